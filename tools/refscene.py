"""Neutral-scene extraction from a *reference* env object (FIXTURE TOOLING).

Turns the state of a reference MiniWorldEnv (running under tools/refshim.py) into the
plain-array "neutral scene" consumed by oracle/pyoracle.py and compared against the
product's own world generation.  Layout documented in oracle/pyoracle.py.
"""
import os

import numpy as np


def _variant_of_path(path):
    """Variant name of a texture file: relative to textures/ without extension (`chars/ch_0x66_1`),
    or `mesh:<name>` for the map_Kd image next to a mesh."""
    path = os.path.normpath(path)
    parts = path.split(os.sep)
    stem = os.path.splitext(parts[-1])[0]
    if "meshes" in parts:
        return "mesh:" + stem
    i = len(parts) - 1 - parts[::-1].index("textures")
    return "/".join(parts[i + 1:-1] + [stem])


def _tex_variant(tex):
    return _variant_of_path(tex.tex.path if hasattr(tex, "tex") else tex.path)


def scene_from_ref_env(env):
    import math
    polys_v, polys_uv, polys_n, polys_nv, polys_tex, polys_rgb, polys_xf = [], [], [], [], [], [], []
    tex_names = []

    def tex_id(tex):
        name = _tex_variant(tex)
        if name not in tex_names:
            tex_names.append(name)
        return tex_names.index(name)

    def add_poly(verts, texcs, normal, tex, rgb=(1, 1, 1), flags=0, xf=(0, 0, 0, 0)):
        n = len(verts)
        if n > 4:
            # GL_POLYGON with n vertices: the driver's triangles are (i, i + 1, 0), i = 1 .. n - 2 (draw_decompose_tmp.h, last
            # vertex convention; a polygon of ours with three vertices (p0, p1, p2) is drawn as (p1, p2, p0))
            for i in range(1, n - 1):
                add_poly([verts[0], verts[i], verts[i + 1]], [texcs[0], texcs[i], texcs[i + 1]], normal, tex, rgb, flags, xf)
            return
        assert n in (3, 4)
        v = np.zeros((4, 3), np.float32)
        uv = np.zeros((4, 2), np.float32)
        v[:n] = np.asarray(verts, np.float64).astype(np.float32)      # glVertex3f
        uv[:n] = np.asarray(texcs, np.float64).astype(np.float32)     # glTexCoord2f
        polys_v.append(v)
        polys_uv.append(uv)
        polys_n.append(np.asarray(normal, np.float64).astype(np.float32))
        polys_nv.append(n | flags)
        polys_tex.append(tex_id(tex) if tex is not None else -1)
        polys_rgb.append(np.asarray(rgb, np.float64).astype(np.float32))
        polys_xf.append(np.asarray(xf, np.float64).astype(np.float32))   # glTranslatef / glRotatef arguments

    for room in env.rooms:                       # Room._render, miniworld.py:401-434
        add_poly(room.floor_verts, room.floor_texcs, (0, 1, 0), room.floor_tex)
        if not room.no_ceiling:
            add_poly(room.ceil_verts, room.ceil_texcs, (0, -1, 0), room.ceil_tex)
        for q in range(room.wall_verts.shape[0] // 4):
            sl = slice(4 * q, 4 * q + 4)
            add_poly(room.wall_verts[sl], room.wall_texcs[sl], room.wall_norms[4 * q], room.wall_tex, flags=0x400)

    kinds, meshes, pos, dirs, sizes, colors, scales, radii, heights, statics = ([] for _ in range(10))
    mesh_names, mesh_tex = [], []
    ents = [e for e in env.entities if e is not env.agent]
    # static ImageFrame / TextFrame quads (entity.py:193-259, 303-383; drawn into display list 1,
    # miniworld.py:1058-1060) in object space, with the arguments of the glTranslatef / glRotatef in front of them
    for e in ents:
        cname = type(e).__name__
        if cname not in ("ImageFrame", "TextFrame"):
            continue
        hz, hy = e.width / 2, e.height / 2
        if cname == "ImageFrame":
            sx, fronts = e.depth, [(e.tex, -hz, +hz)]
        else:
            sx, fronts = 0.05, [(e.texs[i], hz - e.height * (i + 1), hz - e.height * (i + 1) + e.height) for i in range(len(e.str))]
        quads = [([(sx, +hy, z0), (sx, +hy, z1), (sx, -hy, z1), (sx, -hy, z0)], [(1, 1), (0, 1), (0, 0), (1, 0)],
                  (1, 0, 0), (1, 1, 1), t) for t, z0, z1 in fronts]
        uv0, black = [(0, 0)] * 4, (0, 0, 0)
        quads.append(([(0, +hy, -hz), (+sx, +hy, -hz), (+sx, -hy, -hz), (0, -hy, -hz)], uv0, (0, 0, -1), black, None))
        quads.append(([(+sx, +hy, +hz), (0, +hy, +hz), (0, -hy, +hz), (+sx, -hy, +hz)], uv0, (0, 0, 1), black, None))
        quads.append(([(+sx, +hy, +hz), (+sx, +hy, -hz), (0, +hy, -hz), (0, +hy, +hz)], uv0, (0, 1, 0), black, None))
        quads.append(([(+sx, -hy, -hz), (+sx, -hy, +hz), (0, -hy, +hz), (0, -hy, -hz)], uv0, (0, -1, 0), black, None))
        xf = (float(e.pos[0]), float(e.pos[1]), float(e.pos[2]), e.dir * (180 / math.pi))
        for verts, texcs, normal, rgb, tex in quads:
            add_poly(verts, texcs, normal, tex, rgb, 0x100 | 0x200 | 0x400, xf)
    for e in ents:
        cname = type(e).__name__
        pos.append(np.array(e.pos, np.float64))
        dirs.append(float(e.dir))
        radii.append(float(e.radius))
        heights.append(float(e.height))
        statics.append(int(bool(e.is_static)))
        if cname == "Box":
            kinds.append(1); meshes.append(-1)
            sizes.append(np.array(e.size, np.float64))
            colors.append(np.array(e.color_vec, np.float64))
            scales.append(1.0)
        elif hasattr(e, "mesh"):
            mname = mesh_name_of(e)
            if mname not in mesh_names:
                mesh_names.append(mname)
                texs = {(_variant_of_path(t.path) if t else None) for t in e.mesh.textures}
                assert len(texs) == 1, "multi-texture meshes are not used by any environment"
                tv = texs.pop()
                if tv and tv not in tex_names:
                    tex_names.append(tv)
                mesh_tex.append(tex_names.index(tv) if tv else -1)
            kinds.append(2); meshes.append(mesh_names.index(mname))
            sizes.append(np.zeros(3)); colors.append(np.ones(3))
            scales.append(float(e.scale))
        elif cname in ("ImageFrame", "TextFrame"):
            kinds.append(3); meshes.append(-1)
            sizes.append(np.zeros(3)); colors.append(np.ones(3)); scales.append(1.0)
        else:
            raise NotImplementedError(cname)
    E = len(ents)
    carrying = ents.index(env.agent.carrying) if env.agent.carrying is not None else -1
    return {
        "polys_v": np.array(polys_v, np.float32).reshape(-1, 4, 3),
        "polys_uv": np.array(polys_uv, np.float32).reshape(-1, 4, 2),
        "polys_n": np.array(polys_n, np.float32).reshape(-1, 3),
        "polys_nv": np.array(polys_nv, np.int32),
        "polys_tex": np.array(polys_tex, np.int32),
        "polys_rgb": np.array(polys_rgb, np.float32).reshape(-1, 3),
        "polys_xf": np.array(polys_xf, np.float32).reshape(-1, 4),
        "tex_names": np.array(tex_names),
        "ents_kind": np.array(kinds, np.int32),
        "ents_mesh": np.array(meshes, np.int32),
        "ents_pos": np.array(pos, np.float64).reshape(E, 3),
        "ents_dir": np.array(dirs, np.float64),
        "ents_size": np.array(sizes, np.float64).reshape(E, 3),
        "ents_color": np.array(colors, np.float64).reshape(E, 3),
        "ents_scale": np.array(scales, np.float64),
        "ents_radius": np.array(radii, np.float64),
        "ents_height": np.array(heights, np.float64),
        "ents_static": np.array(statics, np.int32),
        "mesh_names": np.array(mesh_names),
        "mesh_tex": np.array(mesh_tex, np.int32),
        "agent_pos": np.array(env.agent.pos, np.float64),
        "agent_dir": np.float64(env.agent.dir),
        "agent_carrying": np.int32(carrying),
        "cam_height": np.float64(env.agent.cam_height),
        "cam_fwd_disp": np.float64(env.agent.cam_fwd_disp),
        "cam_pitch": np.float64(env.agent.cam_pitch),
        "cam_fov_y": np.float64(env.agent.cam_fov_y),
        "sky": np.array(env.sky_color, np.float64),
        "light_pos": np.array(env.light_pos, np.float64),
        "light_color": np.array(env.light_color, np.float64),
        "light_ambient": np.array(env.light_ambient, np.float64),
        "wall_segs": np.ascontiguousarray(np.asarray(env.wall_segs, np.float64)[:, :, [0, 2]]),
        "max_forward_step": np.float64(env.max_forward_step),
        "max_episode_steps": np.float64(env.max_episode_steps),
        "step_count": np.int32(env.step_count),
        "extent": np.array([env.min_x, env.max_x, env.min_z, env.max_z], np.float64),
        "agent_radius": np.float64(env.agent.radius),
        "agent_height": np.float64(env.agent.height),
    }


def mesh_name_of(ent):
    """MeshEnt does not keep its mesh name; recover it from ObjMesh.cache (objmesh.py:17)."""
    from miniworld.objmesh import ObjMesh
    for path, m in ObjMesh.cache.items():
        if m is ent.mesh:
            return os.path.splitext(os.path.basename(path))[0]
    raise KeyError("mesh not in cache")


def ref_mesh_arrays(mesh):
    """Per-face-vertex arrays exactly as the reference's ObjMesh built them (objmesh.py:139-207)."""
    vl = mesh.vlists
    cat = lambda key, k: np.concatenate([v.attrs[key].reshape(-1, 3, k) for v in vl]).astype(np.float32)
    return {"verts": cat("v3f", 3), "norms": cat("n3f", 3), "texcs": cat("t2f", 2), "colors": cat("c3f", 3),
            "min_coords": np.array(mesh.min_coords), "max_coords": np.array(mesh.max_coords)}
