"""World objects -> engine arrays.

``scene_from_env`` flattens a ``MiniWorldEnv`` (rooms, entities, agent, lighting) into the
plain-array "neutral scene" the engine consumes — what the reference would have handed to
OpenGL as display list 1 + entity draws (miniworld.py:401-434, 1019-1077).  ``EngineBinding``
is the batch-of-one engine behind the single-environment API.
"""
from __future__ import annotations

import numpy as np

from . import engine as eng
import math

from .entity import Box, MeshEnt, _Frame
from .texture import Texture


def _f32(a):
    return np.asarray(a, np.float64).astype(np.float32)      # what glVertex3f & co. receive


def scene_from_env(env) -> dict:
    tex_names: list = []

    def tex_id(tex):
        if tex.variant not in tex_names:
            tex_names.append(tex.variant)
        return tex_names.index(tex.variant)

    pv, puv, pn, pnv, ptex, prgb, pxf = [], [], [], [], [], [], []

    def add_poly(verts, texcs, normal, tex, rgb=(1, 1, 1), flags=0, xf=(0, 0, 0, 0)):
        n = len(verts)
        if n > 4:
            # GL_POLYGON with n vertices: the driver's triangles are (i, i + 1, 0), i = 1 .. n - 2 — and a polygon with three
            # vertices (p0, p1, p2) is drawn as (p1, p2, p0): one triangle polygon per fan triangle (tests/golden/gl_ngon_*.npz)
            for i in range(1, n - 1):
                add_poly([verts[0], verts[i], verts[i + 1]], [texcs[0], texcs[i], texcs[i + 1]], normal, tex, rgb, flags, xf)
            return
        v = np.zeros((4, 3), np.float32)
        uv = np.zeros((4, 2), np.float32)
        v[:n], uv[:n] = _f32(verts), _f32(texcs)
        pv.append(v); puv.append(uv); pn.append(_f32(normal)); pnv.append(n | flags)
        ptex.append(tex_id(tex) if tex is not None else -1); prgb.append(_f32(rgb)); pxf.append(_f32(xf))

    for room in env.rooms:       # draw order of Room._render: floor, ceiling, walls
        add_poly(room.floor_verts, room.floor_texcs, (0, 1, 0), room.floor_tex)
        if not room.no_ceiling:
            add_poly(room.ceil_verts, room.ceil_texcs, (0, -1, 0), room.ceil_tex)
        for q in range(room.wall_verts.shape[0] // 4):
            sl = slice(4 * q, 4 * q + 4)
            add_poly(room.wall_verts[sl], room.wall_texcs[sl], room.wall_norms[4 * q], room.wall_tex, flags=eng.POLY_QUAD)

    ents = [e for e in env.entities if e is not env.agent]
    E = len(ents)
    # static ImageFrame / TextFrame quads are part of display list 1 (miniworld.py:1058-1060): appended to the polygon
    # list in OBJECT space together with the arguments of the glTranslatef / glRotatef in front of them
    # (entity.py:205-207, 318-320): the engine composes the modelview the way the GL matrix stack does
    for e in ents:
        if isinstance(e, _Frame):
            xf = (float(e.pos[0]), float(e.pos[1]), float(e.pos[2]), e.dir * (180 / math.pi))
            for verts, texcs, normal, rgb, tex in e.quads():
                add_poly(verts, texcs, normal, tex, rgb, eng.POLY_ENTITY | eng.POLY_XF | eng.POLY_QUAD, xf)
    mesh_names: list = []
    mesh_tex: list = []
    kind = np.zeros(E, np.int32)
    mesh = np.full(E, -1, np.int32)
    size = np.zeros((E, 3)); color = np.ones((E, 3)); scale = np.ones(E)
    for i, e in enumerate(ents):
        if isinstance(e, Box):
            kind[i] = eng.ENT_BOX
            size[i] = e.size
            color[i] = e.color_vec
        elif isinstance(e, MeshEnt):
            kind[i] = eng.ENT_MESH
            if e.mesh_name not in mesh_names:
                mesh_names.append(e.mesh_name)
                tv = e.mesh.tex_variant
                mesh_tex.append(tex_id(Texture.load(tv)) if tv else -1)
            mesh[i] = mesh_names.index(e.mesh_name)
            scale[i] = float(e.scale)
        elif isinstance(e, _Frame):
            kind[i] = eng.ENT_FRAME          # drawn through the polygon list; entity for physics / queries
        else:
            raise NotImplementedError(f"entity type {type(e).__name__} has no engine representation yet")
    carrying = ents.index(env.agent.carrying) if env.agent.carrying is not None else -1
    segs = np.asarray(env.wall_segs, np.float64)
    return {
        "polys_v": np.array(pv, np.float32).reshape(-1, 4, 3),
        "polys_uv": np.array(puv, np.float32).reshape(-1, 4, 2),
        "polys_n": np.array(pn, np.float32).reshape(-1, 3),
        "polys_nv": np.array(pnv, np.int32),
        "polys_tex": np.array(ptex, np.int32),
        "polys_rgb": np.array(prgb, np.float32).reshape(-1, 3),
        "polys_xf": np.array(pxf, np.float32).reshape(-1, 4),
        "tex_names": np.array(tex_names),
        "ents_kind": kind,
        "ents_mesh": mesh,
        "ents_pos": np.array([e.pos for e in ents], np.float64).reshape(E, 3),
        "ents_dir": np.array([e.dir for e in ents], np.float64).reshape(E),
        "ents_size": size,
        "ents_color": color,
        "ents_scale": scale,
        "ents_radius": np.array([float(e.radius) for e in ents], np.float64).reshape(E),
        "ents_height": np.array([float(e.height) for e in ents], np.float64).reshape(E),
        "ents_static": np.array([int(bool(e.is_static)) for e in ents], np.int32).reshape(E),
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
        "wall_segs": np.ascontiguousarray(segs[:, :, [0, 2]]) if len(segs) else np.zeros((0, 2, 2)),
        "max_forward_step": np.float64(env.max_forward_step),
        "max_episode_steps": np.float64(env.max_episode_steps),
        "step_count": np.int32(env.step_count),
        "extent": np.array([env.min_x, env.max_x, env.min_z, env.max_z], np.float64),
        "agent_radius": np.float64(env.agent.radius),
        "agent_height": np.float64(env.agent.height),
    }


def upload_scene_meshes(engine, scene: dict, mesh_ids: dict, tex_ids: dict | None = None) -> dict:
    """Make sure every mesh named by the scene is resident; returns scene-index -> engine mesh id.
    tex_ids: texture variant -> engine texture id (a textured mesh uploads its map_Kd image on demand)."""
    from .objmesh import ObjMesh
    from .texture import Texture
    out = {}
    for i, name in enumerate([str(m) for m in scene.get("mesh_names", [])]):
        if name not in mesh_ids:
            mesh_ids[name] = len(mesh_ids)
            m = ObjMesh.get(name)
            tid = -1
            if m.tex_variant:
                if tex_ids is None:
                    raise ValueError(f"mesh {name!r} is textured: upload_scene_meshes needs the texture id table")
                if m.tex_variant not in tex_ids:
                    tex_ids[m.tex_variant] = len(tex_ids)
                    engine.upload_texture(tex_ids[m.tex_variant], Texture.load(m.tex_variant).rgb_bottom_up())
                tid = tex_ids[m.tex_variant]
            engine.upload_mesh(mesh_ids[name], m.verts, m.norms, m.texcs, m.colors, tid)
        out[i] = mesh_ids[name]
    return out


def polys_array(scene: dict, tex_map=None) -> np.ndarray:
    P = len(scene["polys_nv"])
    polys = np.zeros(P, eng.POLY_DTYPE)
    polys["v"], polys["uv"], polys["n"] = scene["polys_v"], scene["polys_uv"], scene["polys_n"]
    polys["nv"] = scene["polys_nv"]
    polys["rgb"] = scene["polys_rgb"] if "polys_rgb" in scene else 1.0
    if "polys_xf" in scene:
        polys["xf"] = scene["polys_xf"]
    polys["tex"] = scene["polys_tex"] if tex_map is None else [(tex_map[int(t)] if t >= 0 else -1) for t in scene["polys_tex"]]
    return polys


def state_arrays(scenes: list, E: int, mesh_maps: list | None = None) -> dict:
    """mw_set_state arrays for a list of neutral scenes (entity tables padded to E slots).
    mesh_maps[i] translates scene i's mesh indices into engine mesh ids."""
    n = len(scenes)
    st = {
        "agent_pos": np.array([s["agent_pos"] for s in scenes], np.float64),
        "agent_dir": np.array([s["agent_dir"] for s in scenes], np.float64),
        "cam": np.array([[s["cam_height"], s["cam_fwd_disp"], s["cam_pitch"], s["cam_fov_y"]] for s in scenes], np.float64),
        "light": np.array([np.concatenate([s["sky"], s["light_pos"], s["light_color"], s["light_ambient"]])
                           for s in scenes], np.float64),
        "carrying": np.array([int(s.get("agent_carrying", -1)) for s in scenes], np.int32),
        "step_count": np.array([int(s.get("step_count", 0)) for s in scenes], np.int32),
        "num_picked_up": np.zeros(n, np.int32),
        "ent_kind": np.zeros((n, E), np.int32),
        "ent_mesh": np.full((n, E), -1, np.int32),
        "ent_static": np.zeros((n, E), np.int32),
        "ent_pos": np.zeros((n, E, 3), np.float64),
        "ent_dir": np.zeros((n, E), np.float64),
        "ent_geom": np.zeros((n, E, 9), np.float64),
        "extent": np.array([s.get("extent", np.zeros(4)) for s in scenes], np.float64),
    }
    for i, s in enumerate(scenes):
        k = len(s["ents_kind"])
        if k == 0:
            continue
        st["ent_kind"][i, :k] = s["ents_kind"]
        mm = mesh_maps[i] if mesh_maps else None
        st["ent_mesh"][i, :k] = [(-1 if m < 0 else (mm[int(m)] if mm else int(m))) for m in s["ents_mesh"]]
        st["ent_static"][i, :k] = s["ents_static"]
        st["ent_pos"][i, :k] = s["ents_pos"]
        st["ent_dir"][i, :k] = s["ents_dir"]
        st["ent_geom"][i, :k, 0:3] = s["ents_size"]
        st["ent_geom"][i, :k, 3:6] = s["ents_color"]
        st["ent_geom"][i, :k, 6] = s["ents_scale"]
        st["ent_geom"][i, :k, 7] = s["ents_radius"]
        st["ent_geom"][i, :k, 8] = s["ents_height"]
    return st


def base_config(num_envs, width, height, max_ents, max_polys, max_segs, max_visible, params_ranges=None,
                device_id=0) -> eng.MwConfig:
    cfg = eng.MwConfig()
    cfg.device_id = device_id
    cfg.num_envs = num_envs
    cfg.obs_width, cfg.obs_height, cfg.msaa = width, height, 8
    cfg.max_ents, cfg.max_polys, cfg.max_segs, cfg.max_visible = max_ents, max_polys, max_segs, max_visible
    cfg.shared_geometry = 1
    cfg.task = eng.TASK_NONE
    cfg.max_episode_steps = 1 << 30
    cfg.generator = eng.GEN_NONE
    cfg.autoreset = eng.AUTORESET_OFF
    cfg.agent_radius = 0.4
    cfg.agent_height = 1.6
    r = eng.default_ranges()
    if params_ranges:
        r.update({k: v for k, v in params_ranges.items() if k in r})
    eng.fill_ranges(cfg, r)
    return cfg


class EngineBinding:
    """Batch-of-one engine behind ``MiniWorldEnv``: host objects are the source of truth,
    every step / frame is computed by the HIP kernels."""

    def __init__(self, env):
        self.engine = None
        self.caps = (0, 0, 0)
        self.tex_ids: dict = {}
        self.mesh_ids: dict = {}
        self._torch = __import__("torch")

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None

    def _ensure_engine(self, env, scene):
        need = (len(scene["polys_nv"]), len(scene["wall_segs"]), len(scene["ents_kind"]))
        if self.engine is not None and all(n <= c for n, c in zip(need, self.caps)):
            return
        self.close()
        caps = (max(16, need[0]), max(8, need[1]), max(8, need[2] + 2))
        cfg = base_config(1, env.obs_width, env.obs_height, caps[2], caps[0], caps[1],
                          max_visible=-(-(caps[0] + 6 * caps[2]) // 16) * 16, params_ranges=env.params.as_ranges(),
                          device_id=env.device_id)
        cfg.agent_radius = float(env.agent.radius)
        self.engine = eng.Engine(cfg)
        self.caps = caps
        self.tex_ids, self.mesh_ids = {}, {}
        t = self._torch
        dev = self.engine.device
        self.obs = t.zeros((1, env.obs_height, env.obs_width, 3), dtype=t.uint8, device=dev)
        self.depth = t.zeros((1, env.obs_height, env.obs_width, 1), dtype=t.float32, device=dev)
        self.act = t.zeros(1, dtype=t.int32, device=dev)
        self.rew = t.zeros(1, dtype=t.float32, device=dev)
        self.flags = t.zeros(2, dtype=t.uint8, device=dev)

    def upload_world(self, env):
        from .texture import Texture
        scene = scene_from_env(env)
        self._ensure_engine(env, scene)
        tex_map = {}
        for i, variant in enumerate([str(v) for v in scene["tex_names"]]):
            if variant not in self.tex_ids:
                self.tex_ids[variant] = len(self.tex_ids)
                self.engine.upload_texture(self.tex_ids[variant], Texture.tex_cache[variant].rgb_bottom_up())
            tex_map[i] = self.tex_ids[variant]
        self.engine.set_geometry(-1, polys_array(scene, tex_map), scene["wall_segs"])
        self.push_state(env, scene)

    def push_state(self, env, scene=None):
        scene = scene_from_env(env) if scene is None else scene
        mm = upload_scene_meshes(self.engine, scene, self.mesh_ids, self.tex_ids)
        self.engine.set_state(state_arrays([scene], self.engine.E, [mm]))
        self._ents = [e for e in env.entities if e is not env.agent]

    def _pull_state(self, env):
        st = self.engine.get_state()
        env.agent.pos = st["agent_pos"][0].copy()
        env.agent.dir = float(st["agent_dir"][0])
        for i, e in enumerate(self._ents):
            e.pos = st["ent_pos"][0, i].copy()
            e.dir = float(st["ent_dir"][0, i])
        c = int(st["carrying"][0])
        env.agent.carrying = self._ents[c] if c >= 0 else None

    def step(self, env, action, fwd_step, fwd_drift, turn_step):
        self.push_state(env)
        self.engine.set_step_params(np.array([[fwd_step, fwd_drift, turn_step]], np.float64))
        self.act[0] = action
        self.engine.step(self.act, self.obs, None, self.rew, self.flags[0:1], self.flags[1:2])
        self._pull_state(env)
        return self.obs[0].cpu().numpy()

    def visible_ents(self, env):
        self.push_state(env)
        vis = self.engine.visible_ents(0, 1)[0].cpu().numpy()
        return {e for i, e in enumerate(self._ents) if vis[i]}

    def render(self, env, want_depth=False, top_view=False, render_agent=True):
        self.push_state(env)
        if top_view:
            self.engine.render_top(self.obs, self.depth if want_depth else None, render_agent)
        else:
            self.engine.render(self.obs, self.depth if want_depth else None)
        out = {"rgb": self.obs[0].cpu().numpy()}
        if want_depth:
            out["depth"] = self.depth[0].cpu().numpy()
        return out
