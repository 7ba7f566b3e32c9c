"""Placement programs: the ``_gen_world`` of the fixed-floorplan env families compiled into the table the device
generator executes (include/mwengine.h: ``mw_gen_program``, MW_GEN_PROGRAM).

The floorplan of these families never changes; their ``_gen_world`` (fourrooms.py:46-73, tmaze.py:54-81,
ymaze.py:56-108, wallgap.py:48-77, threerooms.py:47-73, putnext.py:45-65, roomobjects.py:44-80, sidewalk.py:51-91,
sign.py:101-150) only draws a few numbers and places entities.  ``compile_program(template, ...)`` takes one
host-generated template world (rooms, entity objects, textures) and the family's op list below and produces

* the rooms (outline, inward normals, extents, the cumulative probabilities numpy's ``choice(p=...)`` searches),
* the texture names with their variants (ids, TEX_DENSITY / size),
* the template entity table and the ops (draws / placements in the reference's call order),
* the template polygons with the metre coordinates their texture coordinates come from, so that a texture-variant
  draw can re-emit them on the device (checked here: metres * density reproduces the template's own texcoords),
* the rule tables of Sidewalk / Sign.

Env i of a batch seeded with s is then the reference's ``reset(seed=s + i)``, generated on the GPU.
"""
from __future__ import annotations

import math

import numpy as np

from . import assets
from . import engine as eng
from .entity import COLOR_NAMES, COLORS, Box, MeshEnt, _Frame

TEX_DENSITY = 512


def _place(slot, room=-1, cond=-1, dir_mode=0, dir=0.0, min_x=None, max_x=None, min_z=None, max_z=None):  # noqa: A002
    flags = (1 if min_x is not None else 0) | (2 if max_x is not None else 0) | (4 if min_z is not None else 0) | (8 if max_z is not None else 0)
    return dict(op=eng.OP_PLACE, slot=slot, room=room, cond=cond, dir_mode=dir_mode, dir=dir, flags=flags,
                lx=min_x or 0.0, hx=max_x or 0.0, lz=min_z or 0.0, hz=max_z or 0.0)


def _fixed(slot, pos, dir=None):  # noqa: A002
    return dict(op=eng.OP_FIXED, slot=slot, lx=float(pos[0]), a=float(pos[1]), lz=float(pos[2]),
                dir_mode=0 if dir is None else 1, dir=0.0 if dir is None else float(dir))


def family_ops(t, slot_of, room_of):
    """The op list of template env ``t`` (a host env right after reset): what its _gen_world does, in call order.
    slot_of(entity) / room_of(room) translate the template's objects; the agent is slot -1."""
    name = type(t).__name__
    A = -1
    ents = [e for e in t.entities if e is not t.agent]
    if name == "FourRooms":
        return [_place(slot_of(t.box)), _place(A)]
    if name in ("TMaze", "TMazeLeft", "TMazeRight", "YMaze", "YMazeLeft", "YMazeRight"):
        ymaze = name.startswith("YMaze")
        stem = t.rooms[0]
        if t.goal_pos is not None:
            g = t.goal_pos
            ops = [_place(slot_of(t.box), min_x=g[0], max_x=g[0], min_z=g[2], max_z=g[2])]
        elif ymaze:
            left, right = t.rooms[2], t.rooms[3]
            ops = [dict(op=eng.OP_COIN, slot=2),
                   _place(slot_of(t.box), room=room_of(left), cond=0, max_z=left.min_z + 2.5),
                   _place(slot_of(t.box), room=room_of(right), cond=1, min_z=right.max_z - 2.5)]
        else:
            bar = t.rooms[1]
            ops = [dict(op=eng.OP_COIN, slot=2),
                   _place(slot_of(t.box), room=room_of(bar), cond=0, max_z=bar.min_z + 2),
                   _place(slot_of(t.box), room=room_of(bar), cond=1, min_z=bar.max_z - 2)]
        return ops + [dict(op=eng.OP_DRAW_DIR, dir=math.pi / 4), _place(A, room=room_of(stem), dir_mode=2)]
    if name == "WallGap":
        building = next(e for e in ents if isinstance(e, MeshEnt))
        return [_place(slot_of(t.box), room=1), _fixed(slot_of(building), building.pos, dir=building.dir), _place(A, room=0)]
    if name == "ThreeRooms":
        ops = []
        for e in ents:      # list order = call order: box, box, picture, duckie, key, ball
            ops.append(dict(op=eng.OP_APPEND, slot=slot_of(e)) if isinstance(e, _Frame) else _place(slot_of(e)))
        return ops + [_place(A)]
    if name == "PutNext":
        ops = []
        for e in ents:
            ops += [dict(op=eng.OP_BOX_SIZE, slot=slot_of(e), a=0.6, b=0.85), _place(slot_of(e))]
        return ops + [_place(A)]
    if name == "Sidewalk":
        ops = []
        for e in ents:
            if isinstance(e, MeshEnt):      # the building (direction given) and the cones (direction drawn)
                ops.append(_fixed(slot_of(e), e.pos, dir=e.dir if e.mesh_name == "building" else None))
        sw = t.rooms[0]
        return ops + [_place(slot_of(t.box), room=0, min_z=sw.max_z - 2, max_z=sw.max_z), _place(A, room=0, min_z=0.0, max_z=1.5)]
    if name == "Sign":
        ops = []
        for e in ents:
            ops.append(dict(op=eng.OP_APPEND, slot=slot_of(e)) if isinstance(e, _Frame) else _fixed(slot_of(e), e.pos))
        return ops + [_place(A, min_x=4.0, max_x=5.0, min_z=4.0, max_z=6.0)]
    if name == "CollectHealth":
        return [_place(slot_of(e)) for e in ents] + [_place(A)]
    raise KeyError(f"no placement program for {name}")


def room_objects_ops(slot_box, slot_ball, slot_key, ball_base, key_base):
    """RoomObjects (roomobjects.py:55-80): a colour choice before each of the three placements."""
    return [dict(op=eng.OP_COLOR, slot=slot_box, room=0), _place(slot_box),
            dict(op=eng.OP_COLOR, slot=slot_ball, room=1, flags=ball_base), _place(slot_ball),
            dict(op=eng.OP_COLOR, slot=slot_key, room=2, flags=key_base), _place(slot_key), _place(-1)]


def _room_polys_metres(room):
    """(surface, metres[4][2]) of the room's polygons in Room._render order (floor, ceiling, wall pieces): the factors
    gen_texcs_floor / gen_texcs_wall (miniworld.py:82-119) multiply by TEX_DENSITY / size."""
    out = []
    fv = room.floor_verts
    out.append((1, [(fv[k][0], fv[k][2]) for k in range(len(fv))]))
    if not room.no_ceiling:
        cv = room.ceil_verts
        out.append((2, [(cv[k][0], cv[k][2]) for k in range(len(cv))]))
    for w in range(room.num_walls):
        p0, p1 = room.outline[w, :], room.outline[(w + 1) % room.num_walls, :]
        width = np.linalg.norm(p1 - p0)
        for start, end, y0, y1 in room._wall_spans(w, width):
            if end == start or y0 == y1:
                continue
            mx, wd, my, ht = start, end - start, y0, y1 - y0
            out.append((0, [(mx, my), (mx, my + ht), (mx + wd, my + ht), (mx + wd, my)]))
    return out


def compile_program(t, scene, tex_ids, mesh_map, ops, task=None):
    """t: template env (domain_rand off); scene: scene_from_env(t); tex_ids: variant name -> engine texture id (every
    variant of every room texture uploaded); mesh_map: scene mesh index -> engine mesh id.  Returns
    (MwGenProgram, polys, poly_room, poly_surf, poly_m, segs)."""
    prog = eng.MwGenProgram()
    rooms = list(t.rooms)
    if len(rooms) > eng.PROG_MAX_ROOMS:
        raise ValueError("too many rooms for a placement program")
    assert t.entities[-1] is t.agent, "the agent is expected to be placed last (Agent.randomize comes last)"
    names = []
    for r in rooms:
        for n in (r.wall_tex_name, r.floor_tex_name, r.ceil_tex_name):
            if n not in names:
                names.append(n)
    if len(names) > eng.PROG_MAX_TEX:
        raise ValueError("too many room textures for a placement program")
    prog.n_tex = len(names)
    for k, n in enumerate(names):
        vs = assets.texture_variants(n)
        prog.tex_nvar[k] = len(vs)
        for j, v in enumerate(vs):
            prog.tex_var_id[k][j] = tex_ids[v] if v in tex_ids else -1
            w, h = assets.texture_size(v)
            prog.tex_var_scale[k][j][0], prog.tex_var_scale[k][j][1] = TEX_DENSITY / w, TEX_DENSITY / h
    # rooms; numpy's Generator.choice(n, p=p): cdf = p.cumsum(); cdf /= cdf[-1]; searchsorted(cdf, u, side="right")
    cdf = np.asarray(t.room_probs, np.float64).cumsum()
    cdf /= cdf[-1]
    prog.n_rooms = len(rooms)
    for i, r in enumerate(rooms):
        pr = prog.rooms[i]
        pr.nverts = r.num_walls
        if r.num_walls > 4:
            raise ValueError("room outlines with more than 4 corners are not supported")
        pr.wall_tex, pr.floor_tex, pr.ceil_tex = (names.index(n) for n in (r.wall_tex_name, r.floor_tex_name, r.ceil_tex_name))
        for k in range(r.num_walls):
            pr.ox[k], pr.oz[k] = float(r.outline[k][0]), float(r.outline[k][2])
            pr.nx[k], pr.nz[k] = float(r.edge_norms[k][0]), float(r.edge_norms[k][2])
        pr.min_x, pr.max_x, pr.min_z, pr.max_z = float(r.min_x), float(r.max_x), float(r.min_z), float(r.max_z)
        pr.cdf = float(cdf[i])
    # template entity table
    E = len(scene["ents_kind"])
    if E > eng.PROG_MAX_ENTS:
        raise ValueError("too many entities for a placement program")
    prog.n_ents = E
    for s in range(E):
        prog.ent_kind[s] = int(scene["ents_kind"][s])
        m = int(scene["ents_mesh"][s])
        prog.ent_mesh[s] = -1 if m < 0 else int(mesh_map[m])
        prog.ent_static[s] = int(scene["ents_static"][s])
        for k in range(3):
            prog.ent_pos[s][k] = float(scene["ents_pos"][s][k])
            prog.ent_geom[s][k] = float(scene["ents_size"][s][k])
            prog.ent_geom[s][3 + k] = float(scene["ents_color"][s][k])
        prog.ent_dir[s] = float(scene["ents_dir"][s])
        prog.ent_geom[s][6] = float(scene["ents_scale"][s])
        prog.ent_geom[s][7] = float(scene["ents_radius"][s])
        prog.ent_geom[s][8] = float(scene["ents_height"][s])
    for c, cname in enumerate(COLOR_NAMES):
        for k in range(3):
            prog.colors[c][k] = float(COLORS[cname][k])
    for k, v in enumerate((t.min_x, t.max_x, t.min_z, t.max_z)):
        prog.extent[k] = float(v)
    if len(ops) > eng.PROG_MAX_OPS:
        raise ValueError("placement program too long")
    prog.n_ops = len(ops)
    for i, o in enumerate(ops):
        po = prog.ops[i]
        po.op, po.slot, po.room, po.cond = o["op"], o.get("slot", 0), o.get("room", -1), o.get("cond", -1)
        po.dir_mode, po.flags = o.get("dir_mode", 0), o.get("flags", 0)
        po.lx, po.hx, po.lz, po.hz = (float(o.get(k, 0.0)) for k in ("lx", "hx", "lz", "hz"))
        po.dir, po.a, po.b = float(o.get("dir", 0.0)), float(o.get("a", 0.0)), float(o.get("b", 0.0))
    # rule tables
    ents = [e for e in t.entities if e is not t.agent]
    if type(t).__name__ == "Sidewalk":
        r = t.street
        for k, v in enumerate((r.min_x, r.max_x, r.min_z, r.max_z)):
            prog.street[k] = float(v)
    if type(t).__name__ == "Sign":
        k = 0
        for obj_index, pair in enumerate(t._objects):
            for color_index, obj in enumerate(pair):
                prog.sign_slot[k] = ents.index(obj)
                prog.sign_reward[k] = float(color_index == t._color_index and obj_index == t._goal) * 2 - 1
                k += 1
        prog.sign_n = k
    # template geometry with the metre coordinates of the room polygons
    from .scene import polys_array
    tex_map = {k: tex_ids[str(v)] for k, v in enumerate(scene["tex_names"])}
    polys = polys_array(scene, tex_map)
    P = len(polys)
    poly_room, poly_surf, poly_m = np.full(P, -1, np.int32), np.zeros(P, np.int32), np.zeros((P, 4, 2), np.float64)
    p = 0
    for i, r in enumerate(rooms):
        for surf, metres in _room_polys_metres(r):
            poly_room[p], poly_surf[p] = i, surf
            tname = (r.wall_tex_name, r.floor_tex_name, r.ceil_tex_name)[surf]
            tex = (r.wall_tex, r.floor_tex, r.ceil_tex)[surf]
            ku, kv = TEX_DENSITY / tex.width, TEX_DENSITY / tex.height
            for k, (mu, mv) in enumerate(metres):
                poly_m[p, k] = (mu, mv)
                want = polys["uv"][p][k]
                got = (np.float32(mu * ku), np.float32(mv * kv))
                if not (got[0] == want[0] and got[1] == want[1]):
                    raise AssertionError(f"room {i} polygon {p} ({tname}): metres do not reproduce the texcoords {want} vs {got}")
            p += 1
    # the remaining polygons are quads of static frames: copied as they are
    segs = np.asarray(scene["wall_segs"], np.float64).reshape(-1, 4)
    return prog, polys, poly_room, poly_surf, poly_m, segs
