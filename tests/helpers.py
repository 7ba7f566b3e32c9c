"""Shared test helpers: golden fixtures -> neutral scenes -> engine state."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    s0 = {k[3:]: d[k] for k in d.files if k.startswith("s0/")}
    tr = {k[3:]: d[k] for k in d.files if k.startswith("tr/")}
    meta = {k[5:]: d[k] for k in d.files if k.startswith("meta/")}
    obs = {}
    for k in d.files:
        if k.startswith("obs/"):
            _, fr, key = k.split("/")
            obs.setdefault(int(fr), {})[key] = d[k]
    return s0, tr, meta, obs


def golden_meshes(scene):
    """mesh name -> arrays dict for oracle.pyoracle.render, from tests/golden/meshes.npz."""
    d = np.load(os.path.join(GOLDEN, "meshes.npz"))
    out = {}
    for name in [str(m) for m in scene.get("mesh_names", [])]:
        base = name.split("_")[0]
        base = base if f"{base}/verts" in d.files else name        # ball / key geometry is stored once per shape
        m = {k: d[f"{base}/{k}"] for k in ("verts", "norms", "texcs")}
        kd = d["kd:" + name] if ("kd:" + name) in d.files else np.ones(3)
        m["colors"] = np.broadcast_to(kd.astype(np.float32), m["verts"].shape).copy()
        out[name] = m
    return out


def frame_scene(s0, frame):
    """Scene s0 with the agent/entity state of a stored frame substituted."""
    sc = dict(s0)
    for k in ("agent_pos", "agent_dir", "ents_pos", "ents_dir", "ents_kind"):
        sc[k] = frame[k]
    return sc


def task_of(meta):
    if rule_of(meta) != "engine":
        return 0
    return {"PickupObjects": 2, "PutNext": 3, "RoomObjects": 0, "ThreeRooms": 0}.get(str(meta["env"]), 1)


def rule_of(meta):
    """"engine": the env's reward / termination rule is one of K1's task rules; "host": it is Python on top
    of the engine's physics (C-level tests compare poses only); "api_only": the host also moves entities."""
    return str(meta["rule"]) if "rule" in meta else "engine"


def env_kwargs_of(meta):
    import ast
    kw = ast.literal_eval(str(meta["kwargs"])) if "kwargs" in meta else {}
    if bool(meta["domain_rand"]):
        kw["domain_rand"] = True
    return kw


def goals_of(meta):
    return int(meta.get("goal_ent", 0)), int(meta.get("goal_ent2", -1))


# ------------------------------------------------------------------ engine glue (GPU tests)

def make_engine_for_scene(s0, n_envs, task=1, max_episode_steps=None, domain_rand=False, max_visible=None,
                          goal_ent=0, goal_ent2=-1, agent_radius=0.4, msaa=8, width=80, height=60):
    """Engine with the scene's shared geometry / textures uploaded (state not yet set)."""
    from miniworld_amd import engine as eng
    from miniworld_amd import assets
    cfg = eng.MwConfig()
    E = max(1, len(s0["ents_kind"]))
    P, S = len(s0["polys_nv"]), len(s0["wall_segs"])
    cfg.device_id = 0
    cfg.num_envs = n_envs
    cfg.obs_width, cfg.obs_height, cfg.msaa = width, height, msaa
    cfg.max_ents, cfg.max_polys, cfg.max_segs = E, P, S
    cfg.max_visible = max_visible or -(-(P + 6 * E) // 16) * 16
    cfg.shared_geometry = 1
    cfg.task = task
    cfg.goal_ent, cfg.goal_ent2 = goal_ent, goal_ent2
    cfg.num_objs = len(s0["ents_kind"])
    mes = max_episode_steps if max_episode_steps is not None else s0["max_episode_steps"]
    cfg.max_episode_steps = int(min(float(mes), 2 ** 30))
    cfg.domain_rand = int(domain_rand)
    cfg.generator = eng.GEN_NONE
    cfg.autoreset = eng.AUTORESET_OFF
    cfg.agent_radius = agent_radius
    eng.fill_ranges(cfg)
    cfg.max_forward_step = float(s0["max_forward_step"])
    e = eng.Engine(cfg)
    for i, name in enumerate([str(t) for t in s0["tex_names"]]):
        e.upload_texture(i, assets.texture_rgb_bottom_up(name))
    polys = np.zeros(P, eng.POLY_DTYPE)
    polys["v"], polys["uv"], polys["n"] = s0["polys_v"], s0["polys_uv"], s0["polys_n"]
    polys["nv"], polys["tex"] = s0["polys_nv"], s0["polys_tex"]
    polys["rgb"] = s0["polys_rgb"] if "polys_rgb" in s0 else 1.0
    if "polys_xf" in s0:
        polys["xf"] = s0["polys_xf"]
    e.set_geometry(-1, polys, s0["wall_segs"])
    from miniworld_amd.scene import upload_scene_meshes
    tex_ids = {str(t): i for i, t in enumerate(s0["tex_names"])}
    e._test_mesh_map = upload_scene_meshes(e, s0, {}, tex_ids)
    return e


def scene_state_arrays(scenes, E=None):
    """Stack neutral scenes (same entity table layout) into mw_set_state arrays."""
    n = len(scenes)
    E = max(1, len(scenes[0]["ents_kind"])) if E is None else E
    Es = len(scenes[0]["ents_kind"])
    st = {
        "agent_pos": np.array([s["agent_pos"] for s in scenes], np.float64),
        "agent_dir": np.array([s["agent_dir"] for s in scenes], np.float64),
        "cam": np.array([[s["cam_height"], s["cam_fwd_disp"], s["cam_pitch"], s["cam_fov_y"]] for s in scenes], np.float64),
        "light": np.array([np.concatenate([s["sky"], s["light_pos"], s["light_color"], s["light_ambient"]]) for s in scenes], np.float64),
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
        if Es == 0:
            continue
        st["ent_kind"][i, :Es] = s["ents_kind"]
        st["ent_mesh"][i, :Es] = s["ents_mesh"]       # scene mesh index == engine mesh id (upload order)
        st["ent_static"][i, :Es] = s["ents_static"]
        st["ent_pos"][i, :Es] = s["ents_pos"]
        st["ent_dir"][i, :Es] = s["ents_dir"]
        st["ent_geom"][i, :Es, 0:3] = s["ents_size"]
        st["ent_geom"][i, :Es, 3:6] = s["ents_color"]
        st["ent_geom"][i, :Es, 6] = s["ents_scale"]
        st["ent_geom"][i, :Es, 7] = s["ents_radius"]
        st["ent_geom"][i, :Es, 8] = s["ents_height"]
    return st


def depth_from_z16(z16):
    """FrameBuffer.get_depth_map's own numpy expression (opengl.py:426-431) on a u16 map."""
    z_near, z_far = 0.04, 100.0
    depth_map = z16.astype(np.float32) / 65535
    clip_z = (depth_map - 0.5) * 2.0
    world_z = -2 * z_far * z_near / (clip_z * (z_far - z_near) - (z_far + z_near))
    return world_z.astype(np.float32)


# ------------------------------------------------------------------ batched env -> oracle scenes

def vec_env_meshes(vec):
    """mesh name -> arrays for pyoracle.render, for every mesh resident in the batched env's engine."""
    from miniworld_amd.objmesh import ObjMesh
    out = {}
    for name in vec.mesh_ids:
        m = ObjMesh.get(name)
        out[name] = {"verts": m.verts, "norms": m.norms, "texcs": m.texcs, "colors": m.colors}
    return out


def scene_of_vec_env(vec, st, i, row=None):
    """Neutral scene (oracle input) of env i of a MiniWorldVecEnv, from what the DEVICE holds: the state arrays `st`
    (mw_get_state; `row` = env i's row in them, default i) and, for envs with their own geometry set (Maze, texture
    domain randomisation), the polygons read back with mw_get_geometry."""
    from miniworld_amd.scene import scene_from_env
    row = i if row is None else row
    sc = scene_from_env(vec.template)
    names = sorted(vec.mesh_ids, key=vec.mesh_ids.get)
    sc["agent_pos"], sc["agent_dir"] = st["agent_pos"][row], st["agent_dir"][row]
    sc["cam_height"], sc["cam_fwd_disp"], sc["cam_pitch"], sc["cam_fov_y"] = st["cam"][row]
    sc["sky"], sc["light_pos"] = st["light"][row, 0:3], st["light"][row, 3:6]
    sc["light_color"], sc["light_ambient"] = st["light"][row, 6:9], st["light"][row, 9:12]
    sc["ents_kind"], sc["ents_mesh"] = st["ent_kind"][row], st["ent_mesh"][row]
    sc["ents_pos"], sc["ents_dir"] = st["ent_pos"][row], st["ent_dir"][row]
    sc["ents_size"], sc["ents_color"] = st["ent_geom"][row, :, 0:3], st["ent_geom"][row, :, 3:6]
    sc["ents_scale"], sc["ents_radius"], sc["ents_height"] = st["ent_geom"][row, :, 6], st["ent_geom"][row, :, 7], st["ent_geom"][row, :, 8]
    sc["ents_static"] = st["ent_static"][row]
    sc["mesh_names"] = np.array(names)
    sc["mesh_tex"] = np.full(len(names), -1, np.int32)         # ball / key meshes are untextured
    if not vec.engine.cfg.shared_geometry:
        polys, segs = vec.engine.get_geometry(i)
        sc["polys_v"], sc["polys_uv"], sc["polys_n"] = polys["v"], polys["uv"], polys["n"]
        sc["polys_nv"], sc["polys_tex"], sc["polys_rgb"] = polys["nv"], polys["tex"], polys["rgb"]
        sc["polys_xf"] = polys["xf"]
        sc["wall_segs"] = segs
        ids = vec.tex_ids
        sc["tex_names"] = np.array(sorted(ids, key=ids.get))
    return sc


class EpisodeMirror:
    """One environment of a device batch replayed on the CPU from the same seed, with nothing of the engine in it:
    worlds come from the host world generator (seed-exact with the reference's reset(), tests/test_host_logic_cpu.py),
    steps from the C oracle's dynamics (pyoracle.Dynamics, pinned on the reference's trajectories), the three per-step
    domain-randomisation draws from the env's own numpy stream in the reference's order (miniworld.py:677-680), and an
    episode's end continues that stream with reset() exactly like the device's same-step auto-reset."""

    def __init__(self, cls, seed, domain_rand, task, **env_kwargs):
        self.h = cls(host_only=True, domain_rand=domain_rand, **env_kwargs)
        self.h.reset(seed=int(seed))
        self.dr, self.task = domain_rand, task
        self.episodes = 0
        self._new_episode()

    def _new_episode(self):
        import pyoracle
        from miniworld_amd.scene import scene_from_env
        h = self.h
        self.sc = scene_from_env(h)
        self.dyn = pyoracle.Dynamics(self.sc, self.task, int(min(float(h.max_episode_steps), 2 ** 30)),
                                     num_objs=len(self.sc["ents_kind"]), max_forward_step=float(h.max_forward_step),
                                     agent_radius=float(h.agent.radius))
        self.fresh = True
        self.episodes += 1

    def step(self, action):
        h = self.h
        rand = h.np_random if self.dr else None
        p = [h.params.sample(rand, k) for k in ("forward_step", "forward_drift", "turn_step")]
        r, te, tr = self.dyn.step(int(action), *p)
        self.fresh = False
        if te or tr:
            h.reset()
            self._new_episode()
        return r, te, tr

    def state(self):
        """(agent_pos, agent_dir, carrying, step_count, alive[E], ent_pos[E,3], ent_dir[E]) after the last step."""
        E = len(self.sc["ents_kind"])
        ag, ents = self.dyn.ag, self.dyn.ents
        return (np.array(ag.pos[:]), float(ag.dir), int(ag.carrying), int(ag.step_count),
                np.array([bool(ents[k].alive) for k in range(E)]),
                np.array([ents[k].pos[:] for k in range(E)]).reshape(E, 3), np.array([ents[k].dir for k in range(E)]))

    def frame_scene(self):
        """The scene the observation returned by the last step shows: a fresh episode's initial world, or the state
        at render time — rendering happens before PickupObjects removes what was picked up (pickupobjects.py:86-88)."""
        sc = dict(self.sc)
        if not self.fresh:
            E = len(sc["ents_kind"])
            ag, ents = self.dyn.ag, self.dyn.render_ents
            sc["agent_pos"], sc["agent_dir"] = np.array(ag.pos[:]), np.float64(ag.dir)
            sc["ents_pos"] = np.array([ents[k].pos[:] for k in range(E)]).reshape(E, 3)
            sc["ents_dir"] = np.array([ents[k].dir for k in range(E)])
            sc["ents_kind"] = np.where([bool(ents[k].alive) for k in range(E)], sc["ents_kind"], 0).astype(np.int32)
        return sc

    def meshes(self):
        from miniworld_amd.objmesh import ObjMesh
        out = {}
        for name in [str(m) for m in self.sc["mesh_names"]]:
            m = ObjMesh.get(name)
            out[name] = {"verts": m.verts, "norms": m.norms, "texcs": m.texcs, "colors": m.colors}
        return out
