"""End-to-end GPU tests of the public API: single-env Gymnasium surface and the batched VecEnv."""
import numpy as np
import pytest

import helpers
from conftest import golden_cases

pytestmark = pytest.mark.gpu

ALL_CASES = golden_cases()


@pytest.mark.parametrize("case", ALL_CASES)
def test_single_env_reset_and_trajectory_match_reference(case):
    """envs.X(...).reset(seed) -> first observation == oracle render of the reference's world;
    then the reference's action sequence reproduces its rewards / flags / poses and the stored
    frames (world generation + HIP step + HIP render, end to end through the Python API)."""
    from miniworld_amd import envs
    s0, tr, meta, obs = helpers.load_case(case)
    env = getattr(envs, str(meta["env"]))(**helpers.env_kwargs_of(meta))
    img = lambda x: x["obs"] if isinstance(x, dict) else x            # Sign returns {"obs", "goal"} (sign.py:170-186)
    o, info = env.reset(seed=int(meta["seed"]))
    assert img(o).shape == (60, 80, 3) and img(o).dtype == np.uint8
    assert np.array_equal(img(o), obs[0]["rgb"])
    poke = meta.get("poke", np.array([-1.0]))
    ents = [e for e in env.entities if e is not env.agent]
    for t in range(len(tr["action"])):
        if int(poke[0]) == t:
            ents[int(poke[1])].pos = np.array(poke[2:5])
        o, r, te, tu, info = env.step(int(tr["action"][t]))
        assert r == tr["reward"][t] and te == bool(tr["term"][t]) and tu == bool(tr["trunc"][t]), (case, t)
        assert np.abs(env.agent.pos - tr["pos"][t]).max() < 1e-12 and abs(env.agent.dir - tr["dir"][t]) < 1e-12
        if (t + 1) in obs:
            assert np.array_equal(img(o), obs[t + 1]["rgb"]), (case, t + 1)
    env.close()


def test_same_seed_same_observation_and_depth():
    from miniworld_amd import envs
    env = envs.OneRoom()
    a, _ = env.reset(seed=11)
    b, _ = env.reset(seed=11)
    assert np.array_equal(a, b)
    d = env.render_depth()
    assert d.shape == (60, 80, 1) and d.dtype == np.float32 and 0.04 < d.min() and d.max() < 100.1
    env.close()


def test_collision_detection_invariant():
    """tests/test_miniworld.py:82-95 of the reference: the agent never leaves the room."""
    from miniworld_amd import envs
    env = envs.OneRoom()
    for _ in range(6):
        env.reset()
        room = env.rooms[0]
        for _ in range(30):
            env.step(env.actions.move_forward)
            x, _, z = env.agent.pos
            assert room.min_x <= x <= room.max_x and room.min_z <= z <= room.max_z
    env.close()


@pytest.mark.parametrize("env_id,n", [("MiniWorld-Hallway-v0", 256), ("MiniWorld-OneRoom-v0", 192)])
def test_vec_env_device_reset_and_autoreset(env_id, n):
    """Device generator + same-step auto-reset: placements are valid (the reference's own
    invariant, test_miniworld.py:112: no intersection after reset), episodes end and restart,
    every env's frame equals the oracle's render of its state."""
    import torch
    import pyoracle
    from miniworld_amd.scene import scene_from_env
    from miniworld_amd.vec_env import MiniWorldVecEnv
    vec = MiniWorldVecEnv(env_id, n, want_depth=True, seed=5)
    vec.reset()
    room = vec.template.rooms[0]
    st = vec.engine.get_state()
    ax, az = st["agent_pos"][:, 0], st["agent_pos"][:, 2]
    bx, bz = st["ent_pos"][:, 0, 0], st["ent_pos"][:, 0, 2]
    assert (ax > room.min_x + 0.4).all() and (ax < room.max_x - 0.4).all()
    assert (az > room.min_z + 0.4).all() and (az < room.max_z - 0.4).all()
    assert (np.hypot(ax - bx, az - bz) >= 0.4 + st["ent_geom"][:, 0, 7]).all()
    assert len(np.unique(ax)) > n // 2                      # different worlds
    if "Hallway" in env_id:
        # place_entity samples x in [min_x - radius, max_x + radius] (miniworld.py:887-890)
        assert (bx >= room.max_x - 2 - st["ent_geom"][:, 0, 7]).all() and (ax <= room.max_x - 2 + 0.4).all()
        assert (np.abs(st["agent_dir"]) <= np.pi / 4).all()
    g = torch.Generator(device="cuda").manual_seed(0)
    n_done = 0
    for t in range(260):
        act = torch.full((n,), 2, dtype=torch.int32, device="cuda") if t % 3 else \
            torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32)
        obs, rew, term, trunc = vec.step(act)
        done = (term | trunc).bool()
        n_done += int(done.sum())
        st2 = vec.engine.get_state()
        # an env that just finished has been regenerated: step counter back to 0
        assert (st2["step_count"][done.cpu().numpy()] == 0).all()
        assert ((rew > 0) == term.bool()).all()
    assert n_done > 0
    vec.engine.check()
    # frame parity for a few envs against the oracle
    st = vec.engine.get_state()
    for i in (0, 1, n // 2, n - 1):
        sc = scene_from_env(vec.template)
        sc["agent_pos"], sc["agent_dir"] = st["agent_pos"][i], st["agent_dir"][i]
        sc["ents_pos"], sc["ents_dir"] = st["ent_pos"][i, :1], st["ent_dir"][i, :1]
        sc["ents_color"] = st["ent_geom"][i, :1, 3:6]
        want = pyoracle.render(sc)
        assert np.array_equal(vec.obs[i].cpu().numpy(), want["rgb"])
        assert np.array_equal(vec.depth[i].cpu().numpy(), want["depth"])
    vec.close()


def test_vec_env_full_size_properties():
    """BASELINE size (4096 envs): determinism (same seed -> identical tensors), frame sanity."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    outs = []
    for _ in range(2):
        vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", 4096, seed=0)
        vec.reset()
        g = torch.Generator(device="cuda").manual_seed(3)
        for _ in range(20):
            vec.step(torch.randint(0, 3, (4096,), generator=g, device="cuda", dtype=torch.int32))
        outs.append((vec.obs.clone(), vec.reward.clone()))
        vec.engine.check()
        vec.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    m = outs[0][0].float().mean(dim=(1, 2, 3))
    assert (m > 20).all() and (m < 235).all()


def _scene_of_env(vec, st, i):
    """Neutral scene of env i of a VecEnv from the engine's state arrays (for the oracle)."""
    from miniworld_amd.scene import scene_from_env
    sc = scene_from_env(vec.template)
    E = st["ent_kind"].shape[1]
    names = sorted(vec.mesh_ids, key=vec.mesh_ids.get)
    sc["agent_pos"], sc["agent_dir"] = st["agent_pos"][i], st["agent_dir"][i]
    sc["cam_height"], sc["cam_fwd_disp"], sc["cam_pitch"], sc["cam_fov_y"] = st["cam"][i]
    sc["sky"], sc["light_pos"] = st["light"][i, 0:3], st["light"][i, 3:6]
    sc["light_color"], sc["light_ambient"] = st["light"][i, 6:9], st["light"][i, 9:12]
    sc["ents_kind"], sc["ents_mesh"] = st["ent_kind"][i], st["ent_mesh"][i]
    sc["ents_pos"], sc["ents_dir"] = st["ent_pos"][i], st["ent_dir"][i]
    sc["ents_size"], sc["ents_color"] = st["ent_geom"][i, :, 0:3], st["ent_geom"][i, :, 3:6]
    sc["ents_scale"], sc["ents_radius"], sc["ents_height"] = st["ent_geom"][i, :, 6], st["ent_geom"][i, :, 7], st["ent_geom"][i, :, 8]
    sc["ents_static"] = st["ent_static"][i]
    sc["mesh_names"] = np.array(names)
    sc["mesh_tex"] = np.full(len(names), -1, np.int32)         # ball / key meshes are untextured
    return sc


def test_vec_env_pickup_device_generator_and_mesh_frames():
    """PickupObjects with domain randomisation, generated and auto-reset on the device: valid
    placements, all three kinds appear, pickups happen, frames (ball / key meshes) equal the oracle."""
    import torch
    import pyoracle
    from miniworld_amd.objmesh import ObjMesh
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 128
    vec = MiniWorldVecEnv("MiniWorld-PickupObjects-v0", n, domain_rand=True, want_depth=True, seed=9)
    vec.reset()
    st = vec.engine.get_state()
    assert (st["ent_kind"][:, :5] != 0).all()
    kinds = {(int(k), int(m) // 6 if k == 2 else -1) for k, m in zip(st["ent_kind"][:, :5].ravel(), st["ent_mesh"][:, :5].ravel())}
    assert kinds == {(1, -1), (2, 0), (2, 1)}            # boxes, balls, keys
    # no two objects (nor the agent) intersect: the reference's own invariant (test_miniworld.py:112)
    for i in range(n):
        p = np.concatenate([st["ent_pos"][i, :5][:, [0, 2]], st["agent_pos"][i][None, [0, 2]]])
        r = np.concatenate([st["ent_geom"][i, :5, 7], [0.4]])
        d = np.hypot(p[:, None, 0] - p[None, :, 0], p[:, None, 1] - p[None, :, 1]) + np.eye(6) * 1e9
        assert (d >= r[:, None] + r[None, :] - 1e-12).all()
        assert (p > r[:, None] - 1e-12).all() and (p < 12 - r[:, None] + 1e-12).all()
    g = torch.Generator(device="cuda").manual_seed(1)
    total_reward = 0.0
    for t in range(300):
        act = torch.randint(0, 5, (n,), generator=g, device="cuda", dtype=torch.int32)
        act[torch.rand(n, generator=g, device="cuda") < 0.5] = 2
        obs, rew, term, trunc = vec.step(act)
        total_reward += float(rew.sum())
    vec.engine.check()
    assert total_reward > 0                                  # something was picked up
    st = vec.engine.get_state()
    meshes = {}
    for name in vec.mesh_ids:
        m = ObjMesh.get(name)
        meshes[name] = {"verts": m.verts, "norms": m.norms, "texcs": m.texcs, "colors": m.colors}
    for i in (0, 17, n - 1):
        want = pyoracle.render(_scene_of_env(vec, st, i), meshes=meshes)
        assert np.array_equal(vec.obs[i].cpu().numpy(), want["rgb"]), f"env {i}"
        assert np.array_equal(vec.depth[i].cpu().numpy(), want["depth"]), f"env {i}"
    vec.close()


def _links_from_device_polys(polys, room_size=3.0, pitch=3.25):
    """Recover (i, j, direction) of every connecting room from a device-generated maze."""
    floors = [k for k in range(len(polys)) if polys["n"][k][1] == 1.0]
    links = []
    for k in floors[64:]:
        P0, P1 = polys["v"][k][0], polys["v"][k][1]
        if P0[2] == P1[2]:
            if P0[0] > P1[0]:
                links.append((round((P1[0] - room_size) / pitch), round(P1[2] / pitch), 0))                  # east
            else:
                links.append((round(P1[0] / pitch), round((P1[2] - room_size) / pitch), 1))                  # west
        elif P0[2] < P1[2]:
            links.append((round(P1[0] / pitch), round(P1[2] / pitch), 2))                                    # north
        else:
            links.append((round((P1[0] - room_size) / pitch), round((P1[2] - room_size) / pitch), 3))        # south
    return links


def test_vec_env_maze_device_generator_matches_host_geometry():
    """Maze generated on the device (recursive backtracker, per-env geometry): a spanning tree of
    the 8x8 grid; every polygon / texcoord / normal / collision segment equals what the host world
    builder (itself seed-exact with the reference) produces for the same carving order; placements
    are collision free; frames equal the oracle; episodes auto-reset with fresh mazes."""
    import torch
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.entity import Box
    from miniworld_amd.scene import scene_from_env
    from miniworld_amd.vec_env import MiniWorldVecEnv
    DI, DJ = (1, -1, 0, 0), (0, 0, -1, 1)

    class FixedMaze(envs.Maze):
        def __init__(self, links):
            self._links = links
            super().__init__(host_only=True)

        def _gen_world(self):
            pitch = self.room_size + self.gap_size
            grid = [[self.add_rect_room(min_x=i * pitch, max_x=i * pitch + self.room_size, min_z=j * pitch,
                                        max_z=j * pitch + self.room_size, wall_tex="brick_wall")
                     for i in range(self.num_cols)] for j in range(self.num_rows)]
            for i, j, d in self._links:
                room, nb = grid[j][i], grid[j + DJ[d]][i + DI[d]]
                if DI[d] == 0:
                    self.connect_rooms(room, nb, min_x=room.min_x, max_x=room.max_x)
                else:
                    self.connect_rooms(room, nb, min_z=room.min_z, max_z=room.max_z)
            self.box = self.place_entity(Box(color="red"))
            self.place_agent()

    n = 16
    vec = MiniWorldVecEnv("MiniWorld-Maze-v0", n, seed=3)
    vec.reset()
    vec.engine.check()
    st = vec.engine.get_state()
    seen = set()
    for i in (0, 5, n - 1):
        polys, segs = vec.engine.get_geometry(i)
        assert len(polys) == 510 and len(segs) == 256
        links = _links_from_device_polys(polys)
        assert len(links) == 63
        # spanning tree: every cell reached exactly once
        reached = {(0, 0)}
        for ci, cj, d in links:
            assert (ci, cj) in reached
            nxt = (ci + DI[d], cj + DJ[d])
            assert nxt not in reached
            reached.add(nxt)
        assert len(reached) == 64
        seen.add(tuple(links))
        host = FixedMaze(links)
        sc = scene_from_env(host)
        for key, field in (("polys_v", "v"), ("polys_uv", "uv"), ("polys_n", "n"), ("polys_nv", "nv"), ("polys_tex", "tex")):
            assert np.array_equal(sc[key], polys[field]), (i, key)
        assert np.array_equal(sc["wall_segs"], segs)
        # placement validity + frame parity
        sc["agent_pos"], sc["agent_dir"] = st["agent_pos"][i], st["agent_dir"][i]
        sc["ents_pos"], sc["ents_dir"] = st["ent_pos"][i, :1], st["ent_dir"][i, :1]
        dyn = pyoracle.Dynamics(sc, pyoracle.TASK_GOTO, 1536)
        assert dyn.intersect(-1, st["agent_pos"][i, 0], st["agent_pos"][i, 2], 0.4) == 0
        assert np.array_equal(vec.obs[i].cpu().numpy(), pyoracle.render(sc)["rgb"])
    assert len(seen) == 3                                        # different mazes per env
    # stepping + auto-reset (forward-biased so that some episode ends)
    g = torch.Generator(device="cuda").manual_seed(2)
    for t in range(60):
        vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
    vec.engine.check()
    vec.close()


def test_vec_env_host_generation_path():
    """Worlds can also be generated on the host (reference-compatible numpy stream) and injected."""
    import pyoracle
    from miniworld_amd.scene import scene_from_env
    from miniworld_amd.vec_env import MiniWorldVecEnv
    vec = MiniWorldVecEnv("MiniWorld-Maze-v0", 4, seed=0)
    vec.host_generate(range(4), [0, 1, 2, 3])
    vec.engine.render(vec.obs, None)
    st = vec.engine.get_state()
    s0, tr, meta, obs = helpers.load_case("maze_s0")
    assert np.array_equal(vec.obs[0].cpu().numpy(), obs[0]["rgb"])       # seed 0 == the reference's reset(seed=0)
    assert np.array_equal(st["agent_pos"][0], s0["agent_pos"])
    vec.close()


def test_vec_env_texture_domain_randomisation_on_device():
    """Hallway with domain_rand: every env draws its own wall texture variant (concrete_1..4, one of
    them 768^2 so the texcoords change too), sky / light / camera parameters; frames == oracle."""
    import torch
    import pyoracle
    from miniworld_amd import assets
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 64
    vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, domain_rand=True, want_depth=True, seed=4)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(5)
    for _ in range(12):
        vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
    vec.engine.check()
    st = vec.engine.get_state()
    names = vec._tex_dr_variants
    walls = set()
    for i in range(n):
        polys, segs = vec.engine.get_geometry(i)
        assert len(polys) == 6 and len(segs) == 4
        wall_variant = names[int(polys["tex"][2])]
        walls.add(wall_variant)
        w, h = assets.texture_size(wall_variant)
        # east wall of the 12 x 4 hallway: 4 m wide, 2.74 m high (gen_texcs_wall, miniworld.py:82-103)
        assert np.allclose(polys["uv"][2][2], [4 * 512 / w, 2.74 * 512 / h], rtol=1e-6)
    assert walls == {"concrete_1", "concrete_2", "concrete_3", "concrete_4"}
    assert len(np.unique(st["light"][:, 0])) > n // 2 and (np.abs(st["cam"][:, 2]) <= 5).all()
    for i in (0, 9, n - 1):
        polys, segs = vec.engine.get_geometry(i)
        sc = _scene_of_env(vec, st, i)
        sc["polys_v"], sc["polys_uv"], sc["polys_n"] = polys["v"], polys["uv"], polys["n"]
        sc["polys_nv"], sc["polys_tex"] = polys["nv"], polys["tex"]
        sc["tex_names"] = np.array(names)
        sc["ents_kind"], sc["ents_mesh"] = st["ent_kind"][i, :1], st["ent_mesh"][i, :1]
        for k in ("ents_pos", "ents_dir", "ents_size", "ents_color", "ents_scale", "ents_radius", "ents_height", "ents_static"):
            sc[k] = sc[k][:1]
        want = pyoracle.render(sc)
        assert np.array_equal(vec.obs[i].cpu().numpy(), want["rgb"]), f"env {i}"
        assert np.array_equal(vec.depth[i].cpu().numpy(), want["depth"]), f"env {i}"
    vec.close()


@pytest.mark.parametrize("case,top", [("hallway_s0", False), ("pickup_s0", False), ("fourrooms_s0", True), ("pickup_dr_s1", True)])
def test_render_800x600x16_matches_oracle(case, top):
    """env.render() (rgb_array): the 800x600 16-sample visualisation buffer (miniworld.py:518, 1354-1362),
    agent view and top view, meshes included; bit-exact against the oracle at the same size."""
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.scene import scene_from_env
    s0, tr, meta, obs = helpers.load_case(case)
    env = getattr(envs, str(meta["env"]))(render_mode="rgb_array", view="top" if top else "agent", **helpers.env_kwargs_of(meta))
    o, _ = env.reset(seed=int(meta["seed"]))
    img = env.render()
    assert img.shape == (600, 800, 3) and img.dtype == np.uint8
    want = pyoracle.render(scene_from_env(env), width=800, height=600, nsamples=16, meshes=helpers.golden_meshes(s0),
                           view="top" if top else "agent", render_agent=top)
    diff = np.abs(img.astype(int) - want["rgb"].astype(int))
    assert diff.max() == 0, f"{np.count_nonzero(diff)} values differ, max {diff.max()}"
    if not top:
        # the reference's own consistency check (tests/test_miniworld.py:26-31)
        assert abs(float(o.mean()) - float(img.mean())) < 5
    env.close()


def test_vec_env_get_visible_ents_matches_oracle():
    """get_visible_ents (miniworld.py:1238-1333) for a whole batch: occlusion queries around the
    0.2 m proxy boxes, compared env by env with the oracle's restatement; removed entities report 0."""
    import torch
    import pyoracle
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 96
    vec = MiniWorldVecEnv("MiniWorld-PickupObjects-v0", n, domain_rand=True, seed=21)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    seen_any = seen_none = 0
    for rnd in range(3):
        for t in range(40):
            act = torch.randint(0, 5, (n,), generator=g, device="cuda", dtype=torch.int32)
            vec.step(act)
        vis = vec.get_visible_ents().cpu().numpy()
        assert vis.shape == (n, vec.engine.E) and vis.dtype == bool
        st = vec.engine.get_state()
        for i in range(n):
            want = pyoracle.visible_ents(_scene_of_env(vec, st, i))
            assert np.array_equal(vis[i, :len(want)], want), f"round {rnd} env {i}: {vis[i]} vs {want}"
            assert not vis[i][st["ent_kind"][i] == 0].any()
        seen_any += int(vis.any(axis=1).sum())
        seen_none += int((~vis.any(axis=1)).sum())
    assert seen_any > 0 and seen_none > 0
    vec.engine.check()
    vec.close()


@pytest.mark.parametrize("case", ["roomobjects_s0", "putnext_s0", "fourrooms_s0", "hallway_s0", "pickup_fwd_s2", "pickup_dr_s1"])
def test_single_env_get_visible_ents(case):
    """MiniWorldEnv.get_visible_ents returns the set of entity objects, along a trajectory."""
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.scene import scene_from_env
    s0, tr, meta, obs = helpers.load_case(case)
    env = getattr(envs, str(meta["env"]))(**helpers.env_kwargs_of(meta))
    env.reset(seed=int(meta["seed"]))
    for t, a in enumerate(tr["action"][:60]):
        if t % 6 == 0:
            got = env.get_visible_ents()
            ents = [e for e in env.entities if e is not env.agent]
            want = pyoracle.visible_ents(scene_from_env(env))
            assert got == {e for e, v in zip(ents, want) if v}, f"step {t}"
            assert env.agent not in got
        _, _, term, trunc, _ = env.step(int(a))
        if term or trunc:
            break
    env.close()


@pytest.mark.parametrize("env_id,kw", [("MiniWorld-Hallway-v0", {}), ("MiniWorld-PickupObjects-v0", {"domain_rand": True})])
def test_vec_env_fused_wrapper_layouts(env_id, kw):
    """obs_layout="cwh" / "grey": the raster kernel's store in the layout of PyTorchObsWrapper /
    GreyscaleWrapper equals those wrappers applied to the plain observation (wrappers.py:24, :44),
    bit for bit (greyscale in numpy's float64 evaluation order), for both raster kernels."""
    import torch
    from miniworld_amd import wrappers
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 64
    vecs = {k: MiniWorldVecEnv(env_id, n, seed=5, obs_layout=k, **kw) for k in ("hwc", "cwh", "grey")}
    g = torch.Generator(device="cuda").manual_seed(0)
    for v in vecs.values():
        v.reset()
    for t in range(12):
        act = torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32)
        act = wrappers.stochastic_actions(act, prob=0.8, random_action=2, generator=g)
        outs = {k: v.step(act)[0].cpu().numpy() for k, v in vecs.items()}
        raw = outs["hwc"]
        assert outs["cwh"].shape == (n, 3, 80, 60) and outs["cwh"].dtype == np.uint8
        assert np.array_equal(outs["cwh"], raw.transpose(0, 3, 2, 1))
        assert outs["grey"].shape == (n, 60, 80, 1) and outs["grey"].dtype == np.float64
        want = 0.30 * raw[..., 0] + 0.59 * raw[..., 1] + 0.11 * raw[..., 2]
        assert np.array_equal(outs["grey"][..., 0], want)
    top = vecs["cwh"].render_top_view()
    assert top.shape == (n, 60, 80, 3) and np.array_equal(top.cpu().numpy(), vecs["hwc"].render_top_view().cpu().numpy())
    for v in vecs.values():
        v.engine.check()
        v.close()


_PROGRAM_FAMILIES = ["FourRooms", "TMaze", "TMazeLeft", "TMazeRight", "YMaze", "YMazeLeft", "YMazeRight", "WallGap",
                     "ThreeRooms", "PutNext", "RoomObjects", "Sidewalk", "Sign", "CollectHealth"]


def _host_scene_with_device_state(h, st, i):
    """Oracle scene of host env h (geometry, textures, entity kinds from the host world) at the device's poses."""
    from miniworld_amd.objmesh import ObjMesh
    from miniworld_amd.scene import scene_from_env
    sc = scene_from_env(h)
    E = len(sc["ents_kind"])
    sc["agent_pos"], sc["agent_dir"] = st["agent_pos"][i], st["agent_dir"][i]
    sc["ents_pos"], sc["ents_dir"] = st["ent_pos"][i, :E], st["ent_dir"][i, :E]
    meshes = {}
    for name in [str(m) for m in sc["mesh_names"]]:
        m = ObjMesh.get(name)
        meshes[name] = {"verts": m.verts, "norms": m.norms, "texcs": m.texcs, "colors": m.colors}
    return sc, meshes


@pytest.mark.parametrize("dr", [False, True])
@pytest.mark.parametrize("cls_name", _PROGRAM_FAMILIES)
def test_placement_program_families_reset_on_the_device_like_the_reference(cls_name, dr):
    """The fixed-floorplan families (placement programs, MW_GEN_PROGRAM): a batch seeded with s holds the worlds of the
    reference's reset(seed=s + i) — coin flips, room choices by area, rejection-sampled placements in rotated rooms,
    drawn box sizes / object colours, pre-drawn directions, fixed entities, with domain randomisation the texture
    variants of every room (re-emitted texcoords), colours, light and camera — episode after episode on one stream;
    frames equal the oracle's render of the host world."""
    import torch
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.scene import polys_array, scene_from_env
    from miniworld_amd.vec_env import MiniWorldVecEnv
    if cls_name == "Sign" and dr:
        pytest.skip("Sign fixes domain_rand=False (sign.py:92-98)")
    n, s, k_steps = 24, 300, 6
    vec = MiniWorldVecEnv(f"MiniWorld-{cls_name}-v0", n, seed=s, domain_rand=dr, autoreset=False)
    assert vec.rng_mode == "pcg64" and not vec.host_autoreset
    obs = vec.reset()
    kw = {} if cls_name == "Sign" else {"domain_rand": dr}
    hosts = [getattr(envs, cls_name)(host_only=True, **kw) for _ in range(n)]
    for i, h in enumerate(hosts):
        h.reset(seed=s + i)
    st = vec.engine.get_state()
    for i, h in enumerate(hosts):
        _assert_same_world(vec, st, i, h, "episode 1")
    if not vec.engine.cfg.shared_geometry:
        for i in (0, n // 2, n - 1):
            polys, segs = vec.engine.get_geometry(i)
            sc = scene_from_env(hosts[i])
            want = polys_array(sc, {k: vec.tex_ids[str(v)] for k, v in enumerate(sc["tex_names"])})
            assert len(polys) == len(want), (i, len(polys), len(want))
            for f in ("v", "uv", "n", "nv", "tex"):
                assert np.array_equal(polys[f], want[f]), (i, f)
            assert np.array_equal(segs, np.asarray(sc["wall_segs"], np.float64).reshape(-1, 2, 2)), i
    for i in (0, 7, n - 1):
        sc, meshes = _host_scene_with_device_state(hosts[i], st, i)
        assert np.array_equal(obs[i].cpu().numpy(), pyoracle.render(sc, meshes=meshes)["rgb"]), (cls_name, i)
    g = torch.Generator(device="cuda").manual_seed(8)
    for t in range(k_steps):
        vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
    vec.engine.reset(None, None)
    for h in hosts:
        if dr:
            for t in range(k_steps):
                for name in ("forward_step", "forward_drift", "turn_step"):
                    h.params.sample(h.np_random, name)
        h.reset()
    st = vec.engine.get_state()
    for i, h in enumerate(hosts):
        _assert_same_world(vec, st, i, h, "episode 2")
    vec.engine.check()
    vec.close()


@pytest.mark.parametrize("env_id,cls_name", [("MiniWorld-Hallway-v0", "Hallway"), ("MiniWorld-OneRoom-v0", "OneRoom"),
                                             ("MiniWorld-OneRoomS6-v0", "OneRoomS6")])
def test_device_reset_draws_the_reference_stream(env_id, cls_name):
    """rng="pcg64" (the default where implemented): env i of a batch seeded with s is, bit for bit, the world of
    env.reset(seed=s + i) of the host classes (themselves seed-exact with the reference, test_host_logic_cpu),
    and the next episode continues the same numpy stream like env.reset() without a seed does — both through
    an explicit mw_reset and through the same-step auto-reset inside K1."""
    import torch
    from miniworld_amd import envs
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n, s = 96, 1000
    vec = MiniWorldVecEnv(env_id, n, seed=s)
    assert vec.rng_mode == "pcg64"
    vec.reset()
    hosts = [getattr(envs, cls_name)(host_only=True) for _ in range(n)]

    def check(tag, which=None):
        st = vec.engine.get_state()
        for i in (range(n) if which is None else which):
            h = hosts[i]
            assert np.array_equal(st["agent_pos"][i], h.agent.pos) and st["agent_dir"][i] == h.agent.dir, (tag, i)
            assert np.array_equal(st["ent_pos"][i, 0], h.box.pos) and st["ent_dir"][i, 0] == h.box.dir, (tag, i)

    for i, h in enumerate(hosts):
        h.reset(seed=s + i)
    check("first episode")
    for h in hosts:
        h.reset()                         # no seed: the episode continues the env's stream (miniworld.py:551)
    vec.engine.reset(None, None)          # same on the device: regenerate every env, keep the streams
    check("second episode")
    # same-step auto-reset: walk until some envs finish; their new worlds are the hosts' third episodes
    g = torch.Generator(device="cuda").manual_seed(4)
    done_once = np.zeros(n, bool)
    for t in range(400):
        act = torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32)
        act[torch.rand(n, generator=g, device="cuda") < 0.6] = 2
        _, _, term, trunc = vec.step(act)
        d = (term | trunc).bool().cpu().numpy()
        fresh = d & ~done_once
        if fresh.any():
            for i in np.nonzero(fresh)[0]:
                hosts[i].reset()
            check(f"auto-reset at step {t}", np.nonzero(fresh)[0])
            done_once |= d
        if done_once.sum() > n // 3:
            break
    assert done_once.any()
    vec.engine.check()
    vec.close()
    # the Philox stream is still there and differs
    vec2 = MiniWorldVecEnv(env_id, 8, seed=s, rng="philox")
    vec2.reset()
    st = vec2.engine.get_state()
    hosts[0].reset(seed=s)
    assert not np.array_equal(st["agent_pos"][0], hosts[0].agent.pos)
    vec2.close()


def test_vector_env_adapter_follows_gymnasium_convention():
    """MiniWorldVectorEnv: reset -> (obs, infos), step -> 5-tuple, same-step autoreset, numpy or torch."""
    from miniworld_amd.vector import MiniWorldVectorEnv
    envs = MiniWorldVectorEnv("MiniWorld-OneRoomS6-v0", 32, to_numpy=True, seed=3)
    assert envs.num_envs == 32 and envs.single_action_space.n == 3
    assert tuple(envs.single_observation_space.shape) == (60, 80, 3) and tuple(envs.observation_space.shape) == (32, 60, 80, 3)
    obs, infos = envs.reset(seed=11)
    assert isinstance(obs, np.ndarray) and obs.shape == (32, 60, 80, 3) and obs.dtype == np.uint8 and infos == {}
    rng = np.random.default_rng(0)
    n_done = 0
    for t in range(150):
        obs, rew, term, trunc, infos = envs.step(rng.integers(0, 3, 32))
        assert obs.shape == (32, 60, 80, 3) and rew.shape == (32,) and term.dtype == bool and trunc.dtype == bool
        n_done += int((term | trunc).sum())
        assert 1.0 < obs.mean() < 254.0
    assert n_done >= 32          # max_episode_steps = 100: every env finished at least once and kept running
    assert envs.render().shape == (32, 60, 80, 3)
    envs.close()


def test_batched_info_matches_the_host_classes():
    """The batched API's `info` (MiniWorldVecEnv.infos / the VectorEnv adapter's fifth value) against the env classes' own
    step(): info["goal_pos"] of TMaze / YMaze (tmaze.py:89, ymaze.py:125) and info["health"] of CollectHealth
    (collecthealth.py:100), env i of a batch seeded with s being env.reset(seed=s + i)."""
    from miniworld_amd import envs as host_envs
    from miniworld_amd.vector import MiniWorldVectorEnv
    for env_id, cls in (("MiniWorld-TMaze-v0", "TMaze"), ("MiniWorld-YMazeLeft-v0", "YMazeLeft")):
        envs = MiniWorldVectorEnv(env_id, 6, to_numpy=True, seed=40, autoreset=False)
        envs.reset(seed=40)
        _, _, _, _, infos = envs.step(np.zeros(6, np.int64))
        assert set(infos) == {"goal_pos"} and infos["goal_pos"].shape == (6, 3) and infos["goal_pos"].dtype == np.float64
        for i in range(6):
            h = getattr(host_envs, cls)()
            h.reset(seed=40 + i)
            _, _, _, _, hi = h.step(0)
            h.close()
            assert np.array_equal(infos["goal_pos"][i], np.asarray(hi["goal_pos"], np.float64)), (env_id, i)
        envs.close()
    envs = MiniWorldVectorEnv("MiniWorld-CollectHealth-v0", 5, to_numpy=True, seed=7, autoreset=False)
    envs.reset(seed=7)
    hosts = []
    for i in range(5):
        h = host_envs.CollectHealth()
        h.reset(seed=7 + i)
        hosts.append(h)
    rng = np.random.default_rng(5)
    for t in range(12):
        a = rng.integers(0, 3, 5)
        _, _, term, _, infos = envs.step(a)
        assert set(infos) == {"health"} and infos["health"].dtype == np.int32
        for i, h in enumerate(hosts):
            _, _, hterm, _, hi = h.step(int(a[i]))
            assert int(infos["health"][i]) == int(hi["health"]) and bool(term[i]) == bool(hterm), (t, i)
    envs.close()
    for h in hosts:
        h.close()
    # every other env: no info keys, like miniworld.py:730 — only the same-step vector env's mask of the envs that just finished
    envs = MiniWorldVectorEnv("MiniWorld-Hallway-v0", 2, seed=0)
    envs.reset(seed=0)
    info = envs.step(np.zeros(2, np.int64))[4]
    assert set(info) == {"_final_info"} and not bool(info["_final_info"].any())
    envs.close()
    envs = MiniWorldVectorEnv("MiniWorld-Hallway-v0", 2, seed=0, autoreset=False)
    envs.reset(seed=0)
    assert envs.step(np.zeros(2, np.int64))[4] == {}
    envs.close()


def test_final_info_is_the_finished_episodes_info_under_same_step_autoreset():
    """With the same-step auto-reset an env that just finished reports its NEW episode in `info`; the finished episode's own info
    (what a gymnasium same-step vector env returns as info["final_info"], masked by info["_final_info"]) is kept by the step kernel
    before it installs the next world: CollectHealth's terminal health (collecthealth.py:100; the wave-per-env K1) and TMaze's
    goal_pos of the episode that was truncated (tmaze.py:89; the dense K1), against the env classes' own episodes."""
    from miniworld_amd import envs as host_envs
    from miniworld_amd.vector import MiniWorldVectorEnv
    n = 4
    envs = MiniWorldVectorEnv("MiniWorld-CollectHealth-v0", n, to_numpy=True, seed=21)
    envs.reset(seed=21)
    hosts = []
    for i in range(n):
        h = host_envs.CollectHealth()
        h.reset(seed=21 + i)
        hosts.append(h)
    ended = np.zeros(n, bool)
    for t in range(60):
        a = np.full(n, t % 2, np.int64)          # turning on the spot: the health runs out after 50 steps
        _, _, term, trunc, infos = envs.step(a)
        assert set(infos) == {"health", "final_info", "_final_info"} and set(infos["final_info"]) == {"health", "_health"}
        assert np.array_equal(infos["_final_info"], term | trunc) and np.array_equal(infos["final_info"]["_health"], term | trunc)
        for i, h in enumerate(hosts):
            if ended[i]:
                continue
            _, _, hterm, htrunc, hi = h.step(int(a[i]))
            assert bool(term[i]) == bool(hterm) and bool(trunc[i]) == bool(htrunc), (t, i)
            if hterm or htrunc:
                ended[i] = True
                assert int(infos["final_info"]["health"][i]) == int(hi["health"]) <= 0, (t, i)
                assert int(infos["health"][i]) == 100, (t, i)                  # the new episode's (collecthealth.py:60)
            else:
                assert int(infos["health"][i]) == int(hi["health"]), (t, i)
    assert ended.all()
    envs.close()
    for h in hosts:
        h.close()

    envs = MiniWorldVectorEnv("MiniWorld-TMaze-v0", n, to_numpy=True, seed=33)
    envs.reset(seed=33)
    first = None
    for t in range(280):                         # max_episode_steps = 280 (tmaze.py:28): every env is truncated on the last one
        _, _, term, trunc, infos = envs.step(np.zeros(n, np.int64))
        first = infos["goal_pos"].copy() if first is None else first
        assert not (term | trunc).any() or t == 279
    assert trunc.all() and infos["_final_info"].all()
    assert np.array_equal(infos["final_info"]["goal_pos"], first)
    for i in range(n):
        h = host_envs.TMaze()
        h.reset(seed=33 + i)
        _, _, _, _, hi = h.step(0)
        h.close()
        assert np.array_equal(infos["final_info"]["goal_pos"][i], np.asarray(hi["goal_pos"], np.float64)), i
    # the running `info` has moved on to the new episodes' boxes (left or right arm at random: not all four can have stayed)
    nxt = envs.step(np.zeros(n, np.int64))[4]
    assert np.array_equal(nxt["goal_pos"], infos["goal_pos"]) and not nxt["_final_info"].any()
    assert np.array_equal(nxt["final_info"]["goal_pos"], first)                # kept until the next episode ends
    envs.close()


def test_final_info_mask_exists_for_every_family_and_final_values_are_copies():
    """A family without info keys (Hallway) still reports which envs finished ("_final_info"); and with tensors left on the
    device the values under final_info are copies — a consumer may hold on to them across the next step."""
    import torch
    from miniworld_amd.vector import MiniWorldVectorEnv
    envs = MiniWorldVectorEnv("MiniWorld-Hallway-v0", 64, seed=3)
    envs.reset(seed=3)
    seen = 0
    for t in range(260):                          # max_episode_steps = 250 (hallway.py:31): every env finishes in here
        _, _, term, trunc, infos = envs.step(torch.full((64,), 2 if t % 3 else 0, dtype=torch.int32, device="cuda"))
        assert set(infos) == {"_final_info"}
        assert torch.equal(infos["_final_info"], (term | trunc))
        seen += int(infos["_final_info"].sum())
    assert seen >= 64
    envs.close()
    envs = MiniWorldVectorEnv("MiniWorld-TMaze-v0", 8, seed=5)
    envs.reset(seed=5)
    for t in range(280):
        infos = envs.step(torch.zeros(8, dtype=torch.int32, device="cuda"))[4]
    held = infos["final_info"]["goal_pos"]
    snapshot = held.clone()
    for t in range(280):                          # the next episodes end too: the engine's buffer is rewritten
        infos = envs.step(torch.ones(8, dtype=torch.int32, device="cuda"))[4]
    assert torch.equal(held, snapshot)
    assert held.data_ptr() != infos["final_info"]["goal_pos"].data_ptr()
    envs.close()


def _assert_same_world(vec, st, i, h, tag):
    """Device state of env i == host env h (same seed, same episode): poses, entity table, per-episode parameters."""
    from miniworld_amd.entity import Box, MeshEnt
    assert np.array_equal(st["agent_pos"][i], h.agent.pos) and st["agent_dir"][i] == h.agent.dir, (tag, i, "agent")
    cam = [h.agent.cam_height, h.agent.cam_fwd_disp, h.agent.cam_pitch, h.agent.cam_fov_y]
    assert np.array_equal(st["cam"][i], np.array(cam, np.float64)), (tag, i, "camera")
    light = np.concatenate([h.sky_color, h.light_pos, h.light_color, h.light_ambient]).astype(np.float64)
    assert np.array_equal(st["light"][i], light), (tag, i, "sky / light")
    ents = [e for e in h.entities if e is not h.agent]
    assert int((st["ent_kind"][i] != 0).sum()) == len(ents), (tag, i, "entity count")
    inv_mesh = {v: k for k, v in vec.mesh_ids.items()}
    for k, e in enumerate(ents):
        assert np.array_equal(st["ent_pos"][i, k], e.pos) and st["ent_dir"][i, k] == e.dir, (tag, i, k, "entity pose")
        if isinstance(e, Box):
            assert st["ent_kind"][i, k] == 1 and np.array_equal(st["ent_geom"][i, k, 3:6], e.color_vec), (tag, i, k, "box")
            assert np.array_equal(st["ent_geom"][i, k, 0:3], np.asarray(e.size, np.float64)), (tag, i, k, "box size")
        elif isinstance(e, MeshEnt):
            assert st["ent_kind"][i, k] == 2 and inv_mesh[int(st["ent_mesh"][i, k])] == e.mesh_name, (tag, i, k, "mesh")
            assert st["ent_geom"][i, k, 6] == e.scale and st["ent_geom"][i, k, 7] == e.radius, (tag, i, k, "mesh scale")


@pytest.mark.parametrize("env_id,cls_name,dr", [
    ("MiniWorld-Hallway-v0", "Hallway", True), ("MiniWorld-OneRoom-v0", "OneRoom", True),
    ("MiniWorld-PickupObjects-v0", "PickupObjects", False), ("MiniWorld-PickupObjects-v0", "PickupObjects", True),
    ("MiniWorld-MazeS3-v0", "MazeS3", False), ("MiniWorld-Maze-v0", "Maze", False),
    ("MiniWorld-MazeS3-v0", "MazeS3", True), ("MiniWorld-Maze-v0", "Maze", True)])
def test_device_reset_reference_stream_all_generators(env_id, cls_name, dr):
    """MW_RNG_PCG64 for every device generator, with domain randomisation: bounded integers (object kinds, colours,
    texture variants, the maze's neighbour orders), the area-weighted room choice, per-episode parameters and the
    three per-step parameter draws all come out of numpy's stream in the reference's order, so a batch seeded
    with s holds the worlds of env.reset(seed=s+i), episode after episode."""
    import torch
    from miniworld_amd import envs
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n, s, k_steps = (24, 77, 7) if "Maze" in env_id else (48, 500, 9)
    vec = MiniWorldVecEnv(env_id, n, seed=s, domain_rand=dr, autoreset=False)
    assert vec.rng_mode == "pcg64"
    vec.reset()
    hosts = [getattr(envs, cls_name)(host_only=True, domain_rand=dr) for _ in range(n)]
    for i, h in enumerate(hosts):
        h.reset(seed=s + i)
    st = vec.engine.get_state()
    for i, h in enumerate(hosts):
        _assert_same_world(vec, st, i, h, "episode 1")
    if "Maze" in env_id or dr:
        # per-env geometry: the room polygons (and with DR their texture variants) are the host's
        from miniworld_amd.scene import polys_array, scene_from_env
        for i in (0, n // 2, n - 1):
            polys, segs = vec.engine.get_geometry(i)
            sc = scene_from_env(hosts[i])
            want = polys_array(sc, {k: vec.tex_ids[str(v)] for k, v in enumerate(sc["tex_names"])})
            assert len(polys) == len(want), (i, len(polys), len(want))
            for f in ("v", "uv", "n", "nv", "tex"):
                assert np.array_equal(polys[f], want[f]), (i, f)
            assert np.array_equal(segs, np.asarray(sc["wall_segs"], np.float64).reshape(-1, 2, 2)), i
    # a few steps (with DR each one draws forward_step, forward_drift, turn_step: miniworld.py:677-680), then episode 2
    g = torch.Generator(device="cuda").manual_seed(8)
    for t in range(k_steps):
        vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
    vec.engine.reset(None, None)
    for h in hosts:
        if dr:
            for t in range(k_steps):
                for name in ("forward_step", "forward_drift", "turn_step"):
                    h.params.sample(h.np_random, name)
        h.reset()
    st = vec.engine.get_state()
    for i, h in enumerate(hosts):
        _assert_same_world(vec, st, i, h, "episode 2")
    vec.engine.check()
    vec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,cls_name", [("MiniWorld-MazeS3-v0", "MazeS3"), ("MiniWorld-Maze-v0", "Maze")])
def test_maze_with_domain_rand_draws_a_texture_variant_per_room(env_id, cls_name, monkeypatch):
    """Room._gen_static_data with an rng draws a wall, a floor and a ceiling variant for EVERY room of a maze, inside the
    first place_entity (miniworld.py:295-297, 856-857; opengl.py:134-138).  The Maze's own textures exist in one variant
    each, so here the asset inventory is given more (sizes differ: the texture coordinates follow): the device generator
    draws them in the reference's order — worlds, polygons (texture ids, texture coordinates) and the stream behind them
    are the host's, episode after episode."""
    import torch
    from miniworld_amd import assets, envs
    from miniworld_amd.scene import polys_array, scene_from_env
    from miniworld_amd.vec_env import MiniWorldVecEnv
    real = assets.texture_variants
    more = {"brick_wall": ["brick_wall_1", "concrete_1", "concrete_3", "asphalt_1"], "floor_tiles_bw": ["floor_tiles_bw_1", "concrete_2"],
            "concrete_tiles": ["concrete_tiles_1", "concrete_4", "brick_wall_1"]}
    monkeypatch.setattr(assets, "texture_variants", lambda name: more.get(name, real(name)))
    n, s, k_steps = 16, 31, 5
    vec = MiniWorldVecEnv(env_id, n, seed=s, domain_rand=True, autoreset=False)
    assert vec.rng_mode == "pcg64"
    vec.reset()
    hosts = [getattr(envs, cls_name)(host_only=True, domain_rand=True) for _ in range(n)]
    for i, h in enumerate(hosts):
        h.reset(seed=s + i)
    seen = set()
    for episode in (1, 2):
        st = vec.engine.get_state()
        for i, h in enumerate(hosts):
            _assert_same_world(vec, st, i, h, f"episode {episode}")
            polys, segs = vec.engine.get_geometry(i)
            sc = scene_from_env(h)
            want = polys_array(sc, {k: vec.tex_ids[str(v)] for k, v in enumerate(sc["tex_names"])})
            assert len(polys) == len(want), (i, len(polys), len(want))
            for f in ("v", "uv", "n", "nv", "tex"):
                assert np.array_equal(polys[f], want[f]), (episode, i, f)
            seen |= {str(v) for v in sc["tex_names"]}
        if episode == 1:
            g = torch.Generator(device="cuda").manual_seed(8)
            for t in range(k_steps):
                vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
            vec.engine.reset(None, None)
            for h in hosts:
                for t in range(k_steps):
                    for name in ("forward_step", "forward_drift", "turn_step"):
                        h.params.sample(h.np_random, name)
                h.reset()
    assert len(seen) >= 6       # the variants really were drawn
    # and the frames are the oracle's for such a world
    import pyoracle
    vec.step(torch.zeros(n, dtype=torch.int32, device="cuda"))
    st = vec.engine.get_state()
    for i in (0, n - 1):
        want = pyoracle.render(helpers.scene_of_vec_env(vec, st, i))
        assert np.array_equal(vec.obs[i].cpu().numpy(), want["rgb"]), i
    vec.engine.check()
    vec.close()


_DEVICE_FAMILIES = {"Hallway": "MiniWorld-Hallway-v0", "OneRoom": "MiniWorld-OneRoom-v0", "Maze": "MiniWorld-Maze-v0",
                    "MazeS3": "MiniWorld-MazeS3-v0", "PickupObjects": "MiniWorld-PickupObjects-v0"}
_DEVICE_FAMILIES.update({c: f"MiniWorld-{c}-v0" for c in _PROGRAM_FAMILIES})


# (putnext_poke teleports a box mid-trajectory: covered through the single-env API and the C-level step test)
@pytest.mark.parametrize("case", [c for c in ALL_CASES if "poke" not in c])
def test_batched_env_reproduces_reference_trajectory_from_seed(case):
    """The whole path with nothing from the host classes in between: a batched env seeded like the reference run
    that produced the fixture (tools/gen_golden.py: the reference's own miniworld.py under GL stubs) generates
    that world on the device (MW_RNG_PCG64), and the reference's action sequence then reproduces its rewards,
    flags, poses — with domain randomisation also its per-step forward_step / drift / turn_step draws — and
    the stored frames."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    s0, tr, meta, obs = helpers.load_case(case)
    env_id = _DEVICE_FAMILIES[str(meta["env"])]
    kw = helpers.env_kwargs_of(meta)
    kw.pop("domain_rand", None)
    vec = MiniWorldVecEnv(env_id, 2, seed=int(meta["seed"]), domain_rand=bool(meta["domain_rand"]), autoreset=False, **kw)
    assert vec.rng_mode == "pcg64" and not vec.host_autoreset
    o = vec.reset()
    st = vec.engine.get_state()
    assert np.array_equal(st["agent_pos"][0], s0["agent_pos"]) and st["agent_dir"][0] == s0["agent_dir"]
    E = len(s0["ents_kind"])
    assert np.array_equal(st["ent_pos"][0, :E], s0["ents_pos"]) and np.array_equal(st["ent_dir"][0, :E], s0["ents_dir"])
    assert np.array_equal(o[0].cpu().numpy(), obs[0]["rgb"])
    act = torch.zeros(2, dtype=torch.int32, device="cuda")
    worst = 0.0
    for t in range(len(tr["action"])):
        act[:] = int(tr["action"][t])
        o, rew, term, trunc = vec.step(act)
        assert np.float32(tr["reward"][t]) == rew[0].item(), (case, t)
        assert bool(term[0].item()) == bool(tr["term"][t]) and bool(trunc[0].item()) == bool(tr["trunc"][t]), (case, t)
        if (t + 1) in obs or t % 16 == 0 or t == len(tr["action"]) - 1:
            st = vec.engine.get_state()
            worst = max(worst, np.abs(st["agent_pos"][0] - tr["pos"][t]).max(), abs(st["agent_dir"][0] - tr["dir"][t]))
            assert int(st["carrying"][0]) == int(tr["carrying"][t]), (case, t)
        if (t + 1) in obs:
            assert np.array_equal(o[0].cpu().numpy(), obs[t + 1]["rgb"]), (case, t + 1)
    assert worst < 1e-12, (case, worst)
    vec.engine.check()
    vec.close()


@pytest.mark.parametrize("w,h", [(160, 120), (96, 64), (16, 4)])
def test_other_observation_sizes_match_oracle(w, h):
    """MiniWorldEnv(obs_width, obs_height) (miniworld.py:473-474): any multiple of the 16x4 tile; the hot kernels
    at another size, including the degenerate single-tile frame, equal the oracle at that size."""
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.scene import scene_from_env
    env = envs.FourRooms(obs_width=w, obs_height=h)
    o, _ = env.reset(seed=5)
    assert o.shape == (h, w, 3)
    for a in (2, 2, 0, 2, 1, 1, 2):
        o, *_ = env.step(a)
    want = pyoracle.render(scene_from_env(env), width=w, height=h)
    assert np.array_equal(o, want["rgb"])
    assert np.array_equal(env.render_depth(), want["depth"])
    env.close()


def test_capacity_overflow_is_reported_not_silent():
    """More triangles in the list than the records max_visible pays for (six per visible primitive): the geometry kernel
    sets a status bit, mw_check turns it into an error (nothing is dropped silently)."""
    import torch
    from miniworld_amd import engine as eng
    s0, tr, meta, obs = helpers.load_case("maze_s0")          # dozens of polygons in view
    e = helpers.make_engine_for_scene(s0, 1, max_visible=2)
    e.set_state(helpers.scene_state_arrays([s0]))
    rgb = torch.zeros((1, 60, 80, 3), dtype=torch.uint8, device="cuda")
    e.render(rgb, None)
    with pytest.raises(eng.EngineError, match="max_visible"):
        e.check()
    e.close()


def test_capacity_overflow_with_a_visiting_order_stays_in_bounds():
    """Big scenes (max_visible > 64) keep a near-to-far visiting order of max_vis + 1 entries per env.  A list longer than
    max_vis (but within the sort's 512 keys) must not be sorted into it: the neighbouring env's order and frame stay what
    they are, and mw_check reports the overflow.  Scene: a stack of 230 small quads in front of env 0's camera (460
    triangles + the room's, more than the 390 records of max_visible = 65), env 1 looks the other way."""
    import pyoracle
    import torch
    from miniworld_amd import engine as eng
    s0, tr, meta, obs = helpers.load_case("hallway_s0")
    base = helpers.frame_scene(s0, obs[sorted(obs)[0]])
    d = float(base["agent_dir"])
    fwd, right = np.array([np.cos(d), 0.0, -np.sin(d)]), np.array([np.sin(d), 0.0, np.cos(d)])
    eye = np.array(base["agent_pos"], np.float64) + np.array([0.0, float(base["cam_height"]), 0.0])
    overflowed = False
    for winding in (1, -1):                                   # one of the two faces the camera
        sc = {k: np.array(v, copy=True) for k, v in base.items()}
        n = 230
        quads = np.zeros((n, 4, 3), np.float32)
        for j in range(n):
            c = eye + fwd * (0.6 + 0.0005 * j)
            corners = [(-1, -1), (1, -1), (1, 1), (-1, 1)][::winding]
            quads[j] = [c + right * (0.05 * u) + np.array([0.0, 0.05 * v, 0.0]) for u, v in corners]
        sc["polys_v"] = np.concatenate([base["polys_v"], quads])
        sc["polys_uv"] = np.concatenate([base["polys_uv"], np.tile(base["polys_uv"][2:3], (n, 1, 1))])
        sc["polys_n"] = np.concatenate([base["polys_n"], np.tile((-fwd).astype(np.float32), (n, 1))])
        sc["polys_nv"] = np.concatenate([base["polys_nv"], np.full(n, 4, np.int32)])
        sc["polys_tex"] = np.concatenate([base["polys_tex"], np.full(n, base["polys_tex"][2], np.int32)])
        sc["polys_rgb"] = np.concatenate([base["polys_rgb"], np.ones((n, 3), np.float32)])
        sc["polys_xf"] = np.concatenate([base["polys_xf"], np.tile(base["polys_xf"][2:3], (n, 1))])
        away = {k: np.array(v, copy=True) for k, v in sc.items()}
        away["agent_dir"] = np.array(d + np.pi)
        want = pyoracle.render(away)["rgb"]
        e = helpers.make_engine_for_scene(sc, 2, max_visible=65)          # 390 records per env, visiting order on
        e.set_state(helpers.scene_state_arrays([sc, away]))
        rgb = torch.zeros((2, 60, 80, 3), dtype=torch.uint8, device="cuda")
        e.render(rgb, None)
        try:
            e.check()
        except eng.EngineError as exc:
            assert "max_visible" in str(exc)
            overflowed = True
        assert np.array_equal(rgb[1].cpu().numpy(), want), "the env beside the overflowing one lost its frame"
        e.close()
    assert overflowed, "neither winding of the stack overflowed the 390 records"


def test_first_mesh_frame_does_not_synchronise():
    """include/mwengine.h: "all device work is enqueued on the stream, nothing synchronises unless documented".  The mesh path's
    buffers (plane cache, sample keys, slow lists) are allocated by mw_upload_mesh, not inside the first frame: with the
    stream kept busy by unrelated work, the very first mw_step of an engine with mesh entities returns while that work is
    still running."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    vec = MiniWorldVecEnv("MiniWorld-PickupObjects-v0", 64, seed=2)
    seeds = np.arange(64, dtype=np.uint64) + np.uint64(2)
    vec.engine.reset(None, seeds)                             # worlds only: no frame has been drawn yet
    x = torch.randn((8192, 8192), device="cuda")
    torch.cuda.synchronize()
    for _ in range(40):                                       # a few hundred milliseconds of matrix products queued ahead
        y = x @ x
    busy = torch.cuda.Event()
    busy.record()
    act = torch.zeros(64, dtype=torch.int32, device="cuda")
    vec.step(act)
    still_running = not busy.query()
    torch.cuda.synchronize()
    assert still_running, "the first frame with mesh entities waited for the stream"
    assert 1.0 < float(vec.obs.float().mean()) < 254.0 and float(y.abs().sum()) > 0
    vec.close()


def test_slow_fragment_heads_survive_the_stamp_wrap():
    """The per-pixel chains of the fragments of mesh triangles that cross a frustum plane are tagged with a 16-bit frame
    stamp instead of being cleared every frame (mw_raster_mesh.hip: slow_pixel): a head that nothing has overwritten since frame F
    would read as valid again at frame F + 65 536, so the engine wipes the heads on the frame whose stamp is 0.  Here a batch
    looks at a view (stamp a), turns away for k frames, the sequence number is moved to 65 536 + a - k (test hook), and k
    turns back bring the first view back exactly on stamp a — through the wrap.  The frames must equal those of the same
    batch stepped without the jump."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n, k = 512, 6

    def run(jump):
        vec = MiniWorldVecEnv("MiniWorld-PickupObjects-v0", n, seed=31)
        vec.reset()                                         # frame 1
        left = torch.zeros(n, dtype=torch.int32, device="cuda")
        right = torch.ones(n, dtype=torch.int32, device="cuda")
        frames = []
        for _ in range(3):                                  # frames 2 .. 4: the view at stamp a = 4 (right, left, right: back and forth)
            vec.step(right if len(frames) % 2 == 0 else left)
            frames.append(vec.obs.clone())
        for _ in range(k):
            vec.step(left)
            frames.append(vec.obs.clone())
        if jump:
            # next frame would be 5 + k; move it to 65 536 + 5 - k: the k frames turning back carry the stamps 5 - k .. 4 (mod 2^16)
            rc = vec.engine.lib.mw_debug_set_mesh_frame_seq(vec.engine.h, 65536 + 5 - k)
            assert rc == 0, vec.engine.lib.mw_last_error(vec.engine.h)
        def stamps():
            from miniworld_amd.engine import _stream_ptr
            h = np.zeros((n, 60, 80), np.uint32)
            assert vec.engine.lib.mw_debug_get_slow_heads(vec.engine.h, h.ctypes.data, _stream_ptr(vec.engine.device)) == 0
            return h >> 16

        if jump:
            assert (stamps() != 0).sum() > 100, "no mesh triangle crossed a frustum plane: the test sees nothing"
        for i in range(k):
            vec.step(right)
            frames.append(vec.obs.clone())
            if jump and i == 1:
                # the frame whose stamp is 0 has been drawn: nothing of the frames before it is left in the heads
                assert (stamps() == 0).all(), "chain heads of earlier frames survived the frame with stamp 0"
        vec.engine.check()
        out = torch.stack(frames).cpu().numpy()
        vec.close()
        return out

    a, b = run(False), run(True)
    assert a.shape == b.shape and 1.0 < a.mean() < 254.0
    bad = np.argwhere((a != b).reshape(a.shape[0], -1).any(axis=1)).ravel()
    assert bad.size == 0, f"frames {bad.tolist()} differ after the stamp wrapped"


def test_two_engines_on_two_streams_stay_exact():
    """Multi-GPU readiness on one GPU: two engines in one process, each stepped on a stream of its own (what two ranks of a
    node do on two devices, here contending for one): frames, rewards and flags equal those of the same batches stepped
    alone on the default stream.  (Engines share nothing: state, records, textures and scratch are per engine.)"""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    mk = lambda: (MiniWorldVecEnv("MiniWorld-Hallway-v0", 768, seed=5), MiniWorldVecEnv("MiniWorld-PickupObjects-v0", 256, seed=9))     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(2)
    acts_a = torch.randint(0, 3, (50, 768), generator=g, device="cuda", dtype=torch.int32)
    acts_b = torch.randint(0, 5, (50, 256), generator=g, device="cuda", dtype=torch.int32)
    a, b = mk()
    a.reset(); b.reset()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    got = []
    for t in range(50):
        with torch.cuda.stream(s1):
            oa, ra, ta, ua = a.step(acts_a[t])
        with torch.cuda.stream(s2):
            ob, rb, tb, ub = b.step(acts_b[t])
        if t % 10 == 9:
            s1.synchronize(); s2.synchronize()
            got.append([x.clone() for x in (oa, ra, ta, ua, ob, rb, tb, ub)])
    torch.cuda.synchronize()
    a.engine.check(); b.engine.check()
    a.close(); b.close()
    a, b = mk()
    a.reset(); b.reset()
    k = 0
    for t in range(50):
        oa, ra, ta, ua = a.step(acts_a[t])
        ob, rb, tb, ub = b.step(acts_b[t])
        if t % 10 == 9:
            torch.cuda.synchronize()
            for x, y in zip(got[k], (oa, ra, ta, ua, ob, rb, tb, ub)):
                assert torch.equal(x, y), (t, k)
            k += 1
    a.close(); b.close()


def test_two_engines_are_independent():
    """Engines do not share state (the reference's display list id 1 and texture cache are process-global,
    miniworld.py:1027, opengl.py:111): two batches of different envs interleave their steps on one device."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    a = MiniWorldVecEnv("MiniWorld-Hallway-v0", 64, seed=1)
    b = MiniWorldVecEnv("MiniWorld-OneRoom-v0", 32, seed=1)
    a2 = MiniWorldVecEnv("MiniWorld-Hallway-v0", 64, seed=1)
    a.reset(); b.reset(); a2.reset()
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(30):
        act = torch.randint(0, 3, (64,), generator=g, device="cuda", dtype=torch.int32)
        oa = a.step(act)[0].clone()
        b.step(act[:32])
        oa2 = a2.step(act)[0]
        assert torch.equal(oa, oa2), t
    for v in (a, b, a2):
        v.engine.check()
        v.close()


@pytest.mark.parametrize("case", ["sidewalk_s0", "sidewalk_s3", "sign_s0", "sign_green_key_s1", "collecthealth_s13"])
def test_vec_env_host_rule_families_follow_reference_trajectory(case):
    """Sidewalk, Sign and CollectHealth in the batched API: Sidewalk's forbidden street, Sign's touch table / extra
    end-of-episode action and CollectHealth's health bookkeeping with kits respawning through the env's own numpy
    stream at the end of the entity list are K1 task rules fed by the placement program.  Env 0, generated from
    the fixture's seed on the device, reproduces the reference's rewards, flags, frames and poses."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    s0, tr, meta, obs = helpers.load_case(case)
    kw = helpers.env_kwargs_of(meta)
    vec = MiniWorldVecEnv("MiniWorld-%s-v0" % str(meta["env"]), 3, seed=int(meta["seed"]), autoreset=False, **kw)
    o = vec.reset()
    assert np.array_equal(o[0].cpu().numpy(), obs[0]["rgb"])
    act = torch.zeros(3, dtype=torch.int32, device="cuda")
    for t in range(len(tr["action"])):
        act[:] = int(tr["action"][t])
        o, rew, term, trunc = vec.step(act)
        assert np.float32(tr["reward"][t]) == rew[0].item(), (case, t)
        assert bool(term[0].item()) == bool(tr["term"][t]) and bool(trunc[0].item()) == bool(tr["trunc"][t]), (case, t)
        if (t + 1) in obs:
            assert np.array_equal(o[0].cpu().numpy(), obs[t + 1]["rgb"]), (case, t + 1)
    st = vec.engine.get_state()
    assert np.abs(st["agent_pos"][0] - tr["pos"][-1]).max() < 1e-12
    vec.close()


@pytest.mark.parametrize("env_id,cls_name,mes", [("MiniWorld-Hallway-v0", "Hallway", None), ("MiniWorld-MazeS3-v0", "MazeS3", None),
                                                 ("MiniWorld-MazeS3-v0", "MazeS3", 2), ("MiniWorld-Maze-v0", "Maze", 3)])
def test_spare_world_mode_keeps_the_reference_stream(env_id, cls_name, mes, monkeypatch):
    """MW_SPARE=1 (the default of small scenes and of the Maze): episodes end by copying a pre-generated
    world into place; blocks appended to K1's grid — for the Maze a kernel on the side stream — regenerate it.  The random stream is consumed in the same order,
    so explicit resets and auto-resets still produce the reference's worlds."""
    import torch
    from miniworld_amd import envs
    from miniworld_amd.vec_env import MiniWorldVecEnv
    monkeypatch.setenv("MW_SPARE", "1")
    n, s = (48 if mes is None else 16), 300
    # mes: episodes of 2 - 3 steps — a Maze env then needs its next spare while the side stream's refill kernel is still
    # on the last one (or has not started): the wait / regenerate-inline branches of the refill_mask protocol
    kw = {} if mes is None else {"max_episode_steps": mes}
    vec = MiniWorldVecEnv(env_id, n, seed=s, **kw)
    vec.reset()
    hosts = [getattr(envs, cls_name)(host_only=True, **kw) for _ in range(n)]
    for i, h in enumerate(hosts):
        h.reset(seed=s + i)
    st = vec.engine.get_state()
    for i, h in enumerate(hosts):
        _assert_same_world(vec, st, i, h, "episode 1")
    for h in hosts:
        h.reset()
    vec.engine.reset(None, None)
    st = vec.engine.get_state()
    for i, h in enumerate(hosts):
        _assert_same_world(vec, st, i, h, "episode 2")
    g = torch.Generator(device="cuda").manual_seed(4)
    episodes = np.full(n, 2)
    for t in range(300):
        act = torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32)
        act[torch.rand(n, generator=g, device="cuda") < 0.6] = 2
        _, _, term, trunc = vec.step(act)
        d = (term | trunc).bool().cpu().numpy()
        if d.any():
            st = vec.engine.get_state()
            for i in np.nonzero(d)[0]:
                hosts[i].reset()
                episodes[i] += 1
                _assert_same_world(vec, st, i, hosts[i], f"auto-reset at step {t} (episode {episodes[i]})")
        if episodes.max() >= (5 if mes is None else 16) and (episodes > 2).sum() > n // 4:
            break
    assert (episodes > 2).any()
    vec.engine.check()
    vec.close()


def test_vec_env_program_family_with_domain_rand_matches_the_reference_fixture():
    """FourRooms with domain_rand=True in the batched API: env 2 of a batch seeded with 0 is the reference's
    FourRooms(domain_rand=True).reset(seed=2) (fixture fourrooms_dr_s2: first frame and the whole trajectory with its
    per-step forward_step / drift / turn_step draws), generated and stepped on the device."""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    s0, tr, meta, obs = helpers.load_case("fourrooms_dr_s2")
    n = 6
    vec = MiniWorldVecEnv("MiniWorld-FourRooms-v0", n, domain_rand=True, seed=0, autoreset=False)
    o = vec.reset()
    assert np.array_equal(o[2].cpu().numpy(), obs[0]["rgb"])
    act = torch.zeros(n, dtype=torch.int32, device="cuda")
    for t in range(len(tr["action"])):
        act[:] = int(tr["action"][t])
        o, rew, term, trunc = vec.step(act)
        assert np.float32(tr["reward"][t]) == rew[2].item() and bool(term[2].item()) == bool(tr["term"][t]), t
        if (t + 1) in obs:
            assert np.array_equal(o[2].cpu().numpy(), obs[t + 1]["rgb"]), t + 1
    st = vec.engine.get_state()
    assert np.abs(st["agent_pos"][2] - tr["pos"][-1]).max() < 1e-12
    vec.engine.check()
    vec.close()


def test_step_accepts_int64_and_strided_actions_and_rejects_bad_outputs():
    """torch.randint / argmax / Categorical.sample give int64, a column of a [N, T] plan is strided: step() converts
    such action tensors instead of reading them as raw int32 words; output tensors of the wrong dtype are an error."""
    import torch
    from miniworld_amd.engine import EngineError
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n, T = 64, 12
    plan = torch.randint(0, 3, (n, T), generator=torch.Generator(device="cuda").manual_seed(4), device="cuda")     # int64 [N, T]
    a = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, seed=3)
    b = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, seed=3)
    a.reset(); b.reset()
    for t in range(T):
        a.step(plan[:, t].to(torch.int32).contiguous())
        if t % 3 == 0:
            b.step(plan[:, t])                      # int64, stride T
        elif t % 3 == 1:
            b.step(plan[:, t].cpu())                # on the host
        else:
            b.step(plan[:, t].to(torch.int16))
    assert torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward)
    with pytest.raises(EngineError):
        a.engine.step(plan[:, 0], a.obs, None, a.reward.double(), a.terminated, a.truncated)
    with pytest.raises(EngineError):
        a.engine.step(plan[:8, 0], a.obs, None, a.reward, a.terminated, a.truncated)
    a.close(); b.close()


def test_render_depth_honours_the_frame_buffer():
    """render_depth(frame_buffer) renders at that buffer's size and sample count (miniworld.py:1223-1236)."""
    import pyoracle
    from miniworld_amd import envs
    env = envs.OneRoom()
    env.reset(seed=3)
    d = env.render_depth(env.vis_fb)
    assert d.shape[:2] == (600, 800) and d.dtype == np.float32
    want = pyoracle.render(env.scene(), 800, 600, 16)["depth"]
    assert np.array_equal(d.reshape(600, 800), want.reshape(600, 800))
    env.close()


@pytest.mark.parametrize("msaa", [4, 1])
@pytest.mark.parametrize("env_id,kw", [("MiniWorld-Hallway-v0", {}), ("MiniWorld-PickupObjects-v0", {"domain_rand": True})])
def test_vec_env_fallback_sample_counts_match_oracle(env_id, kw, msaa):
    """FrameBuffer falls back to the driver's GL_MAX_SAMPLES when it is below 8 (opengl.py:229-231): mw_config.msaa = 4 / 1
    renders the batch with the D3D 4x pattern / the pixel centre; frames and depth equal the oracle's at that count."""
    import torch
    import pyoracle
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 24
    vec = MiniWorldVecEnv(env_id, n, seed=21, want_depth=True, msaa=msaa, **kw)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(3)
    for _ in range(10):
        vec.step(torch.randint(0, vec.n_actions, (n,), generator=g, device="cuda", dtype=torch.int32))
    vec.engine.render(vec.obs, vec.depth)         # the state after the steps (a pickup is removed after its frame)
    vec.engine.check()
    st = vec.engine.get_state()
    meshes = helpers.vec_env_meshes(vec)
    for i in (0, 5, n - 1):
        want = pyoracle.render(helpers.scene_of_vec_env(vec, st, i), nsamples=msaa, meshes=meshes)
        assert np.array_equal(vec.obs[i].cpu().numpy(), want["rgb"]), (env_id, msaa, i)
        assert np.array_equal(vec.depth[i].cpu().numpy(), want["depth"]), (env_id, msaa, i)
    vec.close()


@pytest.mark.gpu
@pytest.mark.parametrize("env_id,kw", [("MiniWorld-Maze-v0", {"max_episode_steps": 40}), ("MiniWorld-Maze-v0", {"obs_width": 160, "obs_height": 120, "max_episode_steps": 40}),
                                       ("MiniWorld-MazeS3-v0", {"domain_rand": True, "max_episode_steps": 40}), ("MiniWorld-FourRooms-v0", {})])
def test_occlusion_culling_never_changes_a_frame(env_id, kw, monkeypatch):
    """The geometry kernel of big scenes drops the room polygons that lie behind full-height walls before they cost a record
    (mw_geom.hip; MW_OCCLUSION=0 keeps them), and whole boxes of polygons outside the frustum, from data it keeps per world
    (MW_OCC_CACHE=0: none of it): a dropped polygon owns no sample, so RGB and depth are identical bit for
    bit, step after step, auto-resets — new worlds behind the same cache — included.  (With domain_rand the camera is pitched and the culling switches itself
    off; FourRooms is a small scene and never had it: both must still agree.)"""
    import torch
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 96
    frames = {}
    # "plain": without the per-world culling cache (no boxes of polygons, no occlusion test): every polygon sifted on its own
    for flag in ("0", "1", "plain"):
        monkeypatch.setenv("MW_OCCLUSION", "0" if flag == "plain" else flag)
        monkeypatch.setenv("MW_OCC_CACHE", "0" if flag == "plain" else "1")
        vec = MiniWorldVecEnv(env_id, n, seed=11, want_depth=True, **kw)
        vec.reset()
        out = [(vec.obs.cpu().numpy().copy(), vec.depth.cpu().numpy().copy())]
        g = torch.Generator(device="cuda").manual_seed(5)
        for t in range(60):
            vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
            out.append((vec.obs.cpu().numpy().copy(), vec.depth.cpu().numpy().copy()))
        vec.engine.check()
        vec.close()
        frames[flag] = out
    for other in ("1", "plain"):
        for t, ((o0, d0), (o1, d1)) in enumerate(zip(frames["0"], frames[other])):
            assert np.array_equal(o0, o1), (env_id, other, t, np.nonzero((o0 != o1).any(axis=(1, 2, 3)))[0][:8])
            assert np.array_equal(d0, d1), (env_id, other, t)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,open_plan,occlusion", [(6, False, "1"), (6, False, "0"), (8, False, "1"), (11, False, "1"),
                                                      (6, True, "1"), (9, True, "1"), (11, True, "1")])
def test_big_host_built_worlds_match_the_oracle(rows, open_plan, occlusion, monkeypatch):
    """Every branch of the big-scene K1 (mw_setup.hip): polygons cached in LDS or not (more than 512), occlusion culling
    or none, records written from the visible list or in place (max_visible above MW_SORT_CAP), rank sort (up to 256
    listed primitives), bitonic sort (up to 512), no visiting order (more) — grids of rooms built through the host API:
    joined by doors (lists of 32 - 255 primitives, tools/debug/grid_nvis.py) or open-plan and seen along the diagonal from a
    corner (6 x 6: 250 - 380; 9 x 9: 400 - 840; 11 x 11: up to 1250)."""
    import math
    import pyoracle
    from miniworld_amd.entity import Box
    from miniworld_amd.miniworld import MiniWorldEnv
    from miniworld_amd.scene import scene_from_env
    monkeypatch.setenv("MW_OCCLUSION", occlusion)
    cols = rows

    class Grid(MiniWorldEnv):
        def __init__(self, **kwargs):
            MiniWorldEnv.__init__(self, max_episode_steps=500, **kwargs)

        def _gen_world(self):
            rooms = [[self.add_rect_room(min_x=3.25 * i, max_x=3.25 * i + 3, min_z=3.25 * j, max_z=3.25 * j + 3) for i in range(cols)]
                     for j in range(rows)]
            for j in range(rows):
                for i in range(cols):
                    if open_plan:
                        if i + 1 < cols:
                            self.connect_rooms(rooms[j][i], rooms[j][i + 1], min_z=3.25 * j + 0.1, max_z=3.25 * j + 2.9)
                        if j + 1 < rows:
                            self.connect_rooms(rooms[j][i], rooms[j + 1][i], min_x=3.25 * i + 0.1, max_x=3.25 * i + 2.9)
                        continue
                    if i + 1 < cols and (i + j) % 3 != 0:
                        self.connect_rooms(rooms[j][i], rooms[j][i + 1], min_z=3.25 * j + 0.5, max_z=3.25 * j + 2.5,
                                           **({"max_y": 2.2} if (i + j) % 2 else {}))
                    if j + 1 < rows and (i * 2 + j) % 4 != 0:
                        self.connect_rooms(rooms[j][i], rooms[j + 1][i], min_x=3.25 * i + 0.75, max_x=3.25 * i + 2.25,
                                           **({"max_y": 2.2} if (i + j) % 2 == 0 else {}))
            self.box = self.place_entity(Box(color="red"))
            if open_plan:
                self.place_agent(pos=np.array([0.6, 0.0, 0.6]), dir=-math.pi / 4)
            else:
                self.place_agent()

    env = Grid()
    env.reset(seed=3)
    n_polys = len(scene_from_env(env)["polys_nv"])
    assert n_polys > {6: 250, 8: 512, 9: 768, 11: 768}[rows], n_polys
    g = np.random.default_rng(rows)
    for t in range(16 if open_plan else 24):
        o, *_ = env.step(int(g.choice([0, 1, 2, 2, 2])))
        want = pyoracle.render(scene_from_env(env))["rgb"]
        assert np.array_equal(o, want), (rows, t, int((o != want).any(-1).sum()))
    env.close()
