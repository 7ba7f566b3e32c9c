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
    env = getattr(envs, str(meta["env"]))(domain_rand=bool(meta["domain_rand"]))
    o, info = env.reset(seed=int(meta["seed"]))
    assert o.shape == env.observation_space.shape and o.dtype == np.uint8
    assert np.array_equal(o, obs[0]["rgb"])
    for t in range(len(tr["action"])):
        o, r, te, tu, info = env.step(int(tr["action"][t]))
        assert r == tr["reward"][t] and te == bool(tr["term"][t]) and tu == bool(tr["trunc"][t]), (case, t)
        assert np.abs(env.agent.pos - tr["pos"][t]).max() < 1e-12 and abs(env.agent.dir - tr["dir"][t]) < 1e-12
        if (t + 1) in obs:
            assert np.array_equal(o, obs[t + 1]["rgb"]), (case, t + 1)
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


def test_vec_env_maze_host_generated_per_env_geometry():
    """Maze: per-env geometry (510 polygons, 256 segments each), host world generation, frames == oracle."""
    import torch
    import pyoracle
    from miniworld_amd.scene import scene_from_env
    from miniworld_amd.vec_env import MiniWorldVecEnv
    n = 8
    vec = MiniWorldVecEnv("MiniWorld-Maze-v0", n, seed=0)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(2)
    for t in range(40):
        vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
    vec.engine.check()
    st = vec.engine.get_state()
    for i in (0, 3, n - 1):
        sc = scene_from_env(vec._host_envs[i])
        sc["agent_pos"], sc["agent_dir"] = st["agent_pos"][i], st["agent_dir"][i]
        want = pyoracle.render(sc)
        assert np.array_equal(vec.obs[i].cpu().numpy(), want["rgb"]), f"env {i}"
    vec.close()
