"""The four BASELINE.json configurations at their real batch sizes, compared with the oracle.

Everything that only matters at full size — the XCD-aware block -> env mapping of the raster kernel, the mesh
kernel's heaviest-first block order and its co-running side stream, 4 waves per env in the big-scene setup kernel,
same-step auto-resets inside full launches — runs here with 4096 / 4096 / 1024 / 2048 envs for 300 steps of random
actions, and 64 envs spread over the first and last blocks and all 8 XCD residues are followed step by step by CPU
mirrors (helpers.EpisodeMirror: reference-exact world generator + oracle dynamics + numpy stream): rewards and flags
every step, poses / entity tables at check-points, and frames bit-exact against pyoracle.render of the MIRROR's
world — at check-points for all 64 and at every step on which one of them auto-reset (the frame returned with
`done` is the first of the next episode, drawn from a world the device generated inside the step kernel).
"""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu

#            env id, host class, envs, depth, domain_rand, n_actions, task, forward bias, steps, env kwargs
CONFIGS = {
    "C2_hallway_4096": ("MiniWorld-Hallway-v0", "Hallway", 4096, False, False, 3, 1, 0.0, 300, {}),
    "C3_oneroom_rgbd_4096": ("MiniWorld-OneRoom-v0", "OneRoom", 4096, True, False, 3, 1, 0.0, 300, {}),
    # a maze episode lasts up to 1536 steps: shortened so that every env regenerates its maze twice inside the run
    # (in-kernel auto-reset with the recursive backtracker at the full batch size); the default length is in the
    # fixtures (maze_s0, maze_s2) and in test_gpu_env_api.py
    "C4_maze_1024": ("MiniWorld-Maze-v0", "Maze", 1024, False, False, 3, 1, 0.5, 300, {"max_episode_steps": 130}),
    # PickupObjects truncates at 400 steps: 420 steps make the whole batch auto-reset in one launch
    "C5_pickup_dr_2048": ("MiniWorld-PickupObjects-v0", "PickupObjects", 2048, False, True, 5, 2, 0.3, 420, {}),
}


def _picked_envs(n):
    """64 envs: the first and the last 8 (first / last blocks, every XCD residue i mod 8), 48 spread over the batch
    with the residue rotating."""
    idx = set(range(8)) | set(range(n - 8, n))
    k = 0
    while len(idx) < 64:
        idx.add((k * (n // 48) + 8 + k % 8 + (k // 8) * 8) % n)
        k += 1
    idx = sorted(idx)
    assert {i % 8 for i in idx} == set(range(8))
    return idx


@pytest.mark.parametrize("name", list(CONFIGS))
def test_baseline_config_at_full_size_matches_oracle(name):
    import torch
    import pyoracle
    from miniworld_amd import envs
    from miniworld_amd.vec_env import MiniWorldVecEnv
    env_id, cls_name, n, depth, dr, n_act, task, fwd_bias, STEPS, kwargs = CONFIGS[name]
    CHECKPOINTS = (0, 1, STEPS // 2, STEPS - 1)
    seed = 1000
    vec = MiniWorldVecEnv(env_id, n, seed=seed, want_depth=depth, domain_rand=dr, **kwargs)
    assert vec.rng_mode == "pcg64"
    obs0 = vec.reset()
    pick = _picked_envs(n)
    pick_t = torch.tensor(pick, device="cuda")
    mirrors = {i: helpers.EpisodeMirror(getattr(envs, cls_name), seed + i, dr, task, **kwargs) for i in pick}
    mesh_cache = {}

    def check_frames(which, tag):
        rgb = vec.obs[torch.tensor(which, device="cuda")].cpu().numpy()
        dep = vec.depth[torch.tensor(which, device="cuda")].cpu().numpy() if depth else None
        for j, i in enumerate(which):
            m = mirrors[i]
            for mname, arrs in m.meshes().items():
                mesh_cache.setdefault(mname, arrs)
            want = pyoracle.render(m.frame_scene(), meshes=mesh_cache)
            nbad = np.count_nonzero(rgb[j] != want["rgb"])
            assert nbad == 0, f"{name} {tag} env {i}: {nbad} RGB values differ from the oracle"
            if depth:
                assert np.array_equal(dep[j], want["depth"]), f"{name} {tag} env {i}: depth differs"

    def check_state(tag):
        st = vec.engine.get_state()
        worst = 0.0
        for i in pick:
            pos, d, carrying, count, alive, epos, edir = mirrors[i].state()
            E = len(alive)
            assert int(st["step_count"][i]) == count and int(st["carrying"][i]) == carrying, (name, tag, i)
            assert np.array_equal(st["ent_kind"][i, :E] != 0, alive), (name, tag, i)
            worst = max(worst, np.abs(st["agent_pos"][i] - pos).max(), abs(st["agent_dir"][i] - d))
            if alive.any():
                worst = max(worst, np.abs(st["ent_pos"][i, :E][alive] - epos[alive]).max(),
                            np.abs(st["ent_dir"][i, :E][alive] - edir[alive]).max())
        assert worst < 1e-12, (name, tag, worst)

    check_state("reset")
    check_frames(pick, "reset")
    g = torch.Generator(device="cuda").manual_seed(77)
    actions = torch.randint(0, n_act, (STEPS, n), generator=g, device="cuda", dtype=torch.int32)
    if fwd_bias:
        actions[torch.rand((STEPS, n), generator=g, device="cuda") < fwd_bias] = 2
    act_host = actions[:, pick_t].cpu().numpy()
    n_done = n_reset_frames = 0
    for t in range(STEPS):
        obs, rew, term, trunc = vec.step(actions[t])
        got = torch.stack([rew[pick_t], term[pick_t].float(), trunc[pick_t].float()]).cpu().numpy()
        just_reset = []
        for j, i in enumerate(pick):
            r, te, tr = mirrors[i].step(act_host[t, j])
            assert np.float32(r) == got[0, j] and te == bool(got[1, j]) and tr == bool(got[2, j]), (name, t, i, r, te, tr, got[:, j])
            if te or tr:
                just_reset.append(i)
        n_done += len(just_reset)
        if t in CHECKPOINTS:
            check_state(f"step {t}")
            check_frames(pick, f"step {t}")
        elif just_reset and n_reset_frames < 96:
            check_frames(just_reset, f"auto-reset at step {t}")
            n_reset_frames += len(just_reset)
    vec.engine.check()
    # every env of the batch produced a real frame and the batch as a whole kept finishing episodes
    m = vec.obs.float().mean(dim=(1, 2, 3))
    assert (m > 5).all() and (m < 250).all()
    assert n_done >= 64, (name, n_done)              # every followed env went through at least one auto-reset
    vec.close()
