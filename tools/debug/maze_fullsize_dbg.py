"""Re-runs the C4 full-size parity case until the first frame mismatch and dumps everything needed to analyse it offline."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import helpers, pyoracle
import test_gpu_full_size_parity as T
from miniworld_amd import envs
from miniworld_amd.vec_env import MiniWorldVecEnv

name = sys.argv[1] if len(sys.argv) > 1 else "C4_maze_1024"
env_id, cls_name, n, depth, dr, n_act, task, fwd_bias, STEPS, kwargs = T.CONFIGS[name]
seed = 1000
vec = MiniWorldVecEnv(env_id, n, seed=seed, want_depth=depth, domain_rand=dr, **kwargs)
vec.reset()
pick = T._picked_envs(n)
pick_t = torch.tensor(pick, device="cuda")
mirrors = {i: helpers.EpisodeMirror(getattr(envs, cls_name), seed + i, dr, task, **kwargs) for i in pick}
g = torch.Generator(device="cuda").manual_seed(77)
actions = torch.randint(0, n_act, (STEPS, n), generator=g, device="cuda", dtype=torch.int32)
if fwd_bias:
    actions[torch.rand((STEPS, n), generator=g, device="cuda") < fwd_bias] = 2
act_host = actions[:, pick_t].cpu().numpy()
out = {}
nbad_total = 0
for t in range(STEPS):
    vec.step(actions[t])
    for j, i in enumerate(pick):
        mirrors[i].step(act_host[t, j])
    rgb = vec.obs[pick_t].cpu().numpy()
    st = None
    for j, i in enumerate(pick):
        m = mirrors[i]
        want = pyoracle.render(m.frame_scene(), meshes=m.meshes(), want_prim=True)
        bad = np.argwhere(rgb[j] != want["rgb"])
        if len(bad):
            nbad_total += 1
            if st is None:
                st = vec.engine.get_state()
            sc_dev = helpers.scene_of_vec_env(vec, st, i)
            want_dev = pyoracle.render(sc_dev, meshes=helpers.vec_env_meshes(vec))
            bad_dev = np.argwhere(rgb[j] != want_dev["rgb"])
            msc = m.frame_scene()
            geo_same = all(np.array_equal(sc_dev[k], msc[k]) for k in ("polys_v", "polys_uv", "polys_n", "polys_nv"))
            pos_same = np.array_equal(sc_dev["agent_pos"], msc["agent_pos"]) and sc_dev["agent_dir"] == msc["agent_dir"]
            epos_same = np.array_equal(sc_dev["ents_pos"][:1], msc["ents_pos"][:1])
            vec.engine.render(vec.obs, vec.depth)
            again = vec.obs[i].cpu().numpy()
            print(f"step {t} env {i} fresh={m.fresh}: {len(bad)} values differ vs mirror-oracle, {len(bad_dev)} vs device-state-oracle; "
                  f"geometry same {geo_same}, agent same {pos_same}, ent same {epos_same}; render-only again equals step frame: {np.array_equal(again, rgb[j])}, "
                  f"render-only vs oracle diffs {np.count_nonzero(again != want['rgb'])}")
            print("  where", bad[:6].tolist(), "got", [int(rgb[j][tuple(b)]) for b in bad[:6]], "want", [int(want['rgb'][tuple(b)]) for b in bad[:6]])
            tag = f"t{t}_e{i}"
            out[tag + "/got"] = rgb[j]; out[tag + "/want"] = want["rgb"]; out[tag + "/prim"] = want["prim"]
            for k, v in msc.items():
                out[tag + "/sc/" + k] = np.asarray(v)
            if nbad_total >= 6:
                break
    if nbad_total >= 6:
        break
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "fullsize_dbg.npz"), **out)
print("mismatching frames:", nbad_total)
