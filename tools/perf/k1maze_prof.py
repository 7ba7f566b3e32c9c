"""Per-phase cycle counts of the big-scene K1 (MW_K1_PROF hook), Maze config: phases, visible polygons, occluders."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "k1maze_prof.bin")
os.environ["MW_K1_PROF"] = out
from miniworld_amd.vec_env import MiniWorldVecEnv
n = 1024
vec = MiniWorldVecEnv("MiniWorld-Maze-v0", n, seed=0)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
for t in range(60):
    vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
vec.close()
raw = np.fromfile(out, np.uint64).reshape(n, 8)
d = raw.astype(np.float64)
t_load = (raw[:, 6] >> np.uint64(8)).astype(np.float64); d[:, 6] = (raw[:, 6] & np.uint64(255)).astype(np.float64)
t_slab = ((raw[:, 7] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64); t_ident = (raw[:, 7] >> np.uint64(40)).astype(np.float64)
d[:, 7] = (raw[:, 7] & np.uint64(0xFFFFF)).astype(np.float64)
names = ["state + physics + rule (+ reset)", "camera", "rooms (occluders + list + records)", "entities + sort"]
reg = d[:, 4] > 0
print("envs regenerated in the last step: %d" % reg.sum())
for k, nm in enumerate(names):
    v = d[~reg, k]
    print("%-36s mean %7.0f  p50 %7.0f  p99 %7.0f cycles" % (nm, v.mean(), np.percentile(v, 50), np.percentile(v, 99)))
print("  of the rooms phase: vertices loaded at %.0f, slab at %.0f, occluders listed at %.0f, bins done at %.0f" % (t_load[~reg].mean(), t_slab[~reg].mean(), t_ident[~reg].mean(), d[~reg, 7].mean()))
print("total (not regenerated) mean %.0f" % d[~reg, :4].sum(1).mean())
for k, nm in ((5, "visible primitives"), (6, "occluder walls")):
    v = d[:, k]
    print("%-20s mean %.1f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f   <=32: %.3f  <=64: %.3f" % (nm, v.mean(), *np.percentile(v, [50, 90, 99]), v.max(), (v <= 32).mean(), (v <= 64).mean()))
