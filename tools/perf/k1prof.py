"""Per-phase cycle counts of K1 (MW_K1_PROF hook), Hallway headline config."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "k1prof.bin")
os.environ["MW_K1_PROF"] = out
from miniworld_amd.vec_env import MiniWorldVecEnv
n = 4096
vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, seed=0)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
for t in range(60):
    vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
vec.close()
d = np.fromfile(out, np.uint64).reshape(n, 8).astype(np.float64)
names = ["state + physics + rule (+ reset)", "camera", "rooms loop", "entities / batches"]
reg = d[:, 4] > 0
print("envs regenerated in the last step: %d" % reg.sum())
for k, nm in enumerate(names):
    v = d[~reg, k]
    print("%-34s mean %7.0f  p50 %7.0f  p99 %7.0f cycles" % (nm, v.mean(), np.percentile(v, 50), np.percentile(v, 99)))
print("total (not regenerated) mean %.0f; regenerated envs: phase 0 mean %.0f" % (d[~reg, :4].sum(1).mean(), d[reg, 0].mean() if reg.any() else 0))
