"""Per-phase cycle counts and launch span of the dense K1 (MW_K1_PROF hook), Hallway headline config."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "k1prof.bin")
os.environ["MW_K1_PROF"] = out
from miniworld_amd.vec_env import MiniWorldVecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
auto = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
vec = MiniWorldVecEnv("MiniWorld-Hallway-v0", n, seed=0, autoreset=auto)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
for t in range(60):
    vec.step(torch.randint(0, 3, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
vec.close()
d = np.fromfile(out, np.uint64).reshape(n, 8)
names = ["state + physics + rule + writes", "auto-reset", "primitive data (loads, box sincos)", "camera + xform + cull + light", "compaction + records + header"]
reg = d[:, 5] > 0
print("envs %d autoreset %s; regenerated in the last step: %d" % (n, auto, reg.sum()))
for k, nm in enumerate(names):
    v = d[:, k].astype(np.float64)
    print("%-34s mean %7.0f  p50 %7.0f  p99 %7.0f  max %7.0f cycles" % (nm, v.mean(), np.percentile(v, 50), np.percentile(v, 99), v.max()))
tot = d[:, :5].sum(1).astype(np.float64)
print("total per wave: mean %.0f p50 %.0f p99 %.0f max %.0f cycles" % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 99), tot.max()))
w0, w1 = d[:, 6].astype(np.int64), d[:, 7].astype(np.int64)
t0 = w0.min()
print("wall clock (100 MHz ticks -> us): first start 0, last start %.2f, first end %.2f, last end %.2f" %
      ((w0.max() - t0) / 100.0, (w1.min() - t0) / 100.0, (w1.max() - t0) / 100.0))
dur = (w1 - w0) / 100.0
print("per-wave duration us: mean %.2f p50 %.2f p99 %.2f max %.2f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 99), dur.max()))
starts = np.sort((w0 - t0) / 100.0)
print("start time percentiles us: p10 %.2f p50 %.2f p90 %.2f p99 %.2f" % tuple(np.percentile(starts, [10, 50, 90, 99])))
