"""Phase cycle counts of the geometry kernel (MW_K1_PROF hook) for a BASELINE config.  usage: kgprof.py <config>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
out = "/tmp/kgprof.bin"
os.environ["MW_K1_PROF"] = out
import bench
from miniworld_amd.vec_env import MiniWorldVecEnv
cfg = sys.argv[1] if len(sys.argv) > 1 else "hallway"
env_id, _, n, depth, dr, n_act, *_ = bench.CONFIGS[cfg]
vec = MiniWorldVecEnv(env_id, n, domain_rand=dr, seed=0)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(40):
    vec.step(torch.randint(0, n_act, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
vec.close()
raw = np.fromfile(out, np.uint64).reshape(n, 32)
d = raw[:, :8].astype(np.float64)
names = ["xform", "walk/occ/sift", "vertex stage", "pass1 setups", "pass1 clip", "scan", "pass2", "tail"]
print(cfg, "cycles per phase (mean over envs, the rounds' phases summed over the rounds, last frame):")
for k, nm in enumerate(names):
    print("  %-14s %9.0f" % (nm, d[:, k].mean()))
print("  sum %.0f cycles = %.1f us at 2.4 GHz" % (d.sum(axis=1).mean(), d.sum(axis=1).mean() / 2400))
# the wavefronts in time (s_memrealtime, 100 MHz): when they start, how long they run — the kernel lasts until the slowest ends
st = raw[:, 8].astype(np.int64); en = raw[:, 9].astype(np.int64)
t0 = st.min(); st = (st - t0) / 100.0; en = (en - t0) / 100.0; dur = en - st
print("  wavefronts: start median %.1f max %.1f us; duration median %.1f p95 %.1f max %.1f us; last end %.1f us" %
      (np.median(st), st.max(), np.median(dur), np.percentile(dur, 95), dur.max(), en.max()))
h, edges = np.histogram(dur, bins=8)
print("  duration histogram (us):", ", ".join("%.0f-%.0f: %d" % (edges[i], edges[i + 1], h[i]) for i in range(len(h))))
slow = dur >= np.percentile(dur, 97)
print("  slowest 3 %%: phases %s" % " ".join("%s %.0f" % (names[k].split("/")[0], d[slow, k].mean()) for k in range(8)))

c = raw[:, 10:16].astype(np.int64)
for k, nm in enumerate(["polygons", "sifted polygons", "records", "rounds", "occluder walls", "boxes kept"]):
    print("  %-16s median %5d p90 %5d max %5d; slowest 3 %%: %5.0f" % (nm, np.median(c[:, k]), np.percentile(c[:, k], 90), c[:, k].max(), c[slow, k].mean()))
sd = raw[:, 16:20].astype(np.float64)
print("  inside the sift (cycles, mean / slowest 3 %): " + ", ".join("%s %.0f / %.0f" % (nm, sd[:, k].mean(), sd[slow, k].mean()) for k, nm in enumerate(["occluder walls", "column bins", "boxes", "polygons"])))
print("  duration vs rounds: " + ", ".join("%d rounds: %d envs %.0f us" % (r, (c[:, 3] == r).sum(), dur[c[:, 3] == r].mean()) for r in sorted(set(c[:, 3].tolist()))))

rr = raw[:, 20:24].astype(np.float64)
print("  rounds (cycles, mean / slowest 3 %%): first round %.0f / %.0f (its vertex stage %.0f / %.0f), second %.0f / %.0f, last round's vertex stage %.0f / %.0f" %
      (rr[:, 0].mean(), rr[slow, 0].mean(), rr[:, 2].mean(), rr[slow, 2].mean(), rr[:, 1].mean(), rr[slow, 1].mean(), rr[:, 3].mean(), rr[slow, 3].mean()))

cl = raw[:, 24:28].astype(np.float64)
print("  clipper (per wavefront, mean / slowest 3 %%): clipping %.0f / %.0f cycles, fans' setup + records %.0f / %.0f cycles; clipped triangles %.1f / %.1f, fan triangles %.1f / %.1f" %
      (cl[:, 0].mean(), cl[slow, 0].mean(), cl[:, 1].mean(), cl[slow, 1].mean(), cl[:, 2].mean(), cl[slow, 2].mean(), cl[:, 3].mean(), cl[slow, 3].mean()))

ee = raw[:, 28:31].astype(np.float64)
print("  inside the fans' pass (cycles per wavefront, mean): owner shuffles %.0f, setup %.0f, record stores %.0f" % (ee[:, 0].mean(), ee[:, 1].mean(), ee[:, 2].mean()))
