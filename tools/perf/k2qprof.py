"""Phase timeline of the quad raster kernel (mw_rasterq.hip): s_memtime stamps of every wavefront at the phase boundaries.
usage (GPU box): python tools/perf/k2qprof.py [config]   (runs a short bench with MW_K2Q_PROF set and summarises the dump)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = sys.argv[1] if len(sys.argv) > 1 else "hallway"
dump = "/tmp/k2qprof.bin"
env = dict(os.environ, MW_K2Q_PROF=dump)
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "12", "--warmup", "4", "--no-cpu-baseline",
                "--no-parity-check"], env=env, check=True, stdout=subprocess.DEVNULL)
raw = np.fromfile(dump, np.uint64)
n_env = raw.size // 80
t = raw[:n_env * 64].reshape(-1, 8, 8).astype(np.int64)      # [env][wave][stamp], shader clock
cls = raw[n_env * 64:].reshape(n_env, 16)[:, :9].astype(np.int64)
print('quads per class, mean per env (FALLBACK BIG EXACT P4 P3 P2 P1 TRIV SKY):', np.round(cls.mean(axis=0), 1).tolist())
names = ["stage", "A tiles", "B quads", "C classes", "D batches", "wait", "D2 exact", "E(rgb)"]
d = np.diff(t, axis=2)
print("per-wave phase durations, us (s_memtime ticks / 100): median / mean / p95 over wavefronts")
for k in range(7):
    v = d[:, :, k].reshape(-1) / 100.0
    print(f"  {names[k]:10s} {np.median(v):7.2f} {v.mean():7.2f} {np.percentile(v, 95):7.2f}")
blk = (t[:, :, 7].max(axis=1) - t[:, :, 0].min(axis=1)) / 100.0
print(f"block lifetime (first stamp to last E start): median {np.median(blk):.2f} us, mean {blk.mean():.2f}, p95 {np.percentile(blk, 95):.2f}")
span = (t[:, :, 7].max() - t[:, :, 0].min()) / 100.0
print(f"kernel span {span:.1f} us; blocks {t.shape[0]}; sum of block lifetimes / span = {blk.sum() / span:.1f} blocks in flight")
