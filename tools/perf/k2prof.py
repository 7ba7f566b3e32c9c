"""Phase cycles and event counts of the tile raster kernel (mw_raster_common.h, MW_PERF_HOOKS build) for a BASELINE config.
usage (GPU box): MW_ENGINE_LIB=miniworld_amd/csrc/_variants/libmwengine_perf.so python tools/perf/k2prof.py [maze]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from miniworld_amd import engine
from miniworld_amd.vec_env import MiniWorldVecEnv
cfg = sys.argv[1] if len(sys.argv) > 1 else "maze"
env_id, _, n, depth, dr, n_act, *_ = bench.CONFIGS[cfg]
vec = MiniWorldVecEnv(env_id, n, domain_rand=dr, seed=0)
vec.reset()
lib = engine.load_library()
out = (C.c_ulonglong * 16)()
g = torch.Generator(device="cuda").manual_seed(1)
for t in range(60):
    vec.step(torch.randint(0, n_act, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
assert lib.mw_debug_k2prof(out) == 0
K = 40
for t in range(K):
    vec.step(torch.randint(0, n_act, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
assert lib.mw_debug_k2prof(out) == 0
v = np.array(list(out), dtype=np.float64)
nv = vec.engine.list_lengths()
vec.close()
tiles = v[0]
print(cfg, "tile kernel, per tile (mean over %d tiles of %d steps):" % (tiles, K))
cyc = v[1:5]
for nm, c in zip(["classification", "coverage+depth loop", "shading loop", "resolve+pack+store"], cyc):
    print("  %-20s %8.0f cycles  %5.1f %%" % (nm, c / tiles, 100 * c / cyc.sum()))
print("  events visited %.2f, covering %.2f, winners shaded %.2f, chunks classified %.2f, tiles left early %.2f" %
      (v[5] / tiles, v[6] / tiles, v[7] / tiles, v[8] / tiles, v[9] / tiles))
print("  cycles per event %.0f, per winner %.0f, per chunk %.0f" % (v[2] / max(v[5], 1), v[3] / max(v[7], 1), v[1] / max(v[8], 1)))
print("  list lengths: median %d p90 %d max %d" % (np.median(nv), np.percentile(nv, 90), nv.max()))
if v[14]:
    print("  per wavefront item: %.0f cycles before the first tile (records staged); mesh tiles: key fetch %.0f cycles, %.2f per-lane mesh winner turns of %.0f cycles each" %
          (v[10] / v[14], v[11] / tiles, v[12] / tiles, v[13] / max(v[12], 1)))
if v[15]:
    print("  shader clock over the tiles' coverage + shading phases: %.0f MHz (s_memtime cycles / s_memrealtime ticks x 100 MHz)" % ((v[1] + v[2] + v[3]) / v[15] * 100.0))
if cfg != "maze":
    print("  (mesh tiles: slot 9 counts the most distinct winners any lane holds: %.2f per tile — the turns of a per-lane winner loop)" % (v[9] / tiles))
