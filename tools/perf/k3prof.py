"""Per-env cycle counts of the mesh kernel (MW_K3_PROF hook) for the PickupObjects config."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "k3prof.bin")
os.environ["MW_K3_PROF"] = out
from miniworld_amd.vec_env import MiniWorldVecEnv
n = 2048
vec = MiniWorldVecEnv("MiniWorld-PickupObjects-v0", n, domain_rand=True, seed=0)
vec.reset()
g = torch.Generator(device="cuda").manual_seed(1234)
for t in range(80):
    vec.step(torch.randint(0, 5, (n,), generator=g, device="cuda", dtype=torch.int32))
torch.cuda.synchronize()
vec.close()
raw = np.fromfile(out, np.uint64).astype(np.float64)
d = raw[:n * 4].reshape(n, 4)
tp = raw[n * 4:].reshape(n, 4)
mesh, tile, nm, nt = d.T
print("envs with meshes in view: %.3f" % (nm > 0).mean())
for name, v in (("mesh phase", mesh), ("tile phase", tile)):
    print(name, "cycles: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (v.mean(), *np.percentile(v, [50, 90, 99]), v.max()))
for lo, hi in ((0, 0), (1, 1), (2, 9)):
    m = (nm >= lo) & (nm <= hi)
    if m.any():
        print(f"  n_mesh in [{lo},{hi}]: {m.mean():.3f} of envs, mesh phase {mesh[m].mean():.0f}, tile phase {tile[m].mean():.0f}, tris {nt[m].mean():.0f}")
big = nt >= 5000
print("  envs with a ball in view: %.3f, mesh phase %.0f, tile phase %.0f" % (big.mean(), mesh[big].mean(), tile[big].mean()))
w = np.argsort(mesh + tile)[-5:]
print("worst envs (mesh, tile, n_mesh, tris):", d[w].astype(int).tolist())
if tp.sum() > 0:       # MW_DEBUG_FLAGS != 0 (general kernel): accumulated over the run, per env: coverage cycles, shading cycles, shading iterations, exact tiles
    m = tp[:, 3] > 0
    print("exact tiles per env-frame (accumulated): coverage cycles/tile %.0f, shading cycles/tile %.0f, shading iterations/tile %.2f, cycles/iteration %.0f"
          % (tp[m, 0].sum() / tp[m, 3].sum(), tp[m, 1].sum() / tp[m, 3].sum(), tp[m, 2].sum() / tp[m, 3].sum(), tp[m, 1].sum() / tp[m, 2].sum()))
