"""Display-list lengths (triangles per env after clipping / culling) of the BASELINE configs: which raster kernel can take them."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from miniworld_amd.vec_env import MiniWorldVecEnv
for cfg in sys.argv[1:] or ["hallway", "oneroom_rgbd", "maze", "pickup_dr"]:
    env_id, _, n, depth, dr, n_act, *_ = bench.CONFIGS[cfg]
    n = min(n, 1024)
    vec = MiniWorldVecEnv(env_id, n, domain_rand=dr, seed=0)
    vec.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    allv = []
    for t in range(60):
        vec.step(torch.randint(0, n_act, (n,), generator=g, device="cuda", dtype=torch.int32))
        if t % 10 == 9:
            allv.append(vec.engine.list_lengths())
    v = np.concatenate(allv)
    print(cfg, "max_vis", vec.engine.cfg.max_visible * 6, "triangles per env: mean %.1f median %d p90 %d p99 %d max %d; <=40: %.1f %%, <=48: %.1f %%, <=64: %.1f %%, <=96: %.1f %%, <=128: %.1f %%" % (
        v.mean(), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), 100 * (v <= 40).mean(), 100 * (v <= 48).mean(), 100 * (v <= 64).mean(), 100 * (v <= 96).mean(), 100 * (v <= 128).mean()))
    vec.close()
