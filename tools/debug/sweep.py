import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import refshim_gl, refscene, pyoracle
sys.path.insert(0, HERE)
from gen_gl_fixtures import read_z16, mesh_arrays
cls = sys.argv[1]; nseeds = int(sys.argv[2]); every = int(sys.argv[3]); kwargs = {"domain_rand": True} if "dr" in sys.argv else {}
tot = dict(frames=0, rgb=0, z=0)
for seed in range(nseeds):
    env = refshim_gl.make_env(cls, **kwargs); env.reset(seed=seed)
    rng = np.random.default_rng(seed)
    for t in range(every * 4):
        _, _, term, trunc, _ = env.step(int(rng.choice(3, p=[0.2, 0.2, 0.6])))
        if term or trunc: break
        if t % every: continue
        sc = refscene.scene_from_ref_env(env)
        rgb = env.render_obs().copy(); z16 = read_z16(env, env.obs_fb)
        r = pyoracle.render(sc, nsamples=4, meshes=mesh_arrays(env), want_prim=True)
        bz = r["z16"] != z16; brgb = (r["rgb"] != rgb).any(axis=2)
        tot["frames"] += 1; tot["rgb"] += int(brgb.sum()); tot["z"] += int(bz.sum())
        for y, x in zip(*np.nonzero(bz | brgb)):
            print(cls, seed, t, "px", y, x, "z", z16[y, x], r["z16"][y, x], "rgb", rgb[y, x], r["rgb"][y, x], "prims", r["prim"][y, x], "npolys", len(sc["polys_nv"]))
print(tot)
