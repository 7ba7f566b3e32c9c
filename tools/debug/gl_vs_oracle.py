"""Scratch: reference on llvmpipe vs oracle(msaa=4) for one env / seed."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import refshim_gl, refscene, pyoracle

def compare(cls, seed, steps, top=False, verbose=True, kwargs={}):
    env = refshim_gl.make_env(cls, **kwargs)
    env.reset(seed=seed)
    rng = np.random.default_rng(seed)
    for _ in range(steps):
        env.step(int(rng.integers(0, 3)))
    gl = refshim_gl.gl
    if top:
        rgb = env.render_top_view(env.obs_fb)
    else:
        rgb = env.render_obs()
    z16 = np.zeros((60, 80), np.uint16)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, env.obs_fb.final_fbo)
    gl.glReadPixels(0, 0, 80, 60, gl.GL_DEPTH_COMPONENT, gl.GL_UNSIGNED_SHORT, z16.ctypes.data)
    z16 = z16[::-1]
    sc = refscene.scene_from_ref_env(env)
    meshes = {}
    for e in env.entities:
        if hasattr(e, "mesh"):
            meshes[refscene.mesh_name_of(e)] = refscene.ref_mesh_arrays(e.mesh)
    r = pyoracle.render(sc, nsamples=4, meshes=meshes, want_prim=True, view="top" if top else "agent", render_agent=top)
    d = np.abs(rgb.astype(int) - r["rgb"].astype(int)).max(axis=2)
    dz = z16.astype(int) - r["z16"].astype(int)
    res = {"rgb_hist": np.bincount(d.ravel(), minlength=4)[:8].tolist(), "rgb_max": int(d.max()), "z_bad": int((dz != 0).sum()),
           "z_max": int(np.abs(dz).max())}
    if verbose:
        print(cls, seed, steps, "top" if top else "", res)
        ys, xs = np.nonzero((d > 0) | (dz != 0))
        for y, x in list(zip(ys, xs))[:int(os.environ.get("SHOW", "8"))]:
            print("   px", y, x, "gl", rgb[y, x], "orc", r["rgb"][y, x], "z", z16[y, x], r["z16"][y, x], "prims", r["prim"][y, x])
    return res

if __name__ == "__main__":
    cls = sys.argv[1] if len(sys.argv) > 1 else "Hallway"
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    compare(cls, seed, steps, "top" in sys.argv)
