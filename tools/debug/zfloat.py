import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..")); sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import refshim_gl, refscene, pyoracle
refshim_gl.load_reference()
gl = refshim_gl.gl
import miniworld.opengl as ogl
cls, seed, t_end, px, py = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
env = refshim_gl.make_env(cls, domain_rand=True); env.reset(seed=seed)
rng = np.random.default_rng(seed)
for t in range(t_end + 1):
    env.step(int(rng.choice(3, p=[0.2, 0.2, 0.6])))
d16 = ogl.GL_DEPTH_COMPONENT16
ogl.GL_DEPTH_COMPONENT16 = gl.GL_DEPTH_COMPONENT32F
fb = ogl.FrameBuffer(80, 60, 8)
ogl.GL_DEPTH_COMPONENT16 = d16
env.render_obs(fb)
z = np.zeros((60, 80), np.float32)
gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, fb.final_fbo)
gl.glReadPixels(0, 0, 80, 60, gl.GL_DEPTH_COMPONENT, gl.GL_FLOAT, z.ctypes.data)
gy = 59 - py
print("GL float z at image (%d,%d) = gl row %d:" % (py, px, gy), float.hex(float(z[gy, px])), "%.9g" % z[gy, px], "*65535 = %.5f" % (float(z[gy, px]) * 65535))
env.render_obs()
from gen_gl_fixtures import read_z16, mesh_arrays
print("GL z16", read_z16(env, env.obs_fb)[py, px])
sc = refscene.scene_from_ref_env(env)
os.environ["MWO_DBG_PX"] = "%d,%d" % (px, gy)
r = pyoracle.render(sc, nsamples=4, meshes=mesh_arrays(env))
print("oracle z16", r["z16"][py, px])
import ctypes as C
zo = np.ones((60, 80), np.float32)
L = pyoracle.lib(); L.mwo_debug_set_zbuf.argtypes = [C.c_void_p]; L.mwo_debug_set_zbuf(zo.ctypes.data)
del os.environ["MWO_DBG_PX"]
r = pyoracle.render(sc, nsamples=4, meshes=mesh_arrays(env), want_prim=True)
L.mwo_debug_set_zbuf(None)
prim0 = r["prim"][::-1, :, 0]      # GL row order
diff = (z.view(np.int32).astype(np.int64) - zo.view(np.int32).astype(np.int64))
for pid in np.unique(prim0):
    m = prim0 == pid
    print("prim", pid, "pixels", m.sum(), "ulp diffs", np.unique(diff[m], return_counts=True))
