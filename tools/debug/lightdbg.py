from glprobe import *
import refshim_gl, itertools
sys.path.insert(0, "..")
from gl_feedback_check import gl_feedback, oracle_tris
f32 = np.float32
cls = sys.argv[1] if len(sys.argv) > 1 else "Maze"; seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
env = refshim_gl.make_env(cls); env.reset(seed=seed)
rng = np.random.default_rng(seed)
for _ in range(steps): env.step(int(rng.integers(0, 3)))
fb = gl_feedback(env); orc = oracle_tris(env)
env.obs_fb.bind()
mv = (c_float * 16)(); gl.glGetFloatv(gl.GL_MODELVIEW_MATRIX, mv); M = np.array(mv, np.float32)
lp = (c_float * 4)(); gl.glGetLightfv(gl.GL_LIGHT0, gl.GL_POSITION, lp); LP = np.array(lp, np.float32)
la = (c_float * 4)(); gl.glGetLightfv(gl.GL_LIGHT0, gl.GL_AMBIENT, la); LA = np.array(la, np.float32)
ld = (c_float * 4)(); gl.glGetLightfv(gl.GL_LIGHT0, gl.GL_DIFFUSE, ld); LD = np.array(ld, np.float32)
print("eye light pos", [float.hex(float(x)) for x in LP], "amb", LA, "dif", LD)
print("python light_pos", env.light_pos, "M", M.reshape(4, 4).T)
# collect (normal -> gl colour) for room polys via the oracle's matching
import refscene
sc = refscene.scene_from_ref_env(env)
j = 0
seen = {}
for t in fb:
    while np.abs(orc[j, :30].reshape(3, 10)[:, :2] - t[:, :2]).max() >= 0.05: j += 1
    d = int(orc[j, 31]); j += 1
    if d < len(sc["polys_n"]):
        n = tuple(sc["polys_n"][d]); seen[n] = (t[0, 4:7].copy(), orc[j - 1, 4:7].copy())
for n, (g, o) in seen.items():
    print("normal", n, "gl", [float.hex(float(x)) for x in g], "orc", [float.hex(float(x)) for x in o], "OK" if (g == o).all() else "BAD")
np.savez("/tmp/light.npz", M=M, LP=LP, LA=LA, LD=LD, normals=np.array(list(seen.keys()), np.float32), gl=np.array([v[0] for v in seen.values()], np.float32))
