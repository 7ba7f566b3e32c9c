"""single-step clip vertices: which interpolation formula does the driver use?"""
from glprobe import *
import refshim_gl, itertools
sys.path.insert(0, "..")
from gl_feedback_check import gl_feedback
f32 = np.float32
one = f32(1)
planes = np.array([[-1,0,0,1],[1,0,0,1],[0,-1,0,1],[0,1,0,1],[0,0,1,1],[0,0,-1,1]], np.float32)
def dot4(c, p): return f32(f32(f32(f32(c[0]*p[0]) + f32(c[1]*p[1])) + f32(c[2]*p[2])) + f32(c[3]*p[3]))
def linterp(t, out, in_): return np.array([f32(o + f32(t * f32(i - o))) for o, i in zip(out, in_)], np.float32)
def matmul4(a, b):
    out = np.zeros(16, np.float32)
    for i in range(4):
        for j in range(4):
            out[j*4+i] = f32(f32(f32(f32(a[i]*b[j*4]) + f32(a[4+i]*b[j*4+1])) + f32(a[8+i]*b[j*4+2])) + f32(a[12+i]*b[j*4+3]))
    return out
for cls, seed in (("Hallway", 0), ("OneRoom", 0), ("FourRooms", 0), ("Hallway", 3)):
    env = refshim_gl.make_env(cls); env.reset(seed=seed)
    fb = gl_feedback(env)
    env.obs_fb.bind()
    mv = (c_float * 16)(); pr = (c_float * 16)()
    gl.glGetFloatv(gl.GL_MODELVIEW_MATRIX, mv); gl.glGetFloatv(gl.GL_PROJECTION_MATRIX, pr)
    MVP = matmul4(np.array(pr, np.float32), np.array(mv, np.float32))
    def xf(p): return np.array([f32(f32(f32(f32(p[0]*MVP[i]) + f32(p[1]*MVP[4+i])) + f32(p[2]*MVP[8+i])) + MVP[12+i]) for i in range(4)], np.float32)
    glverts = {}
    for t in fb:
        for k in range(3):
            glverts[(float(t[k, 8]), float(t[k, 9]))] = t[k]
    for room in env.rooms:
        polys = [(room.floor_verts, room.floor_texcs), (room.ceil_verts, room.ceil_texcs)]
        for q in range(room.wall_verts.shape[0] // 4):
            polys.append((room.wall_verts[4*q:4*q+4], room.wall_texcs[4*q:4*q+4]))
        for V, T in polys:
            V = V.astype(np.float32); T = T.astype(np.float32)
            Cc = [xf(v) for v in V]
            n = len(V)
            for a in range(n):
                b = (a + 1) % n
                for pi in range(4):
                    da, db = dot4(Cc[a], planes[pi]), dot4(Cc[b], planes[pi])
                    if (da >= 0) == (db >= 0): continue
                    # only consider if both endpoints are inside all OTHER lower planes (single step)
                    if any(dot4(Cc[x], planes[pj]) < 0 for x in (a, b) for pj in range(pi)): continue
                    O, I, dO, dI = (a, b, da, db) if da < 0 else (b, a, db, da)
                    cands = {
                        "O1": linterp(f32(dO / f32(dO - dI)), T[O], T[I]),
                        "O2": linterp(f32(one - f32(dI / f32(dI - dO))), T[O], T[I]),
                        "I1": linterp(f32(dI / f32(dI - dO)), T[I], T[O]),
                        "I2": linterp(f32(one - f32(dO / f32(dO - dI))), T[I], T[O]),
                    }
                    hits = [k for k, v in cands.items() if (float(v[0]), float(v[1])) in glverts]
                    allsame = len({(float(v[0]), float(v[1])) for v in cands.values()}) == 1
                    if not allsame:
                        print(cls, "edge", a, b, "plane", pi, "outside", O, "dO %.3f dI %.3f" % (dO, dI), "hits", hits)
