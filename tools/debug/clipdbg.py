from glprobe import *
import refshim_gl, itertools
f32 = np.float32
env = refshim_gl.make_env("Hallway"); env.reset(seed=0); env.render_obs(); env.obs_fb.bind()
mv = (c_float * 16)(); pr = (c_float * 16)()
gl.glGetFloatv(gl.GL_MODELVIEW_MATRIX, mv); gl.glGetFloatv(gl.GL_PROJECTION_MATRIX, pr)
MV = np.array(mv, np.float32); PR = np.array(pr, np.float32)
def matmul4(a, b):
    out = np.zeros(16, np.float32)
    for i in range(4):
        for j in range(4):
            out[j*4+i] = f32(f32(f32(f32(a[i]*b[j*4]) + f32(a[4+i]*b[j*4+1])) + f32(a[8+i]*b[j*4+2])) + f32(a[12+i]*b[j*4+3]))
    return out
MVP = matmul4(PR, MV)
room = env.rooms[0]
V = room.floor_verts.astype(np.float32); T = room.floor_texcs.astype(np.float32)
def xf(p):
    return np.array([f32(f32(f32(f32(p[0]*MVP[i]) + f32(p[1]*MVP[4+i])) + f32(p[2]*MVP[8+i])) + MVP[12+i]) for i in range(4)], np.float32)
C = [xf(v) for v in V]
for c, t in zip(C, T): print([float.hex(float(x)) for x in c], t)
# triangle (1,2,0): which vertices are outside which planes
planes = np.array([[-1,0,0,1],[1,0,0,1],[0,-1,0,1],[0,1,0,1],[0,0,1,1],[0,0,-1,1]], np.float32)
def dot4(c, p): return f32(f32(f32(f32(c[0]*p[0]) + f32(c[1]*p[1])) + f32(c[2]*p[2])) + f32(c[3]*p[3]))
for k, c in enumerate(C): print(k, [float(dot4(c, p)) for p in planes])
print("---- simulate clip of tri (V1,V2,V0)")
def linterp(t, out, in_): return np.array([f32(o + f32(t * f32(i - o))) for o, i in zip(out, in_)], np.float32)
def linterp_fma(t, out, in_): return np.array([f32(np.float64(o) + np.float64(t) * np.float64(f32(i - o))) for o, i in zip(out, in_)], np.float32)
def clip_poly(verts, lin):
    inl = list(verts)
    for pi, pl in enumerate(planes):
        n = len(inl)
        if n < 3: break
        dps = [dot4(v[0], pl) for v in inl]
        if all(d >= 0 for d in dps): continue
        out = []
        prev, dpp = inl[0], dps[0]
        for i in range(1, n + 1):
            v, dp = inl[i % n], dps[i % n]
            if dpp >= 0: out.append(prev)
            if (dp >= 0) != (dpp >= 0):
                if dp < 0:
                    t = f32(dp / f32(dp - dpp)); nv = (lin(t, v[0], prev[0]), lin(t, v[1], prev[1]))
                else:
                    t = f32(dpp / f32(dpp - dp)); nv = (lin(t, prev[0], v[0]), lin(t, prev[1], v[1]))
                out.append(nv)
            prev, dpp = v, dp
        inl = out
    return inl
tri = [(C[1], T[1]), (C[2], T[2]), (C[0], T[0])]
for name, lin in (("plain", linterp), ("fma", linterp_fma)):
    print(name)
    for c, t in clip_poly(tri, lin):
        print("   clip", [float.hex(float(x)) for x in c], "st", [float.hex(float(x)) for x in t])
print("---- direct edge V2->V0 against plane 3")
for (a, b, name) in ((2, 0, "V2->V0 (coming in: out=V2,in=V0)"), (0, 2, "V0->V2 going out: out=V2? ")):
    dpa, dpb = dot4(C[a], planes[3]), dot4(C[b], planes[3])
    # edge prev=a, vert=b
    if dpb < 0:
        t = f32(dpb / f32(dpb - dpa)); st = linterp(t, T[b], T[a]); cl = linterp(t, C[b], C[a])
    else:
        t = f32(dpa / f32(dpa - dpb)); st = linterp(t, T[a], T[b]); cl = linterp(t, C[a], C[b])
    print(name, "t", float.hex(float(t)), [float.hex(float(x)) for x in st], [float.hex(float(x)) for x in cl])
print("---- search plane orders")
target_C = (f32(float.fromhex('0x1.eee2c4p+2')), f32(float.fromhex('0x1.d25cb2p-1')))
target_D = (f32(float.fromhex('0x1.1b8f64p+3')), f32(float.fromhex('-0x1.8c6294p-1')))
import itertools
def clip_order(verts, lin, order):
    inl = list(verts)
    for pi in order:
        pl = planes[pi]
        n = len(inl)
        if n < 3: break
        dps = [dot4(v[0], pl) for v in inl]
        out = []
        prev, dpp = inl[0], dps[0]
        for i in range(1, n + 1):
            v, dp = inl[i % n], dps[i % n]
            if dpp >= 0: out.append(prev)
            if (dp >= 0) != (dpp >= 0):
                if dp < 0:
                    t = f32(dp / f32(dp - dpp)); nv = (lin(t, v[0], prev[0]), lin(t, v[1], prev[1]))
                else:
                    t = f32(dpp / f32(dpp - dp)); nv = (lin(t, prev[0], v[0]), lin(t, prev[1], v[1]))
                out.append(nv)
            prev, dpp = v, dp
        inl = out
    return inl
for order in itertools.permutations(range(5)):
    for name, lin in (("plain", linterp), ("fma", linterp_fma)):
        res = clip_order(tri, lin, order)
        sts = [(v[1][0], v[1][1]) for v in res]
        if target_C in sts and target_D in sts:
            print("MATCH", order, name)
from fractions import Fraction as Fr
def fr(x): return Fr(float(x))
dp2 = fr(C[2][1]) + fr(C[2][3]); dp0 = fr(C[0][1]) + fr(C[0][3])
t = dp2 / (dp2 - dp0)
s = fr(T[2][0]) + t * (fr(T[0][0]) - fr(T[2][0])); tt = fr(T[2][1]) + t * (fr(T[0][1]) - fr(T[2][1]))
print("exact C st", float(s), float.hex(float(f32(float(s)))), float.hex(float(f32(float(tt)))))
# exact corner D: intersection of planes x+w=0, y+w=0 with the triangle's plane: param on triangle
print("---- attribute lerp variants")
def clip_var(verts, lin_pos, lin_attr):
    inl = list(verts)
    for pi in range(6):
        pl = planes[pi]
        n = len(inl)
        if n < 3: break
        dps = [dot4(v[0], pl) for v in inl]
        out = []
        prev, dpp = inl[0], dps[0]
        for i in range(1, n + 1):
            v, dp = inl[i % n], dps[i % n]
            if dpp >= 0: out.append(prev)
            if (dp >= 0) != (dpp >= 0):
                if dp < 0:
                    t = f32(dp / f32(dp - dpp)); nv = (lin_pos(t, v[0], prev[0]), lin_attr(t, v[1], prev[1]))
                else:
                    t = f32(dpp / f32(dpp - dp)); nv = (lin_pos(t, prev[0], v[0]), lin_attr(t, prev[1], v[1]))
                out.append(nv)
            prev, dpp = v, dp
        inl = out
    return inl
one = f32(1)
variants = {
 "out+t*(in-out)": lambda t, o, i: np.array([f32(a + f32(t * f32(b - a))) for a, b in zip(o, i)], np.float32),
 "fma(t,in-out,out)": linterp_fma,
 "(1-t)*out+t*in": lambda t, o, i: np.array([f32(f32(f32(one - t) * a) + f32(t * b)) for a, b in zip(o, i)], np.float32),
 "in+(1-t)*(out-in)": lambda t, o, i: np.array([f32(b + f32(f32(one - t) * f32(a - b))) for a, b in zip(o, i)], np.float32),
 "out-t*(out-in)": lambda t, o, i: np.array([f32(a - f32(t * f32(a - b))) for a, b in zip(o, i)], np.float32),
 "fma(t,in,fma(-t,out,out))": lambda t, o, i: np.array([f32(np.float64(t) * np.float64(b) + np.float64(f32(np.float64(a) - np.float64(t) * np.float64(a)))) for a, b in zip(o, i)], np.float32),
}
for name, f in variants.items():
    res = clip_var(tri, linterp, f)
    sts = [(v[1][0], v[1][1]) for v in res]
    print(name, "C" if target_C in sts else "-", "D" if target_D in sts else "-", [float.hex(float(x)) for x in sts[0]])
print("---- inspect chain for C")
# plane 0 on edge V2->V0 (coming in)
d2, d0 = dot4(C[2], planes[0]), dot4(C[0], planes[0])
t0 = f32(d2 / f32(d2 - d0))
N_c = linterp(t0, C[2], C[0]); N_s = linterp(t0, T[2], T[0])
print("N20 clip", [float.hex(float(x)) for x in N_c], "st", [float.hex(float(x)) for x in N_s], "t0", float.hex(float(t0)))
# plane 3 on edge N20 -> V0 (coming in) 
dn, d0 = dot4(N_c, planes[3]), dot4(C[0], planes[3])
print("dp N", float.hex(float(dn)), "dp V0", float.hex(float(d0)))
t3 = f32(dn / f32(dn - d0))
print("t3", float.hex(float(t3)), [float.hex(float(x)) for x in linterp(t3, N_s, T[0])])
# which t3 values reproduce GL?
for dt in range(-4, 5):
    tt = np.nextafter(t3, f32(2)) if dt > 0 else t3
    tt = t3
    for _ in range(abs(dt)): tt = np.nextafter(tt, f32(2) if dt > 0 else f32(-2))
    r = linterp(tt, N_s, T[0])
    print(dt, float.hex(float(tt)), [float.hex(float(x)) for x in r], "<<<" if (r[0], r[1]) == target_C else "")
print("---- coming-in variants")
def clip_var2(verts, variant):
    inl = list(verts)
    for pi in range(6):
        pl = planes[pi]
        n = len(inl)
        if n < 3: break
        dps = [dot4(v[0], pl) for v in inl]
        out = []
        prev, dpp = inl[0], dps[0]
        for i in range(1, n + 1):
            v, dp = inl[i % n], dps[i % n]
            if dpp >= 0: out.append(prev)
            if (dp >= 0) != (dpp >= 0):
                if dp < 0 or variant == 1:
                    t = f32(dp / f32(dp - dpp)); nv = (linterp(t, v[0], prev[0]), linterp(t, v[1], prev[1]))
                else:
                    t = f32(dpp / f32(dpp - dp)); nv = (linterp(t, prev[0], v[0]), linterp(t, prev[1], v[1]))
                out.append(nv)
            prev, dpp = v, dp
        inl = out
    return inl
for variant in (0, 1):
    res = clip_var2(tri, variant)
    sts = [(v[1][0], v[1][1]) for v in res]
    print(variant, "C" if target_C in sts else "-", "D" if target_D in sts else "-", [[float.hex(float(x)) for x in s] for s in sts])
print("---- 4x4 variants")
def mk(kind, v, prev, dp, dpp):
    if kind == "a": t = f32(dpp / f32(dpp - dp)); return lambda X, Y: linterp(t, Y, X)          # prev + t (v - prev)
    if kind == "b": t = f32(dp / f32(dp - dpp)); return lambda X, Y: linterp(t, X, Y)           # v + t (prev - v)
    if kind == "c": t = f32(one - f32(dpp / f32(dpp - dp))); return lambda X, Y: linterp(t, X, Y)
    if kind == "d": t = f32(one - f32(dp / f32(dp - dpp))); return lambda X, Y: linterp(t, Y, X)
def clip_var3(verts, kin, kout):
    inl = list(verts)
    for pi in range(6):
        pl = planes[pi]
        n = len(inl)
        if n < 3: break
        dps = [dot4(v[0], pl) for v in inl]
        out = []
        prev, dpp = inl[0], dps[0]
        for i in range(1, n + 1):
            v, dp = inl[i % n], dps[i % n]
            if dpp >= 0: out.append(prev)
            if (dp >= 0) != (dpp >= 0):
                f = mk(kout if dp < 0 else kin, v, prev, dp, dpp)
                out.append((f(v[0], prev[0]), f(v[1], prev[1])))
            prev, dpp = v, dp
        inl = out
    return inl
target_B = (f32(11.0), f32(float.fromhex('-0x1.d4a12cp-1')))
for kin in "abcd":
    for kout in "abcd":
        res = clip_var3(tri, kin, kout)
        sts = [(v[1][0], v[1][1]) for v in res]
        print(kin, kout, "B" if target_B in sts else "-", "C" if target_C in sts else "-", "D" if target_D in sts else "-")
print("---- both triangles, more kinds")
def mk2(kind, v, prev, dp, dpp):
    # O = outside vertex, I = inside vertex
    if dp < 0: O, I, dO, dI = 0, 1, dp, dpp      # X=v is outside
    else: O, I, dO, dI = 1, 0, dpp, dp
    def pick(X, Y): return (X, Y) if O == 0 else (Y, X)   # returns (outside, inside)
    if kind == "O1": t = f32(dO / f32(dO - dI)); return lambda X, Y: linterp(t, *pick(X, Y))
    if kind == "O2": t = f32(one - f32(dI / f32(dI - dO))); return lambda X, Y: linterp(t, *pick(X, Y))
    if kind == "I1": t = f32(dI / f32(dI - dO)); return lambda X, Y: linterp(t, *pick(X, Y)[::-1])
    if kind == "I2": t = f32(one - f32(dO / f32(dO - dI))); return lambda X, Y: linterp(t, *pick(X, Y)[::-1])
def clip_var4(verts, kin, kout):
    inl = list(verts)
    for pi in range(6):
        pl = planes[pi]
        n = len(inl)
        if n < 3: break
        dps = [dot4(v[0], pl) for v in inl]
        out = []
        prev, dpp = inl[0], dps[0]
        for i in range(1, n + 1):
            v, dp = inl[i % n], dps[i % n]
            if dpp >= 0: out.append(prev)
            if (dp >= 0) != (dpp >= 0):
                f = mk2(kout if dp < 0 else kin, v, prev, dp, dpp)
                out.append((f(v[0], prev[0]), f(v[1], prev[1])))
            prev, dpp = v, dp
        inl = out
    return inl
tri2 = [(C[2], T[2]), (C[3], T[3]), (C[0], T[0])]
target_E = (f32(float.fromhex('0x1.c0326cp+2')), f32(2.0))
kinds = ["O1", "O2", "I1", "I2"]
for kin in kinds:
    for kout in kinds:
        r1 = [(v[1][0], v[1][1]) for v in clip_var4(tri, kin, kout)]
        r2 = [(v[1][0], v[1][1]) for v in clip_var4(tri2, kin, kout)]
        print(kin, kout, "".join(n if tg in r1 else "-" for n, tg in (("B", target_B), ("C", target_C), ("D", target_D))),
              "".join(n if tg in r2 else "-" for n, tg in (("C", target_C), ("E", target_E))))
