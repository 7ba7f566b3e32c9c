import numpy as np, itertools, struct
f32 = np.float32
d = np.load("/tmp/light.npz")
M, LP, LA, LD = d["M"], d["LP"], d["LA"], d["LD"]
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
def mul(a, b): return f32(f32(a) * f32(b))
def add(a, b): return f32(f32(a) + f32(b))
# VP normalisation variants
def norm_variants(v):
    out = {}
    for sumname, s in (("lr", add(add(mul(v[0], v[0]), mul(v[1], v[1])), mul(v[2], v[2]))),
                       ("fma", fma(v[2], v[2], fma(v[1], v[1], mul(v[0], v[0])))),
                       ("rl", add(mul(v[0], v[0]), add(mul(v[1], v[1]), mul(v[2], v[2]))))):
        out[sumname + ":1/sqrt*"] = np.array([mul(x, f32(1) / f32(np.sqrt(s))) for x in v], np.float32)
        out[sumname + ":/sqrt"] = np.array([f32(x / f32(np.sqrt(s))) for x in v], np.float32)
        out[sumname + ":rsq64"] = np.array([mul(x, f32(1.0 / np.sqrt(np.float64(s)))) for x in v], np.float32)
    return out
def dot_variants(a, b):
    p = [mul(a[i], b[i]) for i in range(3)]
    return {"lr": add(add(p[0], p[1]), p[2]), "rl": add(add(p[2], p[1]), p[0]), "fma_lr": fma(a[2], b[2], fma(a[1], b[1], p[0])),
            "fma_rl": fma(a[0], b[0], fma(a[1], b[1], p[2])), "x+(y+z)": add(p[0], add(p[1], p[2]))}
def normal_variants(n):
    # eye normal = (M^-T) n ; for a near-rotation M^-T ~ M3x3
    R = M.reshape(4, 4).T[:3, :3]
    out = {}
    for name, mat in (("M", R), ):
        for dn, f in (("lr", lambda r: add(add(mul(r[0], n[0]), mul(r[1], n[1])), mul(r[2], n[2]))),
                      ("fma", lambda r: fma(r[2], n[2], fma(r[1], n[1], mul(r[0], n[0]))))):
            out[name + ":" + dn] = np.array([f(mat[i]) for i in range(3)], np.float32)
    # true inverse transpose in float64 then rounded
    Ri = np.linalg.inv(R.astype(np.float64)).T
    out["inv64"] = np.array([f32(np.float64(Ri[i] @ n.astype(np.float64))) for i in range(3)], np.float32)
    return out
def colour_variants(dt, c=f32(1)):
    d0 = dt if dt > 0 else f32(0)
    amb, dif = LA[0], LD[0]
    return {
        "add(mad)": add(mul(d0, mul(dif, c)), add(mul(amb, c), mul(f32(0.2), c))),
        "fma": fma(d0, mul(dif, c), add(mul(amb, c), mul(f32(0.2), c))),
        "fma2": fma(d0, mul(dif, c), fma(f32(0.2), c, mul(amb, c))),
        "scene_fma": fma(d0, mul(dif, c), add(mul(amb, c), fma(f32(0.2), c, f32(0)))),
        "add2": add(mul(d0, mul(dif, c)), add(mul(f32(0.2), c), mul(amb, c))),
    }
for ni, n in enumerate(d["normals"]):
    tgt = d["gl"][ni][0]
    hits = []
    for (nvn, VP), (nn, NE) in itertools.product(norm_variants(LP[:3]).items(), normal_variants(n).items()):
        for dn, dt in dot_variants(NE, VP).items():
            for cn, col in colour_variants(dt).items():
                col = min(max(col, f32(0)), f32(1))
                if col == tgt: hits.append((nvn, nn, dn, cn))
    print(n, float.hex(float(tgt)), len(hits))
    if ni == 3:
        for h in hits: print("   ", h)
