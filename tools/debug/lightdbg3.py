import numpy as np, itertools
f32 = np.float32
d = np.load("/tmp/light.npz")
M, LP, LA, LD = d["M"], d["LP"], d["LA"], d["LD"]
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
def mul(a, b): return f32(f32(a) * f32(b))
def add(a, b): return f32(f32(a) + f32(b))
# object-space light: Minv3 = transpose(M3): Q[i] = Minv[i]*P0 + Minv[4+i]*P1 + Minv[8+i]*P2 (column-major), Minv[col*4+row] = M[row*4+col]
Minv = np.zeros(16, np.float32)
for r in range(3):
    for c in range(3): Minv[c * 4 + r] = M[r * 4 + c]
lo = np.array([add(add(mul(Minv[i], LP[0]), mul(Minv[4 + i], LP[1])), mul(Minv[8 + i], LP[2])) for i in range(3)], np.float32)
print("light obj", lo)
s = add(add(mul(lo[0], lo[0]), mul(lo[1], lo[1])), mul(lo[2], lo[2]))
inv = f32(f32(1) / f32(np.sqrt(s)))
VP = np.array([mul(x, inv) for x in lo], np.float32)
print("VP", [float.hex(float(x)) for x in VP])
for ni, n in enumerate(d["normals"]):
    tgt = d["gl"][ni][0]
    dt = add(add(mul(n[0], VP[0]), mul(n[1], VP[1])), mul(n[2], VP[2]))
    d0 = dt if dt > 0 else f32(0)
    c = f32(1)
    amb, dif = LA[0], LD[0]
    cands = {
        "add(mad)": add(mul(d0, mul(dif, c)), add(mul(amb, c), mul(f32(0.2), c))),
        "fma": fma(d0, mul(dif, c), add(mul(amb, c), mul(f32(0.2), c))),
        "add2": add(mul(d0, mul(dif, c)), add(mul(f32(0.2), c), mul(amb, c))),
    }
    print(n, float.hex(float(tgt)), {k: (float.hex(float(min(v, f32(1)))), min(v, f32(1)) == tgt) for k, v in cands.items()})
