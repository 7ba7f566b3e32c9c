"""setup + interpolation arithmetic: float attribute planes read back exactly from an RGBA32F target."""
from glprobe import *
f32 = np.float32
W, H = 32, 24
rng = np.random.default_rng(5)
make_fbo(W, H)
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
def model(v, a, variant):
    """v: 3x4 window (x,y,z,oow) float32 ; a: 3 attribute values; returns predicted image (nan outside)"""
    # front-facing (ccw in y-up) -> setup order (v1, v0, v2)
    order = [1, 0, 2] if variant.get("swap", True) else [0, 1, 2]
    v0, v1, v2 = (v[i] for i in order); a0, a1, a2 = (f32(a[i] * v[i][3]) for i in order)
    w0, w1, w2 = v0[3], v1[3], v2[3]
    dx01, dy01, dx20, dy20 = f32(v0[0] - v1[0]), f32(v0[1] - v1[1]), f32(v2[0] - v0[0]), f32(v2[1] - v0[1])
    ooa = f32(f32(1) / f32(f32(dx01 * dy20) - f32(dx20 * dy01)))
    dy20o, dy01o, dx20o, dx01o = f32(dy20 * ooa), f32(dy01 * ooa), f32(dx20 * ooa), f32(dx01 * ooa)
    x0c, y0c = f32(v0[0] - f32(0.5)), f32(v0[1] - f32(0.5))
    def coef(b0, b1, b2):
        da01, da20 = f32(b0 - b1), f32(b2 - b0)
        dadx = f32(f32(da01 * dy20o) - f32(da20 * dy01o))
        dady = f32(f32(da20 * dx01o) - f32(da01 * dx20o))
        c0 = f32(b0 - f32(f32(dadx * x0c) + f32(dady * y0c)))
        return c0, dadx, dady
    pa, pw = coef(a0, a1, a2), coef(w0, w1, w2)
    out = np.zeros((H, W), np.float32)
    for y in range(H):
        for x in range(W):
            av = fma(pa[2], f32(y), fma(pa[1], f32(x), pa[0]))
            wv = fma(pw[2], f32(y), fma(pw[1], f32(x), pw[0]))
            out[y, x] = f32(av * f32(f32(1) / wv))
    return out
bad_total = 0
for k in range(30):
    reset_state()
    gl.glClearColor(-1, -1, -1, -1); gl.glClear(gl.GL_COLOR_BUFFER_BIT | gl.GL_DEPTH_BUFFER_BIT)
    # random ccw triangle inside the viewport, random w
    while True:
        p = rng.uniform(1, [W - 1, H - 1], (3, 2))
        area = (p[1, 0] - p[0, 0]) * (p[2, 1] - p[0, 1]) - (p[2, 0] - p[0, 0]) * (p[1, 1] - p[0, 1])
        if area > 20: break
    w = rng.uniform(0.5, 8, 3).astype(np.float32)
    ndc = np.stack([p[:, 0] / W * 2 - 1, p[:, 1] / H * 2 - 1], axis=1)
    clip = np.zeros((3, 4), np.float32)
    clip[:, 0] = (ndc[:, 0] * w).astype(np.float32); clip[:, 1] = (ndc[:, 1] * w).astype(np.float32)
    clip[:, 2] = (rng.uniform(-0.9, 0.9, 3) * w).astype(np.float32); clip[:, 3] = w
    attr = rng.uniform(0, 1, (3, 3)).astype(np.float32)
    gl.glBegin(gl.GL_TRIANGLES)
    for i in range(3):
        gl.glColor3f(*[float(x) for x in attr[i]]); gl.glVertex4f(*[float(x) for x in clip[i]])
    gl.glEnd(); gl.glFlush()
    img = read_rgba_f(W, H)
    # window coords as the vertex shader makes them
    win = np.zeros((3, 4), np.float32)
    for i in range(3):
        oow = f32(f32(1) / clip[i, 3])
        win[i, 0] = fma(f32(clip[i, 0] * oow), f32(W / 2), f32(W / 2))
        win[i, 1] = fma(f32(clip[i, 1] * oow), f32(H / 2), f32(H / 2))
        win[i, 2] = fma(f32(clip[i, 2] * oow), f32(0.5), f32(0.5)); win[i, 3] = oow
    cov = img[:, :, 3] >= 0
    res = {}
    for name, variant in (("swap", {"swap": True}), ("noswap", {"swap": False})):
        nb = 0
        for c in range(3):
            pred = model(win, attr[:, c], variant)
            nb += int((pred[cov].view(np.uint32) != img[:, :, c][cov].view(np.uint32)).sum())
        res[name] = nb
    print(k, "covered", int(cov.sum()), res)
