from glprobe import *
W, H = 32, 24
f32 = np.float32
exec(open("probe_interp.py").read().split("bad_total = 0")[0].split("make_fbo(W, H)")[1])
ms = make_fbo(W, H, samples=4)
ss = make_fbo(W, H)
rng = np.random.default_rng(7)
for k in range(12):
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, ms)
    reset_state()
    gl.glEnable(gl.GL_MULTISAMPLE)
    gl.glDisable(gl.GL_SAMPLE_MASK)
    gl.glClearColor(0, 0, 0, 0); gl.glClear(gl.GL_COLOR_BUFFER_BIT | gl.GL_DEPTH_BUFFER_BIT)
    gl.glEnable(gl.GL_SAMPLE_MASK); gl.glSampleMaski(0, 1 << (k % 4))
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
        gl.glColor4f(*[float(x) for x in attr[i]], 1.0); gl.glVertex4f(*[float(x) for x in clip[i]])
    gl.glEnd(); gl.glFlush()
    gl.glBindFramebuffer(gl.GL_READ_FRAMEBUFFER, ms); gl.glBindFramebuffer(gl.GL_DRAW_FRAMEBUFFER, ss)
    gl.glBlitFramebuffer(0, 0, W, H, 0, 0, W, H, gl.GL_COLOR_BUFFER_BIT, gl.GL_LINEAR)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, ss)
    img = read_rgba_f(W, H) * 4
    win = np.zeros((3, 4), np.float32)
    for i in range(3):
        oow = f32(f32(1) / clip[i, 3])
        win[i, 0] = fma(f32(clip[i, 0] * oow), f32(W / 2), f32(W / 2))
        win[i, 1] = fma(f32(clip[i, 1] * oow), f32(H / 2), f32(H / 2))
        win[i, 2] = fma(f32(clip[i, 2] * oow), f32(0.5), f32(0.5)); win[i, 3] = oow
    cov = img[:, :, 3] > 0.5
    nb = 0
    for c in range(3):
        pred = model(win, attr[:, c], {"swap": True})
        nb += int((pred[cov].view(np.uint32) != img[:, :, c][cov].view(np.uint32)).sum())
    print(k, "sample", k % 4, "covered", int(cov.sum()), "mismatch", nb)
print("---- variants on the last triangle")
def model2(v, a, xoff_center, eval_off, order3=False):
    order = [1, 0, 2]
    v0, v1, v2 = (v[i] for i in order); a0, a1, a2 = (f32(a[i] * v[i][3]) for i in order)
    w0, w1, w2 = v0[3], v1[3], v2[3]
    dx01, dy01, dx20, dy20 = f32(v0[0] - v1[0]), f32(v0[1] - v1[1]), f32(v2[0] - v0[0]), f32(v2[1] - v0[1])
    ooa = f32(f32(1) / f32(f32(dx01 * dy20) - f32(dx20 * dy01)))
    dy20o, dy01o, dx20o, dx01o = f32(dy20 * ooa), f32(dy01 * ooa), f32(dx20 * ooa), f32(dx01 * ooa)
    x0c, y0c = f32(v0[0] - f32(xoff_center)), f32(v0[1] - f32(xoff_center))
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
            xx, yy = f32(x + eval_off), f32(y + eval_off)
            av = fma(pa[2], yy, fma(pa[1], xx, pa[0]))
            wv = fma(pw[2], yy, fma(pw[1], xx, pw[0]))
            out[y, x] = f32(av * f32(f32(1) / wv))
    return out
for name, (xc, eo) in {"center0.5/eval0": (0.5, 0.0), "center0/eval0.5": (0.0, 0.5)}.items():
    nb = 0
    for c in range(3):
        pred = model2(win, attr[:, c], xc, eo)
        nb += int((pred[cov].view(np.uint32) != img[:, :, c][cov].view(np.uint32)).sum())
    print(name, nb)
