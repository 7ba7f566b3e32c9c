"""z at sample 0 in a multisampled target with a 32-bit float depth buffer, read back exactly"""
from glprobe import *
f32 = np.float32
W, H = 32, 24
def fma(a, b, c): return f32(np.float64(a) * np.float64(b) + np.float64(c))
def make_ms_fbo_zf(w, h, samples):
    fbo, tex, dtex = c_uint(0), c_uint(0), c_uint(0)
    gl.glGenFramebuffers(1, byref(fbo)); gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, fbo.value)
    gl.glGenTextures(1, byref(tex)); gl.glGenTextures(1, byref(dtex))
    if samples:
        gl.glBindTexture(gl.GL_TEXTURE_2D_MULTISAMPLE, tex.value)
        gl.glTexImage2DMultisample(gl.GL_TEXTURE_2D_MULTISAMPLE, samples, gl.GL_RGBA32F, w, h, 1)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D_MULTISAMPLE, tex.value, 0)
        gl.glBindTexture(gl.GL_TEXTURE_2D_MULTISAMPLE, dtex.value)
        gl.glTexImage2DMultisample(gl.GL_TEXTURE_2D_MULTISAMPLE, samples, gl.GL_DEPTH_COMPONENT32F, w, h, 1)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_DEPTH_ATTACHMENT, gl.GL_TEXTURE_2D_MULTISAMPLE, dtex.value, 0)
    else:
        gl.glBindTexture(gl.GL_TEXTURE_2D, tex.value)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA32F, w, h, 0, gl.GL_RGBA, gl.GL_FLOAT, None)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D, tex.value, 0)
        gl.glBindTexture(gl.GL_TEXTURE_2D, dtex.value)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_DEPTH_COMPONENT32F, w, h, 0, gl.GL_DEPTH_COMPONENT, gl.GL_FLOAT, None)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_DEPTH_ATTACHMENT, gl.GL_TEXTURE_2D, dtex.value, 0)
    assert gl.glCheckFramebufferStatus(gl.GL_FRAMEBUFFER) == gl.GL_FRAMEBUFFER_COMPLETE
    gl.glViewport(0, 0, w, h)
    return fbo.value
ms = make_ms_fbo_zf(W, H, 4); ss = make_ms_fbo_zf(W, H, 0)
rng = np.random.default_rng(11)
def planes(v):
    order = [1, 0, 2]
    v0, v1, v2 = (v[i] for i in order)
    dx01, dy01, dx20, dy20 = f32(v0[0] - v1[0]), f32(v0[1] - v1[1]), f32(v2[0] - v0[0]), f32(v2[1] - v0[1])
    ooa = f32(f32(1) / f32(f32(dx01 * dy20) - f32(dx20 * dy01)))
    dy20o, dy01o, dx20o, dx01o = f32(dy20 * ooa), f32(dy01 * ooa), f32(dx20 * ooa), f32(dx01 * ooa)
    x0c, y0c = v0[0], v0[1]
    b0, b1, b2 = v0[2], v1[2], v2[2]
    da01, da20 = f32(b0 - b1), f32(b2 - b0)
    dadx = f32(f32(da01 * dy20o) - f32(da20 * dy01o))
    dady = f32(f32(da20 * dx01o) - f32(da01 * dx20o))
    c0 = f32(b0 - f32(f32(dadx * x0c) + f32(dady * y0c)))
    return c0, dadx, dady
for k in range(10):
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, ms)
    reset_state(); gl.glEnable(gl.GL_MULTISAMPLE); gl.glEnable(gl.GL_DEPTH_TEST); gl.glDisable(gl.GL_SAMPLE_MASK)
    gl.glClearColor(0, 0, 0, 0); gl.glClearDepth(1.0); gl.glClear(gl.GL_COLOR_BUFFER_BIT | gl.GL_DEPTH_BUFFER_BIT)
    while True:
        p = rng.uniform(1, [W - 1, H - 1], (3, 2))
        area = (p[1, 0] - p[0, 0]) * (p[2, 1] - p[0, 1]) - (p[2, 0] - p[0, 0]) * (p[1, 1] - p[0, 1])
        if area > 20: break
    w = rng.uniform(0.5, 8, 3).astype(np.float32)
    ndc = np.stack([p[:, 0] / W * 2 - 1, p[:, 1] / H * 2 - 1], axis=1)
    clip = np.zeros((3, 4), np.float32)
    clip[:, 0] = (ndc[:, 0] * w).astype(np.float32); clip[:, 1] = (ndc[:, 1] * w).astype(np.float32)
    clip[:, 2] = (rng.uniform(-0.9, 0.9, 3) * w).astype(np.float32); clip[:, 3] = w
    gl.glBegin(gl.GL_TRIANGLES)
    for i in range(3):
        gl.glColor4f(1, 1, 1, 1); gl.glVertex4f(*[float(x) for x in clip[i]])
    gl.glEnd(); gl.glFlush()
    gl.glBindFramebuffer(gl.GL_READ_FRAMEBUFFER, ms); gl.glBindFramebuffer(gl.GL_DRAW_FRAMEBUFFER, ss)
    gl.glBlitFramebuffer(0, 0, W, H, 0, 0, W, H, gl.GL_DEPTH_BUFFER_BIT, gl.GL_NEAREST)
    gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, ss)
    z = np.zeros((H, W), np.float32)
    gl.glReadPixels(0, 0, W, H, gl.GL_DEPTH_COMPONENT, gl.GL_FLOAT, z.ctypes.data)
    win = np.zeros((3, 4), np.float32)
    for i in range(3):
        oow = f32(f32(1) / clip[i, 3])
        win[i, 0] = fma(f32(clip[i, 0] * oow), f32(W / 2), f32(W / 2))
        win[i, 1] = fma(f32(clip[i, 1] * oow), f32(H / 2), f32(H / 2))
        win[i, 2] = fma(f32(clip[i, 2] * oow), f32(0.5), f32(0.5)); win[i, 3] = oow
    c0, dadx, dady = planes(win)
    cov = z < 1.0
    res = {}
    sx, sy = f32(0.375), f32(0.125)
    for name in ("direct", "center+off", "yx", "off_first"):
        nb = 0
        for y, x in zip(*np.nonzero(cov)):
            if name == "direct": v = fma(dady, f32(y) + sy, fma(dadx, f32(x) + sx, c0))
            elif name == "yx": v = fma(dadx, f32(x) + sx, fma(dady, f32(y) + sy, c0))
            elif name == "center+off":
                zc = fma(dady, f32(y + 0.5), fma(dadx, f32(x + 0.5), c0)); v = fma(dady, sy - f32(0.5), fma(dadx, sx - f32(0.5), zc))
            else:
                zo = fma(dady, sy, fma(dadx, sx, c0)); v = fma(dady, f32(y), fma(dadx, f32(x), zo))
            v = min(max(v, f32(0)), f32(1))
            nb += int(v != z[y, x])
        res[name] = nb
    print(k, "covered", int(cov.sum()), res)
