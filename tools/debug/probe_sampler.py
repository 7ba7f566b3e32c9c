"""Probe llvmpipe's bilinear / mip selection with an ortho quad whose texcoords are affine in pixel coords."""
from glprobe import *
rng = np.random.default_rng(3)

def draw_quad(W, H, tid, s0, t0, dsdx, dtdy, dsdy=0.0, dtdx=0.0):
    """texcoord at pixel centre (x+.5, y+.5) = (s0 + dsdx*(x+.5) + dsdy*(y+.5), t0 + dtdx*(x+.5) + dtdy*(y+.5)); GL coords (y up)"""
    reset_state()
    gl.glClearColor(0, 0, 0, 0); gl.glClear(gl.GL_COLOR_BUFFER_BIT | gl.GL_DEPTH_BUFFER_BIT)
    gl.glMatrixMode(gl.GL_PROJECTION); gl.glLoadIdentity(); gl.glOrtho(0, W, 0, H, -1, 1)
    gl.glEnable(gl.GL_TEXTURE_2D); gl.glBindTexture(gl.GL_TEXTURE_2D, tid)
    gl.glBegin(gl.GL_QUADS)
    for (x, y) in [(0, 0), (W, 0), (W, H), (0, H)]:
        gl.glTexCoord2f(s0 + dsdx * x + dsdy * y, t0 + dtdx * x + dtdy * y); gl.glVertex3f(x, y, 0)
    gl.glEnd(); gl.glFlush()
    return read_rgba_f(W, H)

def lerp8(a, b, w): return a + ((w * (b - a) + 128) >> 8)

def model_bilinear(lv, s, t):
    """lv int[h,w,3]; s,t float32 normalized arrays; POT repeat"""
    h, w = lv.shape[:2]
    fx = np.rint((s * np.float32(w)).astype(np.float32) * np.float32(256)).astype(np.int64) - 128
    fy = np.rint((t * np.float32(h)).astype(np.float32) * np.float32(256)).astype(np.int64) - 128
    i0, wx = (fx >> 8) & (w - 1), fx & 255
    j0, wy = (fy >> 8) & (h - 1), fy & 255
    i1, j1 = (i0 + 1) & (w - 1), (j0 + 1) & (h - 1)
    top = lerp8(lv[j0, i0], lv[j0, i1], wx[..., None])
    bot = lerp8(lv[j1, i0], lv[j1, i1], wx[..., None])
    return lerp8(top, bot, wy[..., None])

if __name__ == "__main__":
    W, H = 32, 16
    make_fbo(W, H)
    tex = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    tid = make_tex(tex)
    L = [l.astype(np.int64) for l in get_levels(tid)]
    # magnification: 0.013 texel per pixel
    s0, t0, dsdx, dtdy = 0.1234, 0.777, 0.00731, 0.00513
    img = draw_quad(W, H, tid, s0, t0, dsdx, dtdy)
    got = np.rint(img[:, :, :3] * 255).astype(int)
    print("exact multiple of 1/255:", np.abs(img[:, :, :3] * 255 - got).max())
    xs = (np.arange(W) + 0.5).astype(np.float32); ys = (np.arange(H) + 0.5).astype(np.float32)
    s = (np.float32(s0) + np.float32(dsdx) * xs)[None, :].repeat(H, 0)
    t = (np.float32(t0) + np.float32(dtdy) * ys)[:, None].repeat(W, 1)
    pred = model_bilinear(L[0], s.astype(np.float32), t.astype(np.float32))
    d = pred - got
    print("mag: mismatching channels", (d != 0).sum(), "of", d.size, "max", np.abs(d).max())
