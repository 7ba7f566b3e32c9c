from probe_sampler import *
f32 = np.float32
W, H = 16, 8
make_fbo(W, H)
tex = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
tid = make_tex(tex)
L = [l.astype(np.int64) for l in get_levels(tid)]
xs = (np.arange(W) + 0.5).astype(np.float32); ys = (np.arange(H) + 0.5).astype(np.float32)
for k in range(40):
    scale = 2.0 ** rng.uniform(-0.5, 4)
    ratio = rng.uniform(0.2, 1.0)
    dsdx, dtdy = scale / 256, scale * ratio / 256
    if k % 2 == 1: dsdx, dtdy = dtdy, dsdx
    s0, t0 = rng.uniform(0, 1), rng.uniform(0, 1)
    img = draw_quad(W, H, tid, s0, t0, dsdx, dtdy)
    got = np.rint(img[:, :, :3] * 255).astype(int)
    s = (f32(s0) + f32(dsdx) * xs)[None, :].repeat(H, 0).astype(np.float32)
    t = (f32(t0) + f32(dtdy) * ys)[:, None].repeat(W, 1).astype(np.float32)
    fits = []
    for l0 in range(len(L) - 1):
        c0 = model_bilinear(L[l0], s, t); c1 = model_bilinear(L[l0 + 1], s, t)
        for w8 in range(256):
            if (lerp8(c0, c1, w8) == got).all(): fits.append((l0, w8))
    rx, ry = dsdx * 256, dtdy * 256
    print("rx %.4f ry %.4f  hyp %.4f" % (rx, ry, np.hypot(rx, ry)), "fits", fits[:4], len(fits))
