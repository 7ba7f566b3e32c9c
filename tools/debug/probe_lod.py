from probe_sampler import *
import struct
def f32(x): return np.float32(x)
PRE = f32((2 * 2 - 0.5) / (np.sqrt(2.0) * 2))
def brilinear(rho):
    rho = f32(f32(rho) * PRE)
    b = struct.unpack("<I", struct.pack("<f", rho))[0]
    ip = ((b >> 23) & 255) - 127
    m = struct.unpack("<f", struct.pack("<I", (b & 0x7fffff) | 0x3f800000))[0]
    return ip, f32(f32(m) * 2 - 3)
def model(L, s, t, rho):
    last = len(L) - 1
    ip, fp = brilinear(rho)
    if ip < 0: l0, fp = 0, 0.0
    elif ip >= last: l0, fp = last, 0.0
    else: l0 = ip
    w8 = int(max(fp, 0.0) * 256)
    c0 = model_bilinear(L[l0], s, t)
    if w8 == 0: return c0, (l0, w8)
    c1 = model_bilinear(L[min(l0 + 1, last)], s, t)
    return lerp8(c0, c1, w8), (l0, w8)
W, H = 16, 8
make_fbo(W, H)
tex = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
tid = make_tex(tex)
L = [l.astype(np.int64) for l in get_levels(tid)]
xs = (np.arange(W) + 0.5).astype(np.float32); ys = (np.arange(H) + 0.5).astype(np.float32)
bad = 0
for k in range(60):
    scale = 2.0 ** rng.uniform(-1, 7)          # texels per pixel along x
    ratio = rng.uniform(0.3, 1.0)
    dsdx, dtdy = scale / 256, scale * ratio / 256
    if k % 3 == 1: dsdx, dtdy = dtdy, dsdx
    s0, t0 = rng.uniform(0, 1), rng.uniform(0, 1)
    img = draw_quad(W, H, tid, s0, t0, dsdx, dtdy)
    got = np.rint(img[:, :, :3] * 255).astype(int)
    s = (f32(s0) + f32(dsdx) * xs)[None, :].repeat(H, 0).astype(np.float32)
    t = (f32(t0) + f32(dtdy) * ys)[:, None].repeat(W, 1).astype(np.float32)
    rho = max(abs(s[0, 1] - s[0, 0]) * f32(256), abs(t[1, 0] - t[0, 0]) * f32(256))
    pred, info = model(L, s, t, rho)
    nb = int((pred != got).any(axis=2).sum())
    bad += nb > 0
    print("rho %.4f" % rho, info, "bad px", nb, "max", np.abs(pred - got).max())
print("bad frames", bad)
