from probe_sampler import *
import struct
f32 = np.float32
def fast_lod(rho2):
    b = struct.unpack("<I", struct.pack("<f", f32(rho2)))[0]
    e = ((b >> 23) & 255) - 127
    m = struct.unpack("<f", struct.pack("<I", (b & 0x7fffff) | 0x3f800000))[0]
    lod = f32(f32(f32(e) + f32(f32(m) - f32(1))) * f32(0.5))
    ip = int(np.floor(lod)); fp = f32(lod - f32(ip))
    return ip, fp
def model(L, s, t, rho2):
    last = len(L) - 1
    ip, fp = fast_lod(rho2)
    if ip < 0: l0, fp = 0, 0.0
    elif ip >= last: l0, fp = last, 0.0
    else: l0 = ip
    w8 = int(fp * 256)
    c0 = model_bilinear(L[l0], s, t)
    if w8 == 0: return c0, (l0, w8)
    return lerp8(c0, model_bilinear(L[min(l0 + 1, last)], s, t), w8), (l0, w8)
if __name__ == "__main__":
    W, H = 16, 8
    make_fbo(W, H)
    tex = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    tid = make_tex(tex)
    L = [l.astype(np.int64) for l in get_levels(tid)]
    xs = (np.arange(W) + 0.5).astype(np.float32); ys = (np.arange(H) + 0.5).astype(np.float32)
    bad = 0
    for k in range(200):
        scale = 2.0 ** rng.uniform(-1, 8.5)
        ratio = rng.uniform(0.2, 1.0)
        dsdx, dtdy = scale / 256, scale * ratio / 256
        if k % 2 == 1: dsdx, dtdy = dtdy, dsdx
        dsdy, dtdx = (rng.uniform(-1, 1) * scale / 256, rng.uniform(-1, 1) * scale / 256) if k % 3 == 0 else (0.0, 0.0)
        s0, t0 = rng.uniform(0, 1), rng.uniform(0, 1)
        img = draw_quad(W, H, tid, s0, t0, dsdx, dtdy, dsdy, dtdx)
        got = np.rint(img[:, :, :3] * 255).astype(int)
        X, Y = np.meshgrid(xs, ys)
        s = (f32(s0) + f32(dsdx) * X + f32(dsdy) * Y).astype(np.float32)
        t = (f32(t0) + f32(dtdx) * X + f32(dtdy) * Y).astype(np.float32)
        sw = f32(256)
        ax = f32(f32(f32(dsdx) * sw) ** 2) + f32(f32(f32(dtdx) * sw) ** 2)
        ay = f32(f32(f32(dsdy) * sw) ** 2) + f32(f32(f32(dtdy) * sw) ** 2)
        pred, info = model(L, s, t, max(ax, ay))
        nb = int((pred != got).any(axis=2).sum())
        bad += nb > 0
        if nb: print("rho2 %.4f" % max(ax, ay), info, "bad px", nb, "max", np.abs(pred - got).max())
    print("bad frames", bad)
