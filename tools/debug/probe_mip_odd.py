from glprobe import *
rng = np.random.default_rng(1)
def up(a, b): return (a + b + 1) >> 1
for (h, w) in [(3, 3), (5, 5), (7, 7), (5, 8), (8, 5), (255, 510), (15, 6)]:
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    L = get_levels(make_tex(rgb))
    print((h, w), [l.shape[:2] for l in L])
    s, d = L[0].astype(int), L[1].astype(int)
    dh, dw = d.shape[:2]
    # hypothesis: bilinear at s = (i+0.5)*sw/dw - 0.5, 8-bit weights, x first then y, rounding lerp
    def lerp(a, b, wgt): return a + ((wgt * (b - a) + 128) >> 8)
    def axis(n, dn):
        idx0, idx1, wt = [], [], []
        for i in range(dn):
            sc = (i + 0.5) * n / dn - 0.5
            i0 = int(np.floor(sc)); f = sc - i0
            idx0.append(min(max(i0, 0), n - 1)); idx1.append(min(max(i0 + 1, 0), n - 1)); wt.append(int(f * 256))
        return np.array(idx0), np.array(idx1), np.array(wt)
    x0, x1, wx = axis(w, dw); y0, y1, wy = axis(h, dh)
    print("  wx", wx[:6], "wy", wy[:6])
    top = lerp(s[y0][:, x0], s[y0][:, x1], wx[None, :, None])
    bot = lerp(s[y1][:, x0], s[y1][:, x1], wx[None, :, None])
    pred = lerp(top, bot, wy[:, None, None])
    print("  mismatches", (pred != d).sum(), "of", d.size, "max", np.abs(pred - d).max())
