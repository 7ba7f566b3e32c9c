from glprobe import *
rng = np.random.default_rng(2)
def lerp(a, b, wgt): return a + ((wgt * (b - a) + 128) >> 8)
for n in [7, 15, 31, 63, 127, 255, 3, 5, 9, 11, 13]:
    row = rng.integers(0, 256, (1, n, 3), dtype=np.uint8)
    rgb = np.repeat(row, 4, axis=0)          # 4 identical rows -> y filter is the identity
    L = get_levels(make_tex(rgb))
    s, d = L[0].astype(int)[0], L[1].astype(int)[0]
    dn = d.shape[0]
    res = []
    for i in range(dn):
        ok = []
        for i0 in range(max(0, 2*i-1), min(n-1, 2*i+2)):
            for wgt in range(0, 257):
                if (lerp(s[i0], s[i0+1], wgt) == d[i]).all(): ok.append((i0, wgt))
        res.append(ok)
    exp = [((i + 0.5) * n / dn - 0.5) for i in range(dn)]
    print(n, dn)
    for i in list(range(min(dn, 6))) + list(range(max(6, dn - 3), dn)):
        ws = [w for (i0, w) in res[i]]
        i0s = sorted(set(i0 for (i0, w) in res[i]))
        print("  i", i, "expect s=%.4f frac*256=%.2f" % (exp[i], (exp[i] - np.floor(exp[i])) * 256), "fit i0", i0s, "w range", (min(ws), max(ws)) if ws else None)
