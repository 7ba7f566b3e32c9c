"""How many (tile, triangle) shading events per tile for different tile shapes (hallway golden frames, 8 samples)."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import helpers, pyoracle
PAT8 = [(9,5),(7,11),(13,9),(5,3),(3,13),(1,7),(11,15),(15,1)]
def tris_of(sc, meshes):
    s, keep = pyoracle.pack_scene(sc, nsamples=8, meshes=meshes)
    L = pyoracle.lib()
    L.mwo_debug_geometry.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    buf = np.zeros((200000, 32), np.float32)
    n = L.mwo_debug_geometry(C.byref(s), 0, buf.ctypes.data, 200000)
    return buf[:n]
def coverage(t):
    """bool[60,80] pixels with any sample inside (float edge functions are fine for counting), front-facing only."""
    v = t[:30].reshape(3, 10)[:, :2].astype(np.float64)
    area = (v[1,0]-v[0,0])*(v[2,1]-v[0,1]) - (v[2,0]-v[0,0])*(v[1,1]-v[0,1])
    if area <= 0: return None           # front faces are counter-clockwise (y up)
    ys, xs = np.mgrid[0:60, 0:80]
    cov = np.zeros((60, 80), bool)
    for sx, sy in PAT8:
        X = xs + sx / 16.0; Y = ys + sy / 16.0
        ins = np.ones((60, 80), bool)
        for i in range(3):
            a, b = v[i], v[(i + 1) % 3]
            e = (b[0]-a[0])*(Y-a[1]) - (b[1]-a[1])*(X-a[0])
            ins &= e >= 0
        cov |= ins
    return cov
for case in sys.argv[1:] or ["hallway_s0", "oneroom_s0"]:
    s0, tr, meta, obs = helpers.load_case(case)
    res = {}
    for f in sorted(obs):
        sc = helpers.frame_scene(s0, obs[f])
        T = tris_of(sc, helpers.golden_meshes(s0))
        covs = [c for c in (coverage(t) for t in T) if c is not None and c.any()]
        for (tw, th) in ((16, 4), (8, 6), (20, 3), (8, 4), (4, 4), (16, 2), (8, 10)):
            ev = 0
            for c in covs:
                ev += c.reshape(60 // th, th, 80 // tw, tw).any(axis=(1, 3)).sum()
            ntiles = (60 // th) * (80 // tw)
            res.setdefault((tw, th), []).append((ev / ntiles, ev * tw * th / 4800.0))
    for k, v in res.items():
        a = np.array(v)
        print(case, k, "events/tile %.2f" % a[:, 0].mean(), " lane-shades per pixel %.2f" % a[:, 1].mean())
