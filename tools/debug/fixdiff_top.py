import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle
exec(open(os.path.join(HERE, "fixdiff.py")).read().split("for k in d")[0])
what = sys.argv[2] if len(sys.argv) > 2 else "top"
for k in d["meta/frames"]:
    sc = {key.split("/", 3)[3]: d[key] for key in d.files if key.startswith(f"gl/{k}/scene/")}
    if what == "top":
        r = pyoracle.render(sc, nsamples=4, meshes=meshes_for(sc), want_prim=True, view="top", render_agent=True); g = d[f"gl/{k}/top"]
        bad = (r["rgb"] != g).any(axis=2)
    else:
        r = pyoracle.render(sc, nsamples=4, meshes=meshes_for(sc), want_prim=True); g = d[f"gl/{k}/z16"]
        bad = r["z16"] != g
    print("frame", k, "bad", bad.sum(), "npolys", len(sc["polys_nv"]))
    for y, x in list(zip(*np.nonzero(bad)))[:6]:
        print("   px", y, x, "gl", g[y, x], "orc", r["rgb"][y, x] if what == "top" else r["z16"][y, x], "prims", r["prim"][y, x])
if what == "top" and len(sys.argv) > 3:
    k = int(sys.argv[3])
    sc = {key.split("/", 3)[3]: d[key] for key in d.files if key.startswith(f"gl/{k}/scene/")}
    r = pyoracle.render(sc, nsamples=4, meshes=meshes_for(sc), want_prim=True, view="top", render_agent=True); g = d[f"gl/{k}/top"]
    bad = (r["rgb"] != g).any(axis=2)
    ys, xs = np.nonzero(bad)
    y0, y1, x0, x1 = max(min(ys) - 2, 0), min(max(ys) + 3, 60), max(min(xs) - 2, 0), min(max(xs) + 3, 80)
    x1 = min(x1, x0 + 30)
    for y in range(y0, y1): print(y, " ".join("%3d" % g[y, x, 0] for x in range(x0, x1)))
    print()
    for y in range(y0, y1): print(y, " ".join("%3d" % r["rgb"][y, x, 0] for x in range(x0, x1)))
    print("extent", sc["extent"], "ents", sc["ents_kind"], sc["mesh_names"])
