import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pyoracle
sys.argv = [sys.argv[0], "pickup_dr_s1"]
exec(open(os.path.join(HERE, "fixdiff.py")).read().split("for k in d")[0])
k = 100
sc = {key.split("/", 3)[3]: d[key] for key in d.files if key.startswith(f"gl/{k}/scene/")}
for view in ("agent", "top"):
    g = d[f"gl/{k}/view_{view}"]
    r = pyoracle.render(sc, width=800, height=600, nsamples=4, meshes=meshes_for(sc), view=view, render_agent=(view == "top"), want_prim=True)
    bad = (r["rgb"] != g).any(axis=2)
    for y, x in zip(*np.nonzero(bad)):
        print(view, "px", y, x, "gl", g[y, x], "orc", r["rgb"][y, x], "prims", r["prim"][y, x], "neigh gl", g[y, x-1], g[y, x+1])
