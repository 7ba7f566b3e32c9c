import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle")); sys.path.insert(0, os.path.join(HERE, "..", "..", "tests"))
import pyoracle
name = sys.argv[1]
d = np.load(os.path.join(HERE, "..", "..", "tests", "golden", "gl_" + name + ".npz"), allow_pickle=False)
mz = np.load(os.path.join(HERE, "..", "..", "tests", "golden", "meshes.npz"))
def meshes_for(sc):
    out = {}
    for m in [str(x) for x in sc["mesh_names"]]:
        base = m.split("_")[0]
        a = {k: mz[f"{base}/{k}"] for k in ("verts", "norms", "texcs")}
        a["colors"] = np.broadcast_to(mz["kd:" + m].astype(np.float32), a["verts"].shape).copy()
        out[m] = a
    return out
for k in d["meta/frames"]:
    sc = {key.split("/", 3)[3]: d[key] for key in d.files if key.startswith(f"gl/{k}/scene/")}
    r = pyoracle.render(sc, nsamples=4, meshes=meshes_for(sc), want_prim=True)
    g = d[f"gl/{k}/rgb"]
    bad = (r["rgb"] != g).any(axis=2)
    for y, x in zip(*np.nonzero(bad)):
        print(name, "frame", k, "px", y, x, "gl", g[y, x], "orc", r["rgb"][y, x], "prims", r["prim"][y, x], "tex", [int(sc["polys_tex"][p]) if 0 <= p < len(sc["polys_tex"]) else -1 for p in r["prim"][y, x]])
