import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gl_feedback_check import *
env = refshim_gl.make_env("Hallway"); env.reset(seed=0)
fb = gl_feedback(env); orc = oracle_tris(env)
j = 0
for ti, t in enumerate(fb):
    while np.abs(orc[j, :30].reshape(3, 10)[:, :2] - t[:, :2]).max() >= 0.05: j += 1
    o = orc[j, :30].reshape(3, 10); j += 1
    print("tri", ti, "draw", int(orc[j - 1, 31]))
    for k in range(3):
        flag = lambda a, b: "ok " if a == b else "BAD"
        print("   win %9.4f %9.4f  st gl %s %s | orc %s %s  %s %s   col %s" % (t[k, 0], t[k, 1], float.hex(float(t[k, 8])), float.hex(float(t[k, 9])),
              float.hex(float(o[k, 8])), float.hex(float(o[k, 9])), flag(t[k, 8], o[k, 8]), flag(t[k, 9], o[k, 9]), flag(tuple(t[k, 4:7]), tuple(o[k, 4:7]))))
