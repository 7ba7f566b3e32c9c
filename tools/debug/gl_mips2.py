import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import refshim_gl, pyoracle
refshim_gl.load_reference()
gl = refshim_gl.gl
from miniworld.opengl import Texture
from ctypes import c_int, byref
def levels(name):
    t = Texture.get(name)
    gl.glBindTexture(gl.GL_TEXTURE_2D, t.tex.id)
    out = []
    lvl = 0
    while True:
        w, h = c_int(0), c_int(0)
        gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_WIDTH, byref(w))
        gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_HEIGHT, byref(h))
        if w.value == 0: break
        buf = np.zeros((h.value, w.value, 4), np.uint8)
        gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
        gl.glGetTexImage(gl.GL_TEXTURE_2D, lvl, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, buf.ctypes.data)
        out.append(buf[:, :, :3].astype(np.int32)); lvl += 1
    return out
def up(a, b): return (a + b + 1) >> 1
for name in sys.argv[1:]:
    L = levels(name)
    for l in range(len(L) - 1):
        s, d = L[l], L[l + 1]
        if s.shape[0] % 2 or s.shape[1] % 2:
            print(name, l, "odd", s.shape); continue
        a, b, c, e = s[0::2, 0::2], s[0::2, 1::2], s[1::2, 0::2], s[1::2, 1::2]
        cands = {"hv": up(up(a, b), up(c, e)), "vh": up(up(a, c), up(b, e)), "rhu": (a + b + c + e + 2) >> 2}
        print(name, l, s.shape, {k: int((v != d).sum()) for k, v in cands.items()})
