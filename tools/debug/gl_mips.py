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
for name in sys.argv[1:] or ["brick_wall", "concrete_tiles", "asphalt", "floor_tiles_bw", "cinder_blocks"]:
    t = Texture.get(name)
    gl.glBindTexture(gl.GL_TEXTURE_2D, t.tex.id)
    rgb = pyoracle.texture_rgb_bottom_up(name + "_1")
    mips = pyoracle.mip_levels(rgb)
    lvl = 0
    while True:
        w, h = c_int(0), c_int(0)
        gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_WIDTH, byref(w))
        gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_HEIGHT, byref(h))
        if w.value == 0: break
        buf = np.zeros((h.value, w.value, 4), np.uint8)
        gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
        gl.glGetTexImage(gl.GL_TEXTURE_2D, lvl, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, buf.ctypes.data)
        if lvl < len(mips):
            d = buf[:, :, :3].astype(int) - mips[lvl].astype(int)
            print(name, lvl, (w.value, h.value), mips[lvl].shape, "mean d %.3f" % d.mean(), "min", d.min(), "max", d.max(), "alpha", buf[:,:,3].min())
        else:
            print(name, lvl, (w.value, h.value), "oracle has no such level")
        lvl += 1
