"""Scratch helpers for black-box probing of llvmpipe (not part of the product or the tests)."""
import os, sys, ctypes
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import refshim_gl
refshim_gl.load_reference()
gl = refshim_gl.gl
from ctypes import c_uint, c_int, c_float, byref

def make_fbo(w, h, samples=0, color=None, depth=True):
    color = color or gl.GL_RGBA32F
    fbo, tex, rb = c_uint(0), c_uint(0), c_uint(0)
    gl.glGenFramebuffers(1, byref(fbo)); gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, fbo.value)
    gl.glGenTextures(1, byref(tex))
    if samples:
        gl.glBindTexture(gl.GL_TEXTURE_2D_MULTISAMPLE, tex.value)
        gl.glTexImage2DMultisample(gl.GL_TEXTURE_2D_MULTISAMPLE, samples, color, w, h, 1)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D_MULTISAMPLE, tex.value, 0)
    else:
        gl.glBindTexture(gl.GL_TEXTURE_2D, tex.value)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, color, w, h, 0, gl.GL_RGBA, gl.GL_FLOAT, None)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D, tex.value, 0)
    if depth:
        gl.glGenRenderbuffers(1, byref(rb)); gl.glBindRenderbuffer(gl.GL_RENDERBUFFER, rb.value)
        if samples:
            gl.glRenderbufferStorageMultisample(gl.GL_RENDERBUFFER, samples, gl.GL_DEPTH_COMPONENT16, w, h)
        else:
            gl.glRenderbufferStorage(gl.GL_RENDERBUFFER, gl.GL_DEPTH_COMPONENT16, w, h)
        gl.glFramebufferRenderbuffer(gl.GL_FRAMEBUFFER, gl.GL_DEPTH_ATTACHMENT, gl.GL_RENDERBUFFER, rb.value)
    assert gl.glCheckFramebufferStatus(gl.GL_FRAMEBUFFER) == gl.GL_FRAMEBUFFER_COMPLETE
    gl.glViewport(0, 0, w, h)
    return fbo.value

def read_rgba_f(w, h):
    buf = np.zeros((h, w, 4), np.float32)
    gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, w, h, gl.GL_RGBA, gl.GL_FLOAT, buf.ctypes.data)
    return buf

def read_z16(w, h):
    buf = np.zeros((h, w), np.uint16)
    gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
    gl.glReadPixels(0, 0, w, h, gl.GL_DEPTH_COMPONENT, gl.GL_UNSIGNED_SHORT, buf.ctypes.data)
    return buf

def make_tex(rgb, mips=True, minf=None, magf=None, wrap=None):
    """rgb u8[h,w,3] rows bottom-up, uploaded like the reference does (GL_RGB internal, RGBA data)."""
    h, w = rgb.shape[:2]
    rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2).copy()
    t = c_uint(0)
    gl.glGenTextures(1, byref(t)); gl.glBindTexture(gl.GL_TEXTURE_2D, t.value)
    gl.glPixelStorei(gl.GL_UNPACK_ALIGNMENT, 1)
    gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGB, w, h, 0, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, rgba.ctypes.data)
    if mips:
        gl.glHint(gl.GL_GENERATE_MIPMAP_HINT, gl.GL_NICEST)
        gl.glGenerateMipmap(gl.GL_TEXTURE_2D)
    gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MAG_FILTER, magf or gl.GL_LINEAR)
    gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MIN_FILTER, minf or (gl.GL_LINEAR_MIPMAP_LINEAR if mips else gl.GL_LINEAR))
    return t.value

def get_levels(tid):
    gl.glBindTexture(gl.GL_TEXTURE_2D, tid)
    out, lvl = [], 0
    while True:
        w, h = c_int(0), c_int(0)
        gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_WIDTH, byref(w))
        gl.glGetTexLevelParameteriv(gl.GL_TEXTURE_2D, lvl, gl.GL_TEXTURE_HEIGHT, byref(h))
        if w.value == 0: break
        buf = np.zeros((h.value, w.value, 4), np.uint8)
        gl.glPixelStorei(gl.GL_PACK_ALIGNMENT, 1)
        gl.glGetTexImage(gl.GL_TEXTURE_2D, lvl, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE, buf.ctypes.data)
        out.append(buf[:, :, :3].copy()); lvl += 1
    return out

def reset_state():
    for cap in ("GL_LIGHTING", "GL_CULL_FACE", "GL_DEPTH_TEST", "GL_TEXTURE_2D", "GL_MULTISAMPLE"):
        gl.glDisable(getattr(gl, cap))
    gl.glMatrixMode(gl.GL_PROJECTION); gl.glLoadIdentity()
    gl.glMatrixMode(gl.GL_MODELVIEW); gl.glLoadIdentity()
    gl.glColor3f(1, 1, 1)
