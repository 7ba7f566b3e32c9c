"""The reference on REAL OpenGL (FIXTURE TOOLING — never imported by the product).

Runs ``/root/reference/miniworld`` unmodified — its own ``FrameBuffer``, ``Texture``, ``drawBox``,
display lists, ``render_obs`` / ``render_depth`` / ``render_top_view`` / ``render`` /
``get_visible_ents`` (opengl.py:102-503, miniworld.py:1019-1362) — on Mesa's llvmpipe through the
headless context of ``oracle/refgl/glctx.c``.  What is substituted is pyglet, which is not installed:

* ``pyglet.gl``: every ``gl*`` name is a ctypes binding to the real entry point (prototypes below, the
  ones pyglet's generated bindings carry), every ``GL_*`` enum is read from ``/usr/include/GL``;
  ``gluPerspective`` / ``gluLookAt`` restate libGLU's published algorithm (SGI libutil/project.c: the
  perspective matrix in double through ``glMultMatrixd``; ``gluLookAt`` in float — forward / side / up
  normalised with float arithmetic — through ``glMultMatrixf`` followed by ``glTranslated(-eye)``).
* ``pyglet.image.load`` decodes with PIL; ``get_texture()`` creates a GL_TEXTURE_2D the way pyglet 1.5
  does (GL_LINEAR min / mag, RGBA upload, bottom-up rows); ``get_image_data().get_data("RGBA", pitch)``
  returns the bottom-up RGBA bytes the reference hands to ``glTexImage2D`` (opengl.py:161-171).
* ``pyglet.graphics.vertex_list(n, ("v3f", ..), ("t2f", ..), ("n3f", ..), ("c3f", ..)).draw(mode)``: client
  arrays + ``glDrawArrays`` (what pyglet's vertex domains issue).
* ``pyglet.window.Window``: the context is already current; ``switch_to`` is a no-op.
* ``gymnasium`` stubs come from tools/refshim.py (seeding identical to gymnasium's).

Only works where /root/reference and the Mesa swrast driver exist (the build container).
"""
from __future__ import annotations

import ctypes
import math
import os
import re
import subprocess
import sys
import types
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_ubyte, c_uint, c_ushort, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import refshim  # noqa: E402  (gymnasium stubs, REFERENCE_ROOT)

GLCTX_SRC = os.path.join(ROOT, "oracle", "refgl", "glctx.c")
GLCTX_SO = os.path.join(ROOT, "oracle", "_ref", "libglctx.so")

_ctx = None


def gl_available() -> bool:
    return os.path.exists("/usr/lib/x86_64-linux-gnu/dri/swrast_dri.so") and refshim.reference_available()


def _context():
    global _ctx
    if _ctx is None:
        if not os.path.exists(GLCTX_SO) or os.path.getmtime(GLCTX_SO) < os.path.getmtime(GLCTX_SRC):
            os.makedirs(os.path.dirname(GLCTX_SO), exist_ok=True)
            subprocess.check_call(["gcc", "-O1", "-shared", "-fPIC", "-o", GLCTX_SO, GLCTX_SRC, "-ldl"])
        lib = ctypes.CDLL(GLCTX_SO)
        lib.glctx_getproc.restype = c_void_p
        lib.glctx_getproc.argtypes = [c_char_p]
        lib.glctx_error.restype = c_char_p
        lib.glctx_create.argtypes = [c_char_p]
        if lib.glctx_create(None) != 0:
            raise RuntimeError("no GL context: " + lib.glctx_error().decode())
        _ctx = lib
    return _ctx


# ---- prototypes (result, args) --------------------------------------------------------------------------
E, I, U, F, D, B = c_uint, c_int, c_uint, c_float, c_double, c_ubyte
P = c_void_p
_PROTO = {
    "glBegin": (None, [E]), "glEnd": (None, []),
    "glVertex3f": (None, [F, F, F]), "glNormal3f": (None, [F, F, F]), "glColor3f": (None, [F, F, F]),
    "glTexCoord2f": (None, [F, F]),
    "glEnable": (None, [E]), "glDisable": (None, [E]),
    "glMatrixMode": (None, [E]), "glLoadIdentity": (None, []), "glPushMatrix": (None, []), "glPopMatrix": (None, []),
    "glTranslatef": (None, [F, F, F]), "glTranslated": (None, [D, D, D]), "glRotatef": (None, [F, F, F, F]),
    "glScalef": (None, [F, F, F]),
    "glLoadMatrixf": (None, [POINTER(F)]), "glMultMatrixf": (None, [POINTER(F)]), "glMultMatrixd": (None, [POINTER(D)]),
    "glOrtho": (None, [D, D, D, D, D, D]),
    "glLightf": (None, [E, E, F]), "glLightfv": (None, [E, E, POINTER(F)]),
    "glShadeModel": (None, [E]), "glColorMaterial": (None, [E, E]),
    "glClearColor": (None, [F, F, F, F]), "glClearDepth": (None, [D]), "glClear": (None, [U]),
    "glViewport": (None, [I, I, I, I]), "glFlush": (None, []), "glFinish": (None, []),
    "glHint": (None, [E, E]), "glPixelStorei": (None, [E, I]),
    "glGenTextures": (None, [I, POINTER(U)]), "glBindTexture": (None, [E, U]),
    "glTexImage2D": (None, [E, I, I, I, I, I, E, E, P]),
    "glTexImage2DMultisample": (None, [E, I, E, I, I, B]),
    "glTexParameteri": (None, [E, E, I]), "glGenerateMipmap": (None, [E]),
    "glGetTexImage": (None, [E, I, E, E, P]), "glGetTexLevelParameteriv": (None, [E, I, E, POINTER(I)]),
    "glGenFramebuffers": (None, [I, POINTER(U)]), "glBindFramebuffer": (None, [E, U]),
    "glFramebufferTexture2D": (None, [E, E, E, U, I]),
    "glGenRenderbuffers": (None, [I, POINTER(U)]), "glBindRenderbuffer": (None, [E, U]),
    "glRenderbufferStorage": (None, [E, E, I, I]), "glRenderbufferStorageMultisample": (None, [E, I, E, I, I]),
    "glFramebufferRenderbuffer": (None, [E, E, E, U]), "glCheckFramebufferStatus": (E, [E]),
    "glBlitFramebuffer": (None, [I, I, I, I, I, I, I, I, U, E]),
    "glReadPixels": (None, [I, I, I, I, E, E, P]),
    "glGetIntegerv": (None, [E, POINTER(I)]), "glGetFloatv": (None, [E, POINTER(F)]),
    "glGetMultisamplefv": (None, [E, U, POINTER(F)]),
    "glGetString": (c_char_p, [E]), "glGetError": (E, []),
    "glNewList": (None, [U, E]), "glEndList": (None, []), "glCallList": (None, [U]), "glDeleteLists": (None, [U, I]),
    "glIsList": (B, [U]),
    "glGenQueries": (None, [I, POINTER(U)]), "glDeleteQueries": (None, [I, POINTER(U)]),
    "glBeginQuery": (None, [E, U]), "glEndQuery": (None, [E]), "glGetQueryObjectuiv": (None, [U, E, POINTER(U)]),
    "glEnableClientState": (None, [E]), "glDisableClientState": (None, [E]),
    "glVertexPointer": (None, [I, E, I, P]), "glNormalPointer": (None, [E, I, P]),
    "glColorPointer": (None, [I, E, I, P]), "glTexCoordPointer": (None, [I, E, I, P]),
    "glDrawArrays": (None, [E, I, I]),
    "glDepthFunc": (None, [E]), "glCullFace": (None, [E]), "glFrontFace": (None, [E]),
    "glFeedbackBuffer": (None, [I, E, POINTER(F)]), "glRenderMode": (I, [E]),
    "glGetLightfv": (None, [E, E, POINTER(F)]), "glIsEnabled": (B, [E]),
    "glColor4f": (None, [F, F, F, F]), "glVertex4f": (None, [F, F, F, F]), "glVertex2f": (None, [F, F]),
    "glDepthRange": (None, [D, D]), "glPolygonMode": (None, [E, E]),
    "glSampleMaski": (None, [U, U]), "glScissor": (None, [I, I, I, I]), "glDepthMask": (None, [B]),
    "glColorMask": (None, [B, B, B, B]),
}


def _read_enums():
    out = {}
    pat = re.compile(r"#define\s+(GL_[A-Za-z0-9_]+)\s+(0x[0-9A-Fa-f]+|\d+)\b")
    for name in ("gl.h", "glext.h"):
        with open(os.path.join("/usr/include/GL", name)) as f:
            for m in pat.finditer(f.read()):
                out.setdefault(m.group(1), int(m.group(2), 0))
    return out


class _GLInfo:
    def __init__(self, gl):
        self._gl = gl

    def have_extension(self, name):
        return True  # llvmpipe 4.5 compat has every extension the reference asks about (FBO multisample)

    def get_renderer(self):
        return self._gl.glGetString(self._gl.GL_RENDERER).decode()

    def get_version(self):
        return self._gl.glGetString(self._gl.GL_VERSION).decode()


class _GLModule(types.ModuleType):
    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []
        self.__dict__.update(_read_enums())
        self.GLfloat, self.GLdouble, self.GLubyte, self.GLuint, self.GLint, self.GLushort = (
            c_float, c_double, c_ubyte, c_uint, c_int, c_ushort)
        self.GLenum, self.GLsizei, self.GLboolean = c_uint, c_int, c_ubyte
        ctx = _context()
        for fn, (res, args) in _PROTO.items():
            addr = ctx.glctx_getproc(fn.encode())
            if not addr:
                raise RuntimeError("GL entry point missing: " + fn)
            self.__dict__[fn] = ctypes.CFUNCTYPE(res, *args)(addr)
        if os.environ.get("MW_REF_FORCE_1SPP"):
            # a driver without multisampled textures: the reference's FrameBuffer takes its own `except` branch and renders
            # into a plain GL_RGBA texture with a 16-bit depth renderbuffer (opengl.py:263-284) — the "_1spp" fixtures
            def no_multisample(*a):
                raise RuntimeError("glTexImage2DMultisample: not supported (MW_REF_FORCE_1SPP)")
            self.__dict__["glTexImage2DMultisample"] = no_multisample
        self.gl_info = _GLInfo(self)
        self.gluPerspective = self._glu_perspective
        self.gluLookAt = self._glu_look_at
        self.Config = lambda **k: None

    # libGLU (SGI libutil/project.c, gluPerspective): all double, glMultMatrixd
    def _glu_perspective(self, fovy, aspect, z_near, z_far):
        radians = float(fovy) / 2 * math.pi / 180
        delta_z = float(z_far) - float(z_near)
        sine = math.sin(radians)
        if delta_z == 0 or sine == 0 or aspect == 0:
            return
        cotangent = math.cos(radians) / sine
        m = [0.0] * 16
        m[0] = cotangent / float(aspect)
        m[5] = cotangent
        m[10] = -(float(z_far) + float(z_near)) / delta_z
        m[11] = -1.0
        m[14] = -2 * float(z_near) * float(z_far) / delta_z
        self.glMultMatrixd((c_double * 16)(*m))

    # libGLU (project.c, gluLookAt): float vectors, glMultMatrixf, then glTranslated(-eye)
    def _glu_look_at(self, ex, ey, ez, cx, cy, cz, ux, uy, uz):
        f32 = np.float32

        def normalize(v):
            r = f32(np.sqrt(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2])))
            if r == 0:
                return v
            return [f32(v[0] / r), f32(v[1] / r), f32(v[2] / r)]

        def cross(a, b):
            return [f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])),
                    f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))]

        fwd = [f32(float(cx) - float(ex)), f32(float(cy) - float(ey)), f32(float(cz) - float(ez))]
        up = [f32(ux), f32(uy), f32(uz)]
        fwd = normalize(fwd)
        side = normalize(cross(fwd, up))
        up = cross(side, fwd)
        m = [0.0] * 16
        m[0], m[4], m[8] = side
        m[1], m[5], m[9] = up
        m[2], m[6], m[10] = [-fwd[0], -fwd[1], -fwd[2]]
        m[15] = 1.0
        self.glMultMatrixf((c_float * 16)(*[float(x) for x in m]))
        self.glTranslated(-float(ex), -float(ey), -float(ez))


class _Tex:
    def __init__(self, target, id_, w, h):
        self.target, self.id, self.width, self.height = target, id_, w, h


class _ImageData:
    def __init__(self, img):
        self._img = img

    def get_data(self, fmt, pitch):
        assert fmt == "RGBA" and pitch == self._img.width * 4
        return self._img._rgba_bottom_up.tobytes()


class _Image:
    """pyglet.image.load(path): an AbstractImage with the decoded RGBA pixels (rows bottom-up)."""

    def __init__(self, gl, path):
        from PIL import Image
        with Image.open(path) as im:
            rgba = np.asarray(im.convert("RGBA"), np.uint8)
        self.path = path
        self.height, self.width = rgba.shape[:2]
        self._rgba_bottom_up = np.ascontiguousarray(rgba[::-1])
        self._gl = gl

    def get_image_data(self):
        return _ImageData(self)

    def get_texture(self):
        gl = self._gl
        tid = c_uint(0)
        gl.glGenTextures(1, ctypes.byref(tid))
        gl.glBindTexture(gl.GL_TEXTURE_2D, tid.value)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MIN_FILTER, gl.GL_LINEAR)
        gl.glTexParameteri(gl.GL_TEXTURE_2D, gl.GL_TEXTURE_MAG_FILTER, gl.GL_LINEAR)
        gl.glPixelStorei(gl.GL_UNPACK_ALIGNMENT, 1)
        gl.glTexImage2D(gl.GL_TEXTURE_2D, 0, gl.GL_RGBA, self.width, self.height, 0, gl.GL_RGBA, gl.GL_UNSIGNED_BYTE,
                        self._rgba_bottom_up.ctypes.data)
        tex = _Tex(gl.GL_TEXTURE_2D, tid.value, self.width, self.height)
        tex.path = self.path
        return tex


class _VertexList:
    def __init__(self, gl, count, *attrs):
        self._gl, self.count = gl, count
        self.attrs = {fmt: np.ascontiguousarray(np.asarray(data, np.float32)) for fmt, data in attrs}

    def draw(self, mode):
        gl = self._gl
        a = self.attrs
        gl.glEnableClientState(gl.GL_VERTEX_ARRAY)
        gl.glVertexPointer(3, gl.GL_FLOAT, 0, a["v3f"].ctypes.data)
        if "t2f" in a:
            gl.glEnableClientState(gl.GL_TEXTURE_COORD_ARRAY)
            gl.glTexCoordPointer(2, gl.GL_FLOAT, 0, a["t2f"].ctypes.data)
        if "n3f" in a:
            gl.glEnableClientState(gl.GL_NORMAL_ARRAY)
            gl.glNormalPointer(gl.GL_FLOAT, 0, a["n3f"].ctypes.data)
        if "c3f" in a:
            gl.glEnableClientState(gl.GL_COLOR_ARRAY)
            gl.glColorPointer(3, gl.GL_FLOAT, 0, a["c3f"].ctypes.data)
        gl.glDrawArrays(mode, 0, self.count)
        for st in ("GL_VERTEX_ARRAY", "GL_TEXTURE_COORD_ARRAY", "GL_NORMAL_ARRAY", "GL_COLOR_ARRAY"):
            gl.glDisableClientState(getattr(gl, st))


class _Window:
    def __init__(self, *a, **k):
        pass

    def switch_to(self):
        pass

    def clear(self):
        pass

    def close(self):
        pass

    def __getattr__(self, name):
        return lambda *a, **k: None


_ref = None
gl = None


def load_reference():
    """Import the reference ``miniworld`` package with pyglet.gl bound to the real GL; returns the module."""
    global _ref, gl
    if _ref is not None:
        return _ref
    if not gl_available():
        raise RuntimeError("needs /root/reference and Mesa's swrast_dri.so")
    assert "pyglet" not in sys.modules, "refshim (stub GL) and refshim_gl cannot share a process"
    gl = _GLModule("pyglet.gl")
    pyglet = types.ModuleType("pyglet")
    pyglet.__path__ = []
    pyglet.options = {}
    pyglet.gl = gl
    pyglet.window = types.SimpleNamespace(Window=_Window)
    pyglet.text = types.SimpleNamespace(Label=refshim._Anything)
    pyglet.image = types.SimpleNamespace(load=lambda path: _Image(gl, path), ImageData=refshim._Anything)
    pyglet.graphics = types.SimpleNamespace(vertex_list=lambda n, *attrs: _VertexList(gl, n, *attrs))
    pyglet.app = refshim._Anything()
    sys.modules["pyglet"] = pyglet
    sys.modules["pyglet.gl"] = gl
    refshim._install_stubs()      # gymnasium only: pyglet is already in sys.modules
    if refshim.REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, refshim.REFERENCE_ROOT)
    import miniworld  # noqa: the reference package
    import miniworld.envs  # noqa
    _ref = miniworld
    return miniworld


def make_env(name: str, **kwargs):
    load_reference()
    import miniworld.envs as envs
    return getattr(envs, name)(**kwargs)


def driver_info():
    load_reference()
    pos = []
    for n in (4,):
        # sample positions of an n-sample FBO as the driver reports them
        fbo, tex = c_uint(0), c_uint(0)
        gl.glGenFramebuffers(1, ctypes.byref(fbo))
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, fbo.value)
        gl.glGenTextures(1, ctypes.byref(tex))
        gl.glBindTexture(gl.GL_TEXTURE_2D_MULTISAMPLE, tex.value)
        gl.glTexImage2DMultisample(gl.GL_TEXTURE_2D_MULTISAMPLE, n, gl.GL_RGBA32F, 8, 8, 1)
        gl.glFramebufferTexture2D(gl.GL_FRAMEBUFFER, gl.GL_COLOR_ATTACHMENT0, gl.GL_TEXTURE_2D_MULTISAMPLE, tex.value, 0)
        for i in range(n):
            v = (c_float * 2)()
            gl.glGetMultisamplefv(gl.GL_SAMPLE_POSITION, i, v)
            pos.append((v[0], v[1]))
        gl.glBindFramebuffer(gl.GL_FRAMEBUFFER, 0)
    return {"renderer": gl.gl_info.get_renderer(), "version": gl.gl_info.get_version(), "sample_positions_4": pos}


if __name__ == "__main__":
    print(driver_info())
    env = make_env("Hallway")
    env.reset(seed=0)
    rgb = env.render_obs()
    dep = env.render_depth()
    print(rgb.shape, rgb.mean(axis=(0, 1)), dep.min(), dep.max(), "GL error", gl.glGetError())
