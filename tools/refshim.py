"""Reference-under-stubs shim (FIXTURE TOOLING — never imported by the product).

Lets the reference's own, unmodified Python (``/root/reference/miniworld``) run in a
container that has neither pyglet, gymnasium nor a GL context, so that its *non-GL*
half — world generation, physics, collision, rewards, RNG consumption — can be used
to generate golden fixtures (``tools/gen_golden.py``) for ``tests/golden/``.

Recipe follows SURVEY.md Appendix D:
  * ``pyglet`` / ``pyglet.gl``: every ``gl*`` function is a no-op, every ``GL_*`` enum a
    distinct int, GL scalar types are ctypes types.
  * ``gymnasium``: ``Env.reset(seed)`` creates ``Generator(PCG64(SeedSequence(seed)))``
    exactly as gymnasium.utils.seeding.np_random does; spaces are plain records.
  * ``Texture.load`` returns a size-only object, ``FrameBuffer`` is a size-only fake.

Nothing from /root/reference is copied; it is put on ``sys.path`` read-only.
This module only works where /root/reference exists (the build container).
"""
from __future__ import annotations

import ctypes
import itertools
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("MINIWORLD_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "miniworld"))


class _Anything:
    """Callable/attribute sink used for pyglet objects that are never inspected."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __iter__(self):
        return iter(())


class _VertexList:
    """Records what ObjMesh hands to pyglet.graphics.vertex_list (objmesh.py:201-207)."""

    def __init__(self, count, *attrs):
        self.count = count
        self.attrs = {fmt: np.array(data) for fmt, data in attrs}

    def draw(self, mode):
        pass


class _Graphics:
    vertex_list = staticmethod(lambda count, *attrs: _VertexList(count, *attrs))


class _GLModule(types.ModuleType):
    _ctypes = {
        "GLfloat": ctypes.c_float,
        "GLubyte": ctypes.c_ubyte,
        "GLuint": ctypes.c_uint,
        "GLint": ctypes.c_int,
        "GLushort": ctypes.c_ushort,
    }

    def __init__(self, name):
        super().__init__(name)
        self._enum = itertools.count(0x1000)
        self._cache = {}
        self.__path__ = []  # behave like a package for "from pyglet.gl import ..."

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name in self._ctypes:
            return self._ctypes[name]
        if name not in self._cache:
            if name.startswith("GL_"):
                self._cache[name] = next(self._enum)
            elif name == "gl_info":
                self._cache[name] = _Anything()
            else:  # gl*/glu* entry points
                self._cache[name] = lambda *a, **k: None
        return self._cache[name]


class _Space:
    def __init__(self, *a, **k):
        self.args, self.kwargs = a, k
        self.n = a[0] if a else k.get("n")
        self.shape = k.get("shape")
        self.dtype = k.get("dtype")


class _Env:
    metadata: dict = {}
    _np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.Generator(np.random.PCG64(np.random.SeedSequence()))
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value


class _EzPickle:
    def __init__(self, *a, **k):
        pass


def _install_stubs():
    if "pyglet" not in sys.modules:
        pyglet = types.ModuleType("pyglet")
        pyglet.options = {}
        pyglet.window = _Anything()
        pyglet.text = _Anything()
        pyglet.graphics = _Graphics()
        pyglet.image = _Anything()
        pyglet.__path__ = []
        gl = _GLModule("pyglet.gl")
        pyglet.gl = gl
        sys.modules["pyglet"] = pyglet
        sys.modules["pyglet.gl"] = gl
    if "gymnasium" not in sys.modules:
        gym = types.ModuleType("gymnasium")
        gym.__path__ = []
        gym.Env = _Env
        gym.ObservationWrapper = type("ObservationWrapper", (), {})
        gym.ActionWrapper = type("ActionWrapper", (), {})
        gym.register = lambda *a, **k: None
        gym.logger = _Anything()
        spaces = types.ModuleType("gymnasium.spaces")
        spaces.Discrete = spaces.Box = spaces.Dict = _Space
        utils = types.ModuleType("gymnasium.utils")
        utils.EzPickle = _EzPickle
        core = types.ModuleType("gymnasium.core")
        core.ObsType = object
        gym.spaces, gym.utils, gym.core = spaces, utils, core
        sys.modules.update({
            "gymnasium": gym, "gymnasium.spaces": spaces,
            "gymnasium.utils": utils, "gymnasium.core": core,
        })


class _SizeOnlyTex:
    def __init__(self, path):
        from PIL import Image
        with Image.open(path) as im:
            self.width, self.height = im.size
        self.path = path
        self.target, self.id = 0, 0


class _FakeFrameBuffer:
    def __init__(self, width, height, num_samples=1):
        self.width, self.height, self.num_samples = width, height, num_samples

    def bind(self):
        pass

    def resolve(self):
        return np.zeros((self.height, self.width, 3), np.uint8)

    def get_depth_map(self, z_near=0.04, z_far=1.0):
        return np.zeros((self.height, self.width, 1), np.float32)


_ref = None


def load_reference():
    """Import the reference ``miniworld`` package under stubs; returns the module."""
    global _ref
    if _ref is not None:
        return _ref
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import miniworld  # noqa: the reference package
    import miniworld.miniworld as mw
    import miniworld.opengl as ogl
    import miniworld.envs  # noqa
    ogl.Texture.load = classmethod(lambda cls, path: _SizeOnlyTex(path))
    ogl.FrameBuffer = _FakeFrameBuffer
    mw.FrameBuffer = _FakeFrameBuffer
    _ref = miniworld
    return miniworld


def make_env(name: str, **kwargs):
    """Construct a reference env class by name, e.g. make_env("Hallway")."""
    load_reference()
    import miniworld.envs as envs
    return getattr(envs, name)(**kwargs)
