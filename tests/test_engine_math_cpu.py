"""The engine's own vertex / fragment arithmetic (miniworld_amd/csrc/mw_glmath.h, mw_frag.h — the functions the HIP kernels
call), compiled for the host and wrapped in a plain frame loop (tests/hostcheck/mwhost.cpp, test infrastructure), against the
reference's frames on real OpenGL (tests/golden/gl_*.npz) and against the oracle.  No GPU: this is the part of the kernels
that can go wrong in the last bit; what only a GPU can show (lanes, LDS, launches) is in the -m gpu tests."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import helpers
import pyoracle
from test_oracle_vs_reference_gl import gl_cases, load_gl

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hostcheck", "mwhost.cpp")
LIB = os.path.join(HERE, "hostcheck", "libmwhost.so")
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def host():
    deps = [SRC, os.path.join(ROOT, "miniworld_amd", "csrc", "mw_glmath.h"), os.path.join(ROOT, "miniworld_amd", "csrc", "mw_frag.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", *fma, "-shared", SRC,
                               os.path.join(ROOT, "oracle", "mwo_math.c"), "-o", LIB])
    lib = C.CDLL(LIB)
    lib.mwhost_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mwhost_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    return lib


def host_render(lib, scene, nsamples, meshes, view="agent", render_agent=False, width=80, height=60):
    sc, keep = pyoracle.pack_scene(scene, width, height, nsamples, meshes, None, view, render_agent)
    rgb = np.zeros((height, width, 3), np.uint8)
    z16 = np.zeros((height, width), np.uint16)
    assert lib.mwhost_render(C.byref(sc), rgb.ctypes.data, z16.ctypes.data) == 0
    return rgb, z16


@pytest.mark.parametrize("case", gl_cases())
def test_engine_math_equals_the_reference_on_opengl(host, case):
    for k, (sc, fr) in load_gl(case).items():
        meshes = helpers.golden_meshes(sc)
        rgb, z16 = host_render(host, sc, 4, meshes)
        assert np.array_equal(z16, fr["z16"]), f"{case} frame {k}: depth"
        assert np.array_equal(rgb, fr["rgb"]), f"{case} frame {k}: {np.count_nonzero(rgb != fr['rgb'])} RGB values differ"
        top, _ = host_render(host, sc, 4, meshes, view="top", render_agent=True)
        assert np.array_equal(top, fr["top"]), f"{case} frame {k}: top view"


@pytest.mark.parametrize("case", ["hallway_s0", "pickup_dr_s1", "maze_s0", "sign_s0", "sidewalk_s0"])
def test_engine_math_equals_the_oracle_at_8_and_1_samples(host, case):
    """the sample counts llvmpipe cannot show: the engine's default (8, what the reference asks for) and 1"""
    for k, (sc, fr) in load_gl(case).items():
        meshes = helpers.golden_meshes(sc)
        for ns in (8, 1):
            want = pyoracle.render(sc, nsamples=ns, meshes=meshes)
            rgb, z16 = host_render(host, sc, ns, meshes)
            assert np.array_equal(z16, want["z16"]) and np.array_equal(rgb, want["rgb"]), f"{case} frame {k} at {ns} samples"


def test_device_sinf_cosf_restate_glibc(host):
    """Mesa's glRotatef calls glibc's sinf / cosf; the device evaluates the same algorithm (mw_glmath.h)."""
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = libm.cosf.restype = C.c_float
    libm.sinf.argtypes = libm.cosf.argtypes = [C.c_float]
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(-20, 20, 20000), rng.uniform(-1e-3, 1e-3, 2000), [0.0, 0.785398185253, 1.57079637051, 3.14159274101]]).astype(np.float32)
    s, c = C.c_float(), C.c_float()
    for x in xs:
        host.mwhost_sincosf(float(x), C.byref(s), C.byref(c))
        assert s.value == libm.sinf(float(x)) and c.value == libm.cosf(float(x)), float(x)


def test_compact_clip_vertices_clip_like_full_ones(host):
    """The geometry kernel's work lists hold mwgl::ClipVert (clip, window, texture coordinates: a flat primitive's colour
    stays in registers); the clipper instantiated on them yields the same polygon as on full vertices, bit for bit."""
    rng = np.random.default_rng(5)
    host.mwhost_clip_variants_agree.restype = C.c_int
    clipped = 0
    for _ in range(4000):
        clip = rng.normal(0, 1.5, (3, 4)).astype(np.float32)
        clip[:, 3] = rng.uniform(-0.5, 2.5, 3).astype(np.float32)
        st = rng.uniform(-2, 2, (3, 2)).astype(np.float32)
        r = host.mwhost_clip_variants_agree(clip.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), 80, 60)
        assert r >= 1, (clip, st)
        clipped += r > 1
    assert clipped > 500        # plenty of them really went through the planes
