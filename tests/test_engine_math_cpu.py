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
    deps = [SRC] + [os.path.join(ROOT, "miniworld_amd", "csrc", h) for h in ("mw_glmath.h", "mw_frag.h", "mw_cover.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", *fma, "-shared", SRC,
                               os.path.join(ROOT, "oracle", "mwo_math.c"), "-o", LIB])
    lib = C.CDLL(LIB)
    lib.mwhost_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.mwhost_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.mwhost_lod_bits_mismatches.argtypes = [C.c_void_p, C.c_long]
    lib.mwhost_lod_bits_mismatches.restype = C.c_long
    lib.mwhost_cover_mismatches.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.POINTER(C.c_long)]
    lib.mwhost_cover_mismatches.restype = C.c_long
    return lib


@pytest.mark.parametrize("size", [(80, 60), (128, 96), (16, 4)])
def test_coverage_by_sample_columns_equals_the_per_sample_definition(host, size):
    """mw_cover.h (the mesh entity kernel's triangle setup in 32 bits and its coverage loop over the sample columns that cross
    a small triangle's bounding box) against setup_triangle_pos + the inside test at every sample of every pixel: the same
    culling, edges, bounds and depth plane, the same samples — each once — with the same 16-bit depths.  Sub-pixel triangles
    (a ball's), triangles of a few pixels, vertices ON sample positions and pixel corners (the fill rule's ties), triangles
    across the frame's border and larger than the frame."""
    W, H = size
    rng = np.random.default_rng(5)
    tris = []

    def add(centre, radius, n, snap=None):
        c = centre[:, None, :] + rng.uniform(-1, 1, (n, 3, 2)) * radius[:, None, None]
        if snap:
            c = np.round(c * snap) / snap
        z = rng.uniform(0.05, 0.999, (n, 3, 1))
        oow = rng.uniform(0.1, 10.0, (n, 3, 1))
        tris.append(np.concatenate([c, z, oow], axis=2).astype(np.float32))

    frame = np.array([W, H], np.float64)
    for radius, n in ((0.15, 30000), (0.5, 30000), (1.5, 20000), (6.0, 4000), (60.0, 400)):
        add(rng.uniform(-0.5, 1.0, (n, 2)) * frame * [1, 1] + rng.uniform(0, 1, (n, 2)) * 0, np.full(n, radius), n)
        add(rng.uniform(0, 1, (n // 2, 2)) * frame, np.full(n // 2, radius), n // 2, snap=16)        # vertices on the sample lattice
        add(rng.uniform(0, 1, (n // 4, 2)) * frame, np.full(n // 4, radius), n // 4, snap=1)         # ... on pixel corners
    win = np.ascontiguousarray(np.concatenate(tris))
    win[:, :, 0] = np.clip(win[:, :, 0], 0, W)       # unclipped vertices lie inside the viewport
    win[:, :, 1] = np.clip(win[:, :, 1], 0, H)
    covered = C.c_long(0)
    bad = host.mwhost_cover_mismatches(win.ctypes.data, len(win), W, H, C.byref(covered))
    assert bad == 0, f"{bad} of {len(win)} triangles differ"
    assert covered.value > 100_000


def test_lod_from_the_bits_of_rho2_equals_the_float_arithmetic(host):
    """mwgl::lod_from_rho2_bits (mw_rasterq.hip's lod: the conversion of exponent.mantissa to float IS the rounding of
    (float)e + (m - 1)) against mwgl::lod_from_rho2 (llvmpipe's arithmetic as the oracle states it): every exponent with
    its extreme and tie mantissas, 4 M random bit patterns, zeros, infinities, NaNs.  (The GPU test runs all 2^32.)"""
    rng = np.random.default_rng(0)
    mant = np.array([0, 1, 2, 3, 0x3FFFFF, 0x400000, 0x400001, 0x7FFFFE, 0x7FFFFF, 0x0FFFF, 0x10000, 0x7F8000, 0x7FFF80, 0x7FFFC0], np.uint32)
    edges = (np.arange(256, dtype=np.uint32)[:, None] << np.uint32(23)) | mant[None, :]
    edges = np.concatenate([edges.ravel(), edges.ravel() | np.uint32(0x80000000)])
    rand = rng.integers(0, 2 ** 32, 4_000_000, dtype=np.uint64).astype(np.uint32)
    # the range where the weight's low bits depend on the rounding: rho2 in [1, 2^24)
    near = (rng.integers(127, 151, 1_000_000, dtype=np.uint64).astype(np.uint32) << np.uint32(23)) | rng.integers(0, 2 ** 23, 1_000_000, dtype=np.uint64).astype(np.uint32)
    bits = np.ascontiguousarray(np.concatenate([edges, rand, near]))
    assert host.mwhost_lod_bits_mismatches(bits.ctypes.data, len(bits)) == 0


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


def test_edge_parallel_clipper_gives_the_serial_clippers_polygon(host):
    """The geometry kernel clips with eight lanes to a triangle, an edge of the polygon per lane, one step per frustum
    plane (mw_geom.hip); that algorithm, with the lanes as a loop over the same per-vertex functions, yields
    clip_triangle's polygon bit for bit — vertex count, order, clip / window / texture coordinates."""
    rng = np.random.default_rng(9)
    host.mwhost_edge_parallel_clip_agrees.restype = C.c_int
    clipped = many = 0
    for _ in range(6000):
        clip = rng.normal(0, 1.5, (3, 4)).astype(np.float32)
        clip[:, 3] = rng.uniform(-0.5, 2.5, 3).astype(np.float32)
        if rng.random() < 0.2:      # big triangles around the whole frustum: many planes, many vertices
            clip[:, :3] *= 6.0
        st = rng.uniform(-2, 2, (3, 2)).astype(np.float32)
        r = host.mwhost_edge_parallel_clip_agrees(clip.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), 80, 60)
        assert r >= 1, (clip, st)
        clipped += r > 1
        many += r > 6
    assert clipped > 1000 and many > 20


def test_clearly_back_facing_triangles_never_leave_setup(host):
    """Big scenes drop polygons seen from behind while sifting (mw_geom.hip): `clearly_back` must only say yes where the
    setup's own test on the snapped vertices drops the triangle — also for slivers, sub-pixel triangles and vertices a hair
    apart, where snapping to 1/256 px can turn the area's sign."""
    rng = np.random.default_rng(10)
    host.mwhost_clearly_back_is_wrong.restype = C.c_int
    host.mwhost_clearly_back.restype = C.c_int
    said_yes = 0
    for k in range(30000):
        kind = k % 4
        if kind == 0:       # anything on the 80 x 60 frame
            w = rng.uniform(-5, 85, (3, 2))
        elif kind == 1:     # slivers: the third vertex a hair off the line through the first two
            a, b = rng.uniform(0, 80, 2), rng.uniform(0, 80, 2)
            t = rng.uniform(0, 1)
            n = np.array([-(b - a)[1], (b - a)[0]]) / max(np.linalg.norm(b - a), 1e-6)
            w = np.stack([a, b, a + t * (b - a) + n * rng.normal(0, 3e-3)])
        elif kind == 2:     # sub-pixel triangles
            w = rng.uniform(0, 80, 2) + rng.normal(0, 5e-3, (3, 2))
        else:               # on the snapping grid's half steps
            w = (np.round(rng.uniform(0, 80, (3, 2)) * 256) + rng.choice([0.0, 0.5, 0.4999, 0.5001], (3, 2))) / 256
        win = np.zeros((3, 4), np.float32)
        win[:, :2] = w
        win[:, 2] = 0.5; win[:, 3] = 1.0
        for ms in (0, 1):
            assert host.mwhost_clearly_back_is_wrong(win.ctypes.data_as(C.c_void_p), ms) == 0, (win, ms)
        said_yes += host.mwhost_clearly_back(win.ctypes.data_as(C.c_void_p))
    assert said_yes > 3000      # and it does say yes for ordinary back faces


def test_a_box_outside_the_frustum_holds_only_vertices_outside_it(host):
    """Big scenes sift boxes of eight polygons before the polygons (mw_geom.hip, mwgl::box_view): a plane the box is
    outside of (with box_view's margin) has every point of the box outside in transform_vertex's arithmetic — the box may
    stand for its polygons —, and a box in front of the eye bounds its points' window x and depth (the occlusion test)."""
    rng = np.random.default_rng(11)
    host.mwhost_box_cull_contradictions.restype = C.c_int
    culled = 0
    for k in range(3000):
        eye = np.array([rng.uniform(-30, 30), rng.uniform(0.5, 2.5), rng.uniform(-30, 30)])
        yaw, pitch = rng.uniform(0, 2 * np.pi), rng.uniform(-0.5, 0.5) * (k % 3 == 0)
        d = np.array([np.cos(yaw) * np.cos(pitch), np.sin(pitch), -np.sin(yaw) * np.cos(pitch)])
        at = eye + d
        c = eye + rng.normal(0, 12, 3) if k % 2 else eye + d * rng.uniform(-3, 30) + rng.normal(0, 4, 3)
        half = rng.uniform(0.05, 4, 3)
        # boxes that graze a frustum plane are the interesting ones: shift some so that a face lies almost in a plane
        mn, mx = (c - half).astype(np.float32), (c + half).astype(np.float32)
        n = 64
        pts = rng.uniform(mn, mx, (n, 3)).astype(np.float32)
        pts[:8] = [[(mx if (j >> a) & 1 else mn)[a] for a in range(3)] for j in range(8)]       # the corners themselves
        planes = C.c_int(0)
        bad = host.mwhost_box_cull_contradictions(eye.ctypes.data_as(C.c_void_p), at.ctypes.data_as(C.c_void_p), C.c_double(60.0), 80, 60,
                                                  mn.ctypes.data_as(C.c_void_p), mx.ctypes.data_as(C.c_void_p),
                                                  pts.ctypes.data_as(C.c_void_p), n, C.byref(planes))
        assert bad == 0, (eye, at, mn, mx, planes.value)
        culled += planes.value != 0
    assert 500 < culled < 2900      # both outcomes occur

