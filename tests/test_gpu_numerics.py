"""Numerical building blocks of the kernels checked exhaustively on the GPU."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fast_reciprocal_equals_ieee_division_for_every_float():
    """rcp_exact (mw_raster_common.h) == 1.0f / x, bit for bit, for all 2^32 inputs (NaNs compared as NaNs)."""
    from miniworld_amd import engine
    lib = engine.load_library()
    bad = (C.c_uint64 * 512)()
    ex = (C.c_uint32 * 64)()
    n = C.c_uint32()
    lib.mw_selftest_rcp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.mw_selftest_rcp(bad, ex, C.byref(n)) == 0
    bad = np.array(bad[:], np.uint64)
    examples = np.array(ex[:min(n.value, 64)], np.uint32).view(np.float32)
    assert n.value == 0, (f"{n.value} inputs differ; binades (sign|exponent): {np.nonzero(bad)[0].tolist()[:40]}; "
                          f"examples: {examples[:8].tolist()}")


def test_fast_division_equals_ieee_division_on_its_domain():
    """div_exact(a, rcp_exact(b), b) == a / b for 2^32 pseudo-random pairs of the domain the kernels use it on."""
    from miniworld_amd import engine
    lib = engine.load_library()
    n = C.c_uint64()
    ex = (C.c_uint32 * 64)()
    lib.mw_selftest_div.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.mw_selftest_div(C.byref(n), ex) == 0
    pairs = np.array(ex[:], np.uint32).view(np.float32).reshape(-1, 2)[:min(int(n.value), 32)]
    assert n.value == 0, f"{n.value} quotients differ, e.g. (a, b) = {pairs[:6].tolist()}"


def test_quad_kernel_shortcuts_equal_the_pinned_arithmetic_for_every_float():
    """mw_rasterq.hip: v_cvt_pk_u8_f32(acc * (255 / S)) == float_to_unorm8(acc * (1 / S)) (S = 4, 8), and the lod taken from
    rho^2's bits == mwgl::lod_from_rho2's float arithmetic (1 .. 12 levels) — for all 2^32 inputs."""
    from miniworld_amd import engine
    lib = engine.load_library()
    n = (C.c_uint64 * 2)()
    ex = (C.c_uint32 * 64)()
    lib.mw_selftest_q.argtypes = [C.c_void_p, C.c_void_p]
    assert lib.mw_selftest_q(n, ex) == 0
    ex = np.array(ex[:], np.uint32)
    assert n[0] == 0, f"unorm8: {n[0]} inputs differ, e.g. {ex[:8].view(np.float32).tolist()}"
    assert n[1] == 0, f"lod: {n[1]} inputs differ, e.g. {ex[32:40].view(np.float32).tolist()}"


def test_visiting_order_sort_sorts():
    """Big scenes' geometry kernel leaves K2 a near-to-far visiting order; K2 stops at the first triangle that lies behind
    everything its tile holds, which is only right if the order is ascending.  mw_selftest_sort runs the kernel's own sort
    (bitonic network, keys in registers: mw_geom.hip::sort_store_keys) on random keys of every length class."""
    from miniworld_amd import engine
    lib = engine.load_library()
    rng = np.random.default_rng(3)
    lens = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 511, 512] + [int(v) for v in rng.integers(1, 513, 50)]
    blocks = len(lens)
    keys = np.zeros((blocks, 512), np.uint32)
    for b, n in enumerate(lens):
        # (depth bound << 16 | list index): the indices make the keys distinct, the bounds repeat
        keys[b, :n] = (rng.integers(0, 400, n).astype(np.uint32) << 16) | rng.permutation(n).astype(np.uint32)
    n_arr = np.array(lens, np.int32)
    order = np.zeros((blocks, 513), np.uint16)
    lib.mw_selftest_sort.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    assert lib.mw_selftest_sort(keys.ctypes.data, n_arr.ctypes.data, blocks, order.ctypes.data) == 0
    for b, n in enumerate(lens):
        want = (np.sort(keys[b, :n]) & 0xFFFF).astype(np.uint16)
        assert np.array_equal(order[b, 1:1 + n], want), (n, order[b, 1:9], want[:8])

