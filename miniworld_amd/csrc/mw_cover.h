// Coverage of a SMALL triangle on an obs-sized 8-sample frame (the mesh entities' triangles: a ball is 5 192 of them on a few
// hundred pixels), host + device: tests/hostcheck/mwhost.cpp runs the same code on the CPU against the per-pixel loop.
//
// The D3D 8-sample pattern (mw_records.h) puts one sample in each of the eight columns x = (2 i + 1) / 16 of a pixel and one
// in each of its eight rows.  A sample inside the triangle lies inside the bounding box of the snapped vertices, so the
// candidates of a triangle are the sample COLUMNS of the frame that cross the box — global column ix = 8 px + i at
// x = 32 ix + 16 (24.8) — times the pixel rows the box touches: one sample each, for a sub-pixel triangle one to four
// candidates instead of 8 samples x 1-4 pixels x 3 edges.
#pragma once
#include "mw_glmath.h"

namespace mwcov {

// column i of a pixel (x = (2 i + 1) / 16): its sample's index in the pattern {9,5} {7,11} {13,9} {5,3} {3,13} {1,7} {11,15} {15,1},
// 3 bits each, and the sample's y in sixteenths, 4 bits each
constexpr uint32_t kColSample = 5u | (4u << 3) | (3u << 6) | (1u << 9) | (0u << 12) | (6u << 15) | (2u << 18) | (7u << 21);
constexpr uint32_t kColY = 7u | (13u << 4) | (3u << 8) | (11u << 12) | (5u << 16) | (15u << 20) | (9u << 24) | (1u << 28);

MW_HD int mul24i(int a, int b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __mul24(a, b);
#else
    return a * b;
#endif
}

// integer edges of a front-facing triangle in 32 bits (frames up to 128 x 96: mw_engine.hip::tile_kernels_exact), the
// snapped vertices' bounds and the depth plane: mw_glmath.h::setup_triangle_pos for a multisampled target.
struct Edges {
    int dcdx[3], dcdy[3], c[3];
    int minx, maxx, miny, maxy;
    mwgl::Plane z;
};

// wa, wb, wc: window coordinates (x, y, z, 1 / w) of the vertices in drawing order.  False: back-facing or empty.
// (edges and bounds only: what the question "does it cover any sample" needs)
MW_HD bool setup_edges_xy(const float wa[4], const float wb[4], const float wc[4], Edges &s)
{
    int fx[3], fy[3];
    fx[0] = mwgl::iround_even(wa[0] * 256.0f); fy[0] = mwgl::iround_even(wa[1] * 256.0f);
    fx[1] = mwgl::iround_even(wb[0] * 256.0f); fy[1] = mwgl::iround_even(wb[1] * 256.0f);
    fx[2] = mwgl::iround_even(wc[0] * 256.0f); fy[2] = mwgl::iround_even(wc[1] * 256.0f);
    {
        const int dx01 = fx[0] - fx[1], dy01 = fy[0] - fy[1], dx20 = fx[2] - fx[0], dy20 = fy[2] - fy[0];
        if (dx01 * dy20 - dx20 * dy01 >= 0) return false;
    }
    // front faces are set up in the order (v1, v0, v2)
    const int X[3] = {fx[1], fx[0], fx[2]}, Y[3] = {fy[1], fy[0], fy[2]};
    for (int i = 0; i < 3; ++i) {
        const int j = i == 2 ? 0 : i + 1;
        s.dcdy[i] = X[i] - X[j];
        s.dcdx[i] = Y[i] - Y[j];
        s.c[i] = s.dcdx[i] * X[i] - s.dcdy[i] * Y[i];
        s.c[i] += (s.dcdx[i] < 0 || (s.dcdx[i] == 0 && s.dcdy[i] > 0)) ? 1 : 0;
    }
    s.minx = X[0] < X[1] ? X[0] : X[1]; s.minx = X[2] < s.minx ? X[2] : s.minx;
    s.maxx = X[0] > X[1] ? X[0] : X[1]; s.maxx = X[2] > s.maxx ? X[2] : s.maxx;
    s.miny = Y[0] < Y[1] ? Y[0] : Y[1]; s.miny = Y[2] < s.miny ? Y[2] : s.miny;
    s.maxy = Y[0] > Y[1] ? Y[0] : Y[1]; s.maxy = Y[2] > s.maxy ? Y[2] : s.maxy;
    return true;
}

// ... and the depth plane of a front-facing triangle
MW_HD void setup_depth_plane(const float wa[4], const float wb[4], const float wc[4], mwgl::Plane &z)
{
    const float *w0 = wb, *w1 = wa, *w2 = wc;
    const float fdx01 = w0[0] - w1[0], fdy01 = w0[1] - w1[1], fdx20 = w2[0] - w0[0], fdy20 = w2[1] - w0[1];
    const float ooa = 1.0f / (fdx01 * fdy20 - fdx20 * fdy01);
    mwgl::plane_coef(z, w0[2], w1[2], w2[2], fdy20 * ooa, fdy01 * ooa, fdx20 * ooa, fdx01 * ooa, w0[0], w0[1]);
}

MW_HD bool setup_edges(const float wa[4], const float wb[4], const float wc[4], Edges &s)
{
    if (!setup_edges_xy(wa, wb, wc, s)) return false;
    setup_depth_plane(wa, wb, wc, s.z);
    return true;
}

// sink(px, gy, s, xs, ys) for every sample of the W x H frame inside the triangle: pixel (px, gy) (GL row), sample index s,
// the sample's position in pixels (the depth plane's coordinates).  Returns whether there was any.
template <class Sink>
MW_HD bool cover_columns(const Edges &e, int W, int H, Sink &&sink)
{
    int ix_lo = (e.minx + 15) >> 5, ix_hi = (e.maxx - 16) >> 5;         // columns with minx <= 32 ix + 16 <= maxx
    int gy_lo = e.miny >> 8, gy_hi = e.maxy >> 8;
    ix_lo = ix_lo < 0 ? 0 : ix_lo; gy_lo = gy_lo < 0 ? 0 : gy_lo;
    ix_hi = ix_hi > W * 8 - 1 ? W * 8 - 1 : ix_hi; gy_hi = gy_hi > H - 1 ? H - 1 : gy_hi;
    if (ix_lo > ix_hi) return false;
    bool any = false;
    int ix = ix_lo, gy = gy_lo;
    while (gy <= gy_hi) {
        const int i = ix & 7, px = ix >> 3;
        const int sy = (int)((kColY >> (4 * i)) & 15u);
        const int fx = ix * 32 + 16, fy = gy * 256 + sy * 16;
        if (fy >= e.miny && fy <= e.maxy) {
            const int E0 = e.c[0] + mul24i(e.dcdy[0], fy) - mul24i(e.dcdx[0], fx);
            const int E1 = e.c[1] + mul24i(e.dcdy[1], fy) - mul24i(e.dcdx[1], fx);
            const int E2 = e.c[2] + mul24i(e.dcdy[2], fy) - mul24i(e.dcdx[2], fx);
            if (E0 > 0 && E1 > 0 && E2 > 0) {
                const int s = (int)((kColSample >> (3 * i)) & 7u);
                sink(px, gy, s, (float)px + (float)(2 * i + 1) * 0.0625f, (float)gy + (float)sy * 0.0625f);
                any = true;
            }
        }
        ++ix;
        if (ix > ix_hi) { ix = ix_lo; ++gy; }
    }
    return any;
}

// The same for a triangle of several pixels: the pixels of the bounding box, the 8 samples of each against the three edges
// (thresholds thr_k[s] = dcdx_k sx_s - dcdy_k sy_s: inside <=> E_k(pixel corner) > thr_k[s]).
// the pattern's x and y in sixteenths by sample index, 4 bits each
constexpr uint32_t kSampleX = 9u | (7u << 4) | (13u << 8) | (5u << 12) | (3u << 16) | (1u << 20) | (11u << 24) | (15u << 28);
constexpr uint32_t kSampleY = 5u | (11u << 4) | (9u << 8) | (3u << 12) | (13u << 16) | (7u << 20) | (15u << 24) | (1u << 28);

MW_HD int lowest_bit(uint32_t m)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)m) - 1;
#else
    return __builtin_ctz(m);
#endif
}

// the pixels of the bounding box inside the frame; false: none
MW_HD bool pixel_box(const Edges &e, int W, int H, int &x0, int &x1, int &y0, int &y1)
{
    x0 = e.minx >> 8; x1 = e.maxx >> 8; y0 = e.miny >> 8; y1 = e.maxy >> 8;
    x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
    x1 = x1 > W - 1 ? W - 1 : x1; y1 = y1 > H - 1 ? H - 1 : y1;
    return x0 <= x1 && y0 <= y1;
}

MW_HD void make_thresholds(const Edges &e, int (&thr)[3][8])
{
    for (int k = 0; k < 3; ++k)
        for (int s = 0; s < 8; ++s)
            thr[k][s] = mul24i(e.dcdx[k], (int)((kSampleX >> (4 * s)) & 15u) * 16) - mul24i(e.dcdy[k], (int)((kSampleY >> (4 * s)) & 15u) * 16);
}

// the samples of pixel (px, gy) inside the triangle, bit s = sample s
MW_HD uint32_t pixel_mask(const Edges &e, const int (&thr)[3][8], int px, int gy)
{
    const int E0 = e.c[0] + mul24i(e.dcdy[0], gy * 256) - mul24i(e.dcdx[0], px * 256);
    const int E1 = e.c[1] + mul24i(e.dcdy[1], gy * 256) - mul24i(e.dcdx[1], px * 256);
    const int E2 = e.c[2] + mul24i(e.dcdy[2], gy * 256) - mul24i(e.dcdx[2], px * 256);
    uint32_t in = 0u;
    for (int s = 0; s < 8; ++s) in |= (E0 > thr[0][s] && E1 > thr[1][s] && E2 > thr[2][s]) ? (1u << s) : 0u;
    return in;
}

template <class Sink>
MW_HD void emit_samples(uint32_t in, int px, int gy, Sink &&sink)
{
    while (in) {
        const int s = lowest_bit(in);
        in &= in - 1u;
        sink(px, gy, s, (float)px + (float)((kSampleX >> (4 * s)) & 15u) * 0.0625f, (float)gy + (float)((kSampleY >> (4 * s)) & 15u) * 0.0625f);
    }
}

template <class Sink>
MW_HD bool cover_pixels(const Edges &e, int W, int H, Sink &&sink)
{
    int x0, x1, y0, y1;
    if (!pixel_box(e, W, H, x0, x1, y0, y1)) return false;
    int thr[3][8];
    make_thresholds(e, thr);
    bool any = false;
    int px = x0, gy = y0;
    while (gy <= y1) {
        const uint32_t in = pixel_mask(e, thr, px, gy);
        emit_samples(in, px, gy, sink);
        any |= in != 0u;
        ++px;
        if (px > x1) { px = x0; ++gy; }
    }
    return any;
}

// sample columns x pixel rows the column form would visit (cover_columns' trip count): beyond a pixel or two the pixel form,
// which tests a pixel's eight samples side by side, is the shorter loop (MW_COVER_COLUMNS_MAX)
MW_HD int column_visits(const Edges &e, int W, int H)
{
    int ix_lo = (e.minx + 15) >> 5, ix_hi = (e.maxx - 16) >> 5, gy_lo = e.miny >> 8, gy_hi = e.maxy >> 8;
    ix_lo = ix_lo < 0 ? 0 : ix_lo; gy_lo = gy_lo < 0 ? 0 : gy_lo;
    ix_hi = ix_hi > W * 8 - 1 ? W * 8 - 1 : ix_hi; gy_hi = gy_hi > H - 1 ? H - 1 : gy_hi;
    return (ix_hi < ix_lo || gy_hi < gy_lo) ? 0 : (ix_hi - ix_lo + 1) * (gy_hi - gy_lo + 1);
}

#define MW_COVER_COLUMNS_MAX 16

// Does the triangle cover any sample at all?  (The same two loops, left at the first sample found; edges and bounds only.)
MW_HD bool covers_any(const Edges &e, int W, int H)
{
    int ix_lo = (e.minx + 15) >> 5, ix_hi = (e.maxx - 16) >> 5, gy_lo = e.miny >> 8, gy_hi = e.maxy >> 8;
    ix_lo = ix_lo < 0 ? 0 : ix_lo; gy_lo = gy_lo < 0 ? 0 : gy_lo;
    ix_hi = ix_hi > W * 8 - 1 ? W * 8 - 1 : ix_hi; gy_hi = gy_hi > H - 1 ? H - 1 : gy_hi;
    if (ix_hi < ix_lo || gy_hi < gy_lo) return false;
    if ((ix_hi - ix_lo + 1) * (gy_hi - gy_lo + 1) <= MW_COVER_COLUMNS_MAX) {
        int ix = ix_lo, gy = gy_lo;
        while (gy <= gy_hi) {
            const int sy = (int)((kColY >> (4 * (ix & 7))) & 15u);
            const int fx = ix * 32 + 16, fy = gy * 256 + sy * 16;
            const int E0 = e.c[0] + mul24i(e.dcdy[0], fy) - mul24i(e.dcdx[0], fx);
            const int E1 = e.c[1] + mul24i(e.dcdy[1], fy) - mul24i(e.dcdx[1], fx);
            const int E2 = e.c[2] + mul24i(e.dcdy[2], fy) - mul24i(e.dcdx[2], fx);
            if (E0 > 0 && E1 > 0 && E2 > 0) return true;
            ++ix;
            if (ix > ix_hi) { ix = ix_lo; ++gy; }
        }
        return false;
    }
    int x0, x1, y0, y1;
    if (!pixel_box(e, W, H, x0, x1, y0, y1)) return false;
    int thr[3][8];
    make_thresholds(e, thr);
    int px = x0, gy = y0;
    while (gy <= y1) {
        if (pixel_mask(e, thr, px, gy)) return true;
        ++px;
        if (px > x1) { px = x0; ++gy; }
    }
    return false;
}

template <class Sink>
MW_HD bool cover(const Edges &e, int W, int H, Sink &&sink)
{
    return column_visits(e, W, H) <= MW_COVER_COLUMNS_MAX ? cover_columns(e, W, H, sink) : cover_pixels(e, W, H, sink);
}

}  // namespace mwcov
