// Per-triangle records the geometry kernel (mw_geom.hip) leaves for the raster kernels, and the sample patterns.
//
// One record triple per triangle that leaves llvmpipe's setup (mw_glmath.h::setup_triangle), in drawing order:
//   raster (MW_RASTER_REC = 64 dwords, read through scalar loads by the wavefront that visits the triangle)
//     [0..2]  A_k = -(dcdx_k << 8)      edge value at GL pixel (px, gy), sample s:  A_k px + B_k gy + C_k > thr_k[s]
//     [3..5]  B_k =   dcdy_k << 8
//     [6..8]  C_k (low 32 bits; all of it for frames up to 128 x 96)     [9] draw id
//     [10..12] z plane a0, dadx, dady                                    [13..15] tmax_k = max_s thr_k[s]
//     [16 + 16 k + s]  thr_k[s] = dcdx_k sx_s - dcdy_k sy_s  (sx, sy: the sample's 24.8 offset inside the pixel)
//   shade (MW_SHADE_REC = 32 dwords, read per lane): w plane + texture id, s, t, r, g, b planes, z plane
//   cull (MW_CULL_REC = 24 dwords, read per lane): A, B, C, tmin, tmax, tile bounds, 16-bit depth lower bound, C high words
#pragma once
#include "mw_device.h"
#include "mw_glmath.h"

namespace mwrec {

// sample positions in 1/16 pixel inside the pixel, GL frame-buffer space (y up): what llvmpipe reports for its 4-sample
// buffers (glGetMultisamplefv), the D3D standard patterns for 8 / 16 (no llvmpipe counterpart)
__device__ __constant__ const unsigned char kPat[4][16][2] = {
    {{8, 8}},
    {{6, 2}, {14, 6}, {2, 10}, {10, 14}},
    {{9, 5}, {7, 11}, {13, 9}, {5, 3}, {3, 13}, {1, 7}, {11, 15}, {15, 1}},
    {{9, 9}, {7, 5}, {5, 10}, {12, 7}, {3, 6}, {10, 13}, {13, 11}, {11, 3}, {6, 14}, {8, 1}, {4, 2}, {2, 12}, {0, 8}, {15, 4}, {14, 15}, {1, 0}}};

__host__ __device__ inline int pat_index(int S) { return S == 1 ? 0 : (S == 4 ? 1 : (S == 8 ? 2 : 3)); }

// sample s of an S-sample buffer: 24.8 offset inside the pixel (0 for a single-sampled buffer, whose snapped coordinates
// are relative to pixel centres)
__device__ inline void sample_offset(int S, int s, int &sx, int &sy)
{
    if (S == 1) { sx = 0; sy = 0; return; }
    const int pi = pat_index(S);
    sx = kPat[pi][s][0] * 16; sy = kPat[pi][s][1] * 16;
}

// the pattern as compile-time constants (the record writer's thresholds fold to multiply-adds)
template <int S> struct Pat;
template <> struct Pat<1> { static constexpr int x[1] = {8}, y[1] = {8}; };
template <> struct Pat<4> { static constexpr int x[4] = {6, 14, 2, 10}, y[4] = {2, 6, 10, 14}; };
template <> struct Pat<8> { static constexpr int x[8] = {9, 7, 13, 5, 3, 1, 11, 15}, y[8] = {5, 11, 9, 3, 13, 7, 15, 1}; };
template <> struct Pat<16> {
    static constexpr int x[16] = {9, 7, 5, 12, 3, 10, 13, 11, 6, 8, 4, 2, 0, 15, 14, 1}, y[16] = {9, 5, 10, 7, 6, 13, 11, 3, 14, 1, 2, 12, 8, 4, 15, 0};
};

// records of one triangle.  S: samples per pixel; draw_id: position in the frame's drawing order
// (returns the 16-bit lower bound of the triangle's depth, the key of the big scenes' visiting order)
template <int S>
__device__ inline uint32_t write_tri_s(const MwArgs &a, int env, int idx, uint32_t draw_id, const mwgl::TriSetup &t, int tex)
{
    float4 *rr = reinterpret_cast<float4 *>(a.rec_raster + ((size_t)env * a.max_vis + idx) * MW_RASTER_REC);
    float4 *sr = reinterpret_cast<float4 *>(a.rec_shade + ((size_t)env * a.max_vis + idx) * MW_SHADE_REC);
    float4 *cr = reinterpret_cast<float4 *>(a.rec_cull + ((size_t)env * a.max_vis + idx) * MW_CULL_REC);
    int A[3], B[3], tmin[3], tmax[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A[k] = -(t.dcdx[k] * 256);
        B[k] = t.dcdy[k] * 256;
        int mn = 0x7fffffff, mx = (int)0x80000000;
        int thr[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            // (a single-sampled buffer's snapped coordinates are relative to pixel centres: offset 0)
            const int sx = S == 1 ? 0 : Pat<S>::x[s < S ? s : 0] * 16, sy = S == 1 ? 0 : Pat<S>::y[s < S ? s : 0] * 16;
            const int v = t.dcdx[k] * sx - t.dcdy[k] * sy;
            thr[s] = s < S ? v : 0x7fffffff;
            if (s < S) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
        }
        tmin[k] = mn; tmax[k] = mx;
        // (the readers of an S-sample frame look at the first S thresholds only)
#pragma unroll
        for (int q = 0; q < (S + 3) / 4; ++q)
            rr[4 + 4 * k + q] = make_float4(__int_as_float(thr[4 * q]), __int_as_float(thr[4 * q + 1]), __int_as_float(thr[4 * q + 2]),
                                            __int_as_float(thr[4 * q + 3]));
    }
    const int clo[3] = {(int)(uint32_t)t.c[0], (int)(uint32_t)t.c[1], (int)(uint32_t)t.c[2]};
    const int chi[3] = {(int)(t.c[0] >> 32), (int)(t.c[1] >> 32), (int)(t.c[2] >> 32)};
    // tile bounds from the snapped vertices: a pixel whose samples the triangle can cover
    const int off = S == 1 ? 128 : 0;       // single-sampled: coordinates are relative to pixel centres
    int px0 = (t.minx + off) >> 8, px1 = (t.maxx + off) >> 8, gy0 = (t.miny + off) >> 8, gy1 = (t.maxy + off) >> 8;
    px0 = px0 < 0 ? 0 : px0; gy0 = gy0 < 0 ? 0 : gy0;
    px1 = px1 > a.W - 1 ? a.W - 1 : px1; gy1 = gy1 > a.H - 1 ? a.H - 1 : gy1;
    // image rows: py = H - 1 - gy
    const int py0 = a.H - 1 - gy1, py1 = a.H - 1 - gy0;
    uint32_t bbox = 0x000000ffu | (0xffu << 16);            // empty
    if (px0 <= px1 && gy0 <= gy1)
        bbox = (uint32_t)(px0 / MW_TILE_W) | ((uint32_t)(px1 / MW_TILE_W) << 8) | ((uint32_t)(py0 / MW_TILE_H) << 16) | ((uint32_t)(py1 / MW_TILE_H) << 24);
    // 16-bit lower bound of the triangle's depth: the plane at the vertex-bounds corner where it is smallest, two steps of slack
    const float bx0 = (float)t.minx * (1.0f / 256.0f), bx1 = (float)t.maxx * (1.0f / 256.0f);
    const float by0 = (float)t.miny * (1.0f / 256.0f), by1 = (float)t.maxy * (1.0f / 256.0f);
    const float zl = fmaf(t.z.dady, t.z.dady > 0.0f ? by0 : by1, fmaf(t.z.dadx, t.z.dadx > 0.0f ? bx0 : bx1, t.z.a0));
    int zlo = (int)(zl * 65535.0f) - 3;
    zlo = zlo < 0 ? 0 : (zlo > 65535 ? 65535 : zlo);
    if (!(zl == zl)) zlo = 0;
    cr[0] = make_float4(__int_as_float(A[0]), __int_as_float(A[1]), __int_as_float(A[2]), __uint_as_float(bbox));
    cr[1] = make_float4(__int_as_float(B[0]), __int_as_float(B[1]), __int_as_float(B[2]), 0.0f);
    cr[2] = make_float4(__int_as_float(clo[0]), __int_as_float(clo[1]), __int_as_float(clo[2]), __uint_as_float((uint32_t)zlo));
    cr[3] = make_float4(__int_as_float(tmin[0]), __int_as_float(tmin[1]), __int_as_float(tmin[2]), 0.0f);
    cr[4] = make_float4(__int_as_float(tmax[0]), __int_as_float(tmax[1]), __int_as_float(tmax[2]), 0.0f);
    cr[5] = make_float4(__int_as_float(chi[0]), __int_as_float(chi[1]), __int_as_float(chi[2]), 0.0f);
    rr[0] = make_float4(__int_as_float(A[0]), __int_as_float(A[1]), __int_as_float(A[2]), __int_as_float(B[0]));
    rr[1] = make_float4(__int_as_float(B[1]), __int_as_float(B[2]), __int_as_float(clo[0]), __int_as_float(clo[1]));
    rr[2] = make_float4(__int_as_float(clo[2]), __uint_as_float(draw_id), t.z.a0, t.z.dadx);
    rr[3] = make_float4(t.z.dady, __int_as_float(tmax[0]), __int_as_float(tmax[1]), __int_as_float(tmax[2]));
    sr[0] = make_float4(t.w.a0, t.w.dadx, t.w.dady, __int_as_float(tex));
    sr[1] = make_float4(t.s.a0, t.s.dadx, t.s.dady, __uint_as_float(draw_id));
    sr[2] = make_float4(t.t.a0, t.t.dadx, t.t.dady, 0.0f);
    sr[3] = make_float4(t.col[0].a0, t.col[0].dadx, t.col[0].dady, 0.0f);
    sr[4] = make_float4(t.col[1].a0, t.col[1].dadx, t.col[1].dady, 0.0f);
    sr[5] = make_float4(t.col[2].a0, t.col[2].dadx, t.col[2].dady, 0.0f);
    sr[6] = make_float4(t.z.a0, t.z.dadx, t.z.dady, 0.0f);
    sr[7] = make_float4(__int_as_float(chi[0]), __int_as_float(chi[1]), __int_as_float(chi[2]), 0.0f);
    return (uint32_t)zlo;
}

__device__ inline uint32_t write_tri(const MwArgs &a, int env, int idx, uint32_t draw_id, const mwgl::TriSetup &t, int tex, int S)
{
    if (S == 8) return write_tri_s<8>(a, env, idx, draw_id, t, tex);
    if (S == 4) return write_tri_s<4>(a, env, idx, draw_id, t, tex);
    if (S == 1) return write_tri_s<1>(a, env, idx, draw_id, t, tex);
    return write_tri_s<16>(a, env, idx, draw_id, t, tex);
}

}  // namespace mwrec
