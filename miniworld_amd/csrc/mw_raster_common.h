// Device functions shared by the raster kernels: coverage, depth, shading, texturing, resolve of one 16x4 tile under the
// pinned GL rules (mw_frag.h: llvmpipe's fragment pipeline, measured), from the triangle records of mw_records.h.
#pragma once
#include "mw_device.h"
#include "mw_frag.h"
#include "mw_records.h"
#include <type_traits>

namespace {

struct RGB { float r, g, b; };

// 1 / x for MW_RCP_LO <= x <= MW_RCP_HI, correctly rounded like the IEEE division llvmpipe performs: hardware estimate
// (1 ulp) + one fused Newton step, 3 instructions instead of the 11 of the compiler's division sequence.  Equality with
// 1.0f / x is measured over all 2^32 bit patterns by mw_selftest_rcp (tests/test_gpu_numerics.py), not assumed.  The
// callers' arguments are an interpolated 1/w (w in [0.04, 100]) and a q within an ulp of 1.
#define MW_RCP_LO 1e-30f
#define MW_RCP_HI 1e30f
__device__ inline bool rcp_domain(float x) { return fabsf(x) >= MW_RCP_LO && fabsf(x) <= MW_RCP_HI; }
__device__ inline float rcp_exact(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, y, 1.0f), y, y);
}
__device__ inline float rcp_safe(float x) { return rcp_domain(x) ? rcp_exact(x) : 1.0f / x; }
// the same for a whole wavefront: one uniform branch instead of both sequences and a select per lane
__device__ inline float rcp_wave(float x)
{
    if (__all(rcp_domain(x))) return rcp_exact(x);
    return 1.0f / x;
}

// a / b given y = rcp_exact(b) (Markstein); kept for the mesh kernel and the self test
#define MW_DIV_LO 1e-10f
#define MW_DIV_HI 1e10f
__device__ inline bool div_domain(float b) { return b >= MW_DIV_LO && b <= MW_DIV_HI; }
__device__ inline float div_exact(float a, float y, float b)
{
    const float q = a * y;
    return fmaf(fmaf(-b, q, a), y, q);
}

typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MW_RSRC_WORD3 0x00020000

// x / d for x < 2^16 as a multiply-high by m = ceil(2^32 / d).  d == 1 has no 32-bit m (it would be 2^32 and the sum wraps
// to 0 — an env with ONE visible triangle, a 16-pixel-wide frame): m = 0 stands for "divide by one".
__device__ inline uint32_t mw_magic16(uint32_t d) { return d > 1u ? 0xFFFFFFFFu / d + 1u : 0u; }
__device__ inline uint32_t mw_div16(uint32_t x, uint32_t m) { return m ? __umulhi(x, m) : x; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline uint32_t ldw(rsrc_t r, uint32_t dword_index)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, dword_index << 2, 0, 0);
}

struct TexEnv {
    rsrc_t td, tx;
    const MwTexDesc *__restrict__ texd;
    int flat;
};

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
// mwgl::lerp8 on two 16-bit lanes at once: a + ((w (b - a) + 128) >> 8) == (a (256 - w) + b w + 128) >> 8, which stays
// below 2^16 for 8-bit a, b, w.  w, iw: the weight and 256 - weight in both halves.
__device__ inline uint32_t lerp8_pk(uint32_t a, uint32_t b, uint32_t w, uint32_t iw)
{
    // (the rounding term comes out of a register the compiler cannot see through: b w + 128 and a iw + that are two
    // packed multiply-adds; a visible constant is moved to the end of the sum and costs a third instruction)
    uint32_t half;
    asm("s_mov_b32 %0, 0x00800080" : "=s"(half));
    const u16x2 av = __builtin_bit_cast(u16x2, a), bv = __builtin_bit_cast(u16x2, b), wv = __builtin_bit_cast(u16x2, w), iv = __builtin_bit_cast(u16x2, iw);
    const u16x2 r = (av * iv + (bv * wv + __builtin_bit_cast(u16x2, half))) >> (u16x2)(8);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ inline uint32_t weight_pk(int w) { return (uint32_t)w | ((uint32_t)w << 16); }
// ... given A = a * 256 + 128 and D = b - a (the footprint records of mw_engine.hip::build_pyramid): (A + w D) >> 8
__device__ inline uint32_t lerp8_ad(uint32_t A, uint32_t D, uint32_t w)
{
    const u16x2 av = __builtin_bit_cast(u16x2, A), dv = __builtin_bit_cast(u16x2, D), wv = __builtin_bit_cast(u16x2, w);
    const u16x2 r = (dv * wv + av) >> (u16x2)(8);
    return __builtin_bit_cast(uint32_t, r);
}

// mwgl::bilerp_rgb on the footprint record whose index in the texel pool is rec: x first, then y; R | B << 16 and G
__device__ inline void bilerp_rec(const TexEnv &te, uint32_t rec, int wx8, int wy8, uint32_t &rb, uint32_t &g)
{
    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(te.tx, rec << 5, 0, 0);
    const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(te.tx, (rec << 5) + 16u, 0, 0);
    const uint32_t wx = weight_pk(wx8), wy = weight_pk(wy8), iy = 0x01000100u - wy;
    rb = lerp8_pk(lerp8_ad(r0.x, r0.y, wx), lerp8_ad(r1.x, r1.y, wx), wy, iy);
    g = lerp8_pk(lerp8_ad(r0.z, r0.w, wx), lerp8_ad(r1.z, r1.w, wx), wy, iy);
}

// GL_LINEAR lookup on mip level l of the texture whose descriptor starts at dword `desc`: one 32-byte footprint record
// (the texel and its right / upper / diagonal neighbours, GL_REPEAT applied: mw_engine.hip::build_pyramid), 8-bit weights
__device__ inline void fetch_level(const TexEnv &te, uint32_t desc, int l, float s, float t, int out[3])
{
    const uint32_t rec = (desc + 4u + (uint32_t)l * 8u) << 2;
    const u32x4 a4 = __builtin_amdgcn_raw_buffer_load_b128(te.td, rec, 0, 0);         // off, w, wmask, hmask
    const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(te.td, rec + 16u, 0, 0);   // fw, fh, h, -
    const int w = (int)a4.y, h = (int)b4.z;
    int i0, j0, wx, wy;
    mwgl::linear_coord(s, w, (w & (w - 1)) == 0, i0, wx);
    mwgl::linear_coord(t, h, (h & (h - 1)) == 0, j0, wy);
    uint32_t rb, g;
    bilerp_rec(te, a4.x + __umul24((uint32_t)j0, (uint32_t)w) + (uint32_t)i0, wx, wy, rb, g);
    out[0] = (int)(rb & 255u); out[1] = (int)(g & 255u); out[2] = (int)((rb >> 16) & 255u);
}

// fragment colour from a triangle's attribute planes at GL pixel (px, gy); eo: where the pixel centre sits in the planes'
// coordinates (0.5 for a multisampled target, 0 for a single-sampled one)
__device__ inline __attribute__((always_inline)) RGB shade_planes_body(const mwgl::Plane &wp, const mwgl::Plane &sp, const mwgl::Plane &tp,
                                                                       const mwgl::Plane &pr, const mwgl::Plane &pg, const mwgl::Plane &pb, int tex,
                                                                       const TexEnv &te, int px, int gy, float eo)
{
    const float x = (float)px + eo, y = (float)gy + eo;
    const float wv = mwgl::plane_at(wp, x, y);
    const float oow = rcp_safe(wv);
    RGB c = {mwgl::plane_at(pr, x, y) * oow, mwgl::plane_at(pg, x, y) * oow, mwgl::plane_at(pb, x, y) * oow};
    if (__any(tex >= 0)) {
        if (tex >= 0) {
            const float invq = rcp_safe(wv * oow);
            const float s = (mwgl::plane_at(sp, x, y) * oow) * invq, t = (mwgl::plane_at(tp, x, y) * oow) * invq;
            // the quad's three corner coordinates
            const float qx = (float)(px & ~1) + eo, qy = (float)(gy & ~1) + eo;
            float sc[3], tc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float cx = qx + (k == 1 ? 1.0f : 0.0f), cy = qy + (k == 2 ? 1.0f : 0.0f);
                const float w2 = mwgl::plane_at(wp, cx, cy);
                const float o2 = rcp_safe(w2);
                const float i2 = rcp_safe(w2 * o2);
                sc[k] = (mwgl::plane_at(sp, cx, cy) * o2) * i2;
                tc[k] = (mwgl::plane_at(tp, cx, cy) * o2) * i2;
            }
            const uint32_t desc = (uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4);
            const uint32_t nlevels = ldw(te.td, desc + 2u);
            const u32x4 l0r = __builtin_amdgcn_raw_buffer_load_b128(te.td, (desc + 8u) << 2, 0, 0);       // lvl[0]: fw, fh, h, -
            int l0, w8;
            mwgl::lod_select(sc[0], tc[0], sc[1], tc[1], sc[2], tc[2], __uint_as_float(l0r.x), __uint_as_float(l0r.y), (int)nlevels, l0, w8);
            int c0[3];
            fetch_level(te, desc, l0, s, t, c0);
            if (w8 > 0) {
                int c1[3];
                const int l1 = l0 + 1 > (int)nlevels - 1 ? (int)nlevels - 1 : l0 + 1;
                fetch_level(te, desc, l1, s, t, c1);
#pragma unroll
                for (int k = 0; k < 3; ++k) c0[k] = mwgl::lerp8(c0[k], c1[k], w8);
            }
            c.r = ((float)c0[0] * (1.0f / 255.0f)) * c.r;
            c.g = ((float)c0[1] * (1.0f / 255.0f)) * c.g;
            c.b = ((float)c0[2] * (1.0f / 255.0f)) * c.b;
        }
    }
    return c;
}
// (out of line: the tile kernels call it from several places; arguments by reference, i.e. through the caller's frame)
__device__ __attribute__((noinline)) RGB shade_planes(const mwgl::Plane &wp, const mwgl::Plane &sp, const mwgl::Plane &tp, const mwgl::Plane &pr,
                                   const mwgl::Plane &pg, const mwgl::Plane &pb, int tex, const TexEnv &te, int px, int gy, float eo)
{
    return shade_planes_body(wp, sp, tp, pr, pg, pb, tex, te, px, gy, eo);
}

// ... of the triangle with shade record sr (per lane)
__device__ inline RGB shade_frag(const float4 *sr, const TexEnv &te, int px, int gy, float eo)
{
    const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], qr = sr[3], qg = sr[4], qb = sr[5];
    const mwgl::Plane wp = {q0.x, q0.y, q0.z}, sp = {q1.x, q1.y, q1.z}, tp = {q2.x, q2.y, q2.z};
    const mwgl::Plane pr = {qr.x, qr.y, qr.z}, pg = {qg.x, qg.y, qg.z}, pb = {qb.x, qb.y, qb.z};
    const int tex = te.flat ? -1 : __float_as_int(q0.w);
    return shade_planes(wp, sp, tp, pr, pg, pb, tex, te, px, gy, eo);
}

// ---- wave-uniform shading --------------------------------------------------------------------------------------------------
// The tile kernels map lanes to pixels quad by quad: lanes 4q .. 4q+3 are the 2x2 pixel quad q of the 16x4 tile,
//   lane bit 0 = x & 1, bit 1 = image row & 1, bits 2-4 = quad column, bit 5 = quad row.
// The frame's height is a multiple of the tile's, so image-row pairs are GL's quads (gy & ~1): lane 2 of a quad is
// GL's (qx, qy), lane 3 (qx + 1, qy), lane 0 (qx, qy + 1).  When all 64 lanes shade the SAME triangle, the texture
// coordinates the lod needs at the quad's corners are the neighbours' own values: three DPP moves per coordinate.
__device__ inline int tile_col(int lane) { return ((lane >> 2) & 7) * 2 + (lane & 1); }
__device__ inline int tile_row(int lane) { return (lane >> 5) * 2 + ((lane >> 1) & 1); }

template <int L> __device__ inline float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), L * 0x55, 0xF, 0xF, true));
}

// GL_LINEAR on level l of a texture whose level-0 sizes are powers of two (lw0, lh0: their logarithms): result as
// R | B << 16 and G (low byte) — mwgl::linear_coord / bilerp_rgb with the size scaling as an exponent add
__device__ inline void fetch_level_pot(const TexEnv &te, uint32_t desc, int lw0, int lh0, int l, float s, float t, uint32_t &rb, uint32_t &ag)
{
    const uint32_t off = ldw(te.td, desc + 4u + (uint32_t)l * 8u);
    const int lw = max(lw0 - l, 0), lh = max(lh0 - l, 0);
    const int fx = (int)rintf(ldexpf(s, lw + 8)) - 128, fy = (int)rintf(ldexpf(t, lh + 8)) - 128;
    const uint32_t i0 = (uint32_t)(fx >> 8) & ((1u << lw) - 1u), j0 = (uint32_t)(fy >> 8) & ((1u << lh) - 1u);
    bilerp_rec(te, off + (j0 << lw) + i0, fx & 255, fy & 255, rb, ag);
}

// fragment colour of the triangle with shade record sr — the SAME record for all 64 lanes — at the lane's pixel of a
// multisampled target; all lanes must be active (quad_bcast)
__device__ inline RGB shade_uniform(const float4 *sr, const TexEnv &te, int px, int gy)
{
    const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], qr = sr[3], qg = sr[4], qb = sr[5];
    const int tex = te.flat ? -1 : __builtin_amdgcn_readfirstlane(__float_as_int(q0.w));
    const float x = (float)px + 0.5f, y = (float)gy + 0.5f;
    const float wv = fmaf(q0.z, y, fmaf(q0.y, x, q0.x));
    const float oow = rcp_wave(wv);
    RGB c = {fmaf(qr.z, y, fmaf(qr.y, x, qr.x)) * oow, fmaf(qg.z, y, fmaf(qg.y, x, qg.x)) * oow, fmaf(qb.z, y, fmaf(qb.y, x, qb.x)) * oow};
    if (tex < 0) return c;
    const float invq = rcp_wave(wv * oow);
    const float s = (fmaf(q1.z, y, fmaf(q1.y, x, q1.x)) * oow) * invq, t = (fmaf(q2.z, y, fmaf(q2.y, x, q2.x)) * oow) * invq;
    const float s00 = quad_bcast<2>(s), s10 = quad_bcast<3>(s), s01 = quad_bcast<0>(s);
    const float t00 = quad_bcast<2>(t), t10 = quad_bcast<3>(t), t01 = quad_bcast<0>(t);
    const MwTexDesc *d = te.texd + tex;
    const uint32_t w0 = d->w, h0 = d->h;
    const int nlevels = (int)d->nlevels;
    const uint32_t desc = (uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4);
    int l0, w8;
    mwgl::lod_select(s00, t00, s10, t10, s01, t01, (float)w0, (float)h0, nlevels, l0, w8);
    int c0[3];
    if (((w0 & (w0 - 1u)) | (h0 & (h0 - 1u))) == 0u) {
        const int lw0 = 31 - __builtin_clz(w0), lh0 = 31 - __builtin_clz(h0);
        uint32_t rb, ag;
        fetch_level_pot(te, desc, lw0, lh0, l0, s, t, rb, ag);
        if (__any(w8 > 0)) {
            uint32_t rb1, ag1;
            const int l1 = l0 + 1 > nlevels - 1 ? nlevels - 1 : l0 + 1;
            fetch_level_pot(te, desc, lw0, lh0, l1, s, t, rb1, ag1);
            const uint32_t wl = weight_pk(w8), il = 0x01000100u - wl;
            rb = lerp8_pk(rb, rb1, wl, il);
            ag = lerp8_pk(ag, ag1, wl, il);
        }
        c0[0] = (int)(rb & 255u); c0[1] = (int)(ag & 255u); c0[2] = (int)((rb >> 16) & 255u);
    } else {
        fetch_level(te, desc, l0, s, t, c0);
        if (__any(w8 > 0)) {
            int c1[3];
            const int l1 = l0 + 1 > nlevels - 1 ? nlevels - 1 : l0 + 1;
            fetch_level(te, desc, l1, s, t, c1);
#pragma unroll
            for (int k = 0; k < 3; ++k) c0[k] = mwgl::lerp8(c0[k], c1[k], w8);
        }
    }
    c.r = ((float)c0[0] * (1.0f / 255.0f)) * c.r;
    c.g = ((float)c0[1] * (1.0f / 255.0f)) * c.g;
    c.b = ((float)c0[2] * (1.0f / 255.0f)) * c.b;
    return c;
}

#ifdef MW_PERF_HOOKS
// tools/perf/k2prof.py (perf build only): cycles and event counts of the tile code's phases, summed over every tile of the launches
// [0] tiles [1] classification cycles [2] coverage / depth loop cycles [3] shading loop cycles [4] resolve / pack / store cycles
// [5] (tile, triangle) events visited [6] ... that covered a sample [7] winners shaded [8] classified chunks [9] tiles left early
// [10] cycles before a wavefront's first tile (records staged) [11] mesh key fetch cycles [12] per-lane mesh winner turns [13] their cycles [14] wavefront items
// (one line of counters per 8 192 workgroups: 800 000 atomics per step on ONE line slowed the kernel tenfold)
#define K2P_SLOTS 8192
__device__ unsigned long long g_k2prof[K2P_SLOTS][16];
#define K2P_NOW() __builtin_readcyclecounter()
#define K2P_ADD(i, v) do { if (cx.lane == 0) atomicAdd(&g_k2prof[blockIdx.x % K2P_SLOTS][i], (unsigned long long)(v)); } while (0)
#else
#define K2P_NOW() 0ull
#define K2P_ADD(i, v) do { } while (0)
#endif

struct TileCtx;
// colour of mesh triangle `id` (entry mj of the env's mesh table) at the lane's pixel: defined by the kernels that see meshes
__device__ inline RGB shade_mesh_winner(const TileCtx &cx, int mj, uint32_t id, int px, int gy);

// Everything a wavefront needs to produce one 16x4 tile of one env.
struct TileCtx {
    const float4 *s_shade;          // [nvis][shade_stride]  shade records: in place (stride 8) or staged in LDS (their 7 used quads)
    const float4 *s_cull;           // [nvis][cull_stride]   classification records: in place (6) or staged (5)
    int shade_stride, cull_stride;
    const float *__restrict__ rr_env;   // [nvis][64] raster records (scalar loads)
    uint8_t *s_pack;                // 192 B of LDS per wavefront
    const float *hdr;               // env header (mesh kernel only)
    const float *ment;              // the env's mesh-entity table
    const float *mesh_pos, *mesh_nrm, *mesh_rgb, *mesh_uv;
    const float *planes;            // the env's plane cache (mesh-aware K2; null elsewhere)
    const float *planes_xtra;       // ... and the textured meshes' fifth quads
    mwgl::Vert *clipbuf;            // view kernels: the wavefront's two clipper work lists in LDS (shade_mesh_tri)
    const float4 *slow_frags;       // the env's slow-fragment list and its length (mw_mesh_slow_kernel)
    const uint32_t *slow_head;      // [H][W] (frame stamp << 16) | newest fragment of the pixel + 1
    uint32_t slow_stamp;            // this frame's stamp (bits 16-31 of the launch flags)
    uint8_t *__restrict__ obs;
    float *__restrict__ depth;
    rsrc_t obs_rsrc;                // this env's uint8[H][W][3] frame as a raw buffer (HWC layout only)
    TexEnv te;
    float sky_r, sky_g, sky_b;
    int env, nvis, W, H, dbg, lane;
    uint64_t pre_touch, pre_full, pre_clip;
    uint64_t pre_edges;
    int have_pre;
    const uint16_t *order;          // SORTED kernels: [0] sorted flag, [1 + k] list index of the k-th nearest triangle
};

// Tile classification of triangle lp against the tile whose pixels span [pxlo, pxhi] x [gylo, gyhi] (GL rows): the edge
// value A px + B gy + C is monotone, so its extremes over the tile's pixels sit at corners:
//   touch = every edge's maximum exceeds its smallest sample threshold, full = every edge's minimum exceeds its largest.
__device__ inline void classify_prim(const float4 *cr, int pxlo, int pxhi, int gylo, int gyhi, bool &touch, bool &full,
                                     uint32_t *edge_open = nullptr)
{
    const float4 A = cr[0], B = cr[1], C = cr[2];
    const float4 TMIN = cr[3], TMAX = cr[4];
    const int ea[3] = {__float_as_int(A.x), __float_as_int(A.y), __float_as_int(A.z)};
    const int eb[3] = {__float_as_int(B.x), __float_as_int(B.y), __float_as_int(B.z)};
    const int ec[3] = {__float_as_int(C.x), __float_as_int(C.y), __float_as_int(C.z)};
    const int tmn[3] = {__float_as_int(TMIN.x), __float_as_int(TMIN.y), __float_as_int(TMIN.z)};
    const int tmx[3] = {__float_as_int(TMAX.x), __float_as_int(TMAX.y), __float_as_int(TMAX.z)};
    touch = true; full = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int emax = ea[k] * (ea[k] > 0 ? pxhi : pxlo) + eb[k] * (eb[k] > 0 ? gyhi : gylo) + ec[k];
        const int emin = ea[k] * (ea[k] > 0 ? pxlo : pxhi) + eb[k] * (eb[k] > 0 ? gylo : gyhi) + ec[k];
        touch &= emax > tmn[k];
        const bool inside = emin > tmx[k];
        full &= inside;
        if (edge_open) *edge_open |= inside ? 0u : (1u << k);
    }
}

// Classification of a GROUP of consecutive tiles in one pass, one (tile, triangle) pair per lane.
__device__ inline void classify_group(const float4 *s_cull, int cull_stride, int lane, int nvis, int tile0, int G, int tiles_x, int H,
                                      uint64_t &T, uint64_t &F, uint64_t (&Eo)[3])
{
    const uint32_t inv_n = (65536u + (uint32_t)nvis - 1u) / (uint32_t)nvis;      // lane / nvis, exact for lane < 64
    const int g = (int)(((uint32_t)lane * inv_n) >> 16);
    const int p = lane - g * nvis;
    bool touch = false, full = false;
    uint32_t eo = 0u;
    if (g < G) {
        const uint32_t idx = (uint32_t)(tile0 + g);
        const uint32_t ty = mw_div16(idx, mw_magic16((uint32_t)tiles_x));        // idx / tiles_x, exact for idx < 2^16
        const uint32_t tx = idx - ty * (uint32_t)tiles_x;
        const int pxlo = (int)(tx * MW_TILE_W), pxhi = pxlo + MW_TILE_W - 1;
        const int gyhi = H - 1 - (int)(ty * MW_TILE_H), gylo = gyhi - (MW_TILE_H - 1);
        classify_prim(s_cull + p * cull_stride, pxlo, pxhi, gylo, gyhi, touch, full, &eo);
    }
    T = __ballot(touch); F = __ballot(full);
#pragma unroll
    for (int k = 0; k < 3; ++k) Eo[k] = __ballot((eo >> k) & 1u);
}

__device__ inline int ffs_mask(uint32_t m) { return __ffs((int)m); }
__device__ inline int ffs_mask(uint64_t m) { return __ffsll((unsigned long long)m); }

// the eight sample offsets inside the pixel as floats (D3D standard 8x pattern, GL space)
__device__ inline float samp_x8(int s) { return (float)mwrec::kPat[2][s][0] * 0.0625f; }
__device__ inline float samp_y8(int s) { return (float)mwrec::kPat[2][s][1] * 0.0625f; }

// 16-bit depth of the triangle with z plane (a0, dadx, dady) at sample s of GL pixel (px, gy)
// (mwgl::z_to_unorm16 with the clamp as one median instruction: the same value for every z that is not a NaN, and a
// finite plane has none)
__device__ inline uint32_t depth16(float a0, float dadx, float dady, int px, int gy, int s)
{
    const float xs = (float)px + samp_x8(s), ys = (float)gy + samp_y8(s);
    const float z = __builtin_amdgcn_fmed3f(fmaf(dady, ys, fmaf(dadx, xs, a0)), 0.0f, 1.0f);
    return __float_as_uint(z * (65535.0f / 65536.0f) + 128.0f) & 0xffffu;
}

__device__ inline uint32_t to_u8(float acc) { return mwgl::float_to_unorm8(acc * 0.125f); }

// lanes whose bit is set in the wave mask m take a, the others keep b: one v_cndmask with the mask in an SGPR pair
__device__ inline float sel_mask(uint64_t m, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}
__device__ inline uint32_t sel_mask(uint64_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
}

// the eight samples' colours of one pixel; the resolve is their sum in sample order (u_blitter's resolve shader)
struct Samples { float r[8], g[8], b[8]; };
__device__ inline RGB resolve_samples(const Samples &q)
{
    RGB acc = {q.r[0], q.g[0], q.b[0]};
#pragma unroll
    for (int s = 1; s < 8; ++s) { acc.r = acc.r + q.r[s]; acc.g = acc.g + q.g[s]; acc.b = acc.b + q.b[s]; }
    return acc;
}

// Is tile (tx, ty) inside the tile rectangle of one of the env's mesh entities (env header, mw_geom.hip)?  Those tiles
// belong to the mesh kernel, all others to K2.
__device__ inline bool tile_in_mesh_rect(const float *hdr, int tx, int ty)
{
    const int n_mesh = __float_as_int(hdr[3]);
    bool in = false;
    for (int j = 0; j < n_mesh; ++j) {
        const uint32_t r = __float_as_uint(hdr[MW_HDR_MESH + MW_HDR_MESH_STRIDE * j + 26]);
        in |= tx >= (int)(r & 255u) && tx <= (int)((r >> 8) & 255u) && ty >= (int)((r >> 16) & 255u) && ty <= (int)(r >> 24);
    }
    return in;
}

// MESH kernels: is draw id `id` a mesh triangle?  (uniform scan of the env's mesh table.)  Returns the table entry or -1,
// and the record index of a non-mesh id (draw ids count every mesh triangle drawn before).
__device__ inline int mesh_entry_of(const TileCtx &cx, uint32_t id, int &rec)
{
    const int n_mesh = __float_as_int(cx.hdr[3]);
    rec = (int)id;
    for (int j = 0; j < n_mesh; ++j) {
        const int start = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * j + 1]);
        const int nt = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * j + 2]);
        if ((int)id >= start + nt) rec -= nt;
        else if ((int)id >= start) return j;
    }
    return -1;
}

// FMT: output layout fixed at compile time (0: the plain observation) or -1: read from the launch flags.
// SORTED: the env's triangles come with a visiting order by ascending depth bound (big scenes).
// HOT: 0 = everything read from the launch (debug flags, depth or not); 1 / 2 = production instantiations RGB / RGB + depth.
// PRE: 1 = the tile's classification masks come from classify_group (at most 32 triangles), 0 = classified here, -1 = cx.have_pre decides
template <bool MESH, int FMT, bool SORTED = false, int HOT = 0, int PRE = -1>
__device__ inline void raster_tile_fmt(const TileCtx &cx, int tx, int ty, const uint32_t *mesh_key)
{
    typedef typename std::conditional<PRE == 1, uint32_t, uint64_t>::type pmask_t;
    const bool have_pre = PRE < 0 ? cx.have_pre != 0 : PRE == 1;
    const int lane = cx.lane, nvis = cx.nvis, dbg = HOT ? 0 : cx.dbg, env = cx.env, W = cx.W, H = cx.H;
    const float4 *s_shade = cx.s_shade, *s_cull = cx.s_cull;
    const float *__restrict__ rr_env = cx.rr_env;
    uint8_t *s_pack = cx.s_pack;
    uint8_t *__restrict__ obs = cx.obs;
    float *__restrict__ depth = HOT == 1 ? nullptr : cx.depth;
    const bool has_depth = HOT == 2 ? true : (HOT == 1 ? false : depth != nullptr);
    const TexEnv &te = cx.te;
    const RGB sky = {cx.sky_r, cx.sky_g, cx.sky_b};
    const int colx = tile_col(lane), row = tile_row(lane);
    const int px = tx * MW_TILE_W + colx, py = ty * MW_TILE_H + row;
    const int gy = H - 1 - py;                              // GL row (the frame buffer's y points up, the image's down)
    const int pxlo = tx * MW_TILE_W, pxhi = pxlo + MW_TILE_W - 1;
    const int gyhi = H - 1 - ty * MW_TILE_H, gylo = gyhi - (MW_TILE_H - 1);
    RGB out = {0.0f, 0.0f, 0.0f};
    uint32_t z16 = 65535u;

    // ============ pass A: "painter without overlap" ==================================
    // As long as no sample is claimed by two triangles, depth is irrelevant: every covered sample belongs to its only
    // claimant, whose colour goes straight into the sample's registers (one v_cndmask per sample and channel, the
    // coverage ballots as selectors).  Contention abandons the tile to pass B (exact keys).
    bool exact = (dbg & 4) != 0;
    const bool sorted = SORTED && cx.order[0] != 0 && !(dbg & 64);
    if (SORTED && sorted) exact = true;
    if (MESH) {
        bool m = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) m |= mesh_key[s] != 0xFFFFFFFFu;
        exact |= __any(m) != 0;
    }
    Samples smp;
#pragma unroll
    for (int s = 0; s < 8; ++s) { smp.r[s] = sky.r; smp.g[s] = sky.g; smp.b[s] = sky.b; }
    if (!exact) {
        uint32_t covbits = 0u;
        uint64_t anycov_m = 0ull;
        for (int chunk = 0; chunk < (PRE == 1 ? 1 : ((dbg & 2) ? 0 : nvis)) && !exact; chunk += 64) {
            pmask_t todo, fullm;
            if (have_pre) {
                todo = (pmask_t)cx.pre_touch; fullm = (pmask_t)cx.pre_full;
            } else {
                const int lp = chunk + lane;
                bool touch = lp < nvis, full = touch;
                if (touch) classify_prim(s_cull + lp * cx.cull_stride, pxlo, pxhi, gylo, gyhi, touch, full);
                todo = (pmask_t)__ballot(touch);
                fullm = (pmask_t)__ballot(full);
            }
            while (todo) {
                const int bit = ffs_mask(todo) - 1;
                const int p = chunk + bit;
                todo &= todo - 1;
                uint64_t in_m[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) in_m[s] = ~0ull;
                if ((fullm >> bit) & 1) {
                    if (anycov_m) { exact = true; break; }
                    covbits = 0xFFu;
                    anycov_m = ~0ull;
                } else {
                    const int *__restrict__ rr = reinterpret_cast<const int *>(rr_env + (size_t)p * MW_RASTER_REC);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        if (PRE == 1 && !((cx.pre_edges >> (16 * k + (bit & 15))) & 1ull)) continue;   // known from the classification
                        const int E = __mul24(rr[k], px) + __mul24(rr[3 + k], gy) + rr[6 + k];
                        if (__all(E > rr[13 + k])) continue;        // tile strictly inside edge k
#pragma unroll
                        for (int s = 0; s < 8; ++s) in_m[s] &= __ballot(E > rr[16 + k * 16 + s]);
                    }
                    uint64_t any_m = 0ull;
#pragma unroll
                    for (int s = 0; s < 8; ++s) any_m |= in_m[s];
                    if (!any_m) continue;
                    uint32_t bits = 0u;
#pragma unroll
                    for (int s = 7; s >= 0; --s)
                        asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(in_m[s]) : "vcc");
                    if (__any((bits & covbits) != 0u)) { exact = true; break; }      // a sample claimed twice
                    covbits |= bits;
                    anycov_m |= any_m;
                }
                const RGB c = shade_uniform(s_shade + p * cx.shade_stride, te, px, gy);
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    smp.r[s] = sel_mask(in_m[s], c.r, smp.r[s]);
                    smp.g[s] = sel_mask(in_m[s], c.g, smp.g[s]);
                    smp.b[s] = sel_mask(in_m[s], c.b, smp.b[s]);
                }
                if (has_depth) {
                    const float4 zp = s_shade[p * cx.shade_stride + 6];
                    z16 = sel_mask(in_m[0], depth16(zp.x, zp.y, zp.z, px, gy, 0), z16);
                }
            }
        }
    }

    // ============ pass B: exact packed-key resolution =================================
#ifdef MW_PERF_HOOKS
    const unsigned long long kp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    [[maybe_unused]] unsigned long long kp_t0 = K2P_NOW(), kp_cls = 0, kp_ev = 0, kp_hit = 0, kp_win = 0, kp_chunks = 0, kp_early = 0, kp_mesh_cyc = 0, kp_mesh_n = 0;
    if (exact) {
        uint32_t key[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) key[s] = MESH ? mesh_key[s] : 0xFFFFFFFFu;
        bool done = false;
        uint32_t far16 = 0xFFFFu;
        // (the visiting order's early exit needs the tile's farthest key, and that is 0xFFFF for as long as any sample is uncovered: the samples
        // covered so far are kept as wave masks, and the maximum over keys and lanes is only taken once they are all ones)
        uint64_t cov_m[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) cov_m[s] = 0ull;
        for (int chunk = 0; chunk < (PRE == 1 ? 1 : nvis) && !done; chunk += 64) {
            pmask_t todo;
            int pidx = chunk + lane;
            uint32_t zlo = 0u;
            [[maybe_unused]] const unsigned long long kp_c0 = K2P_NOW();
            if (have_pre) {
                todo = (pmask_t)cx.pre_touch;
            } else {
                const int lp = chunk + lane;
                bool touch = lp < nvis, full = false;
                if (touch) {
                    if (SORTED && sorted) pidx = (int)cx.order[1 + lp];
                    classify_prim(s_cull + pidx * cx.cull_stride, pxlo, pxhi, gylo, gyhi, touch, full);
                    if (SORTED && sorted) zlo = __float_as_uint(s_cull[pidx * cx.cull_stride + 2].w);
                }
                todo = (pmask_t)__ballot(touch);
            }
#ifdef MW_PERF_HOOKS
            kp_cls += K2P_NOW() - kp_c0; ++kp_chunks;
#endif
            while (todo) {
                const int bit = ffs_mask(todo) - 1;
                const int p = (SORTED && sorted) ? __builtin_amdgcn_readlane(pidx, bit) : chunk + bit;
                todo &= todo - 1;
                if (SORTED && sorted) {
                    if ((uint32_t)__builtin_amdgcn_readlane((int)zlo, bit) > far16) { done = true; ++kp_early; break; }
                }
                ++kp_ev;
                const int *__restrict__ rr = reinterpret_cast<const int *>(rr_env + (size_t)p * MW_RASTER_REC);
                // (the depth plane is asked for with the edges: behind the coverage test its scalar load would be waited for on its own)
                const float za0 = __int_as_float(rr[10]), zdx = __int_as_float(rr[11]), zdy = __int_as_float(rr[12]);
                // (coverage as wave masks in scalar registers, like pass A: one compare per sample and open edge, one select
                // per sample — no per-lane flags, no branches around the samples)
                uint64_t in_m[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) in_m[s] = ~0ull;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    if (PRE == 1 && !((cx.pre_edges >> (16 * k + (bit & 15))) & 1ull)) continue;
                    const int E = __mul24(rr[k], px) + __mul24(rr[3 + k], gy) + rr[6 + k];
                    if (__all(E > rr[13 + k])) continue;
#pragma unroll
                    for (int s = 0; s < 8; ++s) in_m[s] &= __ballot(E > rr[16 + k * 16 + s]);
                }
                uint64_t any_m = 0ull;
#pragma unroll
                for (int s = 0; s < 8; ++s) any_m |= in_m[s];
                if (!any_m) continue;
                ++kp_hit;
                const uint32_t id = MESH ? (uint32_t)rr[9] : (uint32_t)p;           // the mesh kernel's keys carry draw ids
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const uint32_t k = (depth16(za0, zdx, zdy, px, gy, s) << 16) | id;
                    key[s] = sel_mask(in_m[s], min(key[s], k), key[s]);
                }
                if (SORTED && sorted) {
                    uint64_t all_m = ~0ull;
#pragma unroll
                    for (int s = 0; s < 8; ++s) { cov_m[s] |= in_m[s]; all_m &= cov_m[s]; }
                    if (all_m == ~0ull) {
                        uint32_t m = max(max(max(key[0], key[1]), max(key[2], key[3])), max(max(key[4], key[5]), max(key[6], key[7])));
                        far16 = __reduce_max_sync(~0ull, m) >> 16;
                    }
                }
            }
        }
        z16 = key[0] >> 16;
        [[maybe_unused]] const unsigned long long kp_t1 = K2P_NOW();
        // deferred shading: every distinct winner once per pixel (GL multisampling shades a pixel once per triangle).  The
        // tile's distinct winners are visited in ascending draw id, each shaded for the whole wavefront at once
        // (shade_uniform); mesh triangles, which differ from pixel to pixel, per lane, every lane its own next one.  A
        // winner's colour goes to the samples it owns.
        // (the samples' colours start from the sky HERE, behind the coverage / depth loop: 24 registers that loop does not have to carry)
#pragma unroll
        for (int s = 0; s < 8; ++s) { smp.r[s] = sky.r; smp.g[s] = sky.g; smp.b[s] = sky.b; }
        uint32_t pid[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { pid[s] = key[s] & 0xFFFFu; pid[s] = pid[s] == MW_SKY_PID ? 0x10000u : pid[s]; }
#ifdef MW_PERF_HOOKS
        if (MESH) {
            // (k2prof: the turns a per-lane winner loop would take — the most distinct winners any lane of the tile holds)
            uint32_t q[8];
            int cnt = 0;
#pragma unroll
            for (int s = 0; s < 8; ++s) q[s] = pid[s];
            for (;;) {
                uint32_t m = 0x10000u;
#pragma unroll
                for (int s = 0; s < 8; ++s) m = min(m, q[s]);
                if (m == 0x10000u) break;
                ++cnt;
#pragma unroll
                for (int s = 0; s < 8; ++s) q[s] = q[s] == m ? 0x10000u : q[s];
            }
            K2P_ADD(14, 0);
            const int mx = (int)__reduce_max_sync(~0ull, (unsigned)cnt);
            if (cx.lane == 0) atomicAdd(&g_k2prof[blockIdx.x % K2P_SLOTS][9], (unsigned long long)mx);
        }
#endif
        for (;;) {
            uint32_t mine = 0x10000u;
#pragma unroll
            for (int s = 0; s < 8; ++s) mine = min(mine, pid[s]);
            const uint32_t id = __reduce_min_sync(~0ull, mine);
            if (id == 0x10000u) break;
            ++kp_win;
            int rec = (int)id;
            const int mj = MESH ? mesh_entry_of(cx, id, rec) : -1;
            uint32_t sel = id;
            RGB c;
            if (MESH && mj >= 0) {
                // every lane whose next winner belongs to this mesh entity shades its own triangle
                const int start = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * mj + 1]), nt = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * mj + 2]);
                const bool on = (int)mine < start + nt;      // mine >= id >= start
                sel = on ? mine : 0x20000u;
                [[maybe_unused]] const unsigned long long km0 = K2P_NOW();
                c = shade_mesh_winner(cx, mj, on ? mine : id, px, gy);
#ifdef MW_PERF_HOOKS
                kp_mesh_cyc += K2P_NOW() - km0; ++kp_mesh_n;
#endif
            } else {
                c = shade_uniform(s_shade + rec * cx.shade_stride, te, px, gy);
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const bool eq = pid[s] == sel;
                smp.r[s] = eq ? c.r : smp.r[s]; smp.g[s] = eq ? c.g : smp.g[s]; smp.b[s] = eq ? c.b : smp.b[s];
                pid[s] = eq ? 0x10000u : pid[s];
            }
        }
#ifdef MW_PERF_HOOKS
        K2P_ADD(1, kp_cls); K2P_ADD(2, kp_t1 - kp_t0 - kp_cls); K2P_ADD(3, K2P_NOW() - kp_t1); K2P_ADD(15, __builtin_amdgcn_s_memrealtime() - kp_r0);
        K2P_ADD(5, kp_ev); K2P_ADD(6, kp_hit); K2P_ADD(7, kp_win); K2P_ADD(8, kp_chunks); K2P_ADD(9, kp_early); K2P_ADD(12, kp_mesh_n); K2P_ADD(13, kp_mesh_cyc);
#endif
    }
    [[maybe_unused]] const unsigned long long kp_t2 = K2P_NOW();
    out = resolve_samples(smp);
    const uint32_t R = to_u8(out.r), G = to_u8(out.g), B = to_u8(out.b);

    // ---- pack.  Output layout (mw_set_obs_layout; the reference's wrappers.py folded into the store):
    //   0  uint8 [H][W][3]      the observation itself
    //   1  uint8 [3][W][H]      PyTorchObsWrapper: observation.transpose(2, 1, 0)   (wrappers.py:24)
    //   2  double[H][W][1]      GreyscaleWrapper: 0.30 R + 0.59 G + 0.11 B in numpy's float64 (wrappers.py:44)
    const int fmt = FMT >= 0 ? FMT : ((dbg >> 8) & 3);
    if (fmt == 2) {
        const double g = (0.30 * (double)R + 0.59 * (double)G) + 0.11 * (double)B;
        reinterpret_cast<double *>(obs)[((size_t)env * H + py) * W + px] = g;
    } else {
        if (fmt == 0) {
            s_pack[row * 48 + colx * 3 + 0] = (uint8_t)R;
            s_pack[row * 48 + colx * 3 + 1] = (uint8_t)G;
            s_pack[row * 48 + colx * 3 + 2] = (uint8_t)B;
        } else {
            s_pack[0 * 64 + colx * 4 + row] = (uint8_t)R;
            s_pack[1 * 64 + colx * 4 + row] = (uint8_t)G;
            s_pack[2 * 64 + colx * 4 + row] = (uint8_t)B;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < 48) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(s_pack)[lane];
            if (fmt == 0) {
                const int r = lane / 12, d = lane % 12;
                const uint32_t voff = (uint32_t)(r * W * 3 + d * 4);
                const uint32_t soff = (uint32_t)((ty * MW_TILE_H) * W * 3 + tx * (MW_TILE_W * 3));
                __builtin_amdgcn_raw_buffer_store_b32(w, cx.obs_rsrc, voff, soff, 0);
            } else {
                const int ch = lane >> 4, c = lane & 15;
                uint8_t *dst = obs + (((size_t)env * 3 + ch) * W + (tx * MW_TILE_W + c)) * H + ty * MW_TILE_H;
                *reinterpret_cast<uint32_t *>(dst) = w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    if (has_depth) {
        // resolved depth = sample 0; get_depth_map in float32 as numpy evaluates it (opengl.py:426-431)
        const float z = (float)z16;
        const float d = z / 65535.0f;
        const float clip = (d - 0.5f) * 2.0f;
        const float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
        depth[((size_t)env * H + py) * W + px] = (float)(-2.0 * 100.0 * 0.04) / den;
    }
#ifdef MW_PERF_HOOKS
    K2P_ADD(0, 1); K2P_ADD(4, K2P_NOW() - kp_t2);
#endif
}

}  // namespace
