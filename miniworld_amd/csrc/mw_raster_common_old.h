// Device functions shared by the raster kernels (mw_raster.hip, mw_raster_mesh.hip):
// texture LOD + trilinear fetch (R7, R8), fragment shading (R9), resolve conversion (R12).
#pragma once
#include "mw_device.h"
#include <type_traits>

namespace {

__device__ inline float lod_log2(float x)      // R7
{
    const uint32_t b = __float_as_uint(x);
    const int e = (int)((b >> 23) & 255u) - 127;
    const float m = __uint_as_float((b & 0x7fffffu) | 0x3f800000u);
    const float f = m - 1.0f;
    float p = -0.02528550662100315f;
    p = fmaf(p, f, 0.12010025978088379f);
    p = fmaf(p, f, -0.2759689688682556f);
    p = fmaf(p, f, 0.45654040575027466f);
    p = fmaf(p, f, -0.7179135084152222f);
    p = fmaf(p, f, 1.4425272941589355f);
    return fmaf(p, f, (float)e);
}

struct RGB { float r, g, b; };

// 1 / x for MW_RCP_LO <= x <= MW_RCP_HI, correctly rounded like the IEEE division the oracle performs: hardware
// estimate (1 ulp) + one fused Newton step, 3 instructions instead of the 11 of the compiler's division sequence
// (which also handles denormals, overflow and the specials: R7 treats W outside the range like W <= 0, in the oracle
// too, so none of them reaches this function).  Equality with 1.0f / x is measured over all 2^32 bit patterns by
// mw_selftest_rcp (tests/test_gpu_numerics.py), not assumed.
#define MW_RCP_LO 1e-30f
#define MW_RCP_HI 1e30f
__device__ inline bool rcp_domain(float x) { return x >= MW_RCP_LO && x <= MW_RCP_HI; }
__device__ inline float rcp_exact(float x)
{
    const float y = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, y, 1.0f), y, y);
}

// a / b, correctly rounded like the IEEE division, given y = rcp_exact(b) = RN(1 / b): q = RN(a y) is within an ulp of
// a / b, its residual r = a - b q is exact in an fma, and q + r y rounds to RN(a / b) (Markstein 1990: the theorem for a
// correctly rounded reciprocal).  3 instructions per quotient once the reciprocal is there (quotients by one divisor
// share it) instead of the 11 of the compiler's sequence.  Domain (nothing may overflow or go denormal on the way):
// b in [MW_DIV_LO, MW_DIV_HI] and a == 0 or |a| in [1e-25, 1e25].  The callers guard b — eye-space w of a vertex, a
// mesh scale; R4m / R11 treat values outside like the oracle does — and rely on world coordinates below 1e6 m for a.
// mw_selftest_div compares it with a / b on 2^32 pseudo-random pairs of the domain (tests/test_gpu_numerics.py).
#define MW_DIV_LO 1e-10f
#define MW_DIV_HI 1e10f
__device__ inline bool div_domain(float b) { return b >= MW_DIV_LO && b <= MW_DIV_HI; }
__device__ inline float div_exact(float a, float y, float b)
{
    const float q = a * y;
    return fmaf(fmaf(-b, q, a), y, q);
}

// Texel pool and descriptor table are read through raw buffer loads: 32-bit offsets (no 64-bit
// address arithmetic per texel), hardware bounds check, descriptor in SGPRs.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#define MW_RSRC_WORD3 0x00020000

__device__ inline uint32_t ldw(rsrc_t r, uint32_t dword_index)
{
    return __builtin_amdgcn_raw_buffer_load_b32(r, dword_index << 2, 0, 0);
}

// u8 channel -> float through the hardware byte converters.  Inline asm keeps the optimiser from
// rewriting (float)b - (float)a into an integer subtract + convert: on gfx950 every integer /
// conversion op costs twice an f32 add (tools/ubench), so 12 converts + float subtracts win.
__device__ inline float ub0(uint32_t t) { float f; asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(f) : "v"(t)); return f; }
__device__ inline float ub1(uint32_t t) { float f; asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(f) : "v"(t)); return f; }
__device__ inline float ub2(uint32_t t) { float f; asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(f) : "v"(t)); return f; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// R8: GL_LINEAR fetch on mip level `l` of the texture whose descriptor starts at dword `desc` of the table,
// GL_REPEAT, centres at +0.5.  The level record (MwTexDesc::Level) arrives with two 16-byte loads.
// POT: both dims are powers of two (wave-uniform property of the texture): wrap with a mask.
template <bool POT>
__device__ inline RGB bilinear(rsrc_t td, rsrc_t tx, uint32_t desc, int l, float uu, float vv)
{
    const uint32_t rec = (desc + 4u + (uint32_t)l * 8u) << 2;        // byte offset of lvl[l]
    const u32x4 a4 = __builtin_amdgcn_raw_buffer_load_b128(td, rec, 0, 0);         // off, w, wmask, hmask
    const u32x4 b4 = __builtin_amdgcn_raw_buffer_load_b128(td, rec + 16u, 0, 0);   // fw, fh, h, -
    const uint32_t off = a4.x, w = a4.y;
    const float x = fmaf(uu, __uint_as_float(b4.x), -0.5f), y = fmaf(vv, __uint_as_float(b4.y), -0.5f);
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    int i0 = (int)x0f, j0 = (int)y0f;
    if (POT) {
        i0 &= (int)a4.z; j0 &= (int)a4.w;
    } else {
        if (i0 < 0) i0 += (int)w;
        if (j0 < 0) j0 += (int)b4.z;
    }
    // the pool holds, per texel (i, j) of a level, its whole GL_LINEAR footprint: (i, j), (i+1, j), (i, j+1), (i+1, j+1)
    // with GL_REPEAT applied (mw_engine.hip::build_pyramid) — one 16-byte load, no neighbour indices, no second wrap
    const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(tx, (off + __umul24((uint32_t)j0, w) + (uint32_t)i0) << 4, 0, 0);
    const uint32_t t00 = q.x, t10 = q.y, t01 = q.z, t11 = q.w;
    RGB o;
    {
        const float a = ub0(t00), b = ub0(t10), c = ub0(t01), d = ub0(t11);
        const float r0f = fmaf(fx, b - a, a), r1f = fmaf(fx, d - c, c);
        o.r = fmaf(fy, r1f - r0f, r0f);
    }
    {
        const float a = ub1(t00), b = ub1(t10), c = ub1(t01), d = ub1(t11);
        const float r0f = fmaf(fx, b - a, a), r1f = fmaf(fx, d - c, c);
        o.g = fmaf(fy, r1f - r0f, r0f);
    }
    {
        const float a = ub2(t00), b = ub2(t10), c = ub2(t01), d = ub2(t11);
        const float r0f = fmaf(fx, b - a, a), r1f = fmaf(fx, d - c, c);
        o.b = fmaf(fy, r1f - r0f, r0f);
    }
    return o;
}

// Textured fragment colour (R7-R9) for the lanes whose primitive uses texture `tex` (wave-uniform:
// dims and level count live in SGPRs); the attribute planes come from the lane's shade record.
template <bool POT>
__device__ inline RGB shade_tex(const float4 q0, const float4 q1, const float4 q2, rsrc_t td, rsrc_t tx, int tex,
                                float ftw, float fth, int q, float Xc, float Yc)
{
    const float Ua = q0.x, Ub = q0.y, Uc = q0.z, Va = q0.w, Vb = q1.x, Vc = q1.y;
    const float Wa = q1.z, Wb = q1.w, Wc = q2.x;
    const uint32_t desc = (uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4);          // &texd[tex], in dwords
    const float Wq = fmaf(Wa, Xc, fmaf(Wb, Yc, Wc));
    RGB texel;
    // level selection first (all lanes), then at most two bilinear fetches
    int l0 = q, l1 = -1;
    float fr = 0.0f, u = 0.0f, v = 0.0f;
    if (rcp_domain(Wq)) {
        const float iw = rcp_exact(Wq);
        const float Uq = fmaf(Ua, Xc, fmaf(Ub, Yc, Uc));
        const float Vq = fmaf(Va, Xc, fmaf(Vb, Yc, Vc));
        u = Uq * iw; v = Vq * iw;
        const float ux = (Ua - u * Wa) * iw, uy = (Ub - u * Wb) * iw;
        const float vx = (Va - v * Wa) * iw, vy = (Vb - v * Wb) * iw;
        const float sx = ux * ftw, tx_ = vx * fth, sy = uy * ftw, ty_ = vy * fth;
        const float r2x = fmaf(sx, sx, tx_ * tx_), r2y = fmaf(sy, sy, ty_ * ty_);
        const float rho2 = r2x > r2y ? r2x : r2y;
        if (!(rho2 > 1.0f)) {
            l0 = 0;                                   // magnification: GL_LINEAR on level 0
        } else if (rho2 < 1e30f) {
            const float lam = 0.5f * lod_log2(rho2);
            const float lf = floorf(lam);
            const int li = (int)lf;
            if (li < q) { l0 = li; l1 = li + 1; fr = lam - lf; }
        }
    }
    const float uu = u - floorf(u), vv = v - floorf(v);        // GL_REPEAT, shared by both levels
    const RGB c0 = bilinear<POT>(td, tx, desc, l0, uu, vv);
    texel = c0;
    if (l1 >= 0) {
        const RGB c1 = bilinear<POT>(td, tx, desc, l1, uu, vv);
        texel.r = fmaf(fr, c1.r - c0.r, c0.r);
        texel.g = fmaf(fr, c1.g - c0.g, c0.g);
        texel.b = fmaf(fr, c1.b - c0.b, c0.b);
    }
    RGB o;
    o.r = (texel.r * (1.0f / 255.0f)) * q2.y;
    o.g = (texel.g * (1.0f / 255.0f)) * q2.z;
    o.b = (texel.b * (1.0f / 255.0f)) * q2.w;
    return o;
}

__device__ inline uint32_t to_u8(float acc)     // R12: mean of 8, clamp, round half up
{
    // clamp through v_med3_f32: one instruction instead of two compare + select pairs (acc is finite)
    const float v = __builtin_amdgcn_fmed3f(acc * 0.125f, 0.0f, 1.0f);
    return (uint32_t)(int)fmaf(v, 255.0f, 0.5f);
}

}  // namespace


// depth key of primitive `pid` at sample s of this lane's pixel, from the LDS copy of its plane
namespace {
__device__ inline uint32_t lazy_key(const float4 *s_shade, uint32_t pid, int s, float Xc, float Yc)
{
    const float4 pl = s_shade[pid * (MW_SHADE_REC / 4) + 4];                       // zx, zy, zc, -
    const float zo = reinterpret_cast<const float *>(s_shade + pid * (MW_SHADE_REC / 4) + 5)[s];
    const float zc = fmaf(pl.x, Xc, fmaf(pl.y, Yc, pl.z));
    const float t = fmaf(zc + zo, 65535.0f, 0.5f);
    return ((uint32_t)t << 16) | pid;
}

struct TexEnv {
    rsrc_t td, tx;
    const MwTexDesc *__restrict__ texd;
    int flat;
};

// Textured fragment colour for the lanes with tex >= 0: attribute planes q0 / q1 / q2.x, base colour in
// q2.yzw (GL_MODULATE).  The texture id is per lane: in the deferred shading of the exact pass the lanes of a tile hold
// floor, ceiling and wall winners at once, and a waterfall over the distinct ids (scalar descriptors) ran the whole
// fetch up to three times per round.  One pass instead: each lane reads the two descriptor words it needs (level count,
// level-0 size) itself and wraps with the general rule, which gives what the power-of-two mask gives (i0 >= -1).
__device__ inline RGB apply_texture(const float4 q0, const float4 q1, const float4 q2, int tex, const TexEnv &te,
                                    float Xc, float Yc)
{
    RGB c = {q2.y, q2.z, q2.w};                                 // untextured: the base colour
    if (tex >= 0) {
        const uint32_t desc = (uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4);
        const uint32_t nlevels = ldw(te.td, desc + 2u);
        const u32x4 l0 = __builtin_amdgcn_raw_buffer_load_b128(te.td, (desc + 8u) << 2, 0, 0);       // lvl[0]: fw, fh, h, -
        c = shade_tex<false>(q0, q1, q2, te.td, te.tx, tex, __uint_as_float(l0.x), __uint_as_float(l0.y), (int)nlevels - 1, Xc, Yc);
    }
    return c;
}

// fragment colour of the primitive with LDS shade record `sr` at the pixel centre; executed by
// the lanes that need it
__device__ inline RGB shade_prim(const float4 *sr, const TexEnv &te, float Xc, float Yc)
{
    const float4 q2 = sr[2];
    const int tex = te.flat ? -1 : __float_as_int(sr[3].x);
    if (!__any(tex >= 0)) return RGB{q2.y, q2.z, q2.w};
    return apply_texture(sr[0], sr[1], q2, tex, te, Xc, Yc);
}

// the same for a wave-uniform primitive (pass A visits one primitive at a time): its texture id is a scalar, no
// waterfall over the lanes' ids
__device__ inline RGB shade_prim_uniform(const float4 *sr, const TexEnv &te, float Xc, float Yc)
{
    const float4 q2 = sr[2];
    const int tex = te.flat ? -1 : __builtin_amdgcn_readfirstlane(__float_as_int(sr[3].x));
    if (tex < 0) return RGB{q2.y, q2.z, q2.w};
    const MwTexDesc *__restrict__ d = te.texd + tex;
    const int tw = (int)d->w, th = (int)d->h, q = (int)d->nlevels - 1;
    const float ftw = d->lvl[0].fw, fth = d->lvl[0].fh;
    if (((tw & (tw - 1)) | (th & (th - 1))) == 0) return shade_tex<true>(sr[0], sr[1], q2, te.td, te.tx, tex, ftw, fth, q, Xc, Yc);
    return shade_tex<false>(sr[0], sr[1], q2, te.td, te.tx, tex, ftw, fth, q, Xc, Yc);
}

struct TileCtx;
__device__ inline RGB shade_by_draw_id(const TileCtx &cx, uint32_t id, float Xc, float Yc);

// Everything a wavefront needs to produce one 16x4 tile of one env.
struct TileCtx {
    const float4 *s_shade;          // [nvis][8]  shade records (LDS in K2, global in the mesh kernel)
    const float4 *s_cull;           // [nvis][6]  classification records
    const float *__restrict__ rr_env;   // [nvis][64] raster records (scalar loads)
    uint8_t *s_pack;                // 192 B of LDS per wavefront
    const float *hdr;               // env header (mesh kernel only)
    const float *ment;              // the env's mesh-entity table, 12 floats per entry (mesh kernel: a copy in LDS — the
                                    // per-winner lookups of the tile phase are on its critical path; else hdr + MW_HDR_MESH)
    const float *mesh_pos, *mesh_nrm, *mesh_rgb, *mesh_uv;
    uint8_t *__restrict__ obs;
    float *__restrict__ depth;
    rsrc_t obs_rsrc;                // this env's uint8[H][W][3] frame as a raw buffer (HWC layout only)
    TexEnv te;
    float sky_r, sky_g, sky_b;
    int env, nvis, W, H, dbg, lane;
    // tile classification done ahead for a group of tiles (classify_group): valid when have_pre
    uint64_t pre_touch, pre_full, pre_clip;
    uint64_t pre_edges;             // bit 16 k + p: primitive p needs its edge k tested on this tile (PRE 1; all ones with > 16 primitives)
    int have_pre;
    unsigned long long *tprof;      // MW_K3_PROF (general mesh kernel only): [4] cycles in pass B coverage, in deferred shading, iterations, tiles
    const uint16_t *order;          // SORTED kernels: [0] sorted flag, [1 + k] list index of the k-th nearest polygon
};

// Tile classification of primitive lp against the tile whose pixel centres span [Xlo, Xhi] x [Ylo, Yhi].
// The edge function is monotone in X and Y (rounding included), so its extremes over the tile's
// pixel centres sit at corners:  touch = every edge's maximum exceeds its smallest sample threshold,
//                                full  = every edge's minimum exceeds its largest sample threshold.
// edge_open: bit k set unless the whole tile lies strictly inside edge k (full == no bit set)
__device__ inline void classify_prim(const float4 *s_cull, int lp, float Xlo, float Xhi, float Ylo, float Yhi,
                                     bool &touch, bool &full, bool &clipf, uint32_t *edge_open = nullptr)
{
    const float4 A = s_cull[lp * 6 + 0], B = s_cull[lp * 6 + 1], C = s_cull[lp * 6 + 2];
    const float4 TMIN = s_cull[lp * 6 + 3], TMAX = s_cull[lp * 6 + 4];
    const float ea[4] = {A.x, A.y, A.z, A.w}, eb[4] = {B.x, B.y, B.z, B.w}, ec[4] = {C.x, C.y, C.z, C.w};
    const float tmn[4] = {TMIN.x, TMIN.y, TMIN.z, TMIN.w}, tmx[4] = {TMAX.x, TMAX.y, TMAX.z, TMAX.w};
    touch = true; full = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float emax = fmaf(ea[k], ea[k] > 0.0f ? Xhi : Xlo, fmaf(eb[k], eb[k] > 0.0f ? Yhi : Ylo, ec[k]));
        const float emin = fmaf(ea[k], ea[k] > 0.0f ? Xlo : Xhi, fmaf(eb[k], eb[k] > 0.0f ? Ylo : Yhi, ec[k]));
        touch &= emax > tmn[k];
        const bool inside = emin > tmx[k];
        full &= inside;
        if (edge_open) *edge_open |= inside ? 0u : (1u << k);
    }
    clipf = __float_as_uint(s_cull[lp * 6 + 5].x) != 0u;
}

// Classification of a GROUP of consecutive tiles in one pass, one (tile, primitive) pair per lane:
// with the dozen primitives of a typical indoor frame a per-tile pass would leave most lanes idle.
// Lane l < G * nvis handles tile (tile0 + l / nvis), primitive l % nvis; the three ballots hold,
// for the g-th tile of the group, its masks in bits [g * nvis, (g + 1) * nvis).
// Eo[k]: the pairs whose tile is not strictly inside edge k of the primitive (the others need no test of that edge)
__device__ inline void classify_group(const float4 *s_cull, int lane, int nvis, int tile0, int G, int tiles_x,
                                      uint64_t &T, uint64_t &F, uint64_t &Cl, uint64_t (&Eo)[4])
{
    const uint32_t inv_n = (65536u + (uint32_t)nvis - 1u) / (uint32_t)nvis;      // lane / nvis, exact for lane < 64
    const int g = (int)(((uint32_t)lane * inv_n) >> 16);
    const int p = lane - g * nvis;
    bool touch = false, full = false, clipf = false;
    uint32_t eo = 0u;
    if (g < G) {
        const uint32_t idx = (uint32_t)(tile0 + g);
        const uint32_t ty = __umulhi(idx, 0xFFFFFFFFu / (uint32_t)tiles_x + 1u);  // idx / tiles_x, exact for idx < 2^16
        const uint32_t tx = idx - ty * (uint32_t)tiles_x;
        const float Xlo = (float)(tx * MW_TILE_W) + 0.5f, Xhi = Xlo + (float)(MW_TILE_W - 1);
        const float Ylo = (float)(ty * MW_TILE_H) + 0.5f, Yhi = Ylo + (float)(MW_TILE_H - 1);
        classify_prim(s_cull, p, Xlo, Xhi, Ylo, Yhi, touch, full, clipf, &eo);
    }
    T = __ballot(touch); F = __ballot(full); Cl = __ballot(clipf);
#pragma unroll
    for (int k = 0; k < 4; ++k) Eo[k] = __ballot((eo >> k) & 1u);
}

// FMT: output layout fixed at compile time (0: the plain observation, the hot path) or -1: read from
// the launch flags (the wrapper layouts; kept out of the hot instantiation)
// SORTED: the env's polygons come with a visiting order by ascending depth bound (K1, big scenes).  Tiles then
// go straight to the exact pass, walk the polygons front to back and stop as soon as every sample of the tile
// holds something nearer than the next polygon's bound — in a maze that is after a handful of the dozens of
// polygons stacked behind each other in the view.  Keys, winners and colours do not depend on the visiting order.
// HOT: 0 = everything read from the launch (debug flags, depth or not); 1 / 2 = the production instantiations
// without debug flags, RGB only / RGB + depth: the flag tests, the depth bookkeeping (HOT 1) and the SGPRs that keep
// them alive leave the kernel (the general one spills 69 SGPRs to VGPR lanes, ~9 % of its VALU instructions).
// PRE: 1 = the tile's classification masks come from classify_group (cx.pre_*; at most 32 primitives, so the
// masks are 32-bit and there is a single chunk), 0 = classified here, -1 = cx.have_pre decides
__device__ inline int ffs_mask(uint32_t m) { return __ffs((int)m); }
__device__ inline int ffs_mask(uint64_t m) { return __ffsll((unsigned long long)m); }

template <bool MESH, int FMT, bool SORTED = false, int HOT = 0, int PRE = -1>
__device__ inline void raster_tile_fmt(const TileCtx &cx, int tx, int ty, const uint32_t *mesh_key)
{
    typedef typename std::conditional<PRE == 1, uint32_t, uint64_t>::type pmask_t;      // one bit per primitive of a chunk
    const bool have_pre = PRE < 0 ? cx.have_pre != 0 : PRE == 1;
    const int lane = cx.lane, nvis = cx.nvis, dbg = HOT ? 0 : cx.dbg, env = cx.env, W = cx.W, H = cx.H;
    const float4 *s_shade = cx.s_shade, *s_cull = cx.s_cull;
    const float *__restrict__ rr_env = cx.rr_env;
    uint8_t *s_pack = cx.s_pack;
    uint8_t *__restrict__ obs = cx.obs;
    float *__restrict__ depth = HOT == 1 ? nullptr : cx.depth;
    const bool has_depth = HOT == 2 ? true : (HOT == 1 ? false : depth != nullptr);
    const TexEnv &te = cx.te;
    const float sky_r = cx.sky_r, sky_g = cx.sky_g, sky_b = cx.sky_b;
    const int px = tx * MW_TILE_W + (lane & 15), py = ty * MW_TILE_H + (lane >> 4);
    const float Xc = (float)px + 0.5f, Yc = (float)py + 0.5f;
    const float Xlo = (float)(tx * MW_TILE_W) + 0.5f, Xhi = Xlo + (float)(MW_TILE_W - 1);
    const float Ylo = (float)(ty * MW_TILE_H) + 0.5f, Yhi = Ylo + (float)(MW_TILE_H - 1);
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
    uint32_t z16 = 65535u;

    // ============ pass A: "painter without overlap" ==================================
    // As long as no sample is claimed by two primitives and no primitive can be near/far
    // clipped, depth is irrelevant: every covered sample belongs to its only claimant, and
    // visiting the primitives in ascending draw index IS the resolve order of R12.  Each
    // visit shades immediately.  Any contention abandons the tile to pass B (exact keys).
    bool exact = (dbg & 4) != 0;
    const bool sorted = SORTED && cx.order[0] != 0 && !(dbg & 64);     // MW_DEBUG_FLAGS bit 6: ignore the visiting order
    if (SORTED && sorted) exact = true;
    if (MESH) {
        bool m = false;
#pragma unroll
        for (int s = 0; s < 8; ++s) m |= mesh_key[s] != 0xFFFFFFFFu;
        exact |= __any(m) != 0;
    }
    if (!exact) {
        // The samples claimed so far live in ONE VGPR per lane (bit s = sample s of this lane's pixel) plus a
        // wave-level "anything covered" mask; a primitive's own coverage is computed in wave-uniform lane masks
        // (one v_cmp per edge and sample, combined on the SALU) and folded into the lane's bits with add-with-carry.
        // (Eight 64-bit "covered" masks in SGPRs cost 16 scalar registers the tile loop does not have: they were
        // spilled to VGPR lanes, every v_readlane / v_writelane a VALU instruction.)
        uint32_t covbits = 0u;
        uint64_t anycov_m = 0ull;
        uint32_t ncov = 0;                           // covered samples of this lane's pixel
        for (int chunk = 0; chunk < (PRE == 1 ? 1 : ((dbg & 2) ? 0 : nvis)) && !exact; chunk += 64) {
            // Tile classification, one primitive per lane.  The edge function is monotone in X
            // and Y (rounding included), so its extremes over the tile's pixel centres sit at
            // corners:  touch = every edge's maximum exceeds its smallest sample threshold,
            //           full  = every edge's minimum exceeds its largest sample threshold.
            pmask_t todo, fullm, clipm;
            if (have_pre) {
                todo = (pmask_t)cx.pre_touch; fullm = (pmask_t)cx.pre_full; clipm = (pmask_t)cx.pre_clip;
            } else {
                const int lp = chunk + lane;
                bool touch = lp < nvis, full = touch, clipf = false;
                if (touch) classify_prim(s_cull, lp, Xlo, Xhi, Ylo, Yhi, touch, full, clipf);
                todo = (pmask_t)__ballot(touch);
                fullm = (pmask_t)__ballot(full); clipm = (pmask_t)__ballot(clipf);
            }
            if (todo & clipm) { exact = true; break; }
            while (todo) {
                const int bit = ffs_mask(todo) - 1;
                const int p = chunk + bit;
                todo &= todo - 1;
                uint32_t cnt;
                bool in0;
                if ((fullm >> bit) & 1) {
                    // the whole tile lies strictly inside primitive p
                    if (anycov_m) { exact = true; break; }
                    cnt = 8u; in0 = true;
                    covbits = 0xFFu;
                    anycov_m = ~0ull;
                } else {
                    const float *__restrict__ rr = rr_env + (size_t)p * MW_RASTER_REC;
                    uint64_t in_m[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) in_m[s] = ~0ull;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (PRE == 1 && !((cx.pre_edges >> (16 * k + (bit & 15))) & 1ull)) continue;   // known from the classification
                        const float E = fmaf(rr[k], Xc, fmaf(rr[4 + k], Yc, rr[8 + k]));
                        if (__all(E > rr[57 + k])) continue;        // tile strictly inside edge k
#pragma unroll
                        for (int s = 0; s < 8; ++s) in_m[s] &= __ballot(E > rr[16 + k * 8 + s]);
                    }
                    uint64_t any_m = 0ull;
#pragma unroll
                    for (int s = 0; s < 8; ++s) any_m |= in_m[s];
                    if (!any_m) continue;
                    // this lane's eight bits: bits = 2 * bits + (lane's bit of in_m[s]), the mask going in as the carry
                    // of an add-with-carry, sample 7 first so that sample s ends up in bit s
                    uint32_t bits = 0u;
#pragma unroll
                    for (int s = 7; s >= 0; --s)
                        asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(in_m[s]) : "vcc");
                    if (__any((bits & covbits) != 0u)) { exact = true; break; }      // a sample claimed twice
                    covbits |= bits;
                    cnt = (uint32_t)__popc(bits);
                    anycov_m |= any_m;
                    in0 = (bits & 1u) != 0u;
                }
                ncov += cnt;
                if (cnt != 0u) {
                    const RGB c = shade_prim_uniform(s_shade + p * (MW_SHADE_REC / 4), te, Xc, Yc);
                    const float fc = (float)cnt;
                    acc_r = fmaf(fc, c.r, acc_r);
                    acc_g = fmaf(fc, c.g, acc_g);
                    acc_b = fmaf(fc, c.b, acc_b);
                    if (has_depth && in0) z16 = lazy_key(s_shade, (uint32_t)p, 0, Xc, Yc) >> 16;
                }
            }
        }
        if (!exact) {
            const float fs = (float)(8u - ncov);            // uncovered samples: sky, last (R12)
            acc_r = fmaf(fs, sky_r, acc_r);
            acc_g = fmaf(fs, sky_g, acc_g);
            acc_b = fmaf(fs, sky_b, acc_b);
        }
    }

    // ============ pass B: exact packed-key resolution =================================
    if (exact) {
        const bool tp = MESH && !HOT && cx.tprof != nullptr;
        const unsigned long long tp0 = tp ? __builtin_readcyclecounter() : 0ull;
        acc_r = acc_g = acc_b = 0.0f;
        uint32_t key[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) key[s] = MESH ? mesh_key[s] : 0xFFFFFFFFu;
        bool done = false;
        uint32_t far16 = 0xFFFFu;        // SORTED: the farthest depth stored in the tile (0xFFFF while a sample is empty)
        for (int chunk = 0; chunk < (PRE == 1 ? 1 : nvis) && !done; chunk += 64) {
            pmask_t todo;
            int pidx = chunk + lane;        // list index of the polygon this lane classifies
            uint32_t zlo = 0u;              // SORTED: conservative 16-bit lower bound of its depth
            if (have_pre) {
                todo = (pmask_t)cx.pre_touch;
            } else {
                const int lp = chunk + lane;
                bool touch = lp < nvis, full = false, clipf = false;
                if (touch) {
                    if (SORTED && sorted) pidx = (int)cx.order[1 + lp];
                    classify_prim(s_cull, pidx, Xlo, Xhi, Ylo, Yhi, touch, full, clipf);
                    if (SORTED && sorted) {
                        // the key formula of R6 on K1's bound (itself the smallest value the plane + offset expression
                        // takes over the polygon's tiles): no key of this polygon is below it
                        const float zb = fmaf(s_cull[pidx * 6 + 5].y, 65535.0f, 0.5f);
                        zlo = zb >= 1.0f ? (uint32_t)zb : 0u;
                    }
                }
                todo = (pmask_t)__ballot(touch);
            }
            while (todo) {
                const int bit = ffs_mask(todo) - 1;
                const int p = (SORTED && sorted) ? __builtin_amdgcn_readlane(pidx, bit) : chunk + bit;
                todo &= todo - 1;
                if (SORTED && sorted) {
                    if ((uint32_t)__builtin_amdgcn_readlane((int)zlo, bit) > far16) { done = true; break; }
                }
                const float *__restrict__ rr = rr_env + (size_t)p * MW_RASTER_REC;
                bool in[8];
#pragma unroll
                for (int s = 0; s < 8; ++s) in[s] = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (PRE == 1 && !((cx.pre_edges >> (16 * k + (bit & 15))) & 1ull)) continue;
                    const float E = fmaf(rr[k], Xc, fmaf(rr[4 + k], Yc, rr[8 + k]));
                    if (__all(E > rr[57 + k])) continue;
#pragma unroll
                    for (int s = 0; s < 8; ++s) in[s] &= E > rr[16 + k * 8 + s];
                }
                bool any = false;
#pragma unroll
                for (int s = 0; s < 8; ++s) any |= in[s];
                if (!__any(any)) continue;
                const float zc = fmaf(rr[12], Xc, fmaf(rr[13], Yc, rr[14]));
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const float zs = zc + rr[48 + s];
                    const float t = fmaf(zs, 65535.0f, 0.5f);
                    const bool ok = in[s] && t >= 0.5f && t < 65536.0f;
                    const uint32_t id = MESH ? __float_as_uint(rr[61]) : (uint32_t)p;     // draw id
                    const uint32_t k = ((uint32_t)t << 16) | id;
                    key[s] = ok ? min(key[s], k) : key[s];
                }
                if (SORTED && sorted) {
                    uint32_t m = max(max(max(key[0], key[1]), max(key[2], key[3])), max(max(key[4], key[5]), max(key[6], key[7])));
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
                    far16 = m >> 16;
                }
            }
        }
        z16 = key[0] >> 16;
        uint32_t pid[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) pid[s] = key[s] & 0xFFFFu;
        // deferred shading: each distinct winner once, ascending draw index, sky last (R9, R12)
        const unsigned long long tp1 = tp ? __builtin_readcyclecounter() : 0ull;
        int tp_it = 0;
        for (;;) {
            const uint32_t sel = min(min(min(pid[0], pid[1]), min(pid[2], pid[3])), min(min(pid[4], pid[5]), min(pid[6], pid[7])));
            const bool active = sel != 0x10000u;
            if (!__any(active)) break;
            ++tp_it;
            if (active) {
                uint32_t cnt = 0;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bool eq = pid[s] == sel;
                    cnt += eq ? 1u : 0u;
                    pid[s] = eq ? 0x10000u : pid[s];
                }
                RGB c = {sky_r, sky_g, sky_b};
                if (sel != MW_SKY_PID) {
                    if (MESH) c = shade_by_draw_id(cx, sel, Xc, Yc);
                    else c = shade_prim(s_shade + sel * (MW_SHADE_REC / 4), te, Xc, Yc);
                }
                const float fc = (float)cnt;
                acc_r = fmaf(fc, c.r, acc_r);
                acc_g = fmaf(fc, c.g, acc_g);
                acc_b = fmaf(fc, c.b, acc_b);
            }
        }
        if (tp && lane == 0) {
            const unsigned long long tp2 = __builtin_readcyclecounter();
            atomicAdd(cx.tprof + 0, tp1 - tp0); atomicAdd(cx.tprof + 1, tp2 - tp1);
            atomicAdd(cx.tprof + 2, (unsigned long long)tp_it); atomicAdd(cx.tprof + 3, 1ull);
        }
    }
#ifdef MW_VALU_PROBE
    if (dbg & 32) {
        // MW_DEBUG_FLAGS bit 5, perf experiments only: 64 extra dependent-free VALU instructions per tile
        // (is the kernel bound by VALU issue or by latency?)
        float t0 = acc_r, t1 = acc_g, t2 = acc_b, t3 = Xc;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_add_f32 %2, 1.0, %2\n\tv_add_f32 %3, 1.0, %3"
                         : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
        if (t0 + t1 + t2 + t3 == 12345.678f) acc_r = t0;      // keeps the chain alive, never true in practice
    }
#endif
    const uint32_t R = to_u8(acc_r), G = to_u8(acc_g), B = to_u8(acc_b);

    // ---- pack.  Output layout (mw_set_obs_layout; the reference's wrappers.py folded into the store):
    //   0  uint8 [H][W][3]      the observation itself
    //   1  uint8 [3][W][H]      PyTorchObsWrapper: observation.transpose(2, 1, 0)   (wrappers.py:24)
    //   2  double[H][W][1]      GreyscaleWrapper: 0.30 R + 0.59 G + 0.11 B in numpy's float64 (wrappers.py:44)
    const int fmt = FMT >= 0 ? FMT : ((dbg >> 8) & 3);
    const int row = lane >> 4, col = lane & 15;
    if (fmt == 2) {
        const double g = (0.30 * (double)R + 0.59 * (double)G) + 0.11 * (double)B;
        reinterpret_cast<double *>(obs)[((size_t)env * H + py) * W + px] = g;
    } else {
        if (fmt == 0) {
            // tile rows of 16 px * 3 B = 48 B = 12 dwords; 4 rows -> 48 dword stores
            s_pack[row * 48 + col * 3 + 0] = (uint8_t)R;
            s_pack[row * 48 + col * 3 + 1] = (uint8_t)G;
            s_pack[row * 48 + col * 3 + 2] = (uint8_t)B;
        } else {
            // per channel, the 4 rows of one column are contiguous: 16 dwords per channel
            s_pack[0 * 64 + col * 4 + row] = (uint8_t)R;
            s_pack[1 * 64 + col * 4 + row] = (uint8_t)G;
            s_pack[2 * 64 + col * 4 + row] = (uint8_t)B;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < 48) {
            const uint32_t w = reinterpret_cast<const uint32_t *>(s_pack)[lane];
            if (fmt == 0) {
                // raw buffer store: the per-lane part of the address is a 32-bit offset that does not depend
                // on the tile, the tile / env part is scalar (no 64-bit VALU address arithmetic per tile)
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
        // R13 / R14: resolved depth = sample 0; get_depth_map in float32 as numpy evaluates it
        const float z = (float)z16;
        const float d = z / 65535.0f;
        const float clip = (d - 0.5f) * 2.0f;
        const float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
        depth[((size_t)env * H + py) * W + px] = (float)(-2.0 * 100.0 * 0.04) / den;
    }
}

}  // namespace
