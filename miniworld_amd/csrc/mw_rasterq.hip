// K2Q — the raster kernel of small scenes, organised by 2x2-pixel QUADS instead of 16x4 tiles.
//
// Same work as mw_raster.hip (glClear, rasterisation / depth test of display list 1 and the entity draws, GL_MODULATE
// texturing, FrameBuffer.resolve()'s blits + glReadPixels + flip, get_depth_map: miniworld.py:1064-1086, 1193-1195;
// opengl.py:339-435), same arithmetic (mw_frag.h: llvmpipe's fragment pipeline), another mapping.  The tile kernel shades
// every triangle that touches a 16x4 tile on all 64 lanes: 2.15 shading events per Hallway tile, more than half of the
// shaded lanes discarded.  But GL shades per 2x2 quad, and 79 % of the quads of a Hallway / OneRoom frame are covered
// completely by ONE triangle: they need no coverage test, no depth, no per-sample select — only that triangle's colour.
//
// One workgroup (MWQ_THREADS lanes) per env:
//   0  the env's triangle records are staged in LDS (at most MWQ_CAP_MAX; the texture's level-0 geometry is folded in)
//   A  (tile, triangle) pairs, one per lane: touch / full by the edge functions' extremes over the tile
//   B  (quad, triangle) pairs of the tiles a triangle touches without covering them: 16-bit touch / full masks
//   C  every quad collects its triangles (at most four 6-bit ids) and is filed under a class:
//        TRIV  one triangle, every sample of the four pixels inside it          -> shade, sum, store
//        P1-4  n triangles, no quad-level evidence of overlap                   -> "painter without overlap": coverage
//              as wave masks, a contested sample sends the quad to the exact list
//        EXACT overlap (a covering triangle and another one)                    -> packed keys depth16 << 16 | id
//        BIG   more than four triangles: the exact path over the tile's list    FALLBACK: over all triangles
//   D  wavefronts draw batches of 16 quads of one class (heaviest classes first); lanes 4j .. 4j+3 are quad j, every quad
//      with its OWN triangle: records come per lane from LDS, coverage masks stay wave masks in SGPRs (a v_cmp does not
//      care that the lanes' operands belong to different triangles), the lod's texture-coordinate differences are DPP
//      moves inside the quad.  Bytes go to the env's frame in LDS.
//   E  the frame leaves as 16-byte stores (or in the wrappers' layouts, mw_set_obs_layout).
// Results do not depend on the order in which atomics file triangles and quads: unclaimed-sample selection is
// disjoint, the exact path is an unsigned minimum.  Instantiated for 8 samples (the hot path) and for 4 (llvmpipe's
// GL_MAX_SAMPLES, opengl.py:229-231: the reference's own frames in tests/golden/gl_*.npz run through this very code).
#include "mw_mesh.h"

#define MWQ_THREADS 512
#define MWQ_WAVES (MWQ_THREADS / 64)
#define MWQ_CAP_MAX 48        // triangle records staged per env: 48, or 40 with a depth channel (three workgroups per CU either way)
#define MWQ_SLOTS 16          // triangles listed per tile
#define MWQ_EMPTY 63u
#ifndef MWQ_TILE_FALLBACK
#define MWQ_TILE_FALLBACK 1
#endif
#ifndef MWQ_OCC
#define MWQ_OCC 6           // wavefronts per SIMD the register allocation aims at (3 workgroups per CU)
#endif

namespace {

// votes straight from the compare's lane mask (hip's __all / __any go through an integer per lane: two more VALU instructions)
__device__ inline bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_read_exec(); }
__device__ inline bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

enum { QC_FALLBACK = 0, QC_BIG, QC_EXACT, QC_P4, QC_P3, QC_P2, QC_P1, QC_TRIV, QC_SKY, QC_NCLS, QC_NONE = 15 };

// LDS record of one triangle, in 16-byte quads:
//   [0..3]   raster record quads 0-3: A[3] B[3] C[3] id z-plane tmax[3]      (mw_records.h)
//   [4 ..]   thresholds, TQ quads per edge
//   [SH ..]  shade record quads 0-5: (w plane, tex) (s plane, texture info) (t plane, fw) (r plane, fh) (g plane, byte offset
//            of the texture's level table) (b plane)
//   [CT ..]  classification thresholds of the edge values at a tile's / a quad's first pixel (pxlo, gylo):
//            touch iff E_k > T_k for all k, covered iff E_k > F_k:  Tt[3] Ft[3] (16x4 tile)  Tq[3] Fq[3] (2x2 quad)
//              T_k = tmin_k - max(A_k, 0) (w - 1) - max(B_k, 0) (h - 1),  F_k = tmax_k - min(A_k, 0) (w - 1) - min(B_k, 0) (h - 1)
//            (the edge value is monotone along x and y: its extremes over the rectangle sit at corners)
template <int S> struct QRec {
    static constexpr int TQ = (S + 3) / 4;
    static constexpr int SH = 4 + 3 * TQ;
    static constexpr int CT = SH + 6;
    static constexpr int NQ = CT + 3;         // 19 quads for S = 8: 76 dwords, 16 consecutive records' quads fall into 16 different bank groups
};

// LDS plan of one workgroup (host and device compute the same offsets)
struct QPlan {
    int rec, frame, zbuf, qids, queue, xq, tcnt, tfull, tlist, tmask, pe, misc, btab, scratch, total;
};
__host__ __device__ inline int q_cap(bool depth) { return depth ? 40 : MWQ_CAP_MAX; }
__host__ __device__ inline QPlan q_plan(int S, int W, int H, int n_tiles, bool depth)
{
    const int NQ = S == 8 ? QRec<8>::NQ : QRec<4>::NQ;
    const int nquads = (W / 2) * (H / 2);
    QPlan p;
    int o = 0;
    auto take = [&](int bytes) { const int at = o; o += (bytes + 15) & ~15; return at; };
    p.rec = take(q_cap(depth) * NQ * 16);
    p.frame = take(W * H * 3);
    p.zbuf = depth ? take(W * H * 2) : 0;
    p.qids = take(nquads * 4);
    p.queue = take((nquads + 16 * QC_NCLS + 48) * 2);
    p.tcnt = take(n_tiles * 4);
    p.tfull = take(n_tiles * 4);                    // triangles that cover the whole tile: count | three ids << 8, 16, 24
    p.tlist = take(n_tiles * MWQ_SLOTS);
    p.tmask = take(n_tiles * MWQ_SLOTS * 4);
    p.pe = p.frame;                                 // partial (tile, triangle) events, phases A -> B: in the frame (W H 3 >= tiles x 64 bytes), which phase D writes first
    p.xq = take(nquads * 2);                        // the exact list of phase D
    p.misc = take(64 * 4);
    p.btab = take((nquads / 16 + QC_NCLS) * 4);     // per batch: first quad of the queue | class << 12 | quads << 16
    p.scratch = p.rec;                              // one record per wavefront (envs with more triangles than the records hold, 4 samples: no staged records then)
    p.total = o;
    return p;
}

struct QCtx {
    const float4 *s_rec;
    uint8_t *s_frame;
    uint16_t *s_z;
    TexEnv te;
    float sky_r, sky_g, sky_b;
    int W, H;
    bool depth;
};

// ---- classification ---------------------------------------------------------------------------------------------------------
// ... of the rectangle whose first pixel is (pxlo, gylo) against the lane's triangle: T, F = the rectangle size's thresholds
__device__ inline void classify_at(const float4 &a0, const float4 &a1, const float4 &a2, int pxlo, int gylo, const int (&T)[3], const int (&F)[3],
                                    bool &touch, bool &full)
{
    const int ea[3] = {__float_as_int(a0.x), __float_as_int(a0.y), __float_as_int(a0.z)};
    const int eb[3] = {__float_as_int(a0.w), __float_as_int(a1.x), __float_as_int(a1.y)};
    const int ec[3] = {__float_as_int(a1.z), __float_as_int(a1.w), __float_as_int(a2.x)};
    touch = true; full = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int E = __mul24(ea[k], pxlo) + __mul24(eb[k], gylo) + ec[k];
        touch &= E > T[k];
        full &= E > F[k];
    }
}
template <int S>
__device__ inline void classify_tile(const float4 *rec, int pxlo, int gylo, bool &touch, bool &full)
{
    const float4 c0 = rec[QRec<S>::CT], c1 = rec[QRec<S>::CT + 1];
    const int T[3] = {__float_as_int(c0.x), __float_as_int(c0.y), __float_as_int(c0.z)}, F[3] = {__float_as_int(c0.w), __float_as_int(c1.x), __float_as_int(c1.y)};
    classify_at(rec[0], rec[1], rec[2], pxlo, gylo, T, F, touch, full);
}
template <int S>
__device__ inline void classify_quad(const float4 *rec, int pxlo, int gylo, bool &touch, bool &full)
{
    const float4 c1 = rec[QRec<S>::CT + 1], c2 = rec[QRec<S>::CT + 2];
    const int T[3] = {__float_as_int(c1.z), __float_as_int(c1.w), __float_as_int(c2.x)}, F[3] = {__float_as_int(c2.y), __float_as_int(c2.z), __float_as_int(c2.w)};
    classify_at(rec[0], rec[1], rec[2], pxlo, gylo, T, F, touch, full);
}

// ---- coverage of the lane's pixel by the lane's triangle, as wave masks ------------------------------------------------------
template <int S>
__device__ inline void cover_lane(const float4 *rec, int px, int gy, uint64_t valid_m, uint64_t (&in_m)[S])
{
    const float4 a0 = rec[0], a1 = rec[1], a2 = rec[2], a3 = rec[3];
    const int A[3] = {__float_as_int(a0.x), __float_as_int(a0.y), __float_as_int(a0.z)};
    const int B[3] = {__float_as_int(a0.w), __float_as_int(a1.x), __float_as_int(a1.y)};
    const int C[3] = {__float_as_int(a1.z), __float_as_int(a1.w), __float_as_int(a2.x)};
    const int TM[3] = {__float_as_int(a3.y), __float_as_int(a3.z), __float_as_int(a3.w)};
#pragma unroll
    for (int s = 0; s < S; ++s) in_m[s] = valid_m;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int E = __mul24(A[k], px) + __mul24(B[k], gy) + C[k];
        if (wave_all(E > TM[k])) continue;                 // every lane's pixel strictly inside its triangle's edge k
        int thr[S];
#pragma unroll
        for (int q = 0; q < QRec<S>::TQ; ++q) {
            const float4 t = rec[4 + k * QRec<S>::TQ + q];
            thr[4 * q] = __float_as_int(t.x);
            if (4 * q + 1 < S) thr[4 * q + 1] = __float_as_int(t.y);
            if (4 * q + 2 < S) thr[4 * q + 2] = __float_as_int(t.z);
            if (4 * q + 3 < S) thr[4 * q + 3] = __float_as_int(t.w);
        }
#pragma unroll
        for (int s = 0; s < S; ++s) in_m[s] &= __ballot(E > thr[s]);
    }
}

template <int S> __device__ inline float samp_ox(int s) { return (float)mwrec::Pat<S>::x[s] * 0.0625f; }
template <int S> __device__ inline float samp_oy(int s) { return (float)mwrec::Pat<S>::y[s] * 0.0625f; }

// 16-bit depth at sample s of the pixel whose integer coordinates are (fx, fy) as floats (mwgl::z_to_unorm16)
template <int S>
__device__ inline uint32_t depth16_s(float a0, float dadx, float dady, float fx, float fy, int s)
{
    const float xs = fx + samp_ox<S>(s), ys = fy + samp_oy<S>(s);
    const float z = __builtin_amdgcn_fmed3f(fmaf(dady, ys, fmaf(dadx, xs, a0)), 0.0f, 1.0f);
    return __float_as_uint(z * (65535.0f / 65536.0f) + 128.0f) & 0xffffu;
}

// ---- fragment colour of the lane's triangle at the lane's pixel (x, y: pixel centre) ----------------------------------------
// The four lanes of a quad hold the same triangle, so the quad's corner coordinates are the neighbours' own values.

// w | w << 16 of the low byte of v: one byte permute
__device__ inline uint32_t weight_pk8(uint32_t v) { return __builtin_amdgcn_perm(0u, v, 0x0C000C00u); }

// GL_LINEAR on the level whose first record is `off`, lw x lh texels (logarithms; powers of two): R | B << 16 and G
__device__ inline void fetch_pot(const TexEnv &te, uint32_t off, uint32_t lw, uint32_t lh, float s, float t, uint32_t &rb, uint32_t &g)
{
    const int fx = (int)rintf(ldexpf(s, (int)lw + 8)) - 128, fy = (int)rintf(ldexpf(t, (int)lh + 8)) - 128;
    const uint32_t i0 = __builtin_amdgcn_ubfe((uint32_t)fx, 8u, lw), j0 = __builtin_amdgcn_ubfe((uint32_t)fy, 8u, lh);
    const uint32_t rec = ((j0 << lw) + i0 + off) << 5;
    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(te.tx, rec, 0, 0);
    const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(te.tx, rec + 16u, 0, 0);
    const uint32_t wx = weight_pk8((uint32_t)fx), wy = weight_pk8((uint32_t)fy), iy = 0x01000100u - wy;
    rb = lerp8_pk(lerp8_ad(r0.x, r0.y, wx), lerp8_ad(r1.x, r1.y, wx), wy, iy);
    g = lerp8_pk(lerp8_ad(r0.z, r0.w, wx), lerp8_ad(r1.z, r1.w, wx), wy, iy);
}

__device__ inline RGB shade_lane(const float4 *sr, const TexEnv &te, float x, float y)
{
    const float4 q0 = sr[0], qr = sr[3], qg = sr[4], qb = sr[5];
    const float wv = fmaf(q0.z, y, fmaf(q0.y, x, q0.x));
    // (one vote per compare: the votes of a conjunction would go through an integer per lane)
    const bool fast = (__builtin_amdgcn_ballot_w64(fabsf(wv) >= MW_RCP_LO) & __builtin_amdgcn_ballot_w64(fabsf(wv) <= MW_RCP_HI)) == __builtin_amdgcn_read_exec();
    const float oow = fast ? rcp_exact(wv) : 1.0f / wv;
    RGB c = {fmaf(qr.z, y, fmaf(qr.y, x, qr.x)) * oow, fmaf(qg.z, y, fmaf(qg.y, x, qg.x)) * oow, fmaf(qb.z, y, fmaf(qb.y, x, qb.x)) * oow};
    const int tex = __float_as_int(q0.w);
    if (te.flat || !wave_any(tex >= 0)) return c;
    if (tex >= 0) {
        const float4 q1 = sr[1], q2 = sr[2];
        // (with every wv inside rcp_exact's domain the quotient wv * (1 / wv) is within an ulp of 1: inside it too)
        const float qq = wv * oow;
        const float invq = fast ? rcp_exact(qq) : 1.0f / qq;
        const float s = (fmaf(q1.z, y, fmaf(q1.y, x, q1.x)) * oow) * invq, t = (fmaf(q2.z, y, fmaf(q2.y, x, q2.x)) * oow) * invq;
        const float s00 = quad_bcast<2>(s), s10 = quad_bcast<3>(s), s01 = quad_bcast<0>(s);
        const float t00 = quad_bcast<2>(t), t10 = quad_bcast<3>(t), t01 = quad_bcast<0>(t);
        const uint32_t info = __float_as_uint(q1.w);      // lw0 | lh0 << 5 | nlevels << 10 | both sizes powers of two << 15
        const int nlevels = (int)((info >> 10) & 31u);
        const float rho2 = mwgl::lod_rho2(s00, t00, s10, t10, s01, t01, q2.w, qr.w);
        int l0, w8;
        mwgl::lod_from_rho2_bits(rho2, nlevels, l0, w8);
        const int l1 = min(l0 + 1, nlevels - 1);
        uint32_t rb, ag;
        if (wave_all((info >> 15) & 1u)) {
            const uint32_t dbase = __float_as_uint(qg.w);         // byte offset of lvl[0].off in the descriptor table
            const uint32_t off0 = __builtin_amdgcn_raw_buffer_load_b32(te.td, dbase + ((uint32_t)l0 << 5), 0, 0);
            const uint32_t off1 = __builtin_amdgcn_raw_buffer_load_b32(te.td, dbase + ((uint32_t)l1 << 5), 0, 0);
            const int lw0 = (int)(info & 31u), lh0 = (int)((info >> 5) & 31u);
            fetch_pot(te, off0, (uint32_t)max(lw0 - l0, 0), (uint32_t)max(lh0 - l0, 0), s, t, rb, ag);
            if (wave_any(w8 > 0)) {
                uint32_t rb1, ag1;
                fetch_pot(te, off1, (uint32_t)max(lw0 - l1, 0), (uint32_t)max(lh0 - l1, 0), s, t, rb1, ag1);
                const uint32_t wl = weight_pk8((uint32_t)w8), il = 0x01000100u - wl;
                rb = lerp8_pk(rb, rb1, wl, il);
                ag = lerp8_pk(ag, ag1, wl, il);
            }
        } else {
            const uint32_t desc = (uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4);
            int c0[3];
            fetch_level(te, desc, l0, s, t, c0);
            if (wave_any(w8 > 0)) {
                int c1[3];
                fetch_level(te, desc, l1, s, t, c1);
#pragma unroll
                for (int k = 0; k < 3; ++k) c0[k] = mwgl::lerp8(c0[k], c1[k], w8);
            }
            rb = (uint32_t)c0[0] | ((uint32_t)c0[2] << 16); ag = (uint32_t)c0[1];
        }
        c.r = ((float)(rb & 0xFFu) * (1.0f / 255.0f)) * c.r;
        c.g = ((float)(ag & 0xFFu) * (1.0f / 255.0f)) * c.g;
        c.b = ((float)((rb >> 16) & 0xFFu) * (1.0f / 255.0f)) * c.b;
    }
    return c;
}

template <int S> struct SmpQ { float r[S], g[S], b[S]; };

// mwgl::float_to_unorm8(acc * (1 / S)) in two instructions: the scaling by a power of two is exact, so acc * (255 / S) is
// the same single rounding as (acc * (1 / S)) * 255, and v_cvt_pk_u8_f32 rounds to nearest even and saturates to 0 .. 255
// like the clamp before it would (NaN -> 0).  mw_selftest_unorm8 compares the two for all 2^32 floats.
template <int S> __device__ inline uint32_t resolve_u8(float acc)
{
    return __builtin_amdgcn_cvt_pk_u8_f32(acc * (255.0f / S), 0u, 0u);
}

// resolve in sample order (u_blitter's resolve shader), unorm8, bytes into the env's frame; depth = sample 0
template <int S>
__device__ inline void store_pixel(const QCtx &cx, const SmpQ<S> &q, uint32_t z16, int px, int py, bool on)
{
    RGB acc = {q.r[0], q.g[0], q.b[0]};
#pragma unroll
    for (int s = 1; s < S; ++s) { acc.r = acc.r + q.r[s]; acc.g = acc.g + q.g[s]; acc.b = acc.b + q.b[s]; }
    const uint32_t R = resolve_u8<S>(acc.r), G = resolve_u8<S>(acc.g), B = resolve_u8<S>(acc.b);
    if (on) {
        const int pix = py * cx.W + px;
        uint8_t *dst = cx.s_frame + pix * 3;
        dst[0] = (uint8_t)R; dst[1] = (uint8_t)G; dst[2] = (uint8_t)B;
        if (cx.depth) cx.s_z[pix] = (uint16_t)z16;
    }
}

__device__ inline bool quad_any(bool v)
{
    const int x = v ? 1 : 0;
    const int r = x | __builtin_amdgcn_mov_dpp(x, 0x00, 0xF, 0xF, true) | __builtin_amdgcn_mov_dpp(x, 0x55, 0xF, 0xF, true) |
                  __builtin_amdgcn_mov_dpp(x, 0xAA, 0xF, 0xF, true) | __builtin_amdgcn_mov_dpp(x, 0xFF, 0xF, 0xF, true);
    return r != 0;
}

// ---- batch kinds ------------------------------------------------------------------------------------------------------------
// TRIV: one triangle covers every sample of the quad
template <int S>
__device__ inline void batch_trivial(const QCtx &cx, uint32_t ids, int px, int py, bool on)
{
    const int gy = cx.H - 1 - py;
    const float fx = (float)px, fy = (float)gy;
    const float4 *rec = cx.s_rec + (ids & 63u) * QRec<S>::NQ;
    const RGB c = shade_lane(rec + QRec<S>::SH, cx.te, fx + 0.5f, fy + 0.5f);
    SmpQ<S> q;
#pragma unroll
    for (int s = 0; s < S; ++s) { q.r[s] = c.r; q.g[s] = c.g; q.b[s] = c.b; }
    uint32_t z16 = 65535u;
    if (cx.depth) {
        const float4 a2 = rec[2], a3 = rec[3];
        z16 = depth16_s<S>(a2.z, a2.w, a3.x, fx, fy, 0);
    }
    store_pixel<S>(cx, q, z16, px, py, on);
}

// TRIV with a whole quad per LANE (64 quads to a batch): what depends on the quad alone — its triangle's record, the lod, the
// two levels' geometry — is computed once for its four pixels instead of once per pixel-lane, and the lod's corner
// coordinates are the lane's own values.  Pixel i of the quad: column px0 + (i & 1), image row py0 + (i >> 1).
template <int S>
__device__ inline void batch_trivial4(const QCtx &cx, uint32_t ids, int qx, int qy, bool on)
{
    const float4 *rec = cx.s_rec + (ids & 63u) * QRec<S>::NQ;
    const float4 *sr = rec + QRec<S>::SH;
    const float4 q0 = sr[0], qr = sr[3], qg = sr[4], qb = sr[5];
    const int px0 = qx * 2, py0 = qy * 2, gy0 = cx.H - 1 - py0;
    const float fx0 = (float)px0, fy0 = (float)gy0;
    float wv[4], oow[4];
    RGB c[4];
    uint64_t dom = ~0ull;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = fx0 + (0.5f + (float)(i & 1)), y = fy0 + (0.5f - (float)(i >> 1));
        wv[i] = fmaf(q0.z, y, fmaf(q0.y, x, q0.x));
        dom &= __builtin_amdgcn_ballot_w64(fabsf(wv[i]) >= MW_RCP_LO) & __builtin_amdgcn_ballot_w64(fabsf(wv[i]) <= MW_RCP_HI);
    }
    const bool fast = dom == __builtin_amdgcn_read_exec();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x = fx0 + (0.5f + (float)(i & 1)), y = fy0 + (0.5f - (float)(i >> 1));
        oow[i] = fast ? rcp_exact(wv[i]) : 1.0f / wv[i];
        c[i].r = fmaf(qr.z, y, fmaf(qr.y, x, qr.x)) * oow[i];
        c[i].g = fmaf(qg.z, y, fmaf(qg.y, x, qg.x)) * oow[i];
        c[i].b = fmaf(qb.z, y, fmaf(qb.y, x, qb.x)) * oow[i];
    }
    const int tex = __float_as_int(q0.w);
    if (!cx.te.flat && wave_any(tex >= 0)) {
        if (tex >= 0) {
            const float4 q1 = sr[1], q2 = sr[2];
            float s[4], t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float x = fx0 + (0.5f + (float)(i & 1)), y = fy0 + (0.5f - (float)(i >> 1));
                const float qq = wv[i] * oow[i];
                const float invq = fast ? rcp_exact(qq) : 1.0f / qq;
                s[i] = (fmaf(q1.z, y, fmaf(q1.y, x, q1.x)) * oow[i]) * invq;
                t[i] = (fmaf(q2.z, y, fmaf(q2.y, x, q2.x)) * oow[i]) * invq;
            }
            const uint32_t info = __float_as_uint(q1.w);
            const int nlevels = (int)((info >> 10) & 31u);
            // GL's quad: (qx, qy) is its lower row = the image's odd row: pixel 2; (qx + 1, qy) pixel 3; (qx, qy + 1) pixel 0
            const float rho2 = mwgl::lod_rho2(s[2], t[2], s[3], t[3], s[0], t[0], q2.w, qr.w);
            int l0, w8;
            mwgl::lod_from_rho2_bits(rho2, nlevels, l0, w8);
            const int l1 = min(l0 + 1, nlevels - 1);
            const bool two = wave_any(w8 > 0);
            if (wave_all(((info >> 15) & 1u) != 0u)) {
                const uint32_t dbase = __float_as_uint(qg.w);
                const uint32_t off0 = __builtin_amdgcn_raw_buffer_load_b32(cx.te.td, dbase + ((uint32_t)l0 << 5), 0, 0);
                const uint32_t off1 = __builtin_amdgcn_raw_buffer_load_b32(cx.te.td, dbase + ((uint32_t)l1 << 5), 0, 0);
                const int lw0 = (int)(info & 31u), lh0 = (int)((info >> 5) & 31u);
                const uint32_t lwa = (uint32_t)max(lw0 - l0, 0), lha = (uint32_t)max(lh0 - l0, 0), lwb = (uint32_t)max(lw0 - l1, 0), lhb = (uint32_t)max(lh0 - l1, 0);
                const uint32_t wl = weight_pk8((uint32_t)w8), il = 0x01000100u - wl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    uint32_t rb, ag;
                    fetch_pot(cx.te, off0, lwa, lha, s[i], t[i], rb, ag);
                    if (two) {
                        uint32_t rb1, ag1;
                        fetch_pot(cx.te, off1, lwb, lhb, s[i], t[i], rb1, ag1);
                        rb = lerp8_pk(rb, rb1, wl, il);
                        ag = lerp8_pk(ag, ag1, wl, il);
                    }
                    c[i].r = ((float)(rb & 0xFFu) * (1.0f / 255.0f)) * c[i].r;
                    c[i].g = ((float)(ag & 0xFFu) * (1.0f / 255.0f)) * c[i].g;
                    c[i].b = ((float)((rb >> 16) & 0xFFu) * (1.0f / 255.0f)) * c[i].b;
                }
            } else {
                const uint32_t desc = (uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int c0[3];
                    fetch_level(cx.te, desc, l0, s[i], t[i], c0);
                    if (two) {
                        int c1[3];
                        fetch_level(cx.te, desc, l1, s[i], t[i], c1);
#pragma unroll
                        for (int k = 0; k < 3; ++k) c0[k] = mwgl::lerp8(c0[k], c1[k], w8);
                    }
                    c[i].r = ((float)c0[0] * (1.0f / 255.0f)) * c[i].r;
                    c[i].g = ((float)c0[1] * (1.0f / 255.0f)) * c[i].g;
                    c[i].b = ((float)c0[2] * (1.0f / 255.0f)) * c[i].b;
                }
            }
        }
    }
    // resolve: the eight samples of a pixel hold the same colour, summed in sample order like any others; bytes of a row's
    // two pixels leave as three halfwords
    uint32_t u[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        RGB acc = c[i];
#pragma unroll
        for (int k = 1; k < S; ++k) { acc.r = acc.r + c[i].r; acc.g = acc.g + c[i].g; acc.b = acc.b + c[i].b; }
        u[i][0] = __float_as_uint(acc.r * (255.0f / S)); u[i][1] = __float_as_uint(acc.g * (255.0f / S)); u[i][2] = __float_as_uint(acc.b * (255.0f / S));
    }
    if (on) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int pix = (py0 + r) * cx.W + px0;
            auto pk = [](uint32_t lo, uint32_t hi) { return __builtin_amdgcn_cvt_pk_u8_f32(__uint_as_float(hi), 1u, __builtin_amdgcn_cvt_pk_u8_f32(__uint_as_float(lo), 0u, 0u)); };
            uint16_t *dst = reinterpret_cast<uint16_t *>(cx.s_frame + pix * 3);
            dst[0] = (uint16_t)pk(u[2 * r][0], u[2 * r][1]);
            dst[1] = (uint16_t)pk(u[2 * r][2], u[2 * r + 1][0]);
            dst[2] = (uint16_t)pk(u[2 * r + 1][1], u[2 * r + 1][2]);
            if (cx.depth) {
                const float4 a2 = rec[2], a3 = rec[3];
                const uint32_t z0 = depth16_s<S>(a2.z, a2.w, a3.x, fx0, fy0 - (float)r, 0), z1 = depth16_s<S>(a2.z, a2.w, a3.x, fx0 + 1.0f, fy0 - (float)r, 0);
                *reinterpret_cast<uint32_t *>(cx.s_z + pix) = z0 | (z1 << 16);
            }
        }
    }
}

// SKY: nothing touches the quad — the clear colour's bytes (the same for every such pixel of the env: resolved once per
// wavefront, kept in scalar registers)
template <int S>
__device__ inline uint32_t sky_bytes(const QCtx &cx)
{
    RGB acc = {cx.sky_r, cx.sky_g, cx.sky_b};
#pragma unroll
    for (int s = 1; s < S; ++s) { acc.r = acc.r + cx.sky_r; acc.g = acc.g + cx.sky_g; acc.b = acc.b + cx.sky_b; }
    const uint32_t v = resolve_u8<S>(acc.r) | (resolve_u8<S>(acc.g) << 8) | (resolve_u8<S>(acc.b) << 16);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ inline void batch_sky(const QCtx &cx, uint32_t bytes, int px, int py, bool on)
{
    if (on) {
        const int pix = py * cx.W + px;
        uint8_t *dst = cx.s_frame + pix * 3;
        dst[0] = (uint8_t)bytes; dst[1] = (uint8_t)(bytes >> 8); dst[2] = (uint8_t)(bytes >> 16);
        if (cx.depth) cx.s_z[pix] = (uint16_t)65535u;
    }
}

// P1-P4: "painter without overlap" — while no sample is claimed twice depth is irrelevant and a claimant's colour goes
// straight to the samples it covers.  A contested sample sends the quad to the exact list (returns true for its lanes).
template <int S>
__device__ inline bool batch_partial(const QCtx &cx, uint32_t ids, int n, int px, int py, bool on)
{
    const int gy = cx.H - 1 - py;
    const float fx = (float)px, fy = (float)gy;
    SmpQ<S> q;
#pragma unroll
    for (int s = 0; s < S; ++s) { q.r[s] = cx.sky_r; q.g[s] = cx.sky_g; q.b[s] = cx.sky_b; }
    uint32_t z16 = 65535u, covbits = 0u;
    bool cont = false;
    for (int k = 0; k < n; ++k) {
        const float4 *rec = cx.s_rec + ((ids >> (6 * k)) & 63u) * QRec<S>::NQ;
        uint64_t in_m[S];
        cover_lane<S>(rec, px, gy, ~0ull, in_m);
        uint64_t any_m = 0ull;
#pragma unroll
        for (int s = 0; s < S; ++s) any_m |= in_m[s];
        if (!any_m) continue;
        uint32_t bits = 0u;
#pragma unroll
        for (int s = S - 1; s >= 0; --s)
            asm("v_addc_co_u32_e64 %0, vcc, %0, %0, %1" : "+v"(bits) : "s"(in_m[s]) : "vcc");
        cont |= (bits & covbits) != 0u;
        covbits |= bits;
        const RGB c = shade_lane(rec + QRec<S>::SH, cx.te, fx + 0.5f, fy + 0.5f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            q.r[s] = sel_mask(in_m[s], c.r, q.r[s]);
            q.g[s] = sel_mask(in_m[s], c.g, q.g[s]);
            q.b[s] = sel_mask(in_m[s], c.b, q.b[s]);
        }
        if (cx.depth) {
            const float4 a2 = rec[2], a3 = rec[3];
            z16 = sel_mask(in_m[0], depth16_s<S>(a2.z, a2.w, a3.x, fx, fy, 0), z16);
        }
    }
    const bool qc = wave_any(cont) ? quad_any(cont) : false;
    store_pixel<S>(cx, q, z16, px, py, on && !qc);
    return qc;
}

// EXACT / BIG / FALLBACK: packed keys depth16 << 16 | triangle, unsigned min = GL_LESS with first drawn wins (the list
// index of a triangle is its place in the drawing order); then every triangle that owns a sample of the quad is shaded
// once (GL multisampling shades a pixel once per triangle) for the samples it owns.
//   tri(k, p): the lane's k-th candidate (false: none); recp(p): its record
template <int S, class Tri, class RecOf>
__device__ inline void batch_exact(const QCtx &cx, int kmax, Tri tri, RecOf recp, int px, int py, bool on)
{
    const int gy = cx.H - 1 - py;
    const float fx = (float)px, fy = (float)gy;
    uint32_t key[S];
#pragma unroll
    for (int s = 0; s < S; ++s) key[s] = 0xFFFFFFFFu;
    for (int k = 0; k < kmax; ++k) {
        int p = 0;
        const bool valid = tri(k, p);
        const uint64_t vm = __ballot(valid);
        if (!vm) continue;
        const float4 *rec = recp(valid ? p : 0);
        uint64_t in_m[S];
        cover_lane<S>(rec, px, gy, vm, in_m);
        uint64_t any_m = 0ull;
#pragma unroll
        for (int s = 0; s < S; ++s) any_m |= in_m[s];
        if (!any_m) continue;
        const float4 a2 = rec[2], a3 = rec[3];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const uint32_t kk = (depth16_s<S>(a2.z, a2.w, a3.x, fx, fy, s) << 16) | (uint32_t)p;
            key[s] = sel_mask(in_m[s], min(key[s], kk), key[s]);
        }
    }
    SmpQ<S> q;
#pragma unroll
    for (int s = 0; s < S; ++s) { q.r[s] = cx.sky_r; q.g[s] = cx.sky_g; q.b[s] = cx.sky_b; }
    uint32_t pid[S];
#pragma unroll
    for (int s = 0; s < S; ++s) pid[s] = key[s] == 0xFFFFFFFFu ? 0x10000u : (key[s] & 0xFFFFu);
    for (int k = 0; k < kmax; ++k) {
        int p = 0;
        const bool valid = tri(k, p);
        bool wins = false;
#pragma unroll
        for (int s = 0; s < S; ++s) wins |= pid[s] == (uint32_t)p;
        wins &= valid;
        if (!wave_any(wins)) continue;
        // (a quad shades a triangle where any of its pixels holds one of its samples; the other lanes' colour goes nowhere)
        const float4 *rec = recp(valid ? p : 0);
        const RGB c = shade_lane(rec + QRec<S>::SH, cx.te, fx + 0.5f, fy + 0.5f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool eq = valid && pid[s] == (uint32_t)p;
            q.r[s] = eq ? c.r : q.r[s]; q.g[s] = eq ? c.g : q.g[s]; q.b[s] = eq ? c.b : q.b[s];
        }
    }
    store_pixel<S>(cx, q, key[0] >> 16, px, py, on);
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------
template <int S>
__device__ inline void rasterq_body(
    int N, int W, int H, int max_vis, int tiles_x, int n_tiles,
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const float *__restrict__ rec_cull,
    const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr, const uint32_t *__restrict__ texels,
    uint8_t *__restrict__ obs, float *__restrict__ depth, int dbg, int texel_bytes, unsigned long long *__restrict__ prof)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef QRec<S> R;
    const int env = blockIdx.x;
    if (env >= N) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool has_depth = depth != nullptr;
    const QPlan pl = q_plan(S, W, H, n_tiles, has_depth);
    float4 *s_rec = reinterpret_cast<float4 *>(smem + pl.rec);
    uint8_t *s_frame = smem + pl.frame;
    uint16_t *s_z = reinterpret_cast<uint16_t *>(smem + pl.zbuf);
    uint32_t *s_qids = reinterpret_cast<uint32_t *>(smem + pl.qids);
    uint16_t *s_queue = reinterpret_cast<uint16_t *>(smem + pl.queue);
    uint16_t *s_xq = reinterpret_cast<uint16_t *>(smem + pl.xq);
    uint32_t *s_tcnt = reinterpret_cast<uint32_t *>(smem + pl.tcnt);
    uint32_t *s_tfull = reinterpret_cast<uint32_t *>(smem + pl.tfull);
    uint8_t *s_tlist = smem + pl.tlist;
    uint32_t *s_tmask = reinterpret_cast<uint32_t *>(smem + pl.tmask);
    uint32_t *s_pe = reinterpret_cast<uint32_t *>(smem + pl.pe);
    uint32_t *s_misc = reinterpret_cast<uint32_t *>(smem + pl.misc);     // [0..8] class counts, [16] partial events, [17] next batch, [18] exact list, [19] next exact batch
    uint16_t *s_rank = reinterpret_cast<uint16_t *>(s_frame);            // phase C only (the partial events are done with, the frame is not written before phase D)
    uint32_t *s_btab = reinterpret_cast<uint32_t *>(smem + pl.btab);
    const int QW = W / 2, QH = H / 2, nquads = QW * QH;
    // x / d for x < 2^16 as a multiply (the divisors are launch constants)
    const uint32_t m_qw = 0xFFFFFFFFu / (uint32_t)QW + 1u /* QW >= 8 */, m_tx = mw_magic16((uint32_t)tiles_x);
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    const int nvis = nvis_arr[env];
    const int part_mode = (dbg >> 4) & 3;        // 1: every tile but those a mesh entity can touch (the mesh-aware tile kernel draws those)
    const bool mesh_env = part_mode == 1 && __float_as_int(hdr[3]) != 0;
    const float *g_rr = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    const float4 *g_shade = reinterpret_cast<const float4 *>(rec_shade + (size_t)env * max_vis * MW_SHADE_REC);
    const float4 *g_cull = reinterpret_cast<const float4 *>(rec_cull + (size_t)env * max_vis * MW_CULL_REC);
    const MwTexDesc *texd = reinterpret_cast<const MwTexDesc *>(texels);

    // MW_K2Q_PROF (perf experiments only): s_memtime at the phase boundaries of every wavefront, [env][wave][8]
    auto stamp = [&](int k) { if (prof && lane == 0) prof[((size_t)env * MWQ_WAVES + wave) * 8 + k] = __builtin_amdgcn_s_memtime(); };
    stamp(0);
    QCtx cx;
    cx.s_rec = s_rec; cx.s_frame = s_frame; cx.s_z = s_z;
    cx.te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    cx.te.td = cx.te.tx; cx.te.texd = texd; cx.te.flat = dbg & 1;
    cx.sky_r = hdr[0]; cx.sky_g = hdr[1]; cx.sky_b = hdr[2];
    cx.W = W; cx.H = H; cx.depth = has_depth;

    // one record quad from the global records into the LDS layout (q: quad of the LDS record)
    auto stage_quad = [&](int p, int q, float4 *dst, bool with_tex) {
        float4 v;
        if (q < 4) v = reinterpret_cast<const float4 *>(g_rr + (size_t)p * MW_RASTER_REC)[q];
        else if (q < R::SH) { const int k = (q - 4) / R::TQ, j = (q - 4) % R::TQ; v = reinterpret_cast<const float4 *>(g_rr + (size_t)p * MW_RASTER_REC)[4 + 4 * k + j]; }
        else if (q < R::CT) {
            const int j = q - R::SH;
            v = g_shade[(size_t)p * (MW_SHADE_REC / 4) + j];
            if (with_tex && j >= 1 && j <= 4) {
                const int tex = __float_as_int(g_shade[(size_t)p * (MW_SHADE_REC / 4)].w);
                uint32_t w0 = 1u, h0 = 1u, nl = 1u;
                if (tex >= 0) { w0 = texd[tex].w; h0 = texd[tex].h; nl = texd[tex].nlevels; }
                if (j == 1) {
                    const uint32_t pot = ((w0 & (w0 - 1u)) | (h0 & (h0 - 1u))) == 0u ? 1u : 0u;
                    v.w = __uint_as_float((uint32_t)(31 - __builtin_clz(w0)) | ((uint32_t)(31 - __builtin_clz(h0)) << 5) | (nl << 10) | (pot << 15));
                } else if (j == 4) v.w = __uint_as_float(tex >= 0 ? ((uint32_t)tex * (uint32_t)(sizeof(MwTexDesc) / 4) + 4u) * 4u : 0u);
                else v.w = j == 2 ? (float)w0 : (float)h0;
            }
        } else {
            const float4 *rr = reinterpret_cast<const float4 *>(g_rr + (size_t)p * MW_RASTER_REC);
            const float4 r0 = rr[0], r1 = rr[1], r3 = rr[3], tm = g_cull[(size_t)p * (MW_CULL_REC / 4) + 3];
            const int A[3] = {__float_as_int(r0.x), __float_as_int(r0.y), __float_as_int(r0.z)}, B[3] = {__float_as_int(r0.w), __float_as_int(r1.x), __float_as_int(r1.y)};
            const int tmx[3] = {__float_as_int(r3.y), __float_as_int(r3.z), __float_as_int(r3.w)}, tmn[3] = {__float_as_int(tm.x), __float_as_int(tm.y), __float_as_int(tm.z)};
            int c[12];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ap = max(A[k], 0), an = min(A[k], 0), bp = max(B[k], 0), bn = min(B[k], 0);
                c[k] = tmn[k] - ap * (MW_TILE_W - 1) - bp * (MW_TILE_H - 1);
                c[3 + k] = tmx[k] - an * (MW_TILE_W - 1) - bn * (MW_TILE_H - 1);
                c[6 + k] = tmn[k] - ap - bp;
                c[9 + k] = tmx[k] - an - bn;
            }
            const int j = q - R::CT;
            v = make_float4(__int_as_float(j == 0 ? c[0] : (j == 1 ? c[4] : c[8])), __int_as_float(j == 0 ? c[1] : (j == 1 ? c[5] : c[9])),
                            __int_as_float(j == 0 ? c[2] : (j == 1 ? c[6] : c[10])), __int_as_float(j == 0 ? c[3] : (j == 1 ? c[7] : c[11])));
        }
        *dst = v;
    };

    // ---- envs with more triangles than the LDS records hold ---------------------------------------------------------------
    // (flag 0x80, tests: as if the records held 8 triangles — every env with more takes the paths below)
    if (nvis > ((dbg & 0x80) ? 8 : q_cap(has_depth))) {
        if constexpr (S == 8 && MWQ_TILE_FALLBACK) {
            // the tile code of mw_raster.hip, records read in place; every wavefront takes every MWQ_WAVES-th tile
            TileCtx tc;
            tc.s_shade = g_shade; tc.s_cull = g_cull; tc.shade_stride = MW_SHADE_REC / 4; tc.cull_stride = MW_CULL_REC / 4;
            tc.rr_env = g_rr; tc.s_pack = smem + wave * 192; tc.hdr = hdr; tc.ment = hdr + MW_HDR_MESH;
            tc.mesh_pos = tc.mesh_nrm = tc.mesh_rgb = tc.mesh_uv = nullptr; tc.planes = nullptr; tc.planes_xtra = nullptr; tc.clipbuf = nullptr; tc.slow_frags = nullptr; tc.slow_head = nullptr;
            tc.slow_stamp = 0u; tc.obs = obs; tc.depth = depth;
            tc.obs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(obs + (size_t)env * H * W * 3), 0, H * W * 3, MW_RSRC_WORD3);
            tc.te = cx.te; tc.sky_r = cx.sky_r; tc.sky_g = cx.sky_g; tc.sky_b = cx.sky_b;
            tc.env = env; tc.nvis = nvis; tc.W = W; tc.H = H; tc.dbg = dbg & ~0xF0; tc.lane = lane;
            tc.pre_touch = tc.pre_full = tc.pre_clip = tc.pre_edges = 0ull; tc.have_pre = 0;
            tc.order = nullptr;
            for (int tile = wave; tile < n_tiles; tile += MWQ_WAVES) {
                const int tx = tile % tiles_x, ty = tile / tiles_x;
                if (mesh_env && tile_in_mesh_rect(hdr, tx, ty)) continue;
                raster_tile_fmt<false, -1, false, 0, 0>(tc, tx, ty, nullptr);
            }
            return;
        }
        // 4 samples (not a hot path): every quad through the exact path over all triangles, a record at a time in the
        // wavefront's scratch
        float4 *scr = reinterpret_cast<float4 *>(smem + pl.scratch) + wave * R::NQ;
        cx.s_rec = scr;
        for (int b = wave; b * 16 < nquads; b += MWQ_WAVES) {
            const int Q = b * 16 + (lane >> 2);
            const bool on = Q < nquads;
            const int Qc = on ? Q : 0;
            const int qy = (int)__umulhi((uint32_t)Qc, m_qw), qx = Qc - qy * QW;
            const int px = qx * 2 + (lane & 1), py = qy * 2 + ((lane >> 1) & 1);
            auto tri = [&](int k, int &p) { p = k; return true; };
            auto recp = [&](int p) {
                const int pu = __builtin_amdgcn_readfirstlane(p);
                __builtin_amdgcn_wave_barrier();
                if (lane < R::NQ) stage_quad(pu, lane, scr + lane, true);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
                return (const float4 *)scr;
            };
            batch_exact<S>(cx, nvis, tri, recp, px, py, on);
        }
        __syncthreads();
    } else {
        // ---- 0: records -> LDS -------------------------------------------------------------------------------------------
        // (the texture's level-0 geometry follows during phase A: a dependent load that need not hold the copy up)
        // (two loops, so that the copied quads and the computed ones do not share a wavefront's instruction stream: plain copies
        // by source-index arithmetic, then the three threshold quads of every record on their own few lanes)
        for (int i = tid; i < nvis * R::CT; i += MWQ_THREADS) {
            const int p = i / R::CT, q = i - p * R::CT;
            const int kq = q - 4, k = kq / R::TQ, j = kq - k * R::TQ;                    // thresholds: edge k, quad j of it
            const float4 *rr = reinterpret_cast<const float4 *>(g_rr + (size_t)p * MW_RASTER_REC);
            const float4 *src = q < R::SH ? rr + (q < 4 ? q : 4 + 4 * k + j) : g_shade + (size_t)p * (MW_SHADE_REC / 4) + (q - R::SH);
            s_rec[p * R::NQ + q] = *src;
        }
        for (int i = tid; i < nvis * 3; i += MWQ_THREADS) { const int p = i / 3, j = i - p * 3; stage_quad(p, R::CT + j, s_rec + p * R::NQ + R::CT + j, false); }
        for (int i = tid; i < n_tiles; i += MWQ_THREADS) { s_tcnt[i] = 0u; s_tfull[i] = 0u; }
        for (int i = tid; i < nquads; i += MWQ_THREADS) s_qids[i] = 0u;
        if (tid < 64) s_misc[tid] = 0u;
        __syncthreads();
        stamp(1);
        if ((dbg >> 13) == 1) return;       // (bits 13-15: leave after phase 0 / A / B / C1 — instruction counts per phase, frames invalid)

        // ---- A: (tile, triangle) pairs -------------------------------------------------------------------------------------
        const bool force_fallback = (dbg & 8) != 0;
        int my_tex = -1;
        uint32_t my_w = 1u, my_h = 1u, my_nl = 1u;
        if (tid < nvis) {
            my_tex = __float_as_int(s_rec[tid * R::NQ + R::SH].w);
            if (my_tex >= 0) { my_w = texd[my_tex].w; my_h = texd[my_tex].h; my_nl = texd[my_tex].nlevels; }
        }
        const uint32_t m_nvis = (uint32_t)__builtin_amdgcn_readfirstlane((int)mw_magic16((uint32_t)max(nvis, 1)));
        for (int i = tid; i < n_tiles * nvis; i += MWQ_THREADS) {
            const int t = (int)mw_div16((uint32_t)i, m_nvis), p = i - t * nvis;
            const int ty = (int)mw_div16((uint32_t)t, m_tx), tx = t - ty * tiles_x;
            bool touch, full;
            classify_tile<S>(s_rec + p * R::NQ, tx * MW_TILE_W, H - MW_TILE_H - ty * MW_TILE_H, touch, full);
            if (touch) {
                const uint32_t slot = atomicAdd(&s_tcnt[t], 1u);
                if (slot < MWQ_SLOTS) {
                    const uint32_t e = (uint32_t)t * MWQ_SLOTS + slot;
                    s_tlist[e] = (uint8_t)p;
                    if (full) s_tmask[e] = 0xFFFFFFFFu;
                    else s_pe[atomicAdd(&s_misc[16], 1u)] = e | ((uint32_t)p << 16);
                }
                if (full) {
                    const uint32_t f = atomicAdd(&s_tfull[t], 1u) & 255u;
                    if (f < 3u) atomicOr(&s_tfull[t], (uint32_t)p << (8u + 8u * f));
                }
            }
        }
        if (tid < nvis) {
            float *r = reinterpret_cast<float *>(s_rec + tid * R::NQ + R::SH);
            const uint32_t pot = ((my_w & (my_w - 1u)) | (my_h & (my_h - 1u))) == 0u ? 1u : 0u;
            r[4 + 3] = __uint_as_float((uint32_t)(31 - __builtin_clz(my_w)) | ((uint32_t)(31 - __builtin_clz(my_h)) << 5) | (my_nl << 10) | (pot << 15));
            r[8 + 3] = (float)my_w;
            r[12 + 3] = (float)my_h;
            r[16 + 3] = __uint_as_float(my_tex >= 0 ? ((uint32_t)my_tex * (uint32_t)(sizeof(MwTexDesc) / 4) + 4u) * 4u : 0u);
        }
        __syncthreads();

        stamp(2);
        if ((dbg >> 13) == 2) return;
        // ---- B: (quad, triangle) pairs of the tiles a triangle crosses -----------------------------------------------------
        const int npe = __builtin_amdgcn_readfirstlane((int)s_misc[16]);
        for (int i = tid; i < ((npe * 16 + 63) & ~63); i += MWQ_THREADS) {
            const int j = i >> 4, qi = i & 15;
            bool touch = false, full = false;
            uint32_t e = 0u;
            if (j < npe) {
                const uint32_t ep = s_pe[j];
                e = ep & 0xFFFFu;
                const int t = (int)(e / MWQ_SLOTS), p = (int)(ep >> 16);
                const int ty = (int)mw_div16((uint32_t)t, m_tx), tx = t - ty * tiles_x;
                const int qx = tx * 8 + (qi & 7), qy = ty * 2 + (qi >> 3);
                classify_quad<S>(s_rec + p * R::NQ, qx * 2, H - 2 - qy * 2, touch, full);
                if (touch) {
                    // the quad's own list: a place by the counter in bits 24-30, the id into it (four 6-bit places), bit 31 = covered by some triangle
                    uint32_t *qd = s_qids + qy * QW + qx;
                    const uint32_t slot = (atomicAdd(qd, 1u << 24) >> 24) & 127u;
                    const uint32_t v = (full ? 0x80000000u : 0u) | (slot < 4u ? (uint32_t)p << (6u * slot) : 0u);
                    if (v) atomicOr(qd, v);
                }
            }
            const uint64_t tm = __ballot(touch), fm = __ballot(full);
            const int sh = lane & 48;
            if (j < npe && qi == 0) s_tmask[e] = (uint32_t)((tm >> sh) & 0xFFFFull) | ((uint32_t)((fm >> sh) & 0xFFFFull) << 16);
        }
        __syncthreads();

        stamp(3);
        if ((dbg >> 13) == 3) return;
        // ---- C1: every quad collects its triangles and takes a class --------------------------------------------------------
        for (int Q = tid; Q < nquads; Q += MWQ_THREADS) {
            const int qy = (int)__umulhi((uint32_t)Q, m_qw), qx = Q - qy * QW;
            const int tx = qx >> 3, ty = qy >> 1, t = ty * tiles_x + tx;
            const uint32_t v = s_qids[Q], tf = s_tfull[t];
            const uint32_t np = (v >> 24) & 127u, nf = tf & 255u, n = np + nf;
            const bool anyfull = (v >> 31) != 0u || nf != 0u;
            uint32_t ids = v & 0xFFFFFFu;
            if (nf != 0u && np < 4u) ids |= ((tf >> 8) & 63u) << (6u * np);
            if (wave_any(nf > 1u)) {
#pragma unroll
                for (uint32_t j = 1; j < 3u; ++j)
                    if (j < nf && np + j < 4u) ids |= ((tf >> (8u + 8u * j)) & 63u) << (6u * (np + j));
            }
            ids |= (0xFFFFFFu << (6u * min(n, 4u))) & 0xFFFFFFu;          // the places behind the list: MWQ_EMPTY
            int cls;
            if (mesh_env && tile_in_mesh_rect(hdr, tx, ty)) cls = QC_NONE;
            else if (s_tcnt[t] > MWQ_SLOTS || force_fallback) cls = QC_FALLBACK;
            else if (n == 0u) cls = QC_SKY;
            else if (n == 1u && anyfull && !(dbg & 4)) cls = QC_TRIV;
            else if (n > 4u || nf > 3u) cls = QC_BIG;
            else if ((n >= 2u && anyfull) || (dbg & 4)) cls = QC_EXACT;
            else cls = QC_P1 + 1 - (int)n;
            s_qids[Q] = ids | ((uint32_t)cls << 24);
            // place within the class: one LDS atomic per wavefront for the class nearly every quad is in, one per lane otherwise
            const uint64_t tm = __ballot(cls == QC_TRIV);
            uint32_t tbase = 0u;
            if (tm != 0ull) {
                const int leader = __ffsll((unsigned long long)tm) - 1;
                if (lane == leader) tbase = atomicAdd(&s_misc[QC_TRIV], (uint32_t)__popcll(tm));
                tbase = (uint32_t)__builtin_amdgcn_readlane((int)tbase, leader);
            }
            if (cls == QC_TRIV) s_rank[Q] = (uint16_t)(tbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(tm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tm, 0u)));
            else if (cls != QC_NONE) s_rank[Q] = (uint16_t)atomicAdd(&s_misc[cls], 1u);
        }
        __syncthreads();

        if ((dbg >> 13) == 4) return;
        // ---- C2: class queues in whole slots of 16 quads; a batch is one slot — four for TRIV, whose lanes are whole quads ----
        uint32_t cnt_c[QC_NCLS], first_b[QC_NCLS + 1], first_bat[QC_NCLS + 1];
        first_b[0] = 0u; first_bat[0] = 0u;
#pragma unroll
        for (int c = 0; c < QC_NCLS; ++c) {
            cnt_c[c] = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[c]);
            first_b[c + 1] = first_b[c] + (cnt_c[c] + 15u) / 16u;
            first_bat[c + 1] = first_bat[c] + (c == QC_TRIV ? (cnt_c[c] + 63u) / 64u : (cnt_c[c] + 15u) / 16u);
        }
        // (the class's first slot: lane c of a register holds first_b[c], a quad fetches its own with one ds_bpermute)
        uint32_t fb_lane = 0u;
#pragma unroll
        for (int c = 0; c < QC_NCLS; ++c) fb_lane = lane == c ? first_b[c] : fb_lane;
        for (int Q0 = 0; Q0 < nquads; Q0 += MWQ_THREADS) {
            const int Q = Q0 + tid;
            const uint32_t cls = Q < nquads ? s_qids[Q] >> 24 : (uint32_t)QC_NONE;
            const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(cls & 15u) << 2, (int)fb_lane);
            if (cls != QC_NONE) s_queue[base * 16u + s_rank[Q]] = (uint16_t)Q;
        }
        for (uint32_t b = tid; b < first_bat[QC_NCLS]; b += MWQ_THREADS) {
            uint32_t cls = 0u;
#pragma unroll
            for (int c = 1; c < QC_NCLS; ++c) cls += b >= first_bat[c] ? 1u : 0u;
            uint32_t cfirst = 0u, cslot = 0u, ccnt = 0u;
#pragma unroll
            for (int c = 0; c < QC_NCLS; ++c) { cfirst = cls == (uint32_t)c ? first_bat[c] : cfirst; cslot = cls == (uint32_t)c ? first_b[c] : cslot; ccnt = cls == (uint32_t)c ? cnt_c[c] : ccnt; }
            const uint32_t size = cls == QC_TRIV ? 64u : 16u, bi = b - cfirst;
            s_btab[b] = ((cslot + bi * (size / 16u)) * 16u) | (cls << 12) | (min(size, ccnt - bi * size) << 16);      // first quad of the queue | class | quads
        }
        __syncthreads();

        stamp(4);
        // ---- D: batches ------------------------------------------------------------------------------------------------------
        if (prof && tid < QC_NCLS) prof[(size_t)N * MWQ_WAVES * 8 + (size_t)env * 16 + tid] = s_misc[tid];       // (class sizes, beside the stamps)
        const uint32_t sky_u8 = sky_bytes<S>(cx);
        const uint32_t n_batches = (dbg & 0x400) ? 0u : first_bat[QC_NCLS];      // (0x400, 0x800, 0x1000: phase timing experiments, frames invalid)
        for (;;) {
            uint32_t b = 0u;
            if (lane == 0) b = atomicAdd(&s_misc[17], 1u);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b >= n_batches) break;
            const uint32_t be = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_btab[b]);
            const int cls = (int)((be >> 12) & 15u);
            if ((dbg & 0x800) && cls != QC_TRIV) continue;
            if ((dbg & 0x1000) && cls == QC_TRIV) continue;
            if (cls == QC_TRIV) {
                const bool on4 = (uint32_t)lane < (be >> 16);
                const int Q4 = on4 ? (int)s_queue[(be & 0xFFFu) + (uint32_t)lane] : 0;
                const int qy4 = (int)__umulhi((uint32_t)Q4, m_qw);
                batch_trivial4<S>(cx, on4 ? s_qids[Q4] : 0u, Q4 - qy4 * QW, qy4, on4);
                continue;
            }
            const bool on = (uint32_t)(lane >> 2) < (be >> 16);
            const int Q = on ? (int)s_queue[(be & 0xFFFu) + (uint32_t)(lane >> 2)] : 0;
            const int qy = (int)__umulhi((uint32_t)Q, m_qw), qx = Q - qy * QW;
            const int px = qx * 2 + (lane & 1), py = qy * 2 + ((lane >> 1) & 1);
            const uint32_t ids = on ? s_qids[Q] : 0u;
            if (cls == QC_SKY) batch_sky(cx, sky_u8, px, py, on);
            else if (cls >= QC_P4) {
                const bool qc = batch_partial<S>(cx, ids, QC_P1 + 1 - cls, px, py, on);
                if (qc && on && (lane & 3) == 0) s_xq[atomicAdd(&s_misc[18], 1u)] = (uint16_t)Q;
            } else if (cls == QC_EXACT) {
                auto tri = [&](int k, int &p) { p = (int)((ids >> (6 * k)) & 63u); return p != (int)MWQ_EMPTY; };
                auto recp = [&](int p) { return (const float4 *)(s_rec + p * R::NQ); };
                batch_exact<S>(cx, 4, tri, recp, px, py, on);
            } else if (cls == QC_BIG) {
                const int tx = qx >> 3, ty = qy >> 1, t = ty * tiles_x + tx, bit = ((qy & 1) << 3) | (qx & 7);
                const int cnt = on ? (int)s_tcnt[t] : 0;
                int kmax = cnt;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o));
                auto tri = [&](int k, int &p) {
                    if (k >= cnt) { p = 0; return false; }
                    p = s_tlist[t * MWQ_SLOTS + k];
                    return ((s_tmask[t * MWQ_SLOTS + k] >> bit) & 1u) != 0u;
                };
                auto recp = [&](int p) { return (const float4 *)(s_rec + p * R::NQ); };
                batch_exact<S>(cx, kmax, tri, recp, px, py, on);
            } else {
                auto tri = [&](int k, int &p) { p = k; return true; };
                auto recp = [&](int p) { return (const float4 *)(s_rec + p * R::NQ); };
                batch_exact<S>(cx, nvis, tri, recp, px, py, on);
            }
        }
        stamp(5);
        __syncthreads();
        stamp(6);

        // ---- D2: the quads whose samples turned out to be contested ---------------------------------------------------------
        const uint32_t nx = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[18]);
        for (;;) {
            uint32_t b = 0u;
            if (lane == 0) b = atomicAdd(&s_misc[19], 1u);
            b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
            if (b * 16u >= nx) break;
            const bool on = b * 16u + (uint32_t)(lane >> 2) < nx;
            const int Q = on ? (int)s_xq[b * 16u + (uint32_t)(lane >> 2)] : 0;
            const int qy = (int)__umulhi((uint32_t)Q, m_qw), qx = Q - qy * QW;
            const int px = qx * 2 + (lane & 1), py = qy * 2 + ((lane >> 1) & 1);
            const uint32_t ids = on ? s_qids[Q] : 0u;
            auto tri = [&](int k, int &p) { p = (int)((ids >> (6 * k)) & 63u); return p != (int)MWQ_EMPTY; };
            auto recp = [&](int p) { return (const float4 *)(s_rec + p * R::NQ); };
            batch_exact<S>(cx, 4, tri, recp, px, py, on);
        }
        __syncthreads();
    }

    // ---- E: the frame leaves ------------------------------------------------------------------------------------------------
    // Output layouts (mw_set_obs_layout; the reference's wrappers.py folded into the store):
    //   0  uint8 [H][W][3]      the observation itself
    //   1  uint8 [3][W][H]      PyTorchObsWrapper: observation.transpose(2, 1, 0)   (wrappers.py:24)
    //   2  double[H][W][1]      GreyscaleWrapper: 0.30 R + 0.59 G + 0.11 B in numpy's float64 (wrappers.py:44)
    const int fmt = (dbg >> 8) & 3;
    const int npix = W * H;
    auto in_mesh_tile = [&](int px, int py) { return mesh_env && tile_in_mesh_rect(hdr, px / MW_TILE_W, py / MW_TILE_H); };
    if (fmt == 0) {
        uint8_t *dst = obs + (size_t)env * npix * 3;
        if (!mesh_env && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0u) {
            const uint4 *src = reinterpret_cast<const uint4 *>(s_frame);
            for (int i = tid; i < npix * 3 / 16; i += MWQ_THREADS) reinterpret_cast<uint4 *>(dst)[i] = src[i];
        } else if ((reinterpret_cast<uintptr_t>(dst) & 3u) == 0u) {
            // (dwords never straddle a tile: a tile row is 48 bytes)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(s_frame);
            for (int i = tid; i < npix * 3 / 4; i += MWQ_THREADS) {
                const int pix = (i * 4) / 3;
                if (!in_mesh_tile(pix % W, pix / W)) reinterpret_cast<uint32_t *>(dst)[i] = src[i];
            }
        } else {
            for (int i = tid; i < npix * 3; i += MWQ_THREADS) {
                const int pix = i / 3;
                if (!in_mesh_tile(pix % W, pix / W)) dst[i] = s_frame[i];
            }
        }
    } else if (fmt == 1) {
        uint8_t *dst = obs + (size_t)env * npix * 3;
        for (int i = tid; i < npix * 3; i += MWQ_THREADS) {
            const int ch = i / npix, r = i - ch * npix, x = r / H, y = r - x * H;
            if (!in_mesh_tile(x, y)) dst[i] = s_frame[(y * W + x) * 3 + ch];
        }
    } else {
        double *dst = reinterpret_cast<double *>(obs) + (size_t)env * npix;
        for (int i = tid; i < npix; i += MWQ_THREADS) {
            if (in_mesh_tile(i % W, i / W)) continue;
            const double Rv = (double)s_frame[i * 3], Gv = (double)s_frame[i * 3 + 1], Bv = (double)s_frame[i * 3 + 2];
            dst[i] = (0.30 * Rv + 0.59 * Gv) + 0.11 * Bv;
        }
    }
    stamp(7);
    if (has_depth) {
        // get_depth_map in float32 as numpy evaluates it (opengl.py:426-431)
        float *dst = depth + (size_t)env * npix;
        for (int i = tid; i < npix; i += MWQ_THREADS) {
            if (in_mesh_tile(i % W, i / W)) continue;
            const float z = (float)s_z[i];
            const float d = z / 65535.0f;
            const float clip = (d - 0.5f) * 2.0f;
            const float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
            dst[i] = (float)(-2.0 * 100.0 * 0.04) / den;
        }
    }
}

}  // namespace

#define MWQ_ARGS \
    int N, int W, int H, int max_vis, int tiles_x, int n_tiles, \
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const float *__restrict__ rec_cull, \
    const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr, const uint32_t *__restrict__ texels, \
    uint8_t *__restrict__ obs, float *__restrict__ depth, int dbg, int texel_bytes, unsigned long long *__restrict__ prof
#define MWQ_FWD N, W, H, max_vis, tiles_x, n_tiles, rec_raster, rec_shade, rec_cull, nvis_arr, envhdr, texels, obs, depth, dbg, texel_bytes, prof

extern "C" __global__ __launch_bounds__(MWQ_THREADS, MWQ_OCC) void mw_rasterq_kernel(MWQ_ARGS) { rasterq_body<8>(MWQ_FWD); }
extern "C" __global__ __launch_bounds__(MWQ_THREADS) void mw_rasterq4_kernel(MWQ_ARGS) { rasterq_body<4>(MWQ_FWD); }

// bytes of dynamic LDS a launch needs (mw_engine.hip)
extern "C" int mw_rasterq_lds_bytes(int S, int W, int H, int n_tiles, int depth) { return q_plan(S, W, H, n_tiles, depth != 0).total; }
// the longest display list the quad path draws (longer ones: the tile code)
extern "C" int mw_rasterq_cap(int depth) { return q_cap(depth != 0); }

#ifdef MWQ_PROBE
// static ISA inspection only (tools/perf/isa_k2q.sh): the trivial batch on its own
extern "C" __global__ __launch_bounds__(MWQ_THREADS) void mwq_probe_trivial(const uint32_t *texels, int texel_bytes, int W, int H, const uint32_t *ids_in, float sky, int dbg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    QCtx cx;
    cx.s_rec = reinterpret_cast<float4 *>(smem); cx.s_frame = smem + 20000; cx.s_z = reinterpret_cast<uint16_t *>(smem + 40000);
    cx.te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    cx.te.td = cx.te.tx; cx.te.texd = nullptr; cx.te.flat = dbg & 1;
    cx.sky_r = cx.sky_g = cx.sky_b = sky; cx.W = W; cx.H = H; cx.depth = false;
    const int lane = threadIdx.x & 63;
    const uint32_t ids = ids_in[threadIdx.x >> 2];
    const int Q = (int)(ids >> 8);
    const int qy = Q / 40, qx = Q - qy * 40;
    batch_trivial<8>(cx, ids, qx * 2 + (lane & 1), qy * 2 + ((lane >> 1) & 1), true);
}
#endif
