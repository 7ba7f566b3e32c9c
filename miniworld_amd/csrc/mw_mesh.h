// Mesh entities (Ball / Key / Building / ...: entity.py:124-165, 435-452; ObjMesh.render objmesh.py:280-292) on the device:
// the per-triangle vertex stage, triangle setup and key scatter of the mesh scatter kernel (mw_raster_mesh.hip), and the
// shading of a mesh triangle that won a sample (K2's mesh tiles, the view kernels).
//
// A ball is 5 192 sub-pixel triangles, so the pixel-per-lane scheme of K2 would waste 60+ lanes per triangle.  Mesh
// triangles are rasterised one per LANE instead, scattering packed keys (depth16 << 16 | draw id) with atomic unsigned
// minima — GL_LESS / first-drawn-wins is an unsigned min, hence order independent — into the env's sample-key buffer;
// K2 then starts the tiles a mesh can touch from those keys.  Per-triangle arithmetic is mw_glmath.h's (the driver's
// vertex stage and triangle setup): the entity's MVP matrix, its object-space light and normal scale come from the
// geometry kernel's mesh table.
#pragma once
#include "mw_raster_common.h"
#include "mw_cover.h"

namespace {

struct MeshEnt {            // one entry of the env header's mesh table (mw_geom.hip)
    int slot, start, ntris, first, tex;
    mwgl::Xform x;
};

__device__ inline MeshEnt load_ment(const float *table, int j)
{
    const float *m = table + MW_HDR_MESH_STRIDE * j;
    MeshEnt e;
    e.slot = __float_as_int(m[0]); e.start = __float_as_int(m[1]); e.ntris = __float_as_int(m[2]);
    e.first = __float_as_int(m[3]); e.tex = __float_as_int(m[4]);
    e.x.nscale = m[5];
    e.x.light[0] = m[6]; e.x.light[1] = m[7]; e.x.light[2] = m[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) e.x.mvp.m[k] = m[9 + k];
    return e;
}

// what the per-vertex functions need of the frame: viewport and light colours (env header)
__device__ inline void frame_lite(const float *hdr, int W, int H, mwgl::Frame &f)
{
    f.vp_scale[0] = (float)W * 0.5f; f.vp_trans[0] = (float)W * 0.5f;
    f.vp_scale[1] = (float)H * 0.5f; f.vp_trans[1] = (float)H * 0.5f;
    f.vp_scale[2] = 0.5f; f.vp_trans[2] = 0.5f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { f.l_amb[i] = hdr[4 + i]; f.l_dif[i] = hdr[8 + i]; }
}

// the i-th triangle of the rasterisation order (sorted by face-normal direction; mw_device.h: MW_MESH_POS_STRIDE)
__device__ inline int tri_sorted(const float *mesh_pos, int first, int i)
{
    return (int)__float_as_uint(mesh_pos[(size_t)(first + i) * MW_MESH_POS_STRIDE + 9]);
}
__device__ inline int tri_sorted(const float *mesh_pos, const MeshEnt &e, int i) { return tri_sorted(mesh_pos, e.first, i); }

__device__ inline void tri_load(const float *mesh_pos, const MeshEnt &e, int tri, float (&p)[9])
{
    static_assert(MW_MESH_POS_STRIDE % 2 == 0, "8-byte aligned triangles");
    const float2 *src = reinterpret_cast<const float2 *>(mesh_pos + (size_t)(e.first + tri) * MW_MESH_POS_STRIDE);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float2 v = src[k]; p[2 * k] = v.x; p[2 * k + 1] = v.y; }
    p[8] = reinterpret_cast<const float *>(src)[8];
}

// sample s of an S-sample pixel: offset inside the pixel in pixels (the planes' coordinates)
template <int S> __device__ inline float samp_fx(int s) { return S == 1 ? 0.0f : (float)mwrec::kPat[mwrec::pat_index(S)][s][0] * 0.0625f; }
template <int S> __device__ inline float samp_fy(int s) { return S == 1 ? 0.0f : (float)mwrec::kPat[mwrec::pat_index(S)][s][1] * 0.0625f; }

// scatter one set-up triangle's keys: every sample inside gets min(key, depth16 << 16 | id)
template <int S>
__device__ inline void scatter_tri(const mwgl::TriEdges &t, int W, int H, uint32_t id, uint32_t *keys)
{
    const int off = S == 1 ? 128 : 0;
    int x0 = (t.minx + off) >> 8, x1 = (t.maxx + off) >> 8, y0 = (t.miny + off) >> 8, y1 = (t.maxy + off) >> 8;
    x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
    x1 = x1 > W - 1 ? W - 1 : x1; y1 = y1 > H - 1 ? H - 1 : y1;
    for (int gy = y0; gy <= y1; ++gy)
        for (int px = x0; px <= x1; ++px) {
            uint32_t *kp = keys + ((size_t)(H - 1 - gy) * W + px) * S;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                int sx, sy;
                mwrec::sample_offset(S, s, sx, sy);
                const int64_t fx = (int64_t)px * 256 + sx, fy = (int64_t)gy * 256 + sy;
                bool in = true;
#pragma unroll
                for (int k = 0; k < 3; ++k) in &= (t.c[k] + (int64_t)t.dcdy[k] * fy - (int64_t)t.dcdx[k] * fx) > 0;
                if (in) {
                    const float xs = (float)px + samp_fx<S>(s), ys = (float)gy + samp_fy<S>(s);
                    atomicMin(kp + s, (mwgl::z_to_unorm16(mwgl::plane_at(t.z, xs, ys)) << 16) | id);
                }
            }
        }
}

// the view kernels' clipper: lanes take turns with work lists in LDS, MW_CLIP_TURN lanes at a time
#define MW_CLIP_TURN 8
template <int K> __device__ inline uint64_t drop_lowest(uint64_t m)
{
#pragma unroll
    for (int k = 0; k < K; ++k) m &= m - 1ull;
    return m;
}

// rasterise one mesh triangle (triangle tri of entry j of the env's mesh table) into the key buffer, one lane per triangle.
// Out of line, and every argument a scalar: structures by reference live in the caller's frame, i.e. in scratch memory — the
// frame, the table entry and the positions were 352 B of it in every lane of mw_view_mesh_kernel; the callee reads them itself.
template <int S>
__device__ __attribute__((noinline)) void raster_tri(const float *hdr, int j, int tri, const float *mesh_pos, int W, int H, uint32_t *keys, mwgl::Vert *clipbuf)
{
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    const MeshEnt e = load_ment(hdr + MW_HDR_MESH, j);
    float pos[9];
    tri_load(mesh_pos, e, tri, pos);
    mwgl::Vert v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p[3] = {pos[k * 3], pos[k * 3 + 1], pos[k * 3 + 2]};
        mwgl::transform_vertex(f, e.x, p, v[k]);
    }
    const uint32_t id = (uint32_t)(e.start + tri);
    const uint32_t m = v[0].clipmask | v[1].clipmask | v[2].clipmask;
    if (v[0].clipmask & v[1].clipmask & v[2].clipmask) return;
    mwgl::TriEdges te;
    if (m == 0u) {
        if (mwgl::setup_triangle_pos(v[0].win, v[1].win, v[2].win, S > 1, te)) scatter_tri<S>(te, W, H, id, keys);
        return;
    }
    // a triangle that crosses a frustum plane (rare: a mesh at the screen's edge or the near plane): the wavefront's lanes that
    // hold one take turns, MW_CLIP_TURN at a time, with a pair of work lists each in LDS (clipbuf: MW_CLIP_TURN x 2 x
    // MWGL_MAX_CLIP_VERTS vertices) — as private arrays they were 1.1 KB of scratch in every lane of the kernel
#pragma unroll
    for (int k = 0; k < 3; ++k) { v[k].st[0] = v[k].st[1] = 0.0f; v[k].col[0] = v[k].col[1] = v[k].col[2] = 0.0f; }
    const int lane = (int)(threadIdx.x & 63u);
    for (uint64_t pend = __ballot(true); pend; pend = drop_lowest<MW_CLIP_TURN>(pend)) {
        const int rank = __popcll((unsigned long long)(pend & ((1ull << lane) - 1ull)));
        if (!((pend >> lane) & 1ull) || rank >= MW_CLIP_TURN) continue;
        mwgl::Vert *buf = clipbuf + rank * 2 * MWGL_MAX_CLIP_VERTS, *r;
        const int n = mwgl::clip_triangle<false>(f, v[0], v[1], v[2], buf, buf + MWGL_MAX_CLIP_VERTS, &r);
        for (int i = 2; i < n; ++i)
            if (mwgl::setup_triangle_pos(r[i - 1].win, r[i].win, r[0].win, S > 1, te)) scatter_tri<S>(te, W, H, id, keys);
    }
}

// ---- the obs-sized frame's fast path ----------------------------------------------------------------------------------------
// For frames up to 128 x 96 the 24.8 coordinates stay below 2^15 and every sum of the triangle setup fits 32 bits.
// A triangle inside the frustum is set up and scattered with 32-bit integers; what crosses a frustum plane is listed for
// mw_mesh_slow_kernel.  A triangle that covers a sample leaves its attribute planes in the env's plane
// cache (MW_PLANE_REC floats per mesh triangle in view, indexed like the draw ids), so that the tile phase shades a mesh
// winner with a 80-byte lookup instead of re-deriving its three vertices.

// every sample inside the set-up triangle gets min(key, depth16 << 16 | id); true if there was any (mw_cover.h: the sample
// columns that cross a small triangle's bounding box, the box's pixels otherwise)
__device__ inline bool scatter_tri_cols(const mwcov::Edges &ed, int W, int H, uint32_t id, uint32_t *keys)
{
    // (device-scope minima.  XCD-local ones — minima without the system-coherence bits, for keys that only one XCD's workgroups
    // touch — were tried in round 5: no faster, the same HBM write bytes, and only as safe as the claim "these workgroups run on
    // that XCD", which a long-lived process does not keep: tools/experiments/README.md)
    return mwcov::cover(ed, W, H, [&](int px, int gy, int s, float xs, float ys) {
        atomicMin(keys + ((size_t)(H - 1 - gy) * W + px) * 8 + s, (mwgl::z_to_unorm16(mwgl::plane_at(ed.z, xs, ys)) << 16) | id);
    });
}

__device__ inline void store_planes(float *rec, float *xtra, const mwgl::TriSetup &ts, int tex, int state)
{
    float4 *q = reinterpret_cast<float4 *>(rec);
    q[0] = make_float4(ts.w.a0, ts.w.dadx, ts.w.dady, __int_as_float(tex));
    q[1] = make_float4(ts.col[0].a0, ts.col[0].dadx, ts.col[0].dady, __int_as_float(state));
    q[2] = make_float4(ts.col[1].a0, ts.col[1].dadx, ts.col[1].dady, ts.s.a0);
    q[3] = make_float4(ts.col[2].a0, ts.col[2].dadx, ts.col[2].dady, ts.s.dadx);
    if (tex >= 0) *reinterpret_cast<float4 *>(xtra) = make_float4(ts.s.dady, ts.t.a0, ts.t.dadx, ts.t.dady);
}

// a triangle that crosses a frustum plane (rare): left to mw_mesh_slow_kernel, which clips it, scatters its keys and lists its
// fragments
struct SlowEnvs { int32_t *n; uint32_t *envs; int env; };       // the frame's list of envs with such triangles (the slow kernel's work list)

__device__ inline void list_slow_tri(int j, int tri, float *rec, int32_t *slow_count, uint32_t *slow_tris, const SlowEnvs &se)
{
    const int k = atomicAdd(slow_count, 1);
    if (k == 0) se.envs[atomicAdd(se.n, 1)] = (uint32_t)se.env;        // the env's first: the env joins the list (at most N entries)
    if (k < MW_SLOW_TRIS) slow_tris[k] = ((uint32_t)j << 16) | (uint32_t)tri;
    reinterpret_cast<float4 *>(rec)[1] = make_float4(0.0f, 0.0f, 0.0f, __int_as_float(MW_PLANE_SLOW));
}

// a triangle that covers a sample: its vertices lit (attr: normals, colours, texture coordinates of the three, 96 bytes), its
// attribute planes set up and stored in the env's plane cache.  v: the vertex stage's window coordinates (x, y, z, 1 / w), drawing
// order; the triangle is front-facing (it went through setup_edges).  mw_glmath.h::setup_triangle's plane arithmetic for a
// multisampled target — (v1, v0, v2), calc_coef on the unsnapped floats, attributes times 1 / w — without the edges it also builds:
// a plane leaves for its record as soon as it is known (the whole TriSetup alive cost this kernel its place at 80 registers).
__device__ inline void setup_winner(const mwgl::Frame &f, const MeshEnt &e, const float4 &va, const float4 &vb, const float4 &vc, const float4 *attr,
                                    float *rec, float *xtra)
{
    const float4 a0 = attr[0], a1 = attr[1], a2 = attr[2], a3 = attr[3], a4 = attr[4], a5 = attr[5];
    // front faces are set up in the order (v1, v0, v2)
    const float4 &p0 = vb, &p1 = va, &p2 = vc;
    const float fdx01 = p0.x - p1.x, fdy01 = p0.y - p1.y, fdx20 = p2.x - p0.x, fdy20 = p2.y - p0.y;
    const float ooa = 1.0f / (fdx01 * fdy20 - fdx20 * fdy01);
    const float dy20_ooa = fdy20 * ooa, dy01_ooa = fdy01 * ooa, dx20_ooa = fdx20 * ooa, dx01_ooa = fdx01 * ooa;
    const float x0c = p0.x, y0c = p0.y;
    float4 *q = reinterpret_cast<float4 *>(rec);
    mwgl::Plane pl;
#define MW_COEF(q0, q1, q2) mwgl::plane_coef(pl, q0, q1, q2, dy20_ooa, dy01_ooa, dx20_ooa, dx01_ooa, x0c, y0c)
    // vertex colours (drawing order a, b, c -> setup order b, a, c)
    float ca[3], cb[3], cc[3];
    {
        const float na[3] = {a0.x, a0.y, a0.z}, nb[3] = {a0.w, a1.x, a1.y}, nc[3] = {a1.z, a1.w, a2.x};
        const float ka[3] = {a2.y, a2.z, a2.w}, kb[3] = {a3.x, a3.y, a3.z}, kc[3] = {a3.w, a4.x, a4.y};
        mwgl::light_vertex(f, e.x, na, ka, ca);
        mwgl::light_vertex(f, e.x, nb, kb, cb);
        mwgl::light_vertex(f, e.x, nc, kc, cc);
    }
    const bool textured = e.tex >= 0;
    mwgl::Plane sp = {0.0f, 0.0f, 0.0f}, tp = {0.0f, 0.0f, 0.0f};
    if (textured) {
        // (s, t) of a, b, c: a4.zw, a5.xy, a5.zw
        MW_COEF(a5.x * p0.w, a4.z * p1.w, a5.z * p2.w); sp = pl;
        MW_COEF(a5.y * p0.w, a4.w * p1.w, a5.w * p2.w); tp = pl;
    }
    MW_COEF(p0.w, p1.w, p2.w);
    q[0] = make_float4(pl.a0, pl.dadx, pl.dady, __int_as_float(e.tex));
    MW_COEF(cb[0] * p0.w, ca[0] * p1.w, cc[0] * p2.w);
    q[1] = make_float4(pl.a0, pl.dadx, pl.dady, __int_as_float(1));
    MW_COEF(cb[1] * p0.w, ca[1] * p1.w, cc[1] * p2.w);
    q[2] = make_float4(pl.a0, pl.dadx, pl.dady, sp.a0);
    MW_COEF(cb[2] * p0.w, ca[2] * p1.w, cc[2] * p2.w);
    q[3] = make_float4(pl.a0, pl.dadx, pl.dady, sp.dadx);
    if (textured) *reinterpret_cast<float4 *>(xtra) = make_float4(sp.dady, tp.a0, tp.dadx, tp.dady);
#undef MW_COEF
}

// One mesh triangle of an obs-sized 8-sample frame from its own three positions (a mesh without a vertex table): vertex
// stage, setup, keys; a triangle that covers a sample leaves its attribute planes in the plane cache.
__device__ inline void raster_tri_obs(const mwgl::Frame &f, const MeshEnt &e, int tri, const float (&pos)[9], int W, int H, uint32_t *keys,
                                      const float4 *attr, float *cache, float *xcache, int j, int32_t *slow_count, uint32_t *slow_tris, const SlowEnvs &se)
{
    mwgl::Vert v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p[3] = {pos[k * 3], pos[k * 3 + 1], pos[k * 3 + 2]};
        mwgl::transform_vertex(f, e.x, p, v[k]);
    }
    if (v[0].clipmask & v[1].clipmask & v[2].clipmask) return;
    float *rec = cache + (size_t)tri * MW_PLANE_REC;
    if ((v[0].clipmask | v[1].clipmask | v[2].clipmask) != 0u) { list_slow_tri(j, tri, rec, slow_count, slow_tris, se); return; }
    mwcov::Edges ed;
    if (!mwcov::setup_edges(v[0].win, v[1].win, v[2].win, ed)) return;        // back-facing or empty
    if (!scatter_tri_cols(ed, W, H, (uint32_t)(e.start + tri), keys)) return;
    setup_winner(f, e, make_float4(v[0].win[0], v[0].win[1], v[0].win[2], v[0].win[3]), make_float4(v[1].win[0], v[1].win[1], v[1].win[2], v[1].win[3]),
                 make_float4(v[2].win[0], v[2].win[1], v[2].win[2], v[2].win[3]), attr, rec, xcache + (size_t)tri * MW_PLANE_XTRA);
}

// ... from the entity's vertex table in LDS (mw_mesh_entity_kernel): a vertex is (window x, y, z, 1 / w), or, outside the
// frustum, z = -(its clip mask).  The first pass only ASKS: 1 = the triangle covers a sample (a quarter of a distant ball's front
// faces do: their keys, lighting and attribute planes are the work of consecutive lanes afterwards, scatter_winner), 2 = its
// bounding box holds more than MW_ENT_BIG_PIXELS pixels — a lane is no place for a loop over hundreds of pixels, a wavefront
// takes it pixel per lane (scatter_tri_wave) —, 0 = nothing to do (culled, no sample, or listed for the slow path).
__device__ inline int classify_tri_table(int tri, const float4 &va, const float4 &vb, const float4 &vc, int W, int H,
                                         float *cache, int j, int32_t *slow_count, uint32_t *slow_tris, const SlowEnvs &se)
{
    if (va.z < 0.0f || vb.z < 0.0f || vc.z < 0.0f) {
        const uint32_t ma = va.z < 0.0f ? (uint32_t)(int)(-va.z) : 0u, mb = vb.z < 0.0f ? (uint32_t)(int)(-vb.z) : 0u,
                       mc = vc.z < 0.0f ? (uint32_t)(int)(-vc.z) : 0u;
        if (!(ma & mb & mc)) list_slow_tri(j, tri, cache + (size_t)tri * MW_PLANE_REC, slow_count, slow_tris, se);
        return 0;
    }
    const float wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
    mwcov::Edges ed;
    if (!mwcov::setup_edges_xy(wa, wb, wc, ed)) return 0;
    int x0, x1, y0, y1;
    if (!mwcov::pixel_box(ed, W, H, x0, x1, y0, y1)) return 0;
    if ((x1 - x0 + 1) * (y1 - y0 + 1) > MW_ENT_BIG_PIXELS) return 2;
    return mwcov::covers_any(ed, W, H) ? 1 : 0;
}

// a triangle that covers a sample (unclipped vertices of the table): its keys — unless a wavefront has scattered them —, its
// vertices lit, its attribute planes into the plane cache
__device__ inline void scatter_winner(const mwgl::Frame &f, const MeshEnt &e, int tri, const float4 &va, const float4 &vb, const float4 &vc, int W, int H,
                                      uint32_t *keys, bool keys_done, const float4 *attr, float *rec, float *xtra)
{
    if (!keys_done) {
        const float wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
        mwcov::Edges ed;
        if (mwcov::setup_edges(wa, wb, wc, ed)) scatter_tri_cols(ed, W, H, (uint32_t)(e.start + tri), keys);
    }
    setup_winner(f, e, va, vb, vc, attr, rec, xtra);
}

// A triangle of many pixels, the same for all lanes of the wavefront (vertices of the table, unclipped, front-facing): lane l
// takes the pixels l, l + 64, ... of its bounding box.  True (in every lane) if it covers a sample.
__device__ inline bool scatter_tri_wave(const MeshEnt &e, int tri, const float4 &va, const float4 &vb, const float4 &vc, int W, int H, uint32_t *keys, int lane)
{
    const float wa[4] = {va.x, va.y, va.z, va.w}, wb[4] = {vb.x, vb.y, vb.z, vb.w}, wc[4] = {vc.x, vc.y, vc.z, vc.w};
    mwcov::Edges ed;
    int x0, x1, y0, y1;
    if (!mwcov::setup_edges(wa, wb, wc, ed) || !mwcov::pixel_box(ed, W, H, x0, x1, y0, y1)) return false;
    int thr[3][8];
    mwcov::make_thresholds(ed, thr);
    const uint32_t id = (uint32_t)(e.start + tri);
    const int bw = x1 - x0 + 1, npix = bw * (y1 - y0 + 1);
    const int q0 = lane / bw, dq = 64 / bw, dr = 64 - dq * bw;
    int px = x0 + (lane - q0 * bw), gy = y0 + q0;
    bool any = false;
    for (int m = lane; m < npix; m += 64) {
        const uint32_t in = mwcov::pixel_mask(ed, thr, px, gy);
        mwcov::emit_samples(in, px, gy, [&](int sx, int sy, int s, float xs, float ys) {
            atomicMin(keys + ((size_t)(H - 1 - sy) * W + sx) * 8 + s, (mwgl::z_to_unorm16(mwgl::plane_at(ed.z, xs, ys)) << 16) | id);
        });
        any |= in != 0u;
        px += dr; gy += dq;
        if (px > x1) { px -= bw; ++gy; }
    }
    return __any(any) != 0;
}

// Attribute planes of mesh triangle (e, tri) for the pixel (px, gy): the triangle is taken through the vertex stage again
// (lighting per vertex: Gouraud), clipped if it has to be — then the part of the fan that covers the pixel — and set up.
// (out of line with scalar arguments, like raster_tri: the tile context and the table entry by reference were 500 B of scratch)
template <int S>
__device__ __attribute__((noinline)) RGB shade_mesh_tri(const float *hdr, int W, int H, int j, int tri, int px, int gy, const float *mesh_pos,
                                                        const float *mesh_nrm, const float *mesh_rgb, const float *mesh_uv, rsrc_t tex_rsrc,
                                                        int flat, mwgl::Vert *clipbuf)
{
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    const MeshEnt e = load_ment(hdr + MW_HDR_MESH, j);
    const int lane = (int)(threadIdx.x & 63u);
    float pos[9];
    tri_load(mesh_pos, e, tri, pos);
    const float *nrm = mesh_nrm + (size_t)(e.first + tri) * 9, *rgb = mesh_rgb + (size_t)(e.first + tri) * 9;
    const float *uv = mesh_uv + (size_t)(e.first + tri) * 6;
    mwgl::Vert v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float p[3] = {pos[k * 3], pos[k * 3 + 1], pos[k * 3 + 2]};
        mwgl::transform_vertex(f, e.x, p, v[k]);
        const float n[3] = {nrm[k * 3], nrm[k * 3 + 1], nrm[k * 3 + 2]}, c[3] = {rgb[k * 3], rgb[k * 3 + 1], rgb[k * 3 + 2]};
        mwgl::light_vertex(f, e.x, n, c, v[k].col);
        v[k].st[0] = e.tex >= 0 ? uv[k * 2] : 0.0f;
        v[k].st[1] = e.tex >= 0 ? uv[k * 2 + 1] : 0.0f;
    }
    const float eo = S > 1 ? 0.5f : 0.0f;
    mwgl::TriSetup ts;
    bool have = false;
    if ((v[0].clipmask | v[1].clipmask | v[2].clipmask) == 0u) {
        have = mwgl::setup_triangle(v[0], v[1], v[2], S > 1, e.tex >= 0, ts);
    } else for (uint64_t pend = __ballot(true); pend; pend = drop_lowest<MW_CLIP_TURN>(pend)) {
        // (the lanes whose triangle crosses a frustum plane take turns, MW_CLIP_TURN at a time, with the wavefront's work lists in
        // LDS, cx.clipbuf)
        const int rank = __popcll((unsigned long long)(pend & ((1ull << lane) - 1ull)));
        if (!((pend >> lane) & 1ull) || rank >= MW_CLIP_TURN) continue;
        mwgl::Vert *buf = clipbuf + rank * 2 * MWGL_MAX_CLIP_VERTS, *r;
        const int n = mwgl::clip_triangle<true>(f, v[0], v[1], v[2], buf, buf + MWGL_MAX_CLIP_VERTS, &r);
        // the first triangle of the fan with a sample of this pixel inside
        for (int i = 2; i < n && !have; ++i) {
            mwgl::TriSetup t2;
            if (!mwgl::setup_triangle(r[i - 1], r[i], r[0], S > 1, e.tex >= 0, t2)) continue;
            bool any = false;
            for (int s = 0; s < S; ++s) {
                int sx, sy;
                mwrec::sample_offset(S, s, sx, sy);
                const int64_t fx = (int64_t)px * 256 + sx, fy = (int64_t)gy * 256 + sy;
                bool in = true;
                for (int k = 0; k < 3; ++k) in &= (t2.c[k] + (int64_t)t2.dcdy[k] * fy - (int64_t)t2.dcdx[k] * fx) > 0;
                any |= in;
            }
            if (any) { ts = t2; have = true; }
        }
    }
    if (!have) return RGB{0.0f, 0.0f, 0.0f};
    TexEnv te;
    te.td = te.tx = tex_rsrc; te.texd = nullptr; te.flat = flat;
    return shade_planes_body(ts.w, ts.s, ts.t, ts.col[0], ts.col[1], ts.col[2], flat ? -1 : e.tex, te, px, gy, eo);
}

// the view kernels' shade_frag: out of line with the record's address as the argument (shade_planes takes its planes by reference)
__device__ __attribute__((noinline)) RGB shade_frag_view(const float4 *sr, rsrc_t tex_rsrc, int flat, int px, int gy, float eo)
{
    TexEnv te;
    te.td = te.tx = tex_rsrc; te.texd = nullptr; te.flat = flat;
    const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], qr = sr[3], qg = sr[4], qb = sr[5];
    const mwgl::Plane wp = {q0.x, q0.y, q0.z}, sp = {q1.x, q1.y, q1.z}, tp = {q2.x, q2.y, q2.z};
    const mwgl::Plane pr = {qr.x, qr.y, qr.z}, pg = {qg.x, qg.y, qg.z}, pb = {qb.x, qb.y, qb.z};
    return shade_planes_body(wp, sp, tp, pr, pg, pb, flat ? -1 : __float_as_int(q0.w), te, px, gy, eo);
}

// draw id -> fragment colour: ids inside a mesh entity's range are triangles, the others index the record list once the
// mesh triangles drawn before them are subtracted
template <int S>
__device__ inline RGB shade_by_draw_id_s(const TileCtx &cx, uint32_t id, int px, int gy)
{
    const int n_mesh = __float_as_int(cx.hdr[3]);
    int vis = (int)id;
    for (int j = 0; j < n_mesh; ++j) {
        const int start = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * j + 1]);
        const int nt = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * j + 2]);
        if ((int)id >= start + nt) {
            vis -= nt;
        } else if ((int)id >= start) {
            return shade_mesh_tri<S>(cx.hdr, cx.W, cx.H, j, (int)id - start, px, gy, cx.mesh_pos, cx.mesh_nrm, cx.mesh_rgb, cx.mesh_uv, cx.te.tx,
                                     cx.te.flat, cx.clipbuf);
        }
    }
    return shade_frag_view(cx.s_shade + vis * (MW_SHADE_REC / 4), cx.te.tx, cx.te.flat, px, gy, S > 1 ? 0.5f : 0.0f);
}

// the tile kernels (obs path): mesh triangle `id` of table entry mj from the plane cache (raster_tri_obs)
__device__ inline RGB shade_mesh_winner(const TileCtx &cx, int mj, uint32_t id, int px, int gy)
{
    const int start = __float_as_int(cx.ment[MW_HDR_MESH_STRIDE * mj + 1]);
    const size_t slot = (size_t)__float_as_int(cx.ment[MW_HDR_MESH_STRIDE * mj + 25]) + (size_t)((int)id - start);
    const float4 *q = reinterpret_cast<const float4 *>(cx.planes + slot * MW_PLANE_REC);
    const float4 *q4p = reinterpret_cast<const float4 *>(cx.planes_xtra + slot * MW_PLANE_XTRA);
    if (__float_as_int(q[1].w) == MW_PLANE_SLOW) {
        // a triangle that crosses a frustum plane: mw_mesh_slow_kernel clipped it and left, per pixel, a chain of the pieces
        // of its fan that cover a sample there, and the pieces' planes (of the entries for this id the first piece's counts)
        const uint32_t h0 = cx.slow_head[(cx.H - 1 - gy) * cx.W + px];
        uint32_t k = (h0 >> 16) == cx.slow_stamp ? (h0 & 0xFFFFu) : 0u;        // a head of an earlier frame is empty
        uint32_t best = 8u, piece = 0xFFFFFFFFu;
        for (int guard = 0; k != 0u && k <= MW_SLOW_FRAGS && guard < MW_SLOW_FRAGS; ++guard) {
            const float4 fr = cx.slow_frags[k - 1u];
            const uint32_t w = __float_as_uint(fr.x), pc = (w >> 13) & 7u;
            if ((w >> 16) == id && pc < best) { piece = __float_as_uint(fr.y); best = pc; }
            k = w & 0x1FFFu;
        }
        if (piece >= (uint32_t)MW_SLOW_PIECES) return RGB{0.0f, 0.0f, 0.0f};
        q = cx.slow_frags + (MW_SLOW_FRAGS + 1) + (size_t)piece * (MW_PIECE_REC / 4);
        q4p = q + 4;
    }
    const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
    const int tex = cx.te.flat ? -1 : __float_as_int(q0.w);
    const mwgl::Plane wp = {q0.x, q0.y, q0.z}, pr = {q1.x, q1.y, q1.z}, pg = {q2.x, q2.y, q2.z}, pb = {q3.x, q3.y, q3.z};
    if (tex < 0) {      // an untextured mesh (the common case): three planes over 1 / w
        const float x = (float)px + 0.5f, y = (float)gy + 0.5f;
        const float oow = rcp_safe(mwgl::plane_at(wp, x, y));
        return RGB{mwgl::plane_at(pr, x, y) * oow, mwgl::plane_at(pg, x, y) * oow, mwgl::plane_at(pb, x, y) * oow};
    }
    const float4 q4 = *q4p;
    const mwgl::Plane sp = {q2.w, q3.w, q4.x}, tp = {q4.y, q4.z, q4.w};
    return shade_planes(wp, sp, tp, pr, pg, pb, tex, cx.te, px, gy, 0.5f);
}

// a mesh tile's sample keys: read, and left cleared for the next frame's scatter
__device__ inline void take_mesh_keys(uint32_t *env_keys, int W, int tx, int ty, int lane, uint32_t (&mk)[8])
{
    const int px = tx * MW_TILE_W + tile_col(lane), py = ty * MW_TILE_H + tile_row(lane);
    uint4 *kp = reinterpret_cast<uint4 *>(env_keys + ((size_t)py * W + px) * 8);
    const uint4 k0 = kp[0], k1 = kp[1];
    mk[0] = k0.x; mk[1] = k0.y; mk[2] = k0.z; mk[3] = k0.w; mk[4] = k1.x; mk[5] = k1.y; mk[6] = k1.z; mk[7] = k1.w;
    const uint4 ones = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    if ((k0.x & k0.y & k0.z & k0.w) != 0xFFFFFFFFu) kp[0] = ones;
    if ((k1.x & k1.y & k1.z & k1.w) != 0xFFFFFFFFu) kp[1] = ones;
}

}  // namespace
