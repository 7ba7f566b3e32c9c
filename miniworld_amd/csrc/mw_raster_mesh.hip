// The mesh entity kernel (mesh triangles -> sample keys + attribute planes of the winners), the kernel of the triangles that
// cross a frustum plane, and the generic-resolution view kernels.  See mw_mesh.h.
#include "mw_mesh.h"

// HW_REG_XCC_ID of 256 workgroups: which XCDs does this device show (mw_create)
extern "C" __global__ void mw_xcc_probe_kernel(uint32_t *out)
{
    uint32_t x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x & 15u;
}

namespace {
// a workgroup barrier that orders LDS only: the global minima and plane records in flight need no other wave's attention,
// and waiting for them (what __syncthreads' fences do) was most of this kernel's time
__device__ inline void lds_barrier()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
}  // namespace

// The mesh entity kernel: a workgroup takes a mesh entity in view through the vertex stage ONCE PER VERTEX
// — ObjMesh draws per-face vertex arrays (objmesh.py:280-292), but a ball's 15 576 face vertices are 2 600 positions, and
// transform_vertex is a function of the position and the entity's matrix alone — into a table in LDS (16 bytes per vertex:
// window x, y, z, 1 / w; a vertex outside the frustum keeps its clip mask instead).  Then a triangle per lane, in the order
// sorted by face normal (mw_upload_mesh: the 64 triangles of a wavefront face the same way, back-face culling retires whole
// waves): three 16-bit indices, three LDS reads, 32-bit setup, coverage by sample columns (mw_cover.h), atomic minima of the
// packed keys into the env's key buffer (all 0xFFFFFFFF on entry inside the entities' tile rectangles; K2 resets what it
// reads).  The first pass only finds the triangles that cover a sample — a quarter of a distant ball's front faces —; they
// are queued in LDS and consecutive lanes scatter their keys, light them and set them up (vertex attributes are read for those only): their attribute planes go to the env's plane
// cache [N][plane_cap][MW_PLANE_REC], where K2's mesh tiles shade the winners from.  A mesh with more than MW_MESH_VCAP
// distinct positions has no table and takes every triangle through the vertex stage by itself (raster_tri_obs).
// The workgroups (MW_ENT_THREADS lanes; dynamic LDS: 16 bytes x the largest uploaded vertex table — a launch of one workgroup
// per env would queue 2 048 of them for LDS, most to find no mesh in view) are persistent: each draws envs from a counter and
// goes through the env's mesh entities in view.
extern "C" __global__ __launch_bounds__(MW_ENT_THREADS, MW_ENT_OCC) void mw_mesh_entity_kernel(
    int N, int W, int H, const float *__restrict__ envhdr, const MwMeshDesc *__restrict__ meshes, const float4 *__restrict__ mesh_vpos,
    const uint2 *__restrict__ mesh_idx, const float *__restrict__ mesh_stream, const float *__restrict__ mesh_attr, uint32_t *__restrict__ keys_all,
    float *__restrict__ plane_cache, int plane_cap, int32_t *__restrict__ slow_count, uint32_t *__restrict__ slow_tris, const uint32_t *__restrict__ ent_list, int ent_list_cap, int32_t *ent_n, int32_t *ent_n_after,
    uint32_t *__restrict__ slow_envs, int n_xcc, unsigned long long *prof)
{
    // (the mesh kernels are the frame's critical path: their wavefronts issue ahead of the quad kernel's on a shared SIMD)
    __builtin_amdgcn_s_setprio(3);
    extern __shared__ __attribute__((aligned(16))) float4 s_vert[];
    __shared__ uint16_t s_queue[MW_ENT_ROUND], s_big[MW_ENT_ROUND];
    __shared__ int s_qn, s_bn, s_env;
    static_assert(MW_MESH_VCAP * 16 + sizeof(uint16_t) * 2 * MW_ENT_ROUND + 3 * sizeof(int) <= 65536, "the vertex table and the queues share a workgroup's 64 KB of LDS");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ent_n: this frame's list lengths (the geometry kernel's) and the cursor into them; ent_n_after: the next frame's, zeroed
    // here (the two swap places from frame to frame; nothing of the next frame starts before this kernel has ended)
    if (blockIdx.x == 0 && tid < MW_CNT_WORDS) ent_n_after[tid] = 0;
    if (tid == 0) { s_qn = 0; s_bn = 0; }
    // the lists of class b % n_xcc (mw_geom.hip files env e under e % n_xcc; the launch has at least n_xcc workgroups).  A fresh
    // process dispatches workgroup b to XCD b % 8 (tools/ubench/xcc_probe.hip), so an env's entities, its mesh tiles and their
    // records then meet one L2; nothing depends on it — later launches of a long-lived process start their round-robin elsewhere
    // (a version that picked the lists by HW_REG_XCC_ID left whole lists undrawn when a small launch missed an XCD).
    const uint32_t xcc = blockIdx.x % (uint32_t)n_xcc;
    const uint32_t *l_long = ent_list + (size_t)xcc * ent_list_cap, *l_short = ent_list + (size_t)(8u + xcc) * ent_list_cap;
    const int n_long = min(ent_n[MW_CNT_LONG + xcc], ent_list_cap), n_items = n_long + min(ent_n[MW_CNT_SHORT + xcc], ent_list_cap);
    for (;;) {
        lds_barrier();
        if (tid == 0) s_env = atomicAdd(ent_n + MW_CNT_CURSOR + xcc, 1);
        lds_barrier();
        const int item_i = s_env;
        if (item_i >= n_items) break;
        const uint32_t item = item_i < n_long ? l_long[item_i] : l_short[item_i - n_long];
        const int env = (int)(item & 0xFFFFFFu), j = (int)(item >> 24);
        const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
        uint32_t *keys = keys_all + (size_t)env * W * H * 8;
        mwgl::Frame f;
        frame_lite(hdr, W, H, f);
        // (MW_ENT_PROF, perf experiments: per env start / end time, entities, triangles, wavefront-sized triangles, winners, workgroup)
        unsigned long long pr_t0 = 0ull, pr_tris = 0ull, pr_big = 0ull, pr_win = 0ull, pr_ph[4] = {0ull, 0ull, 0ull, 0ull}, pr_last = 0ull;
        if (prof) pr_t0 = pr_last = __builtin_amdgcn_s_memrealtime();
#define MW_ENT_PHASE(i) if (prof) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); pr_ph[i] += now_ - pr_last; pr_last = now_; }
        {
            const MeshEnt e = load_ment(hdr + MW_HDR_MESH, j);
            const size_t slot0 = (size_t)env * plane_cap + (size_t)__float_as_int(hdr[MW_HDR_MESH + MW_HDR_MESH_STRIDE * j + 25]);
            float *cache_e = plane_cache + slot0 * MW_PLANE_REC, *xcache_e = plane_cache + (size_t)N * plane_cap * MW_PLANE_REC + slot0 * MW_PLANE_XTRA;
            const MwMeshDesc md = meshes[__float_as_int(hdr[MW_HDR_MESH + MW_HDR_MESH_STRIDE * j + 27])];
            const int nverts = (int)md.nverts;
            const float4 *attr_e = reinterpret_cast<const float4 *>(mesh_attr) + (size_t)e.first * 6;
            if (nverts == 0) {
                // no vertex table: record t of the entity's stream is the t-th triangle of the rasterisation order (48 bytes)
                for (int t = tid; t < e.ntris; t += MW_ENT_THREADS) {
                    const float4 *rec = reinterpret_cast<const float4 *>(mesh_stream) + (size_t)(e.first + t) * 3;
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                    const float pos[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
                    raster_tri_obs(f, e, __float_as_int(r2.y), pos, W, H, keys, attr_e + (size_t)t * 6, cache_e, xcache_e, j, slow_count + env, slow_tris + (size_t)env * MW_SLOW_TRIS, SlowEnvs{ent_n + MW_CNT_SLOW_ENVS, slow_envs, env});
                }
                continue;       // (the next item)
            }
            // ---- the vertex stage, a position per lane (the table's last readers are behind a barrier)
            // (all of a lane's positions requested before the first is used)
            float4 p4[MW_ENT_VPL];
#pragma unroll
            for (int i = 0; i < MW_ENT_VPL; ++i) { const int v = tid + i * MW_ENT_THREADS; p4[i] = v < nverts ? mesh_vpos[md.vfirst + v] : make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
#pragma unroll
            for (int i = 0; i < MW_ENT_VPL; ++i) {
                const int v = tid + i * MW_ENT_THREADS;
                if (v < nverts) {
                    const float p[3] = {p4[i].x, p4[i].y, p4[i].z};
                    mwgl::Vert V;
                    mwgl::transform_vertex(f, e.x, p, V);
                    s_vert[v] = V.clipmask ? make_float4(0.0f, 0.0f, -(float)V.clipmask, 0.0f) : make_float4(V.win[0], V.win[1], V.win[2], V.win[3]);
                }
            }
            lds_barrier();
            MW_ENT_PHASE(0)
            const uint2 *idx_e = mesh_idx + e.first;
            for (int base = 0; base < e.ntris; base += MW_ENT_ROUND) {
                // ---- a triangle per lane: keys; what covers a sample is queued (the round's indices are requested up front: a
                // load behind the first minima would wait for them)
                static_assert(MW_ENT_ROUND == MW_ENT_TPL * MW_ENT_THREADS, "whole triangles per lane and round");
                const int end = min(base + MW_ENT_ROUND, e.ntris);
                uint2 ixr[MW_ENT_TPL];
#pragma unroll
                for (int i = 0; i < MW_ENT_TPL; ++i) { const int t = base + tid + i * MW_ENT_THREADS; ixr[i] = t < end ? idx_e[t] : make_uint2(0u, 0u); }
#pragma unroll
                for (int i = 0; i < MW_ENT_TPL; ++i) {
                    const int t = base + tid + i * MW_ENT_THREADS;
                    if (t < end) {
                        const uint2 ix = ixr[i];
                        const float4 va = s_vert[ix.x & 0xFFFFu], vb = s_vert[ix.x >> 16], vc = s_vert[ix.y & 0xFFFFu];
                        const int r = classify_tri_table((int)(ix.y >> 16), va, vb, vc, W, H, cache_e, j, slow_count + env, slow_tris + (size_t)env * MW_SLOW_TRIS, SlowEnvs{ent_n + MW_CNT_SLOW_ENVS, slow_envs, env});
                        if (r == 1) s_queue[atomicAdd(&s_qn, 1)] = (uint16_t)(t - base);
                        else if (r == 2) s_big[atomicAdd(&s_bn, 1)] = (uint16_t)(t - base);
                    }
                }
                lds_barrier();
                // ---- the triangles of many pixels, one per wavefront at a time (a key or a medkit at arm's length: one lane would
                // loop over hundreds of pixels while its wavefront — and the kernel, which ends with its slowest workgroup — waits);
                // entry k0 + 8 l + w is wavefront w's l-th: its lanes fetch the indices of 64 of them at once
                const int nb = s_bn;
                MW_ENT_PHASE(1)
                if (prof) { pr_big += (unsigned long long)nb; pr_tris += (unsigned long long)(end - base); }
                for (int k0 = 0; k0 < nb; k0 += MW_ENT_THREADS) {
                    const int kk = k0 + lane * (MW_ENT_THREADS / 64) + wave;
                    const int tq = kk < nb ? (int)s_big[kk] : 0;
                    const uint2 iq = kk < nb ? idx_e[base + tq] : make_uint2(0u, 0u);
                    const int cnt = (min(nb - k0, MW_ENT_THREADS) - wave + (MW_ENT_THREADS / 64 - 1)) / (MW_ENT_THREADS / 64);
                    for (int l = 0; l < cnt; ++l) {
                        const uint32_t ixx = (uint32_t)__builtin_amdgcn_readlane((int)iq.x, l), ixy = (uint32_t)__builtin_amdgcn_readlane((int)iq.y, l);
                        const int tl = __builtin_amdgcn_readlane(tq, l);
                        const float4 va = s_vert[ixx & 0xFFFFu], vb = s_vert[ixx >> 16], vc = s_vert[ixy & 0xFFFFu];
                        if (scatter_tri_wave(e, (int)(ixy >> 16), va, vb, vc, W, H, keys, lane) && lane == 0) s_queue[atomicAdd(&s_qn, 1)] = (uint16_t)(tl | 0x8000);      // (keys done)
                    }
                }
                if (nb) lds_barrier();
                MW_ENT_PHASE(2)
                // ---- a queued triangle per lane: keys, lighting, attribute planes
                const int nq = s_qn;
                if (prof) pr_win += (unsigned long long)nq;
                for (int k = tid; k < nq; k += MW_ENT_THREADS) {
                    const int qe = (int)s_queue[k], tq = base + (qe & 0x7FFF);
                    const uint2 iq = idx_e[tq];
                    const float4 va = s_vert[iq.x & 0xFFFFu], vb = s_vert[iq.x >> 16], vc = s_vert[iq.y & 0xFFFFu];
                    scatter_winner(f, e, (int)(iq.y >> 16), va, vb, vc, W, H, keys, (qe & 0x8000) != 0, attr_e + (size_t)tq * 6, cache_e + (size_t)(iq.y >> 16) * MW_PLANE_REC, xcache_e + (size_t)(iq.y >> 16) * MW_PLANE_XTRA);
                }
                lds_barrier();
                if (tid == 0) { s_qn = 0; s_bn = 0; }
                lds_barrier();
                MW_ENT_PHASE(3)
            }
        }
        if (prof && tid == 0) {
            unsigned long long *p = prof + ((size_t)xcc * ent_list_cap + (size_t)item_i) * 8;
            p[0] = pr_t0; p[1] = __builtin_amdgcn_s_memrealtime(); p[2] = 1ull | (pr_tris << 8) | (pr_win << 32); p[3] = pr_ph[0]; p[4] = pr_ph[1]; p[5] = pr_ph[2] | (pr_ph[3] << 32);
            p[6] = (unsigned long long)blockIdx.x; p[7] = (unsigned long long)item | (pr_big << 32);
        }
    }
}

// The mesh triangles that cross a frustum plane (a mesh at the frame's edge; the scatter kernel lists them): clipped, every
// piece set up on its own (llvmpipe's clipper output), its keys scattered, its attribute planes left in the env's piece
// table and one entry (draw id, piece) per pixel with a covered sample chained to the pixel — K2 shades what wins from
// there — so that neither the scatter kernel nor K2 carries the clipper.  A wavefront takes eight triangles at a time, eight lanes to a triangle: a vertex per lane
// (transform, light), an edge of the clipped polygon per lane (the geometry kernel's clipper, with colours), a piece of the
// fan per lane (setup), then a (piece, pixel) pair per lane over all pieces of the eight.  Exits at once for an env without
// such triangles.
#define MW_SLOW_GROUPS 8        // triangles per wavefront and turn
#define MW_SLOW_SHARE 16        // wavefronts that share a listed env's triangles

namespace {

struct SlowPiece {       // what the pixel loop needs of one set-up piece
    int dcdx[3], dcdy[3], c[3];
    mwgl::Plane z;
    int x0, x1, y0, y1;
    uint32_t id;        // draw id << 16 | piece of the fan << 13
    uint32_t rec;       // the piece's record (attribute planes) in the env's piece table
};

// the piece's samples in pixel (px, gy): keys scattered; true if any
__device__ inline bool slow_cover(const SlowPiece &p, int px, int gy, int W, int H, uint32_t *keys)
{
    uint32_t *kp = keys + ((size_t)(H - 1 - gy) * W + px) * 8;
    bool any = false;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int fx = px * 256 + (int)mwrec::kPat[2][s][0] * 16, fy = gy * 256 + (int)mwrec::kPat[2][s][1] * 16;
        bool in = true;
#pragma unroll
        for (int k = 0; k < 3; ++k) in &= p.c[k] + __mul24(p.dcdy[k], fy) - __mul24(p.dcdx[k], fx) > 0;
        if (in) {
            const float xs = (float)px + samp_fx<8>(s), ys = (float)gy + samp_fy<8>(s);
            atomicMin(kp + s, (mwgl::z_to_unorm16(mwgl::plane_at(p.z, xs, ys)) << 16) | (p.id >> 16));
            any = true;
        }
    }
    return any;
}

// the piece covers a sample of pixel (px, gy): entry k of the env's list, chained to the pixel (K2 shades it from the
// piece's record if it wins)
// (a pixel's chain head is (frame stamp << 16) | index + 1: heads of earlier frames read as empty, nothing is cleared)
__device__ inline void slow_pixel(const SlowPiece &p, int px, int gy, int W, int H, uint32_t *keys, int32_t *frag_count,
                                  float4 *frags, uint32_t stamp, uint32_t *head, uint32_t *status)
{
    if (!slow_cover(p, px, gy, W, H, keys)) return;
    const int k = atomicAdd(frag_count, 1);
    if (k >= MW_SLOW_FRAGS) { atomicOr(status, MW_ST_VIS_OVERFLOW); return; }
    const uint32_t pix = (uint32_t)((H - 1 - gy) * W + px);
    const uint32_t old = atomicExch(head + pix, (stamp << 16) | ((uint32_t)k + 1u));
    const uint32_t next = (old >> 16) == stamp ? (old & 0xFFFFu) : 0u;
    frags[k] = make_float4(__uint_as_float(p.id | next), __uint_as_float(p.rec), 0.0f, 0.0f);
}

}  // namespace

extern "C" __global__ __launch_bounds__(64) void mw_mesh_slow_kernel(int W, int H, const float *__restrict__ envhdr, const float *__restrict__ mesh_pos,
                                                                    const float *__restrict__ mesh_nrm, const float *__restrict__ mesh_rgb,
                                                                    const float *__restrict__ mesh_uv, const uint32_t *__restrict__ texels, int texel_bytes,
                                                                    uint32_t *__restrict__ keys_all, int32_t *__restrict__ counts, int N, int parity,
                                                                    const uint32_t *__restrict__ slow_tris, float4 *__restrict__ frags_all,
                                                                    uint32_t *__restrict__ heads_all, uint32_t stamp, uint32_t *__restrict__ status,
                                                                    const uint32_t *__restrict__ slow_envs, const int32_t *__restrict__ slow_env_n)
{
    // the clipper's work lists, the pieces of one turn, where each piece's pixels start in the turn's pixel list (18 KB: the
    // grid is mostly empty workgroups, which must not queue for LDS)
    __builtin_amdgcn_s_setprio(3);
    __shared__ mwgl::Vert s_list[MW_SLOW_GROUPS][2][MWGL_MAX_CLIP_VERTS];
    __shared__ SlowPiece s_piece[64];
    __shared__ int s_pref[65];
    // counts: [2 parities][2][N] — listed triangles (the entity kernel's) and fragments of this frame's parity; the other
    // parity's are zeroed here for the next frame.
    // Work: the envs the entity kernel listed (slow_envs[0 .. *slow_env_n): those with a triangle across a frustum plane, one env in
    // five), MW_SLOW_SHARE wavefronts to an env, MW_SLOW_GROUPS triangles per wavefront and turn; wavefront w of the launch takes
    // the items w, w + grid, ...
    const int lane = threadIdx.x;
    for (int i = (int)blockIdx.x * 64 + lane; i < 2 * N; i += (int)gridDim.x * 64) counts[(size_t)(parity ^ 1) * 2 * N + i] = 0;
    int32_t *slow_count = counts + ((size_t)parity * 2 + 0) * N, *frag_count = counts + ((size_t)parity * 2 + 1) * N;
    const int n_items = min(*slow_env_n, N) * MW_SLOW_SHARE;
    for (int item = (int)blockIdx.x; item < n_items; item += (int)gridDim.x) {
    const int env = (int)slow_envs[item / MW_SLOW_SHARE], share = item % MW_SLOW_SHARE;
    const int n = slow_count[env];
    if (share * MW_SLOW_GROUPS >= n) continue;
    uint32_t *head = heads_all + (size_t)env * W * H;
    float4 *frags = frags_all + (size_t)env * MW_SLOW_STRIDE;
    float *pieces = reinterpret_cast<float *>(frags + (MW_SLOW_FRAGS + 1));
    if (n > MW_SLOW_TRIS && lane == 0 && share == 0) atomicOr(status, MW_ST_VIS_OVERFLOW);
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    uint32_t *keys = keys_all + (size_t)env * W * H * 8;
    (void)texels; (void)texel_bytes;
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    const int nn = min(n, MW_SLOW_TRIS);
    const int g = lane >> 3, e = lane & 7, g0 = lane & ~7;
    mwgl::Vert (&L)[2][MWGL_MAX_CLIP_VERTS] = s_list[g];
    for (int base = share * MW_SLOW_GROUPS; base < nn; base += MW_SLOW_SHARE * MW_SLOW_GROUPS) {
        const int i = base + g;
        const bool valid = i < nn;
        // ---- a vertex per lane (lanes 0 .. 2 of the group)
        const uint32_t it = valid ? slow_tris[(size_t)env * MW_SLOW_TRIS + i] : 0u;
        const MeshEnt me = load_ment(hdr + MW_HDR_MESH, (int)(it >> 16));
        const int tri = (int)(it & 0xFFFFu), tex = me.tex;
        const uint32_t id = (uint32_t)(me.start + tri);
        uint32_t cmv = 0u;
        if (valid && e < 3) {
            const float *ps = mesh_pos + (size_t)(me.first + tri) * MW_MESH_POS_STRIDE + 3 * e;
            const float *nrm = mesh_nrm + (size_t)(me.first + tri) * 9 + 3 * e, *rgb = mesh_rgb + (size_t)(me.first + tri) * 9 + 3 * e;
            const float *uv = mesh_uv + (size_t)(me.first + tri) * 6 + 2 * e;
            const float p[3] = {ps[0], ps[1], ps[2]}, nv3[3] = {nrm[0], nrm[1], nrm[2]}, c[3] = {rgb[0], rgb[1], rgb[2]};
            mwgl::Vert v;
            mwgl::transform_vertex(f, me.x, p, v);
            mwgl::light_vertex(f, me.x, nv3, c, v.col);
            v.st[0] = tex >= 0 ? uv[0] : 0.0f;
            v.st[1] = tex >= 0 ? uv[1] : 0.0f;
            cmv = v.clipmask;
            L[0][e] = v;
        }
        const uint32_t c0 = (uint32_t)__shfl((int)cmv, g0), c1 = (uint32_t)__shfl((int)cmv, g0 + 1), c2 = (uint32_t)__shfl((int)cmv, g0 + 2);
        uint32_t cm = valid && !(c0 & c1 & c2) ? (c0 | c1 | c2) : 0u;
        int nvx = valid && !(c0 & c1 & c2) ? 3 : 0, cur = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // ---- an edge per lane: the clipper (mw_geom.hip's, with the colours clipped like the texture coordinates).  A
        // triangle across all six planes can grow to nine vertices: clip_triangle then, on the group's first lane.
        const bool six = __popc(cm) == 6;
        if (six) {
            if (e == 0) {
                mwgl::Vert *r;
                const mwgl::Vert a = L[0][0], b = L[0][1], c = L[0][2];
                nvx = mwgl::clip_triangle<true>(f, a, b, c, L[0], L[1], &r);
                cur = r == L[1] ? 1 : 0;
            }
            nvx = __shfl(nvx, g0); cur = __shfl(cur, g0);
            cm = 0u;
        }
        while (__any(cm != 0u && nvx >= 3)) {
            const bool act = cm != 0u && nvx >= 3;
            const int plane = act ? __ffs((int)cm) - 1 : 0;
            if (act) cm &= cm - 1u;
            const mwgl::Vert *in = L[cur];
            mwgl::Vert *out = L[cur ^ 1];
            const bool mine = act && e < nvx;
            const int nxt = e + 1 < nvx ? e + 1 : 0;
            mwgl::Vert V;
            if (mine) V = in[e];
            const float dp_prev = mine ? mwgl::clip_dist(V, plane) : 0.0f;
            const float dp = __shfl(dp_prev, g0 + nxt);
            const bool bad = mine && (!(dp_prev == dp_prev) || dp_prev - dp_prev != 0.0f);
            const bool emit = mine && dp_prev >= 0.0f, cross = mine && ((dp >= 0.0f) != (dp_prev >= 0.0f));
            const uint32_t bg = (uint32_t)(__ballot(bad) >> g0) & 0xFFu;
            const uint32_t eg = (uint32_t)(__ballot(emit) >> g0) & 0xFFu, xg = (uint32_t)(__ballot(cross) >> g0) & 0xFFu;
            const uint32_t lowm = (1u << e) - 1u;
            const int pos = __popc(eg & lowm) + __popc(xg & lowm);
            if (emit) out[pos] = V;
            if (cross) {
                const mwgl::Vert Vn = in[nxt];
                const bool from_cur = fabsf(dp) < fabsf(dp_prev);
                const float t = (from_cur ? dp : dp_prev) / (from_cur ? dp - dp_prev : dp_prev - dp);
                mwgl::clip_interp<true>(f, out[pos + (emit ? 1 : 0)], t, from_cur ? Vn : V, from_cur ? V : Vn);
            }
            if (act) { nvx = bg ? 0 : __popc(eg) + __popc(xg); cur ^= 1; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
        if (nvx < 3) nvx = 0;
        // ---- a piece per lane: (r[q-1], r[q], r[0]), q = e + 2
        const mwgl::Vert *r = L[cur];
        const int q = e + 2;
        SlowPiece p;
        bool have = false;
        if (q < nvx) {
            mwgl::TriSetup ts;
            if (mwgl::setup_triangle(r[q - 1], r[q], r[0], true, tex >= 0, ts)) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { p.dcdx[k] = ts.dcdx[k]; p.dcdy[k] = ts.dcdy[k]; p.c[k] = (int)ts.c[k]; }
                p.z = ts.z;
                p.x0 = max(ts.minx >> 8, 0); p.x1 = min(ts.maxx >> 8, W - 1);
                p.y0 = max(ts.miny >> 8, 0); p.y1 = min(ts.maxy >> 8, H - 1);
                p.id = (id << 16) | ((uint32_t)(q - 2) << 13);
                p.rec = (uint32_t)(i * 7 + (q - 2));
                have = p.x0 <= p.x1 && p.y0 <= p.y1;
                if (have) store_planes(pieces + (size_t)p.rec * MW_PIECE_REC, pieces + (size_t)p.rec * MW_PIECE_REC + 16, ts, tex, 1);
            }
        }
        // ---- a (piece, pixel) pair per lane and turn over all pieces
        const int npx = have ? (p.x1 - p.x0 + 1) * (p.y1 - p.y0 + 1) : 0;
        int incl = npx;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off); if (lane >= off) incl += y; }
        s_pref[lane] = incl - npx;
        if (have) s_piece[lane] = p;
        if (lane == 63) s_pref[64] = incl;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        const int total = s_pref[64];
        for (int k = lane; k < total; k += 64) {
            int j = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) if (s_pref[j + step] <= k) j += step;      // the last piece that starts at or before k
            const SlowPiece u = s_piece[j];
            const int kk = k - s_pref[j], bw = u.x1 - u.x0 + 1;
            slow_pixel(u, u.x0 + kk % bw, u.y0 + kk / bw, W, H, keys, frag_count + env, frags, stamp, head, status);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    }
}

// ======================================================================================
// Generic-resolution path: render()/vis_fb 800x600 (miniworld.py:518, 1340-1362), the fallback sample counts of
// FrameBuffer (opengl.py:229-231: 4 on the reference's CI driver) and any other frame buffer size.  Exact packed-key
// resolution only; the mesh keys go through a global buffer; 64-bit edge values (the frame may be large).  Not the hot path.
// ======================================================================================
// (The out-of-line functions of this path take scalars only, so LLVM would mark their calls `tail` — they touch nothing of the caller's
// frame — and a function with such a call site saves every callee-saved register it uses, 178 of them around each triangle of
// raster_tri<16>: without the marks the calls' clobbers are known to the callers instead and nothing is saved.)
#define MW_NO_TAIL_MARKS __attribute__((disable_tail_calls))
template <int S>
__device__ inline void view_mesh_body(int W, int H, const float *hdr, const float *mesh_pos, uint32_t *keys, mwgl::Vert *clipbuf)
{
    const int n_mesh = __float_as_int(hdr[3]);
    const int stride = gridDim.x * blockDim.x;
    for (int j = 0; j < n_mesh; ++j) {
        const float *m = hdr + MW_HDR_MESH + MW_HDR_MESH_STRIDE * j;         // load_ment's layout: [2] ntris, [3] first
        const int ntris = __float_as_int(m[2]), first = __float_as_int(m[3]);
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < ntris; t += stride)
            raster_tri<S>(hdr, j, tri_sorted(mesh_pos, first, t), mesh_pos, W, H, keys, clipbuf);
    }
}

// grid (x, count): blockIdx.y = env first_env + y of the batch, its keys at keys + y * W * H * S
extern "C" __global__ __launch_bounds__(256) MW_NO_TAIL_MARKS void mw_view_mesh_kernel(int W, int H, int S, int first_env, const float *__restrict__ envhdr,
                                                                     const float *__restrict__ mesh_pos, uint32_t *keys)
{
    __shared__ mwgl::Vert s_clip[4][MW_CLIP_TURN * 2 * MWGL_MAX_CLIP_VERTS];       // the clipper's work lists, MW_CLIP_TURN pairs per wavefront
    mwgl::Vert *clipbuf = s_clip[threadIdx.x >> 6];
    const float *hdr = envhdr + (size_t)(first_env + (int)blockIdx.y) * MW_ENVHDR;
    keys += (size_t)blockIdx.y * W * H * S;
    if (S == 16) view_mesh_body<16>(W, H, hdr, mesh_pos, keys, clipbuf);
    else if (S == 4) view_mesh_body<4>(W, H, hdr, mesh_pos, keys, clipbuf);
    else if (S == 1) view_mesh_body<1>(W, H, hdr, mesh_pos, keys, clipbuf);
    else view_mesh_body<8>(W, H, hdr, mesh_pos, keys, clipbuf);
}

template <int S>
__device__ inline void view_tile_body(TileCtx &cx, int tiles_x, const uint32_t *mesh_keys)
{
    const int lane = cx.lane, W = cx.W, H = cx.H, nvis = cx.nvis;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * MW_TILE_W + (lane & 15), py = ty * MW_TILE_H + (lane >> 4);
    const int gy = H - 1 - py;
    uint32_t key[S];
#pragma unroll
    for (int s = 0; s < S; ++s) key[s] = mesh_keys ? mesh_keys[((size_t)py * W + px) * S + s] : 0xFFFFFFFFu;
    for (int p = 0; p < nvis; ++p) {
        const int *__restrict__ rr = reinterpret_cast<const int *>(cx.rr_env + (size_t)p * MW_RASTER_REC);
        const float4 *cr = cx.s_cull + (size_t)p * (MW_CULL_REC / 4);
        const uint32_t bb = __float_as_uint(cr[0].w);
        const int bx0 = bb & 255u, bx1 = (bb >> 8) & 255u, by0 = (bb >> 16) & 255u, by1 = bb >> 24;
        if (tx < bx0 || tx > bx1 || ty < by0 || ty > by1) continue;
        const float4 chi = cr[5];
        const int hi[3] = {__float_as_int(chi.x), __float_as_int(chi.y), __float_as_int(chi.z)};
        int64_t E[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int64_t c = ((int64_t)hi[k] << 32) | (uint32_t)rr[6 + k];
            E[k] = c + (int64_t)rr[k] * px + (int64_t)rr[3 + k] * gy;
        }
        const mwgl::Plane zp = {__int_as_float(rr[10]), __int_as_float(rr[11]), __int_as_float(rr[12])};
        const uint32_t id = (uint32_t)rr[9];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const bool in = E[0] > (int64_t)rr[16 + s] && E[1] > (int64_t)rr[32 + s] && E[2] > (int64_t)rr[48 + s];
            const float xs = (float)px + samp_fx<S>(s), ys = (float)gy + samp_fy<S>(s);
            const uint32_t k = (mwgl::z_to_unorm16(mwgl::plane_at(zp, xs, ys)) << 16) | id;
            key[s] = in ? min(key[s], k) : key[s];
        }
    }
    const uint32_t z16 = key[0] >> 16;
    // resolve: the samples' colours summed in sample order; a sample whose winner is the previous sample's reuses its colour
    const RGB sky = {cx.sky_r, cx.sky_g, cx.sky_b};
    RGB acc = {0.0f, 0.0f, 0.0f}, last = sky;
    uint32_t last_id = MW_SKY_PID;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const uint32_t w = key[s] & 0xFFFFu;
        const bool need = w != last_id && w != MW_SKY_PID;
        if (__any(need)) {
            const RGB c = shade_by_draw_id_s<S>(cx, need ? w : 0u, px, gy);
            if (need) { last = c; last_id = w; }
        }
        if (w == MW_SKY_PID) { last = sky; last_id = MW_SKY_PID; }
        if (s == 0) acc = last;
        else { acc.r = acc.r + last.r; acc.g = acc.g + last.g; acc.b = acc.b + last.b; }
    }
    const float inv = 1.0f / (float)S;
    const float v[3] = {acc.r * inv, acc.g * inv, acc.b * inv};
    uint8_t *dst = cx.obs + ((size_t)py * W + px) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[c] = (uint8_t)mwgl::float_to_unorm8(v[c]);
    if (cx.depth) {
        const float z = (float)z16;
        const float d = z / 65535.0f;
        const float clip = (d - 0.5f) * 2.0f;
        const float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
        cx.depth[(size_t)py * W + px] = (float)(-2.0 * 100.0 * 0.04) / den;
    }
}

// grid (n_tiles, count): blockIdx.y = env first_env + y of the batch; its frame at out + y * H * W * 3, its mesh keys
// at mesh_keys + y * W * H * S
extern "C" __global__ __launch_bounds__(64) MW_NO_TAIL_MARKS void mw_view_raster_kernel(
    int first_env, int W, int H, int S, int max_vis, int tiles_x, const float *__restrict__ rec_raster,
    const float *__restrict__ rec_shade, const float *__restrict__ rec_cull, const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr,
    const MwTexDesc *__restrict__ texd, const uint32_t *__restrict__ texels, const float *__restrict__ mesh_pos,
    const float *__restrict__ mesh_nrm, const float *__restrict__ mesh_rgb, const float *__restrict__ mesh_uv, const uint32_t *mesh_keys,
    uint8_t *__restrict__ out, float *__restrict__ depth, int texel_bytes)
{
    __shared__ mwgl::Vert s_clip[MW_CLIP_TURN * 2 * MWGL_MAX_CLIP_VERTS];
    const int env = first_env + (int)blockIdx.y;
    out += (size_t)blockIdx.y * H * W * 3;
    if (depth) depth += (size_t)blockIdx.y * H * W;
    if (mesh_keys) mesh_keys += (size_t)blockIdx.y * W * H * S;
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    TileCtx cx;
    cx.s_shade = reinterpret_cast<const float4 *>(rec_shade + (size_t)env * max_vis * MW_SHADE_REC);
    cx.s_cull = reinterpret_cast<const float4 *>(rec_cull + (size_t)env * max_vis * MW_CULL_REC);
    cx.rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    cx.s_pack = nullptr;
    cx.hdr = hdr; cx.ment = hdr + MW_HDR_MESH;
    cx.mesh_pos = mesh_pos; cx.mesh_nrm = mesh_nrm; cx.mesh_rgb = mesh_rgb; cx.mesh_uv = mesh_uv;
    cx.obs = out; cx.depth = depth;
    cx.obs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, H * W * 3, MW_RSRC_WORD3);
    cx.te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    cx.te.td = cx.te.tx;
    cx.te.texd = texd;
    cx.te.flat = 0;
    cx.sky_r = hdr[0]; cx.sky_g = hdr[1]; cx.sky_b = hdr[2];
    cx.env = 0; cx.nvis = nvis_arr[env]; cx.W = W; cx.H = H; cx.dbg = 0; cx.lane = threadIdx.x; cx.have_pre = 0; cx.order = nullptr;
    cx.planes = nullptr; cx.planes_xtra = nullptr; cx.clipbuf = s_clip; cx.slow_frags = nullptr; cx.slow_head = nullptr; cx.slow_stamp = 0u;
    cx.pre_touch = cx.pre_full = cx.pre_clip = cx.pre_edges = 0ull;
    if (S == 16) view_tile_body<16>(cx, tiles_x, mesh_keys);
    else if (S == 4) view_tile_body<4>(cx, tiles_x, mesh_keys);
    else if (S == 1) view_tile_body<1>(cx, tiles_x, mesh_keys);
    else view_tile_body<8>(cx, tiles_x, mesh_keys);
}

