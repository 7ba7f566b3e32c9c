// K3 — raster kernel for envs that contain mesh entities (Ball / Key / Building / ...: entity.py:124-165, 435-452;
// ObjMesh.render objmesh.py:280-292), and the generic-resolution view kernels.  One 1024-thread workgroup per env.
//
// A ball is 5 192 sub-pixel triangles, so the pixel-per-lane scheme of K2 would waste 60+ lanes per triangle.  Here the
// whole env's sample-key buffer (80*60*8 dwords = 153 600 B) lives in the CU's 160 KiB LDS and the mesh triangles are
// rasterised one per lane, scattering packed keys (depth16 << 16 | draw id) with ds_min_u32 — GL_LESS / first-drawn-wins
// is an unsigned min, hence order independent.  After a barrier the 16 wavefronts walk the tiles exactly like K2, starting
// from the mesh keys; winners whose draw id falls into a mesh entity's range are shaded by re-deriving that triangle.
//
// Per-triangle arithmetic is mw_glmath.h's (the driver's vertex stage and triangle setup): the entity's MVP matrix, its
// object-space light and normal scale come from the geometry kernel's mesh table.
#include "mw_raster_common.h"

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
__device__ inline int tri_sorted(const float *mesh_pos, const MeshEnt &e, int i)
{
    return (int)__float_as_uint(mesh_pos[(size_t)(e.first + i) * MW_MESH_POS_STRIDE + 9]);
}

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

// rasterise one mesh triangle into the key buffer (one lane per triangle)
template <int S>
__device__ inline void raster_tri(const mwgl::Frame &f, const MeshEnt &e, int tri, const float (&pos)[9], int W, int H, uint32_t *keys)
{
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
    // a triangle that crosses a frustum plane (rare: a mesh at the screen's edge or the near plane): clipped in private memory
#pragma unroll
    for (int k = 0; k < 3; ++k) { v[k].st[0] = v[k].st[1] = 0.0f; v[k].col[0] = v[k].col[1] = v[k].col[2] = 0.0f; }
    mwgl::Vert buf0[MWGL_MAX_CLIP_VERTS], buf1[MWGL_MAX_CLIP_VERTS], *r;
    const int n = mwgl::clip_triangle<false>(f, v[0], v[1], v[2], buf0, buf1, &r);
    for (int i = 2; i < n; ++i)
        if (mwgl::setup_triangle_pos(r[i - 1].win, r[i].win, r[0].win, S > 1, te)) scatter_tri<S>(te, W, H, id, keys);
}

// Attribute planes of mesh triangle (e, tri) for the pixel (px, gy): the triangle is taken through the vertex stage again
// (lighting per vertex: Gouraud), clipped if it has to be — then the part of the fan that covers the pixel — and set up.
template <int S>
__device__ inline RGB shade_mesh_tri(const TileCtx &cx, const MeshEnt &e, int tri, int px, int gy)
{
    mwgl::Frame f;
    frame_lite(cx.hdr, cx.W, cx.H, f);
    float pos[9];
    tri_load(cx.mesh_pos, e, tri, pos);
    const float *nrm = cx.mesh_nrm + (size_t)(e.first + tri) * 9, *rgb = cx.mesh_rgb + (size_t)(e.first + tri) * 9;
    const float *uv = cx.mesh_uv + (size_t)(e.first + tri) * 6;
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
    } else {
        mwgl::Vert buf0[MWGL_MAX_CLIP_VERTS], buf1[MWGL_MAX_CLIP_VERTS], *r;
        const int n = mwgl::clip_triangle<true>(f, v[0], v[1], v[2], buf0, buf1, &r);
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
    return shade_planes(ts.w, ts.s, ts.t, ts.col[0], ts.col[1], ts.col[2], cx.te.flat ? -1 : e.tex, cx.te, px, gy, eo);
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
            const MeshEnt e = load_ment(cx.ment, j);
            return shade_mesh_tri<S>(cx, e, (int)id - start, px, gy);
        }
    }
    return shade_frag(cx.s_shade + vis * (MW_SHADE_REC / 4), cx.te, px, gy, S > 1 ? 0.5f : 0.0f);
}

__device__ inline RGB shade_by_draw_id(const TileCtx &cx, uint32_t id, int px, int gy) { return shade_by_draw_id_s<8>(cx, id, px, gy); }

}  // namespace

#define MW_MESH_ARGS \
    int N, int W, int H, int max_vis, int tiles_x, int n_tiles, \
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const float *__restrict__ rec_cull, \
    const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr, const MwTexDesc *__restrict__ texd, \
    const uint32_t *__restrict__ texels, const float *__restrict__ mesh_pos, const float *__restrict__ mesh_nrm, \
    const float *__restrict__ mesh_rgb, const float *__restrict__ mesh_uv, uint8_t *__restrict__ obs, float *__restrict__ depth, int dbg, int texel_bytes, \
    unsigned long long *__restrict__ prof, const int32_t *__restrict__ env_order
#define MW_MESH_FWD N, W, H, max_vis, tiles_x, n_tiles, rec_raster, rec_shade, rec_cull, nvis_arr, envhdr, texd, texels, mesh_pos, \
    mesh_nrm, mesh_rgb, mesh_uv, obs, depth, dbg, texel_bytes, prof, env_order

#define MW_K3_TAIL (16 * MW_K3_WAVE_LDS + 16 + MW_MAX_MESH_ENTS * MW_HDR_MESH_STRIDE * 4)    // LDS beside the keys: pack buffers, tile counter, mesh table

// FMT / HOT as in mw_raster.hip
template <int FMT, int HOT>
__device__ inline void mesh_kernel_body(MW_MESH_ARGS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem);                     // [H][W][8], image rows
    const unsigned long long t_start = prof ? __builtin_readcyclecounter() : 0ull;
    const int nkeys = W * H * 8;
    // block b draws the b-th env in order of decreasing mesh work (mw_mesh_order_kernel)
    const int env = env_order ? env_order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t *s_pack = smem + (size_t)nkeys * 4 + wave * MW_K3_WAVE_LDS;
    // co-run mode (flag 16): the envs without a mesh in view are drawn at the same time by mw_raster_big_kernel
    if ((dbg & 16) && __float_as_int(envhdr[(size_t)env * MW_ENVHDR + 3]) == 0) return;
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    float *s_ment = reinterpret_cast<float *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS + 16);
    {
        uint4 *k4 = reinterpret_cast<uint4 *>(keys);
        for (int i = tid; i < nkeys / 4; i += 1024) k4[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (tid == 0) *reinterpret_cast<int *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS) = 0;       // the tile counter of phase 2
        if (tid < MW_MAX_MESH_ENTS * MW_HDR_MESH_STRIDE) s_ment[tid] = hdr[MW_HDR_MESH + tid];
    }
    __syncthreads();

    TileCtx cx;
    cx.s_shade = reinterpret_cast<const float4 *>(rec_shade + (size_t)env * max_vis * MW_SHADE_REC);
    cx.s_cull = reinterpret_cast<const float4 *>(rec_cull + (size_t)env * max_vis * MW_CULL_REC);
    cx.rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    cx.s_pack = s_pack;
    cx.hdr = hdr;
    cx.ment = s_ment;
    cx.mesh_pos = mesh_pos; cx.mesh_nrm = mesh_nrm; cx.mesh_rgb = mesh_rgb; cx.mesh_uv = mesh_uv;
    cx.obs = obs; cx.depth = depth;
    cx.obs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(obs + (size_t)env * H * W * 3), 0, H * W * 3, MW_RSRC_WORD3);
    cx.te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    cx.te.td = cx.te.tx;
    cx.te.texd = texd;
    cx.te.flat = HOT ? 0 : (dbg & 1);
    cx.sky_r = hdr[0]; cx.sky_g = hdr[1]; cx.sky_b = hdr[2];
    cx.env = env; cx.nvis = nvis_arr[env]; cx.W = W; cx.H = H; cx.dbg = dbg; cx.lane = lane; cx.have_pre = 0; cx.order = nullptr;
    cx.pre_touch = cx.pre_full = cx.pre_clip = cx.pre_edges = 0ull;
    cx.tprof = nullptr;

    // ---- phase 1: every mesh triangle -> LDS keys, one triangle per lane -------------------
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    const int n_mesh = (!HOT && (dbg & 8)) ? 0 : __float_as_int(hdr[3]);
    for (int j = 0; j < n_mesh; ++j) {
        const MeshEnt e = load_ment(cx.ment, j);
        int t = tid;
        int tri = t < e.ntris ? tri_sorted(mesh_pos, e, t) : 0;
        int tri_n = t + 1024 < e.ntris ? tri_sorted(mesh_pos, e, t + 1024) : 0;
        float pos[9];
        tri_load(mesh_pos, e, tri, pos);
        while (t < e.ntris) {
            const int tri_nn = t + 2048 < e.ntris ? tri_sorted(mesh_pos, e, t + 2048) : 0;
            float pos_n[9];
            tri_load(mesh_pos, e, tri_n, pos_n);
            raster_tri<8>(f, e, tri, pos, W, H, keys);
            t += 1024; tri = tri_n; tri_n = tri_nn;
#pragma unroll
            for (int k = 0; k < 9; ++k) pos[k] = pos_n[k];
        }
    }
    __syncthreads();
    const unsigned long long t_mesh = prof ? __builtin_readcyclecounter() : 0ull;

    // ---- phase 2: tiles, taken by the 16 wavefronts from a shared counter --------------------------------------
    int *s_next = reinterpret_cast<int *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS);
    for (;;) {
        int tile = 0;
        if (lane == 0) tile = atomicAdd(s_next, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= n_tiles) break;
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int px = tx * MW_TILE_W + tile_col(lane), py = ty * MW_TILE_H + tile_row(lane);
        uint32_t mk[8];
        const uint4 k0 = *reinterpret_cast<const uint4 *>(keys + ((size_t)py * W + px) * 8);
        const uint4 k1 = *reinterpret_cast<const uint4 *>(keys + ((size_t)py * W + px) * 8 + 4);
        mk[0] = k0.x; mk[1] = k0.y; mk[2] = k0.z; mk[3] = k0.w; mk[4] = k1.x; mk[5] = k1.y; mk[6] = k1.z; mk[7] = k1.w;
        raster_tile_fmt<true, FMT, false, HOT, 0>(cx, tx, ty, mk);
    }
    if (prof) {     // MW_K3_PROF: per-env cycle counts (perf experiments only)
        __syncthreads();
        if (tid == 0) {
            const unsigned long long t_end = __builtin_readcyclecounter();
            prof[(size_t)env * 4 + 0] = t_mesh - t_start;
            prof[(size_t)env * 4 + 1] = t_end - t_mesh;
            prof[(size_t)env * 4 + 2] = (unsigned long long)n_mesh;
            unsigned long long nt = 0;
            for (int j = 0; j < n_mesh; ++j) nt += (unsigned long long)__float_as_int(hdr[MW_HDR_MESH + MW_HDR_MESH_STRIDE * j + 2]);
            prof[(size_t)env * 4 + 3] = nt;
        }
    }
}

extern "C" __global__ __launch_bounds__(1024) void mw_raster_mesh_kernel(MW_MESH_ARGS) { mesh_kernel_body<0, 1>(MW_MESH_FWD); }
extern "C" __global__ __launch_bounds__(1024) void mw_raster_mesh_depth_kernel(MW_MESH_ARGS) { mesh_kernel_body<0, 2>(MW_MESH_FWD); }
extern "C" __global__ __launch_bounds__(1024) void mw_raster_mesh_wrap_kernel(MW_MESH_ARGS) { mesh_kernel_body<-1, 0>(MW_MESH_FWD); }

// ======================================================================================
// Generic-resolution path: render()/vis_fb 800x600 (miniworld.py:518, 1340-1362), the fallback sample counts of
// FrameBuffer (opengl.py:229-231: 4 on the reference's CI driver) and any other frame buffer size.  Exact packed-key
// resolution only; the mesh keys go through a global buffer; 64-bit edge values (the frame may be large).  Not the hot path.
// ======================================================================================
template <int S>
__device__ inline void view_mesh_body(int W, int H, const float *hdr, const float *mesh_pos, uint32_t *keys)
{
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    const int n_mesh = __float_as_int(hdr[3]);
    const int stride = gridDim.x * blockDim.x;
    for (int j = 0; j < n_mesh; ++j) {
        const MeshEnt e = load_ment(hdr + MW_HDR_MESH, j);
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e.ntris; t += stride) {
            const int tri = tri_sorted(mesh_pos, e, t);
            float pos[9];
            tri_load(mesh_pos, e, tri, pos);
            raster_tri<S>(f, e, tri, pos, W, H, keys);
        }
    }
}

// grid (x, count): blockIdx.y = env first_env + y of the batch, its keys at keys + y * W * H * S
extern "C" __global__ __launch_bounds__(256) void mw_view_mesh_kernel(int W, int H, int S, int first_env, const float *__restrict__ envhdr,
                                                                     const float *__restrict__ mesh_pos, uint32_t *keys)
{
    const float *hdr = envhdr + (size_t)(first_env + (int)blockIdx.y) * MW_ENVHDR;
    keys += (size_t)blockIdx.y * W * H * S;
    if (S == 16) view_mesh_body<16>(W, H, hdr, mesh_pos, keys);
    else if (S == 4) view_mesh_body<4>(W, H, hdr, mesh_pos, keys);
    else if (S == 1) view_mesh_body<1>(W, H, hdr, mesh_pos, keys);
    else view_mesh_body<8>(W, H, hdr, mesh_pos, keys);
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
extern "C" __global__ __launch_bounds__(64) void mw_view_raster_kernel(
    int first_env, int W, int H, int S, int max_vis, int tiles_x, const float *__restrict__ rec_raster,
    const float *__restrict__ rec_shade, const float *__restrict__ rec_cull, const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr,
    const MwTexDesc *__restrict__ texd, const uint32_t *__restrict__ texels, const float *__restrict__ mesh_pos,
    const float *__restrict__ mesh_nrm, const float *__restrict__ mesh_rgb, const float *__restrict__ mesh_uv, const uint32_t *mesh_keys,
    uint8_t *__restrict__ out, float *__restrict__ depth, int texel_bytes)
{
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
    cx.env = 0; cx.nvis = nvis_arr[env]; cx.W = W; cx.H = H; cx.dbg = 0; cx.lane = threadIdx.x; cx.have_pre = 0; cx.order = nullptr; cx.tprof = nullptr;
    cx.pre_touch = cx.pre_full = cx.pre_clip = cx.pre_edges = 0ull;
    if (S == 16) view_tile_body<16>(cx, tiles_x, mesh_keys);
    else if (S == 4) view_tile_body<4>(cx, tiles_x, mesh_keys);
    else if (S == 1) view_tile_body<1>(cx, tiles_x, mesh_keys);
    else view_tile_body<8>(cx, tiles_x, mesh_keys);
}

// Order in which mw_raster_mesh_kernel's blocks take the envs: a counting sort of the envs by mesh triangles in
// view (the geometry kernel's k3_cost), most first; one workgroup, bins of 2048 triangles.
extern "C" __global__ __launch_bounds__(1024) void mw_mesh_order_kernel(int N, const int32_t *__restrict__ cost, int32_t *__restrict__ order)
{
    constexpr int BINS = 16;
    __shared__ int hist[BINS], start[BINS];
    const int tid = threadIdx.x;
    if (tid < BINS) hist[tid] = 0;
    __syncthreads();
    for (int e = tid; e < N; e += 1024) atomicAdd(&hist[(BINS - 1) - min(cost[e] >> 11, BINS - 1)], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int b = 0; b < BINS; ++b) { start[b] = acc; acc += hist[b]; }
    }
    __syncthreads();
    for (int e = tid; e < N; e += 1024) order[atomicAdd(&start[(BINS - 1) - min(cost[e] >> 11, BINS - 1)], 1)] = e;
}
