// The mesh scatter kernel (mesh triangles -> sample keys + attribute planes of the winners) and the generic-resolution
// view kernels.  See mw_mesh.h.
#include "mw_mesh.h"

// Grid (MW_SCATTER_BLOCKS_X, N): the blocks of row y walk the mesh entities in view of env y, 256 triangles per block and
// step, one triangle per lane, read from the mesh's stream in the order sorted by face normal (mw_upload_mesh: the 64
// triangles of a wavefront face the same way, so back-face culling retires whole waves; 48 contiguous bytes per lane).  keys: [N][H][W][8] dwords, all 0xFFFFFFFF on entry inside the
// entities' tile rectangles (K2 resets what it reads); planes: [N][plane_cap][MW_PLANE_REC].
extern "C" __global__ __launch_bounds__(256) void mw_mesh_scatter_kernel(int W, int H, const float *__restrict__ envhdr,
                                                                        const float *__restrict__ mesh_stream, const float *__restrict__ mesh_attr,
                                                                        uint32_t *__restrict__ keys_all, float *__restrict__ plane_cache, int plane_cap,
                                                                        int32_t *__restrict__ slow_count, uint32_t *__restrict__ slow_tris)
{
    const int env = blockIdx.y;
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    const int n_mesh = __float_as_int(hdr[3]);
    if (n_mesh == 0) return;
    uint32_t *keys = keys_all + (size_t)env * W * H * 8;
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    for (int j = 0; j < n_mesh; ++j) {
        const MeshEnt e = load_ment(hdr + MW_HDR_MESH, j);
        float *cache_e = plane_cache + ((size_t)env * plane_cap + (size_t)__float_as_int(hdr[MW_HDR_MESH + MW_HDR_MESH_STRIDE * j + 25])) * MW_PLANE_REC;
        for (int t = (int)blockIdx.x * 256 + (int)threadIdx.x; t < e.ntris; t += (int)gridDim.x * 256) {
            // record t of the entity's stream: the t-th triangle of the rasterisation order, coalesced
            const float4 *rec = reinterpret_cast<const float4 *>(mesh_stream) + (size_t)(e.first + t) * 3;
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const float pos[9] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x};
            const int tri = __float_as_int(r2.y);
            raster_tri_obs(f, e, tri, pos, W, H, keys, reinterpret_cast<const float4 *>(mesh_attr) + (size_t)(e.first + t) * 6, cache_e, j, slow_count + env, slow_tris + (size_t)env * MW_SLOW_TRIS);
        }
    }
}

// The mesh triangles that cross a frustum plane (a mesh at the frame's edge; the scatter kernel lists them): clipped, every
// piece set up on its own (llvmpipe's clipper output), its keys scattered and its fragments shaded right here — one entry
// (draw id, colour) per pixel with a covered sample, chained per pixel, in fan order — so that neither the scatter kernel
// nor K2 carries the clipper.  A triangle per lane (work lists in LDS) up to the pieces' setup, then a (piece, pixel)
// pair per lane.  Exits at once for an env without such triangles.
#define MW_SLOW_LANES 16

namespace {

struct SlowPiece {       // what the pixel loop needs of one set-up piece
    int dcdx[3], dcdy[3], c[3];
    mwgl::Plane z, w, s, t, r, g, b;
    int x0, x1, y0, y1;
    uint32_t id;
    int tex;
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
            atomicMin(kp + s, (mwgl::z_to_unorm16(mwgl::plane_at(p.z, xs, ys)) << 16) | p.id);
            any = true;
        }
    }
    return any;
}

// the piece's fragment of pixel (px, gy) as entry k of the env's list
// (a pixel's chain head is (frame stamp << 16) | index + 1: heads of earlier frames read as empty, nothing is cleared)
__device__ inline void slow_frag(const SlowPiece &p, int px, int gy, int W, int H, const TexEnv &te, int k, float4 *frags, uint32_t stamp,
                                 uint32_t *head, uint32_t *status)
{
    if (k >= MW_SLOW_FRAGS) { atomicOr(status, MW_ST_VIS_OVERFLOW); return; }
    const RGB c = shade_planes(p.w, p.s, p.t, p.r, p.g, p.b, p.tex, te, px, gy, 0.5f);
    const uint32_t pix = (uint32_t)((H - 1 - gy) * W + px);
    const uint32_t old = atomicExch(head + pix, (stamp << 16) | ((uint32_t)k + 1u));
    const uint32_t next = (old >> 16) == stamp ? (old & 0xFFFFu) : 0u;
    frags[k] = make_float4(__uint_as_float((p.id << 16) | next), c.r, c.g, c.b);
}

__device__ inline void slow_pixel(const SlowPiece &p, int px, int gy, int W, int H, uint32_t *keys, const TexEnv &te, int32_t *frag_count,
                                  float4 *frags, uint32_t stamp, uint32_t *head, uint32_t *status)
{
    if (slow_cover(p, px, gy, W, H, keys)) slow_frag(p, px, gy, W, H, te, atomicAdd(frag_count, 1), frags, stamp, head, status);
}

}  // namespace

extern "C" __global__ __launch_bounds__(64) void mw_mesh_slow_kernel(int W, int H, const float *__restrict__ envhdr, const float *__restrict__ mesh_pos,
                                                                    const float *__restrict__ mesh_nrm, const float *__restrict__ mesh_rgb,
                                                                    const float *__restrict__ mesh_uv, const uint32_t *__restrict__ texels, int texel_bytes,
                                                                    uint32_t *__restrict__ keys_all, int32_t *__restrict__ counts, int N, int parity,
                                                                    const uint32_t *__restrict__ slow_tris, float4 *__restrict__ frags_all,
                                                                    uint32_t *__restrict__ heads_all, uint32_t stamp, uint32_t *__restrict__ status)
{
    // the clipper's work lists: [MW_SLOW_LANES][2][MWGL_MAX_CLIP_VERTS] vertices (18 KB: the grid is mostly empty workgroups, which
    // must not queue for LDS)
    __shared__ mwgl::Vert slow_lists[MW_SLOW_LANES][2][MWGL_MAX_CLIP_VERTS];
    __shared__ SlowPiece s_piece[MW_SLOW_LANES];        // the pieces of one step of the fans, and where each one's pixels start in the step's pixel list
    __shared__ int s_pref[MW_SLOW_LANES + 1];
    // grid (MW_SLOW_BLOCKS_X, N): the blocks of row y share env y's list, MW_SLOW_LANES triangles per block and turn.
    // counts: [2 parities][2][N] — listed triangles (the scatter kernel's) and fragments of this frame's parity; the other
    // parity's are zeroed here for the next frame.
    const int env = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
    int32_t *slow_count = counts + ((size_t)parity * 2 + 0) * N, *frag_count = counts + ((size_t)parity * 2 + 1) * N;
    if (blockIdx.x == 0 && tid == 0) { counts[((size_t)(parity ^ 1) * 2 + 0) * N + env] = 0; counts[((size_t)(parity ^ 1) * 2 + 1) * N + env] = 0; }
    const int n = slow_count[env];
    if ((int)blockIdx.x * MW_SLOW_LANES >= n) return;
    uint32_t *head = heads_all + (size_t)env * W * H;
    float4 *frags = frags_all + (size_t)env * MW_SLOW_FRAGS;
    if (n > MW_SLOW_TRIS && tid == 0) atomicOr(status, MW_ST_VIS_OVERFLOW);
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    uint32_t *keys = keys_all + (size_t)env * W * H * 8;
    TexEnv te;
    te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    te.td = te.tx;
    te.texd = reinterpret_cast<const MwTexDesc *>(texels);
    te.flat = 0;
    mwgl::Frame f;
    frame_lite(hdr, W, H, f);
    const int nn = min(n, MW_SLOW_TRIS);
    mwgl::Vert *buf0 = slow_lists[tid & (MW_SLOW_LANES - 1)][0], *buf1 = slow_lists[tid & (MW_SLOW_LANES - 1)][1];
    for (int base = (int)blockIdx.x * MW_SLOW_LANES; base < nn; base += (int)gridDim.x * MW_SLOW_LANES) {
        const int i = base + tid;
        const bool valid = tid < MW_SLOW_LANES && i < nn;
        mwgl::Vert *r = buf0;
        int nv = 0, tex = -1;
        uint32_t id = 0u;
        if (valid) {
            const uint32_t it = slow_tris[(size_t)env * MW_SLOW_TRIS + i];
            const MeshEnt e = load_ment(hdr + MW_HDR_MESH, (int)(it >> 16));
            const int tri = (int)(it & 0xFFFFu);
            id = (uint32_t)(e.start + tri);
            tex = e.tex;
            float pos[9];
            tri_load(mesh_pos, e, tri, pos);
            const float *nrm = mesh_nrm + (size_t)(e.first + tri) * 9, *rgb = mesh_rgb + (size_t)(e.first + tri) * 9;
            const float *uv = mesh_uv + (size_t)(e.first + tri) * 6;
            mwgl::Vert v[3];
            for (int k = 0; k < 3; ++k) {
                const float p[3] = {pos[k * 3], pos[k * 3 + 1], pos[k * 3 + 2]};
                mwgl::transform_vertex(f, e.x, p, v[k]);
                const float nv3[3] = {nrm[k * 3], nrm[k * 3 + 1], nrm[k * 3 + 2]}, c[3] = {rgb[k * 3], rgb[k * 3 + 1], rgb[k * 3 + 2]};
                mwgl::light_vertex(f, e.x, nv3, c, v[k].col);
                v[k].st[0] = e.tex >= 0 ? uv[k * 2] : 0.0f;
                v[k].st[1] = e.tex >= 0 ? uv[k * 2 + 1] : 0.0f;
            }
            nv = mwgl::clip_triangle<true>(f, v[0], v[1], v[2], buf0, buf1, &r);
        }
        // the pieces (r[q-1], r[q], r[0]), q = 2 .. nv - 1, in fan order
        for (int q = 2; __any(q < nv); ++q) {
            SlowPiece p;
            bool have = false;
            if (q < nv) {
                mwgl::TriSetup ts;
                if (mwgl::setup_triangle(r[q - 1], r[q], r[0], true, tex >= 0, ts)) {
                    have = true;
#pragma unroll
                    for (int k = 0; k < 3; ++k) { p.dcdx[k] = ts.dcdx[k]; p.dcdy[k] = ts.dcdy[k]; p.c[k] = (int)ts.c[k]; }
                    p.z = ts.z; p.w = ts.w; p.s = ts.s; p.t = ts.t; p.r = ts.col[0]; p.g = ts.col[1]; p.b = ts.col[2];
                    p.x0 = max(ts.minx >> 8, 0); p.x1 = min(ts.maxx >> 8, W - 1);
                    p.y0 = max(ts.miny >> 8, 0); p.y1 = min(ts.maxy >> 8, H - 1);
                    p.id = id; p.tex = tex;
                    have = p.x0 <= p.x1 && p.y0 <= p.y1;
                }
            }
            // the pixels of all pieces of this step of the fans, one (piece, pixel) pair per lane and turn: a fragment costs
            // two dependent texture reads, and a lane walking its own piece pays them pixel after pixel
            const int npx = have ? (p.x1 - p.x0 + 1) * (p.y1 - p.y0 + 1) : 0;
            int incl = npx;
#pragma unroll
            for (int off = 1; off < MW_SLOW_LANES; off <<= 1) { const int y = __shfl_up(incl, off); if (lane >= off) incl += y; }
            if (tid < MW_SLOW_LANES) { s_pref[tid] = incl - npx; if (have) s_piece[tid] = p; }
            if (tid == MW_SLOW_LANES - 1) s_pref[MW_SLOW_LANES] = incl;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const int total = s_pref[MW_SLOW_LANES];
            for (int k = lane; k < total; k += 64) {
                int j = 0;
#pragma unroll
                for (int step = MW_SLOW_LANES / 2; step > 0; step >>= 1) if (s_pref[j + step] <= k) j += step;      // the last piece that starts at or before k
                const SlowPiece u = s_piece[j];
                const int kk = k - s_pref[j], bw = u.x1 - u.x0 + 1;
                slow_pixel(u, u.x0 + kk % bw, u.y0 + kk / bw, W, H, keys, te, frag_count + env, frags, stamp, head, status);
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
    cx.env = 0; cx.nvis = nvis_arr[env]; cx.W = W; cx.H = H; cx.dbg = 0; cx.lane = threadIdx.x; cx.have_pre = 0; cx.order = nullptr;
    cx.planes = nullptr; cx.slow_frags = nullptr; cx.slow_head = nullptr; cx.slow_stamp = 0u;
    cx.pre_touch = cx.pre_full = cx.pre_clip = cx.pre_edges = 0ull;
    if (S == 16) view_tile_body<16>(cx, tiles_x, mesh_keys);
    else if (S == 4) view_tile_body<4>(cx, tiles_x, mesh_keys);
    else if (S == 1) view_tile_body<1>(cx, tiles_x, mesh_keys);
    else view_tile_body<8>(cx, tiles_x, mesh_keys);
}

