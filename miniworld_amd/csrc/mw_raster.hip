// K2 — the dominant kernel: 8x-MSAA coverage + 16-bit depth + GL_LINEAR_MIPMAP_LINEAR shading + resolve + pack, for every
// environment.  Replaces what the reference delegates to the GL driver per step: glClear, rasterisation / depth test of
// display list 1 and the entity draws, GL_MODULATE texturing, FrameBuffer.resolve()'s two blits + glReadPixels + flip
// (miniworld.py:1064-1086, 1193-1195; opengl.py:339-398) and get_depth_map (opengl.py:400-435) — llvmpipe's fragment
// pipeline (mw_frag.h).
//
// Mapping: one 64-lane wavefront (= one workgroup) owns a run of 16x4-pixel tiles of one env; lanes are pixels, quad by
// quad (mw_raster_common.h), each with its 8 samples' colours — and, where triangles contend for samples, their packed
// keys (depth16 << 16 | draw id) — in registers: no colour or depth buffer in memory at all.
//   coverage : per (tile, triangle) the raster record is wave-uniform and arrives through scalar loads (SGPRs); a sample
//              is inside edge k iff E_k(pixel) > thr_k[s] (integer edge functions, fill rule folded into C_k): 2 mul24 +
//              8 v_cmp per edge, the mask algebra on the SALU.
//   depth    : GL_LESS with first-drawn-wins == unsigned min of the packed keys; not needed while no sample is claimed twice.
//   shading  : each distinct triangle of the tile once for the whole wavefront (GL multisampling shades a pixel once per
//              triangle, at the pixel centre), the quad's texture-coordinate differences through DPP.
//   output   : RGB bytes staged through 192 B of LDS so the tile leaves as dword stores.
// HBM traffic per env-step is the observation (14 400 B, + 19 200 B with depth) plus the geometry kernel's records; textures
// and records are L2-resident.
#include <cstring>
#include "mw_mesh.h"

// LDS_RECS = true : the env's shade / classification records are staged in LDS (small scenes);
// LDS_RECS = false: they are read in place from global memory (L1/L2) — scenes with hundreds of
//                   visible primitives (Maze) would not leave room for enough resident waves.
// MESHAWARE   : envs may hold mesh entities — the tiles inside a mesh entity's tile rectangle (env header) start from the
//               sample keys the scatter kernel left (mw_raster_mesh.hip) and give them back cleared.
template <bool LDS_RECS, int FMT, int HOT = 0, int MESHAWARE = 0>
__device__ inline void raster_env_tiles(
    int N, int env, int t_begin, int t_end, int part_mode, int W, int H, int max_vis, int tiles_x, int n_tiles,
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const float *__restrict__ rec_cull,
    const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr, const MwTexDesc *__restrict__ texd,
    const uint32_t *__restrict__ texels, uint8_t *__restrict__ obs, float *__restrict__ depth, int dbg, int texel_bytes,
    const uint16_t *__restrict__ rec_order, const float *__restrict__ mesh_pos, const float *__restrict__ mesh_nrm,
    const float *__restrict__ mesh_rgb, const float *__restrict__ mesh_uv, uint32_t *__restrict__ mesh_keys,
    const float *__restrict__ plane_cache, int plane_cap, const float4 *__restrict__ slow_frags, const uint32_t *__restrict__ slow_head)
{
    // (one wavefront, the tiles t_begin .. t_end - 1 of env; part_mode: below)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lds_recs = LDS_RECS ? (max_vis < MW_LDS_RECS ? max_vis : MW_LDS_RECS) : 0;
    float4 *s_shade = reinterpret_cast<float4 *>(smem);                                       // [max_vis][8]
    float4 *s_cull = reinterpret_cast<float4 *>(smem + (size_t)lds_recs * MW_LDS_SHADE_Q * 16);       // [lds_recs][5]
    uint8_t *s_pack = smem + (size_t)lds_recs * (MW_LDS_SHADE_Q + MW_LDS_CULL_Q) * 16;                 // 192 B

    const int lane = threadIdx.x;
    [[maybe_unused]] const unsigned long long kp_e0 = K2P_NOW();
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    const bool mesh_env = MESHAWARE && __float_as_int(hdr[3]) != 0;
    uint32_t *env_keys = MESHAWARE == 1 ? mesh_keys + (size_t)env * W * H * 8 : nullptr;
    // mesh-aware launches come in two parts: 1 = every tile but those a mesh can touch — they need nothing of the mesh kernels and
    // run beside them —, 2 = those tiles only, behind the mesh kernels; 0 = all tiles
    if (part_mode == 2) {
        if (!mesh_env) return;
        bool any = false;
        for (int t = t_begin; t < t_end; ++t) any |= tile_in_mesh_rect(hdr, t % tiles_x, t / tiles_x);
        if (!any) return;
    }
    const int nvis = nvis_arr[env];
    const float *__restrict__ rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    const float4 *g_shade = reinterpret_cast<const float4 *>(rec_shade + (size_t)env * max_vis * MW_SHADE_REC);
    const float4 *g_cull = reinterpret_cast<const float4 *>(rec_cull + (size_t)env * max_vis * MW_CULL_REC);
    const bool in_lds = LDS_RECS && nvis <= lds_recs;
    if (in_lds) {
        __syncthreads();        // (a persistent wavefront: the previous item's readers are done)
        // (the quads K2 reads: 7 of a shade record's 8, 5 of a classification record's 6)
        for (int i = lane; i < nvis * MW_LDS_SHADE_Q; i += 64) { const int r = i / MW_LDS_SHADE_Q, q = i - r * MW_LDS_SHADE_Q; s_shade[i] = g_shade[r * (MW_SHADE_REC / 4) + q]; }
        for (int i = lane; i < nvis * MW_LDS_CULL_Q; i += 64) { const int r = i / MW_LDS_CULL_Q, q = i - r * MW_LDS_CULL_Q; s_cull[i] = g_cull[r * (MW_CULL_REC / 4) + q]; }
        __syncthreads();
    }
    const float sky_r = envhdr[(size_t)env * MW_ENVHDR + 0], sky_g = envhdr[(size_t)env * MW_ENVHDR + 1],
                sky_b = envhdr[(size_t)env * MW_ENVHDR + 2];
    TexEnv te;
    te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    te.td = te.tx;      // the descriptor table is the head of the texel block (upload_textures): texd == texels
    te.texd = reinterpret_cast<const MwTexDesc *>(texels);
    te.flat = HOT ? 0 : (dbg & 1);

    TileCtx cx;
    cx.s_shade = in_lds ? s_shade : g_shade; cx.s_cull = in_lds ? s_cull : g_cull;
    cx.shade_stride = in_lds ? MW_LDS_SHADE_Q : MW_SHADE_REC / 4; cx.cull_stride = in_lds ? MW_LDS_CULL_Q : MW_CULL_REC / 4; cx.rr_env = rr_env; cx.s_pack = s_pack; cx.hdr = hdr; cx.ment = hdr + MW_HDR_MESH;
    cx.mesh_pos = mesh_pos; cx.mesh_nrm = mesh_nrm; cx.mesh_rgb = mesh_rgb; cx.mesh_uv = mesh_uv;
    cx.planes = MESHAWARE == 1 ? plane_cache + (size_t)env * plane_cap * MW_PLANE_REC : nullptr;
    cx.planes_xtra = MESHAWARE == 1 ? plane_cache + (size_t)N * plane_cap * MW_PLANE_REC + (size_t)env * plane_cap * MW_PLANE_XTRA : nullptr;
    cx.clipbuf = nullptr;
    cx.slow_frags = MESHAWARE == 1 ? slow_frags + (size_t)env * MW_SLOW_STRIDE : nullptr;
    cx.slow_head = MESHAWARE == 1 ? slow_head + (size_t)env * W * H : nullptr;
    cx.slow_stamp = (uint32_t)dbg >> 16;
    cx.obs = obs; cx.depth = depth;
    cx.obs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(obs + (size_t)env * H * W * 3), 0, H * W * 3, MW_RSRC_WORD3); cx.te = te;
    cx.sky_r = sky_r; cx.sky_g = sky_g; cx.sky_b = sky_b;
    cx.env = env; cx.nvis = nvis; cx.W = W; cx.H = H; cx.dbg = dbg; cx.lane = lane;
#ifdef MW_PERF_HOOKS
    K2P_ADD(10, K2P_NOW() - kp_e0); K2P_ADD(14, 1);
#endif
    int tx = t_begin % tiles_x, ty = t_begin / tiles_x;
    // small scenes: classify (tile, primitive) pairs for as many tiles as fit in the 64 lanes at once
    const bool pairs = in_lds && nvis > 0 && nvis <= 32 && (HOT || !(dbg & 2));
    const int per_group = pairs ? 64 / nvis : 0;
    const uint64_t prim_mask = pairs ? ((1ull << nvis) - 1ull) : 0ull;
    cx.order = (!LDS_RECS && rec_order) ? rec_order + (size_t)env * (max_vis + 1) : nullptr;
    cx.pre_touch = cx.pre_full = cx.pre_clip = cx.pre_edges = 0ull;
    cx.have_pre = 0;
    if (pairs) {
        // the masks of the g-th tile of the current group live in lane g of three VGPRs (<= 32 primitives: 32 bits
        // each); a tile fetches its three with v_readlane instead of carrying 64-bit group masks through the tile
        // loop in SGPRs
        uint32_t vT = 0u, vF = 0u;
        uint32_t vE01 = ~0u, vE23 = ~0u;    // per-edge "needs a test" masks, 16 bits each (more than 16 primitives: test all)
        int gi = 0, G = 0;
        cx.have_pre = 1;
        for (int tile = t_begin; tile < t_end; ++tile, tx = (tx + 1 == tiles_x) ? 0 : tx + 1, ty += (tx == 0)) {
            if (gi == G) {
                G = min(per_group, t_end - tile);
                gi = 0;
                uint64_t T, F, Eo[3];
                classify_group(s_cull, MW_LDS_CULL_Q, lane, nvis, tile, G, tiles_x, H, T, F, Eo);
                const int sh = lane < G ? lane * nvis : 0;
                vT = (uint32_t)((T >> sh) & prim_mask); vF = (uint32_t)((F >> sh) & prim_mask);
                if (nvis <= 16) {
                    vE01 = (uint32_t)((Eo[0] >> sh) & prim_mask) | ((uint32_t)((Eo[1] >> sh) & prim_mask) << 16);
                    vE23 = (uint32_t)((Eo[2] >> sh) & prim_mask) | 0xFFFF0000u;
                }
            }
            cx.pre_edges = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)vE01, gi) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)vE23, gi) << 32);
            cx.pre_touch = (uint32_t)__builtin_amdgcn_readlane((int)vT, gi);
            cx.pre_full = (uint32_t)__builtin_amdgcn_readlane((int)vF, gi);
            ++gi;
            const bool mesh_tile = MESHAWARE && mesh_env && tile_in_mesh_rect(hdr, tx, ty);
            if (MESHAWARE == 2 && mesh_tile) continue;
            if (MESHAWARE && part_mode != 0 && mesh_tile != (part_mode == 2)) continue;
            if (MESHAWARE == 1 && mesh_tile) {
                uint32_t mk[8];
                [[maybe_unused]] const unsigned long long kk0 = K2P_NOW();
                take_mesh_keys(env_keys, W, tx, ty, lane, mk);
#ifdef MW_PERF_HOOKS
                { uint32_t o = 0; for (int q = 0; q < 8; ++q) o |= mk[q]; if (o == 0x12345u) K2P_ADD(14, 0); K2P_ADD(11, K2P_NOW() - kk0); }
#endif
                raster_tile_fmt<true, FMT, false, HOT, 1>(cx, tx, ty, mk);
                continue;
            }
            raster_tile_fmt<false, FMT, false, HOT, 1>(cx, tx, ty, nullptr);
        }
        return;
    }
    for (int tile = t_begin; tile < t_end; ++tile, tx = (tx + 1 == tiles_x) ? 0 : tx + 1, ty += (tx == 0)) {
        const bool mesh_tile = MESHAWARE && mesh_env && tile_in_mesh_rect(hdr, tx, ty);
        if (MESHAWARE == 2 && mesh_tile) continue;
        if (MESHAWARE && part_mode != 0 && mesh_tile != (part_mode == 2)) continue;
        if (MESHAWARE == 1 && mesh_tile) {
            uint32_t mk[8];
            [[maybe_unused]] const unsigned long long kk0 = K2P_NOW();
            take_mesh_keys(env_keys, W, tx, ty, lane, mk);
#ifdef MW_PERF_HOOKS
            { uint32_t o = 0; for (int q = 0; q < 8; ++q) o |= mk[q]; if (o == 0x12345u) K2P_ADD(14, 0); K2P_ADD(11, K2P_NOW() - kk0); }
#endif
            raster_tile_fmt<true, FMT, false, HOT, 0>(cx, tx, ty, mk);
            continue;
        }
        if (!LDS_RECS && rec_order) raster_tile_fmt<false, FMT, true, HOT, 0>(cx, tx, ty, nullptr);
        else raster_tile_fmt<false, FMT, false, HOT, 0>(cx, tx, ty, nullptr);
    }
}


// One wavefront (= one workgroup) per run of tiles of one env; or, for the tiles a mesh can touch (launch flags bits 4-5 = 3),
// persistent wavefronts drawing (env, tile) items from the list the geometry kernel left (mw_geom.hip: the union of the
// entities' tile rectangles; a grid of one wavefront per tile of every env spent most of its time starting wavefronts that
// found no mesh in their env).
template <bool LDS_RECS, int FMT, int HOT = 0, int MESHAWARE = 0>
__device__ inline void raster_kernel_body(
    int N, int W, int H, int max_vis, int tiles_x, int n_tiles, int waves_per_env, int tiles_per_wave,
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const float *__restrict__ rec_cull,
    const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr, const MwTexDesc *__restrict__ texd,
    const uint32_t *__restrict__ texels, uint8_t *__restrict__ obs, float *__restrict__ depth, int dbg, int texel_bytes,
    const uint16_t *__restrict__ rec_order, const float *__restrict__ mesh_pos, const float *__restrict__ mesh_nrm,
    const float *__restrict__ mesh_rgb, const float *__restrict__ mesh_uv, uint32_t *__restrict__ mesh_keys,
    const float *__restrict__ plane_cache, int plane_cap, const float4 *__restrict__ slow_frags, const uint32_t *__restrict__ slow_head,
    const uint32_t *__restrict__ tile_list, int32_t *__restrict__ tile_n, int tile_list_cap, int n_xcc)
{
    if (MESHAWARE == 1) __builtin_amdgcn_s_setprio(3);      // (the mesh tiles end the frame's critical path: ahead of the quad kernel's wavefronts)
    // (one call site for both forms: two copies of the tile code in one kernel cost it 150 registers)
    const int part_mode = MESHAWARE ? (dbg >> 4) & 3 : 0;
    const bool listed = MESHAWARE == 1 && part_mode == 3;
    // tile_n[MW_CNT_TILES + x]: items of XCD x's list (mw_mesh_entity_kernel zeroes them for the frame after the next).  Workgroup b is
    // dispatched to XCD b % n_xcc (tools/ubench/xcc_probe.hip: all of 4 096): it takes the items b / n_xcc, + grid / n_xcc, ... of that
    // XCD's list — the envs e % n_xcc == x, whose records and keys thus meet one L2 only.  (A dispatch that did otherwise would cost
    // locality, not frames.)
    const int xl = listed ? (int)blockIdx.x % n_xcc : 0, stride = listed ? ((int)gridDim.x - xl + n_xcc - 1) / n_xcc : 1;      // (the launch's wavefronts b % n_xcc == xl: every item once)
    const int n_items = listed ? min(tile_n[MW_CNT_TILES + xl], tile_list_cap) : 1;
    if (listed) tile_list += (size_t)xl * tile_list_cap;
    for (int i = listed ? (int)blockIdx.x / n_xcc : 0; i < n_items; i += stride) {
        int env, t_begin, t_end;
        if (listed) {
            const uint32_t item = tile_list[i];
            env = (int)(item & 0xFFFFFFu); t_begin = (int)(item >> 24); t_end = t_begin + 1;
        } else {
            // XCD-aware block -> (env, part): the parts of one env run on the same XCD (block b is
            // dispatched to XCD b % 8), adjacent in time, so its records are fetched into one L2 once.
            const int b = blockIdx.x;
            const int xcd = b & 7, slot = b >> 3;
            env = (slot / waves_per_env) * 8 + xcd;
            if (env >= N) return;
            t_begin = (slot % waves_per_env) * tiles_per_wave; t_end = min(t_begin + tiles_per_wave, n_tiles);
        }
        raster_env_tiles<LDS_RECS, FMT, HOT, MESHAWARE>(N, env, t_begin, t_end, listed ? 2 : part_mode, W, H, max_vis, tiles_x, n_tiles, rec_raster, rec_shade, rec_cull, nvis_arr,
                                                        envhdr, texd, texels, obs, depth, dbg, texel_bytes, rec_order, mesh_pos, mesh_nrm, mesh_rgb, mesh_uv, mesh_keys,
                                                        plane_cache, plane_cap, slow_frags, slow_head);
        if (!listed) return;
    }
}

// (texd == texels: the descriptor table is the head of the texel block, mw_engine.hip::upload_textures; the kernels
// use `texels` for both)
#define MW_RASTER_ARGS \
    int N, int W, int H, int max_vis, int tiles_x, int n_tiles, int waves_per_env, int tiles_per_wave, \
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const float *__restrict__ rec_cull, \
    const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr, const MwTexDesc *__restrict__ texd, \
    const uint32_t *__restrict__ texels, uint8_t *__restrict__ obs, float *__restrict__ depth, int dbg, int texel_bytes, \
    const uint16_t *__restrict__ rec_order, const float *__restrict__ mesh_pos, const float *__restrict__ mesh_nrm, \
    const float *__restrict__ mesh_rgb, const float *__restrict__ mesh_uv, uint32_t *__restrict__ mesh_keys, \
    const float *__restrict__ plane_cache, int plane_cap, const float4 *__restrict__ slow_frags, const uint32_t *__restrict__ slow_head, \
    const uint32_t *__restrict__ tile_list, int32_t *__restrict__ tile_n, int tile_list_cap, int n_xcc
#define MW_RASTER_FWD N, W, H, max_vis, tiles_x, n_tiles, waves_per_env, tiles_per_wave, rec_raster, rec_shade, rec_cull, \
    nvis_arr, envhdr, texd, texels, obs, depth, dbg, texel_bytes, rec_order, mesh_pos, mesh_nrm, mesh_rgb, mesh_uv, mesh_keys, plane_cache, plane_cap, slow_frags, slow_head, tile_list, tile_n, tile_list_cap, n_xcc

// the production kernels of small scenes: no debug flags (mw_engine.hip launches the general kernel below when
// MW_DEBUG_FLAGS asks for any), RGB only / RGB + depth
extern "C" __global__ __launch_bounds__(64) void mw_raster_kernel(MW_RASTER_ARGS)
{
    raster_kernel_body<true, 0, 1>(MW_RASTER_FWD);
}

extern "C" __global__ __launch_bounds__(64) void mw_raster_depth_kernel(MW_RASTER_ARGS)
{
    raster_kernel_body<true, 0, 2>(MW_RASTER_FWD);
}

#ifndef MW_K2BIG_OCC
#define MW_K2BIG_OCC
#endif
extern "C" __global__ __launch_bounds__(64) MW_K2BIG_OCC void mw_raster_big_kernel(MW_RASTER_ARGS)
{
    raster_kernel_body<false, 0, 1>(MW_RASTER_FWD);
}

extern "C" __global__ __launch_bounds__(64) void mw_raster_big_depth_kernel(MW_RASTER_ARGS)
{
    raster_kernel_body<false, 0, 2>(MW_RASTER_FWD);
}

// the general kernels: output layout (mw_set_obs_layout; dbg bits 8-9), debug flags and depth read from the launch
extern "C" __global__ __launch_bounds__(64) void mw_raster_wrap_kernel(MW_RASTER_ARGS)
{
    raster_kernel_body<true, -1>(MW_RASTER_FWD);
}

extern "C" __global__ __launch_bounds__(64) void mw_raster_big_wrap_kernel(MW_RASTER_ARGS)
{
    raster_kernel_body<false, -1>(MW_RASTER_FWD);
}

#ifndef MW_MESH_TILE_OCC
#define MW_MESH_TILE_OCC 4
#endif
// the same for envs that may hold mesh entities (PickupObjects, Sign, CollectHealth, ...)
// ... the tiles no mesh can touch, of envs that hold mesh entities (K2's first part, beside the mesh kernels): the plain tile code,
// at the plain kernels' register count
extern "C" __global__ __launch_bounds__(64) void mw_raster_nomesh_kernel(MW_RASTER_ARGS) { raster_kernel_body<true, 0, 1, 2>(MW_RASTER_FWD); }
extern "C" __global__ __launch_bounds__(64) void mw_raster_nomesh_depth_kernel(MW_RASTER_ARGS) { raster_kernel_body<true, 0, 2, 2>(MW_RASTER_FWD); }
extern "C" __global__ __launch_bounds__(64, MW_MESH_TILE_OCC) void mw_raster_mesh_kernel(MW_RASTER_ARGS) { raster_kernel_body<true, 0, 1, 1>(MW_RASTER_FWD); }
extern "C" __global__ __launch_bounds__(64, MW_MESH_TILE_OCC) void mw_raster_mesh_depth_kernel(MW_RASTER_ARGS) { raster_kernel_body<true, 0, 2, 1>(MW_RASTER_FWD); }
extern "C" __global__ __launch_bounds__(64) void mw_raster_mesh_wrap_kernel(MW_RASTER_ARGS) { raster_kernel_body<true, -1, 0, 1>(MW_RASTER_FWD); }
extern "C" __global__ __launch_bounds__(64) void mw_raster_big_mesh_wrap_kernel(MW_RASTER_ARGS) { raster_kernel_body<false, -1, 0, 1>(MW_RASTER_FWD); }

#ifdef MW_PERF_HOOKS
// tools/perf/k2prof.py: read (and zero) this translation unit's phase counters
extern "C" int mw_debug_k2prof(unsigned long long *out16)
{
    static unsigned long long h[K2P_SLOTS][16];
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_k2prof), sizeof h) != hipSuccess) return -2;
    for (int i = 0; i < 16; ++i) { out16[i] = 0; for (int k = 0; k < K2P_SLOTS; ++k) out16[i] += h[k][i]; }
    memset(h, 0, sizeof h);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_k2prof), h, sizeof h) == hipSuccess ? 0 : -3;
}
#endif
