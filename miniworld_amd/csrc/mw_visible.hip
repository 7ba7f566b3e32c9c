// get_visible_ents — occlusion queries over the proxy boxes (miniworld.py:1238-1333).
//
// K1 (mw_step_setup_kernel, view_flags bit 2) has left, per env, the raster records of the room
// polygons followed by the front faces of one 0.2 m proxy box per entity, in self.entities (slot)
// order, each tagged 0x10000 | slot in its draw-id field.  One workgroup per env keeps the
// per-sample 16-bit depth buffer of the obs frame (W x H x 8 samples, one dword each) in LDS:
//   phase 1  rooms: depth only, unsigned min == GL_LESS;
//   phase 2  proxies, one entity at a time (a barrier between entities keeps GL's draw order):
//            a sample passes iff its depth is strictly below the stored one (GL_LESS) — the faces
//            of one convex box never share a sample, so test and write fuse into one atomic min.
// vis[env][slot] = 1 iff any sample passed == GL_ANY_SAMPLES_PASSED (:1296, :1325).
#include "mw_device.h"

namespace {

__device__ inline void visit_prim(const float *__restrict__ rr, uint32_t *zbuf, int W, int tid, int nthreads,
                                  bool query, int *passed)
{
    const uint32_t bb = __float_as_uint(rr[15]);
    const int x0 = (int)(bb & 255u) * MW_TILE_W, x1 = (int)((bb >> 8) & 255u) * MW_TILE_W + MW_TILE_W - 1;
    const int y0 = (int)((bb >> 16) & 255u) * MW_TILE_H, y1 = (int)(bb >> 24) * MW_TILE_H + MW_TILE_H - 1;
    const int bw = x1 - x0 + 1, npx = bw * (y1 - y0 + 1);
    bool any_pass = false;
    for (int i = tid; i < npx; i += nthreads) {
        const int px = x0 + i % bw, py = y0 + i / bw;
        const float Xc = (float)px + 0.5f, Yc = (float)py + 0.5f;
        float E[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) E[k] = fmaf(rr[k], Xc, fmaf(rr[4 + k], Yc, rr[8 + k]));
        const float zc = fmaf(rr[12], Xc, fmaf(rr[13], Yc, rr[14]));
        uint32_t *zp = zbuf + ((size_t)py * W + px) * 8;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            bool in = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) in &= E[k] > rr[16 + k * 8 + s];
            const float zs = zc + rr[48 + s];
            const float t = fmaf(zs, 65535.0f, 0.5f);
            if (in && t >= 0.5f && t < 65536.0f) {        // R6 near / far clip
                const uint32_t z16 = (uint32_t)t;
                const uint32_t old = atomicMin(zp + s, z16);
                any_pass |= z16 < old;
            }
        }
    }
    if (query && any_pass) *passed = 1;
}

}  // namespace

extern "C" __global__ __launch_bounds__(256) void mw_visible_kernel(int env_base, int W, int H, int max_vis, int E,
                                                                    const float *__restrict__ rec_raster,
                                                                    const int32_t *__restrict__ nvis,
                                                                    uint8_t *__restrict__ vis)
{
    extern __shared__ uint32_t zbuf[];          // [H][W][8]
    __shared__ int s_passed[64];
    __shared__ int s_nroom;
    const int env = env_base + blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nv = nvis[env];
    const float *__restrict__ rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    for (int i = tid; i < W * H * 8; i += nt) zbuf[i] = 65535u;
    if (tid < 64) s_passed[tid] = 0;
    if (tid == 0) s_nroom = nv;
    __syncthreads();
    for (int p = tid; p < nv; p += nt)
        if (__float_as_uint(rr_env[(size_t)p * MW_RASTER_REC + 61]) >= 0x10000u) atomicMin(&s_nroom, p);
    __syncthreads();
    const int nroom = s_nroom;
    for (int p = 0; p < nroom; ++p) visit_prim(rr_env + (size_t)p * MW_RASTER_REC, zbuf, W, tid, nt, false, nullptr);
    int cur = -1;
    for (int p = nroom; p < nv; ++p) {
        const float *__restrict__ rr = rr_env + (size_t)p * MW_RASTER_REC;
        const int slot = (int)(__float_as_uint(rr[61]) & 0xFFFFu);
        if (slot != cur) { __syncthreads(); cur = slot; }
        visit_prim(rr, zbuf, W, tid, nt, true, &s_passed[slot & 63]);
    }
    __syncthreads();
    if (tid < E) vis[(size_t)blockIdx.x * E + tid] = (uint8_t)(s_passed[tid] != 0);
}
