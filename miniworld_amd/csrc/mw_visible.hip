// get_visible_ents — occlusion queries over the proxy boxes (miniworld.py:1238-1333).
//
// The geometry kernel (mw_geom.hip, view_flags bit 2) has left, per env, the triangle records of the room
// polygons followed by the front faces of one 0.2 m proxy box per entity, in self.entities (slot)
// order, each tagged 0x10000 | slot in its draw-id field.  One workgroup per env keeps the
// per-sample 16-bit depth buffer of the obs frame (W x H x S samples, one dword each) in LDS:
//   phase 1  rooms: depth only, unsigned min == GL_LESS;
//   phase 2  proxies, one entity at a time (a barrier between entities keeps GL's draw order):
//            a sample passes iff its depth is strictly below the stored one (GL_LESS) — the faces
//            of one convex box never share a sample, so test and write fuse into one atomic min.
// vis[env][slot] = 1 iff any sample passed == GL_ANY_SAMPLES_PASSED (:1296, :1325).
// Coverage and depth are the raster kernels' (mw_records.h: integer edge functions, z plane at the sample position).
#include "mw_records.h"
#include "mw_frag.h"

namespace {

__device__ inline void visit_prim(const int *__restrict__ rr, const int *__restrict__ cr, uint32_t *zbuf, int W, int H, int S, int tid,
                                  int nthreads, bool query, int *passed)
{
    const uint32_t bb = (uint32_t)cr[3];
    const int x0 = (int)(bb & 255u) * MW_TILE_W, x1 = min((int)((bb >> 8) & 255u) * MW_TILE_W + MW_TILE_W - 1, W - 1);
    const int y0 = (int)((bb >> 16) & 255u) * MW_TILE_H, y1 = min((int)(bb >> 24) * MW_TILE_H + MW_TILE_H - 1, H - 1);
    if (x1 < x0 || y1 < y0) return;
    const int bw = x1 - x0 + 1, npx = bw * (y1 - y0 + 1);
    const int pi = mwrec::pat_index(S);
    const mwgl::Plane zp = {__int_as_float(rr[10]), __int_as_float(rr[11]), __int_as_float(rr[12])};
    bool any_pass = false;
    for (int i = tid; i < npx; i += nthreads) {
        const int px = x0 + i % bw, py = y0 + i / bw, gy = H - 1 - py;
        int64_t E[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            E[k] = (((int64_t)cr[20 + k] << 32) | (uint32_t)rr[6 + k]) + (int64_t)rr[k] * px + (int64_t)rr[3 + k] * gy;
        uint32_t *zrow = zbuf + ((size_t)py * W + px) * S;
        for (int s = 0; s < S; ++s) {
            const bool in = E[0] > (int64_t)rr[16 + s] && E[1] > (int64_t)rr[32 + s] && E[2] > (int64_t)rr[48 + s];
            if (in) {
                const float xs = (float)px + (S == 1 ? 0.0f : (float)mwrec::kPat[pi][s][0] * 0.0625f);
                const float ys = (float)gy + (S == 1 ? 0.0f : (float)mwrec::kPat[pi][s][1] * 0.0625f);
                const uint32_t z16 = mwgl::z_to_unorm16(mwgl::plane_at(zp, xs, ys));
                const uint32_t old = atomicMin(zrow + s, z16);
                any_pass |= z16 < old;
            }
        }
    }
    if (query && any_pass) *passed = 1;
}

}  // namespace

extern "C" __global__ __launch_bounds__(256) void mw_visible_kernel(int env_base, int W, int H, int S, int max_vis, int E,
                                                                    const float *__restrict__ rec_raster, const float *__restrict__ rec_cull,
                                                                    const int32_t *__restrict__ nvis,
                                                                    uint8_t *__restrict__ vis)
{
    extern __shared__ uint32_t zbuf[];          // [H][W][S]
    __shared__ int s_passed[64];
    __shared__ int s_nroom;
    const int env = env_base + blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
    const int nv = nvis[env];
    const int *__restrict__ rr_env = reinterpret_cast<const int *>(rec_raster + (size_t)env * max_vis * MW_RASTER_REC);
    const int *__restrict__ cr_env = reinterpret_cast<const int *>(rec_cull + (size_t)env * max_vis * MW_CULL_REC);
    for (int i = tid; i < W * H * S; i += nt) zbuf[i] = 65535u;
    if (tid < 64) s_passed[tid] = 0;
    if (tid == 0) s_nroom = nv;
    __syncthreads();
    for (int p = tid; p < nv; p += nt)
        if ((uint32_t)rr_env[(size_t)p * MW_RASTER_REC + 9] >= 0x10000u) atomicMin(&s_nroom, p);
    __syncthreads();
    const int nroom = s_nroom;
    for (int p = 0; p < nroom; ++p)
        visit_prim(rr_env + (size_t)p * MW_RASTER_REC, cr_env + (size_t)p * MW_CULL_REC, zbuf, W, H, S, tid, nt, false, nullptr);
    int cur = -1;
    for (int p = nroom; p < nv; ++p) {
        const int *__restrict__ rr = rr_env + (size_t)p * MW_RASTER_REC;
        const uint32_t tag = (uint32_t)rr[9];
        if (tag < 0x10000u) continue;           // an unused clipper slot
        const int slot = (int)(tag & 0xFFFFu);
        if (slot != cur) { __syncthreads(); cur = slot; }
        visit_prim(rr, cr_env + (size_t)p * MW_CULL_REC, zbuf, W, H, S, tid, nt, true, &s_passed[slot & 63]);
    }
    __syncthreads();
    if (tid < E) vis[(size_t)blockIdx.x * E + tid] = (uint8_t)(s_passed[tid] != 0);
}
