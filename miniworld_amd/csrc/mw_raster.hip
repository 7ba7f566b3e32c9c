// K2 — the dominant kernel: 8x-MSAA coverage + 16-bit depth + deferred trilinear shading +
// resolve + pack, for every environment.  Replaces what the reference delegates to the GL
// driver per step: glClear, rasterisation/depth test of display list 1 and the entity
// draws, GL_MODULATE texturing, FrameBuffer.resolve()'s two blits + glReadPixels + flip
// (miniworld.py:1064-1086, 1193-1195; opengl.py:339-398) and get_depth_map (opengl.py:400-435).
//
// Mapping: one 64-lane wavefront (= one workgroup) owns a run of 16x4-pixel tiles of one
// env; lane l is pixel (l & 15, l >> 4) of the tile and keeps its 8 samples' packed keys
// (depth16 << 16 | draw index) in 8 VGPRs — no depth buffer in memory at all.
//   coverage : per (tile, primitive) the 64-dword raster record is wave-uniform and arrives
//              through scalar loads (SGPRs); a sample is inside edge k iff E_k(pixel centre)
//              > thr_k[s] (R4/R5: thresholds precomputed by K1 with the top-left rule folded
//              in), so coverage is 2 FMA + 8 v_cmp per edge with the mask algebra on the SALU.
//   depth    : GL_LESS with first-drawn-wins == unsigned min of the packed keys (R6).
//   shading  : deferred — each lane shades each *distinct* winning primitive of its pixel once
//              at the pixel centre (GL multisample semantics, R9), ascending draw index (R12).
//   output   : RGB bytes staged through 192 B of LDS so the tile leaves as dword stores.
// HBM traffic per env-step is the observation (14 400 B, + 19 200 B with depth) plus the
// K1 records; textures and records are L2-resident.
#include "mw_device.h"

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

// R8: GL_LINEAR fetch on one mip level, GL_REPEAT, texel centres at +0.5
__device__ inline RGB bilinear(const uint32_t *__restrict__ texels, uint32_t off, int w, int h, float u, float v)
{
    const float uu = u - floorf(u), vv = v - floorf(v);
    const float x = fmaf(uu, (float)w, -0.5f), y = fmaf(vv, (float)h, -0.5f);
    const float x0f = floorf(x), y0f = floorf(y);
    const float fx = x - x0f, fy = y - y0f;
    int i0 = (int)x0f, j0 = (int)y0f;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += w;
    if (i1 >= w) i1 -= w;
    if (j0 < 0) j0 += h;
    if (j1 >= h) j1 -= h;
    const uint32_t *base = texels + off;
    const uint32_t t00 = base[j0 * w + i0], t10 = base[j0 * w + i1];
    const uint32_t t01 = base[j1 * w + i0], t11 = base[j1 * w + i1];
    RGB o;
    {
        const float a = (float)(t00 & 255u), b = (float)(t10 & 255u), c = (float)(t01 & 255u), d = (float)(t11 & 255u);
        const float r0 = fmaf(fx, b - a, a), r1 = fmaf(fx, d - c, c);
        o.r = fmaf(fy, r1 - r0, r0);
    }
    {
        const float a = (float)((t00 >> 8) & 255u), b = (float)((t10 >> 8) & 255u);
        const float c = (float)((t01 >> 8) & 255u), d = (float)((t11 >> 8) & 255u);
        const float r0 = fmaf(fx, b - a, a), r1 = fmaf(fx, d - c, c);
        o.g = fmaf(fy, r1 - r0, r0);
    }
    {
        const float a = (float)((t00 >> 16) & 255u), b = (float)((t10 >> 16) & 255u);
        const float c = (float)((t01 >> 16) & 255u), d = (float)((t11 >> 16) & 255u);
        const float r0 = fmaf(fx, b - a, a), r1 = fmaf(fx, d - c, c);
        o.b = fmaf(fy, r1 - r0, r0);
    }
    return o;
}

__device__ inline int level_dim(int d, int l) { const int s = d >> l; return s > 0 ? s : 1; }

// fragment colour of primitive record `sr` at the pixel centre (R7-R9)
__device__ inline RGB shade(const float4 *sr, const MwTexDesc *__restrict__ texd,
                            const uint32_t *__restrict__ texels, float Xc, float Yc)
{
    const float4 q0 = sr[0], q1 = sr[1], q2 = sr[2], q3 = sr[3];
    const float Ua = q0.x, Ub = q0.y, Uc = q0.z, Va = q0.w, Vb = q1.x, Vc = q1.y;
    const float Wa = q1.z, Wb = q1.w, Wc = q2.x;
    RGB base = {q2.y, q2.z, q2.w};
    const int tex = __float_as_int(q3.x);
    if (tex < 0) return base;
    const MwTexDesc *t = texd + tex;
    const int tw = (int)t->w, th = (int)t->h, q = (int)t->nlevels - 1;
    const float Wq = fmaf(Wa, Xc, fmaf(Wb, Yc, Wc));
    RGB texel;
    if (!(Wq > 0.0f)) {
        texel = bilinear(texels, t->off[q], level_dim(tw, q), level_dim(th, q), 0.0f, 0.0f);
    } else {
        const float iw = 1.0f / Wq;
        const float Uq = fmaf(Ua, Xc, fmaf(Ub, Yc, Uc));
        const float Vq = fmaf(Va, Xc, fmaf(Vb, Yc, Vc));
        const float u = Uq * iw, v = Vq * iw;
        const float ux = (Ua - u * Wa) * iw, uy = (Ub - u * Wb) * iw;
        const float vx = (Va - v * Wa) * iw, vy = (Vb - v * Wb) * iw;
        const float ftw = (float)tw, fth = (float)th;
        const float sx = ux * ftw, tx = vx * fth, sy = uy * ftw, ty = vy * fth;
        const float r2x = fmaf(sx, sx, tx * tx), r2y = fmaf(sy, sy, ty * ty);
        const float rho2 = r2x > r2y ? r2x : r2y;
        if (!(rho2 > 1.0f)) {
            texel = bilinear(texels, t->off[0], tw, th, u, v);
        } else if (!(rho2 < 1e30f)) {
            texel = bilinear(texels, t->off[q], level_dim(tw, q), level_dim(th, q), u, v);
        } else {
            const float lam = 0.5f * lod_log2(rho2);
            const float lf = floorf(lam);
            const int l0 = (int)lf;
            if (l0 >= q) {
                texel = bilinear(texels, t->off[q], level_dim(tw, q), level_dim(th, q), u, v);
            } else {
                const float fr = lam - lf;
                const RGB c0 = bilinear(texels, t->off[l0], level_dim(tw, l0), level_dim(th, l0), u, v);
                const RGB c1 = bilinear(texels, t->off[l0 + 1], level_dim(tw, l0 + 1), level_dim(th, l0 + 1), u, v);
                texel.r = fmaf(fr, c1.r - c0.r, c0.r);
                texel.g = fmaf(fr, c1.g - c0.g, c0.g);
                texel.b = fmaf(fr, c1.b - c0.b, c0.b);
            }
        }
    }
    RGB o;
    o.r = (texel.r * (1.0f / 255.0f)) * base.r;
    o.g = (texel.g * (1.0f / 255.0f)) * base.g;
    o.b = (texel.b * (1.0f / 255.0f)) * base.b;
    return o;
}

__device__ inline uint32_t to_u8(float acc)     // R12: mean of 8, clamp, round half up
{
    float v = acc * 0.125f;
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return (uint32_t)(int)fmaf(v, 255.0f, 0.5f);
}

}  // namespace

extern "C" __global__ __launch_bounds__(64) void mw_raster_kernel(
    int N, int W, int H, int max_vis, int tiles_x, int n_tiles, int waves_per_env, int tiles_per_wave,
    const float *__restrict__ rec_raster, const float *__restrict__ rec_shade, const int32_t *__restrict__ nvis_arr,
    const float *__restrict__ envhdr, const MwTexDesc *__restrict__ texd, const uint32_t *__restrict__ texels,
    uint8_t *__restrict__ obs, float *__restrict__ depth)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *s_shade = reinterpret_cast<float4 *>(smem);                       // [max_vis][4]
    uint8_t *s_pack = smem + (size_t)max_vis * MW_SHADE_REC * 4;              // 192 B, 16-aligned

    // XCD-aware block -> (env, part): the parts of one env run on the same XCD (block b is
    // dispatched to XCD b % 8), adjacent in time, so its records are fetched into one L2 once.
    const int b = blockIdx.x;
    const int xcd = b & 7, slot = b >> 3;
    const int env = (slot / waves_per_env) * 8 + xcd;
    const int part = slot % waves_per_env;
    if (env >= N) return;
    const int lane = threadIdx.x;
    const int nvis = nvis_arr[env];
    const float *__restrict__ rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    {
        const float4 *src = reinterpret_cast<const float4 *>(rec_shade + (size_t)env * max_vis * MW_SHADE_REC);
        for (int i = lane; i < nvis * 4; i += 64) s_shade[i] = src[i];
    }
    __syncthreads();
    const float sky_r = envhdr[(size_t)env * 4 + 0], sky_g = envhdr[(size_t)env * 4 + 1],
                sky_b = envhdr[(size_t)env * 4 + 2];

    const int t_begin = part * tiles_per_wave;
    const int t_end = min(t_begin + tiles_per_wave, n_tiles);
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int px = tx * MW_TILE_W + (lane & 15), py = ty * MW_TILE_H + (lane >> 4);
        const float Xc = (float)px + 0.5f, Yc = (float)py + 0.5f;
        uint32_t key[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) key[s] = 0xFFFFFFFFu;

        for (int p = 0; p < nvis; ++p) {
            const float *__restrict__ rr = rr_env + (size_t)p * MW_RASTER_REC;
            const uint32_t bb = __float_as_uint(rr[15]);
            const int bx0 = bb & 255u, bx1 = (bb >> 8) & 255u, by0 = (bb >> 16) & 255u, by1 = bb >> 24;
            if (tx < bx0 || tx > bx1 || ty < by0 || ty > by1) continue;
            bool in[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) in[s] = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float E = fmaf(rr[k], Xc, fmaf(rr[4 + k], Yc, rr[8 + k]));
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
                const uint32_t k = ((uint32_t)t << 16) | (uint32_t)p;
                key[s] = ok ? min(key[s], k) : key[s];
            }
        }

        // ---- deferred shading + resolve (R9, R12) --------------------------------
        uint32_t pid[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) pid[s] = key[s] & 0xFFFFu;
        float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
        for (;;) {
            uint32_t sel = min(min(min(pid[0], pid[1]), min(pid[2], pid[3])), min(min(pid[4], pid[5]), min(pid[6], pid[7])));
            const bool active = sel != 0x10000u;
            if (!__any(active)) break;
            if (active) {
                uint32_t cnt = 0;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const bool eq = pid[s] == sel;
                    cnt += eq ? 1u : 0u;
                    pid[s] = eq ? 0x10000u : pid[s];
                }
                RGB c;
                if (sel == MW_SKY_PID) {
                    c.r = sky_r; c.g = sky_g; c.b = sky_b;
                } else {
                    c = shade(s_shade + sel * 4, texd, texels, Xc, Yc);
                }
                const float fc = (float)cnt;
                acc_r = fmaf(fc, c.r, acc_r);
                acc_g = fmaf(fc, c.g, acc_g);
                acc_b = fmaf(fc, c.b, acc_b);
            }
        }
        const uint32_t R = to_u8(acc_r), G = to_u8(acc_g), B = to_u8(acc_b);

        // ---- pack: tile rows of 16 px * 3 B = 48 B = 12 dwords; 4 rows -> 48 dword stores
        const int row = lane >> 4, col = lane & 15;
        s_pack[row * 48 + col * 3 + 0] = (uint8_t)R;
        s_pack[row * 48 + col * 3 + 1] = (uint8_t)G;
        s_pack[row * 48 + col * 3 + 2] = (uint8_t)B;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        if (lane < 48) {
            const int r = lane / 12, d = lane % 12;
            const uint32_t w = reinterpret_cast<const uint32_t *>(s_pack)[r * 12 + d];
            uint8_t *dst = obs + ((size_t)env * H + (ty * MW_TILE_H + r)) * W * 3 + (size_t)tx * (MW_TILE_W * 3) + d * 4;
            *reinterpret_cast<uint32_t *>(dst) = w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (depth) {
            // R13 / R14: resolved depth = sample 0; get_depth_map in float32 as numpy evaluates it
            const float z = (float)(key[0] >> 16);
            const float d = z / 65535.0f;
            const float clip = (d - 0.5f) * 2.0f;
            const float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
            depth[((size_t)env * H + py) * W + px] = (float)(-2.0 * 100.0 * 0.04) / den;
        }
    }
}
