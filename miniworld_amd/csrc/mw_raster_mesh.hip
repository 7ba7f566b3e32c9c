// K3 — raster kernel for envs that contain mesh entities (Ball / Key: entity.py:124-165, 435-452;
// ObjMesh.render objmesh.py:280-292).  One 1024-thread workgroup per env.
//
// A ball is 5 192 sub-pixel triangles, so the pixel-per-lane scheme of K2 would waste 60+
// lanes per triangle.  Here the whole env's sample-key buffer (80*60*8 dwords = 153 600 B)
// lives in the CU's 160 KiB LDS and the mesh triangles are rasterised one per lane, scattering
// packed keys (depth16 << 16 | draw id) with ds_min_u32 — GL_LESS / first-drawn-wins is an
// unsigned min, hence order independent (R6).  After a barrier the 16 wavefronts walk the
// tiles exactly like K2, starting from the mesh keys; winners whose draw id falls into a mesh
// entity's range are shaded by re-deriving that triangle (Gouraud, R11).
//
// Per-triangle arithmetic (transform, unnormalised-normal lighting, edge / depth / colour
// planes) follows DESIGN.md R3, R4, R6, R10, R11 exactly like the oracle's mesh path.
#include "mw_raster_common_old.h"

namespace {

struct HV { float hx, hy, hw, cz; };

struct MeshEnt {            // one entry of the env header's mesh table (written by K1)
    int slot, start, ntris, first, tex;
    float c, s, scale, px, py, pz;
};

__device__ inline MeshEnt load_ment(const float *table, int j)
{
    const float *m = table + 12 * j;
    MeshEnt e;
    e.slot = __float_as_int(m[0]); e.start = __float_as_int(m[1]); e.ntris = __float_as_int(m[2]);
    e.first = __float_as_int(m[3]);
    e.c = m[4]; e.s = m[5]; e.scale = m[6]; e.px = m[7]; e.py = m[8]; e.pz = m[9];
    e.tex = __float_as_int(m[10]);
    return e;
}

__device__ inline HV xform_hdr(const float *hdr, float halfw, float halfh, float x, float y, float z)
{
    const float *m = hdr + 4;
    const float ex = fmaf(m[0], x, fmaf(m[1], y, fmaf(m[2], z, m[3])));
    const float ey = fmaf(m[4], x, fmaf(m[5], y, fmaf(m[6], z, m[7])));
    const float ez = fmaf(m[8], x, fmaf(m[9], y, fmaf(m[10], z, m[11])));
    float cx = hdr[16] * ex, cy = hdr[17] * ey, cw = -ez;
    if (__float_as_int(hdr[23])) {      // top view: glOrtho (translation terms, w = 1)
        cx = fmaf(hdr[16], ex, hdr[27]);
        cy = fmaf(hdr[17], ey, hdr[31]);
        cw = 1.0f;
    }
    HV h;
    h.cz = fmaf(hdr[18], ez, hdr[19]);
    h.hx = (cx + cw) * halfw;
    h.hy = (cw - cy) * halfh;
    h.hw = cw;
    return h;
}

__device__ inline void edge_coef(const HV &a, const HV &b, float &ea, float &eb, float &ec)
{
    ea = b.hy * a.hw - b.hw * a.hy;
    eb = b.hw * a.hx - b.hx * a.hw;
    ec = b.hx * a.hy - b.hy * a.hx;
}

__device__ inline float below(float x)
{
    if (x == 0.0f) return __uint_as_float(0x80000001u);
    const uint32_t b = __float_as_uint(x);
    return __uint_as_float(x > 0.0f ? b - 1u : b + 1u);
}

__constant__ float kDx[8] = {0.0625f, -0.0625f, 0.3125f, -0.1875f, -0.3125f, -0.4375f, 0.1875f, 0.4375f};
__constant__ float kDy[8] = {-0.1875f, 0.1875f, 0.0625f, -0.3125f, 0.3125f, -0.0625f, 0.4375f, -0.4375f};
// R5 for the 16-sample visualisation buffer (FrameBuffer(800, 600, 16), miniworld.py:518): the
// D3D standard 16x pattern (9,9)(7,5)(5,10)(12,7)(3,6)(10,13)(13,11)(11,3)(6,14)(8,1)(4,2)(2,12)(0,8)(15,4)(14,15)(1,0)
__constant__ float kDx16[16] = {0.0625f, -0.0625f, -0.1875f, 0.25f, -0.3125f, 0.125f, 0.3125f, 0.1875f,
                                -0.125f, 0.0f, -0.25f, -0.375f, -0.5f, 0.4375f, 0.375f, -0.4375f};
__constant__ float kDy16[16] = {0.0625f, -0.1875f, 0.125f, -0.0625f, -0.125f, 0.3125f, 0.1875f, -0.3125f,
                                0.375f, -0.4375f, -0.375f, 0.25f, 0.0f, -0.25f, 0.4375f, -0.5f};
// the fallback sample counts of FrameBuffer (opengl.py:229-231: a driver that clamps GL_MAX_SAMPLES): 4 = the D3D
// standard 4x pattern (6,2)(14,6)(2,10)(10,14), 1 = the pixel centre
__constant__ float kDx4[4] = {-0.125f, 0.375f, -0.375f, 0.125f};
__constant__ float kDy4[4] = {-0.375f, -0.125f, 0.125f, 0.375f};
template <int S> __device__ inline float sample_dx(int s) { return S == 16 ? kDx16[s] : (S == 4 ? kDx4[s] : (S == 1 ? 0.0f : kDx[s])); }
template <int S> __device__ inline float sample_dy(int s) { return S == 16 ? kDy16[s] : (S == 4 ? kDy4[s] : (S == 1 ? 0.0f : kDy[s])); }

// the i-th triangle of the rasterisation order (sorted by face-normal direction; mw_device.h: MW_MESH_POS_STRIDE)
__device__ inline int tri_sorted(const TileCtx &cx, const MeshEnt &e, int i)
{
    return (int)__float_as_uint(cx.mesh_pos[(size_t)(e.first + i) * MW_MESH_POS_STRIDE + 9]);
}

// object-space vertices of triangle `tri` (drawing order) of mesh entity e
__device__ inline void tri_load(const TileCtx &cx, const MeshEnt &e, int tri, float (&p)[9])
{
    // (a triangle's 10 words are 8-byte aligned: four 8-byte loads and one 4-byte load instead of nine)
    static_assert(MW_MESH_POS_STRIDE % 2 == 0, "8-byte aligned triangles");
    const float2 *src = reinterpret_cast<const float2 *>(cx.mesh_pos + (size_t)(e.first + tri) * MW_MESH_POS_STRIDE);
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float2 v = src[k]; p[2 * k] = v.x; p[2 * k + 1] = v.y; }
    p[8] = reinterpret_cast<const float *>(src)[8];
}

// homogeneous image-space vertices from the object-space ones (R11: pos + scale * R_y(dir) * v)
__device__ inline void tri_verts(const TileCtx &cx, const MeshEnt &e, const float (&p)[9], float halfw, float halfh, HV h[3])
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float lx = p[k * 3 + 0], ly = p[k * 3 + 1], lz = p[k * 3 + 2];
        const float rx = fmaf(e.c, lx, e.s * lz), rz = fmaf(e.c, lz, -(e.s * lx));
        const float wx = fmaf(e.scale, rx, e.px), wy = fmaf(e.scale, ly, e.py), wz = fmaf(e.scale, rz, e.pz);
        h[k] = xform_hdr(cx.hdr, halfw, halfh, wx, wy, wz);
    }
}

// Mesh triangle (e, tri) at the pixel centre (R9-R11): Gouraud colour in q2.yzw and, for a textured
// mesh (objmesh.py:209-216), the texcoord / 1/w planes in q0, q1, q2.x for apply_texture
__device__ inline void mesh_tri_fragment(const TileCtx &cx, const MeshEnt &e, int tri, float Xc, float Yc,
                                         float4 &q0, float4 &q1, float4 &q2)
{
    const float halfw = (float)cx.W * 0.5f, halfh = (float)cx.H * 0.5f;
    HV h[3];
    float pos[9];
    tri_load(cx, e, tri, pos);
    tri_verts(cx, e, pos, halfw, halfh, h);
    float ga[3], gb[3], gc[3];
    edge_coef(h[1], h[2], ga[0], gb[0], gc[0]);
    edge_coef(h[2], h[0], ga[1], gb[1], gc[1]);
    edge_coef(h[0], h[1], ga[2], gb[2], gc[2]);
    const float *L = cx.hdr + 20, *amb = cx.hdr + 24, *lcol = cx.hdr + 28;
    const float *nrm = cx.mesh_nrm + (size_t)(e.first + tri) * 9, *rgb = cx.mesh_rgb + (size_t)(e.first + tri) * 9;
    float col[3][3];
    // R11: normal through the inverse transpose, R_y n / scale, three exact quotients per vertex by one divisor
    const bool fast_div = div_domain(e.scale);
    const float inv_scale = fast_div ? rcp_exact(e.scale) : 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float nx0 = nrm[k * 3 + 0], ny0 = nrm[k * 3 + 1], nz0 = nrm[k * 3 + 2];
        const float rnx = fmaf(e.c, nx0, e.s * nz0), rnz = fmaf(e.c, nz0, -(e.s * nx0));
        float n[3];
        if (fast_div) {
            n[0] = div_exact(rnx, inv_scale, e.scale); n[1] = div_exact(ny0, inv_scale, e.scale); n[2] = div_exact(rnz, inv_scale, e.scale);
        } else {
            n[0] = rnx / e.scale; n[1] = ny0 / e.scale; n[2] = rnz / e.scale;
        }
        const float ndl = fmaf(n[2], L[2], fmaf(n[1], L[1], n[0] * L[0]));
        const float d = ndl > 0.0f ? ndl : 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float kk = fmaf(lcol[i], d, amb[i]);
            const float v = rgb[k * 3 + i] * kk;
            col[k][i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        }
    }
    const float Wa = (ga[0] + ga[1]) + ga[2], Wb = (gb[0] + gb[1]) + gb[2], Wc = (gc[0] + gc[1]) + gc[2];
    const float Wq = fmaf(Wa, Xc, fmaf(Wb, Yc, Wc));
    float out[3] = {col[0][0], col[0][1], col[0][2]};
    if (rcp_domain(Wq)) {
        const float iw = rcp_exact(Wq);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float Ca = fmaf(col[2][i], ga[2], fmaf(col[1][i], ga[1], col[0][i] * ga[0]));
            const float Cb = fmaf(col[2][i], gb[2], fmaf(col[1][i], gb[1], col[0][i] * gb[0]));
            const float Cc = fmaf(col[2][i], gc[2], fmaf(col[1][i], gc[1], col[0][i] * gc[0]));
            out[i] = fmaf(Ca, Xc, fmaf(Cb, Yc, Cc)) * iw;
        }
    }
    float U[3] = {0.0f, 0.0f, 0.0f}, V[3] = {0.0f, 0.0f, 0.0f};
    if (e.tex >= 0) {
        const float *uv = cx.mesh_uv + (size_t)(e.first + tri) * 6;
        U[0] = fmaf(uv[4], ga[2], fmaf(uv[2], ga[1], uv[0] * ga[0]));
        U[1] = fmaf(uv[4], gb[2], fmaf(uv[2], gb[1], uv[0] * gb[0]));
        U[2] = fmaf(uv[4], gc[2], fmaf(uv[2], gc[1], uv[0] * gc[0]));
        V[0] = fmaf(uv[5], ga[2], fmaf(uv[3], ga[1], uv[1] * ga[0]));
        V[1] = fmaf(uv[5], gb[2], fmaf(uv[3], gb[1], uv[1] * gb[0]));
        V[2] = fmaf(uv[5], gc[2], fmaf(uv[3], gc[1], uv[1] * gc[0]));
    }
    q0 = make_float4(U[0], U[1], U[2], V[0]);
    q1 = make_float4(V[1], V[2], Wa, Wb);
    q2 = make_float4(Wc, out[0], out[1], out[2]);
}

// draw id -> fragment colour: ids inside a mesh entity's range are triangles, the others index the
// visible-primitive list once the triangles drawn before them are subtracted
__device__ inline RGB shade_by_draw_id(const TileCtx &cx, uint32_t id, float Xc, float Yc)
{
    const int n_mesh = __float_as_int(cx.hdr[3]);
    int vis = (int)id;
    float4 q0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), q1 = q0, q2 = q0;
    int tex = -1;
    bool is_tri = false;
    for (int j = 0; j < n_mesh; ++j) {
        const int start = __float_as_int(cx.ment[12 * j + 1]);
        const int nt = __float_as_int(cx.ment[12 * j + 2]);
        if ((int)id >= start + nt) {
            vis -= nt;
        } else if ((int)id >= start) {
            const MeshEnt e = load_ment(cx.ment, j);
            mesh_tri_fragment(cx, e, (int)id - start, Xc, Yc, q0, q1, q2);
            tex = e.tex;
            is_tri = true;
            break;
        }
    }
    if (!is_tri) {
        const float4 *sr = cx.s_shade + vis * (MW_SHADE_REC / 4);
        q0 = sr[0]; q1 = sr[1]; q2 = sr[2];
        tex = __float_as_int(sr[3].x);
    }
    if (cx.te.flat) tex = -1;
    if (!__any(tex >= 0)) return RGB{q2.y, q2.z, q2.w};
    return apply_texture(q0, q1, q2, tex, cx.te, Xc, Yc);
}

// rasterise one mesh triangle into the LDS key buffer (one lane per triangle)
template <int S>
__device__ inline void raster_tri(const TileCtx &cx, const MeshEnt &e, int tri, const float (&pos)[9], uint32_t *keys)
{
    const int W = cx.W, H = cx.H;
    const float halfw = (float)W * 0.5f, halfh = (float)H * 0.5f;
    HV h[3];
    tri_verts(cx, e, pos, halfw, halfh, h);
    float ga[3], gb[3], gc[3];
    edge_coef(h[1], h[2], ga[0], gb[0], gc[0]);
    edge_coef(h[2], h[0], ga[1], gb[1], gc[1]);
    edge_coef(h[0], h[1], ga[2], gb[2], gc[2]);
    const float D = fmaf(h[0].hx, ga[0], fmaf(h[0].hy, gb[0], h[0].hw * gc[0]));
    if (!(D > 0.0f)) return;                                  // back face (R4)
    // Pixel bounds (R4m).  A mesh triangle with all three vertices in front of the eye is rasterised
    // inside the pixel bounding box of its projected vertices, floor(min) .. floor(max) — part of the
    // pinned semantics (the oracle computes the same box from the same divisions), not only an
    // optimisation: a near-degenerate sliver's edge functions are rounding noise and would otherwise
    // claim samples away from it.  A triangle reaching behind the eye is bounded loosely instead, by the
    // part in front of w = 0.01 plus a pixel (coverage itself never clips, R4).
    int x0, y0, x1, y1;
    if (div_domain(h[0].hw) && div_domain(h[1].hw) && div_domain(h[2].hw)) {     // R4m: every w in [1e-10, 1e10]
        float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float iw = rcp_exact(h[k].hw);            // the oracle's hx / hw, hy / hw (exact quotients, mw_raster_common.h)
            const float X = div_exact(h[k].hx, iw, h[k].hw), Y = div_exact(h[k].hy, iw, h[k].hw);
            xmin = fminf(xmin, X); xmax = fmaxf(xmax, X); ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
        }
        const float fx0 = floorf(xmin), fx1 = floorf(xmax), fy0 = floorf(ymin), fy1 = floorf(ymax);
        if (!(fx1 >= 0.0f && fy1 >= 0.0f && fx0 <= (float)(W - 1) && fy0 <= (float)(H - 1))) return;     // off screen (or NaN)
        x0 = (int)fmaxf(fx0, 0.0f); x1 = (int)fminf(fx1, (float)(W - 1));
        y0 = (int)fmaxf(fy0, 0.0f); y1 = (int)fminf(fy1, (float)(H - 1));
    } else {
        const float wc = 0.01f;
        float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
        bool some = false;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const HV p = h[k], q = h[k == 2 ? 0 : k + 1];
            const bool pin = p.hw >= wc, qin = q.hw >= wc;
            if (pin) {
                const float X = p.hx / p.hw, Y = p.hy / p.hw;
                xmin = fminf(xmin, X); xmax = fmaxf(xmax, X); ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
                some = true;
            }
            if (pin != qin) {
                const float t = (wc - p.hw) / (q.hw - p.hw);
                const float X = fmaf(t, q.hx - p.hx, p.hx) / wc, Y = fmaf(t, q.hy - p.hy, p.hy) / wc;
                xmin = fminf(xmin, X); xmax = fmaxf(xmax, X); ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
                some = true;
            }
        }
        if (!some) return;                                    // entirely behind the eye
        const float mx = 1.0f + 1e-4f * fmaxf(fabsf(xmin), fabsf(xmax)), my = 1.0f + 1e-4f * fmaxf(fabsf(ymin), fabsf(ymax));
        if (xmax + mx < 0.0f || ymax + my < 0.0f || xmin - mx > (float)W || ymin - my > (float)H) return;
        x0 = (int)fminf(fmaxf(floorf(xmin - mx), 0.0f), (float)(W - 1));
        x1 = (int)fminf(fmaxf(floorf(xmax + mx), 0.0f), (float)(W - 1));
        y0 = (int)fminf(fmaxf(floorf(ymin - my), 0.0f), (float)(H - 1));
        y1 = (int)fminf(fmaxf(floorf(ymax + my), 0.0f), (float)(H - 1));
    }
    // edges 0->1, 1->2, 2->0 in drawing order: coefficients = G2, G0, G1
    const float ea[3] = {ga[2], ga[0], ga[1]}, eb[3] = {gb[2], gb[0], gb[1]}, ec[3] = {gc[2], gc[0], gc[1]};
    float thr[3][S];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const bool tl = (ea[k] > 0.0f) || (ea[k] == 0.0f && eb[k] > 0.0f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float t = -fmaf(ea[k], sample_dx<S>(s), eb[k] * sample_dy<S>(s));
            thr[k][s] = tl ? below(t) : t;
        }
    }
    const float invD = rcp_domain(D) ? rcp_exact(D) : 1.0f / D;
    const float ta = fmaf(h[2].cz, ga[2], fmaf(h[1].cz, ga[1], h[0].cz * ga[0]));
    const float tb = fmaf(h[2].cz, gb[2], fmaf(h[1].cz, gb[1], h[0].cz * gb[0]));
    const float tc = fmaf(h[2].cz, gc[2], fmaf(h[1].cz, gc[1], h[0].cz * gc[0]));
    const float zx = (ta * invD) * 0.5f, zy = (tb * invD) * 0.5f, zcc = fmaf(tc * invD, 0.5f, 0.5f);
    float zo[S];
#pragma unroll
    for (int s = 0; s < S; ++s) zo[s] = fmaf(zx, sample_dx<S>(s), zy * sample_dy<S>(s));
    const uint32_t id = (uint32_t)(e.start + tri);      // draw id = position in drawing order
    for (int py = y0; py <= y1; ++py)
        for (int px = x0; px <= x1; ++px) {
            const float Xc = (float)px + 0.5f, Yc = (float)py + 0.5f;
            const float E0 = fmaf(ea[0], Xc, fmaf(eb[0], Yc, ec[0]));
            const float E1 = fmaf(ea[1], Xc, fmaf(eb[1], Yc, ec[1]));
            const float E2 = fmaf(ea[2], Xc, fmaf(eb[2], Yc, ec[2]));
            const float zc = fmaf(zx, Xc, fmaf(zy, Yc, zcc));
            uint32_t *kp = keys + ((size_t)py * W + px) * S;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (E0 > thr[0][s] && E1 > thr[1][s] && E2 > thr[2][s]) {
                    const float t = fmaf(zc + zo[s], 65535.0f, 0.5f);
                    if (t >= 0.5f && t < 65536.0f) atomicMin(kp + s, ((uint32_t)t << 16) | id);
                }
            }
        }
}

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

// FMT / HOT as in mw_raster.hip: the production kernels (plain observation layout, RGB or RGB-D) carry no debug flags
// and no run-time depth / layout switches in their tile loop; the general one reads all of that from `dbg`.
template <int FMT, int HOT>
__device__ inline void mesh_kernel_body(MW_MESH_ARGS)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t *keys = reinterpret_cast<uint32_t *>(smem);                     // [H][W][8]
    const unsigned long long t_start = prof ? __builtin_readcyclecounter() : 0ull;
    const int nkeys = W * H * 8;
    // block b draws the b-th env in order of decreasing mesh work (mw_mesh_order_kernel): blocks are dispatched
    // roughly in index order, one per CU at a time, and their costs are heavy-tailed, so the heavy envs must start
    // first and the light ones fill the gaps at the end
    const int env = env_order ? env_order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint8_t *s_pack = smem + (size_t)nkeys * 4 + wave * MW_K3_WAVE_LDS;
    // co-run mode (flag 16): the envs without a mesh in view are being drawn at the same time by
    // mw_raster_big_kernel on a second stream (they come last in the block order, so these blocks retire at once)
    if ((dbg & 16) && __float_as_int(envhdr[(size_t)env * MW_ENVHDR + 3]) == 0) return;
    const float *hdr = envhdr + (size_t)env * MW_ENVHDR;
    {
        uint4 *k4 = reinterpret_cast<uint4 *>(keys);
        for (int i = tid; i < nkeys / 4; i += 1024) k4[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        if (tid == 0) *reinterpret_cast<int *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS) = 0;       // the tile counter of phase 2
        float *s_ment = reinterpret_cast<float *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS + 16);       // [MW_MAX_MESH_ENTS][12]
        if (tid < MW_MAX_MESH_ENTS * 12) s_ment[tid] = hdr[MW_HDR_MESH + tid];
    }
    __syncthreads();

    TileCtx cx;
    cx.s_shade = reinterpret_cast<const float4 *>(rec_shade + (size_t)env * max_vis * MW_SHADE_REC);
    cx.s_cull = reinterpret_cast<const float4 *>(rec_cull + (size_t)env * max_vis * MW_CULL_REC);
    cx.rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    cx.s_pack = s_pack;
    cx.hdr = hdr;
    cx.ment = reinterpret_cast<const float *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS + 16);
    cx.mesh_pos = mesh_pos; cx.mesh_nrm = mesh_nrm; cx.mesh_rgb = mesh_rgb; cx.mesh_uv = mesh_uv;
    cx.obs = obs; cx.depth = depth;
    cx.obs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(obs + (size_t)env * H * W * 3), 0, H * W * 3, MW_RSRC_WORD3);
    cx.te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    cx.te.td = cx.te.tx;    // the descriptor table is the head of the texel block (upload_textures)
    cx.te.texd = texd;
    cx.te.flat = HOT ? 0 : (dbg & 1);
    cx.sky_r = hdr[0]; cx.sky_g = hdr[1]; cx.sky_b = hdr[2];
    cx.env = env; cx.nvis = nvis_arr[env]; cx.W = W; cx.H = H; cx.dbg = dbg; cx.lane = lane; cx.have_pre = 0; cx.order = nullptr;
    cx.tprof = prof ? prof + (size_t)N * 4 + (size_t)env * 4 : nullptr;        // MW_K3_PROF: second half of the buffer

    // ---- phase 1: every mesh triangle -> LDS keys, one triangle per lane -------------------
    const int n_mesh = (!HOT && (dbg & 8)) ? 0 : __float_as_int(hdr[3]);       // MW_DEBUG_FLAGS bit 3: perf experiments only
    for (int j = 0; j < n_mesh; ++j) {
        const MeshEnt e = load_ment(cx.ment, j);
        // two loads deep: the index of the triangle after next and the vertices of the next one are in flight while this
        // one is rasterised (the kernel waits more than it computes: 4 waves per SIMD, two dependent gathers per triangle)
        int t = tid;
        int tri = t < e.ntris ? tri_sorted(cx, e, t) : 0;
        int tri_n = t + 1024 < e.ntris ? tri_sorted(cx, e, t + 1024) : 0;
        float pos[9];
        tri_load(cx, e, tri, pos);
        while (t < e.ntris) {
            const int tri_nn = t + 2048 < e.ntris ? tri_sorted(cx, e, t + 2048) : 0;
            float pos_n[9];
            tri_load(cx, e, tri_n, pos_n);
            raster_tri<8>(cx, e, tri, pos, keys);
            t += 1024; tri = tri_n; tri_n = tri_nn;
#pragma unroll
            for (int k = 0; k < 9; ++k) pos[k] = pos_n[k];
        }
    }
    __syncthreads();
    const unsigned long long t_mesh = prof ? __builtin_readcyclecounter() : 0ull;

    // ---- phase 2: tiles, taken by the 16 wavefronts from a shared counter --------------------------------------
    // (a tile under a ball costs several times a plain one: with a static round-robin the waves that drew the mesh
    // tiles decide the workgroup's duration while the others idle — and nothing else fits on the CU beside its LDS)
    int *s_next = reinterpret_cast<int *>(smem + (size_t)nkeys * 4 + 16 * MW_K3_WAVE_LDS);
    for (;;) {
        int tile = 0;
        if (lane == 0) tile = atomicAdd(s_next, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile >= n_tiles) break;
        const int tx = tile % tiles_x, ty = tile / tiles_x;
        const int px = tx * MW_TILE_W + (lane & 15), py = ty * MW_TILE_H + (lane >> 4);
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
            for (int j = 0; j < n_mesh; ++j) nt += (unsigned long long)__float_as_int(hdr[MW_HDR_MESH + 12 * j + 2]);
            prof[(size_t)env * 4 + 3] = nt;
        }
    }
}

extern "C" __global__ __launch_bounds__(1024) void mw_raster_mesh_kernel(MW_MESH_ARGS) { mesh_kernel_body<0, 1>(MW_MESH_FWD); }
extern "C" __global__ __launch_bounds__(1024) void mw_raster_mesh_depth_kernel(MW_MESH_ARGS) { mesh_kernel_body<0, 2>(MW_MESH_FWD); }
extern "C" __global__ __launch_bounds__(1024) void mw_raster_mesh_wrap_kernel(MW_MESH_ARGS) { mesh_kernel_body<-1, 0>(MW_MESH_FWD); }

// ======================================================================================
// Generic-resolution path: render()/vis_fb 800x600x16 (miniworld.py:518, 1340-1362) and any other
// frame buffer size / sample count.  One env at a time, exact packed-key resolution only; the mesh
// keys go through a global buffer (the image no longer fits LDS).  Not the hot path.
// ======================================================================================
template <int S>
__device__ inline void view_mesh_body(int W, int H, const float *hdr, const float *mesh_pos, uint32_t *keys)
{
    TileCtx cx{};
    cx.tprof = nullptr;
    cx.hdr = hdr; cx.ment = hdr + MW_HDR_MESH; cx.mesh_pos = mesh_pos; cx.W = W; cx.H = H;
    const int n_mesh = __float_as_int(hdr[3]);
    const int stride = gridDim.x * blockDim.x;
    for (int j = 0; j < n_mesh; ++j) {
        const MeshEnt e = load_ment(cx.ment, j);
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < e.ntris; t += stride) {
            const int tri = tri_sorted(cx, e, t);
            float pos[9];
            tri_load(cx, e, tri, pos);
            raster_tri<S>(cx, e, tri, pos, keys);
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
    const int lane = cx.lane, W = cx.W, nvis = cx.nvis;
    const int tile = blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * MW_TILE_W + (lane & 15), py = ty * MW_TILE_H + (lane >> 4);
    const float Xc = (float)px + 0.5f, Yc = (float)py + 0.5f;
    uint32_t key[S];
#pragma unroll
    for (int s = 0; s < S; ++s) key[s] = mesh_keys ? mesh_keys[((size_t)py * W + px) * S + s] : 0xFFFFFFFFu;
    for (int p = 0; p < nvis; ++p) {
        const float *__restrict__ rr = cx.rr_env + (size_t)p * MW_RASTER_REC;
        const uint32_t bb = __float_as_uint(rr[15]);
        const int bx0 = bb & 255u, bx1 = (bb >> 8) & 255u, by0 = (bb >> 16) & 255u, by1 = bb >> 24;
        if (tx < bx0 || tx > bx1 || ty < by0 || ty > by1) continue;
        float E[4];
        bool tl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            E[k] = fmaf(rr[k], Xc, fmaf(rr[4 + k], Yc, rr[8 + k]));
            tl[k] = (rr[k] > 0.0f) || (rr[k] == 0.0f && rr[4 + k] > 0.0f);
        }
        const bool tri = rr[3] == 0.0f && rr[7] == 0.0f && rr[11] == 1.0f;      // K1's always-true 4th edge
        const float zc = fmaf(rr[12], Xc, fmaf(rr[13], Yc, rr[14]));
        const uint32_t id = __float_as_uint(rr[61]);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float dx = sample_dx<S>(s), dy = sample_dy<S>(s);
            bool in = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = -fmaf(rr[k], dx, rr[4 + k] * dy);
                const float thr = (tl[k] && !(k == 3 && tri)) ? below(t) : t;
                in &= E[k] > thr;
            }
            const float zs = zc + fmaf(rr[12], dx, rr[13] * dy);
            const float t = fmaf(zs, 65535.0f, 0.5f);
            const bool ok = in && t >= 0.5f && t < 65536.0f;
            const uint32_t k = ((uint32_t)t << 16) | id;
            key[s] = ok ? min(key[s], k) : key[s];
        }
    }
    const uint32_t z16 = key[0] >> 16;
    uint32_t pid[S];
#pragma unroll
    for (int s = 0; s < S; ++s) pid[s] = key[s] & 0xFFFFu;
    float acc_r = 0.0f, acc_g = 0.0f, acc_b = 0.0f;
    for (;;) {
        uint32_t sel = 0x10000u;
#pragma unroll
        for (int s = 0; s < S; ++s) sel = min(sel, pid[s]);
        const bool active = sel != 0x10000u;
        if (!__any(active)) break;
        if (active) {
            uint32_t cnt = 0;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bool eq = pid[s] == sel;
                cnt += eq ? 1u : 0u;
                pid[s] = eq ? 0x10000u : pid[s];
            }
            RGB c = {cx.sky_r, cx.sky_g, cx.sky_b};
            if (sel != MW_SKY_PID) c = shade_by_draw_id(cx, sel, Xc, Yc);
            const float fc = (float)cnt;
            acc_r = fmaf(fc, c.r, acc_r);
            acc_g = fmaf(fc, c.g, acc_g);
            acc_b = fmaf(fc, c.b, acc_b);
        }
    }
    const float inv = 1.0f / (float)S;
    float v[3] = {acc_r * inv, acc_g * inv, acc_b * inv};
    uint8_t *dst = cx.obs + ((size_t)py * W + px) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = v[c] < 0.0f ? 0.0f : (v[c] > 1.0f ? 1.0f : v[c]);
        dst[c] = (uint8_t)(int)fmaf(x, 255.0f, 0.5f);
    }
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
    const float *__restrict__ rec_shade, const int32_t *__restrict__ nvis_arr, const float *__restrict__ envhdr,
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
    cx.s_cull = nullptr;
    cx.rr_env = rec_raster + (size_t)env * max_vis * MW_RASTER_REC;
    cx.s_pack = nullptr;
    cx.hdr = hdr; cx.ment = hdr + MW_HDR_MESH;
    cx.mesh_pos = mesh_pos; cx.mesh_nrm = mesh_nrm; cx.mesh_rgb = mesh_rgb; cx.mesh_uv = mesh_uv;
    cx.obs = out; cx.depth = depth;
    cx.obs_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, H * W * 3, MW_RSRC_WORD3);
    cx.te.tx = __builtin_amdgcn_make_buffer_rsrc((void *)texels, 0, texel_bytes, MW_RSRC_WORD3);
    cx.te.td = cx.te.tx;    // the descriptor table is the head of the texel block (upload_textures)
    cx.te.texd = texd;
    cx.te.flat = 0;
    cx.sky_r = hdr[0]; cx.sky_g = hdr[1]; cx.sky_b = hdr[2];
    cx.env = 0; cx.nvis = nvis_arr[env]; cx.W = W; cx.H = H; cx.dbg = 0; cx.lane = threadIdx.x; cx.have_pre = 0; cx.order = nullptr; cx.tprof = nullptr;
    if (S == 16) view_tile_body<16>(cx, tiles_x, mesh_keys);
    else if (S == 4) view_tile_body<4>(cx, tiles_x, mesh_keys);
    else if (S == 1) view_tile_body<1>(cx, tiles_x, mesh_keys);
    else view_tile_body<8>(cx, tiles_x, mesh_keys);
}

// Order in which mw_raster_mesh_kernel's blocks take the envs: a counting sort of the envs by mesh triangles in
// view (K1's k3_cost), most first; one workgroup, bins of 2048 triangles.  The order inside a bin is whatever the
// LDS atomics produce — no output depends on which block draws which env.
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
