// KG — the vertex half of a frame: camera, lighting, transform, clipping, triangle setup.
//
// Replaces, per env and per frame (reference file:line), what the reference hands to OpenGL and what the driver does with
// it before a single sample is touched:
//   render_obs / render_top_view          miniworld.py:1088-1221   gluPerspective, gluLookAt / glOrtho, glLoadMatrixf
//   _render_static, Room._render          miniworld.py:401-434, 1019-1062   light, colour material, display list 1
//   _render_world                         miniworld.py:1064-1086   draw order: list 1 (rooms, static entities), dynamic entities
//   Box.render / drawBox                  entity.py:409-432, opengl.py:460-503
//   ImageFrame / TextFrame.render         entity.py:193-259, 303-383   (quads of the static polygon list with their own transform)
//   MeshEnt.render                        entity.py:150-161   (described to the mesh kernel: transform, light, draw-id range)
//   Agent.render                          entity.py:518-539   (top view's marker, lit by the stale current normal)
//   get_visible_ents' proxy boxes         miniworld.py:1291-1313
// with the arithmetic of mw_glmath.h (Mesa 23.2.1 / llvmpipe, measured).  Output: per env the list of the triangles that
// leave llvmpipe's setup, in drawing order and without gaps (mw_records.h), the env header (sky colour, mesh-entity
// table), and the entity removals the step left pending (a picked-up object is still drawn in the frame of the step that
// picked it up: pickupobjects.py:86-88 runs after :717).
//
// Lanes: a wavefront serves 64 / L envs, L lanes each (L: the power of two that holds an env's triangles — two per polygon
// and box face, the agent marker — or 64, with several rounds).  One triangle per lane and round.
// A round runs twice over its triangles: pass 1 counts what survives clipping and culling (the snapped-area test only), a
// segmented scan turns the counts into list positions, pass 2 sets the survivors up and writes the records.  A triangle
// that crosses a frustum plane goes through one of kClipSlots work lists in LDS, eight lanes to a list: an edge of the
// polygon per lane and one step per plane, then a triangle of the fan per lane (setup, record).
// Big scenes (one env per wavefront): the polygons are sifted first — boxes of eight polygons, then the polygons, against
// the frustum, full-height walls in front of them and their own facing — from data kept per world (occ_cache).
#include <cstddef>
#include "mw_setup_common.h"
#include "mw_records.h"

namespace {

constexpr int kClipSlots = 24;      // work lists of the clipper in LDS: 2 x 10 vertices of 48 bytes each (23 KB)

// The lanes of ONE wavefront (= the workgroup) exchange data through LDS: the LDS pipeline serves a wavefront's accesses in program order, so
// an exchange needs the compiler to keep that order and the data to have arrived (lgkmcnt) — not the vmcnt(0) of a full
// workgroup fence, which also waits for every record store in flight (one to two microseconds each time at one wavefront per SIMD).
__device__ inline void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// exclusive scan inside the env's L-lane group; total: the group's sum
__device__ inline int group_excl_scan(int v, int sub, int L, int &total)
{
    int x = v;
    for (int off = 1; off < L; off <<= 1) {
        const int y = __shfl_up(x, off, L);
        if (sub >= off) x += y;
    }
    total = __shfl(x, L - 1, L);
    return x - v;
}

// a polygon's 128 bytes in one go: vertices [0..11], uv [12..19], normal [20..22], nv, tex, rgb [25..27], xf [28..31]
__device__ inline void load_poly(const mw_poly *p, float (&q)[32])
{
    const float4 *q4 = reinterpret_cast<const float4 *>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float4 w = q4[i]; q[4 * i] = w.x; q[4 * i + 1] = w.y; q[4 * i + 2] = w.z; q[4 * i + 3] = w.w; }
}

// a work-list vertex with the primitive's flat colour
__device__ inline mwgl::Vert to_vert(const mwgl::ClipVert &c, const float col[3])
{
    mwgl::Vert v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v.clip[i] = c.clip[i]; v.win[i] = c.win[i]; }
    v.st[0] = c.st[0]; v.st[1] = c.st[1];
    v.col[0] = col[0]; v.col[1] = col[1]; v.col[2] = col[2];
    v.clipmask = 0u;
    return v;
}

// does the triangle (window coordinates) leave setup?  (the snapped-area cull of setup_triangle alone)
__device__ inline bool tri_front(const float wa[4], const float wb[4], const float wc[4], bool multisampled)
{
    const float off = multisampled ? 0.0f : 0.5f;
    const int32_t x0 = mwgl::iround_even((wa[0] - off) * 256.0f), y0 = mwgl::iround_even((wa[1] - off) * 256.0f);
    const int32_t x1 = mwgl::iround_even((wb[0] - off) * 256.0f), y1 = mwgl::iround_even((wb[1] - off) * 256.0f);
    const int32_t x2 = mwgl::iround_even((wc[0] - off) * 256.0f), y2 = mwgl::iround_even((wc[1] - off) * 256.0f);
    const int64_t dx01 = x0 - x1, dy01 = y0 - y1, dx20 = x2 - x0, dy20 = y2 - y0;
    return dx01 * dy20 - dx20 * dy01 < 0;
}

// Bitonic sort of the n <= 64 R keys in keys[] (ascending; unused places count as the largest key) by one wavefront, the
// low halves of the sorted keys stored to order[1 ..].
template <int R>
__device__ inline void sort_store_keys(const uint32_t *keys, int n, int lane, uint16_t *order)
{
    uint32_t x[R];
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = 64 * r + lane < n ? keys[64 * r + lane] : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 2; k <= 64 * R; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                // partners in two registers of the same lane
                const int jj = j >> 6;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (r & jj) continue;
                    const bool up = ((64 * r) & k) == 0;        // (k >= 128: the lane's bits do not matter)
                    const uint32_t lo = min(x[r], x[r ^ jj]), hi = max(x[r], x[r ^ jj]);
                    x[r] = up ? lo : hi; x[r ^ jj] = up ? hi : lo;
                }
            } else {
                // partners in two lanes, the same register
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const uint32_t y = (uint32_t)__shfl_xor((int)x[r], j);
                    const bool up = (((64 * r) | lane) & k) == 0;
                    const bool keep_min = ((lane & j) == 0) == up;
                    x[r] = keep_min ? min(x[r], y) : max(x[r], y);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (64 * r + lane < n) order[1 + 64 * r + lane] = (uint16_t)(x[r] & 0xFFFFu);
}

// ---------------------------------------------------------------- occlusion culling (big scenes)
// A Maze view holds ~95 front-facing polygons inside the frustum and ~13 that own a sample: everything else lies behind
// walls.  With an unpitched camera a wall that spans the whole height of the world (the slab [lo, hi] of all room
// polygons, the eye inside it) hides every room polygon behind it in the screen columns it covers: a ray to a point
// of the slab farther away crosses the wall's plane inside the slab.  So: every such wall in front of the eye marks the
// column bins it covers completely with its farthest depth there (the nearest wall wins the bin), and a polygon
// whose columns are all marked with depths in front of its nearest vertex is dropped before it costs a record and the
// raster kernel's visits.  Frames do not change: a dropped polygon owns no sample —
//   * its fragments are no nearer than its nearest vertex up to the rounding of the float32 depth plane (1e-6 of the depth
//     range), and "in front" demands more than three steps of the 16-bit depth buffer: 1/z_wall - 1/z_poly > 1.2e-3, one
//     step of D16 over [0.04, 100] being 3.8e-4 in 1/z (GL_LESS ties go to the polygon drawn first);
//   * all margins are on the keeping side: walls shrink by 0.05 px and are cut at z = 0.1 (the near plane is at 0.04),
//     polygons grow by 0.1 px, polygons with a vertex nearer than 0.1 are kept untested.
// tests/test_gpu_env_api.py::test_occlusion_culling_never_changes_a_frame compares MW_OCCLUSION=0 / 1 bit for bit.
#define MW_OCC_BINS 256
#define MW_OCC_CAP 192
#define MW_ORDER_CAP 512       // big scenes: lists up to this length get a near-to-far visiting order

// occ_z[0 .. BINS): the bins; occ_z[BINS .. BINS + BINS / 16): the largest value of every group of 16 bins
// Is everything with window x in [xlo, xhi] and depth >= zq hidden?
__device__ inline bool occluded_span(const float *occ_z, float xlo, float xhi, float zq, float bins_per_px)
{
    const float fb0 = floorf(fmaxf(xlo, 0.0f) * bins_per_px), fb1 = floorf(fmaxf(xhi, 0.0f) * bins_per_px);
    const int b0 = (int)fminf(fb0, (float)(MW_OCC_BINS - 1)), b1 = (int)fminf(fb1, (float)(MW_OCC_BINS - 1));
    const float thr = zq * __builtin_amdgcn_rcpf(fmaf(1.2e-3f, zq, 1.0f)) * 0.9999f;
    // every bin of b0 .. b1 in front of thr: whole groups through their maxima
    const int g0 = (b0 + 15) >> 4, g1 = (b1 + 1) >> 4;
    if (g0 >= g1) {
        for (int b = b0; b <= b1; ++b)
            if (!(occ_z[b] < thr)) return false;
        return true;
    }
    for (int b = b0; b < (g0 << 4); ++b)
        if (!(occ_z[b] < thr)) return false;
    for (int g = g0; g < g1; ++g)
        if (!(occ_z[MW_OCC_BINS + g] < thr)) return false;
    for (int b = g1 << 4; b <= b1; ++b)
        if (!(occ_z[b] < thr)) return false;
    return true;
}

__device__ inline bool occluded(const float *occ_z, const mwgl::Vert v[4], float bins_per_px)
{
    float zq = 1e30f, xmn = 1e30f, xmx = -1e30f;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ok &= v[k].clip[3] >= 0.1f;
        xmn = fminf(xmn, v[k].win[0]); xmx = fmaxf(xmx, v[k].win[0]); zq = fminf(zq, v[k].clip[3]);
    }
    return ok && occluded_span(occ_z, xmn - 0.1f, xmx + 0.1f, zq, bins_per_px);
}

// The same question answered from a sparse table over the bins (rmq[k * BINS + b] = the largest value of bins b .. b + 2^k - 1):
// two reads instead of a walk over up to 46 bins and group maxima whose length differs from lane to lane (the wavefront
// waited for its widest polygon at every call).
__device__ inline bool occluded_span_rmq(const float *rmq, float xlo, float xhi, float zq, float bins_per_px)
{
    const float fb0 = floorf(fmaxf(xlo, 0.0f) * bins_per_px), fb1 = floorf(fmaxf(xhi, 0.0f) * bins_per_px);
    const int b0 = (int)fminf(fb0, (float)(MW_OCC_BINS - 1)), b1 = (int)fminf(fb1, (float)(MW_OCC_BINS - 1));
    const float thr = zq * __builtin_amdgcn_rcpf(fmaf(1.2e-3f, zq, 1.0f)) * 0.9999f;
    if (b1 < b0) return true;       // (an empty span, like the walk's)
    const int k = 31 - __builtin_clz((unsigned)(b1 - b0 + 1));
    const float m = fmaxf(rmq[k * MW_OCC_BINS + b0], rmq[k * MW_OCC_BINS + b1 - (1 << k) + 1]);
    return m < thr;
}

__device__ inline bool occluded_rmq(const float *rmq, const mwgl::Vert v[4], float bins_per_px)
{
    float zq = 1e30f, xmn = 1e30f, xmx = -1e30f;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ok &= v[k].clip[3] >= 0.1f;
        xmn = fminf(xmn, v[k].win[0]); xmx = fmaxf(xmx, v[k].win[0]); zq = fminf(zq, v[k].clip[3]);
    }
    return ok && occluded_span_rmq(rmq, xmn - 0.1f, xmx + 0.1f, zq, bins_per_px);
}

}  // namespace

// view_flags: bit 0 top view, bit 1 draw the agent marker, bit 2 get_visible_ents' proxy pass (rooms untextured + one
// 0.2 m box per entity, tagged 0x10000 | slot).  S: samples per pixel of the target (1, 4, 8, 16).  L: lanes per env
// (8, 16, 32 or 64); n_env: envs a.env_base .. a.env_base + n_env - 1.
// BIG: one env per wavefront (L = 64) with the sifting and occlusion culling of big scenes — their LDS stays out of the
// small scenes' kernel, whose workgroups then fit four to a CU.
// SFIX: the target's samples per pixel when fixed at compile time (8: the observation path — a quarter of the record
// writer's code, 36 registers less), 0: taken from the launch
template <bool BIG, int SFIX>
__device__ inline void geom_body(const MwArgs &a, int view_flags, int S_, int L, int n_env)
{
    const int S = SFIX ? SFIX : S_;
    // (244 dwords per slot: consecutive slots start in different LDS banks)
    struct ClipSlot { mwgl::ClipVert l[2][MWGL_MAX_CLIP_VERTS]; float pad[4]; };
    __shared__ ClipSlot s_clip[kClipSlots];
    __shared__ int s_pos[BIG ? 1 : 8][66];       // (the big scenes' kernel serves one env per wavefront: its LDS must stay within a quarter of a CU's)
    __shared__ uint32_t s_meta[64], s_res[64];      // per work list: owner lane | clip planes << 8; vertices | list << 4 | front-facing fan triangles << 8
    __shared__ float s_occ_z[BIG ? MW_OCC_BINS + MW_OCC_BINS / 16 : 1];      // occlusion culling: farthest depth of the nearest wall per column bin, group maxima
    __shared__ int s_occ_n;
    float *s_rmq = reinterpret_cast<float *>(&s_clip[0]);      // big scenes, during the sift: range maxima of the column bins (9 levels x 256)
    float4 *s_occ_wall = reinterpret_cast<float4 *>(s_rmq + 9 * MW_OCC_BINS);       // ... and the occluders, two quads each (xl, xr, A2, B2), (D, -, -, -)
    static_assert(!BIG || sizeof(ClipSlot) * kClipSlots >= (9 * MW_OCC_BINS + 8 * MW_OCC_CAP) * sizeof(float), "the sift borrows the clipper's LDS");
    __shared__ uint16_t s_list[BIG ? 4096 : 1];
    __shared__ uint32_t s_key[BIG ? MW_ORDER_CAP : 1];        // big scenes: (depth bound << 16 | list index) of every record, for the visiting order      // big scenes: the polygons that pass the cheap tests (frustum, occlusion), in drawing order
#ifdef MW_PERF_HOOKS      // (tools/perf/kgprof.py; the product build carries no time stamps)
#define KGP_ON (a.k1_prof != nullptr)
#else
#define KGP_ON false
#endif
    const unsigned long long tstart = KGP_ON ? __builtin_readcyclecounter() : 0ull;
    const unsigned long long rt_start = KGP_ON ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz, the same clock on every CU
    [[maybe_unused]] unsigned long long ts[4] = {0, 0, 0, 0};     // inside the sift: occluder walls, column bins, boxes, polygons
    [[maybe_unused]] int kp_nocc = 0, kp_nkept = 0, kp_nclip = 0;
    [[maybe_unused]] unsigned long long kp_clip = 0, kp_emit = 0, kp_e_shfl = 0, kp_e_setup = 0, kp_e_write = 0;
    const int lane = threadIdx.x;
    const int epw = 64 / L, sub = lane & (L - 1), grp = lane / L;
    const int rel = (int)blockIdx.x * epw + grp;
    const bool live = rel < n_env;
    const int env = a.env_base + (live ? rel : n_env - 1);      // a padding group recomputes the last env and writes nothing
    const int set = a.shared_geom ? 0 : env;
    const bool top = (view_flags & 1) != 0, proxy = (view_flags & 4) != 0, ms = S > 1;
    // big scenes: the per-world culling data (occ_cache: full-height walls, boxes of eight polygons) lives in HBM, one to two
    // microseconds away at one wavefront per SIMD — requested here, ahead of the camera's double-precision prologue, instead of
    // one round trip per 64 walls inside the sift (29 k of the kernel's 150 k cycles)
    constexpr int kWallPf = BIG ? 4 : 1;
    float4 pf_wall[kWallPf], pf_box0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), pf_box1 = pf_box0;
    float pf_sgn[kWallPf], pf_hdr[3] = {0.0f, 0.0f, 0.0f};
    auto prefetch_occ = [&]() {
        if (!BIG || L != 64 || a.occ_cache == nullptr) return;
        const float *pc = a.occ_cache + (size_t)set * MW_OCC_CACHE_STRIDE(a.max_polys);
#pragma unroll
        for (int q = 0; q < kWallPf; ++q) {
            const int i = lane + 64 * q;
            if (i < a.max_polys) {
                pf_wall[q] = reinterpret_cast<const float4 *>(pc + MW_OCC_CACHE_HDR + 8 * (size_t)i)[0];
                pf_sgn[q] = pc[MW_OCC_CACHE_HDR + 8 * (size_t)i + 4];
            }
        }
        if (lane < (a.max_polys + 7) / 8) {
            const float4 *b4 = reinterpret_cast<const float4 *>(pc + MW_OCC_CACHE_HDR + 8 * (size_t)a.max_polys + 8 * (size_t)lane);
            pf_box0 = b4[0]; pf_box1 = b4[1];
        }
        pf_hdr[0] = pc[0]; pf_hdr[1] = pc[1]; pf_hdr[2] = pc[2];
    };
#pragma unroll
    for (int q = 0; q < kWallPf; ++q) { pf_wall[q] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); pf_sgn[q] = 0.0f; }
    prefetch_occ();
    // the env's entity table, one slot per lane of its group (the walks below ask for slot after slot: a chain of dependent
    // global loads otherwise — with five slots and two passes, a quarter of PickupObjects' kernel)
    const bool ent_pre = a.E <= L;
    const int my_ekind = (ent_pre && sub < a.E) ? a.ekind[(size_t)sub * a.N + env] : 0;
    const int my_estatic = (ent_pre && sub < a.E) ? a.estatic[(size_t)sub * a.N + env] : 0;
    const int my_emesh = (ent_pre && sub < a.E && my_ekind == MW_ENT_MESH) ? a.emesh[(size_t)sub * a.N + env] : 0;
    const int my_mesh_ntris = (ent_pre && sub < a.E && my_ekind == MW_ENT_MESH) ? (int)a.mesh[my_emesh].ntris : 0;
    auto ent_kind = [&](int s0) { return ent_pre ? __shfl(my_ekind, s0, L) : a.ekind[(size_t)s0 * a.N + env]; };
    auto ent_static = [&](int s0) { return ent_pre ? __shfl(my_estatic, s0, L) : a.estatic[(size_t)s0 * a.N + env]; };
    auto ent_mesh_ntris = [&](int s0) { return ent_pre ? __shfl(my_mesh_ntris, s0, L) : (int)a.mesh[a.emesh[(size_t)s0 * a.N + env]].ntris; };
    // ---- the frame's GL state (every lane of the group evaluates it: same instruction stream)
    mwgl::Frame f;
    const double px = a.ax[env], py = a.ay[env], pz = a.az[env], dir = a.adir[env];
    float eye_x = 0.0f, eye_y = 0.0f, eye_z = 0.0f;
    {
        double lpos[3], lcol[3], lamb[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            lpos[i] = a.light[(size_t)(3 + i) * a.N + env];
            lcol[i] = a.light[(size_t)(6 + i) * a.N + env];
            lamb[i] = a.light[(size_t)(9 + i) * a.N + env];
        }
        if (top) {
            double min_x = a.extent[(size_t)0 * a.N + env] - 1, max_x = a.extent[(size_t)1 * a.N + env] + 1;
            double min_z = a.extent[(size_t)2 * a.N + env] - 1, max_z = a.extent[(size_t)3 * a.N + env] + 1;
            const double width = max_x - min_x, height = max_z - min_z;
            const double aspect = width / height, fb_aspect = (double)a.W / (double)a.H;
            if (aspect > fb_aspect) {
                const double new_h = width / fb_aspect, h_diff = new_h - height;
                min_z -= h_diff / 2; max_z += h_diff / 2;
            } else if (aspect < fb_aspect) {
                const double new_w = height * fb_aspect, w_diff = new_w - width;
                min_x -= w_diff / 2; max_x += w_diff / 2;
            }
            mwgl::frame_top(f, min_x, max_x, min_z, max_z);
        } else {
            // Agent.cam_pos / cam_dir via gen_rot_matrix (math.py:11-27, entity.py:476-503) as numpy evaluates them
            const double cam_height = a.cam[(size_t)0 * a.N + env], fwd_disp = a.cam[(size_t)1 * a.N + env];
            const double pitch_deg = a.cam[(size_t)2 * a.N + env], fov_y = a.cam[(size_t)3 * a.N + env];
            const mw::SinCos hd = mw::sincos_det(dir / 2.0);
            const double ya = hd.c, yc = -1.0 * hd.s;
            const double ry00 = ya * ya - yc * yc, ry02 = 2.0 * (ya * yc), ry11 = ya * ya + yc * yc;
            const double pitch = pitch_deg * kPi / 180.0;
            const mw::SinCos hp = mw::sincos_det(pitch / 2.0);
            const double za = hp.c, zd = -1.0 * hp.s;
            const double rz00 = za * za - zd * zd, rz01 = 2.0 * (0.0 - za * zd);
            const double eye[3] = {px + fwd_disp * ry00, py + cam_height * ry11, pz + fwd_disp * ry02};
            const double cd[3] = {rz00 * ry00, rz01 * ry11, rz00 * ry02};
            const double at[3] = {eye[0] + cd[0], eye[1] + cd[1], eye[2] + cd[2]};
            const mw::SinCos hf = mw::sincos_det(fov_y / 2 * kPi / 180);
            mwgl::frame_perspective(f, eye, at, hf.c / hf.s, a.W, a.H);
            eye_x = (float)eye[0]; eye_y = (float)eye[1]; eye_z = (float)eye[2];
        }
        mwgl::frame_finish(f, a.W, a.H, lpos, lcol, lamb);
    }
    mwgl::Xform cam;
    mwgl::make_xform(f, f.view, f.view_flags, cam);
    unsigned long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    [[maybe_unused]] unsigned long long racc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0, rstart = 0, rtot[2] = {0, 0}, rvert[2] = {0, 0};      // per-round phases, summed over the rounds
#define KGP_R(k) do { if (KGP_ON) { const unsigned long long n_ = __builtin_readcyclecounter(); racc[k] += n_ - tprev; tprev = n_; } } while (0)
    if (KGP_ON) tp[0] = __builtin_readcyclecounter();

    const mw_poly *polys = a.polys + (size_t)set * a.max_polys;
    const int np = a.npolys[set];
    float *hdr = a.envhdr + (size_t)env * MW_ENVHDR;

    // ---- the env's draw list behind the polygons: boxes in drawing order (static entities first, inside display list 1,
    // then the dynamic ones, each in slot order).  A mesh entity's triangles belong to the mesh kernel; here it gets its
    // place in the drawing order (one draw id per triangle) and its transform (env header, written by the group's lane 0).
    // This lane's primitive of round r is item r * L + sub: a polygon, else face (item - np) % 6 of box (item - np) / 6,
    // else the agent marker.
    int total_meshes = 0, total_mesh_tris = 0, total_boxes = 0;
    uint64_t mesh_in_view = 0ull;       // per entity slot: a mesh entity that is drawn
    uint32_t tile_mask = 0u;            // tile sub + k L lies in a drawn mesh entity's tile rectangle: bit k
    // What a mesh entity's place in the frame takes — its transform, the view test on its bounding sphere, the tile rectangle —
    // depends on the entity alone: lane s of the group works it out for slot s, all slots side by side (a serial walk with every
    // lane repeating every entity was 26 k of PickupObjects' 95 k cycles); the walk behind it only counts and writes.
    struct MeshView { bool in_view; uint32_t rect; int ntris, mid, first, tex; mwgl::Xform ex; };
    auto mesh_view = [&](int s0, MeshView &mv) {
        const int mid = a.emesh[(size_t)s0 * a.N + env];
        const MwMeshDesc *mdp = a.mesh + mid;
        mv.mid = mid; mv.ntris = (int)mdp->ntris; mv.first = (int)mdp->first; mv.tex = (int)mdp->tex;
        const float pos[3] = {(float)a.epos[((size_t)0 * a.E + s0) * a.N + env], (float)a.epos[((size_t)1 * a.E + s0) * a.N + env],
                              (float)a.epos[((size_t)2 * a.E + s0) * a.N + env]};
        const float scale = (float)a.egeom[((size_t)6 * a.E + s0) * a.N + env];
        mwgl::entity_xform(f, pos, (float)(a.edir[(size_t)s0 * a.N + env] * 180 / kPi), scale, true, mv.ex);
        // whole-entity frustum test on the bounding sphere (conservative): clip-space distance to the five planes
        bool in_view = true;
        uint32_t rect;      // the tiles the entity's bounding sphere can touch: tx0 | tx1 << 8 | ty0 << 16 | ty1 << 24 (image rows)
        {
            // (the sphere about the bounding box's centre: a ball's origin lies at its foot)
            const float brad = mdp->radius * scale * 1.001f + 1e-3f;
            mwgl::Vert o;
            const float ctr[3] = {mdp->center[0], mdp->center[1], mdp->center[2]};
            mwgl::transform_vertex(f, mv.ex, ctr, o);
            const float w = o.clip[3];
            const float p00 = f.proj.m[0], p11 = f.proj.m[5];
            float xlo = -1.0f, xhi = 1.0f, ylo = -1.0f, yhi = 1.0f;
            if (!top) {
                const float lx = sqrtf(fmaf(p00, p00, 1.0f)), ly = sqrtf(fmaf(p11, p11, 1.0f));
                in_view = !(w + brad < 0.04f) && !(w - fabsf(o.clip[0]) < -(brad * lx)) && !(w - fabsf(o.clip[1]) < -(brad * ly));
                if (w - brad > 0.04f) {
                    // eye-space box around the sphere, projected: x / d with d in [w - r, w + r]
                    const float dn = 1.0f / (w - brad), df = 1.0f / (w + brad);
                    const float nxl = o.clip[0] - brad * p00, nxh = o.clip[0] + brad * p00, nyl = o.clip[1] - brad * p11, nyh = o.clip[1] + brad * p11;
                    xlo = fminf(nxl * dn, nxl * df); xhi = fmaxf(nxh * dn, nxh * df);
                    ylo = fminf(nyl * dn, nyl * df); yhi = fmaxf(nyh * dn, nyh * df);
                }
            } else {
                xlo = o.clip[0] - brad * fabsf(p00); xhi = o.clip[0] + brad * fabsf(p00);
                ylo = o.clip[1] - brad * fabsf(p11); yhi = o.clip[1] + brad * fabsf(p11);
            }
            const float Wf = (float)a.W, Hf = (float)a.H;
            int x0 = (int)floorf(fmaxf((xlo * 0.5f + 0.5f) * Wf - 1.5f, 0.0f)), x1 = (int)fminf((xhi * 0.5f + 0.5f) * Wf + 1.5f, Wf - 1.0f);
            int g0 = (int)floorf(fmaxf((ylo * 0.5f + 0.5f) * Hf - 1.5f, 0.0f)), g1 = (int)fminf((yhi * 0.5f + 0.5f) * Hf + 1.5f, Hf - 1.0f);
            // ... cut down to the window bounds of the bounding box's corners when all eight lie in front of the eye (a convex
            // combination of the corners then projects to a convex combination of their projections; a key is a thin
            // slab inside a sphere of its length)
            if (in_view) {
                bool front = true;
                float cxl = 1e30f, cxh = -1e30f, cyl = 1e30f, cyh = -1e30f;
                for (int c = 0; c < 8; ++c) {
                    const float p[3] = {(c & 1) ? mdp->bmax[0] : mdp->bmin[0], (c & 2) ? mdp->bmax[1] : mdp->bmin[1], (c & 4) ? mdp->bmax[2] : mdp->bmin[2]};
                    mwgl::Vert q;
                    mwgl::transform_vertex(f, mv.ex, p, q);
                    front &= top || q.clip[3] > 0.05f;
                    cxl = fminf(cxl, q.win[0]); cxh = fmaxf(cxh, q.win[0]); cyl = fminf(cyl, q.win[1]); cyh = fmaxf(cyh, q.win[1]);
                }
                if (front && cxl <= cxh && cyl <= cyh) {
                    x0 = max(x0, (int)floorf(fmaxf(cxl - 1.5f, 0.0f))); x1 = min(x1, (int)fminf(cxh + 1.5f, Wf - 1.0f));
                    g0 = max(g0, (int)floorf(fmaxf(cyl - 1.5f, 0.0f))); g1 = min(g1, (int)fminf(cyh + 1.5f, Hf - 1.0f));
                }
            }
            if (x1 < x0 || g1 < g0) in_view = false;
            const int y0 = a.H - 1 - g1, y1 = a.H - 1 - g0;
            rect = (uint32_t)(x0 / MW_TILE_W) | ((uint32_t)(x1 / MW_TILE_W) << 8) | ((uint32_t)(y0 / MW_TILE_H) << 16) | ((uint32_t)(y1 / MW_TILE_H) << 24);
        }
        mv.in_view = in_view; mv.rect = rect;
    };
    // (a slot's lane keeps what its header entry still needs at the end of the kernel — nothing written by one lane is read back by another)
    int my_mesh_j = -1, my_boxes_before = 0, my_tris_before = 0;
    uint32_t big_mask = 0u;             // the drawn meshes of 1 024 triangles and more (entry numbers): first in the entity kernel's list
    static_assert(MW_MAX_MESH_ENTS <= 32, "one bit per drawn mesh");
    MeshView my_mv;
    my_mv.in_view = false; my_mv.rect = 0u; my_mv.ntris = 0; my_mv.mid = 0; my_mv.first = 0; my_mv.tex = -1;
    if (ent_pre && !proxy && sub < a.E && my_ekind == MW_ENT_MESH) mesh_view(sub, my_mv);
    for (int pass = 0; pass < 2; ++pass) {
        for (int s0 = 0; s0 < a.E; ++s0) {
            const int kind = ent_kind(s0);
            if (kind == MW_ENT_NONE) continue;
            const bool stat = proxy ? true : ent_static(s0) != 0;
            if (stat != (pass == 0)) continue;
            if (kind == MW_ENT_MESH && !proxy) {
                // (with the slots' views on their lanes: the slot's lane holds everything, the others ask for what they need)
                MeshView walk_mv;
                if (!ent_pre) mesh_view(s0, walk_mv);
                const bool in_view = ent_pre ? __shfl((int)my_mv.in_view, s0, L) != 0 : walk_mv.in_view;
                if (!in_view) continue;
                const uint32_t rect = ent_pre ? (uint32_t)__shfl((int)my_mv.rect, s0, L) : walk_mv.rect;
                const int md_ntris = ent_pre ? __shfl(my_mv.ntris, s0, L) : walk_mv.ntris;
                if (total_meshes < MW_MAX_MESH_ENTS && total_mesh_tris + md_ntris < 0xC000 && s0 < 64) {
                    if (md_ntris >= 1024) big_mask |= 1u << total_meshes;
                    if (ent_pre && sub == s0) { my_mesh_j = total_meshes; my_boxes_before = total_boxes; my_tris_before = total_mesh_tris; }
                    if ((ent_pre ? sub == s0 : sub == 0) && live) {
                        const MeshView &mv = ent_pre ? my_mv : walk_mv;
                        float *m = hdr + MW_HDR_MESH + MW_HDR_MESH_STRIDE * total_meshes;
                        m[0] = __int_as_float(s0);
                        m[1] = __int_as_float(total_boxes);         // boxes drawn before: turned into the first draw id at the end
                        m[2] = __int_as_float(md_ntris);
                        m[3] = __int_as_float(mv.first);
                        m[4] = __int_as_float(mv.tex);
                        m[5] = mv.ex.nscale;
                        m[6] = mv.ex.light[0]; m[7] = mv.ex.light[1]; m[8] = mv.ex.light[2];
#pragma unroll
                        for (int k = 0; k < 16; ++k) m[9 + k] = mv.ex.mvp.m[k];
                        m[25] = __int_as_float(total_mesh_tris);    // mesh triangles drawn before
                        m[26] = __uint_as_float(rect);
                        m[27] = __int_as_float(mv.mid);
                    }
                    if (a.tile_list) {
                        // the tiles this lane answers for — sub, sub + L, ... — inside the entity's tile rectangle
                        const int tx0 = (int)(rect & 255u), tx1 = (int)((rect >> 8) & 255u), ty0 = (int)((rect >> 16) & 255u), ty1 = (int)(rect >> 24);
                        for (int k = 0, t = sub; t < a.n_tiles && k < 32; ++k, t += L) {
                            const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
                            if (tx >= tx0 && tx <= tx1 && ty >= ty0 && ty <= ty1) tile_mask |= 1u << k;
                        }
                    }
                    mesh_in_view |= 1ull << s0;
                    total_mesh_tris += md_ntris;      // one draw id per triangle (a mesh out of view takes none)
                    ++total_meshes;
                } else {
                    atomicOr(a.status, MW_ST_VIS_OVERFLOW);
                }
            } else if (kind == MW_ENT_BOX || proxy) {
                ++total_boxes;
            }
        }
    }
    // the want-th box of the drawing order: its slot and the mesh triangles drawn before it
    auto find_box = [&](int want, int &slot_out, int &idbase_out) {
        int nb = 0, mt = 0;
        slot_out = -1; idbase_out = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int s0 = 0; s0 < a.E; ++s0) {
                const int kind = ent_kind(s0);
                if (kind == MW_ENT_NONE) continue;
                const bool stat = proxy ? true : ent_static(s0) != 0;
                if (stat != (pass == 0)) continue;
                if (kind == MW_ENT_MESH && !proxy) {
                    if (s0 < 64 && ((mesh_in_view >> s0) & 1ull)) mt += ent_mesh_ntris(s0);
                } else if (kind == MW_ENT_BOX || proxy) {
                    if (nb == want) { slot_out = s0; idbase_out = mt; }
                    ++nb;
                }
            }
        }
    };

    // ---- big scenes (one env per wavefront): what the culling below needs of the world alone — the slab, the full-height
    // walls, a bounding box for every eight consecutive polygons — is derived once per world and kept (a.occ_cache; a Maze
    // world lives for hundreds of frames, and three passes over its 65 KB of polygons per frame were a third of this
    // kernel's time).  Whatever rewrites a set's polygons zeroes occ_valid[set] (mw_gen.h, mw_set_world).
    const bool sifted = BIG && L == 64 && np > 64 && np <= 4096;
    bool occ_on = BIG && L == 64 && a.occlusion && !top && f.view.m[4] == 0.0f && f.view.m[6] == 0.0f;      // an unpitched camera
    const bool cached = BIG && L == 64 && a.occ_cache != nullptr && np > 0 && np <= a.max_polys;
    if (!cached) occ_on = false;
    float *oc = cached ? a.occ_cache + (size_t)set * MW_OCC_CACHE_STRIDE(a.max_polys) : nullptr;
    float *oc_wall = oc + MW_OCC_CACHE_HDR, *oc_box = oc + MW_OCC_CACHE_HDR + 8 * (size_t)a.max_polys;
    if (cached && (occ_on || sifted) && a.occ_valid[set] != np + 1) {
        // the slab: lowest and highest point of the room polygons
        // (the loops over the env's polygons keep the next polygon's 128 bytes in flight while they work on the current one)
        float lo = 1e30f, hi = -1e30f;
        {
            float qn[32];
            if (lane < np) load_poly(polys + lane, qn);
            for (int i = lane; i < np; i += 64) {
                float q[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) q[k] = qn[k];
                if (i + 64 < np) load_poly(polys + i + 64, qn);
                const int nv = __float_as_int(q[23]) & 0xFF;
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nv) { lo = fminf(lo, q[3 * k + 1]); hi = fmaxf(hi, q[3 * k + 1]); }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        int n_walls = 0;
        float qn[32];
        if (lane < np) load_poly(polys + lane, qn);
        for (int base = 0; base < np; base += 64) {
            const int i = base + lane;
            float q[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) q[k] = qn[k];
            if (i + 64 < np) load_poly(polys + i + 64, qn);
            const int nvf = i < np ? __float_as_int(q[23]) : 0, nv = nvf & 0xFF;
            // -- a wall that can hide things: a vertical rectangle from lo to hi (triangles, and the quads of static
            // entities (flag bits), are no walls)
            bool wall = i < np && (nvf == 4 || nvf == (4 | MW_POLY_QUAD));
            float vx[4], vy[4], vz[4];
            bool ys = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vx[k] = q[3 * k]; vy[k] = q[3 * k + 1]; vz[k] = q[3 * k + 2];
                ys &= vy[k] == lo || vy[k] == hi;
            }
            const bool pa = vx[0] == vx[1] && vz[0] == vz[1] && vx[2] == vx[3] && vz[2] == vz[3] && vy[0] != vy[1] && vy[2] != vy[3];
            const bool pb = vx[1] == vx[2] && vz[1] == vz[2] && vx[3] == vx[0] && vz[3] == vz[0] && vy[1] != vy[2] && vy[3] != vy[0];
            const float bx = pa ? vx[2] : vx[1], bz = pa ? vz[2] : vz[1];
            wall = wall && ys && (pa || pb) && !(bx == vx[0] && bz == vz[0]);
            const uint64_t wm = __ballot(wall);
            if (wall) {
                float4 *w4 = reinterpret_cast<float4 *>(oc_wall + 8 * (size_t)(n_walls + (int)__popcll((unsigned long long)(wm & ((1ull << lane) - 1ull)))));
                w4[0] = make_float4(vx[0], vz[0], bx, bz);
                w4[1] = make_float4(pa ? vy[1] - vy[0] : vy[1] - vy[2], 0.0f, 0.0f, 0.0f);        // which way round the rectangle is drawn
            }
            n_walls += (int)__popcll((unsigned long long)wm);
            // -- the bounding box of polygons 8c .. 8c + 7 (eight consecutive lanes) and what may be concluded from it:
            // bit 0 every polygon stands in world coordinates (the frustum test applies), bit 1 every polygon is a room
            // quad (the occlusion test applies)
            float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (i < np && k < nv) {
                    mn[0] = fminf(mn[0], vx[k]); mn[1] = fminf(mn[1], vy[k]); mn[2] = fminf(mn[2], vz[k]);
                    mx[0] = fmaxf(mx[0], vx[k]); mx[1] = fmaxf(mx[1], vy[k]); mx[2] = fmaxf(mx[2], vz[k]);
                }
            int fl = i < np ? (((nvf & MW_POLY_XF) ? 0 : 1) | ((nv == 4 && !(nvf & MW_POLY_XF)) ? 2 : 0)) : 3;
#pragma unroll
            for (int o = 1; o < 8; o <<= 1) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], o)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], o)); }
                fl &= __shfl_xor(fl, o);
            }
            if ((lane & 7) == 0 && i < np) {
                float4 *b4 = reinterpret_cast<float4 *>(oc_box + 8 * (size_t)(i >> 3));
                b4[0] = make_float4(mn[0], mn[1], mn[2], mx[0]);
                b4[1] = make_float4(mx[1], mx[2], __int_as_float(fl), 0.0f);
            }
        }
        if (lane == 0) { oc[0] = __int_as_float(n_walls); oc[1] = lo; oc[2] = hi; }
        __threadfence();
        __syncthreads();
        if (lane == 0) a.occ_valid[set] = np + 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        prefetch_occ();         // (what was requested at the top was the cache before this world)
    } else if (cached && a.shared_geom) {
        // (a shared set may have been filled by another wavefront of this launch after this one asked for it)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        prefetch_occ();
    }

    // ---- occlusion culling: the walls that hide what lies behind them
    const float bins_per_px = (float)MW_OCC_BINS / (float)a.W;
    if (occ_on) {
        const float lo = pf_hdr[1], hi = pf_hdr[2];
        const int n_walls = __float_as_int(pf_hdr[0]);
        occ_on = eye_y > lo + 1e-3f && eye_y < hi - 1e-3f;
        if (!occ_on && lane == 0) s_occ_n = 0;
        if (occ_on) {
            const float wc = 0.1f;
            const float *V = f.view.m;          // column major: eye x = V[0] x + V[8] z + V[12], depth = -(V[2] x + V[10] z + V[14])
            const float p00 = f.proj.m[0], halfw = (float)a.W * 0.5f;
            const float inv_p00 = 1.0f / p00, inv_hp = 1.0f / (halfw * p00), px_per_bin = 1.0f / bins_per_px;
            // (walls 0 .. 255 were requested at the top of the kernel; the occluders are numbered by ballots: a place per wall in
            // the order of the world's list, no atomics)
            int n_found = 0;
            for (int it = 0, i = lane; it * 64 < n_walls; ++it, i += 64) {
                float4 w0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                float sgn = 0.0f;
                if (it < kWallPf) {
#pragma unroll
                    for (int q = 0; q < kWallPf; ++q) if (q == it) { w0 = pf_wall[q]; sgn = pf_sgn[q]; }
                } else if (i < n_walls) {
                    w0 = reinterpret_cast<const float4 *>(oc_wall + 8 * (size_t)i)[0];
                    sgn = oc_wall[8 * (size_t)i + 4];
                }
                bool found = false;
                float o_xl = 0.0f, o_xr = 0.0f, o_a2 = 0.0f, o_b2 = 0.0f, o_d = 0.0f;
                do {
                if (i >= n_walls) break;
                const float vx0 = w0.x, vz0 = w0.y, bx = w0.z, bz = w0.w;
                // drawn at all?  GL_CCW front faces (miniworld.py:512): the winding normal, s * (tz, 0, -tx) for a vertical
                // rectangle over the foot line B0 -> B1 = (tx, tz), points at the eye — by a centimetre at least
                const float tx = bx - vx0, tz = bz - vz0;
                const float side = tz * (eye_x - vx0) - tx * (eye_z - vz0);
                const float facing = sgn > 0.0f ? side : -side;
                if (!(facing > 0.0f && facing * facing > 1e-4f * (tx * tx + tz * tz))) break;
                // its foot line in eye space: (x, depth) of the two vertical edges, cut at depth wc
                float ea = fmaf(V[0], vx0, fmaf(V[8], vz0, V[12]));
                float wa = -fmaf(V[2], vx0, fmaf(V[10], vz0, V[14]));
                float eb = fmaf(V[0], bx, fmaf(V[8], bz, V[12]));
                float wb = -fmaf(V[2], bx, fmaf(V[10], bz, V[14]));
                if (!(wa >= wc) && !(wb >= wc)) break;
                // (hardware reciprocals: their last-bit error moves a column by 1e-5 px, the margins are 0.05)
                if (!(wa >= wc)) { const float t = (wc - wa) * __builtin_amdgcn_rcpf(wb - wa); ea = fmaf(t, eb - ea, ea); wa = wc; }
                else if (!(wb >= wc)) { const float t = (wc - wb) * __builtin_amdgcn_rcpf(wa - wb); eb = fmaf(t, ea - eb, eb); wb = wc; }
                if (!(fmaxf(wa, wb) < 95.0f)) break;         // the far plane is at 100
                const float xa = halfw * fmaf(p00, ea * __builtin_amdgcn_rcpf(wa), 1.0f);
                const float xb = halfw * fmaf(p00, eb * __builtin_amdgcn_rcpf(wb), 1.0f);
                const float xl = fminf(xa, xb) + 0.05f, xr = fmaxf(xa, xb) - 0.05f;
                if (!(xr > 0.0f && xl < (float)a.W && xr - xl >= px_per_bin)) break;
                // depth along the wall as a function of the pixel column: A ex + B w = D with ex / w = (x / halfw - 1) / p00
                float A = wb - wa, B = -(eb - ea), D = A * ea + B * wa;
                if (D < 0.0f) { A = -A; B = -B; D = -D; }
                if (!(D > 1e-4f)) break;
                const float A2 = A * inv_hp, B2 = B - A * inv_p00;
                const float dl = fmaf(A2, xl, B2), dr = fmaf(A2, xr, B2);
                if (!(dl * 100.0f > D && dr * 100.0f > D)) break;       // depths below 100 at both ends (and positive denominators)
                found = true; o_xl = xl; o_xr = xr; o_a2 = A2; o_b2 = B2; o_d = D;
                } while (false);
                const uint64_t fm = __ballot(found);
                const int j = n_found + (int)__popcll((unsigned long long)(fm & ((1ull << lane) - 1ull)));
                if (found && j < MW_OCC_CAP) {
                    s_occ_wall[2 * j] = make_float4(o_xl, o_xr, o_a2, o_b2);
                    s_occ_wall[2 * j + 1] = make_float4(o_d, 0.0f, 0.0f, 0.0f);
                }
                n_found += (int)__popcll((unsigned long long)fm);
            }
            if (lane == 0) s_occ_n = n_found;
        }
        wave_lds_sync();
        if (KGP_ON) ts[0] = __builtin_readcyclecounter();
        if (occ_on) {
            const int n_occ = s_occ_n < MW_OCC_CAP ? s_occ_n : MW_OCC_CAP;
            kp_nocc = n_occ;
            // (four bins per lane in registers, the walls in the outer loop: one broadcast read of a wall serves them all)
            static_assert(MW_OCC_BINS == 256, "four column bins per lane");
            float zb[4] = {1e30f, 1e30f, 1e30f, 1e30f}, xa[4], xb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int b = lane + 64 * q; xa[q] = (float)b / bins_per_px; xb[q] = (float)(b + 1) / bins_per_px; }
            // (a bin inside the wall's span has positive denominators at both ends — the wall was listed with positive ones at its own ends and
            // they are linear in x —: the farther end is the smaller one, one reciprocal.  Its last-bit error is 1e-7 of the depth; the
            // margin below is 1e-4.)
#pragma unroll 2
            for (int j = 0; j < n_occ; ++j) {
                const float4 w = s_occ_wall[2 * j];
                const float o4 = s_occ_wall[2 * j + 1].x;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float far = o4 * __builtin_amdgcn_rcpf(fminf(fmaf(w.z, xa[q], w.w), fmaf(w.z, xb[q], w.w)));
                    zb[q] = (xa[q] >= w.x && xb[q] <= w.y) ? fminf(zb[q], far) : zb[q];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = lane + 64 * q;
                const float z = zb[q] * 1.0001f;
                s_occ_z[b] = z;
                // the largest value of every group of 16 bins (= 16 consecutive lanes)
                float gz = z;
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) gz = fmaxf(gz, __shfl_xor(gz, o));
                if ((lane & 15) == 0) s_occ_z[MW_OCC_BINS + (b >> 4)] = gz;
                s_rmq[b] = z;
            }
            // the sparse table of the sift's range queries, level by level (in the clipper's work lists: idle until the rounds)
            static_assert(sizeof(ClipSlot) * kClipSlots >= 9 * MW_OCC_BINS * sizeof(float), "the range-maximum table borrows the clipper's LDS");
            for (int k = 1; k <= 8; ++k) {
                wave_lds_sync();
                const int half = 1 << (k - 1);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = lane + 64 * q;
                    if (b + 2 * half <= MW_OCC_BINS) s_rmq[k * MW_OCC_BINS + b] = fmaxf(s_rmq[(k - 1) * MW_OCC_BINS + b], s_rmq[(k - 1) * MW_OCC_BINS + b + half]);
                }
            }
        }
        wave_lds_sync();
    }

    // ---- big scenes: most polygons lie outside the frustum or behind walls — sift them with the cheap tests first, so that
    // the rounds below (clipping, setup) run over the survivors only.  First the boxes of eight polygons each: a box
    // that fails a test takes its polygons along unseen.  A box fails only where each of its polygons would (its margins
    // are the wider ones), so the list is the one the polygons' own tests leave.
    int n_polys_drawn = np;
    if (KGP_ON) ts[1] = ts[2] = ts[3] = __builtin_readcyclecounter();
    if (sifted) {
        uint16_t *s_box = reinterpret_cast<uint16_t *>(s_key);      // the boxes that stay, ascending (s_key is not in use yet)
        const int nbox = (np + 7) >> 3;
        int nkept = 0;
        for (int base = 0; base < nbox; base += 64) {
            const int c = base + lane;
            bool keep = c < nbox;
            if (keep && cached) {
                // (boxes 0 .. 63 were requested at the top of the kernel)
                const float4 b0 = base == 0 ? pf_box0 : reinterpret_cast<const float4 *>(oc_box + 8 * (size_t)c)[0];
                const float4 b1 = base == 0 ? pf_box1 : reinterpret_cast<const float4 *>(oc_box + 8 * (size_t)c)[1];
                const int fl = __float_as_int(b1.z);
                if (fl & 1) {
                    const float mn[3] = {b0.x, b0.y, b0.z}, mx[3] = {b0.w, b1.x, b1.y};
                    const mwgl::BoxView bv = mwgl::box_view(mn, mx, cam.mvp.m, f.vp_scale[0], f.vp_trans[0]);
                    const uint32_t all = bv.all;
                    const bool front = bv.front;
                    const float xmn = bv.xmn, xmx = bv.xmx, zq = bv.zq;
                    keep = all == 0u;
                    if (keep && occ_on && (fl & 2) && front && occluded_span_rmq(s_rmq, xmn - 0.15f, xmx + 0.15f, zq * 0.9999f, bins_per_px)) keep = false;
                }
            }
            const uint64_t m = __ballot(keep);
            if (keep) s_box[nkept + __popcll((unsigned long long)(m & ((1ull << lane) - 1ull)))] = (uint16_t)c;
            nkept += __popcll((unsigned long long)m);
        }
        wave_lds_sync();
        kp_nkept = nkept;
        if (KGP_ON) ts[2] = __builtin_readcyclecounter();
        int ns = 0;
        // (the next turn's polygon is in flight while this turn's is tested)
        auto cand = [&](int k) { return k < 8 * nkept ? 8 * (int)s_box[k >> 3] + (k & 7) : np; };
        float qn[32];
        if (cand(lane) < np) load_poly(polys + cand(lane), qn);
        for (int base = 0; base < 8 * nkept; base += 64) {
            const int i = cand(base + lane);
            bool keep = false;
            float q[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) q[k] = qn[k];
            if (cand(base + 64 + lane) < np) load_poly(polys + cand(base + 64 + lane), qn);
            if (i < np) {
                const int nvf = __float_as_int(q[23]), nv = nvf & 0xFF;
                keep = !(proxy && (nvf & MW_POLY_ENTITY));
                if (keep && !(nvf & MW_POLY_XF)) {
                    mwgl::Vert v[4];
                    uint32_t all = 0x3Fu;
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) {
                        const float pk[3] = {k2 < nv ? q[3 * k2] : q[0], k2 < nv ? q[3 * k2 + 1] : q[1], k2 < nv ? q[3 * k2 + 2] : q[2]};
                        mwgl::transform_vertex(f, cam, pk, v[k2]);
                        all &= v[k2].clipmask;
                    }
                    keep = all == 0u;       // no frustum plane has every vertex outside
                    if (keep && occ_on && nv == 4 && occluded_rmq(s_rmq, v, bins_per_px)) keep = false;
                    // seen from behind — the other side of most walls of other rooms: with no vertex clipped its two
                    // triangles are the ones the rounds would set up, and when both are back-facing by more than snapping
                    // the vertices can change (clearly_back), setup would drop both: no record either way
                    if (keep && (v[0].clipmask | v[1].clipmask | v[2].clipmask | v[3].clipmask) == 0u) {
                        const bool fan = nv == 3 || !(nvf & MW_POLY_QUAD);
                        // v[] holds the polygon's own vertices k (v[3] = v[0] for a triangle): fan (1,2,0) (2,3,0), list quad (0,1,3) (1,2,3)
                        const bool b0 = fan ? mwgl::clearly_back(v[1].win, v[2].win, v[0].win) : mwgl::clearly_back(v[0].win, v[1].win, v[3].win);
                        const bool b1 = nv == 3 || (fan ? mwgl::clearly_back(v[2].win, v[3].win, v[0].win) : mwgl::clearly_back(v[1].win, v[2].win, v[3].win));
                        if (b0 && b1) keep = false;
                    }
                }
            }
            const uint64_t m = __ballot(keep);
            if (keep) s_list[ns + __popcll((unsigned long long)(m & ((1ull << lane) - 1ull)))] = (uint16_t)i;
            ns += __popcll((unsigned long long)m);
        }
        n_polys_drawn = ns;
        wave_lds_sync();
        if (KGP_ON) ts[3] = __builtin_readcyclecounter();
    }

    if (KGP_ON) tp[1] = tprev = __builtin_readcyclecounter();
    int count = 0;          // the env's list length so far (uniform in the group)
    float stale_n[3] = {0.0f, 1.0f, 0.0f};
    const int marker = (view_flags & 2) ? 1 : 0;
    if (marker && np > 0) { stale_n[0] = polys[np - 1].n[0]; stale_n[1] = polys[np - 1].n[1]; stale_n[2] = polys[np - 1].n[2]; }      // (only the marker is lit by it)
    if (total_boxes > 0) { stale_n[0] = 0.0f; stale_n[1] = -1.0f; stale_n[2] = 0.0f; }     // drawBox ends with glNormal3f(0, -1, 0)
    const float white[3] = {1.0f, 1.0f, 1.0f};
    const int npd = n_polys_drawn;      // polygons among the items
    const int n_items = npd + 6 * total_boxes + marker;
    // list position of each box's first record and of the end of the boxes: a mesh's first draw id is the number of
    // records drawn before it plus the mesh triangles drawn before it
    int *pos = s_pos[BIG ? 0 : grp];      // (the big kernel is launched with L = 64 only)
    if (sub == 0) for (int i = 0; i <= (total_boxes < 64 ? total_boxes : 64); ++i) pos[i] = -1;

    // one TRIANGLE per lane: lanes 2k and 2k + 1 of a round hold the two triangles of primitive k (both evaluate its vertex
    // stage; each sets up, clips and writes its own triangle)
    // Rounds: the polygons' triangles fill rounds of their own, then every box gets twelve consecutive lanes of ONE round
    // (L / 12 boxes to a round), the agent marker a last round — a round is either all polygons or all boxes, so its lanes
    // share one branch of the vertex stage, and a box never straddles two rounds (its clipped-vertex vote below needs all
    // twelve lanes).  Groups of 8 lanes cannot hold a box: they keep the plain order, two triangles per primitive.
    const bool aligned = L >= 16;
    const int bpr = L / 12;
    const int poly_rounds = (2 * npd + L - 1) / L, box_rounds = aligned ? (total_boxes + bpr - 1) / bpr : 0;
    const int n_rounds = aligned ? poly_rounds + box_rounds + marker : (2 * n_items + L - 1) / L;
    // (the envs of a wavefront go through the rounds together — the clipper's work lists are served by all 64 lanes, eight to a
    // list —: an env with fewer rounds than its neighbours idles through the rest)
    const int n_rounds_wave = L < 64 ? (int)__reduce_max_sync(~0ull, (unsigned)n_rounds) : n_rounds;
    for (int rnd = 0; rnd < n_rounds_wave; ++rnd) {
        if (KGP_ON) rstart = tprev;
        int item, tsel;
        if (rnd >= n_rounds) {
            item = n_items; tsel = 0;
        } else if (!aligned || rnd < poly_rounds) {
            const int t = rnd * L + sub;
            item = t >> 1; tsel = t & 1;
            if (aligned && item >= npd) item = n_items;         // (behind the polygons of the last polygon round: nothing)
        } else if (rnd < poly_rounds + box_rounds) {
            const int b = (rnd - poly_rounds) * bpr + sub / 12, tri = sub % 12;
            item = (sub < 12 * bpr && b < total_boxes) ? npd + 6 * b + (tri >> 1) : n_items;
            tsel = tri & 1;
        } else {
            item = sub == 0 ? n_items - 1 : n_items;            // the marker: one triangle
            tsel = 0;
        }
        int box_slot = -1, box_idbase = 0;
        if (__any(item >= npd && item < npd + 6 * total_boxes)) find_box(item >= npd ? (item - npd) / 6 : -1, box_slot, box_idbase);
        if (!(item >= npd && item < npd + 6 * total_boxes)) box_slot = -1;
        mwgl::Vert v[4];
        int nt = 0, tex = -1;
        bool direct = false;        // the triangles of v[]: (0,1,3) (1,2,3), or (0,1,2) (0,2,3) for a direct quad
        uint32_t id_base = 0u, tag = 0u;
        bool is_box = false, clipped_l = false, in_list = false;
        if (item < npd) {
            // the polygon's 128 bytes in one go: vertices [0..11], uv [12..19], normal [20..22], nv, tex, rgb [25..27], xf [28..31]
            float q[32];
            load_poly(polys + (sifted ? (int)s_list[item] : item), q);
            static_assert(sizeof(mw_poly) == 128 && offsetof(mw_poly, uv) == 48 && offsetof(mw_poly, n) == 80 && offsetof(mw_poly, nv) == 92 &&
                          offsetof(mw_poly, tex) == 96 && offsetof(mw_poly, rgb) == 100 && offsetof(mw_poly, xf) == 112, "mw_poly layout");
            const int nvf = __float_as_int(q[23]), nv = nvf & 0xFF;
            if (!(proxy && (nvf & MW_POLY_ENTITY))) {       // the queries draw rooms only
                mwgl::Xform ex;
                const bool own = (nvf & MW_POLY_XF) != 0;
                const float qxf[4] = {q[28], q[29], q[30], q[31]};
                if (own) mwgl::entity_xform(f, qxf, qxf[3], 1.0f, false, ex);
                const mwgl::Xform &x = own ? ex : cam;
                float col[3];
                const float qn[3] = {q[20], q[21], q[22]}, qc[3] = {q[25], q[26], q[27]};
                mwgl::light_vertex(f, x, qn, proxy ? white : qc, col);
                tex = proxy ? -1 : __float_as_int(q[24]);
                // a polygon is the fan (1,2,0) (2,3,0), a quad of the list (0,1,3) (1,2,3): the same triangles of v[] once
                // the polygon's vertices are taken one further round
                const bool fan = nv == 3 || !(nvf & MW_POLY_QUAD);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // source vertex: k for a quad of the list, k + 1 for a polygon, (1, 2, 0, 0) for a triangle
                    const int ka = k, kb = (k + 1) & 3, kc = k == 0 ? 1 : (k == 1 ? 2 : 0);
                    float pk[3], st[2];
#pragma unroll
                    for (int c = 0; c < 3; ++c) pk[c] = nv == 3 ? q[3 * kc + c] : (fan ? q[3 * kb + c] : q[3 * ka + c]);
#pragma unroll
                    for (int c = 0; c < 2; ++c) st[c] = nv == 3 ? q[12 + 2 * kc + c] : (fan ? q[12 + 2 * kb + c] : q[12 + 2 * ka + c]);
                    mwgl::transform_vertex(f, x, pk, v[k]);
                    v[k].st[0] = tex >= 0 ? st[0] : 0.0f; v[k].st[1] = tex >= 0 ? st[1] : 0.0f;
                    v[k].col[0] = col[0]; v[k].col[1] = col[1]; v[k].col[2] = col[2];
                }
                nt = nv == 3 ? 1 : 2;
                if (occ_on && !sifted && nv == 4 && !own && occluded(s_occ_z, v, bins_per_px)) nt = 0;      // hidden behind a full-height wall (a sifted list has passed this very test)
            }
        } else if (box_slot >= 0) {
            const int slot = box_slot, fc = (item - npd) % 6;
            is_box = true;
            id_base = (uint32_t)box_idbase;
            tag = proxy ? (0x10000u | (uint32_t)slot) : 0u;
            const double ex_ = a.epos[((size_t)0 * a.E + slot) * a.N + env], ey_ = a.epos[((size_t)1 * a.E + slot) * a.N + env],
                         ez_ = a.epos[((size_t)2 * a.E + slot) * a.N + env];
            float lo[3], hi[3], base_col[3];
            mwgl::Xform ex;
            const mwgl::Xform *x = &cam;
            if (proxy) {
                // drawBox(pos -+ 0.1, pos.y .. pos.y + 0.2): python doubles through glVertex3f, under the camera alone
                lo[0] = (float)(ex_ - 0.1); lo[1] = (float)ey_; lo[2] = (float)(ez_ - 0.1);
                hi[0] = (float)(ex_ + 0.1); hi[1] = (float)(ey_ + 0.2); hi[2] = (float)(ez_ + 0.1);
                base_col[0] = base_col[1] = base_col[2] = 1.0f;
            } else {
                const float pos[3] = {(float)ex_, (float)ey_, (float)ez_};
                mwgl::entity_xform(f, pos, (float)(a.edir[(size_t)slot * a.N + env] * (180 / kPi)), 1.0f, false, ex);
                x = &ex;
                const double sx = a.egeom[((size_t)0 * a.E + slot) * a.N + env], sy = a.egeom[((size_t)1 * a.E + slot) * a.N + env],
                             sz = a.egeom[((size_t)2 * a.E + slot) * a.N + env];
                lo[0] = (float)(-sx / 2); lo[1] = 0.0f; lo[2] = (float)(-sz / 2);
                hi[0] = (float)(sx / 2); hi[1] = (float)sy; hi[2] = (float)(sz / 2);
#pragma unroll
                for (int k = 0; k < 3; ++k) base_col[k] = (float)a.egeom[((size_t)(3 + k) * a.E + slot) * a.N + env];
                in_list = a.estatic[(size_t)slot * a.N + env] != 0;
            }
            float n[3], col[3];
            mwgl::box_normal(fc, n);
            mwgl::light_vertex(f, *x, n, base_col, col);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sel = mwgl::box_sel(fc, k);
                const float p[3] = {(sel & 1) ? hi[0] : lo[0], (sel & 2) ? hi[1] : lo[1], (sel & 4) ? hi[2] : lo[2]};
                mwgl::transform_vertex(f, *x, p, v[k]);
                v[k].st[0] = v[k].st[1] = 0.0f;
                v[k].col[0] = col[0]; v[k].col[1] = col[1]; v[k].col[2] = col[2];
                clipped_l |= v[k].clipmask != 0u;
            }
            nt = 2;
        } else if (marker && item == n_items - 1) {
            // Agent.render (entity.py:518-539): no glNormal3f => lit with the normal the last immediate-mode glNormal3f or the
            // end of the display list left current (glDrawArrays with a normal array leaves it alone)
            const mw::SinCos sc = mw::sincos_det(dir);
            const double rad = a.agent_radius, hgt = a.agent_height;
            const double p[3] = {px + 0 * hgt, py + 1 * hgt, pz + 0 * hgt};
            const double dv[3] = {sc.c * rad, 0 * rad, -sc.s * rad}, rv[3] = {sc.s * rad, 0 * rad, sc.c * rad};
            float pv[3][3];
            for (int i = 0; i < 3; ++i) {
                pv[0][i] = (float)(p[i] + dv[i]);
                pv[2][i] = (float)(p[i] + 0.75 * (rv[i] - dv[i]));
                pv[1][i] = (float)(p[i] + 0.75 * (-rv[i] - dv[i]));
            }
            const float red[3] = {1.0f, 0.0f, 0.0f};
            float col[3];
            mwgl::light_vertex(f, cam, stale_n, red, col);
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                mwgl::Vert &d = k == 2 ? v[3] : v[k];          // the triangle (0, 1, 3) of v[]
                mwgl::transform_vertex(f, cam, pv[k], d);
                d.st[0] = d.st[1] = 0.0f;
                d.col[0] = col[0]; d.col[1] = col[1]; d.col[2] = col[2];
            }
            nt = 1;
            id_base = (uint32_t)total_mesh_tris;
        }
        // one glBegin / glEnd per box: a clipped vertex anywhere in the call sends all six faces through the draw module's
        // pipeline, whose quads split (0,1,3) (1,2,3); an immediate-mode call without one splits (0,1,2) (0,2,3); a static
        // box sits in display list 1 (always the first split).  The six faces of a box are six consecutive items.
        {
            const uint64_t cm = __ballot(clipped_l && is_box);
            bool box_clipped = clipped_l;
            if (is_box) {
                const int fc = (item - npd) % 6;
                // the faces of this box in this round: twelve consecutive lanes of the same group (a box may straddle two
                // rounds: then the other faces' flags are recomputed from the box's vertices — all 8 corners appear in any 2 faces,
                // so a straddling box is handled by testing the corners directly)
                const int first = lane - (2 * fc + tsel);
                const bool whole = sub - (2 * fc + tsel) >= 0 && sub - (2 * fc + tsel) + 11 < L;
                if (whole) {
                    box_clipped = ((cm >> first) & 0xFFFull) != 0ull;
                } else {
                    // recompute: any corner of the box clipped?  (faces 0 and 1 hold all eight corners)
                    box_clipped = false;        // filled below by the slow path
                }
                if (!whole) {
                    const int slot = box_slot;
                    const double ex_ = a.epos[((size_t)0 * a.E + slot) * a.N + env], ey_ = a.epos[((size_t)1 * a.E + slot) * a.N + env],
                                 ez_ = a.epos[((size_t)2 * a.E + slot) * a.N + env];
                    float lo[3], hi[3];
                    mwgl::Xform ex;
                    const mwgl::Xform *x = &cam;
                    if (proxy) {
                        lo[0] = (float)(ex_ - 0.1); lo[1] = (float)ey_; lo[2] = (float)(ez_ - 0.1);
                        hi[0] = (float)(ex_ + 0.1); hi[1] = (float)(ey_ + 0.2); hi[2] = (float)(ez_ + 0.1);
                    } else {
                        const float pos[3] = {(float)ex_, (float)ey_, (float)ez_};
                        mwgl::entity_xform(f, pos, (float)(a.edir[(size_t)slot * a.N + env] * (180 / kPi)), 1.0f, false, ex);
                        x = &ex;
                        const double sx = a.egeom[((size_t)0 * a.E + slot) * a.N + env], sy = a.egeom[((size_t)1 * a.E + slot) * a.N + env],
                                     sz = a.egeom[((size_t)2 * a.E + slot) * a.N + env];
                        lo[0] = (float)(-sx / 2); lo[1] = 0.0f; lo[2] = (float)(-sz / 2);
                        hi[0] = (float)(sx / 2); hi[1] = (float)sy; hi[2] = (float)(sz / 2);
                    }
                    for (int c = 0; c < 8; ++c) {
                        const float p[3] = {(c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]};
                        mwgl::Vert o;
                        mwgl::transform_vertex(f, *x, p, o);
                        box_clipped |= o.clipmask != 0u;
                    }
                }
                direct = !box_clipped && !in_list;
            }
        }

        KGP_R(2);
        if (KGP_ON) { if (rnd == 0) rvert[0] = tprev - rstart; rvert[1] = tprev - rstart; }
        // this lane's triangle of v[]: (0,1,3) / (1,2,3), or (0,1,2) / (0,2,3) for a direct quad
        const mwgl::Vert va = tsel ? (direct ? v[0] : v[1]) : v[0], vb = tsel ? v[2] : v[1], vc = tsel ? v[3] : (direct ? v[2] : v[3]);
        // ---- pass 1: does it survive, and as how many triangles
        // (the setup itself waits for pass 2: nothing of it has to live through the clipper and the scan)
        int cnt = 0;
        bool clipped = false;
        if (tsel < nt && !(va.clipmask & vb.clipmask & vc.clipmask)) {
            if ((va.clipmask | vb.clipmask | vc.clipmask) == 0u) cnt = tri_front(va.win, vb.win, vc.win, ms) ? 1 : 0;
            else clipped = true;
        }
        KGP_R(3);
        // A triangle that crosses a frustum plane goes through a work list in LDS: kClipSlots of the wave's clipped
        // triangles at a time are clipped, counted, placed in the env's list and written (the serial form, for a triangle
        // across all six planes: clipped in pass 1, and again in pass 2 when the wave has more of them than lists).
        const uint64_t cmask = __ballot(clipped);
        const uint64_t below = (1ull << lane) - 1ull;
        const int n_clip = (int)__popcll((unsigned long long)cmask);
        const bool keep_lists = n_clip <= kClipSlots;
        const uint32_t cm_union = va.clipmask | vb.clipmask | vc.clipmask;
        // The triangles are clipped eight at a time, eight lanes to a list — lane e owns the polygon's edge e -> e + 1: it passes
        // vertex e on if that lies inside the plane and makes the crossing's vertex; the coverage ballots give everybody's
        // place in the output list.  Same arithmetic per vertex as clip_triangle, one step per plane instead of one per
        // plane and vertex.  (A triangle that crosses all six planes could grow to nine vertices: then
        // clip_triangle does the work, one lane per triangle.)
        const bool par = !__any(clipped && __popc(cm_union) == 6);
        mwgl::ClipVert *kept = nullptr;
        int kept_n = 0;
        const int cg0 = lane & ~7, ce = lane & 7;
        const int my_v = (int)__popcll((unsigned long long)(cmask & below));        // this lane's clipped triangle among the wave's
        // clips triangles v0 .. v0 + kClipSlots - 1 of the wave's clipped ones (list = number - v0) and notes what became of each
        auto clip_lists = [&](int v0) {
            if (clipped && my_v >= v0 && my_v < v0 + kClipSlots) {
                ClipSlot &cs = s_clip[my_v - v0];
                mwgl::clip_copy_in(cs.l[0][0], va); mwgl::clip_copy_in(cs.l[0][1], vb); mwgl::clip_copy_in(cs.l[0][2], vc);
            }
            wave_lds_sync();
            const int nb = n_clip - v0 < kClipSlots ? n_clip - v0 : kClipSlots;
            for (int b0 = 0; b0 < nb; b0 += 8) {
                const int slot = b0 + (lane >> 3);
                const bool valid = slot < nb;
                ClipSlot &cs = s_clip[valid ? slot : 0];
                uint32_t cm = valid ? s_meta[v0 + slot] >> 8 : 0u;
                int n = valid ? 3 : 0, cur = 0;
                while (__any(cm != 0u && n >= 3)) {
                    const bool act = cm != 0u && n >= 3;
                    const int plane = act ? __ffs((int)cm) - 1 : 0;
                    if (act) cm &= cm - 1u;
                    const mwgl::ClipVert *in = cs.l[cur];
                    mwgl::ClipVert *out = cs.l[cur ^ 1];
                    const bool me = act && ce < n;
                    const int nxt = ce + 1 < n ? ce + 1 : 0;
                    mwgl::ClipVert V;
                    if (me) V = in[ce];
                    const float dp_prev = me ? mwgl::clip_dist(V, plane) : 0.0f;
                    const float dp = __shfl(dp_prev, cg0 + nxt);
                    const bool bad = me && (!(dp_prev == dp_prev) || dp_prev - dp_prev != 0.0f);      // NaN / Inf: the triangle is dropped
                    const bool emit = me && dp_prev >= 0.0f, cross = me && ((dp >= 0.0f) != (dp_prev >= 0.0f));
                    const uint32_t bg = (uint32_t)(__ballot(bad) >> cg0) & 0xFFu;
                    const uint32_t eg = (uint32_t)(__ballot(emit) >> cg0) & 0xFFu, xg = (uint32_t)(__ballot(cross) >> cg0) & 0xFFu;
                    const uint32_t lowm = (1u << ce) - 1u;
                    const int pos = __popc(eg & lowm) + __popc(xg & lowm);
                    if (emit) out[pos] = V;
                    if (cross) {
                        const mwgl::ClipVert Vn = in[nxt];
                        // the new vertex is interpolated from the endpoint that is closer to the plane (clip_triangle)
                        const bool from_cur = fabsf(dp) < fabsf(dp_prev);
                        const float t = (from_cur ? dp : dp_prev) / (from_cur ? dp - dp_prev : dp_prev - dp);
                        mwgl::clip_interp<false>(f, out[pos + (emit ? 1 : 0)], t, from_cur ? Vn : V, from_cur ? V : Vn);
                    }
                    if (act) { n = bg ? 0 : __popc(eg) + __popc(xg); cur ^= 1; }
                    wave_lds_sync();
                }
                if (n < 3) n = 0;
                // the fan (r[e-1], r[e], r[0]), e = 2 .. n-1: which of its triangles leave setup
                const mwgl::ClipVert *r = cs.l[cur];
                bool front = false;
                if (ce >= 2 && ce < n) front = tri_front(r[ce - 1].win, r[ce].win, r[0].win, ms);
                const uint32_t fm = (uint32_t)(__ballot(front) >> cg0) & 0xFFu;
                if (valid && ce == 0) s_res[v0 + slot] = (uint32_t)n | ((uint32_t)cur << 4) | (fm << 8);
            }
            wave_lds_sync();
        };
        // the records of the fans in lists 0 .. of triangles v0 ..: every triangle of every fan on a lane of its own — list
        // position from the owner's (base_now), setup, record
        auto emit_fans = [&](int v0, int base_now) {
            const int nb = n_clip - v0 < kClipSlots ? n_clip - v0 : kClipSlots;
            [[maybe_unused]] unsigned long long e_prev = KGP_ON ? __builtin_readcyclecounter() : 0ull;
            for (int b0 = 0; b0 < nb; b0 += 8) {
                const int slot = b0 + (lane >> 3);
                const bool valid = slot < nb;
                const uint32_t res = valid ? s_res[v0 + slot] : 0u;
                const int owner = valid ? (int)(s_meta[v0 + slot] & 63u) : 0;
                const int n = (int)(res & 15u), cur = (int)((res >> 4) & 1u);
                const uint32_t fm = (res >> 8) & 0xFFu;
                const int o_base = __shfl(base_now, owner), o_tex = __shfl(tex, owner), o_env = __shfl(env, owner), o_live = __shfl((int)live, owner);
                const uint32_t o_tag = (uint32_t)__shfl((int)tag, owner), o_idb = (uint32_t)__shfl((int)id_base, owner);
                const float o_col[3] = {__shfl(va.col[0], owner), __shfl(va.col[1], owner), __shfl(va.col[2], owner)};
                [[maybe_unused]] const unsigned long long e0 = KGP_ON ? __builtin_readcyclecounter() : 0ull;
                if (KGP_ON) kp_e_shfl += e0 - e_prev;
                bool e_ok = false;
                mwgl::TriSetup t2;
                int idx = 0;
                const mwgl::ClipVert *r = s_clip[slot].l[cur];
                if (ce >= 2 && ce < n && ((fm >> ce) & 1u)) {
                    idx = o_base + __popc(fm & ((1u << ce) - 1u));
                    e_ok = mwgl::setup_triangle(to_vert(r[ce - 1], o_col), to_vert(r[ce], o_col), to_vert(r[0], o_col), ms, o_tex >= 0, t2) && o_live && idx < a.max_vis;
                }
                [[maybe_unused]] const unsigned long long e1 = KGP_ON ? __builtin_readcyclecounter() : 0ull;
                if (KGP_ON) kp_e_setup += e1 - e0;
                if (e_ok) {
                    {
                        const uint32_t zlo = (SFIX == 8 ? mwrec::write_tri_s<8>(a, o_env, idx, o_tag ? o_tag : (uint32_t)idx + o_idb, t2, o_tex) : mwrec::write_tri(a, o_env, idx, o_tag ? o_tag : (uint32_t)idx + o_idb, t2, o_tex, S));
                        if (BIG && idx < MW_ORDER_CAP) s_key[idx] = (zlo << 16) | (uint32_t)idx;
                    }
                }
                if (KGP_ON) { e_prev = __builtin_readcyclecounter(); kp_e_write += e_prev - e1; }
            }
            wave_lds_sync();
        };
        int total = 0, base = 0;
        bool scanned = false;
        if (par && n_clip) {
            // kClipSlots triangles at a time: clipped, counted, placed (the scan sees zero for the clipped triangles still
            // to come: they lie behind these in the list), written — then the lists serve the next ones
            if (clipped) s_meta[my_v] = (uint32_t)lane | (cm_union << 8);
            for (int v0 = 0; v0 < n_clip; v0 += kClipSlots) {
                [[maybe_unused]] const unsigned long long c0 = KGP_ON ? __builtin_readcyclecounter() : 0ull;
                clip_lists(v0);
                [[maybe_unused]] const unsigned long long c1 = KGP_ON ? __builtin_readcyclecounter() : 0ull;
                if (clipped && my_v >= v0 && my_v < v0 + kClipSlots) cnt = __popc(s_res[my_v] >> 8);
                base = count + group_excl_scan(cnt, sub, L, total);
                emit_fans(v0, base);
                if (KGP_ON) { kp_clip += c1 - c0; kp_emit += __builtin_readcyclecounter() - c1; kp_nclip += n_clip > v0 + kClipSlots ? kClipSlots : n_clip - v0; }
            }
            scanned = true;     // the last scan saw every count
        } else {
            uint64_t pend = cmask;
            if (keep_lists && pend) pend = 1ull;        // one turn for everybody
            while (pend) {
                uint64_t batch = ~0ull, rest = 0ull;
                if (!keep_lists) {
                    batch = 0ull; rest = pend;
                    for (int k = 0; k < kClipSlots && rest; ++k) { batch |= rest & (0ull - rest); rest &= rest - 1ull; }
                }
                if (clipped && ((batch >> lane) & 1ull)) {
                    const int slot = (int)__popcll((unsigned long long)((keep_lists ? cmask : batch) & below));
                    kept_n = mwgl::clip_triangle<false>(f, va, vb, vc, s_clip[slot].l[0], s_clip[slot].l[1], &kept);
                    int c = 0;
                    for (int i = 2; i < kept_n; ++i) c += tri_front(kept[i - 1].win, kept[i].win, kept[0].win, ms) ? 1 : 0;
                    cnt = c;
                }
                pend = rest;
            }
        }
        KGP_R(4);
        // ---- list positions
        if (!scanned) base = count + group_excl_scan(cnt, sub, L, total);
        count += total;
        if (base + cnt > a.max_vis && cnt) atomicOr(a.status, MW_ST_VIS_OVERFLOW);
        KGP_R(5);
        // ---- pass 2: the records
        if (cnt == 1 && !clipped && live && base < a.max_vis) {
            mwgl::TriSetup ts;
            if (mwgl::setup_triangle(va, vb, vc, ms, tex >= 0, ts)) {
                const uint32_t zlo = (SFIX == 8 ? mwrec::write_tri_s<8>(a, env, base, tag ? tag : (uint32_t)base + id_base, ts, tex) : mwrec::write_tri(a, env, base, tag ? tag : (uint32_t)base + id_base, ts, tex, S));
                if (BIG && base < MW_ORDER_CAP) s_key[base] = (zlo << 16) | (uint32_t)base;
            }
        }
        if (!par) {
            const bool mine = clipped && cnt > 0;
            uint64_t pend = __ballot(mine);
            if (keep_lists && pend) pend = 1ull;
            while (pend) {
                uint64_t batch = ~0ull, rest = 0ull;
                if (!keep_lists) {
                    batch = 0ull; rest = pend;
                    for (int k = 0; k < kClipSlots && rest; ++k) { batch |= rest & (0ull - rest); rest &= rest - 1ull; }
                }
                if (mine && ((batch >> lane) & 1ull)) {
                    const mwgl::ClipVert *r = kept;
                    int n = kept_n;
                    if (!keep_lists) {
                        const int slot = (int)__popcll((unsigned long long)(batch & below));
                        mwgl::ClipVert *r2;
                        n = mwgl::clip_triangle<false>(f, va, vb, vc, s_clip[slot].l[0], s_clip[slot].l[1], &r2);
                        r = r2;
                    }
                    int idx = base;
                    for (int i = 2; i < n; ++i) {
                        mwgl::TriSetup t2;
                        if (mwgl::setup_triangle(to_vert(r[i - 1], va.col), to_vert(r[i], va.col), to_vert(r[0], va.col), ms, tex >= 0, t2)) {
                            if (live && idx < a.max_vis) {
                                const uint32_t zlo = (SFIX == 8 ? mwrec::write_tri_s<8>(a, env, idx, tag ? tag : (uint32_t)idx + id_base, t2, tex) : mwrec::write_tri(a, env, idx, tag ? tag : (uint32_t)idx + id_base, t2, tex, S));
                                if (BIG && idx < MW_ORDER_CAP) s_key[idx] = (zlo << 16) | (uint32_t)idx;
                            }
                            ++idx;
                        }
                    }
                }
                pend = rest;
            }
        }
        KGP_R(6);
        if (KGP_ON && rnd < 2) rtot[rnd] = tprev - rstart;
        // list positions the meshes' draw ids need
        if (is_box && tsel == 0 && (item - npd) % 6 == 0 && (item - npd) / 6 < 64) pos[(item - npd) / 6] = base;
        if (marker && tsel == 0 && item == n_items - 1) pos[total_boxes < 64 ? total_boxes : 64] = base;
    }
    wave_lds_sync();
    if (BIG && a.rec_order && L < 64) {
        if (sub == 0 && live) a.rec_order[(size_t)env * (a.max_vis + 1)] = 0;      // no order
    } else if (BIG && a.rec_order) {
        // big scenes (one env per wavefront): the records' visiting order by ascending depth bound — K2 visits a tile's
        // triangles near to far and stops at the first one that lies behind everything the tile holds by then
        uint16_t *order = a.rec_order + (size_t)env * (a.max_vis + 1);
        const int n = count;
        // (a list that overflowed its capacity has no order either: the records behind max_vis were never written, their keys
        // never set, and the order array holds max_vis entries — mw_check reports the overflow, the frame stays in bounds)
        if (n > MW_ORDER_CAP || n > a.max_vis || !live) {
            if (lane == 0 && live) order[0] = 0;
        } else {
            wave_lds_sync();
            // (64 R keys in R registers per lane, key i = 64 r + lane: the network's exchanges are lane shuffles and
            // register swaps — no LDS round trip per stage)
            if (n <= 64) sort_store_keys<1>(s_key, n, lane, order);
            else if (n <= 128) sort_store_keys<2>(s_key, n, lane, order);
            else if (n <= 256) sort_store_keys<4>(s_key, n, lane, order);
            else sort_store_keys<8>(s_key, n, lane, order);
            if (lane == 0) order[0] = 1;
        }
    }
    if (KGP_ON && sub == 0 && live) {
        unsigned long long *pp = a.k1_prof + (size_t)env * MW_K1_PROF_SLOTS;
        pp[0] = tp[0] - tstart; pp[1] = tp[1] - tp[0]; for (int i = 2; i < 7; ++i) pp[i] = racc[i];
        pp[7] = __builtin_readcyclecounter() - tprev;
        pp[20] = rtot[0]; pp[21] = rtot[1]; pp[22] = rvert[0]; pp[23] = rvert[1];
        pp[24] = kp_clip; pp[25] = kp_emit; pp[26] = (unsigned long long)kp_nclip; pp[27] = 0; pp[28] = kp_e_shfl; pp[29] = kp_e_setup; pp[30] = kp_e_write;
        pp[8] = rt_start; pp[9] = __builtin_amdgcn_s_memrealtime();
        pp[10] = (unsigned long long)np; pp[11] = (unsigned long long)npd; pp[12] = (unsigned long long)count; pp[13] = (unsigned long long)n_rounds;
        pp[14] = (unsigned long long)kp_nocc; pp[15] = (unsigned long long)kp_nkept;
        pp[16] = ts[0] - tp[0]; pp[17] = ts[1] - ts[0]; pp[18] = ts[2] - ts[1]; pp[19] = ts[3] - ts[2];
    }
    if (ent_pre && my_mesh_j >= 0 && live) {
        // the mesh's first draw id: the records drawn before it plus the mesh triangles drawn before it
        const int nb = total_boxes < 64 ? total_boxes : 64;
        const int end_pos = marker ? pos[nb] : count;
        const int p0 = my_boxes_before < nb ? pos[my_boxes_before] : end_pos;
        hdr[MW_HDR_MESH + MW_HDR_MESH_STRIDE * my_mesh_j + 1] = __int_as_float(p0 + my_tris_before);
    }
    if (sub == 0 && live) {
        a.nvis[env] = count < a.max_vis ? count : a.max_vis;
        hdr[0] = (float)a.light[(size_t)0 * a.N + env]; hdr[1] = (float)a.light[(size_t)1 * a.N + env]; hdr[2] = (float)a.light[(size_t)2 * a.N + env];
        hdr[3] = __int_as_float(total_meshes);
#pragma unroll
        for (int i = 0; i < 3; ++i) { hdr[4 + i] = f.l_amb[i]; hdr[8 + i] = f.l_dif[i]; }
        const int nb = total_boxes < 64 ? total_boxes : 64;
        const int end_pos = marker ? pos[nb] : count;
        if (!ent_pre) {
            for (int j = 0; j < total_meshes; ++j) {
                float *m = hdr + MW_HDR_MESH + MW_HDR_MESH_STRIDE * j;
                const int before = __float_as_int(m[1]);
                const int p0 = before < nb ? pos[before] : end_pos;
                m[1] = __int_as_float(p0 + __float_as_int(m[25]));
            }
        }
        if (count + total_mesh_tris >= 0xFFF0) atomicOr(a.status, MW_ST_VIS_OVERFLOW);      // 16-bit draw ids
        if (a.ent_list && total_meshes > 0) {
            // the work list of the mesh entity kernel: balls before keys (a workgroup per entity; the long ones start first)
            const int nbig = __popc(big_mask);
            const int xl = env % a.n_xcc;        // the XCD whose workgroups draw this env's entities
            int ib = nbig ? atomicAdd(a.ent_list_n + MW_CNT_LONG + xl, nbig) : 0;
            int is = total_meshes - nbig ? atomicAdd(a.ent_list_n + MW_CNT_SHORT + xl, total_meshes - nbig) : 0;
            uint32_t *l_long = a.ent_list + (size_t)xl * a.ent_list_cap, *l_short = a.ent_list + (size_t)(8 + xl) * a.ent_list_cap;
            for (int j = 0; j < total_meshes; ++j) {
                const uint32_t item = (uint32_t)env | ((uint32_t)j << 24);
                if ((big_mask >> j) & 1u) { if (ib < a.ent_list_cap) l_long[ib] = item; ++ib; }
                else { if (is < a.ent_list_cap) l_short[is] = item; ++is; }
            }
        }
        // the step's pending removal: the picked-up object leaves the entity list after its last frame
        // (pickupobjects.py:86-88); CollectHealth's consumed kit respawns instead, with draws from the env's stream:
        // mw_collect_respawn_kernel, launched behind this one
        const int rs = a.pending_remove[env];
        if (rs >= 0 && !proxy && a.task != MW_TASK_COLLECT) {
            a.ekind[(size_t)rs * a.N + env] = MW_ENT_NONE;
            a.pending_remove[env] = -1;
        }
    }
    if (a.tile_list) {
        // the mesh tiles' work list: every tile inside a drawn mesh entity's rectangle once, the group's lanes side by side
        int total;
        const int cnt = live ? __popc(tile_mask) : 0, excl = group_excl_scan(cnt, sub, L, total);
        int o = 0;
        const int xl = env % a.n_xcc;
        if (sub == 0 && total > 0) o = atomicAdd(a.ent_list_n + MW_CNT_TILES + xl, total);
        o = __shfl(o, 0, L) + excl;
        uint32_t *list = a.tile_list + (size_t)xl * a.tile_list_cap;
        for (uint32_t m = live ? tile_mask : 0u; m; m &= m - 1u, ++o)
            if (o < a.tile_list_cap) list[o] = (uint32_t)env | ((uint32_t)(sub + (__ffs((int)m) - 1) * L) << 24);
    }
}

#ifndef MW_GEOM_OCC
#define MW_GEOM_OCC
#endif
extern "C" __global__ __launch_bounds__(64) MW_GEOM_OCC void mw_geom_kernel(MwArgs a, int view_flags, int S, int L, int n_env) { geom_body<false, 8>(a, view_flags, S, L, n_env); }
extern "C" __global__ __launch_bounds__(64) MW_GEOM_OCC void mw_geom_big_kernel(MwArgs a, int view_flags, int S, int L, int n_env) { geom_body<true, 8>(a, view_flags, S, L, n_env); }
// ... for frame buffers with 1, 4 or 16 samples per pixel
extern "C" __global__ __launch_bounds__(64) void mw_geom_any_kernel(MwArgs a, int view_flags, int S, int L, int n_env) { geom_body<false, 0>(a, view_flags, S, L, n_env); }
extern "C" __global__ __launch_bounds__(64) void mw_geom_big_any_kernel(MwArgs a, int view_flags, int S, int L, int n_env) { geom_body<true, 0>(a, view_flags, S, L, n_env); }

// mw_selftest_sort: the visiting order's sort on keys of the caller's (tests/test_gpu_numerics.py): block b sorts the
// n[b] <= 512 keys at keys + 512 b into order + 513 b (order[0] unused, then the keys' low halves in ascending key order)
extern "C" __global__ __launch_bounds__(64) void mw_selftest_sort_kernel(const uint32_t *keys, const int32_t *n, uint16_t *order)
{
    __shared__ uint32_t s_key[MW_ORDER_CAP];
    const int lane = threadIdx.x, cnt = n[blockIdx.x];
    for (int i = lane; i < cnt; i += 64) s_key[i] = keys[(size_t)blockIdx.x * MW_ORDER_CAP + i];
    __syncthreads();
    uint16_t *out = order + (size_t)blockIdx.x * (MW_ORDER_CAP + 1);
    if (cnt <= 64) sort_store_keys<1>(s_key, cnt, lane, out);
    else if (cnt <= 128) sort_store_keys<2>(s_key, cnt, lane, out);
    else if (cnt <= 256) sort_store_keys<4>(s_key, cnt, lane, out);
    else sort_store_keys<8>(s_key, cnt, lane, out);
}

extern "C" int mw_selftest_sort(const uint32_t *host_keys /*[blocks][512]*/, const int32_t *host_n /*[blocks]*/, int32_t blocks, uint16_t *host_order /*[blocks][513]*/)
{
    uint32_t *d_keys = nullptr; int32_t *d_n = nullptr; uint16_t *d_order = nullptr;
    const size_t kb = (size_t)blocks * MW_ORDER_CAP * 4, ob = (size_t)blocks * (MW_ORDER_CAP + 1) * 2;
    if (blocks <= 0 || hipMalloc((void **)&d_keys, kb) != hipSuccess || hipMalloc((void **)&d_n, (size_t)blocks * 4) != hipSuccess ||
        hipMalloc((void **)&d_order, ob) != hipSuccess) return -1;
    (void)hipMemcpy(d_keys, host_keys, kb, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_n, host_n, (size_t)blocks * 4, hipMemcpyHostToDevice);
    (void)hipMemset(d_order, 0, ob);
    hipLaunchKernelGGL(mw_selftest_sort_kernel, dim3(blocks), dim3(64), 0, 0, d_keys, d_n, d_order);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    (void)hipMemcpy(host_order, d_order, ob, hipMemcpyDeviceToHost);
    (void)hipFree(d_keys); (void)hipFree(d_n); (void)hipFree(d_order);
    return rc;
}
