// K1 — per-env step + primitive setup.  One 64-lane wavefront per environment.
//
// Replaces, per env and per step (reference file:line):
//   MiniWorldEnv.step / move_agent / turn_agent / _get_carry_pos   miniworld.py:606-730
//   MiniWorldEnv.intersect + intersect_circle_segs                 miniworld.py:937-963, math.py:30-62
//   near / _reward + env rules                                     miniworld.py:965-975,1012-1017; hallway.py:67-74; pickupobjects.py:83-95
//   Agent.cam_pos / cam_dir, gluLookAt, gluPerspective             entity.py:476-503; miniworld.py:1198-1219
//   the fixed-function transform + lighting of every primitive     miniworld.py:401-434,1019-1077; entity.py:150-161,409-432
// and leaves, for the raster kernel (K2), a draw-ordered list of front-facing on-screen
// primitives per env: a 64-dword raster record (edge functions, per-sample thresholds,
// depth plane) read by K2 through scalar loads, and a 16-dword shade record.
//
// Lanes cooperate: collision segments and entities are tested one per lane (ballot),
// polygons are set up one per lane and compacted in draw order with ballot + popcount.
// All double-precision dynamics follow numpy's evaluation order;
// all float32 raster setup follows DESIGN.md section 3 rules R1-R11.
#include "mw_device.h"
#include "mw_math.h"
#include "mw_rng.h"
#include "mw_gen.h"

#ifndef MW_SORT_VIS
#define MW_SORT_VIS 0       // 1: also emit the depth-sorted visiting order of big scenes (mw_setup_sort*.hip)
#endif
#define MW_SORT_CAP 1024    // polygons sorted per env (their packed sort keys sit in 8 KiB of LDS)

namespace {

constexpr double kPi = 3.14159265358979323846;

struct HV { float hx, hy, hw, cz; };

struct Cam {
    float m[3][4];
    float p00, p11, p22, p23, halfw, halfh;
    float p03, p13;         // orthographic (top view) only
    int ortho;
    float L[3], amb[3], lcol[3];
};

// R5: D3D standard 8x pattern, offsets from the pixel centre in pixels
__constant__ float kSampleDx[8] = {0.0625f, -0.0625f, 0.3125f, -0.1875f, -0.3125f, -0.4375f, 0.1875f, 0.4375f};
__constant__ float kSampleDy[8] = {-0.1875f, 0.1875f, 0.0625f, -0.3125f, 0.3125f, -0.0625f, 0.4375f, -0.4375f};

// drawBox vertex selectors and normals, in the order of opengl.py:460-503
__constant__ unsigned char kBoxSel[6][4] = {
    // bit0: x max, bit1: y max, bit2: z max
    {7, 6, 4, 5}, {2, 3, 1, 0}, {6, 2, 0, 4}, {3, 7, 5, 1}, {7, 3, 2, 6}, {1, 5, 4, 0}};
__constant__ float kBoxN[6][3] = {{0, 0, 1}, {0, 0, -1}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}};

__device__ inline uint64_t ballot(bool p) { return __ballot(p); }

// wave-uniform values computed on the VALU are moved to SGPRs so they do not occupy a VGPR each
__device__ inline float uni(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ inline int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline double uni(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---------------------------------------------------------------- dynamics (f64)

struct StepCtx {
    const MwArgs &a;
    int env, lane, set;
    double px, py, pz, dir;        // agent
    double cam_height;
    int carry;                     // slot the agent carries, -1 none
    int live;                      // slot whose pos/dir live in cpos/cdir this step, -1 none
    double cpos[3], cdir;
};

__device__ inline double ent_pos(const StepCtx &c, int slot, int comp)
{
    if (slot == c.live) return c.cpos[comp];
    return c.a.epos[((size_t)comp * c.a.E + slot) * c.a.N + c.env];
}

__device__ inline double ent_geom(const MwArgs &a, int env, int slot, int k)
{
    return a.egeom[((size_t)k * a.E + slot) * a.N + env];
}

// MiniWorldEnv.intersect (miniworld.py:937-963): 0 none, -1 wall, 1+slot entity, 1+E agent.
// Every lane passes the same arguments; segments / entities are spread over the lanes.
__device__ int intersect(const StepCtx &c, int self_slot, double x, double z, double radius)
{
    const MwArgs &a = c.a;
    const double *segs = a.segs + (size_t)c.set * a.max_segs * 4;
    const int ns = a.nsegs[c.set];
    bool hit = false;
    for (int i = c.lane; i < ns; i += 64) {
        const double sax = segs[i * 4 + 0], saz = segs[i * 4 + 1], sbx = segs[i * 4 + 2], sbz = segs[i * 4 + 3];
        const double abx = sbx - sax, abz = sbz - saz;
        const double apx = x - sax, apz = z - saz;
        const double dap = apx * abx + apz * abz;
        const double dab = abx * abx + abz * abz;
        double t = dap / dab;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double cx = sax + t * abx, cz = saz + t * abz;
        const double dx = cx - x, dz = cz - z;
        hit |= sqrt(dx * dx + dz * dz) < radius;
    }
    if (ballot(hit)) return -1;
    for (int base = 0; base < a.E; base += 64) {
        const int slot = base + c.lane;
        bool h = false;
        if (slot < a.E && slot != self_slot && a.ekind[(size_t)slot * a.N + c.env] != MW_ENT_NONE) {
            const double dx = ent_pos(c, slot, 0) - x, dz = ent_pos(c, slot, 2) - z;
            h = sqrt(dx * dx + dz * dz) < radius + ent_geom(a, c.env, slot, 7);
        }
        const uint64_t m = ballot(h);
        if (m) return 1 + base + (__ffsll((unsigned long long)m) - 1);
    }
    if (self_slot >= 0) {
        const double dx = c.px - x, dz = c.pz - z;
        if (sqrt(dx * dx + dz * dz) < radius + a.agent_radius) return 1 + a.E;
    }
    return 0;
}

// _get_carry_pos (miniworld.py:606-618)
__device__ inline void carry_pos(const StepCtx &c, int slot, double ax, double ay, double az, double dvx,
                                 double dvz, double out[3])
{
    const double dist = c.a.agent_radius + ent_geom(c.a, c.env, slot, 7) + c.a.max_forward_step;
    out[0] = ax + dvx * 1.05 * dist;
    out[1] = ay + 0.0 * 1.05 * dist;
    out[2] = az + dvz * 1.05 * dist;
    const double y = c.cam_height - ent_geom(c.a, c.env, slot, 8) - 0.3;
    out[1] = out[1] + 1.0 * (y > 0.0 ? y : 0.0);
}

__device__ void move_agent(StepCtx &c, double fwd_dist, double fwd_drift)
{
    const mw::SinCos sc = mw::sincos_det(c.dir);
    const double dvx = sc.c, dvz = -sc.s, rvx = sc.s, rvz = sc.c;
    const double nx = c.px + dvx * fwd_dist + rvx * fwd_drift;
    const double ny = c.py + 0.0 * fwd_dist + 0.0 * fwd_drift;
    const double nz = c.pz + dvz * fwd_dist + rvz * fwd_drift;
    if (intersect(c, -1, nx, nz, c.a.agent_radius)) return;
    if (c.carry >= 0) {
        double cp[3];
        carry_pos(c, c.carry, nx, ny, nz, dvx, dvz, cp);
        if (intersect(c, c.carry, cp[0], cp[2], ent_geom(c.a, c.env, c.carry, 7))) return;
        c.cpos[0] = cp[0]; c.cpos[1] = cp[1]; c.cpos[2] = cp[2];
    }
    c.px = nx; c.py = ny; c.pz = nz;
}

__device__ void turn_agent(StepCtx &c, double turn_deg)
{
    const double turn = turn_deg * (kPi / 180.0);
    const double orig = c.dir;
    c.dir = c.dir + turn;
    if (c.carry >= 0) {
        const mw::SinCos sc = mw::sincos_det(c.dir);
        double cp[3];
        carry_pos(c, c.carry, c.px, c.py, c.pz, sc.c, -sc.s, cp);
        if (intersect(c, c.carry, cp[0], cp[2], ent_geom(c.a, c.env, c.carry, 7))) {
            c.dir = orig;
            return;
        }
        c.cpos[0] = cp[0]; c.cpos[1] = cp[1]; c.cpos[2] = cp[2];
        c.cdir = c.dir;
    }
}

// ---------------------------------------------------------------- camera (R1, R2, R10)

__device__ void build_camera(const MwArgs &a, int env, double px, double py, double pz, double dir, Cam &cam,
                             float sky[3], bool top_view)
{
    cam.ortho = 0; cam.p03 = 0.0f; cam.p13 = 0.0f;
    cam.halfw = (float)a.W * 0.5f;
    cam.halfh = (float)a.H * 0.5f;
    if (top_view) {
        // render_top_view (miniworld.py:1108-1160): extents +-1 m widened to the buffer's aspect,
        // glOrtho(min_x, max_x, -max_z, -min_z, -100, 100), modelview (x, y, z) -> (x, -z, y)
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
        const double l = min_x, r = max_x, b = -max_z, t = -min_z, n = -100.0, f = 100.0;
        cam.ortho = 1;
        cam.p00 = (float)(2.0 / (r - l)); cam.p03 = (float)(-(r + l) / (r - l));
        cam.p11 = (float)(2.0 / (t - b)); cam.p13 = (float)(-(t + b) / (t - b));
        cam.p22 = (float)(-2.0 / (f - n)); cam.p23 = (float)(-(f + n) / (f - n));
        cam.m[0][0] = 1; cam.m[0][1] = 0; cam.m[0][2] = 0; cam.m[0][3] = 0;
        cam.m[1][0] = 0; cam.m[1][1] = 0; cam.m[1][2] = -1; cam.m[1][3] = 0;
        cam.m[2][0] = 0; cam.m[2][1] = 1; cam.m[2][2] = 0; cam.m[2][3] = 0;
    } else {
    const double cam_height = a.cam[(size_t)0 * a.N + env], fwd_disp = a.cam[(size_t)1 * a.N + env];
    const double pitch_deg = a.cam[(size_t)2 * a.N + env], fov_y = a.cam[(size_t)3 * a.N + env];
    const mw::SinCos hd = mw::sincos_det(dir / 2.0);
    const double ya = hd.c, yc = -1.0 * hd.s;
    const double ry00 = ya * ya - yc * yc;
    const double ry02 = 2.0 * (ya * yc);
    const double ry11 = ya * ya + yc * yc;
    const double pitch = pitch_deg * kPi / 180.0;
    const mw::SinCos hp = mw::sincos_det(pitch / 2.0);
    const double za = hp.c, zd = -1.0 * hp.s;
    const double rz00 = za * za - zd * zd;
    const double rz01 = 2.0 * (0.0 - za * zd);
    const double eye[3] = {px + fwd_disp * ry00, py + cam_height * ry11, pz + fwd_disp * ry02};
    const double cd[3] = {rz00 * ry00, rz01 * ry11, rz00 * ry02};
    const double at[3] = {eye[0] + cd[0], eye[1] + cd[1], eye[2] + cd[2]};
    double F[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
    const double fl = sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]);
    F[0] /= fl; F[1] /= fl; F[2] /= fl;
    double s[3] = {-F[2], 0.0, F[0]};
    const double sl = sqrt(s[0] * s[0] + s[2] * s[2]);
    s[0] /= sl; s[2] /= sl;
    const double u[3] = {s[1] * F[2] - s[2] * F[1], s[2] * F[0] - s[0] * F[2], s[0] * F[1] - s[1] * F[0]};
    const double R[3][3] = {{s[0], s[1], s[2]}, {u[0], u[1], u[2]}, {-F[0], -F[1], -F[2]}};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        cam.m[i][0] = (float)R[i][0];
        cam.m[i][1] = (float)R[i][1];
        cam.m[i][2] = (float)R[i][2];
        cam.m[i][3] = (float)(-(R[i][0] * eye[0] + R[i][1] * eye[1] + R[i][2] * eye[2]));
    }
    const double half = fov_y / 2.0 * kPi / 180.0;
    const mw::SinCos hf = mw::sincos_det(half);
    const double cot = hf.c / hf.s;
    const double aspect = (double)a.W / (double)a.H;
    const double zn = 0.04, zf = 100.0;
    cam.p00 = (float)(cot / aspect);
    cam.p11 = (float)cot;
    cam.p22 = (float)(-(zf + zn) / (zf - zn));
    cam.p23 = (float)(-2.0 * zn * zf / (zf - zn));
    }
    float lp[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) lp[i] = (float)(a.light[(size_t)(3 + i) * a.N + env] + 1.0);
    const float ll = sqrtf(fmaf(lp[2], lp[2], fmaf(lp[1], lp[1], lp[0] * lp[0])));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        cam.L[i] = lp[i] / ll;
        cam.amb[i] = 0.2f + (float)a.light[(size_t)(9 + i) * a.N + env];
        cam.lcol[i] = (float)a.light[(size_t)(6 + i) * a.N + env];
        sky[i] = (float)a.light[(size_t)i * a.N + env];
    }
}

__device__ inline HV xform(const Cam &c, float x, float y, float z)
{
    const float ex = fmaf(c.m[0][0], x, fmaf(c.m[0][1], y, fmaf(c.m[0][2], z, c.m[0][3])));
    const float ey = fmaf(c.m[1][0], x, fmaf(c.m[1][1], y, fmaf(c.m[1][2], z, c.m[1][3])));
    const float ez = fmaf(c.m[2][0], x, fmaf(c.m[2][1], y, fmaf(c.m[2][2], z, c.m[2][3])));
    float cx = c.p00 * ex, cy = c.p11 * ey, cw = -ez;
    if (c.ortho) {              // glOrtho: translation terms, w = 1
        cx = fmaf(c.p00, ex, c.p03);
        cy = fmaf(c.p11, ey, c.p13);
        cw = 1.0f;
    }
    HV h;
    h.cz = fmaf(c.p22, ez, c.p23);
    h.hx = (cx + cw) * c.halfw;
    h.hy = (cw - cy) * c.halfh;
    h.hw = cw;
    return h;
}

__device__ inline void light(const Cam &c, const float n[3], const float base[3], float out[3])
{
    const float ndl = fmaf(n[2], c.L[2], fmaf(n[1], c.L[1], n[0] * c.L[0]));
    const float d = ndl > 0.0f ? ndl : 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float k = fmaf(c.lcol[i], d, c.amb[i]);
        const float v = base[i] * k;
        out[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

// edge function of a->b: the homogeneous cross product b x a (R4)
__device__ inline void edge_coef(const HV &a, const HV &b, float &ea, float &eb, float &ec)
{
    ea = b.hy * a.hw - b.hw * a.hy;
    eb = b.hw * a.hx - b.hx * a.hw;
    ec = b.hx * a.hy - b.hy * a.hx;
}

// largest float strictly below x: E >= x  <=>  E > below(x)   (top-left tie rule folded in)
__device__ inline float below(float x)
{
    if (x == 0.0f) return __uint_as_float(0x80000001u);
    const uint32_t b = __float_as_uint(x);
    return __uint_as_float(x > 0.0f ? b - 1u : b + 1u);
}

// R4 is split in two so that nothing big stays live across the ordered compaction: cull_poly()
// decides visibility (orientation + conservative tile bounds), write_poly() derives the records of
// a visible polygon and stores them straight into the env's lists.
struct PolyGeom {
    float ga[3], gb[3], gc[3];      // interpolation basis G0 = edge(1->2), G1 = edge(2->0), G2 = edge(0->1)
    float D;
    uint32_t bbox;                  // tile bounds tx0 | tx1<<8 | ty0<<16 | ty1<<24
};

__device__ __forceinline__ bool cull_poly(const MwArgs &a, const HV h[4], int nv, PolyGeom &g)
{
    edge_coef(h[1], h[2], g.ga[0], g.gb[0], g.gc[0]);
    edge_coef(h[2], h[0], g.ga[1], g.gb[1], g.gc[1]);
    edge_coef(h[0], h[1], g.ga[2], g.gb[2], g.gc[2]);
    g.D = fmaf(h[0].hx, g.ga[0], fmaf(h[0].hy, g.gb[0], h[0].hw * g.gc[0]));
    if (!(g.D > 0.0f)) return false;          // back-face cull (miniworld.py:512)
    // conservative screen bounds: is anything of the polygon on the screen at all?  The polygon is clipped against
    // w >= 0.01 (well in front of the 0.04 near plane) only to bound its projection; coverage itself never clips
    // (R4).  The bounds feed this yes / no and a tile range, both with a margin of a pixel, so the perspective
    // divides are hardware reciprocals (1 ulp) instead of IEEE divisions (a dozen dependent instructions each).
    const float wc = 0.01f;
    float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
    bool some = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nv) {
            const HV p = h[k];
            const HV q = (k + 1 == nv || k == 3) ? h[0] : h[k < 3 ? k + 1 : 0];
            const bool pin = p.hw >= wc, qin = q.hw >= wc;
            if (pin) {
                const float iw = __builtin_amdgcn_rcpf(p.hw);
                const float X = p.hx * iw, Y = p.hy * iw;
                xmin = fminf(xmin, X); xmax = fmaxf(xmax, X); ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
                some = true;
            }
            if (pin != qin) {
                const float t = (wc - p.hw) * __builtin_amdgcn_rcpf(q.hw - p.hw);
                const float X = fmaf(t, q.hx - p.hx, p.hx) * 100.0f, Y = fmaf(t, q.hy - p.hy, p.hy) * 100.0f;
                xmin = fminf(xmin, X); xmax = fmaxf(xmax, X); ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
                some = true;
            }
        }
    }
    if (!some) return false;                         // entirely behind the eye
    // generous margin: the clipped outline is computed in float and huge coordinates lose precision
    const float mx = 1.0f + 1e-3f * fmaxf(fabsf(xmin), fabsf(xmax)), my = 1.0f + 1e-3f * fmaxf(fabsf(ymin), fabsf(ymax));
    if (xmax + mx < 0.0f || ymax + my < 0.0f || xmin - mx > (float)a.W || ymin - my > (float)a.H) return false;
    // tile range for the large views (mw_raster_mesh.hip::view_tile_body skips primitives by it)
    const float fx0 = fminf(fmaxf(floorf(xmin - mx), 0.0f), (float)(a.W - 1));
    const float fx1 = fminf(fmaxf(floorf(xmax + mx), 0.0f), (float)(a.W - 1));
    const float fy0 = fminf(fmaxf(floorf(ymin - my), 0.0f), (float)(a.H - 1));
    const float fy1 = fminf(fmaxf(floorf(ymax + my), 0.0f), (float)(a.H - 1));
    g.bbox = (uint32_t)((int)fx0 / MW_TILE_W) | ((uint32_t)((int)fx1 / MW_TILE_W) << 8) |
             ((uint32_t)((int)fy0 / MW_TILE_H) << 16) | ((uint32_t)((int)fy1 / MW_TILE_H) << 24);
    return true;
}

__device__ __forceinline__ void write_poly(const MwArgs &a, int env, int idx, uint32_t draw_id, const HV h[4], int nv,
                           const PolyGeom &g, const float uv[3][2], const float col[3], int tex, unsigned long long *s_zmin)
{
    float4 *rr = reinterpret_cast<float4 *>(a.rec_raster + ((size_t)env * a.max_vis + idx) * MW_RASTER_REC);
    float4 *sr = reinterpret_cast<float4 *>(a.rec_shade + ((size_t)env * a.max_vis + idx) * MW_SHADE_REC);
    float4 *cr = reinterpret_cast<float4 *>(a.rec_cull + ((size_t)env * a.max_vis + idx) * MW_CULL_REC);
    float ea[4], eb[4], ec[4], tmaxv[4], tminv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ea[k] = 0.0f; eb[k] = 0.0f; ec[k] = 1.0f;       // always-true edge for triangles
        if (k < nv) {
            const HV nxt = (k + 1 == nv || k == 3) ? h[0] : h[k < 3 ? k + 1 : 0];
            edge_coef(h[k], nxt, ea[k], eb[k], ec[k]);
        }
        const bool tl = (ea[k] > 0.0f) || (ea[k] == 0.0f && eb[k] > 0.0f);
        float thr[8];
        float tmax = -1e30f, tmin = 1e30f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float t = -fmaf(ea[k], kSampleDx[s], eb[k] * kSampleDy[s]);
            thr[s] = (k < nv && tl) ? below(t) : t;      // top-left tie rule folded into the threshold
            tmax = fmaxf(tmax, thr[s]);
            tmin = fminf(tmin, thr[s]);
        }
        rr[4 + 2 * k] = make_float4(thr[0], thr[1], thr[2], thr[3]);
        rr[5 + 2 * k] = make_float4(thr[4], thr[5], thr[6], thr[7]);
        tmaxv[k] = tmax;            // E > tmax  =>  every sample of the pixel is inside edge k
        tminv[k] = tmin;            // E <= tmin =>  no sample of the pixel is inside edge k
    }
    const float invD = 1.0f / g.D;
    const float ta = fmaf(h[2].cz, g.ga[2], fmaf(h[1].cz, g.ga[1], h[0].cz * g.ga[0]));
    const float tb = fmaf(h[2].cz, g.gb[2], fmaf(h[1].cz, g.gb[1], h[0].cz * g.gb[0]));
    const float tc = fmaf(h[2].cz, g.gc[2], fmaf(h[1].cz, g.gc[1], h[0].cz * g.gc[0]));
    const float zx = (ta * invD) * 0.5f, zy = (tb * invD) * 0.5f, zc = fmaf(tc * invD, 0.5f, 0.5f);
    float zo[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) zo[s] = fmaf(zx, kSampleDx[s], zy * kSampleDy[s]);
    // may_clip: can a sample inside this polygon fail the near / far test of R6?  Conservative:
    // far  - some vertex in front of the eye is deeper than 99 m;
    // near - some vertex is nearer than 5 cm AND the polygon's 1/w plane exceeds 1/0.05 at one of
    //        the screen corners (1/w is linear on screen, so its maximum is at a corner).
    const float Wa = (g.ga[0] + g.ga[1]) + g.ga[2], Wb = (g.gb[0] + g.gb[1]) + g.gb[2], Wc = (g.gc[0] + g.gc[1]) + g.gc[2];
    bool may_clip = false;
    {
        float wmin = 1e30f, wmax = -1e30f;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) { wmin = fminf(wmin, h[k].hw); wmax = fmaxf(wmax, h[k].hw); }
        may_clip |= !(wmax <= 99.0f);
        if (!(wmin >= 0.05f)) {
            const float na = Wa * invD, nb = Wb * invD, nc = Wc * invD;
            const float fw = (float)a.W, fh = (float)a.H;
            const float c00 = nc, c10 = fmaf(na, fw, nc), c01 = fmaf(nb, fh, nc), c11 = fmaf(na, fw, fmaf(nb, fh, nc));
            may_clip |= !(fmaxf(fmaxf(c00, c10), fmaxf(c01, c11)) <= 20.0f);
        }
    }
    const float flag = __uint_as_float(may_clip ? 1u : 0u);
    const float4 e0 = make_float4(ea[0], ea[1], ea[2], ea[3]), e1 = make_float4(eb[0], eb[1], eb[2], eb[3]),
                 e2 = make_float4(ec[0], ec[1], ec[2], ec[3]);
    rr[0] = e0; rr[1] = e1; rr[2] = e2;
    rr[3] = make_float4(zx, zy, zc, __uint_as_float(g.bbox));
    rr[12] = make_float4(zo[0], zo[1], zo[2], zo[3]);
    rr[13] = make_float4(zo[4], zo[5], zo[6], zo[7]);
    rr[14] = make_float4(flag, tmaxv[0], tmaxv[1], tmaxv[2]);
    rr[15] = make_float4(tmaxv[3], __uint_as_float(draw_id), 0.0f, 0.0f);       // draw id = list index + mesh triangles drawn before
    cr[0] = e0; cr[1] = e1; cr[2] = e2;
    cr[3] = make_float4(tminv[0], tminv[1], tminv[2], tminv[3]);
    cr[4] = make_float4(tmaxv[0], tmaxv[1], tmaxv[2], tmaxv[3]);
    float zmin = 0.0f;
#if MW_SORT_VIS
    {
        // Lower bound of every depth KEY the raster kernel can compute for this polygon: its depth plane, evaluated
        // exactly like there (R6: fmaf(zx, Xc, fmaf(zy, Yc, zc)) + zo[s], every step monotone in Xc and Yc, rounding
        // included), at the corner of the polygon's tile bounds where it is smallest, plus the smallest sample offset.
        // (The minimum over the VERTEX depths is not such a bound: the plane coefficients of a thin or grazing
        // polygon carry rounding error, and the key computed at a sample can fall below the depth of every vertex —
        // the full-size parity test caught single samples of far polygons, seen through cracks, lost that way.)
        // 0 = nearest possible when a vertex is behind the eye (bounds are the whole screen then).
        // The raster kernel visits polygons in ascending order of this bound and stops once a tile's farthest
        // stored sample is nearer than the next bound.
        bool allpos = true;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < nv) allpos &= h[k].hw > 0.0f;
        const float Xlo = (float)((g.bbox & 255u) * MW_TILE_W) + 0.5f, Xhi = (float)(((g.bbox >> 8) & 255u) * MW_TILE_W + (MW_TILE_W - 1)) + 0.5f;
        const float Ylo = (float)(((g.bbox >> 16) & 255u) * MW_TILE_H) + 0.5f, Yhi = (float)((g.bbox >> 24) * MW_TILE_H + (MW_TILE_H - 1)) + 0.5f;
        const float zcmin = fmaf(zx, zx > 0.0f ? Xlo : Xhi, fmaf(zy, zy > 0.0f ? Ylo : Yhi, zc));
        float zomin = zo[0];
#pragma unroll
        for (int s = 1; s < 8; ++s) zomin = fminf(zomin, zo[s]);
        zmin = allpos ? zcmin + zomin : 0.0f;
        if (!(zmin >= 0.0f)) zmin = 0.0f;
        // sort key: the bound's bit pattern (monotone for non-negative floats), ties broken by the list index
        if (idx < MW_SORT_CAP) s_zmin[idx] = ((unsigned long long)__float_as_uint(zmin) << 16) | (unsigned long long)idx;
    }
#endif
    cr[5] = make_float4(flag, zmin, 0.0f, 0.0f);
    // shade record: attribute planes (unnormalised), face colour, texture, depth plane again
    float U[3] = {0, 0, 0}, V[3] = {0, 0, 0};
    if (tex >= 0) {
        U[0] = fmaf(uv[2][0], g.ga[2], fmaf(uv[1][0], g.ga[1], uv[0][0] * g.ga[0]));
        U[1] = fmaf(uv[2][0], g.gb[2], fmaf(uv[1][0], g.gb[1], uv[0][0] * g.gb[0]));
        U[2] = fmaf(uv[2][0], g.gc[2], fmaf(uv[1][0], g.gc[1], uv[0][0] * g.gc[0]));
        V[0] = fmaf(uv[2][1], g.ga[2], fmaf(uv[1][1], g.ga[1], uv[0][1] * g.ga[0]));
        V[1] = fmaf(uv[2][1], g.gb[2], fmaf(uv[1][1], g.gb[1], uv[0][1] * g.gb[0]));
        V[2] = fmaf(uv[2][1], g.gc[2], fmaf(uv[1][1], g.gc[1], uv[0][1] * g.gc[0]));
    }
    sr[0] = make_float4(U[0], U[1], U[2], V[0]);
    sr[1] = make_float4(V[1], V[2], Wa, Wb);
    sr[2] = make_float4(Wc, col[0], col[1], col[2]);
    sr[3] = make_float4(__int_as_float(tex), 0.0f, 0.0f, 0.0f);
    sr[4] = make_float4(zx, zy, zc, 0.0f);
    sr[5] = make_float4(zo[0], zo[1], zo[2], zo[3]);
    sr[6] = make_float4(zo[4], zo[5], zo[6], zo[7]);
    sr[7] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ordered compaction: list index of this lane's primitive (valid if vis); advances count
__device__ inline int compact(int lane, bool vis, int &count)
{
    const uint64_t m = ballot(vis);
    const int idx = count + __popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
    count += __popcll((unsigned long long)m);
    return idx;
}

}  // namespace

// ---------------------------------------------------------------- the kernel

#ifndef MW_SETUP_KERNEL_NAME
#define MW_SETUP_KERNEL_NAME mw_step_setup_kernel
#endif
// MW_K1_WAVES wavefronts per env.  Small scenes: 1 (thousands of envs already fill the chip with one wave each).
// Big scenes (MW_SORT_VIS: a Maze has 510 polygons and the BASELINE batch is 1024 envs per GPU = one wave per
// SIMD): 4 — every wave runs the scalar part (physics, camera) redundantly, thread 0 alone writes state, the
// polygon batches are dealt round-robin to the waves with an ordered compaction across them, entities stay on
// wave 0, and all 256 threads sort.
#if MW_SORT_VIS
#define MW_K1_WAVES 4
#else
#define MW_K1_WAVES 1
#endif
extern "C" __global__ __launch_bounds__(64 * MW_K1_WAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void MW_SETUP_KERNEL_NAME(
    MwArgs a, int do_step, int view_flags, const int32_t *__restrict__ actions, float *__restrict__ reward,
    uint8_t *__restrict__ term, uint8_t *__restrict__ trunc)
{
    constexpr int KW = MW_K1_WAVES;
    __shared__ int s_cnt[2][KW];
    __shared__ unsigned char gen_ws[MW_GEN_WS_BYTES];
    // spare mode: blocks appended to the grid regenerate the spare worlds consumed in earlier steps, beside the step
    // itself (measured both ways: at the head of the grid they delay the env blocks more than they hide)
    if ((int)blockIdx.x >= a.N) {
        mw::refill_spares(a, (int)blockIdx.x - a.N, (int)threadIdx.x, gen_ws);
        return;
    }
#if MW_SORT_VIS
    __shared__ unsigned long long s_zmin_buf[MW_SORT_CAP];
    unsigned long long *s_zmin = s_zmin_buf;
#else
    unsigned long long *s_zmin = nullptr;
#endif
    const unsigned long long pt0 = a.k1_prof ? __builtin_readcyclecounter() : 0ull;
    const int env = a.env_base + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool writer = threadIdx.x == 0;       // the one thread that writes the env's state
    StepCtx c{a, env, lane, a.shared_geom ? 0 : env, 0, 0, 0, 0, 0, -1, -1, {0, 0, 0}, 0};
    c.px = a.ax[env]; c.py = a.ay[env]; c.pz = a.az[env]; c.dir = a.adir[env];
    c.cam_height = a.cam[env];
    c.carry = a.carry[env];
    c.live = -1;
    c.cpos[0] = c.cpos[1] = c.cpos[2] = 0.0; c.cdir = 0.0;
    if (c.carry >= 0) {
        const int k = c.carry;
        c.cpos[0] = ent_pos(c, k, 0); c.cpos[1] = ent_pos(c, k, 1); c.cpos[2] = ent_pos(c, k, 2);
        c.cdir = a.edir[(size_t)k * a.N + env];
        c.live = k;
    }
    int remove_slot = -1;
    int tm = 0, tr = 0;             // terminated / truncated, uniform over the wave

    if (do_step) {
        int step_count = a.step[env] + 1;
        int picked = a.picked[env];
        // the three per-step parameters (miniworld.py:677-680)
        double fwd_step = a.fwd.def, fwd_drift = a.drift.def, turn_step = a.turn.def;
        mw::Rng rng{};
        bool drew = false;
        if (a.step_override) {
            fwd_step = a.step_override[(size_t)env * 3 + 0];
            fwd_drift = a.step_override[(size_t)env * 3 + 1];
            turn_step = a.step_override[(size_t)env * 3 + 2];
        } else if (a.domain_rand) {
            rng = mw::rng_load(a.rng, a.N, env);
            fwd_step = mw::rng_uniform(rng, a.fwd.lo, a.fwd.hi);
            fwd_drift = mw::rng_uniform(rng, a.drift.lo, a.drift.hi);
            turn_step = mw::rng_uniform(rng, a.turn.lo, a.turn.hi);
            drew = true;            // stored below, once every wave of the env has read the old state
        }
        const int action = actions[env];
        switch (action) {
        case 2: move_agent(c, fwd_step, fwd_drift); break;
        case 3: move_agent(c, -fwd_step, fwd_drift); break;
        case 0: turn_agent(c, turn_step); break;
        case 1: turn_agent(c, -turn_step); break;
        case 4: {   // pickup (miniworld.py:695-702)
            const mw::SinCos sc = mw::sincos_det(c.dir);
            const double tx = c.px + sc.c * 1.5 * a.agent_radius;
            const double tz = c.pz + (-sc.s) * 1.5 * a.agent_radius;
            const int hit = intersect(c, -1, tx, tz, 1.2 * a.agent_radius);
            if (c.carry < 0 && hit > 0 && hit <= a.E && !a.estatic[(size_t)(hit - 1) * a.N + env]) {
                const int k = hit - 1;
                c.cpos[0] = ent_pos(c, k, 0); c.cpos[1] = ent_pos(c, k, 1); c.cpos[2] = ent_pos(c, k, 2);
                c.cdir = a.edir[(size_t)k * a.N + env];
                c.carry = k;
                c.live = k;
            }
            break;
        }
        case 5:     // drop (miniworld.py:705-708)
            if (c.carry >= 0) {
                c.cpos[1] = 0.0;
                c.carry = -1;       // the live copy is written back at the end of the step
            }
            break;
        default: break;
        }
        if (c.carry >= 0) {     // carried object follows (miniworld.py:711-714)
            const mw::SinCos sc = mw::sincos_det(c.dir);
            double cp[3];
            carry_pos(c, c.carry, c.px, c.py, c.pz, sc.c, -sc.s, cp);
            c.cpos[0] = cp[0]; c.cpos[1] = cp[1]; c.cpos[2] = cp[2];
            c.cdir = c.dir;
        }
        // reward / termination (miniworld.py:720-730 + env rule)
        double rew = 0.0;
        tr = step_count >= a.max_steps ? 1 : 0;
        if (a.task == MW_TASK_GOTO) {
            const int g = a.goal_ent;
            const double dx = ent_pos(c, g, 0) - c.px, dy = ent_pos(c, g, 1) - c.py, dz = ent_pos(c, g, 2) - c.pz;
            const double dist = sqrt(dx * dx + dy * dy + dz * dz);
            if (dist < ent_geom(a, env, g, 7) + a.agent_radius + 1.1 * a.max_forward_step) {
                rew += 1.0 - 0.2 * ((double)step_count / (double)a.max_steps);
                tm = 1;
            }
        } else if (a.task == MW_TASK_PUTNEXT) {
            if (c.carry < 0) {      // putnext.py:74-78
                const int g0 = a.goal_ent, g1 = a.goal_ent2;
                const double dx = ent_pos(c, g0, 0) - ent_pos(c, g1, 0), dy = ent_pos(c, g0, 1) - ent_pos(c, g1, 1),
                             dz = ent_pos(c, g0, 2) - ent_pos(c, g1, 2);
                const double dist = sqrt(dx * dx + dy * dy + dz * dz);
                if (dist < ent_geom(a, env, g0, 7) + ent_geom(a, env, g1, 7) + 1.1 * a.max_forward_step) {
                    rew += 1.0 - 0.2 * ((double)step_count / (double)a.max_steps);
                    tm = 1;
                }
            }
        } else if (a.task == MW_TASK_PICKUP) {
            if (c.carry >= 0) {
                remove_slot = c.carry;      // still drawn this frame (pickupobjects.py:86-88 runs after :717)
                picked += 1;
                rew = 1.0;
                if (picked == a.num_objs) tm = 1;
            }
        }
        if (KW > 1) __syncthreads();        // every wave has read the state it needs: thread 0 may now overwrite it
        if (writer) {
            if (drew) mw::rng_store(a.rng, a.N, env, rng);
            reward[env] = (float)rew;
            term[env] = (uint8_t)tm;
            trunc[env] = (uint8_t)tr;
            a.step[env] = step_count;
            a.picked[env] = picked;
        }
    }

    // ---- state write-back (agent + carried entity) ------------------------------
    bool regenerated = false;
    if (do_step && writer) {
        a.ax[env] = c.px; a.ay[env] = c.py; a.az[env] = c.pz; a.adir[env] = c.dir;
        if (c.live >= 0) {
            a.epos[((size_t)0 * a.E + c.live) * a.N + env] = c.cpos[0];
            a.epos[((size_t)1 * a.E + c.live) * a.N + env] = c.cpos[1];
            a.epos[((size_t)2 * a.E + c.live) * a.N + env] = c.cpos[2];
            a.edir[(size_t)c.live * a.N + env] = c.cdir;
        }
        a.carry[env] = remove_slot >= 0 ? -1 : c.carry;
    }
    if (do_step && a.autoreset == MW_AUTORESET_SAME_STEP && a.generator != MW_GEN_NONE) {
        // same-step auto-reset: the observation returned with done=1 is the first one of the
        // next episode (the reference leaves the reset to the caller, scripts/benchmark.py:36-37)
        regenerated = (tm | tr) != 0;
        if (regenerated) {
            if (a.spare) {
                // the next world was generated ahead (by a refill block of an earlier launch): claim it
                if (writer) s_cnt[0][0] = (int)atomicCAS(a.refill_mask + env, 1u, 3u);
                __syncthreads();
                const int old = s_cnt[0][0];
                if (old == 1) {
                    // the previous episode lasted one step and the refill has not run yet: generate in place
                    if (wave == 0) mw::generate_world(*a.gen_live, env, gen_ws, lane);
                } else {
                    if (old == 2) {     // a refill block of this very launch is on it
                        if (writer) while (__hip_atomic_load(a.refill_mask + env, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_amdgcn_s_sleep(16);
                        __syncthreads();
                    }
                    if (wave == 0) mw::take_spare(a, env, lane);
                }
                __threadfence();
                __syncthreads();
                if (writer) atomicExch(a.refill_mask + env, 1u);        // the spare is missing again
            } else if (wave == 0) {
                mw::generate_world(*a.gen_live, env, gen_ws, lane);
            }
            __syncthreads();
            c.px = a.ax[env]; c.py = a.ay[env]; c.pz = a.az[env]; c.dir = a.adir[env];
            c.carry = -1; c.live = -1;
            remove_slot = -1;
        }
    }

    const unsigned long long pt1 = a.k1_prof ? __builtin_readcyclecounter() : 0ull;
    // ---- camera + primitive setup -----------------------------------------------
    Cam cam;
    float sky[3];
    build_camera(a, env, c.px, c.py, c.pz, c.dir, cam, sky, (view_flags & 1) != 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cam.m[i][j] = uni(cam.m[i][j]);
        cam.L[i] = uni(cam.L[i]); cam.amb[i] = uni(cam.amb[i]); cam.lcol[i] = uni(cam.lcol[i]);
        sky[i] = uni(sky[i]);
    }
    cam.p00 = uni(cam.p00); cam.p11 = uni(cam.p11); cam.p22 = uni(cam.p22); cam.p23 = uni(cam.p23);
    cam.p03 = uni(cam.p03); cam.p13 = uni(cam.p13);
    c.px = uni(c.px); c.py = uni(c.py); c.pz = uni(c.pz); c.dir = uni(c.dir);
    c.cpos[0] = uni(c.cpos[0]); c.cpos[1] = uni(c.cpos[1]); c.cpos[2] = uni(c.cpos[2]); c.cdir = uni(c.cdir);
    const unsigned long long pt2 = a.k1_prof ? __builtin_readcyclecounter() : 0ull;
    float stale_n[3] = {0.0f, 1.0f, 0.0f};       // GL's current normal after the last draw (top-view agent marker)
    int count = 0;
    const mw_poly *polys = a.polys + (size_t)c.set * a.max_polys;
    const int np = a.npolys[c.set];
    for (int base = 0, round = 0; base < np; base += 64 * KW, ++round) {         // display list 1: rooms
        const int i = base + wave * 64 + lane;
        bool vis = false;
        HV h[4];
        PolyGeom g;
        mw_poly q;
        if (i < np) {
            q = polys[i];
            const bool ent_poly = (q.nv & MW_POLY_ENTITY) != 0;
            q.nv &= 0xFF;
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = xform(cam, q.v[k][0], q.v[k][1], q.v[k][2]);
            vis = cull_poly(a, h, q.nv, g) && !(ent_poly && (view_flags & 4));     // the queries draw rooms only
        }
        int idx;
        if (KW == 1) {
            idx = compact(lane, vis, count);
        } else {
            // ordered compaction across the waves of the env: wave w's batch comes after those of waves < w
            const uint64_t m = ballot(vis);
            if (lane == 0) s_cnt[round & 1][wave] = __popcll((unsigned long long)m);
            __syncthreads();
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < KW; ++w) {
                const int cw = s_cnt[round & 1][w];
                before += w < wave ? cw : 0;
                total += cw;
            }
            idx = count + before + __popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
            count += total;
        }
        if (vis) {
            if (idx < a.max_vis) {
                float col[3];
                light(cam, q.n, q.rgb, col);
                const float uv[3][2] = {{q.uv[0][0], q.uv[0][1]}, {q.uv[1][0], q.uv[1][1]}, {q.uv[2][0], q.uv[2][1]}};
                write_poly(a, env, idx, (uint32_t)idx, h, q.nv, g, uv, col, q.tex, s_zmin);
            } else {
                atomicOr(a.status, MW_ST_VIS_OVERFLOW);
            }
        }
    }
    if (np > 0) {
        const mw_poly &lastq = polys[np - 1];
        stale_n[0] = lastq.n[0]; stale_n[1] = lastq.n[1]; stale_n[2] = lastq.n[2];
    }
    // entities in draw order: static ones first, then dynamic (miniworld.py:1058-1060, 1075-1077).
    // Boxes become 6 polygons each (runs of up to 10 consecutive boxes share one 64-lane batch);
    // a mesh entity only reserves its range of draw ids and is described to the mesh raster kernel.
    const unsigned long long pt3 = a.k1_prof ? __builtin_readcyclecounter() : 0ull;
    int mesh_tris = 0, n_mesh = 0;
    float *hdr = a.envhdr + (size_t)env * MW_ENVHDR;
    if (KW == 1 || wave == 0) {     // entities (and the agent marker) are few: wave 0 of the env alone
    // kind / static flag of every slot, one slot per lane, as wave-uniform bit masks (max_ents <= 64)
    uint64_t box_m, mesh_m, frame_m, static_m;
    {
        int kind_l = MW_ENT_NONE, static_l = 0;
        if (lane < a.E) {
            kind_l = a.ekind[(size_t)lane * a.N + env];
            static_l = a.estatic[(size_t)lane * a.N + env];
        }
        box_m = ballot(kind_l == MW_ENT_BOX);
        mesh_m = ballot(kind_l == MW_ENT_MESH);
        frame_m = ballot(kind_l == MW_ENT_FRAME);
        static_m = ballot(static_l != 0);
    }
    // get_visible_ents (miniworld.py:1296-1313): instead of the entities themselves, an axis-aligned
    // 0.2 m proxy box per entity, in self.entities (= slot) order, tagged with its slot
    const bool proxy = (view_flags & 4) != 0;
    if (proxy) { box_m |= mesh_m | frame_m; mesh_m = 0ull; static_m = ~0ull; }
    for (int pass = 0; pass < 2; ++pass) {
        const uint64_t mine_m = pass == 0 ? static_m : ~static_m;
        const uint64_t mesh_mine = mesh_m & mine_m, box_mine = box_m & mine_m;
        int s0 = 0;
        while (s0 < a.E) {
            if ((mesh_mine >> s0) & 1ull) {
                const int mid = uni(a.emesh[(size_t)s0 * a.N + env]);
                const MwMeshDesc *mdp = a.mesh + mid;
                const int md_ntris = uni((int)mdp->ntris), md_first = uni((int)mdp->first), md_tex = uni((int)mdp->tex);
                // whole-entity frustum cull (perspective views): the bounding sphere of the scaled mesh about its
                // origin against the near and the four side planes, conservative — a skipped mesh has no pixel
                bool in_view = true;
                if (!cam.ortho) {
                    const float brad = __int_as_float(uni((int)mdp->bound_bits)) * (float)ent_geom(a, env, s0, 6) * 1.001f + 1e-3f;
                    const float wx = (float)ent_pos(c, s0, 0), wy = (float)ent_pos(c, s0, 1), wz = (float)ent_pos(c, s0, 2);
                    const float ex = fmaf(cam.m[0][0], wx, fmaf(cam.m[0][1], wy, fmaf(cam.m[0][2], wz, cam.m[0][3])));
                    const float ey = fmaf(cam.m[1][0], wx, fmaf(cam.m[1][1], wy, fmaf(cam.m[1][2], wz, cam.m[1][3])));
                    const float ez = fmaf(cam.m[2][0], wx, fmaf(cam.m[2][1], wy, fmaf(cam.m[2][2], wz, cam.m[2][3])));
                    const float w = -ez;
                    const float lx = sqrtf(fmaf(cam.p00, cam.p00, 1.0f)), ly = sqrtf(fmaf(cam.p11, cam.p11, 1.0f));
                    in_view = !(w + brad < 0.04f) && !(w - fabsf(cam.p00 * ex) < -(brad * lx)) &&
                              !(w - fabsf(cam.p11 * ey) < -(brad * ly));
                    in_view = uni((int)in_view) != 0;
                }
                if (!in_view) {
                    // nothing to draw; GL's current normal still ends up at the mesh's last one (below)
                } else if (n_mesh < MW_MAX_MESH_ENTS && count + mesh_tris + md_ntris < 0xFFF0) {
                    if (lane == 0) {
                        const double edir = (s0 == c.live) ? c.cdir : a.edir[(size_t)s0 * a.N + env];
                        const mw::SinCos sc = mw::sincos_det(edir);
                        float *m = hdr + MW_HDR_MESH + 12 * n_mesh;
                        m[0] = __int_as_float(s0);
                        m[1] = __int_as_float(count + mesh_tris);
                        m[2] = __int_as_float(md_ntris);
                        m[3] = __int_as_float(md_first);
                        m[4] = (float)sc.c; m[5] = (float)sc.s;
                        m[6] = (float)ent_geom(a, env, s0, 6);
                        m[7] = (float)ent_pos(c, s0, 0); m[8] = (float)ent_pos(c, s0, 1); m[9] = (float)ent_pos(c, s0, 2);
                        m[10] = __int_as_float(md_tex);
                        m[11] = 0.0f;
                    }
                    mesh_tris += md_ntris;
                    ++n_mesh;
                } else {
                    atomicOr(a.status, MW_ST_VIS_OVERFLOW);
                }
                if (md_ntris > 0) {     // the mesh's last vertex normal stays current
                    const float *ln = a.mesh_nrm + ((size_t)(md_first + md_ntris - 1) * 3 + 2) * 3;
                    stale_n[0] = uni(ln[0]); stale_n[1] = uni(ln[1]); stale_n[2] = uni(ln[2]);
                }
                ++s0;
                continue;
            }
            // the run of slots [s0, s1): up to 10 slots, ends before the next mesh of this pass
            int s1 = s0 + 10 < a.E ? s0 + 10 : a.E;
            {
                const uint64_t ahead = mesh_mine >> s0;       // bit 0 is clear here
                if (ahead) {
                    const int nxt = s0 + __builtin_ctzll(ahead);
                    s1 = nxt < s1 ? nxt : s1;
                }
            }
            const uint64_t run_boxes = (box_mine >> s0) & ((1ull << (s1 - s0)) - 1ull);
            if (run_boxes) {
                const int i = lane;
                bool vis = false;
                HV h[4];
                PolyGeom g;
                float col[3] = {0.0f, 0.0f, 0.0f};
                const int slot = s0 + i / 6, f = i % 6;
                if (i < (s1 - s0) * 6 && ((run_boxes >> (i / 6)) & 1ull)) {
                    // Box.render (entity.py:409-432): T(pos) R_y(dir) drawBox(...)
                    const double edir = (slot == c.live) ? c.cdir : a.edir[(size_t)slot * a.N + env];
                    const mw::SinCos sc = mw::sincos_det(edir);
                    const float cs = (float)sc.c, sn = (float)sc.s;
                    const float ex = (float)ent_pos(c, slot, 0), ey = (float)ent_pos(c, slot, 1), ez = (float)ent_pos(c, slot, 2);
                    const float hx = (float)(ent_geom(a, env, slot, 0) / 2), sy = (float)ent_geom(a, env, slot, 1),
                                hz = (float)(ent_geom(a, env, slot, 2) / 2);
                    const float lo[3] = {-hx, 0.0f, -hz};
                    const float hi[3] = {hx, sy, hz};
                    const float base_col[3] = {(float)ent_geom(a, env, slot, 3), (float)ent_geom(a, env, slot, 4),
                                               (float)ent_geom(a, env, slot, 5)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int sel = kBoxSel[f][k];
                        const float lx = (sel & 1) ? hi[0] : lo[0];
                        const float ly = (sel & 2) ? hi[1] : lo[1];
                        const float lz = (sel & 4) ? hi[2] : lo[2];
                        float wx = fmaf(cs, lx, sn * lz) + ex;
                        float wy = ly + ey;
                        float wz = fmaf(cs, lz, -(sn * lx)) + ez;
                        if (proxy) {    // drawBox(pos -+ 0.1, pos.y .. pos.y + 0.2) evaluated in double, glVertex3f
                            wx = (float)(ent_pos(c, slot, 0) + ((sel & 1) ? 0.1 : -0.1));
                            wy = (sel & 2) ? (float)(ent_pos(c, slot, 1) + 0.2) : (float)ent_pos(c, slot, 1);
                            wz = (float)(ent_pos(c, slot, 2) + ((sel & 4) ? 0.1 : -0.1));
                        }
                        h[k] = xform(cam, wx, wy, wz);
                    }
                    const float n[3] = {fmaf(cs, kBoxN[f][0], sn * kBoxN[f][2]), kBoxN[f][1],
                                        fmaf(cs, kBoxN[f][2], -(sn * kBoxN[f][0]))};
                    light(cam, n, base_col, col);
                    vis = cull_poly(a, h, 4, g);
                }
                const int idx = compact(lane, vis, count);
                if (vis) {
                    if (idx < a.max_vis) {
                        const float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                        write_poly(a, env, idx, proxy ? (0x10000u | (uint32_t)slot) : (uint32_t)(idx + mesh_tris), h, 4, g, uv, col, -1, s_zmin);
                    } else {
                        atomicOr(a.status, MW_ST_VIS_OVERFLOW);
                    }
                }
                // drawBox ends with glNormal3f(0, -1, 0) (opengl.py:495): current after the last box of the run
                stale_n[0] = 0.0f; stale_n[1] = -1.0f; stale_n[2] = 0.0f;
            }
            s0 = s1;
        }
    }
    if (view_flags & 2) {
        // Agent.render (entity.py:518-539): red triangle on top of the agent's cylinder, no glNormal3f
        // => lit with the stale current normal.  Last in draw order.
        bool vis = false;
        HV h[4];
        PolyGeom g;
        float col[3] = {0.0f, 0.0f, 0.0f};
        if (lane == 0) {
            const mw::SinCos sc = mw::sincos_det(c.dir);
            const double rad = a.agent_radius, hgt = a.agent_height;
            const double p[3] = {c.px + 0 * hgt, c.py + 1 * hgt, c.pz + 0 * hgt};
            const double dv[3] = {sc.c * rad, 0 * rad, -sc.s * rad}, rv[3] = {sc.s * rad, 0 * rad, sc.c * rad};
            double q0[3], q1[3], q2[3];
            for (int i = 0; i < 3; ++i) {
                q0[i] = p[i] + dv[i];
                q1[i] = p[i] + 0.75 * (rv[i] - dv[i]);
                q2[i] = p[i] + 0.75 * (-rv[i] - dv[i]);
            }
            h[0] = xform(cam, (float)q0[0], (float)q0[1], (float)q0[2]);
            h[1] = xform(cam, (float)q2[0], (float)q2[1], (float)q2[2]);
            h[2] = xform(cam, (float)q1[0], (float)q1[1], (float)q1[2]);
            h[3] = h[0];
            const float red[3] = {1.0f, 0.0f, 0.0f};
            light(cam, stale_n, red, col);
            vis = cull_poly(a, h, 3, g);
        }
        const int idx = compact(lane, vis, count);
        if (vis) {
            if (idx < a.max_vis) {
                const float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
                write_poly(a, env, idx, (uint32_t)(idx + mesh_tris), h, 3, g, uv, col, -1, s_zmin);
            } else {
                atomicOr(a.status, MW_ST_VIS_OVERFLOW);
            }
        }
    }
    }       // wave 0
#if MW_SORT_VIS
    if (KW > 1) {       // the other waves learn the final count
        __syncthreads();
        if (writer) s_cnt[0][0] = count;
        __syncthreads();
        count = s_cnt[0][0];
    }
    if (a.rec_order) {
        // visiting order for the raster kernel: list indices sorted by ascending depth bound; order[0] = 1 marks
        // the list as sorted
        const int n = count < a.max_vis ? count : a.max_vis;
        uint16_t *ord = a.rec_order + (size_t)env * (a.max_vis + 1);
        __syncthreads();
        if (n <= MW_SORT_CAP) {
            // bitonic sort of the packed (bound, index) keys in LDS by the 64 lanes of the wave
            int P = 64;
            while (P < n) P <<= 1;
            for (int i = n + (int)threadIdx.x; i < P; i += 64 * KW) s_zmin[i] = ~0ull;
            __syncthreads();
            for (int k = 2; k <= P; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = (int)threadIdx.x; t < (P >> 1); t += 64 * KW) {
                        const int i = 2 * t - (t & (j - 1));        // index with a zero inserted at bit log2(j)
                        const int l = i | j;
                        const unsigned long long x = s_zmin[i], y = s_zmin[l];
                        const bool up = (i & k) == 0;
                        if ((x > y) == up) { s_zmin[i] = y; s_zmin[l] = x; }
                    }
                    __syncthreads();
                }
            }
            for (int r = (int)threadIdx.x; r < n; r += 64 * KW) ord[1 + r] = (uint16_t)(s_zmin[r] & 0xFFFFull);
            if (writer) ord[0] = 1;
        } else if (writer) {
            ord[0] = 0;
        }
    }
#endif
    if (writer && a.k1_prof) {      // MW_K1_PROF: cycles of the phases (perf experiments only, tools/perf/k1prof.py)
        const unsigned long long pt4 = __builtin_readcyclecounter();
        unsigned long long *pp = a.k1_prof + (size_t)env * 8;
        pp[0] = pt1 - pt0; pp[1] = pt2 - pt1; pp[2] = pt3 - pt2; pp[3] = pt4 - pt3; pp[4] = (unsigned long long)regenerated;
    }
    if (writer) {
        a.nvis[env] = count < a.max_vis ? count : a.max_vis;
        a.k3_cost[env] = mesh_tris;             // mesh triangles in view: the mesh kernel's scheduling weight
        hdr[0] = sky[0]; hdr[1] = sky[1]; hdr[2] = sky[2];
        hdr[3] = __int_as_float(n_mesh);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            hdr[4 + 4 * i + 0] = cam.m[i][0]; hdr[4 + 4 * i + 1] = cam.m[i][1];
            hdr[4 + 4 * i + 2] = cam.m[i][2]; hdr[4 + 4 * i + 3] = cam.m[i][3];
            hdr[20 + i] = cam.L[i]; hdr[24 + i] = cam.amb[i]; hdr[28 + i] = cam.lcol[i];
        }
        hdr[16] = cam.p00; hdr[17] = cam.p11; hdr[18] = cam.p22; hdr[19] = cam.p23;
        hdr[23] = __int_as_float(cam.ortho); hdr[27] = cam.p03; hdr[31] = cam.p13;
        if (remove_slot >= 0) a.ekind[(size_t)remove_slot * a.N + env] = MW_ENT_NONE;
    }
}
