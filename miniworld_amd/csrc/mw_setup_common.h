// Device functions shared by the step + setup kernels (mw_setup.hip: one wavefront per env, any scene;
// mw_setup_dense.hip: one lane per (env, primitive slot), small scenes): the f64 dynamics of
// MiniWorldEnv.step (miniworld.py:606-730, 937-963; math.py:30-62), the camera (R1, R2, R10) and the
// per-primitive raster / shade / classification records (R3-R6) the raster kernels consume.
#pragma once
#include "mw_device.h"
#include "mw_math.h"
#include "mw_rng.h"
#include "mw_gen.h"

#ifndef MW_SORT_VIS
#define MW_SORT_VIS 0       // 1: also emit the depth-sorted visiting order of big scenes (mw_setup_sort*.hip)
#endif
#define MW_SORT_CAP 768     // visible polygons listed per env in LDS (their packed sort keys sit in 6 KiB)
#define MW_SORT_POW2 512    // ... and sorted: the bitonic network pads to a power of two, the largest one inside the key buffer
static_assert(MW_SORT_POW2 <= MW_SORT_CAP && 2 * MW_SORT_POW2 > MW_SORT_CAP && (MW_SORT_POW2 & (MW_SORT_POW2 - 1)) == 0, "MW_SORT_POW2");

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
__device__ int intersect_wave(const StepCtx &c, int self_slot, double x, double z, double radius)
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

// The same query evaluated by ONE lane for its own env (mw_setup_dense.hip: a wavefront holds several envs, the
// lanes of one env all walk its segments and entities and arrive at the same answer).
__device__ int intersect_lane(const StepCtx &c, int self_slot, double x, double z, double radius)
{
    const MwArgs &a = c.a;
    const double *segs = a.segs + (size_t)c.set * a.max_segs * 4;
    const int ns = a.nsegs[c.set];
    bool hit = false;
#pragma unroll 2
    for (int i = 0; i < ns; ++i) {      // independent iterations: two divisions / square roots in flight
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
    if (hit) return -1;
    for (int slot = 0; slot < a.E; ++slot) {
        if (slot == self_slot || a.ekind[(size_t)slot * a.N + c.env] == MW_ENT_NONE) continue;
        const double dx = ent_pos(c, slot, 0) - x, dz = ent_pos(c, slot, 2) - z;
        if (sqrt(dx * dx + dz * dz) < radius + ent_geom(a, c.env, slot, 7)) return 1 + slot;
    }
    if (self_slot >= 0) {
        const double dx = c.px - x, dz = c.pz - z;
        if (sqrt(dx * dx + dz * dz) < radius + a.agent_radius) return 1 + a.E;
    }
    return 0;
}

template <bool PER_LANE>
__device__ inline int intersect(const StepCtx &c, int self_slot, double x, double z, double radius)
{
    return PER_LANE ? intersect_lane(c, self_slot, x, z, radius) : intersect_wave(c, self_slot, x, z, radius);
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

// near(ent) (miniworld.py:965-975): 3D distance agent - entity below the two radii + 1.1 * max_forward_step
__device__ inline bool near_agent(const StepCtx &c, int slot)
{
    const double dx = ent_pos(c, slot, 0) - c.px, dy = ent_pos(c, slot, 1) - c.py, dz = ent_pos(c, slot, 2) - c.pz;
    return sqrt(dx * dx + dy * dy + dz * dz) < ent_geom(c.a, c.env, slot, 7) + c.a.agent_radius + 1.1 * c.a.max_forward_step;
}

// The env rules that live in the placement program's tables (include/mwengine.h):
// Sidewalk.step (sidewalk.py:93-104): the street ends the episode and zeroes the reward, the box adds the GOTO reward;
// Sign.step (sign.py:152-170): action move_forward + 1 ends the episode, touching an object ends it with +-1.
__device__ inline void program_rules(const StepCtx &c, int action, int step_count, double &rew, int &tm)
{
    const MwArgs &a = c.a;
    if (a.task == MW_TASK_SIDEWALK) {
        const double *st = a.prog->p.street;
        if (c.px > st[0] && c.px < st[1] && c.pz > st[2] && c.pz < st[3]) { rew = 0.0; tm = 1; }     // Room.point_inside
        if (near_agent(c, a.goal_ent)) {
            rew += 1.0 - 0.2 * ((double)step_count / (double)a.max_steps);
            tm = 1;
        }
    } else if (a.task == MW_TASK_SIGN) {
        if (action == 3) tm = 1;                    // actions.move_forward + 1: the custom end-of-episode action
        const mw_gen_program &g = a.prog->p;
        for (int k = 0; k < g.sign_n; ++k)
            if (near_agent(c, g.sign_slot[k])) { tm = 1; rew = g.sign_reward[k]; }
    }
}

template <bool PER_LANE>
__device__ void move_agent(StepCtx &c, double fwd_dist, double fwd_drift)
{
    const mw::SinCos sc = mw::sincos_det(c.dir);
    const double dvx = sc.c, dvz = -sc.s, rvx = sc.s, rvz = sc.c;
    const double nx = c.px + dvx * fwd_dist + rvx * fwd_drift;
    const double ny = c.py + 0.0 * fwd_dist + 0.0 * fwd_drift;
    const double nz = c.pz + dvz * fwd_dist + rvz * fwd_drift;
    if (intersect<PER_LANE>(c, -1, nx, nz, c.a.agent_radius)) return;
    if (c.carry >= 0) {
        double cp[3];
        carry_pos(c, c.carry, nx, ny, nz, dvx, dvz, cp);
        if (intersect<PER_LANE>(c, c.carry, cp[0], cp[2], ent_geom(c.a, c.env, c.carry, 7))) return;
        c.cpos[0] = cp[0]; c.cpos[1] = cp[1]; c.cpos[2] = cp[2];
    }
    c.px = nx; c.py = ny; c.pz = nz;
}

template <bool PER_LANE>
__device__ void turn_agent(StepCtx &c, double turn_deg)
{
    const double turn = turn_deg * (kPi / 180.0);
    const double orig = c.dir;
    c.dir = c.dir + turn;
    if (c.carry >= 0) {
        const mw::SinCos sc = mw::sincos_det(c.dir);
        double cp[3];
        carry_pos(c, c.carry, c.px, c.py, c.pz, sc.c, -sc.s, cp);
        if (intersect<PER_LANE>(c, c.carry, cp[0], cp[2], ent_geom(c.a, c.env, c.carry, 7))) {
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
    float zx = (ta * invD) * 0.5f, zy = (tb * invD) * 0.5f, zc = fmaf(tc * invD, 0.5f, 0.5f);
    if (h[0].hw > 0.0f && h[1].hw > 0.0f && h[2].hw > 0.0f) {
        // R6p: first three vertices in front of the eye — the plane through their window coordinates (X, Y, z_w), solved
        // in binary64 and rounded to binary32 (the sums above cancel catastrophically for a polygon seen edge-on: a far
        // floor two pixels high, a wall stub a tenth of a pixel wide).  Differences of the window coordinates over common
        // denominators, X1 - X0 = (hx1 w0 - hx0 w1) / (w0 w1): the products are exact in binary64 and the denominators
        // cancel between the numerators and the determinant.
        const double w0 = h[0].hw, w1 = h[1].hw, w2 = h[2].hw;
        const double nax = (double)h[1].hx * w0 - (double)h[0].hx * w1, nay = (double)h[1].hy * w0 - (double)h[0].hy * w1;
        const double naz = (double)h[1].cz * w0 - (double)h[0].cz * w1;
        const double nbx = (double)h[2].hx * w0 - (double)h[0].hx * w2, nby = (double)h[2].hy * w0 - (double)h[0].hy * w2;
        const double nbz = (double)h[2].cz * w0 - (double)h[0].cz * w2;
        const double det = nax * nby - nbx * nay;
        if (det != 0.0) {
            const double r = 1.0 / det, i0 = 1.0 / w0;
            const double zxd = 0.5 * ((naz * nby - nbz * nay) * r), zyd = 0.5 * ((nax * nbz - nbx * naz) * r);
            zx = (float)zxd;
            zy = (float)zyd;
            zc = (float)(0.5 + ((0.5 * (double)h[0].cz - zxd * (double)h[0].hx) - zyd * (double)h[0].hy) * i0);
        }
    }
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

