// Device functions shared by the step kernels (mw_setup.hip: one wavefront per env, any scene; mw_setup_dense.hip:
// several envs per wavefront, small scenes) and the geometry kernel: the f64 dynamics of MiniWorldEnv.step
// (miniworld.py:606-730, 937-963; math.py:30-62).
#pragma once
#include "mw_device.h"
#include "mw_math.h"
#include "mw_rng.h"
#include "mw_gen.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

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

}  // namespace
