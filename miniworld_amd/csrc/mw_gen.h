// Device-side world generators: MiniWorldEnv.reset (miniworld.py:544-604) for the single
// rectangular-room envs, i.e. the env's _gen_world (hallway.py:55-65, oneroom.py:59-62),
// place_entity's rejection sampling (miniworld.py:872-905), DomainParams.sample of the
// per-episode parameters (miniworld.py:576-585; entity.py:405-407, 505-515).
// Same distributions as the reference, drawn from the engine's Philox stream (the numpy
// PCG64 stream is reproduced host-side only).  Executed by ONE lane.
#pragma once
#include "mw_device.h"
#include "mw_rng.h"

namespace mw {

constexpr double kGenPi = 3.14159265358979323846;

__device__ inline double gen_param(Rng &r, const mw_range &p, bool dr)
{
    return dr ? rng_uniform(r, p.lo, p.hi) : p.def;
}

// circle vs the env's wall segments (math.py:30-62), scalar version
__device__ inline bool gen_hits_wall(const MwArgs &a, int set, double x, double z, double radius)
{
    const double *segs = a.segs + (size_t)set * a.max_segs * 4;
    const int ns = a.nsegs[set];
    for (int i = 0; i < ns; ++i) {
        const double sax = segs[i * 4 + 0], saz = segs[i * 4 + 1], sbx = segs[i * 4 + 2], sbz = segs[i * 4 + 3];
        const double abx = sbx - sax, abz = sbz - saz, apx = x - sax, apz = z - saz;
        double t = (apx * abx + apz * abz) / (abx * abx + abz * abz);
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        const double dx = sax + t * abx - x, dz = saz + t * abz - z;
        if (sqrt(dx * dx + dz * dz) < radius) return true;
    }
    return false;
}

// place_entity in the rectangular room gen_args[0..3] (miniworld.py:872-905); lx/hx narrow
// the sampled x range like the min_x / max_x keyword arguments do.
__device__ inline bool gen_place(const MwArgs &a, int env, int set, Rng &r, double radius, int n_placed,
                                 double lx, double hx, double &ox, double &oz)
{
    const double rx0 = a.gen_args[0], rx1 = a.gen_args[1], rz0 = a.gen_args[2], rz1 = a.gen_args[3];
    for (int attempt = 0; attempt < 4096; ++attempt) {
        const double x = rng_uniform(r, lx - radius, hx + radius);
        const double z = rng_uniform(r, rz0 - radius, rz1 + radius);
        // Room.point_inside (miniworld.py:272-284): strictly inside every edge
        if (!(x > rx0 && x < rx1 && z > rz0 && z < rz1)) continue;
        if (gen_hits_wall(a, set, x, z, radius)) continue;
        bool hit = false;
        for (int s = 0; s < n_placed; ++s) {
            const double dx = a.epos[((size_t)0 * a.E + s) * a.N + env] - x;
            const double dz = a.epos[((size_t)2 * a.E + s) * a.N + env] - z;
            hit |= sqrt(dx * dx + dz * dz) < radius + a.egeom[((size_t)7 * a.E + s) * a.N + env];
        }
        if (hit) continue;
        ox = x; oz = z;
        return true;
    }
    atomicOr(a.status, MW_ST_PLACEMENT_FAIL);
    ox = 0.5 * (rx0 + rx1); oz = 0.5 * (rz0 + rz1);
    return false;
}

__device__ inline void gen_store_box(const MwArgs &a, int env, int slot, double x, double z, double dir, double size,
                                     const double col[3])
{
    const size_t N = a.N, E = a.E;
    a.ekind[(size_t)slot * N + env] = MW_ENT_BOX;
    a.emesh[(size_t)slot * N + env] = -1;
    a.estatic[(size_t)slot * N + env] = 0;
    a.epos[((size_t)0 * E + slot) * N + env] = x;
    a.epos[((size_t)1 * E + slot) * N + env] = 0.0;
    a.epos[((size_t)2 * E + slot) * N + env] = z;
    a.edir[(size_t)slot * N + env] = dir;
    for (int k = 0; k < 3; ++k) a.egeom[((size_t)k * E + slot) * N + env] = size;
    for (int k = 0; k < 3; ++k) a.egeom[((size_t)(3 + k) * E + slot) * N + env] = col[k];
    a.egeom[((size_t)6 * E + slot) * N + env] = 1.0;
    a.egeom[((size_t)7 * E + slot) * N + env] = sqrt(size * size + size * size) / 2.0;   // Box.radius entity.py:401
    a.egeom[((size_t)8 * E + slot) * N + env] = size;
}

// One full reset of env `env`.  Writes every per-env state array.
__device__ inline void generate_world(const MwArgs &a, int env)
{
    const int set = a.shared_geom ? 0 : env;
    const bool dr = a.domain_rand != 0;
    Rng r = rng_load(a.rng, a.N, env);
    const size_t N = a.N;
    for (int s = 0; s < a.E; ++s) a.ekind[(size_t)s * N + env] = MW_ENT_NONE;
    double ax = 0, az = 0, adir = 0;
    if (a.generator == MW_GEN_HALLWAY || a.generator == MW_GEN_ONEROOM) {
        // the red box (hallway.py:59, oneroom.py:61), then the agent (hallway.py:62-65, oneroom.py:62)
        const double size = a.gen_args[7];
        const double brad = sqrt(size * size + size * size) / 2.0;
        double bx, bz;
        gen_place(a, env, set, r, brad, 0, a.gen_args[4], a.gen_args[1], bx, bz);
        const double bdir = rng_uniform(r, -kGenPi, kGenPi);
        double col[3] = {1.0, 0.0, 0.0};                       // COLORS["red"] entity.py:31
        // written first so that the agent's placement sees it; colour bias applied below
        gen_store_box(a, env, 0, bx, bz, bdir, size, col);
        adir = rng_uniform(r, -a.gen_args[6], a.gen_args[6]);
        gen_place(a, env, set, r, a.agent_radius, 1, a.gen_args[0], a.gen_args[5], ax, az);
        // per-episode parameters (miniworld.py:576-578), then entity randomisation (:584-585)
        for (int k = 0; k < 3; ++k) a.light[(size_t)(0 + k) * N + env] = gen_param(r, a.sky[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(3 + k) * N + env] = gen_param(r, a.light_pos[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(6 + k) * N + env] = gen_param(r, a.light_color[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(9 + k) * N + env] = gen_param(r, a.light_ambient[k], dr);
        for (int k = 0; k < 3; ++k) {
            const double v = col[k] + gen_param(r, a.color_bias[k], dr);      // Box.randomize entity.py:405-407
            a.egeom[((size_t)(3 + k) * a.E + 0) * N + env] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
        }
        a.cam[(size_t)0 * N + env] = gen_param(r, a.cam_height, dr);          // Agent.randomize entity.py:505-515
        a.cam[(size_t)1 * N + env] = gen_param(r, a.cam_fwd_disp, dr);
        a.cam[(size_t)2 * N + env] = gen_param(r, a.cam_pitch, dr);
        a.cam[(size_t)3 * N + env] = gen_param(r, a.cam_fov_y, dr);
    }
    if (a.generator == MW_GEN_PICKUP) {
        // pickupobjects.py:55-81: num_objs objects of random kind (Ball, Box, Key) and colour, placed
        // one after the other, then the agent; gen_tab holds the per-kind constants computed by the
        // host from the meshes (radius, height, scale) and the first mesh id of each kind.
        const int n = a.num_objs;
        for (int s = 0; s < n && s < a.E; ++s) {
            const int kind = (int)rng_below(r, 3);          // 0 ball, 1 box, 2 key (obj_types order)
            const int color = (int)rng_below(r, 6);         // index into the sorted COLOR_NAMES
            const double radius = a.gen_tab[kind * 4 + 0], height = a.gen_tab[kind * 4 + 1];
            double x, z;
            gen_place(a, env, set, r, radius, s, a.gen_args[0], a.gen_args[1], x, z);
            const double dir = rng_uniform(r, -kGenPi, kGenPi);
            const size_t E = a.E;
            a.ekind[(size_t)s * N + env] = kind == 1 ? MW_ENT_BOX : MW_ENT_MESH;
            a.emesh[(size_t)s * N + env] = kind == 1 ? -1 : (int)a.gen_tab[kind * 4 + 3] + color;
            a.estatic[(size_t)s * N + env] = 0;
            a.epos[((size_t)0 * E + s) * N + env] = x;
            a.epos[((size_t)1 * E + s) * N + env] = 0.0;
            a.epos[((size_t)2 * E + s) * N + env] = z;
            a.edir[(size_t)s * N + env] = dir;
            const double size = kind == 1 ? 0.9 : 0.0;
            for (int k = 0; k < 3; ++k) a.egeom[((size_t)k * E + s) * N + env] = size;
            for (int k = 0; k < 3; ++k) a.egeom[((size_t)(3 + k) * E + s) * N + env] = a.gen_colors[color * 3 + k];
            a.egeom[((size_t)6 * E + s) * N + env] = a.gen_tab[kind * 4 + 2];
            a.egeom[((size_t)7 * E + s) * N + env] = radius;
            a.egeom[((size_t)8 * E + s) * N + env] = height;
        }
        gen_place(a, env, set, r, a.agent_radius, n < a.E ? n : a.E, a.gen_args[0], a.gen_args[1], ax, az);
        adir = rng_uniform(r, -kGenPi, kGenPi);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(0 + k) * N + env] = gen_param(r, a.sky[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(3 + k) * N + env] = gen_param(r, a.light_pos[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(6 + k) * N + env] = gen_param(r, a.light_color[k], dr);
        for (int k = 0; k < 3; ++k) a.light[(size_t)(9 + k) * N + env] = gen_param(r, a.light_ambient[k], dr);
        for (int s = 0; s < n && s < a.E; ++s)               // Box.randomize: colour bias (entity.py:405-407)
            if (a.ekind[(size_t)s * N + env] == MW_ENT_BOX)
                for (int k = 0; k < 3; ++k) {
                    const size_t gi = ((size_t)(3 + k) * a.E + s) * N + env;
                    const double v = a.egeom[gi] + gen_param(r, a.color_bias[k], dr);
                    a.egeom[gi] = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
                }
        a.cam[(size_t)0 * N + env] = gen_param(r, a.cam_height, dr);
        a.cam[(size_t)1 * N + env] = gen_param(r, a.cam_fwd_disp, dr);
        a.cam[(size_t)2 * N + env] = gen_param(r, a.cam_pitch, dr);
        a.cam[(size_t)3 * N + env] = gen_param(r, a.cam_fov_y, dr);
    }
    a.ax[env] = ax; a.ay[env] = 0.0; a.az[env] = az; a.adir[env] = adir;
    a.carry[env] = -1; a.step[env] = 0; a.picked[env] = 0;
    rng_store(a.rng, a.N, env, r);
}

}  // namespace mw
