/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Double-precision restatement of the reference's per-step dynamics:
 *   MiniWorldEnv.step            miniworld/miniworld.py:670-730
 *   move_agent / turn_agent      miniworld/miniworld.py:620-668
 *   _get_carry_pos               miniworld/miniworld.py:606-618
 *   intersect                    miniworld/miniworld.py:937-963
 *   intersect_circle_segs        miniworld/math.py:30-62
 *   near / _reward               miniworld/miniworld.py:965-975, 1012-1017
 *   Entity.dir_vec / right_vec   miniworld/entity.py:95-113
 *   env rules: Hallway/OneRoom/Maze (hallway.py:67-74, oneroom.py:64-71, maze.py:155-162),
 *              PickupObjects (pickupobjects.py:83-95)
 * The evaluation order of every floating-point expression follows numpy's (left to
 * right, reductions over (x, y, z) with y == 0 terms kept where they matter).
 * Pinned against the reference run under GL stubs: tests/golden/dyn_*.npz.
 */
#include "mwo.h"
#include <math.h>
#include <string.h>

int mwo_intersect(const mwo_agent_state *ag, const mwo_phys_ent *ents, int32_t self_idx,
                  double px, double pz, double radius, const double *segs, int32_t n_segs)
{
    /* math.py:30-62 — any segment whose clamped closest point is nearer than radius */
    for (int i = 0; i < n_segs; ++i) {
        double ax = segs[i * 4 + 0], az = segs[i * 4 + 1], bx = segs[i * 4 + 2], bz = segs[i * 4 + 3];
        double abx = bx - ax, abz = bz - az;
        double apx = px - ax, apz = pz - az;
        double dotAPAB = apx * abx + apz * abz;
        double dotABAB = abx * abx + abz * abz;
        double t = dotAPAB / dotABAB;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);       /* np.clip; NaN stays NaN */
        double cx = ax + t * abx, cz = az + t * abz;
        double dx = cx - px, dz = cz - pz;
        double dist = sqrt(dx * dx + dz * dz);
        if (dist < radius) return -1;
    }
    /* miniworld.py:951-961 — entities in list order, skipping self */
    for (int i = 0; i < ag->n_ents; ++i) {
        if (i == self_idx || !ents[i].alive) continue;
        double dx = ents[i].pos[0] - px, dz = ents[i].pos[2] - pz;
        double d = sqrt(dx * dx + dz * dz);
        if (d < radius + ents[i].radius) return 1 + i;
    }
    if (self_idx >= 0) {
        double dx = ag->pos[0] - px, dz = ag->pos[2] - pz;
        double d = sqrt(dx * dx + dz * dz);
        if (d < radius + ag->radius) return 1 + ag->n_ents;
    }
    return 0;
}

static void carry_pos(const mwo_agent_state *ag, const mwo_phys_ent *e, double apx, double apy,
                      double apz, double dvx, double dvz, double out[3])
{
    /* miniworld.py:606-618 */
    double dist = ag->radius + e->radius + ag->max_forward_step;
    out[0] = apx + dvx * 1.05 * dist;
    out[1] = apy + 0.0 * 1.05 * dist;
    out[2] = apz + dvz * 1.05 * dist;
    double y = ag->cam_height - e->height - 0.3;
    double y_pos = y > 0.0 ? y : 0.0;
    out[1] = out[1] + 1.0 * y_pos;
}

static int move_agent(mwo_agent_state *ag, mwo_phys_ent *ents, const double *segs, int n_segs,
                      double fwd_dist, double fwd_drift)
{
    /* miniworld.py:620-645 */
    double s, c;
    mwo_sincos(ag->dir, &s, &c);
    double dvx = c, dvz = -s;       /* dir_vec   (entity.py:95-103)  */
    double rvx = s, rvz = c;        /* right_vec (entity.py:105-113) */
    double nx = ag->pos[0] + dvx * fwd_dist + rvx * fwd_drift;
    double ny = ag->pos[1] + 0.0 * fwd_dist + 0.0 * fwd_drift;
    double nz = ag->pos[2] + dvz * fwd_dist + rvz * fwd_drift;
    if (mwo_intersect(ag, ents, -1, nx, nz, ag->radius, segs, n_segs)) return 0;
    if (ag->carrying >= 0) {
        mwo_phys_ent *ce = &ents[ag->carrying];
        double cp[3];
        carry_pos(ag, ce, nx, ny, nz, dvx, dvz, cp);
        if (mwo_intersect(ag, ents, ag->carrying, cp[0], cp[2], ce->radius, segs, n_segs)) return 0;
        memcpy(ce->pos, cp, sizeof cp);
    }
    ag->pos[0] = nx; ag->pos[1] = ny; ag->pos[2] = nz;
    return 1;
}

static int turn_agent(mwo_agent_state *ag, mwo_phys_ent *ents, const double *segs, int n_segs,
                      double turn_angle_deg)
{
    /* miniworld.py:647-668 */
    double turn = turn_angle_deg * (3.14159265358979323846 / 180.0);
    double orig = ag->dir;
    ag->dir = ag->dir + turn;
    if (ag->carrying >= 0) {
        mwo_phys_ent *ce = &ents[ag->carrying];
        double s, c, cp[3];
        mwo_sincos(ag->dir, &s, &c);
        carry_pos(ag, ce, ag->pos[0], ag->pos[1], ag->pos[2], c, -s, cp);
        if (mwo_intersect(ag, ents, ag->carrying, cp[0], cp[2], ce->radius, segs, n_segs)) {
            ag->dir = orig;
            return 0;
        }
        memcpy(ce->pos, cp, sizeof cp);
        ce->dir = ag->dir;
    }
    return 1;
}

int mwo_step(mwo_agent_state *ag, mwo_phys_ent *ents, mwo_phys_ent *ents_at_render,
             const double *segs, int32_t n_segs, int32_t action,
             double fwd_step, double fwd_drift, double turn_step,
             double *reward, int32_t *terminated, int32_t *truncated)
{
    ag->step_count += 1;
    switch (action) {
    case 2: move_agent(ag, ents, segs, n_segs, fwd_step, fwd_drift); break;   /* move_forward */
    case 3: move_agent(ag, ents, segs, n_segs, -fwd_step, fwd_drift); break;  /* move_back    */
    case 0: turn_agent(ag, ents, segs, n_segs, turn_step); break;             /* turn_left    */
    case 1: turn_agent(ag, ents, segs, n_segs, -turn_step); break;            /* turn_right   */
    case 4: {                                                                 /* pickup :695-702 */
        double s, c;
        mwo_sincos(ag->dir, &s, &c);
        double tx = ag->pos[0] + c * 1.5 * ag->radius;
        double tz = ag->pos[2] + (-s) * 1.5 * ag->radius;
        int hit = mwo_intersect(ag, ents, -1, tx, tz, 1.2 * ag->radius, segs, n_segs);
        if (ag->carrying < 0 && hit > 0 && !ents[hit - 1].is_static) ag->carrying = hit - 1;
        break;
    }
    case 5:                                                                   /* drop :705-708 */
        if (ag->carrying >= 0) {
            ents[ag->carrying].pos[1] = 0.0;
            ag->carrying = -1;
        }
        break;
    default: break;                                                           /* toggle, done: no-ops */
    }
    if (ag->carrying >= 0) {                                                  /* :711-714 */
        double s, c, cp[3];
        mwo_sincos(ag->dir, &s, &c);
        carry_pos(ag, &ents[ag->carrying], ag->pos[0], ag->pos[1], ag->pos[2], c, -s, cp);
        memcpy(ents[ag->carrying].pos, cp, sizeof cp);
        ents[ag->carrying].dir = ag->dir;
    }
    /* obs = render_obs() happens here (:717) */
    if (ents_at_render) memcpy(ents_at_render, ents, sizeof(mwo_phys_ent) * (size_t)ag->n_ents);

    double rew = 0.0;
    int term = 0, trunc = 0;
    if (ag->step_count >= ag->max_episode_steps) trunc = 1;                   /* :720-724 */
    if (ag->task == MWO_TASK_GOTO) {
        /* near(box) (miniworld.py:965-975): full 3-D distance */
        const mwo_phys_ent *b = &ents[ag->goal_ent];
        double dx = b->pos[0] - ag->pos[0], dy = b->pos[1] - ag->pos[1], dz = b->pos[2] - ag->pos[2];
        double dist = sqrt(dx * dx + dy * dy + dz * dz);
        if (dist < b->radius + ag->radius + 1.1 * ag->max_forward_step) {
            rew += 1.0 - 0.2 * ((double)ag->step_count / (double)ag->max_episode_steps);
            term = 1;
        }
    } else if (ag->task == MWO_TASK_PUTNEXT) {
        /* putnext.py:74-78: not carrying and near(red_box, yellow_box) */
        if (ag->carrying < 0) {
            const mwo_phys_ent *e0 = &ents[ag->goal_ent], *e1 = &ents[ag->goal_ent2];
            double dx = e0->pos[0] - e1->pos[0], dy = e0->pos[1] - e1->pos[1], dz = e0->pos[2] - e1->pos[2];
            double dist = sqrt(dx * dx + dy * dy + dz * dz);
            if (dist < e0->radius + e1->radius + 1.1 * ag->max_forward_step) {
                rew += 1.0 - 0.2 * ((double)ag->step_count / (double)ag->max_episode_steps);
                term = 1;
            }
        }
    } else if (ag->task == MWO_TASK_PICKUP) {
        if (ag->carrying >= 0) {                                              /* pickupobjects.py:86-93 */
            ents[ag->carrying].alive = 0;
            ag->carrying = -1;
            ag->num_picked_up += 1;
            rew = 1.0;
            if (ag->num_picked_up == ag->num_objs) term = 1;
        }
    }
    *reward = rew; *terminated = term; *truncated = trunc;
    return 0;
}

/* cpu_baseline helper for bench.py: `steps` iterations of [mwo_step + mwo_render_obs] on one
 * env with a deterministic action sequence (LCG), resetting the pose when the episode ends.
 * Returns the elapsed wall-clock seconds (CLOCK_MONOTONIC) of the loop. */
#include <time.h>
double mwo_bench_loop(mwo_scene *sc, mwo_agent_state *ag, mwo_phys_ent *ents, const double *segs,
                      int32_t n_segs, int32_t n_actions, int32_t steps, uint8_t *rgb)
{
    struct timespec t0, t1;
    mwo_agent_state ag0 = *ag;
    mwo_phys_ent ents0[64];
    int ne = ag->n_ents < 64 ? ag->n_ents : 64;
    memcpy(ents0, ents, (size_t)ne * sizeof(mwo_phys_ent));
    uint32_t lcg = 12345u;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < steps; ++i) {
        lcg = lcg * 1664525u + 1013904223u;
        int action = (int)((lcg >> 16) % (uint32_t)n_actions);
        double rew;
        int32_t te, tr;
        mwo_step(ag, ents, 0, segs, n_segs, action, 0.15, 0.0, 15.0, &rew, &te, &tr);
        if (te || tr) { *ag = ag0; memcpy(ents, ents0, (size_t)ne * sizeof(mwo_phys_ent)); }   /* episode restart */
        sc->agent_pos[0] = ag->pos[0]; sc->agent_pos[1] = ag->pos[1]; sc->agent_pos[2] = ag->pos[2];
        sc->agent_dir = ag->dir;
        mwo_render_obs(sc, rgb, 0, 0, 0);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
