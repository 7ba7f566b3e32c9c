// The step of one env of a SMALL scene as one device function: the body of the dense K1 (mw_setup_dense.hip, which see), also
// run by the geometry kernel's fused form (mw_geom.hip: mw_geom_step_kernel) as its prologue.  Every lane of the env calls
// it with the same arguments and evaluates the env's step itself; `leader` (one lane of the env) writes state and flags and —
// on an episode's end — installs the next world.
#pragma once
#include "mw_setup_common.h"

namespace {

__device__ inline void dense_step(const MwArgs &a, int do_step, int env, int lane, bool leader, const int32_t *__restrict__ actions,
                                  float *__restrict__ reward, uint8_t *__restrict__ term, uint8_t *__restrict__ trunc, unsigned char *gen_ws)
{
    // MW_K1_PROF (perf experiments only): cycle stamps of the phases, written by the env's leading lane
    StepCtx c{a, env, lane, a.shared_geom ? 0 : env, 0, 0, 0, 0, 0, -1, -1, {0, 0, 0}, 0};
    c.px = a.ax[env]; c.py = a.ay[env]; c.pz = a.az[env]; c.dir = a.adir[env];
    c.cam_height = a.cam[env];
    c.carry = a.carry[env];
    if (c.carry >= 0) {
        const int k = c.carry;
        c.cpos[0] = ent_pos(c, k, 0); c.cpos[1] = ent_pos(c, k, 1); c.cpos[2] = ent_pos(c, k, 2);
        c.cdir = a.edir[(size_t)k * a.N + env];
        c.live = k;
    }
    int remove_slot = -1;
    int tm = 0, tr = 0;

    if (do_step) {
        const int step_count = a.step[env] + 1;
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
            drew = true;
        }
        const int action = actions[env];
        switch (action) {
        case 2: move_agent<true>(c, fwd_step, fwd_drift); break;
        case 3: move_agent<true>(c, -fwd_step, fwd_drift); break;
        case 0: turn_agent<true>(c, turn_step); break;
        case 1: turn_agent<true>(c, -turn_step); break;
        case 4: {   // pickup (miniworld.py:695-702)
            const mw::SinCos sc = mw::sincos_det(c.dir);
            const double tx = c.px + sc.c * 1.5 * a.agent_radius;
            const double tz = c.pz + (-sc.s) * 1.5 * a.agent_radius;
            const int hit = intersect<true>(c, -1, tx, tz, 1.2 * a.agent_radius);
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
                c.carry = -1;
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
        if (a.task >= MW_TASK_SIDEWALK) program_rules(c, action, step_count, rew, tm);
        // every lane of the env has read the old state (the lanes of a wavefront run in lockstep, and each lane only
        // reads its own env): the leading lane writes the new one
        __builtin_amdgcn_wave_barrier();
        if (leader) {
            if (drew) mw::rng_store(a.rng, a.N, env, rng);
            reward[env] = (float)rew;
            term[env] = (uint8_t)tm;
            trunc[env] = (uint8_t)tr;
            a.step[env] = step_count;
            a.picked[env] = picked;
            a.ax[env] = c.px; a.ay[env] = c.py; a.az[env] = c.pz; a.adir[env] = c.dir;
            if (c.live >= 0) {
                a.epos[((size_t)0 * a.E + c.live) * a.N + env] = c.cpos[0];
                a.epos[((size_t)1 * a.E + c.live) * a.N + env] = c.cpos[1];
                a.epos[((size_t)2 * a.E + c.live) * a.N + env] = c.cpos[2];
                a.edir[(size_t)c.live * a.N + env] = c.cdir;
            }
            a.carry[env] = remove_slot >= 0 ? -1 : c.carry;
        }
        if (a.autoreset == MW_AUTORESET_SAME_STEP && a.generator != MW_GEN_NONE && (tm | tr)) {
            // same-step auto-reset: the observation returned with done = 1 is the first one of the next episode.
            // The env's leading lane installs the next world (several envs of the wave may do so side by side); the
            // env's other lanes then read it like the leader does.
            if (leader) {
                mw::keep_final_info(a, env);
                if (a.spare) {
                    // spare mode: the next world was generated ahead by a refill block of an earlier launch (the
                    // blocks behind the env blocks of this grid): claim it.  States of refill_mask: mw_device.h.
                    const unsigned old = atomicCAS(a.refill_mask + env, 1u, 3u);
                    __threadfence();        // acquire: the spare's contents are read behind the claim
                    if (old == 1u) {
                        // the previous episode lasted one step and the refill has not run yet: generate in place
                        mw::generate_world(*a.gen_live, env, gen_ws, 0);
                    } else {
                        if (old == 2u)      // a refill block of this very launch is on it
                            while (__hip_atomic_load(a.refill_mask + env, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0u) __builtin_amdgcn_s_sleep(16);
                        mw::take_spare_lane(a, env);
                    }
                    __threadfence();
                    atomicExch(a.refill_mask + env, 1u);        // the spare is missing again
                } else {
                    mw::generate_world(*a.gen_live, env, gen_ws, 0);
                }
            }
            __threadfence();
            __builtin_amdgcn_wave_barrier();
            c.px = a.ax[env]; c.py = a.ay[env]; c.pz = a.az[env]; c.dir = a.adir[env];
            c.carry = -1; c.live = -1;
            remove_slot = -1;
        }
    }

    // the frame's vertex half is mw_geom_kernel's (mw_geom.hip); see mw_setup.hip
    if (leader && do_step) a.pending_remove[env] = remove_slot;
}

}  // namespace
