// K1 — per-env step (the wave-per-env form; small scenes use the dense form, mw_setup_dense.hip).
//
// Replaces, per env and per step (reference file:line):
//   MiniWorldEnv.step / move_agent / turn_agent / _get_carry_pos   miniworld.py:606-730
//   MiniWorldEnv.intersect + intersect_circle_segs                 miniworld.py:937-963, math.py:30-62
//   near / _reward + env rules                                     miniworld.py:965-975,1012-1017; hallway.py:67-74; pickupobjects.py:83-95
// The frame itself — camera, transform, lighting, clipping, triangle setup — is the geometry kernel's (mw_geom.hip).
//
// Lanes cooperate: collision segments and entities are tested one per lane (ballot).
// All double-precision dynamics follow numpy's evaluation order.
#include "mw_setup_common.h"

// ---------------------------------------------------------------- the kernel

#ifndef MW_SETUP_KERNEL_NAME
#define MW_SETUP_KERNEL_NAME mw_step_setup_kernel
#endif
#define MW_K1_WAVES 1
#ifndef MW_K1_OCC
#define MW_K1_OCC 3      // waves per SIMD the register allocation aims at
#endif

extern "C" __global__ __launch_bounds__(64 * MW_K1_WAVES) __attribute__((amdgpu_waves_per_eu(MW_K1_OCC, 4))) void MW_SETUP_KERNEL_NAME(
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
        case 2: move_agent<false>(c, fwd_step, fwd_drift); break;
        case 3: move_agent<false>(c, -fwd_step, fwd_drift); break;
        case 0: turn_agent<false>(c, turn_step); break;
        case 1: turn_agent<false>(c, -turn_step); break;
        case 4: {   // pickup (miniworld.py:695-702)
            const mw::SinCos sc = mw::sincos_det(c.dir);
            const double tx = c.px + sc.c * 1.5 * a.agent_radius;
            const double tz = c.pz + (-sc.s) * 1.5 * a.agent_radius;
            const int hit = intersect<false>(c, -1, tx, tz, 1.2 * a.agent_radius);
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
        if (a.task >= MW_TASK_SIDEWALK) program_rules(c, action, step_count, rew, tm);
        int health = 0;
        if (a.task == MW_TASK_COLLECT) {        // collecthealth.py:79-98
            health = a.health[env] - 2;
            if (action == 4 && c.carry >= 0) {  // the kit in hand is consumed — after this frame was drawn (remove_slot)
                remove_slot = c.carry;
                health = 100;
            }
            if (health > 0) rew = 2.0; else { rew = -100.0; tm = 1; }
        }
        if (KW > 1) __syncthreads();        // every wave has read the state it needs: thread 0 may now overwrite it
        if (writer) {
            if (drew) mw::rng_store(a.rng, a.N, env, rng);
            reward[env] = (float)rew;
            term[env] = (uint8_t)tm;
            trunc[env] = (uint8_t)tr;
            a.step[env] = step_count;
            a.picked[env] = picked;
            if (a.task == MW_TASK_COLLECT) a.health[env] = health;
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
            if (writer) mw::keep_final_info(a, env);
            if (a.spare) {
                // the next world was generated ahead (by a refill block of an earlier launch): claim it
                if (writer) s_cnt[0][0] = (int)atomicCAS(a.refill_mask + env, 1u, 3u);
                __syncthreads();
                __threadfence();        // acquire: the spare's contents (written by another block, released with its state) are read behind the claim
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

    // The frame's vertex half — camera, lighting, transform, clipping, triangle setup under the pinned GL rules — is
    // mw_geom_kernel's (mw_geom.hip); what this step leaves for after its frame (a picked-up object is drawn one last time,
    // pickupobjects.py:86-88) goes with it.
    if (writer && do_step) a.pending_remove[env] = remove_slot;
}
