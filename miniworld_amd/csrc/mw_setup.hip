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
#include "mw_setup_common.h"

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
#define MW_K1_OCC 4
#else
#define MW_K1_WAVES 1
#define MW_K1_OCC 3      // waves per SIMD the register allocation aims at (168 VGPRs: the wave-per-env batch sizes leave 2-4 resident)
#endif
#if MW_SORT_VIS
// ---------------------------------------------------------------- occlusion culling (big scenes)
// A Maze view holds ~95 front-facing polygons inside the frustum and ~13 that own a sample: everything else lies behind
// walls.  With an unpitched camera a wall that spans the whole height of the world (the slab [lo, hi] of all room
// polygons, the eye inside it) hides every room polygon behind it in the screen columns it covers: a ray to a point
// of the slab farther away crosses the wall's plane inside the slab.  So: every such wall in front of the eye marks the
// column bins it covers completely with its farthest depth there (the nearest wall wins the bin), and a polygon
// whose columns are all marked with depths in front of its nearest vertex is dropped before it costs a record, a
// sort slot and the raster kernel's visits.  Frames do not change: a dropped polygon owns no sample —
//   * its fragments are no nearer than its nearest vertex: R6p (mw_setup_common.h::write_poly) computes the depth plane
//     from window coordinates in binary64, so a fragment's depth stays inside its polygon's vertex range (with the 2DH
//     sums of R6 a wall stub 0.1 px wide came out dozens of D16 steps in front of the wall hiding it);
//   * not even through the 16-bit depth quantisation (GL_LESS ties go to the polygon drawn first): "in front" demands
//     more than three depth-buffer steps, 1/z_wall - 1/z_poly > 1.2e-3, one step of D16 over [0.04, 100] being 3.8e-4 in 1/z;
//   * all margins are on the keeping side: walls shrink by 0.05 px and are cut at z = 0.1 (the near plane is at 0.04),
//     polygons grow by 0.1 px (an edge function's rounding moves an edge by < 1e-3 px), polygons with a vertex nearer than
//     0.1 are kept untested.
// tests/test_gpu_env_api.py::test_occlusion_culling_never_changes_a_frame compares MW_OCCLUSION=0 / 1 bit for bit.
#define MW_OCC_BINS 256
#define MW_OCC_CAP 192

// occ_z[0 .. BINS): the bins; occ_z[BINS .. BINS + BINS / 16): the largest value of every group of 16 bins
__device__ inline bool occluded(const float *occ_z, const HV h[4], int nv, float bins_per_px)
{
    float zq = 1e30f, xmn = 1e30f, xmx = -1e30f;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < nv) {
            ok &= h[k].hw >= 0.1f;
            const float X = h[k].hx * __builtin_amdgcn_rcpf(h[k].hw);
            xmn = fminf(xmn, X); xmx = fmaxf(xmx, X); zq = fminf(zq, h[k].hw);
        }
    if (!ok) return false;
    const float fb0 = floorf(fmaxf(xmn - 0.1f, 0.0f) * bins_per_px), fb1 = floorf(fmaxf(xmx + 0.1f, 0.0f) * bins_per_px);
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
#if MW_SORT_VIS
    __shared__ unsigned long long s_zmin_buf[MW_SORT_CAP];
    unsigned long long *s_zmin = s_zmin_buf;
    __shared__ float s_occ_z[MW_OCC_BINS + MW_OCC_BINS / 16];      // occlusion culling (below): farthest depth of the nearest wall per column bin, group maxima
    __shared__ float s_occ_wall[MW_OCC_CAP * 5];
    __shared__ float s_slab[2 * KW];
    __shared__ int s_occ_n;
    __shared__ uint16_t s_list[MW_SORT_CAP];        // the visible room polygons, in draw order
    __shared__ float4 s_pv[2 * 3 * 64 * KW];        // the vertices of up to 512 room polygons (24 KB)
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

    {
        // The frame's vertex half — camera, lighting, transform, clipping, triangle setup under the pinned GL rules — is
        // mw_geom_kernel's (mw_geom.hip); what this step leaves for after its frame (a picked-up object is drawn one last
        // time, pickupobjects.py:86-88) goes with it.
        if (writer && do_step) a.pending_remove[env] = remove_slot;
        return;
    }
    const unsigned long long pt1 = a.k1_prof ? __builtin_readcyclecounter() : 0ull;
    // ---- camera + primitive setup (superseded: kept until the occlusion culling moves to the geometry kernel) -----
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
#if MW_SORT_VIS
    // ---- display list 1 (rooms), big scenes: vertices read once into registers -> slab, occluder walls, column bins ->
    // visibility of every polygon (back face, frustum, occlusion) -> ordered list -> records of the listed ones
    constexpr int PR = 2;                               // polygon rounds held in LDS (2 * 256 = 512 polygons)
    const bool cached = np <= PR * 64 * KW;
    // the twelve vertex floats of thread t's polygon of round r: s_pv[(r * 3 + c) * 256 + t], c = 0 .. 2 (consecutive
    // threads, consecutive 16 bytes).  In LDS, not in registers: 24 more live VGPRs through the occluder passes were
    // 50 scratch reloads in their inner loops.
    int pnv[PR];
#pragma unroll
    for (int r = 0; r < PR; ++r) {
        const int i = r * 64 * KW + wave * 64 + lane;
        pnv[r] = 0;
        if (cached && i < np) {
            const float4 *src = reinterpret_cast<const float4 *>(polys[i].v);      // 12 floats at the head of the 112-byte struct
#pragma unroll
            for (int c = 0; c < 3; ++c) s_pv[(r * 3 + c) * 64 * KW + (int)threadIdx.x] = src[c];
            pnv[r] = polys[i].nv;
        }
    }
    auto load_pv = [&](int r, float (&v)[12]) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float4 q = s_pv[(r * 3 + c) * 64 * KW + (int)threadIdx.x];
            v[4 * c] = q.x; v[4 * c + 1] = q.y; v[4 * c + 2] = q.z; v[4 * c + 3] = q.w;
        }
    };
    unsigned long long pq[4] = {0ull, 0ull, 0ull, 0ull};
    if (a.k1_prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pq[0] = __builtin_readcyclecounter(); }
    bool occ_on = cached && a.occlusion && !cam.ortho && cam.m[0][1] == 0.0f && cam.m[2][1] == 0.0f;
    const float bins_per_px = (float)MW_OCC_BINS / (float)a.W;
    if (occ_on) {
        // the slab: lowest and highest point of the room polygons
        float lo = 1e30f, hi = -1e30f;
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            if (pnv[r] == 0) continue;
            float v[12];
            load_pv(r, v);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < (pnv[r] & 0xFF)) { lo = fminf(lo, v[3 * k + 1]); hi = fmaxf(hi, v[3 * k + 1]); }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        if (lane == 0) { s_slab[wave] = lo; s_slab[KW + wave] = hi; }
        if (threadIdx.x == 0) s_occ_n = 0;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < KW; ++w) { lo = fminf(lo, s_slab[w]); hi = fmaxf(hi, s_slab[KW + w]); }
        if (a.k1_prof) pq[1] = __builtin_readcyclecounter();
        const float eye_y = -cam.m[1][3];           // the unpitched camera's up row is (0, 1, 0)
        occ_on = eye_y > lo + 1e-3f && eye_y < hi - 1e-3f;
        if (occ_on) {
            const float wc = 0.1f;
            // the eye in world space (the rotation part of the modelview is orthonormal)
            const float eye_x = -(cam.m[0][0] * cam.m[0][3] + cam.m[1][0] * cam.m[1][3] + cam.m[2][0] * cam.m[2][3]);
            const float eye_z = -(cam.m[0][2] * cam.m[0][3] + cam.m[1][2] * cam.m[1][3] + cam.m[2][2] * cam.m[2][3]);
            const float inv_p00 = 1.0f / cam.p00, inv_hp = 1.0f / (cam.halfw * cam.p00), px_per_bin = 1.0f / bins_per_px;
#pragma unroll
            for (int r = 0; r < PR; ++r) {
                if (pnv[r] != 4) continue;          // triangles, and the quads of static entities (flag bit), are no walls
                float vx[4], vy[4], vz[4], v[12];
                load_pv(r, v);
                bool ys = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vx[k] = v[3 * k]; vy[k] = v[3 * k + 1]; vz[k] = v[3 * k + 2];
                    ys &= vy[k] == lo || vy[k] == hi;
                }
                // a vertical rectangle from lo to hi: two vertical edges
                const bool pa = vx[0] == vx[1] && vz[0] == vz[1] && vx[2] == vx[3] && vz[2] == vz[3] && vy[0] != vy[1] && vy[2] != vy[3];
                const bool pb = vx[1] == vx[2] && vz[1] == vz[2] && vx[3] == vx[0] && vz[3] == vz[0] && vy[1] != vy[2] && vy[3] != vy[0];
                if (!ys || !(pa || pb)) continue;
                const float bx = pa ? vx[2] : vx[1], bz = pa ? vz[2] : vz[1];
                if (bx == vx[0] && bz == vz[0]) continue;
                // drawn at all?  GL_CCW front faces (miniworld.py:512): the winding normal, s * (tz, 0, -tx) for a vertical
                // rectangle over the foot line B0 -> B1 = (tx, tz), points at the eye — by a centimetre at least
                const float tx = bx - vx[0], tz = bz - vz[0];
                const float sgn = pa ? vy[1] - vy[0] : vy[1] - vy[2];
                const float side = tz * (eye_x - vx[0]) - tx * (eye_z - vz[0]);
                const float facing = sgn > 0.0f ? side : -side;
                if (!(facing > 0.0f && facing * facing > 1e-4f * (tx * tx + tz * tz))) continue;
                // its foot line in eye space: (x, depth) of the two vertical edges, cut at depth wc
                float ea = fmaf(cam.m[0][0], vx[0], fmaf(cam.m[0][2], vz[0], cam.m[0][3]));
                float wa = -fmaf(cam.m[2][0], vx[0], fmaf(cam.m[2][2], vz[0], cam.m[2][3]));
                float eb = fmaf(cam.m[0][0], bx, fmaf(cam.m[0][2], bz, cam.m[0][3]));
                float wb = -fmaf(cam.m[2][0], bx, fmaf(cam.m[2][2], bz, cam.m[2][3]));
                if (!(wa >= wc) && !(wb >= wc)) continue;
                // (hardware reciprocals: their last-bit error moves a column by 1e-5 px, the margins are 0.05)
                if (!(wa >= wc)) { const float t = (wc - wa) * __builtin_amdgcn_rcpf(wb - wa); ea = fmaf(t, eb - ea, ea); wa = wc; }
                else if (!(wb >= wc)) { const float t = (wc - wb) * __builtin_amdgcn_rcpf(wa - wb); eb = fmaf(t, ea - eb, eb); wb = wc; }
                if (!(fmaxf(wa, wb) < 95.0f)) continue;         // the far plane is at 100
                const float xa = cam.halfw * fmaf(cam.p00, ea * __builtin_amdgcn_rcpf(wa), 1.0f);
                const float xb = cam.halfw * fmaf(cam.p00, eb * __builtin_amdgcn_rcpf(wb), 1.0f);
                const float xl = fminf(xa, xb) + 0.05f, xr = fmaxf(xa, xb) - 0.05f;
                if (!(xr > 0.0f && xl < (float)a.W && xr - xl >= px_per_bin)) continue;
                // depth along the wall as a function of the pixel column: A ex + B w = D with ex / w = (x / halfw - 1) / p00
                float A = wb - wa, B = -(eb - ea), D = A * ea + B * wa;
                if (D < 0.0f) { A = -A; B = -B; D = -D; }
                if (!(D > 1e-4f)) continue;
                const float A2 = A * inv_hp, B2 = B - A * inv_p00;
                const float dl = fmaf(A2, xl, B2), dr = fmaf(A2, xr, B2);
                if (!(dl * 100.0f > D && dr * 100.0f > D)) continue;       // depths below 100 at both ends (and positive denominators)
                const int j = atomicAdd(&s_occ_n, 1);
                if (j < MW_OCC_CAP) {
                    float *ow = s_occ_wall + 5 * j;
                    ow[0] = xl; ow[1] = xr; ow[2] = A2; ow[3] = B2; ow[4] = D;
                }
            }
        }
        __syncthreads();
        if (a.k1_prof) pq[2] = __builtin_readcyclecounter();
        if (occ_on) {
            const int n_occ = s_occ_n < MW_OCC_CAP ? s_occ_n : MW_OCC_CAP;
            static_assert(MW_OCC_BINS == 64 * KW, "one column bin per thread");
            const int b = (int)threadIdx.x;
            const float xa = (float)b / bins_per_px, xb = (float)(b + 1) / bins_per_px;
            float z = 1e30f;
#pragma unroll 4
            for (int j = 0; j < n_occ; ++j) {
                const float *ow = s_occ_wall + 5 * j;
                const float far = ow[4] * fmaxf(__builtin_amdgcn_rcpf(fmaf(ow[2], xa, ow[3])), __builtin_amdgcn_rcpf(fmaf(ow[2], xb, ow[3])));
                z = (xa >= ow[0] && xb <= ow[1]) ? fminf(z, far) : z;
            }
            z *= 1.0001f;
            s_occ_z[b] = z;
            // the largest value of every group of 16 bins (= 16 consecutive lanes)
            float gz = z;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) gz = fmaxf(gz, __shfl_xor(gz, o));
            if ((lane & 15) == 0) s_occ_z[MW_OCC_BINS + (b >> 4)] = gz;
        }
        __syncthreads();
    }
    const unsigned long long ptocc = a.k1_prof ? __builtin_readcyclecounter() : 0ull;
    const bool dense_write = a.max_vis <= MW_SORT_CAP;      // visible polygons are listed first and set up with every lane busy
    // one round of 64 * KW polygons: visibility, ordered compaction across the waves, list entry (or the record at once)
    auto list_round = [&](const float (&v)[12], int nvflags, int i, int round) {
        bool vis = false;
        HV h[4];
        PolyGeom g;
        const int nv = nvflags & 0xFF;
        if (i < np) {
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = xform(cam, v[3 * k], v[3 * k + 1], v[3 * k + 2]);
            vis = cull_poly(a, h, nv, g) && !((nvflags & MW_POLY_ENTITY) && (view_flags & 4));     // the queries draw rooms only
            if (vis && occ_on) vis = !occluded(s_occ_z, h, nv, bins_per_px);
        }
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
        const int idx = count + before + __popcll((unsigned long long)(m & ((1ull << lane) - 1ull)));
        count += total;
        if (vis) {
            if (idx >= a.max_vis) {
                atomicOr(a.status, MW_ST_VIS_OVERFLOW);
            } else if (dense_write) {
                s_list[idx] = (uint16_t)i;
            } else {
                const mw_poly &q = polys[i];
                float col[3];
                light(cam, q.n, q.rgb, col);
                const float uv[3][2] = {{q.uv[0][0], q.uv[0][1]}, {q.uv[1][0], q.uv[1][1]}, {q.uv[2][0], q.uv[2][1]}};
                write_poly(a, env, idx, (uint32_t)idx, h, nv, g, uv, col, q.tex, s_zmin);
            }
        }
    };
    if (cached && dense_write) {
        // the usual case: both rounds decided first, one ordered compaction (one barrier) for the two
        bool vis[PR];
        uint64_t vm[PR];
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            const int i = r * 64 * KW + wave * 64 + lane;
            vis[r] = false;
            if (i < np) {
                float v[12];
                HV h[4];
                PolyGeom g;
                load_pv(r, v);
                const int nv = pnv[r] & 0xFF;
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = xform(cam, v[3 * k], v[3 * k + 1], v[3 * k + 2]);
                vis[r] = cull_poly(a, h, nv, g) && !((pnv[r] & MW_POLY_ENTITY) && (view_flags & 4));     // the queries draw rooms only
                if (vis[r] && occ_on) vis[r] = !occluded(s_occ_z, h, nv, bins_per_px);
            }
            vm[r] = ballot(vis[r]);
            if (lane == 0) s_cnt[r][wave] = __popcll((unsigned long long)vm[r]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < PR; ++r) {
            int before = 0, total = 0;
#pragma unroll
            for (int w = 0; w < KW; ++w) {
                const int cw = s_cnt[r][w];
                before += w < wave ? cw : 0;
                total += cw;
            }
            const int idx = count + before + __popcll((unsigned long long)(vm[r] & ((1ull << lane) - 1ull)));
            count += total;
            if (vis[r]) {
                if (idx < a.max_vis) s_list[idx] = (uint16_t)(r * 64 * KW + wave * 64 + lane);
                else atomicOr(a.status, MW_ST_VIS_OVERFLOW);
            }
        }
    } else if (cached) {
#pragma unroll
        for (int r = 0; r < PR; ++r)
            if (r * 64 * KW < np) {
                float v[12];
                load_pv(r, v);
                list_round(v, pnv[r], r * 64 * KW + wave * 64 + lane, r);
            }
    } else {
        for (int base = 0, round = 0; base < np; base += 64 * KW, ++round) {
            const int i = base + wave * 64 + lane;
            float v[12];
            int nvflags = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) v[k] = i < np ? polys[i].v[k / 3][k % 3] : 0.0f;
            if (i < np) nvflags = polys[i].nv;
            list_round(v, nvflags, i, round);
        }
    }
    if (dense_write) {
        // the records of the listed polygons: a few dozen of the hundreds looked at, one per thread
        __syncthreads();
        const int n_room = count < a.max_vis ? count : a.max_vis;
        // (wave 1 takes the first 64: wave 0 goes on to the entities meanwhile)
        for (int t = ((int)threadIdx.x + 64 * (KW - 1)) % (64 * KW); t < n_room; t += 64 * KW) {
            mw_poly q = polys[s_list[t]];
            q.nv &= 0xFF;
            HV h[4];
            PolyGeom g;
#pragma unroll
            for (int k = 0; k < 4; ++k) h[k] = xform(cam, q.v[k][0], q.v[k][1], q.v[k][2]);
            (void)cull_poly(a, h, q.nv, g);
            float col[3];
            light(cam, q.n, q.rgb, col);
            const float uv[3][2] = {{q.uv[0][0], q.uv[0][1]}, {q.uv[1][0], q.uv[1][1]}, {q.uv[2][0], q.uv[2][1]}};
            write_poly(a, env, t, (uint32_t)t, h, q.nv, g, uv, col, q.tex, s_zmin);
        }
    }
#else
    for (int base = 0; base < np; base += 64) {         // display list 1: rooms
        const int i = base + lane;
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
        const int idx = compact(lane, vis, count);
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
#endif
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
                    stale_n[0] = uni(mdp->last_n[0]); stale_n[1] = uni(mdp->last_n[1]); stale_n[2] = uni(mdp->last_n[2]);
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
        if (n <= 64 * KW) {
            // a short list (the usual case once the hidden polygons are gone): every thread ranks its own key —
            // the keys are distinct, their low bits being the list index
            const unsigned long long key = (int)threadIdx.x < n ? s_zmin[threadIdx.x] : ~0ull;
            int rank = 0;
            for (int j = 0; j < n; ++j) rank += s_zmin[j] < key ? 1 : 0;
            if ((int)threadIdx.x < n) ord[1 + rank] = (uint16_t)(key & 0xFFFFull);
            if (writer) ord[0] = 1;
        } else if (n <= MW_SORT_POW2) {
            // bitonic sort of the packed (bound, index) keys in LDS (more polygons than that in view: no visiting order,
            // the raster kernel then walks the list as it is)
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
        pp[5] = (unsigned long long)count;
#if MW_SORT_VIS
        pp[6] = (unsigned long long)s_occ_n | ((pq[0] - pt2) << 8); pp[7] = (ptocc - pt2) | ((pq[1] - pt2) << 20) | ((pq[2] - pt2) << 40);
#endif
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
        if (remove_slot >= 0) {
            if (a.task == MW_TASK_COLLECT) mw::collect_respawn(a, env, c.set, remove_slot, c.px, c.pz);     // the kit respawns
            else a.ekind[(size_t)remove_slot * a.N + env] = MW_ENT_NONE;
        }
    }
}
