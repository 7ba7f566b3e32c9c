// K1, dense variant — step + primitive setup for SMALL scenes, one lane per (env, primitive slot).
//
// Same work and same arithmetic as mw_setup.hip (which see for the reference file:line map), other lane mapping.
// The wave-per-env kernel spends a wavefront's 64 lanes on one env: per-env scalar work (f64 physics, camera) is
// repeated 64 times, and the polygon batches of a single room run with 6 of 64 lanes active, twice (rooms, boxes);
// 4096 Hallway envs are 4096 waves of ~72 k cycles each, four per SIMD, latency bound.
// Here an env owns L = max_polys + 6 * max_ents consecutive lanes (Hallway / OneRoom: 6 + 6 = 12, five envs per
// wave): every lane of an env evaluates the env's step and camera itself — the same instruction stream as before,
// now serving five envs — and then sets up ITS primitive: room polygon `slot`, or face (slot - max_polys) % 6 of
// entity (slot - max_polys) / 6 when that is a Box.  One pass, ~60 lanes busy, an ordered compaction per env
// through ballot masks (draw order: rooms, static entities, dynamic entities; miniworld.py:1058-1077), one fifth
// of the waves, no cross-lane traffic and no spills (up to 256 VGPRs: with fewer waves than SIMDs occupancy is moot).
// The leading lane of an env writes its state, flags, header and — on an episode's end — runs the generator.
//
// Mesh entities reserve their draw-id ranges and are described in the env header like in mw_setup.hip.
// Not handled here (the engine launches mw_setup.hip instead): top / proxy views, scenes whose L exceeds 64, big scenes
// with a visiting order, MW_TASK_COLLECT.
#include "mw_setup_common.h"

// MW_DENSE_MESH = 1: the instantiation for engines with meshes (mw_setup_dense_mesh*.hip): the two mesh walks are
// compiled in and the camera comes first (the walk needs it, and the primitive's object-space data would otherwise be
// live across it: 64 spilled dwords); the plain instantiation carries none of it (175 VGPRs, no scratch).
#ifndef MW_DENSE_MESH
#define MW_DENSE_MESH 0
#endif
#ifndef MW_DENSE_KERNEL_NAME
#define MW_DENSE_KERNEL_NAME mw_step_setup_dense_kernel
#endif

namespace {
// whole-entity frustum cull of a mesh entity: the bounding sphere of the scaled mesh about its origin against the near
// and the four side planes, conservative — a skipped mesh has no pixel
__device__ inline bool mesh_in_view(const MwArgs &a, const StepCtx &c, const Cam &cam, int env, int es, const MwMeshDesc *mdp)
{
    const float brad = __uint_as_float(mdp->bound_bits) * (float)ent_geom(a, env, es, 6) * 1.001f + 1e-3f;
    const float wx = (float)ent_pos(c, es, 0), wy = (float)ent_pos(c, es, 1), wz = (float)ent_pos(c, es, 2);
    const float ex = fmaf(cam.m[0][0], wx, fmaf(cam.m[0][1], wy, fmaf(cam.m[0][2], wz, cam.m[0][3])));
    const float ey = fmaf(cam.m[1][0], wx, fmaf(cam.m[1][1], wy, fmaf(cam.m[1][2], wz, cam.m[1][3])));
    const float ez = fmaf(cam.m[2][0], wx, fmaf(cam.m[2][1], wy, fmaf(cam.m[2][2], wz, cam.m[2][3])));
    const float w = -ez;
    const float lx = sqrtf(fmaf(cam.p00, cam.p00, 1.0f)), ly = sqrtf(fmaf(cam.p11, cam.p11, 1.0f));
    return !(w + brad < 0.04f) && !(w - fabsf(cam.p00 * ex) < -(brad * lx)) && !(w - fabsf(cam.p11 * ey) < -(brad * ly));
}
}  // namespace

extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void MW_DENSE_KERNEL_NAME(
    MwArgs a, int do_step, int lanes_per_env, const int32_t *__restrict__ actions, float *__restrict__ reward,
    uint8_t *__restrict__ term, uint8_t *__restrict__ trunc)
{
    __shared__ unsigned char gen_ws[MW_GEN_WS_BYTES];      // generator scratch (used by the Maze generator only)
    const int lane = threadIdx.x;
    const int L = lanes_per_env;
    const int epw = 64 / L;                                                   // envs per wavefront
    const int el = (int)(((uint32_t)lane * ((65536u + (uint32_t)L - 1u) / (uint32_t)L)) >> 16);     // lane / L, exact for lane < 64
    const int slot = lane - el * L;
    // spare mode: blocks appended to the grid regenerate the spare worlds consumed in earlier steps, beside the step
    const int env_blocks = (a.N + epw - 1) / epw;
    if ((int)blockIdx.x >= env_blocks) {
        mw::refill_spares(a, (int)blockIdx.x - env_blocks, lane, gen_ws);
        return;
    }
    const int env = (int)blockIdx.x * epw + el;
    if (el >= epw || env >= a.N) return;
    const bool leader = slot == 0;
    // MW_K1_PROF (perf experiments only): cycle stamps of the phases, written by the env's leading lane
    const bool prof = a.k1_prof != nullptr;
    unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long wall0 = prof ? wall_clock64() : 0ull;
    if (prof) pt[0] = __builtin_readcyclecounter();
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
        if (prof) pt[1] = __builtin_readcyclecounter();
        if (a.autoreset == MW_AUTORESET_SAME_STEP && a.generator != MW_GEN_NONE && (tm | tr)) {
            // same-step auto-reset: the observation returned with done = 1 is the first one of the next episode.
            // The env's leading lane installs the next world (several envs of the wave may do so side by side); the
            // env's other lanes then read it like the leader does.
            if (leader) {
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

    {
        // the frame's vertex half is mw_geom_kernel's (mw_geom.hip); see mw_setup.hip
        if (leader && do_step) a.pending_remove[env] = remove_slot;
        return;
    }
    if (prof) pt[2] = __builtin_readcyclecounter();
    // ---- camera (superseded) ---------------------------------------------------------------------
    Cam cam;
    float sky[3];
    build_camera(a, env, c.px, c.py, c.pz, c.dir, cam, sky, false);

#if MW_DENSE_MESH
    // ---- mesh entities, first walk: each reserves a range of draw ids at its place in the drawing order (static
    // entities first, then dynamic ones, each in slot order: miniworld.py:1058-1060, 1075-1077) and is described to the
    // mesh raster kernel in the env header; a box drawn after it has its draw id shifted by the triangles before it.
    // Every lane of the env walks the (few) slots; the leading lane writes the table.
    int mesh_tris = 0, n_mesh = 0, mesh_before = 0;
    float *hdr = a.envhdr + (size_t)env * MW_ENVHDR;
    {
        const int my_es = slot >= a.max_polys ? (slot - a.max_polys) / 6 : -1;
        const int my_cls = (my_es >= 0 && my_es < a.E && a.ekind[(size_t)my_es * a.N + env] == MW_ENT_BOX)
                               ? (a.estatic[(size_t)my_es * a.N + env] ? 1 : 2) : 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int es = 0; es < a.E; ++es) {
                const int kind = a.ekind[(size_t)es * a.N + env];
                if (kind == MW_ENT_NONE || (a.estatic[(size_t)es * a.N + env] != 0) != (pass == 0)) continue;
                if (es == my_es && my_cls == 1 + pass) mesh_before = mesh_tris;         // this lane's box is drawn here
                if (kind != MW_ENT_MESH) continue;
                const MwMeshDesc *mdp = a.mesh + a.emesh[(size_t)es * a.N + env];
                if (!mesh_in_view(a, c, cam, env, es, mdp)) continue;
                const int md_ntris = (int)mdp->ntris;
                if (n_mesh < MW_MAX_MESH_ENTS && L + mesh_tris + md_ntris < 0xFFF0) {
                    const double edir = (es == c.live) ? c.cdir : a.edir[(size_t)es * a.N + env];
                    const mw::SinCos sc = mw::sincos_det(edir);
                    if (leader) {
                        float *m = hdr + MW_HDR_MESH + 12 * n_mesh;     // m[1], the first draw id, follows in the second walk
                        m[0] = __int_as_float(es);
                        m[2] = __int_as_float(md_ntris);
                        m[3] = __int_as_float((int)mdp->first);
                        m[4] = (float)sc.c; m[5] = (float)sc.s;
                        m[6] = (float)ent_geom(a, env, es, 6);
                        m[7] = (float)ent_pos(c, es, 0); m[8] = (float)ent_pos(c, es, 1); m[9] = (float)ent_pos(c, es, 2);
                        m[10] = __int_as_float((int)mdp->tex);
                        m[11] = 0.0f;
                    }
                    mesh_tris += md_ntris;
                    ++n_mesh;
                } else {
                    atomicOr(a.status, MW_ST_VIS_OVERFLOW);
                }
            }
        }
    }

#else
    int mesh_tris = 0, n_mesh = 0, mesh_before = 0;
    float *hdr = a.envhdr + (size_t)env * MW_ENVHDR;
#endif

    // ---- this lane's primitive: object-space data (after the camera: held across it, it cost 56 spilled dwords) ----
    const mw_poly *polys = a.polys + (size_t)c.set * a.max_polys;
    const int np = a.npolys[c.set];
    bool have = false;
    int cls = 0;                    // 0 room polygon, 1 face of a static box, 2 face of a dynamic box
    float wv[4][3];                 // world-space vertices as glVertex3f receives them
    float nrm[3] = {0.0f, 1.0f, 0.0f}, base[3] = {0.0f, 0.0f, 0.0f};
    float uv[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    int nv = 4, tex = -1;
    if (slot < a.max_polys) {
        if (slot < np) {            // display list 1: the rooms (and static frame quads), in list order
            const mw_poly q = polys[slot];
            nv = q.nv & 0xFF;
            tex = q.tex;
#pragma unroll
            for (int k = 0; k < 4; ++k) { wv[k][0] = q.v[k][0]; wv[k][1] = q.v[k][1]; wv[k][2] = q.v[k][2]; }
#pragma unroll
            for (int k = 0; k < 3; ++k) { uv[k][0] = q.uv[k][0]; uv[k][1] = q.uv[k][1]; nrm[k] = q.n[k]; base[k] = q.rgb[k]; }
            have = true;
        }
    } else {
        const int bi = slot - a.max_polys;
        const int es = bi / 6, f = bi - es * 6;
        if (es < a.E && a.ekind[(size_t)es * a.N + env] == MW_ENT_BOX) {
            // Box.render (entity.py:409-432): T(pos) R_y(dir) drawBox(...)
            cls = a.estatic[(size_t)es * a.N + env] ? 1 : 2;
            const double edir = (es == c.live) ? c.cdir : a.edir[(size_t)es * a.N + env];
            const float ex = (float)ent_pos(c, es, 0), ey = (float)ent_pos(c, es, 1), ez = (float)ent_pos(c, es, 2);
            const float hx = (float)(ent_geom(a, env, es, 0) / 2), sy = (float)ent_geom(a, env, es, 1),
                        hz = (float)(ent_geom(a, env, es, 2) / 2);
            base[0] = (float)ent_geom(a, env, es, 3); base[1] = (float)ent_geom(a, env, es, 4); base[2] = (float)ent_geom(a, env, es, 5);
            const mw::SinCos sc = mw::sincos_det(edir);
            const float cs = (float)sc.c, sn = (float)sc.s;
            const float lo[3] = {-hx, 0.0f, -hz};
            const float hi[3] = {hx, sy, hz};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sel = kBoxSel[f][k];
                const float lx = (sel & 1) ? hi[0] : lo[0];
                const float ly = (sel & 2) ? hi[1] : lo[1];
                const float lz = (sel & 4) ? hi[2] : lo[2];
                wv[k][0] = fmaf(cs, lx, sn * lz) + ex;
                wv[k][1] = ly + ey;
                wv[k][2] = fmaf(cs, lz, -(sn * lx)) + ez;
            }
            nrm[0] = fmaf(cs, kBoxN[f][0], sn * kBoxN[f][2]);
            nrm[1] = kBoxN[f][1];
            nrm[2] = fmaf(cs, kBoxN[f][2], -(sn * kBoxN[f][0]));
            have = true;
        }
    }
    if (prof) pt[3] = __builtin_readcyclecounter();

    // ---- transform, cull, light: once, whatever the lane holds ---------------------------------------------
    bool vis = false;
    HV h[4];
    PolyGeom g;
    float col[3] = {0.0f, 0.0f, 0.0f};
    if (have) {
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = xform(cam, wv[k][0], wv[k][1], wv[k][2]);
        vis = cull_poly(a, h, nv, g);
        light(cam, nrm, base, col);
    }
    if (prof) pt[4] = __builtin_readcyclecounter();
    // ---- ordered compaction within the env: rooms, static boxes, dynamic boxes, each in slot order ----------
    const uint64_t env_mask = (L >= 64 ? ~0ull : ((1ull << L) - 1ull)) << (el * L);
    const uint64_t below = (1ull << lane) - 1ull;
    const uint64_t mv = ballot(vis) & env_mask;
    const uint64_t m_room = ballot(cls == 0), m_sbox = ballot(cls == 1);
    int idx;
    if (cls == 0) idx = __popcll((unsigned long long)(mv & m_room & below));
    else if (cls == 1) idx = __popcll((unsigned long long)(mv & m_room)) + __popcll((unsigned long long)(mv & m_sbox & below));
    else idx = __popcll((unsigned long long)(mv & (m_room | m_sbox))) + __popcll((unsigned long long)(mv & ~(m_room | m_sbox) & below));
    const int count = __popcll((unsigned long long)mv);
    // ---- mesh entities, second walk: now that the visible primitives are counted, each mesh's first draw id =
    // visible primitives drawn before it (the rooms; the static boxes — all of them for a dynamic mesh, those in lower
    // slots for a static one; the dynamic boxes in lower slots) + mesh triangles before it
    if (MW_DENSE_MESH && n_mesh > 0) {
        int tris = 0, j = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int es = 0; es < a.E; ++es) {
                const int kind = a.ekind[(size_t)es * a.N + env];
                if (kind != MW_ENT_MESH || (a.estatic[(size_t)es * a.N + env] != 0) != (pass == 0)) continue;
                const MwMeshDesc *mdp = a.mesh + a.emesh[(size_t)es * a.N + env];
                if (!mesh_in_view(a, c, cam, env, es, mdp)) continue;
                const int md_ntris = (int)mdp->ntris;
                if (!(j < MW_MAX_MESH_ENTS && L + tris + md_ntris < 0xFFF0)) continue;
                const uint64_t lower = ((1ull << (a.max_polys + 6 * es)) - 1ull) << (el * L);
                const uint64_t m_dbox = ~(m_room | m_sbox);
                const uint64_t before_m = pass == 0 ? (m_room | (m_sbox & lower)) : (m_room | m_sbox | (m_dbox & lower));
                if (leader) hdr[MW_HDR_MESH + 12 * j + 1] = __int_as_float(__popcll((unsigned long long)(mv & before_m)) + tris);
                tris += md_ntris;
                ++j;
            }
        }
    }
    if (vis) {
        if (idx < a.max_vis) write_poly(a, env, idx, (uint32_t)(idx + (cls == 0 ? 0 : mesh_before)), h, nv, g, uv, col, tex, nullptr);
        else atomicOr(a.status, MW_ST_VIS_OVERFLOW);
    }
    if (leader) {
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
        hdr[23] = __int_as_float(0); hdr[27] = 0.0f; hdr[31] = 0.0f;
        if (remove_slot >= 0) a.ekind[(size_t)remove_slot * a.N + env] = MW_ENT_NONE;
    }
    if (prof && leader) {
        __builtin_amdgcn_s_waitcnt(0);          // the record stores have left the wave
        pt[5] = __builtin_readcyclecounter();
        unsigned long long *pp = a.k1_prof + (size_t)env * 8;
        pp[0] = pt[1] - pt[0]; pp[1] = pt[2] - pt[1]; pp[2] = pt[3] - pt[2]; pp[3] = pt[4] - pt[3]; pp[4] = pt[5] - pt[4];
        pp[5] = (unsigned long long)(tm | tr);
        pp[6] = wall0; pp[7] = wall_clock64();
    }
}
