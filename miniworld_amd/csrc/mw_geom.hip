// KG — the vertex half of a frame: camera, lighting, transform, clipping, triangle setup.  One wavefront per environment.
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
// with the arithmetic of mw_glmath.h (Mesa 23.2.1 / llvmpipe, measured).  Output: per env a list of triangle records in
// drawing order (mw_records.h), the env header (sky colour, mesh-entity table), and the entity removals the step left pending
// (a picked-up object is still drawn in the frame of the step that picked it up: pickupobjects.py:86-88 runs after :717).
//
// Lanes: one GL primitive (polygon / box face) per lane and round.  A primitive yields up to two triangles; a triangle that
// needs clipping takes one of 8 work slots in LDS (two vertex lists), the lanes with such a triangle go through the slots in
// batches.  List positions come from an exclusive scan of per-lane UPPER BOUNDS (a triangle cut by k planes becomes at most
// k + 1): drawing order is kept, what clipping or culling removes is left as a NULL record that touches no tile.
#include "mw_setup_common.h"
#include "mw_records.h"

namespace {

constexpr int kClipSlots = 8;

__device__ inline int wave_excl_scan(int v, int lane, int &total)
{
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    total = __shfl(x, 63);
    return x - v;
}

// Draw ids order the frame's triangles for the depth test's ties (GL_LESS: the first drawn wins) and name them in the
// sample keys: a record's id is its list position plus the mesh triangles drawn before it (a mesh entity takes one id
// per triangle); the proxy boxes of get_visible_ents carry their entity's tag instead.
struct Emit {
    const MwArgs &a;
    int env, S, tex;
    uint32_t id_base, tag;
    int idx, end;            // next list position / one past this triangle's range
    __device__ void operator()(const mwgl::TriSetup &t)
    {
        if (idx < end && idx < a.max_vis) mwrec::write_tri(a, env, idx, tag ? tag : (uint32_t)idx + id_base, t, tex, S);
        ++idx;
    }
};

// upper bound of the triangles (a, b, c) turns into
__device__ inline int tri_bound(const mwgl::Vert &a, const mwgl::Vert &b, const mwgl::Vert &c)
{
    const uint32_t m = a.clipmask | b.clipmask | c.clipmask;
    if (a.clipmask & b.clipmask & c.clipmask) return 0;
    return 1 + __popc(m);
}

// One round: every lane holds a primitive of `nt` triangles (0: none) given as vertex indices into v[4]; emits them at
// list positions [base, base + bound) in order and fills what stays unused with NULL records.
template <bool GOURAUD>
__device__ inline void emit_round(const MwArgs &a, const mwgl::Frame &f, int env, int lane, int S, const mwgl::Vert v[4], int nt,
                                  const int tri[2][3], int tex, uint32_t id_base, uint32_t tag, int &count, mwgl::Vert (*s_clip)[2][MWGL_MAX_CLIP_VERTS])
{
    int bound[2] = {0, 0};
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < nt) bound[t] = tri_bound(v[tri[t][0]], v[tri[t][1]], v[tri[t][2]]);
    int total;
    const int base = count + wave_excl_scan(bound[0] + bound[1], lane, total);
    count += total;
    if (base + bound[0] + bound[1] > a.max_vis && (bound[0] | bound[1])) atomicOr(a.status, MW_ST_VIS_OVERFLOW);
    const bool ms = S > 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int b0 = base + (t ? bound[0] : 0);
        Emit em{a, env, S, tex, id_base, tag, b0, b0 + bound[t]};
        const mwgl::Vert &va = v[tri[t][0]], &vb = v[tri[t][1]], &vc = v[tri[t][2]];
        const bool clipped = bound[t] > 1;
        if (bound[t] == 1) {
            mwgl::TriSetup ts;
            if (mwgl::setup_triangle(va, vb, vc, ms, tex >= 0, ts)) em(ts);
        }
        // triangles that cross a frustum plane: kClipSlots lanes at a time through the LDS work lists
        uint64_t pend = __ballot(clipped);
        while (pend) {
            // the batch: the lowest kClipSlots set bits
            uint64_t batch = 0ull, rest = pend;
            for (int k = 0; k < kClipSlots && rest; ++k) { batch |= rest & (0ull - rest); rest &= rest - 1ull; }
            if (clipped && ((batch >> lane) & 1ull)) {
                const int slot = __popcll((unsigned long long)(batch & ((1ull << lane) - 1ull)));
                mwgl::Vert *r;
                const int n = mwgl::clip_triangle<GOURAUD>(f, va, vb, vc, s_clip[slot][0], s_clip[slot][1], &r);
                for (int i = 2; i < n; ++i) {
                    mwgl::TriSetup ts;
                    if (mwgl::setup_triangle(r[i - 1], r[i], r[0], ms, tex >= 0, ts)) em(ts);
                }
            }
            pend = rest;
        }
        for (int i = em.idx; i < em.end && i < a.max_vis; ++i) mwrec::write_null(a, env, i);
    }
}

}  // namespace

// view_flags: bit 0 top view, bit 1 draw the agent marker, bit 2 get_visible_ents' proxy pass (rooms untextured + one
// 0.2 m box per entity, tagged 0x10000 | slot).  S: samples per pixel of the target (1, 4, 8, 16).
extern "C" __global__ __launch_bounds__(64) void mw_geom_kernel(MwArgs a, int view_flags, int S)
{
    __shared__ mwgl::Vert s_clip[kClipSlots][2][MWGL_MAX_CLIP_VERTS];
    const int env = a.env_base + blockIdx.x;
    const int lane = threadIdx.x;
    const int set = a.shared_geom ? 0 : env;
    const bool top = (view_flags & 1) != 0, proxy = (view_flags & 4) != 0;
    // ---- the frame's GL state (every lane evaluates it: same instruction stream)
    mwgl::Frame f;
    const double px = a.ax[env], py = a.ay[env], pz = a.az[env], dir = a.adir[env];
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
    }
    mwgl::frame_finish(f, a.W, a.H, lpos, lcol, lamb);
    mwgl::Xform cam;
    mwgl::make_xform(f, f.view, f.view_flags, cam);

    int count = 0;
    float stale_n[3] = {0.0f, 1.0f, 0.0f};
    const mw_poly *polys = a.polys + (size_t)set * a.max_polys;
    const int np = a.npolys[set];
    const float white[3] = {1.0f, 1.0f, 1.0f};
    // ---- display list 1: rooms, frames
    for (int base = 0; base < np; base += 64) {
        const int i = base + lane;
        mwgl::Vert v[4];
        int nt = 0, tex = -1;
        int tri[2][3] = {{0, 1, 2}, {0, 2, 3}};
        if (i < np) {
            const mw_poly q = polys[i];
            const int nv = q.nv & 0xFF;
            if (!(proxy && (q.nv & MW_POLY_ENTITY))) {       // the queries draw rooms only
                mwgl::Xform ex;
                const bool own = (q.nv & MW_POLY_XF) != 0;
                if (own) mwgl::entity_xform(f, q.xf, q.xf[3], 1.0f, false, ex);
                const mwgl::Xform &x = own ? ex : cam;
                float col[3];
                mwgl::light_vertex(f, x, q.n, proxy ? white : q.rgb, col);
                tex = proxy ? -1 : q.tex;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < nv) {
                        mwgl::transform_vertex(f, x, q.v[k], v[k]);
                        v[k].st[0] = tex >= 0 ? q.uv[k][0] : 0.0f; v[k].st[1] = tex >= 0 ? q.uv[k][1] : 0.0f;
                        v[k].col[0] = col[0]; v[k].col[1] = col[1]; v[k].col[2] = col[2];
                    }
                }
                if (nv == 3) { nt = 1; tri[0][0] = 1; tri[0][1] = 2; tri[0][2] = 0; }
                else if (q.nv & MW_POLY_QUAD) { nt = 2; tri[0][0] = 0; tri[0][1] = 1; tri[0][2] = 3; tri[1][0] = 1; tri[1][1] = 2; tri[1][2] = 3; }
                else { nt = 2; tri[0][0] = 1; tri[0][1] = 2; tri[0][2] = 0; tri[1][0] = 2; tri[1][1] = 3; tri[1][2] = 0; }
            }
        }
        emit_round<false>(a, f, env, lane, S, v, nt, tri, tex, 0u, 0u, count, s_clip);
    }
    if (np > 0) { stale_n[0] = polys[np - 1].n[0]; stale_n[1] = polys[np - 1].n[1]; stale_n[2] = polys[np - 1].n[2]; }

    // ---- entities: static ones first (inside display list 1), then dynamic ones, each in slot order
    float *hdr = a.envhdr + (size_t)env * MW_ENVHDR;
    int n_mesh = 0, mesh_tris = 0;
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
    if (proxy) { box_m |= mesh_m | frame_m; mesh_m = 0ull; static_m = ~0ull; }
    for (int pass = 0; pass < 2; ++pass) {
        const uint64_t mine_m = pass == 0 ? static_m : ~static_m;
        const uint64_t mesh_mine = mesh_m & mine_m, box_mine = box_m & mine_m;
        int s0 = 0;
        while (s0 < a.E) {
            if ((mesh_mine >> s0) & 1ull) {
                // a mesh entity: its triangles belong to the mesh kernel; here its place in the drawing order and its transform
                const int mid = a.emesh[(size_t)s0 * a.N + env];
                const MwMeshDesc *mdp = a.mesh + mid;
                const int md_ntris = (int)mdp->ntris;
                const float pos[3] = {(float)a.epos[((size_t)0 * a.E + s0) * a.N + env], (float)a.epos[((size_t)1 * a.E + s0) * a.N + env],
                                      (float)a.epos[((size_t)2 * a.E + s0) * a.N + env]};
                const float scale = (float)a.egeom[((size_t)6 * a.E + s0) * a.N + env];
                mwgl::Xform ex;
                mwgl::entity_xform(f, pos, (float)(a.edir[(size_t)s0 * a.N + env] * 180 / kPi), scale, true, ex);
                // whole-entity frustum test on the bounding sphere (conservative): clip-space distance to the five planes
                bool in_view = true;
                if (!top) {
                    const float brad = __uint_as_float(mdp->bound_bits) * scale * 1.001f + 1e-3f;
                    mwgl::Vert o;
                    const float zero[3] = {0.0f, 0.0f, 0.0f};
                    mwgl::transform_vertex(f, ex, zero, o);
                    const float w = o.clip[3];
                    const float p00 = f.proj.m[0], p11 = f.proj.m[5];
                    const float lx = sqrtf(fmaf(p00, p00, 1.0f)), ly = sqrtf(fmaf(p11, p11, 1.0f));
                    in_view = !(w + brad < 0.04f) && !(w - fabsf(o.clip[0]) < -(brad * lx)) && !(w - fabsf(o.clip[1]) < -(brad * ly));
                }
                if (in_view) {
                    if (n_mesh < MW_MAX_MESH_ENTS && count + mesh_tris + md_ntris < 0xFFF0) {
                        if (lane == 0) {
                            float *m = hdr + MW_HDR_MESH + MW_HDR_MESH_STRIDE * n_mesh;
                            m[0] = __int_as_float(s0);
                            m[1] = __int_as_float(count + mesh_tris);
                            m[2] = __int_as_float(md_ntris);
                            m[3] = __int_as_float((int)mdp->first);
                            m[4] = __int_as_float((int)mdp->tex);
                            m[5] = ex.nscale;
                            m[6] = ex.light[0]; m[7] = ex.light[1]; m[8] = ex.light[2];
#pragma unroll
                            for (int k = 0; k < 16; ++k) m[9 + k] = ex.mvp.m[k];
                        }
                        mesh_tris += md_ntris;      // one draw id per triangle (a mesh out of view takes none)
                        ++n_mesh;
                    } else {
                        atomicOr(a.status, MW_ST_VIS_OVERFLOW);
                    }
                }
                ++s0;
                continue;
            }
            // a run of up to 10 box slots (6 faces each), ending before the next mesh of this pass
            int s1 = s0 + 10 < a.E ? s0 + 10 : a.E;
            {
                const uint64_t ahead = mesh_mine >> s0;
                if (ahead) {
                    const int nxt = s0 + __builtin_ctzll(ahead);
                    s1 = nxt < s1 ? nxt : s1;
                }
            }
            const uint64_t run_boxes = (box_mine >> s0) & ((1ull << (s1 - s0)) - 1ull);
            if (run_boxes) {
                const int bi = lane / 6, fc = lane - bi * 6, slot = s0 + bi;
                const bool mine = lane < (s1 - s0) * 6 && ((run_boxes >> bi) & 1ull);
                mwgl::Vert v[4];
                int nt = 0;
                int tri[2][3] = {{0, 1, 3}, {1, 2, 3}};
                bool clipped_l = false;
                if (mine) {
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
                }
                // one glBegin / glEnd per box: a clipped vertex anywhere in the call sends all six faces through the draw
                // module's pipeline, whose quads split (0,1,3) (1,2,3); an immediate-mode call without one splits (0,1,2)
                // (0,2,3); a static box sits in display list 1 (always the first split)
                const uint64_t cm = __ballot(clipped_l);
                const bool box_clipped = ((cm >> (bi * 6)) & 0x3Full) != 0ull;
                const bool in_list = !proxy && mine && a.estatic[(size_t)slot * a.N + env] != 0;
                if (mine && !box_clipped && !in_list) { tri[0][0] = 0; tri[0][1] = 1; tri[0][2] = 2; tri[1][0] = 0; tri[1][1] = 2; tri[1][2] = 3; }
                emit_round<false>(a, f, env, lane, S, v, nt, tri, -1, (uint32_t)mesh_tris, proxy ? (0x10000u | (uint32_t)slot) : 0u, count, s_clip);
                stale_n[0] = 0.0f; stale_n[1] = -1.0f; stale_n[2] = 0.0f;     // drawBox ends with glNormal3f(0, -1, 0)
            }
            s0 = s1;
        }
    }
    if (view_flags & 2) {
        // Agent.render (entity.py:518-539): no glNormal3f => lit with the normal the last immediate-mode glNormal3f or the
        // end of the display list left current (glDrawArrays with a normal array leaves it alone)
        mwgl::Vert v[4];
        int nt = 0;
        const int tri[2][3] = {{0, 1, 2}, {0, 1, 2}};
        if (lane == 0) {
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
            for (int k = 0; k < 3; ++k) {
                mwgl::transform_vertex(f, cam, pv[k], v[k]);
                v[k].st[0] = v[k].st[1] = 0.0f;
                v[k].col[0] = col[0]; v[k].col[1] = col[1]; v[k].col[2] = col[2];
            }
            nt = 1;
        }
        emit_round<false>(a, f, env, lane, S, v, nt, tri, -1, (uint32_t)mesh_tris, 0u, count, s_clip);
    }
    if (lane == 0) {
        a.nvis[env] = count < a.max_vis ? count : a.max_vis;
        a.k3_cost[env] = mesh_tris;
        hdr[0] = (float)a.light[(size_t)0 * a.N + env]; hdr[1] = (float)a.light[(size_t)1 * a.N + env]; hdr[2] = (float)a.light[(size_t)2 * a.N + env];
        hdr[3] = __int_as_float(n_mesh);
#pragma unroll
        for (int i = 0; i < 3; ++i) { hdr[4 + i] = f.l_amb[i]; hdr[8 + i] = f.l_dif[i]; }
        // the step's pending removal: the picked-up object leaves the entity list after its last frame
        // (pickupobjects.py:86-88); CollectHealth's consumed kit respawns instead, with draws from the env's stream:
        // mw_collect_respawn_kernel, launched behind this one
        const int rs = a.pending_remove[env];
        if (rs >= 0 && !proxy && a.task != MW_TASK_COLLECT) {
            a.ekind[(size_t)rs * a.N + env] = MW_ENT_NONE;
            a.pending_remove[env] = -1;
        }
    }
}
