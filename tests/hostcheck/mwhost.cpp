// TEST INFRASTRUCTURE ONLY — the engine's own per-vertex / per-fragment arithmetic (miniworld_amd/csrc/mw_glmath.h,
// mw_frag.h: the functions the HIP kernels call) compiled for the HOST and wrapped in a plain frame loop, so that
// `-m "not gpu"` tests can compare it with the oracle and with the reference's GL frames without a GPU
// (tests/test_engine_math_cpu.py).  Never part of the product: the engine has no CPU path.
// Input: the oracle's scene struct (oracle/mwo.h), so that tests feed both from the same arrays.
#include "../../miniworld_amd/csrc/mw_frag.h"
#include "../../miniworld_amd/csrc/mw_cover.h"
#include "../../oracle/mwo.h"
#include <vector>
#include <cstdlib>
#include <cstring>

using namespace mwgl;

namespace {
struct Tri { TriSetup ts; int tex; int draw; };
struct Sink {
    std::vector<Tri> *out; int tex, draw;
    void operator()(const TriSetup &t) { out->push_back(Tri{t, tex, draw}); }
};

const int PAT1[1][2] = {{8, 8}};
const int PAT4[4][2] = {{6, 2}, {14, 6}, {2, 10}, {10, 14}};
const int PAT8[8][2] = {{9, 5}, {7, 11}, {13, 9}, {5, 3}, {3, 13}, {1, 7}, {11, 15}, {15, 1}};
const int PAT16[16][2] = {{9, 9}, {7, 5}, {5, 10}, {12, 7}, {3, 6}, {10, 13}, {13, 11}, {11, 3},
                          {6, 14}, {8, 1}, {4, 2}, {2, 12}, {0, 8}, {15, 4}, {14, 15}, {1, 0}};

void build_frame(const mwo_scene *sc, Frame &f)
{
    if (sc->view == 1) {
        double min_x = sc->extent[0] - 1, max_x = sc->extent[1] + 1, min_z = sc->extent[2] - 1, max_z = sc->extent[3] + 1;
        const double width = max_x - min_x, height = max_z - min_z;
        const double aspect = width / height, fb_aspect = (double)sc->width / (double)sc->height;
        if (aspect > fb_aspect) { const double nh = width / fb_aspect, d = nh - height; min_z -= d / 2; max_z += d / 2; }
        else if (aspect < fb_aspect) { const double nw = height * fb_aspect, d = nw - width; min_x -= d / 2; max_x += d / 2; }
        frame_top(f, min_x, max_x, min_z, max_z);
    } else {
        double sh, ch, sp, cp;
        mwo_sincos(sc->agent_dir / 2.0, &sh, &ch);
        const double a = ch, c = -1.0 * sh;
        const double ry00 = a * a - c * c, ry02 = 2.0 * (a * c), ry11 = a * a + c * c;
        const double pitch = sc->cam_pitch * 3.14159265358979323846 / 180.0;
        mwo_sincos(pitch / 2.0, &sp, &cp);
        const double az = cp, dz = -1.0 * sp;
        const double rz00 = az * az - dz * dz, rz01 = 2.0 * (0.0 - az * dz);
        const double eye[3] = {sc->agent_pos[0] + sc->cam_fwd_disp * ry00, sc->agent_pos[1] + sc->cam_height * ry11,
                               sc->agent_pos[2] + sc->cam_fwd_disp * ry02};
        const double dir[3] = {rz00 * ry00, rz01 * ry11, rz00 * ry02};
        const double at[3] = {eye[0] + dir[0], eye[1] + dir[1], eye[2] + dir[2]};
        double sf, cf;
        mwo_sincos(sc->cam_fov_y / 2 * 3.14159265358979323846 / 180, &sf, &cf);
        frame_perspective(f, eye, at, cf / sf, sc->width, sc->height);
    }
    frame_finish(f, sc->width, sc->height, sc->light_pos, sc->light_color, sc->light_ambient);
}

void geometry(const mwo_scene *sc, bool ms, std::vector<Tri> &out)
{
    Frame f;
    build_frame(sc, f);
    Xform cam;
    make_xform(f, f.view, f.view_flags, cam);
    Vert buf0[MWGL_MAX_CLIP_VERTS], buf1[MWGL_MAX_CLIP_VERTS];
    int draw = 0;
    float stale_n[3] = {0, 1, 0};
    for (int i = 0; i < sc->n_polys; ++i, ++draw) {
        const mwo_poly *q = &sc->polys[i];
        const int nv = q->nv & 0xFF;
        Xform ex;
        const Xform *x = &cam;
        if (q->nv & MWO_POLY_XF) { entity_xform(f, q->xf, q->xf[3], 1.0f, false, ex); x = &ex; }
        float col[3];
        light_vertex(f, *x, q->n, q->rgb, col);
        Vert v[4];
        for (int k = 0; k < nv; ++k) {
            transform_vertex(f, *x, q->v[k], v[k]);
            v[k].st[0] = q->tex >= 0 ? q->uv[k][0] : 0.0f; v[k].st[1] = q->tex >= 0 ? q->uv[k][1] : 0.0f;
            memcpy(v[k].col, col, sizeof col);
        }
        Sink sink{&out, q->tex, draw};
        emit_primitive<false>(f, v, nv, (q->nv & MWO_POLY_QUAD) ? SPLIT_QUAD_LIST : SPLIT_POLYGON, buf0, buf1, ms, q->tex >= 0, sink);
        memcpy(stale_n, q->n, sizeof stale_n);
    }
    for (int e = 0; e < sc->n_ents; ++e) {
        const mwo_ent *en = &sc->ents[e];
        if (en->kind == MWO_ENT_BOX) {
            const float pos[3] = {(float)en->pos[0], (float)en->pos[1], (float)en->pos[2]};
            Xform ex;
            entity_xform(f, pos, (float)(en->dir * (180 / 3.14159265358979323846)), 1.0f, false, ex);
            const float lo[3] = {(float)(-en->size[0] / 2), 0.0f, (float)(-en->size[2] / 2)};
            const float hi[3] = {(float)(en->size[0] / 2), (float)en->size[1], (float)(en->size[2] / 2)};
            const float base[3] = {(float)en->color[0], (float)en->color[1], (float)en->color[2]};
            Vert v[6][4];
            bool clipped = false;
            for (int fc = 0; fc < 6; ++fc) {
                float n[3], col[3];
                box_normal(fc, n);
                light_vertex(f, ex, n, base, col);
                for (int k = 0; k < 4; ++k) {
                    const int sel = box_sel(fc, k);
                    const float p[3] = {(sel & 1) ? hi[0] : lo[0], (sel & 2) ? hi[1] : lo[1], (sel & 4) ? hi[2] : lo[2]};
                    transform_vertex(f, ex, p, v[fc][k]);
                    v[fc][k].st[0] = v[fc][k].st[1] = 0.0f;
                    memcpy(v[fc][k].col, col, sizeof col);
                    clipped |= v[fc][k].clipmask != 0;
                }
            }
            const int split = (en->is_static || clipped) ? SPLIT_QUAD_LIST : SPLIT_QUAD_DIRECT;
            for (int fc = 0; fc < 6; ++fc, ++draw) {
                Sink sink{&out, -1, draw};
                emit_primitive<false>(f, v[fc], 4, split, buf0, buf1, ms, false, sink);
            }
            stale_n[0] = 0; stale_n[1] = -1; stale_n[2] = 0;
        } else if (en->kind == MWO_ENT_MESH) {
            const mwo_mesh *m = &sc->meshes[en->mesh];
            const float pos[3] = {(float)en->pos[0], (float)en->pos[1], (float)en->pos[2]};
            Xform ex;
            entity_xform(f, pos, (float)(en->dir * 180 / 3.14159265358979323846), (float)en->scale, true, ex);
            for (int t = 0; t < m->ntris; ++t, ++draw) {
                Vert v[4];
                for (int k = 0; k < 3; ++k) {
                    transform_vertex(f, ex, &m->pos[(size_t)(t * 3 + k) * 3], v[k]);
                    light_vertex(f, ex, &m->nrm[(size_t)(t * 3 + k) * 3], &m->rgb[(size_t)(t * 3 + k) * 3], v[k].col);
                    v[k].st[0] = m->tex >= 0 ? m->uv[(size_t)(t * 3 + k) * 2] : 0.0f;
                    v[k].st[1] = m->tex >= 0 ? m->uv[(size_t)(t * 3 + k) * 2 + 1] : 0.0f;
                }
                Sink sink{&out, m->tex, draw};
                emit_primitive<true>(f, v, 3, SPLIT_TRIANGLE, buf0, buf1, ms, m->tex >= 0, sink);
            }
        }
    }
    if (sc->render_agent) {
        double sd, cd;
        mwo_sincos(sc->agent_dir, &sd, &cd);
        const double rad = sc->agent_radius, hgt = sc->agent_height;
        const double p[3] = {sc->agent_pos[0] + 0 * hgt, sc->agent_pos[1] + 1 * hgt, sc->agent_pos[2] + 0 * hgt};
        const double dv[3] = {cd * rad, 0 * rad, -sd * rad}, rv[3] = {sd * rad, 0 * rad, cd * rad};
        float pv[3][3];
        for (int i = 0; i < 3; ++i) {
            pv[0][i] = (float)(p[i] + dv[i]);
            pv[2][i] = (float)(p[i] + 0.75 * (rv[i] - dv[i]));
            pv[1][i] = (float)(p[i] + 0.75 * (-rv[i] - dv[i]));
        }
        const float red[3] = {1, 0, 0};
        float col[3];
        light_vertex(f, cam, stale_n, red, col);
        Vert v[4];
        for (int k = 0; k < 3; ++k) { transform_vertex(f, cam, pv[k], v[k]); v[k].st[0] = v[k].st[1] = 0; memcpy(v[k].col, col, sizeof col); }
        Sink sink{&out, -1, draw};
        emit_primitive<false>(f, v, 3, SPLIT_TRIANGLE, buf0, buf1, ms, false, sink);
    }
}

const uint8_t *tex_level(const mwo_tex *t, int level, int *lw, int *lh)
{
    const uint8_t *p = t->rgb;
    int w = t->w, h = t->h;
    for (int l = 0; l < level; ++l) { p += (int64_t)w * h * 3; w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1; }
    *lw = w; *lh = h;
    return p;
}

void fetch_level(const mwo_tex *t, int level, float s, float tt, int out[3])
{
    int w, h, i0, j0, wx, wy;
    const uint8_t *px = tex_level(t, level, &w, &h);
    linear_coord(s, w, (w & (w - 1)) == 0, i0, wx);
    linear_coord(tt, h, (h & (h - 1)) == 0, j0, wy);
    const int i1 = i0 + 1 == w ? 0 : i0 + 1, j1 = j0 + 1 == h ? 0 : j0 + 1;
    auto tex = [&](int i, int j) { const uint8_t *q = px + ((int64_t)j * w + i) * 3; return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16); };
    bilerp_rgb(tex(i0, j0), tex(i1, j0), tex(i0, j1), tex(i1, j1), wx, wy, out);
}
}  // namespace

extern "C" int mwhost_render(const mwo_scene *sc, uint8_t *rgb, uint16_t *z16out)
{
    const int W = sc->width, H = sc->height, S = sc->nsamples;
    const int (*pat)[2] = S == 1 ? PAT1 : (S == 4 ? PAT4 : (S == 8 ? PAT8 : PAT16));
    const bool ms = S > 1;
    std::vector<Tri> tris;
    geometry(sc, ms, tris);
    std::vector<uint16_t> zb((size_t)W * H * S, 65535);
    std::vector<float> cb((size_t)W * H * S * 3);
    for (size_t i = 0; i < (size_t)W * H * S; ++i) for (int c = 0; c < 3; ++c) cb[i * 3 + c] = (float)sc->sky[c];
    const float eo = ms ? 0.5f : 0.0f;
    for (const Tri &tr : tris) {
        const TriSetup &p = tr.ts;
        int x0 = (p.minx >> 8) - 1, x1 = (p.maxx >> 8) + 1, y0 = (p.miny >> 8) - 1, y1 = (p.maxy >> 8) + 1;
        if (x0 < 0) x0 = 0; if (y0 < 0) y0 = 0; if (x1 > W - 1) x1 = W - 1; if (y1 > H - 1) y1 = H - 1;
        for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) {
                bool shaded = false;
                float col[3];
                for (int s = 0; s < S; ++s) {
                    const int32_t fx = px * 256 + (ms ? pat[s][0] * 16 : 0), fy = py * 256 + (ms ? pat[s][1] * 16 : 0);
                    bool in = true;
                    for (int k = 0; k < 3; ++k) in &= (p.c[k] + (int64_t)p.dcdy[k] * fy - (int64_t)p.dcdx[k] * fx) > 0;
                    if (!in) continue;
                    const float xs = (float)px + (ms ? (float)pat[s][0] * 0.0625f : 0.0f), ys = (float)py + (ms ? (float)pat[s][1] * 0.0625f : 0.0f);
                    const uint16_t z = (uint16_t)z_to_unorm16(plane_at(p.z, xs, ys));
                    const size_t idx = ((size_t)py * W + px) * S + s;
                    if (!(z < zb[idx])) continue;
                    if (!shaded) {
                        shaded = true;
                        const float x = (float)px + eo, y = (float)py + eo;
                        float ss, tt, oow;
                        tex_coords(p.w, p.s, p.t, x, y, ss, tt, oow);
                        for (int k = 0; k < 3; ++k) col[k] = plane_at(p.col[k], x, y) * oow;
                        if (tr.tex >= 0) {
                            const mwo_tex *tx = &sc->tex[tr.tex];
                            const float qx = (float)(px & ~1) + eo, qy = (float)(py & ~1) + eo;
                            float s00, t00, s10, t10, s01, t01, d;
                            tex_coords(p.w, p.s, p.t, qx, qy, s00, t00, d);
                            tex_coords(p.w, p.s, p.t, qx + 1.0f, qy, s10, t10, d);
                            tex_coords(p.w, p.s, p.t, qx, qy + 1.0f, s01, t01, d);
                            int l0, w8, c0[3], c1[3];
                            lod_select(s00, t00, s10, t10, s01, t01, (float)tx->w, (float)tx->h, tx->nlevels, l0, w8);
                            fetch_level(tx, l0, ss, tt, c0);
                            if (w8 > 0) {
                                fetch_level(tx, l0 + 1 > tx->nlevels - 1 ? tx->nlevels - 1 : l0 + 1, ss, tt, c1);
                                for (int k = 0; k < 3; ++k) c0[k] = lerp8(c0[k], c1[k], w8);
                            }
                            for (int k = 0; k < 3; ++k) col[k] = ((float)c0[k] * (1.0f / 255.0f)) * col[k];
                        }
                    }
                    zb[idx] = z;
                    memcpy(&cb[idx * 3], col, sizeof col);
                }
            }
    }
    const float inv = 1.0f / (float)S;
    for (int gy = 0; gy < H; ++gy)
        for (int px = 0; px < W; ++px) {
            const size_t base = ((size_t)gy * W + px) * S, o = (size_t)(H - 1 - gy) * W + px;
            for (int c = 0; c < 3; ++c) {
                float acc = cb[base * 3 + c];
                for (int s = 1; s < S; ++s) acc = acc + cb[(base + s) * 3 + c];
                rgb[o * 3 + c] = (uint8_t)float_to_unorm8(acc * inv);
            }
            if (z16out) z16out[o] = zb[base];
        }
    return 0;
}

// how many triangles the display list of the scene holds after clipping, culling and setup (the engine's nvis)
extern "C" int mwhost_list_length(const mwo_scene *sc)
{
    std::vector<Tri> tris;
    geometry(sc, sc->nsamples > 1, tris);
    return (int)tris.size();
}

// glibc's sinf / cosf against the restatement the device uses (exhaustive range test in tests/)
extern "C" void mwhost_sincosf(float x, float *s, float *c) { sincosf_glibc(x, *s, *c); }

// mw_frag.h: the lod from rho^2's bits (the quad kernel's form) against llvmpipe's float arithmetic, on n bit patterns;
// returns the number of inputs where level or weight differ for some pyramid of 1 .. 12 levels
extern "C" long mwhost_lod_bits_mismatches(const uint32_t *bits, long n)
{
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        const float x = mwgl::u2f(bits[i]);
        bool b = false;
        for (int nl = 1; nl <= 12; ++nl) {
            int l0, w8, l0b, w8b;
            mwgl::lod_from_rho2(x, nl, l0, w8);
            mwgl::lod_from_rho2_bits(x, nl, l0b, w8b);
            b |= l0 != l0b || w8 != w8b;
        }
        bad += b ? 1 : 0;
    }
    return bad;
}

// The geometry kernel clips flat-shaded triangles on compact work-list vertices (mwgl::ClipVert: no colour, no clip mask);
// this hook runs both instantiations of the clipper on the same clip-space triangle and reports whether vertex count,
// clip coordinates, window coordinates and texture coordinates agree bit for bit.
extern "C" int mwhost_clip_variants_agree(const float clip[3][4], const float st[3][2], int W, int H)
{
    Frame f{};
    f.vp_scale[0] = (float)W * 0.5f; f.vp_trans[0] = (float)W * 0.5f;
    f.vp_scale[1] = (float)H * 0.5f; f.vp_trans[1] = (float)H * 0.5f;
    f.vp_scale[2] = 0.5f; f.vp_trans[2] = 0.5f;
    Vert v[3];
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < 4; ++i) v[k].clip[i] = clip[k][i];
        const float w = v[k].clip[3];
        uint32_t m = 0;
        if (v[k].clip[0] > w) m |= 1u;
        if (v[k].clip[0] + w < 0.0f) m |= 2u;
        if (v[k].clip[1] > w) m |= 4u;
        if (v[k].clip[1] + w < 0.0f) m |= 8u;
        if (v[k].clip[2] + w < 0.0f) m |= 16u;
        if (v[k].clip[2] > w) m |= 32u;
        v[k].clipmask = m;
        const float oow = 1.0f / w;
        for (int i = 0; i < 3; ++i) v[k].win[i] = fmaf(v[k].clip[i] * oow, f.vp_scale[i], f.vp_trans[i]);
        v[k].win[3] = oow;
        v[k].st[0] = st[k][0]; v[k].st[1] = st[k][1];
        v[k].col[0] = 0.25f; v[k].col[1] = 0.5f; v[k].col[2] = 0.75f;
    }
    Vert a0[MWGL_MAX_CLIP_VERTS], a1[MWGL_MAX_CLIP_VERTS], *ra;
    ClipVert b0[MWGL_MAX_CLIP_VERTS], b1[MWGL_MAX_CLIP_VERTS], *rb;
    const int na = clip_triangle<false>(f, v[0], v[1], v[2], a0, a1, &ra);
    const int nb = clip_triangle<false>(f, v[0], v[1], v[2], b0, b1, &rb);
    if (na != nb) return 0;
    for (int i = 0; i < na; ++i) {
        if (memcmp(ra[i].clip, rb[i].clip, 16) || memcmp(ra[i].win, rb[i].win, 16) || memcmp(ra[i].st, rb[i].st, 8)) return 0;
        if (ra[i].col[0] != 0.25f || ra[i].col[1] != 0.5f || ra[i].col[2] != 0.75f) return 0;
    }
    return 1 + na;
}

// The big scenes' sift drops a polygon whose triangles are "clearly back-facing" (mw_glmath.h: the float area beyond what
// snapping the vertices to 1/256 px can change).  Returns 1 if the claim and the exact test disagree: clearly_back says
// yes although the setup (snapped integers) keeps the triangle.
extern "C" int mwhost_clearly_back_is_wrong(const float win[3][4], int multisampled)
{
    TriEdges te;
    const bool kept = setup_triangle_pos(win[0], win[1], win[2], multisampled != 0, te);
    return clearly_back(win[0], win[1], win[2]) && kept ? 1 : 0;
}

extern "C" int mwhost_clearly_back(const float win[3][4]) { return clearly_back(win[0], win[1], win[2]) ? 1 : 0; }

// The geometry kernel's clipper works eight lanes to a triangle: lane e owns the polygon's edge e -> e + 1, passes vertex e
// on if it is inside the plane and makes the crossing's vertex; everybody's place in the output list is the number of
// vertices the lanes before it put out.  This is that algorithm with the lanes as a loop, on the same per-vertex functions:
// it must give clip_triangle's polygon bit for bit (vertex count, order, coordinates).  Returns 1 + n on agreement, 0 else.
extern "C" int mwhost_edge_parallel_clip_agrees(const float clip[3][4], const float st[3][2], int W, int H)
{
    Frame f{};
    f.vp_scale[0] = (float)W * 0.5f; f.vp_trans[0] = (float)W * 0.5f;
    f.vp_scale[1] = (float)H * 0.5f; f.vp_trans[1] = (float)H * 0.5f;
    f.vp_scale[2] = 0.5f; f.vp_trans[2] = 0.5f;
    Vert v[3];
    uint32_t un = 0u, in = 0x3Fu;
    for (int k = 0; k < 3; ++k) {
        for (int i = 0; i < 4; ++i) v[k].clip[i] = clip[k][i];
        const float w = v[k].clip[3];
        uint32_t m = 0;
        if (v[k].clip[0] > w) m |= 1u;
        if (v[k].clip[0] + w < 0.0f) m |= 2u;
        if (v[k].clip[1] > w) m |= 4u;
        if (v[k].clip[1] + w < 0.0f) m |= 8u;
        if (v[k].clip[2] + w < 0.0f) m |= 16u;
        if (v[k].clip[2] > w) m |= 32u;
        v[k].clipmask = m; un |= m; in &= m;
        const float oow = 1.0f / w;
        for (int i = 0; i < 3; ++i) v[k].win[i] = fmaf(v[k].clip[i] * oow, f.vp_scale[i], f.vp_trans[i]);
        v[k].win[3] = oow;
        v[k].st[0] = st[k][0]; v[k].st[1] = st[k][1];
        v[k].col[0] = v[k].col[1] = v[k].col[2] = 0.5f;
    }
    ClipVert a0[MWGL_MAX_CLIP_VERTS], a1[MWGL_MAX_CLIP_VERTS], *ra;
    const int na = clip_triangle<false>(f, v[0], v[1], v[2], a0, a1, &ra);
    if (in) return na == 0 ? 1 : 0;
    // the lanes' version
    ClipVert l[2][MWGL_MAX_CLIP_VERTS + 2];
    for (int k = 0; k < 3; ++k) clip_copy_in(l[0][k], v[k]);
    int n = 3, cur = 0;
    uint32_t cm = un;
    while (cm != 0u && n >= 3) {
        int plane = 0;
        while (!((cm >> plane) & 1u)) ++plane;
        cm &= cm - 1u;
        const ClipVert *inl = l[cur];
        ClipVert *out = l[cur ^ 1];
        float dpv[MWGL_MAX_CLIP_VERTS + 2];
        bool bad = false;
        for (int e = 0; e < n; ++e) { dpv[e] = clip_dist(inl[e], plane); bad |= !(dpv[e] == dpv[e]) || dpv[e] - dpv[e] != 0.0f; }
        int total = 0;
        for (int e = 0; e < n; ++e) {       // "lane" e: its place = what the lanes before it put out
            const int nxt = e + 1 < n ? e + 1 : 0;
            const float dp_prev = dpv[e], dp = dpv[nxt];
            const bool emit = dp_prev >= 0.0f, cross = (dp >= 0.0f) != (dp_prev >= 0.0f);
            int pos = 0;
            for (int b = 0; b < e; ++b) {
                const int bn = b + 1 < n ? b + 1 : 0;
                pos += (dpv[b] >= 0.0f ? 1 : 0) + (((dpv[bn] >= 0.0f) != (dpv[b] >= 0.0f)) ? 1 : 0);
            }
            if (emit) out[pos] = inl[e];
            if (cross) {
                const bool from_cur = fabsf(dp) < fabsf(dp_prev);
                const float t = (from_cur ? dp : dp_prev) / (from_cur ? dp - dp_prev : dp_prev - dp);
                clip_interp<false>(f, out[pos + (emit ? 1 : 0)], t, from_cur ? inl[nxt] : inl[e], from_cur ? inl[e] : inl[nxt]);
            }
            total += (emit ? 1 : 0) + (cross ? 1 : 0);
        }
        n = bad ? 0 : total;
        cur ^= 1;
    }
    if (n < 3) n = 0;
    if (n != na) return 0;
    for (int i = 0; i < n; ++i)
        if (memcmp(l[cur][i].clip, ra[i].clip, 16) || memcmp(l[cur][i].win, ra[i].win, 16) || memcmp(l[cur][i].st, ra[i].st, 8)) return 0;
    return 1 + n;
}

// Big scenes test a bounding box of eight polygons against the frustum before the polygons themselves (mw_geom.hip,
// box_view): for every plane the box is found outside of, every point inside the box must be outside in
// transform_vertex's own arithmetic.  Camera as render_obs builds it (eye, direction of view in the xz plane + pitch).
// Returns the number of points that contradict the box; *planes gets the box's plane mask.
extern "C" int mwhost_box_cull_contradictions(const double eye[3], const double at[3], double fov_y_deg, int W, int H, const float mn[3],
                                              const float mx[3], const float *pts, int n, int *planes)
{
    Frame f;
    double sf, cf;
    mwo_sincos(fov_y_deg / 2 * 3.14159265358979323846 / 180, &sf, &cf);
    frame_perspective(f, eye, at, cf / sf, W, H);
    const double lp[3] = {0, 2.5, 0}, lc[3] = {0.7, 0.7, 0.7}, la[3] = {0.45, 0.45, 0.45};
    frame_finish(f, W, H, lp, lc, la);
    Xform cam;
    make_xform(f, f.view, f.view_flags, cam);
    const BoxView b = box_view(mn, mx, cam.mvp.m, f.vp_scale[0], f.vp_trans[0]);
    *planes = (int)b.all;
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        Vert v;
        transform_vertex(f, cam, pts + 3 * i, v);
        if ((v.clipmask & b.all) != b.all) ++bad;
        if (b.front && (v.win[0] < b.xmn - 0.05f || v.win[0] > b.xmx + 0.05f || v.clip[3] < b.zq * 0.9999f)) ++bad;
    }
    return bad;
}


// mw_cover.h: the mesh entity kernel's coverage of a small triangle by sample columns (setup_edges + cover_columns, 32-bit
// integers) against the definition — mw_glmath.h's setup_triangle_pos (64-bit edge constants) tested at every sample of every
// pixel of the frame, and the packed depth keys at the covered samples.  win: [n][3][4] window coordinates; returns the
// number of triangles whose sample sets, depth keys or culling differ; *covered = samples covered in total.
extern "C" long mwhost_cover_mismatches(const float *win, long n, int W, int H, long *covered)
{
    long bad = 0, cov = 0;
    std::vector<uint32_t> want((size_t)W * H * 8), got((size_t)W * H * 8);
    for (long t = 0; t < n; ++t) {
        const float *wa = win + t * 12, *wb = wa + 4, *wc = wa + 8;
        TriEdges te;
        const bool kept = setup_triangle_pos(wa, wb, wc, true, te);
        mwcov::Edges e;
        const bool kept2 = mwcov::setup_edges(wa, wb, wc, e);
        if (kept != kept2) { ++bad; continue; }
        if (!kept) continue;
        bool same = e.minx == te.minx && e.maxx == te.maxx && e.miny == te.miny && e.maxy == te.maxy &&
                    e.z.a0 == te.z.a0 && e.z.dadx == te.z.dadx && e.z.dady == te.z.dady;
        for (int k = 0; k < 3; ++k) same &= e.dcdx[k] == te.dcdx[k] && e.dcdy[k] == te.dcdy[k] && (int64_t)e.c[k] == te.c[k];
        std::fill(want.begin(), want.end(), 0xFFFFFFFFu);
        std::fill(got.begin(), got.end(), 0xFFFFFFFFu);
        for (int gy = 0; gy < H; ++gy)
            for (int px = 0; px < W; ++px)
                for (int s = 0; s < 8; ++s) {
                    const int64_t fx = (int64_t)px * 256 + PAT8[s][0] * 16, fy = (int64_t)gy * 256 + PAT8[s][1] * 16;
                    bool in = true;
                    for (int k = 0; k < 3; ++k) in &= (te.c[k] + (int64_t)te.dcdy[k] * fy - (int64_t)te.dcdx[k] * fx) > 0;
                    if (!in) continue;
                    const float xs = (float)px + (float)PAT8[s][0] * 0.0625f, ys = (float)gy + (float)PAT8[s][1] * 0.0625f;
                    want[((size_t)gy * W + px) * 8 + s] = z_to_unorm16(plane_at(te.z, xs, ys));
                    ++cov;
                }
        int dup = 0;
        auto sink = [&](int px, int gy, int s, float xs, float ys) {
            uint32_t &g = got[((size_t)gy * W + px) * 8 + s];
            if (g != 0xFFFFFFFFu) ++dup;
            g = z_to_unorm16(plane_at(e.z, xs, ys));
        };
        // both forms on every triangle, whichever mwcov::cover would choose
        mwcov::cover_columns(e, W, H, sink);
        if (!same || dup || want != got) { ++bad; continue; }
        std::fill(got.begin(), got.end(), 0xFFFFFFFFu);
        mwcov::cover_pixels(e, W, H, sink);
        if (dup || want != got) { ++bad; continue; }
        std::fill(got.begin(), got.end(), 0xFFFFFFFFu);
        const bool any = mwcov::cover(e, W, H, sink);
        if (dup || want != got) { ++bad; continue; }
        // ... and the question alone (edges and bounds only, no depth plane)
        mwcov::Edges exy;
        if (!mwcov::setup_edges_xy(wa, wb, wc, exy) || mwcov::covers_any(exy, W, H) != any) ++bad;
    }
    if (covered) *covered = cov;
    return bad;
}
