/* ORACLE — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED for pixels (see mwo.h).
 *
 * CPU restatement of the reference's per-step GL frame:
 *   MiniWorldEnv.render_obs        miniworld/miniworld.py:1177-1221
 *   MiniWorldEnv._render_static    miniworld/miniworld.py:1019-1062   (light, material state)
 *   Room._render                   miniworld/miniworld.py:401-434
 *   MiniWorldEnv._render_world     miniworld/miniworld.py:1064-1086   (draw order)
 *   Box.render / drawBox           miniworld/entity.py:409-432, miniworld/opengl.py:460-503
 *   MeshEnt.render / ObjMesh.render miniworld/entity.py:150-161, miniworld/objmesh.py:280-292
 *   Texture.load                   miniworld/opengl.py:148-184       (RGB8, trilinear, REPEAT)
 *   FrameBuffer (8x MSAA RGBA32F + DEPTH16, resolve, readback, flip) opengl.py:202-398
 *   FrameBuffer.get_depth_map      miniworld/opengl.py:400-435
 *   Agent.cam_pos / cam_dir        miniworld/entity.py:476-503, miniworld/math.py:11-27
 *
 * The arithmetic the reference delegates to the GL driver is restated from the
 * OpenGL 2.1 specification; every implementation-defined choice is pinned in
 * DESIGN.md section 3 (rules R1..R14, quoted in the comments below).
 *
 * Straightforward "immediate mode" structure on purpose: primitives are drawn in GL
 * order into explicit per-sample colour / depth / id buffers with a GL_LESS test,
 * then resolved.  (The HIP engine is organised completely differently — tile
 * hierarchical, min-reduction of packed keys, deferred shading — and must still match.)
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include "mwo.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXS 16

/* R5: sample positions in 1/16 pixel, image space (x right, y down), origin = the
 * pixel's upper-left corner: the D3D / Vulkan standard patterns. */
static const int PAT1[1][2]  = {{8, 8}};
static const int PAT4[4][2]  = {{6, 2}, {14, 6}, {2, 10}, {10, 14}};
static const int PAT8[8][2]  = {{9, 5}, {7, 11}, {13, 9}, {5, 3}, {3, 13}, {1, 7}, {11, 15}, {15, 1}};
static const int PAT16[16][2] = {{9, 9}, {7, 5}, {5, 10}, {12, 7}, {3, 6}, {10, 13}, {13, 11}, {11, 3},
                                 {6, 14}, {8, 1}, {4, 2}, {2, 12}, {0, 8}, {15, 4}, {14, 15}, {1, 0}};

typedef struct { float hx, hy, hw, cz; } hvert;

typedef struct {
    float m[3][4];          /* modelview rows (f32)                */
    float p00, p11, p22, p23;
    float p03, p13;         /* orthographic only (R3o)             */
    int ortho;
    float halfw, halfh;
    float L[3];             /* unit light direction, world space   */
    float amb[3];           /* 0.2 + light_ambient                 */
    float lcol[3];
    float sky[3];
} camera;

typedef struct {
    int nv;
    float ea[4], eb[4], ec[4];
    int tl[4];
    float zx, zy, zc;
    float Ua, Ub, Uc, Va, Vb, Vc, Wa, Wb, Wc;
    int gouraud;
    float col[3];           /* flat colour                         */
    float Ca[3], Cb[3], Cc[3];
    float col0[3];          /* vertex 0 colour (fallback, R9)      */
    int tex;
    int x0, x1, y0, y1;     /* conservative pixel bbox, inclusive  */
} prim;

/* ------------------------------------------------------------------ camera (R1, R2) */

static void build_camera(const mwo_scene *sc, camera *cam)
{
    /* Agent.cam_pos / cam_dir via gen_rot_matrix (math.py:11-27, entity.py:476-503),
     * evaluated in double exactly as numpy does (terms multiplied by exact zeros dropped). */
    double sh, ch, sp, cp;
    mwo_sincos(sc->agent_dir / 2.0, &sh, &ch);
    double a = ch, c = -1.0 * sh;              /* b = d = 0 for the Y axis          */
    double ry00 = a * a - c * c;               /* = cos(dir)                        */
    double ry02 = 2.0 * (a * c);               /* = -sin(dir)                       */
    double ry11 = a * a + c * c;               /* ~ 1                               */
    double pitch = sc->cam_pitch * 3.14159265358979323846 / 180.0;
    mwo_sincos(pitch / 2.0, &sp, &cp);
    double az = cp, dz = -1.0 * sp;            /* b = c = 0 for the Z axis          */
    double rz00 = az * az - dz * dz;           /* cos(pitch)                        */
    double rz01 = 2.0 * (0.0 - az * dz);       /* sin(pitch)                        */
    cam->ortho = 0; cam->p03 = 0.0f; cam->p13 = 0.0f;
    cam->halfw = (float)sc->width * 0.5f;
    cam->halfh = (float)sc->height * 0.5f;
    if (sc->view == 1) {
        /* R1o / R2o: render_top_view (miniworld.py:1108-1160).  Extents +-1 m, widened to the frame
         * buffer's aspect; glOrtho(min_x, max_x, -max_z, -min_z, -100, 100); modelview maps
         * (x, y, z) -> (x, -z, y).  All in double, then float32. */
        double min_x = sc->extent[0] - 1, max_x = sc->extent[1] + 1, min_z = sc->extent[2] - 1, max_z = sc->extent[3] + 1;
        double width = max_x - min_x, height = max_z - min_z;
        double aspect = width / height, fb_aspect = (double)sc->width / (double)sc->height;
        if (aspect > fb_aspect) {
            double new_h = width / fb_aspect, h_diff = new_h - height;
            min_z -= h_diff / 2; max_z += h_diff / 2;
        } else if (aspect < fb_aspect) {
            double new_w = height * fb_aspect, w_diff = new_w - width;
            min_x -= w_diff / 2; max_x += w_diff / 2;
        }
        double l = min_x, r = max_x, b = -max_z, t = -min_z, n = -100.0, f = 100.0;
        cam->ortho = 1;
        cam->p00 = (float)(2.0 / (r - l)); cam->p03 = (float)(-(r + l) / (r - l));
        cam->p11 = (float)(2.0 / (t - b)); cam->p13 = (float)(-(t + b) / (t - b));
        cam->p22 = (float)(-2.0 / (f - n)); cam->p23 = (float)(-(f + n) / (f - n));
        const float M[3][4] = {{1, 0, 0, 0}, {0, 0, -1, 0}, {0, 1, 0, 0}};
        memcpy(cam->m, M, sizeof M);
    } else {
    double eye[3], dir[3];
    eye[0] = sc->agent_pos[0] + sc->cam_fwd_disp * ry00;
    eye[1] = sc->agent_pos[1] + sc->cam_height * ry11;
    eye[2] = sc->agent_pos[2] + sc->cam_fwd_disp * ry02;
    dir[0] = rz00 * ry00;
    dir[1] = rz01 * ry11;
    dir[2] = rz00 * ry02;

    /* R1: gluLookAt(eye, eye+dir, (0,1,0)) (miniworld.py:1210-1219) in double, then f32 */
    double at[3] = {eye[0] + dir[0], eye[1] + dir[1], eye[2] + dir[2]};
    double F[3] = {at[0] - eye[0], at[1] - eye[1], at[2] - eye[2]};
    double fl = sqrt(F[0] * F[0] + F[1] * F[1] + F[2] * F[2]);
    F[0] /= fl; F[1] /= fl; F[2] /= fl;
    /* s = F x up, up = (0,1,0) */
    double s[3] = {-F[2], 0.0, F[0]};
    double sl = sqrt(s[0] * s[0] + s[2] * s[2]);
    s[0] /= sl; s[2] /= sl;
    /* u = s x F */
    double u[3] = {s[1] * F[2] - s[2] * F[1], s[2] * F[0] - s[0] * F[2], s[0] * F[1] - s[1] * F[0]};
    double R[3][3] = {{s[0], s[1], s[2]}, {u[0], u[1], u[2]}, {-F[0], -F[1], -F[2]}};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) cam->m[i][j] = (float)R[i][j];
        cam->m[i][3] = (float)(-(R[i][0] * eye[0] + R[i][1] * eye[1] + R[i][2] * eye[2]));
    }
    /* R2: gluPerspective(fovy, W/H, 0.04, 100) (miniworld.py:1200-1205) in double, then f32 */
    double half = sc->cam_fov_y / 2.0 * 3.14159265358979323846 / 180.0;
    double sf, cf;
    mwo_sincos(half, &sf, &cf);
    double cot = cf / sf;
    double aspect = (double)sc->width / (double)sc->height;
    double zn = 0.04, zf = 100.0;
    cam->p00 = (float)(cot / aspect);
    cam->p11 = (float)cot;
    cam->p22 = (float)(-(zf + zn) / (zf - zn));
    cam->p23 = (float)(-2.0 * zn * zf / (zf - zn));
    }

    /* R10: light.  (GLfloat*4)(*light_pos + [1]) (miniworld.py:1031): ndarray + [1] adds
     * 1 to every component and leaves w = 0 => directional light along light_pos + 1. */
    float lp[3];
    for (int i = 0; i < 3; ++i) lp[i] = (float)(sc->light_pos[i] + 1.0);
    float ll = sqrtf(fmaf(lp[2], lp[2], fmaf(lp[1], lp[1], lp[0] * lp[0])));
    for (int i = 0; i < 3; ++i) {
        cam->L[i] = lp[i] / ll;
        cam->amb[i] = 0.2f + (float)sc->light_ambient[i];   /* scene ambient 0.2 (GL default) */
        cam->lcol[i] = (float)sc->light_color[i];
        cam->sky[i] = (float)sc->sky[i];
    }
}

/* R3: vertex -> homogeneous pixel coordinates */
static hvert xform(const camera *cam, float x, float y, float z)
{
    float ex = fmaf(cam->m[0][0], x, fmaf(cam->m[0][1], y, fmaf(cam->m[0][2], z, cam->m[0][3])));
    float ey = fmaf(cam->m[1][0], x, fmaf(cam->m[1][1], y, fmaf(cam->m[1][2], z, cam->m[1][3])));
    float ez = fmaf(cam->m[2][0], x, fmaf(cam->m[2][1], y, fmaf(cam->m[2][2], z, cam->m[2][3])));
    float cx = cam->p00 * ex, cy = cam->p11 * ey;
    float cw = -ez;
    if (cam->ortho) {           /* R3o: glOrtho has translation terms and w = 1 */
        cx = fmaf(cam->p00, ex, cam->p03);
        cy = fmaf(cam->p11, ey, cam->p13);
        cw = 1.0f;
    }
    hvert h;
    h.cz = fmaf(cam->p22, ez, cam->p23);
    h.hx = (cx + cw) * cam->halfw;
    h.hy = (cw - cy) * cam->halfh;
    h.hw = cw;
    return h;
}

/* R10: per-vertex fixed-function lighting, diffuse only, no renormalisation */
static void light(const camera *cam, const float n[3], const float base[3], float out[3])
{
    float ndl = fmaf(n[2], cam->L[2], fmaf(n[1], cam->L[1], n[0] * cam->L[0]));
    float d = ndl > 0.0f ? ndl : 0.0f;
    for (int i = 0; i < 3; ++i) {
        float k = fmaf(cam->lcol[i], d, cam->amb[i]);
        float v = base[i] * k;
        out[i] = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
}

/* cross product b x a of homogeneous points (edge function of the edge a->b, R4) */
static void edge_coef(const hvert *a, const hvert *b, float *ea, float *eb, float *ec)
{
    *ea = b->hy * a->hw - b->hw * a->hy;
    *eb = b->hw * a->hx - b->hx * a->hw;
    *ec = b->hx * a->hy - b->hy * a->hx;
}

static int clampi(float f, int lo, int hi)
{
    if (!(f > (float)lo)) return lo;
    if (!(f < (float)hi)) return hi;
    return (int)f;
}

/* R4: polygon setup.  Returns 0 when back-facing / degenerate. */
static int setup_prim(const mwo_scene *sc, const hvert *h, int nv, const float (*uv)[2],
                      const float (*vcol)[3], int gouraud, int tex, prim *p)
{
    float ga[3], gb[3], gc[3];
    /* interpolation basis from vertices 0,1,2: G0 = edge(1->2), G1 = edge(2->0), G2 = edge(0->1) */
    edge_coef(&h[1], &h[2], &ga[0], &gb[0], &gc[0]);
    edge_coef(&h[2], &h[0], &ga[1], &gb[1], &gc[1]);
    edge_coef(&h[0], &h[1], &ga[2], &gb[2], &gc[2]);
    float D = fmaf(h[0].hx, ga[0], fmaf(h[0].hy, gb[0], h[0].hw * gc[0]));
    if (!(D > 0.0f)) return 0;             /* back-face cull (miniworld.py:512), CCW front */
    p->nv = nv;
    for (int k = 0; k < nv; ++k) {
        const hvert *a = &h[k], *b = &h[(k + 1) % nv];
        edge_coef(a, b, &p->ea[k], &p->eb[k], &p->ec[k]);
        p->tl[k] = (p->ea[k] > 0.0f) || (p->ea[k] == 0.0f && p->eb[k] > 0.0f);
    }
    float invD = 1.0f / D;
    float ta = fmaf(h[2].cz, ga[2], fmaf(h[1].cz, ga[1], h[0].cz * ga[0]));
    float tb = fmaf(h[2].cz, gb[2], fmaf(h[1].cz, gb[1], h[0].cz * gb[0]));
    float tc = fmaf(h[2].cz, gc[2], fmaf(h[1].cz, gc[1], h[0].cz * gc[0]));
    p->zx = (ta * invD) * 0.5f;
    p->zy = (tb * invD) * 0.5f;
    p->zc = fmaf(tc * invD, 0.5f, 0.5f);
    if (!gouraud && h[0].hw > 0.0f && h[1].hw > 0.0f && h[2].hw > 0.0f) {
        /* R6p: a polygon whose first three vertices lie in front of the eye takes the plane through their window
         * coordinates (X, Y, z_w), solved in binary64 and rounded to binary32.  Same plane, different arithmetic: the
         * sums above cancel catastrophically for a polygon seen edge-on (a far floor two pixels high, a wall stub a
         * tenth of a pixel wide) — depths dozens of D16 steps outside the range of the polygon's own vertices, which a
         * rasteriser working on snapped window coordinates never produces. */
        /* differences of the window coordinates over common denominators: X1 - X0 = (hx1 w0 - hx0 w1) / (w0 w1), the
         * products exact in binary64; the denominators cancel between the plane's numerators and its determinant */
        double w0 = h[0].hw, w1 = h[1].hw, w2 = h[2].hw;
        double nax = (double)h[1].hx * w0 - (double)h[0].hx * w1, nay = (double)h[1].hy * w0 - (double)h[0].hy * w1;
        double naz = (double)h[1].cz * w0 - (double)h[0].cz * w1;
        double nbx = (double)h[2].hx * w0 - (double)h[0].hx * w2, nby = (double)h[2].hy * w0 - (double)h[0].hy * w2;
        double nbz = (double)h[2].cz * w0 - (double)h[0].cz * w2;
        double det = nax * nby - nbx * nay;
        if (det != 0.0) {
            double r = 1.0 / det, i0 = 1.0 / w0;
            double zx = 0.5 * ((naz * nby - nbz * nay) * r), zy = 0.5 * ((nax * nbz - nbx * naz) * r);
            p->zx = (float)zx;
            p->zy = (float)zy;
            p->zc = (float)(0.5 + ((0.5 * (double)h[0].cz - zx * (double)h[0].hx) - zy * (double)h[0].hy) * i0);
        }
    }
    p->Wa = (ga[0] + ga[1]) + ga[2];
    p->Wb = (gb[0] + gb[1]) + gb[2];
    p->Wc = (gc[0] + gc[1]) + gc[2];
    p->tex = tex;
    if (tex >= 0) {
        p->Ua = fmaf(uv[2][0], ga[2], fmaf(uv[1][0], ga[1], uv[0][0] * ga[0]));
        p->Ub = fmaf(uv[2][0], gb[2], fmaf(uv[1][0], gb[1], uv[0][0] * gb[0]));
        p->Uc = fmaf(uv[2][0], gc[2], fmaf(uv[1][0], gc[1], uv[0][0] * gc[0]));
        p->Va = fmaf(uv[2][1], ga[2], fmaf(uv[1][1], ga[1], uv[0][1] * ga[0]));
        p->Vb = fmaf(uv[2][1], gb[2], fmaf(uv[1][1], gb[1], uv[0][1] * gb[0]));
        p->Vc = fmaf(uv[2][1], gc[2], fmaf(uv[1][1], gc[1], uv[0][1] * gc[0]));
    }
    p->gouraud = gouraud;
    for (int i = 0; i < 3; ++i) {
        p->col[i] = vcol[0][i];
        p->col0[i] = vcol[0][i];
        if (gouraud) {
            p->Ca[i] = fmaf(vcol[2][i], ga[2], fmaf(vcol[1][i], ga[1], vcol[0][i] * ga[0]));
            p->Cb[i] = fmaf(vcol[2][i], gb[2], fmaf(vcol[1][i], gb[1], vcol[0][i] * gb[0]));
            p->Cc[i] = fmaf(vcol[2][i], gc[2], fmaf(vcol[1][i], gc[1], vcol[0][i] * gc[0]));
        }
    }
    /* pixel bounds: conservative for polygons (an optimisation that never changes their coverage in
     * practice), exact and part of the semantics for mesh triangles (R4m below) */
    int allpos = 1, allneg = 1;
    for (int k = 0; k < nv; ++k) { allpos &= (h[k].hw > 0.0f); allneg &= !(h[k].hw > 0.0f); }
    /* R4m applies to a mesh triangle whose every w lies in [1e-10, 1e10] (a vertex closer to the eye plane than that
     * is treated like one behind it: no box) */
    if (gouraud) for (int k = 0; k < nv; ++k) allpos &= (h[k].hw >= 1e-10f && h[k].hw <= 1e10f);
    /* a primitive with every vertex on or behind the eye plane is clipped away as a whole (GL clips geometrically;
     * in 2DH terms every point inside it has w <= 0, i.e. z_ndc > 1, and fails R6's range test anyway) */
    if (allneg) return 0;
    p->x0 = 0; p->y0 = 0; p->x1 = sc->width - 1; p->y1 = sc->height - 1;
    if (allpos) {
        float xmin = 1e30f, xmax = -1e30f, ymin = 1e30f, ymax = -1e30f;
        for (int k = 0; k < nv; ++k) {
            float X = h[k].hx / h[k].hw, Y = h[k].hy / h[k].hw;
            xmin = fminf(xmin, X); xmax = fmaxf(xmax, X);
            ymin = fminf(ymin, Y); ymax = fmaxf(ymax, Y);
        }
        if (gouraud) {
            /* R4m: a mesh triangle (GL_TRIANGLES of a vertex list) in front of the eye is rasterised inside
             * the pixel bounding box of its projected vertices, floor(min) .. floor(max).  This is
             * semantics, not only speed: the edge functions of a near-degenerate sliver are rounding
             * noise and would otherwise claim samples away from it (real rasterisers snap vertices to
             * a sub-pixel grid, where such slivers collapse to nothing). */
            float fx0 = floorf(xmin), fx1 = floorf(xmax), fy0 = floorf(ymin), fy1 = floorf(ymax);
            if (!(fx1 >= 0.0f && fy1 >= 0.0f && fx0 <= (float)(sc->width - 1) && fy0 <= (float)(sc->height - 1)))
                return 0;
            p->x0 = (int)fmaxf(fx0, 0.0f); p->x1 = (int)fminf(fx1, (float)(sc->width - 1));
            p->y0 = (int)fmaxf(fy0, 0.0f); p->y1 = (int)fminf(fy1, (float)(sc->height - 1));
            return 1;
        }
        p->x0 = clampi(floorf(xmin) - 1.0f, 0, sc->width - 1);
        p->x1 = clampi(floorf(xmax) + 1.0f, 0, sc->width - 1);
        p->y0 = clampi(floorf(ymin) - 1.0f, 0, sc->height - 1);
        p->y1 = clampi(floorf(ymax) + 1.0f, 0, sc->height - 1);
        if (xmax < -1.0f || ymax < -1.0f || xmin > (float)sc->width + 1.0f || ymin > (float)sc->height + 1.0f)
            return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------ texturing (R7, R8) */

static float lod_log2(float x)      /* R7: log2 for x >= 1 via exponent + degree-6 polynomial */
{
    uint32_t b;
    memcpy(&b, &x, 4);
    int e = (int)((b >> 23) & 255u) - 127;
    uint32_t mb = (b & 0x7fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mb, 4);
    float f = m - 1.0f;
    float p = -0.02528550662100315f;
    p = fmaf(p, f, 0.12010025978088379f);
    p = fmaf(p, f, -0.2759689688682556f);
    p = fmaf(p, f, 0.45654040575027466f);
    p = fmaf(p, f, -0.7179135084152222f);
    p = fmaf(p, f, 1.4425272941589355f);
    return fmaf(p, f, (float)e);
}

static const uint8_t *tex_level(const mwo_tex *t, int level, int *lw, int *lh)
{
    const uint8_t *p = t->rgb;
    int w = t->w, h = t->h;
    for (int l = 0; l < level; ++l) {
        p += (int64_t)w * h * 3;
        w = w > 1 ? w / 2 : 1;
        h = h > 1 ? h / 2 : 1;
    }
    *lw = w; *lh = h;
    return p;
}

static void bilinear(const mwo_tex *t, int level, float u, float v, float out[3])
{
    int w, h;
    const uint8_t *px = tex_level(t, level, &w, &h);
    float uu = u - floorf(u), vv = v - floorf(v);       /* GL_REPEAT */
    float x = fmaf(uu, (float)w, -0.5f), y = fmaf(vv, (float)h, -0.5f);
    float x0f = floorf(x), y0f = floorf(y);
    float fx = x - x0f, fy = y - y0f;
    int i0 = (int)x0f, j0 = (int)y0f;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += w;
    if (i1 >= w) i1 -= w;
    if (j0 < 0) j0 += h;
    if (j1 >= h) j1 -= h;
    const uint8_t *t00 = px + ((int64_t)j0 * w + i0) * 3, *t10 = px + ((int64_t)j0 * w + i1) * 3;
    const uint8_t *t01 = px + ((int64_t)j1 * w + i0) * 3, *t11 = px + ((int64_t)j1 * w + i1) * 3;
    for (int c = 0; c < 3; ++c) {
        float a = (float)t00[c], b = (float)t10[c], cc = (float)t01[c], d = (float)t11[c];
        float r0 = fmaf(fx, b - a, a);
        float r1 = fmaf(fx, d - cc, cc);
        out[c] = fmaf(fy, r1 - r0, r0);
    }
}

/* fragment colour at the pixel centre (R6, R7, R8, R9) */
static void shade(const mwo_scene *sc, const prim *p, float Xc, float Yc, float out[3])
{
    float Wq = fmaf(p->Wa, Xc, fmaf(p->Wb, Yc, p->Wc));
    int wok = Wq >= 1e-30f && Wq <= 1e30f;      /* R7: W <= 0 (or absurdly small / large) = degenerate */
    float iw = wok ? 1.0f / Wq : 0.0f;
    float base[3];
    if (p->gouraud && wok) {
        for (int i = 0; i < 3; ++i)
            base[i] = fmaf(p->Ca[i], Xc, fmaf(p->Cb[i], Yc, p->Cc[i])) * iw;
    } else if (p->gouraud) {
        for (int i = 0; i < 3; ++i) base[i] = p->col0[i];
    } else {
        for (int i = 0; i < 3; ++i) base[i] = p->col[i];
    }
    if (p->tex < 0) {
        for (int i = 0; i < 3; ++i) out[i] = base[i];
        return;
    }
    const mwo_tex *t = &sc->tex[p->tex];
    int q = t->nlevels - 1;
    float texel[3];
    if (!wok) {
        bilinear(t, q, 0.0f, 0.0f, texel);
    } else {
        float Uq = fmaf(p->Ua, Xc, fmaf(p->Ub, Yc, p->Uc));
        float Vq = fmaf(p->Va, Xc, fmaf(p->Vb, Yc, p->Vc));
        float u = Uq * iw, v = Vq * iw;
        float ux = (p->Ua - u * p->Wa) * iw, uy = (p->Ub - u * p->Wb) * iw;
        float vx = (p->Va - v * p->Wa) * iw, vy = (p->Vb - v * p->Wb) * iw;
        float tw = (float)t->w, th = (float)t->h;
        float sx = ux * tw, tx = vx * th, sy = uy * tw, ty = vy * th;
        float r2x = fmaf(sx, sx, tx * tx), r2y = fmaf(sy, sy, ty * ty);
        float rho2 = r2x > r2y ? r2x : r2y;
        if (!(rho2 > 1.0f)) {                       /* magnification: GL_LINEAR on level 0 */
            bilinear(t, 0, u, v, texel);
        } else if (!(rho2 < 1e30f)) {               /* inf / nan guard */
            bilinear(t, q, u, v, texel);
        } else {
            float lam = 0.5f * lod_log2(rho2);
            float lf = floorf(lam);
            int l0 = (int)lf;
            if (l0 >= q) {
                bilinear(t, q, u, v, texel);
            } else {
                float fr = lam - lf;
                float c0[3], c1[3];
                bilinear(t, l0, u, v, c0);
                bilinear(t, l0 + 1, u, v, c1);
                for (int i = 0; i < 3; ++i) texel[i] = fmaf(fr, c1[i] - c0[i], c0[i]);
            }
        }
    }
    for (int i = 0; i < 3; ++i) out[i] = (texel[i] * (1.0f / 255.0f)) * base[i];   /* GL_MODULATE */
}

/* ------------------------------------------------------------------ raster (R4, R5, R6) */

typedef struct {
    int W, H, S;
    const int (*pat)[2];
    uint16_t *zbuf;     /* [H][W][S] */
    int32_t *ibuf;      /* draw index, -1 = clear */
    float *cbuf;        /* [H][W][S][3] */
} target;

/* returns the number of samples that passed the depth test (what an occlusion query counts) */
static int draw_prim(const mwo_scene *sc, target *tg, const prim *p, int draw_index)
{
    int passed = 0;
    float thr[4][MAXS], zo[MAXS];
    for (int s = 0; s < tg->S; ++s) {
        float dx = (float)(tg->pat[s][0] - 8) * 0.0625f, dy = (float)(tg->pat[s][1] - 8) * 0.0625f;
        for (int k = 0; k < p->nv; ++k) thr[k][s] = -fmaf(p->ea[k], dx, p->eb[k] * dy);
        zo[s] = fmaf(p->zx, dx, p->zy * dy);
    }
    for (int py = p->y0; py <= p->y1; ++py)
        for (int px = p->x0; px <= p->x1; ++px) {
            float Xc = (float)px + 0.5f, Yc = (float)py + 0.5f;
            float E[4];
            for (int k = 0; k < p->nv; ++k) E[k] = fmaf(p->ea[k], Xc, fmaf(p->eb[k], Yc, p->ec[k]));
            float zc = fmaf(p->zx, Xc, fmaf(p->zy, Yc, p->zc));
            int64_t base = ((int64_t)py * tg->W + px) * tg->S;
            int shaded = 0;
            float col[3];
            for (int s = 0; s < tg->S; ++s) {
                int in = 1;
                for (int k = 0; k < p->nv; ++k)
                    in &= (E[k] > thr[k][s]) || (E[k] == thr[k][s] && p->tl[k]);
                if (!in) continue;
                float zs = zc + zo[s];
                float t = fmaf(zs, 65535.0f, 0.5f);
                if (!(t >= 0.5f && t < 65536.0f)) continue;      /* near / far clip (R6) */
                uint16_t z16 = (uint16_t)(uint32_t)t;
                if (!(z16 < tg->zbuf[base + s])) continue;        /* GL_LESS */
                if (!shaded) { shade(sc, p, Xc, Yc, col); shaded = 1; }
                ++passed;
                tg->zbuf[base + s] = z16;
                tg->ibuf[base + s] = draw_index;
                memcpy(&tg->cbuf[(base + s) * 3], col, sizeof col);
            }
        }
    return passed;
}

static int draw_poly(const mwo_scene *sc, const camera *cam, target *tg, const float (*v)[3],
                     const float (*uv)[2], const float n[3], const float base[3], int nv, int tex,
                     int draw_index)
{
    hvert h[4];
    float vcol[3][3];
    for (int k = 0; k < nv; ++k) h[k] = xform(cam, v[k][0], v[k][1], v[k][2]);
    light(cam, n, base, vcol[0]);
    memcpy(vcol[1], vcol[0], sizeof vcol[0]);
    memcpy(vcol[2], vcol[0], sizeof vcol[0]);
    prim p;
    if (setup_prim(sc, h, nv, uv, (const float (*)[3])vcol, 0, tex, &p)) return draw_prim(sc, tg, &p, draw_index);
    return 0;
}

/* opengl.py:460-503 drawBox, vertex order and normals as listed there */
static const int BOXV[6][4][3] = {
    {{1, 1, 1}, {0, 1, 1}, {0, 0, 1}, {1, 0, 1}},
    {{0, 1, 0}, {1, 1, 0}, {1, 0, 0}, {0, 0, 0}},
    {{0, 1, 1}, {0, 1, 0}, {0, 0, 0}, {0, 0, 1}},
    {{1, 1, 0}, {1, 1, 1}, {1, 0, 1}, {1, 0, 0}},
    {{1, 1, 1}, {1, 1, 0}, {0, 1, 0}, {0, 1, 1}},
    {{1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 0, 0}},
};
static const float BOXN[6][3] = {{0, 0, 1}, {0, 0, -1}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}};

int mwo_render_obs(const mwo_scene *sc, uint8_t *rgb, uint16_t *z16out, float *depth, int32_t *primout)
{
    target tg;
    tg.W = sc->width; tg.H = sc->height; tg.S = sc->nsamples;
    switch (sc->nsamples) {
    case 1: tg.pat = PAT1; break;
    case 4: tg.pat = PAT4; break;
    case 8: tg.pat = PAT8; break;
    case 16: tg.pat = PAT16; break;
    default: return -1;
    }
    int64_t ns = (int64_t)tg.W * tg.H * tg.S;
    tg.zbuf = (uint16_t *)malloc(ns * sizeof(uint16_t));
    tg.ibuf = (int32_t *)malloc(ns * sizeof(int32_t));
    tg.cbuf = (float *)malloc(ns * 3 * sizeof(float));
    if (!tg.zbuf || !tg.ibuf || !tg.cbuf) return -2;
    camera cam;
    build_camera(sc, &cam);
    /* glClear (miniworld.py:1193-1195) */
    for (int64_t i = 0; i < ns; ++i) {
        tg.zbuf[i] = 65535;
        tg.ibuf[i] = -1;
        tg.cbuf[i * 3 + 0] = cam.sky[0]; tg.cbuf[i * 3 + 1] = cam.sky[1]; tg.cbuf[i * 3 + 2] = cam.sky[2];
    }
    int draw = 0;
    float stale_n[3] = {0.0f, 1.0f, 0.0f};      /* the GL "current normal" left behind by the last draw */
    /* display list 1: rooms (miniworld.py:1053-1055) */
    for (int i = 0; i < sc->n_polys; ++i, ++draw) {
        const mwo_poly *q = &sc->polys[i];
        draw_poly(sc, &cam, &tg, q->v, q->uv, q->n, q->rgb, q->nv & 0xFF, q->tex, draw);
        memcpy(stale_n, q->n, sizeof stale_n);
    }
    /* entities, already in draw order */
    for (int e = 0; e < sc->n_ents; ++e) {
        const mwo_ent *en = &sc->ents[e];
        if (en->kind == MWO_ENT_NONE) continue;
        double sd, cd;
        mwo_sincos(en->dir, &sd, &cd);
        float c = (float)cd, s = (float)sd;
        float px = (float)en->pos[0], py = (float)en->pos[1], pz = (float)en->pos[2];
        if (en->kind == MWO_ENT_BOX) {
            /* Box.render (entity.py:409-432): T(pos) * R_y(dir) * drawBox(-sx/2..sx/2, 0..sy, -sz/2..sz/2) */
            float lo[3] = {(float)(-en->size[0] / 2), 0.0f, (float)(-en->size[2] / 2)};
            float hi[3] = {(float)(en->size[0] / 2), (float)en->size[1], (float)(en->size[2] / 2)};
            float base[3] = {(float)en->color[0], (float)en->color[1], (float)en->color[2]};
            for (int f = 0; f < 6; ++f, ++draw) {
                float v[4][3], n[3];
                for (int k = 0; k < 4; ++k) {
                    float lx = BOXV[f][k][0] ? hi[0] : lo[0];
                    float ly = BOXV[f][k][1] ? hi[1] : lo[1];
                    float lz = BOXV[f][k][2] ? hi[2] : lo[2];
                    v[k][0] = fmaf(c, lx, s * lz) + px;        /* R11: R_y(dir) then translate */
                    v[k][1] = ly + py;
                    v[k][2] = fmaf(c, lz, -(s * lx)) + pz;
                }
                n[0] = fmaf(c, BOXN[f][0], s * BOXN[f][2]);
                n[1] = BOXN[f][1];
                n[2] = fmaf(c, BOXN[f][2], -(s * BOXN[f][0]));
                draw_poly(sc, &cam, &tg, (const float (*)[3])v, NULL, n, base, 4, -1, draw);
            }
            stale_n[0] = 0.0f; stale_n[1] = -1.0f; stale_n[2] = 0.0f;      /* drawBox ends with glNormal3f(0,-1,0) */
        } else if (en->kind == MWO_ENT_MESH) {
            /* MeshEnt.render (entity.py:150-161): T(pos) * S(scale) * R_y(dir); normals through
             * the inverse transpose WITHOUT renormalisation => R_y(dir) n / scale (R11) */
            const mwo_mesh *m = &sc->meshes[en->mesh];
            float sc_ = (float)en->scale;
            for (int t = 0; t < m->ntris; ++t, ++draw) {
                hvert h[3];
                float vcol[3][3], uv[3][2];
                for (int k = 0; k < 3; ++k) {
                    const float *lp = &m->pos[(t * 3 + k) * 3], *ln = &m->nrm[(t * 3 + k) * 3];
                    float rx = fmaf(c, lp[0], s * lp[2]), rz = fmaf(c, lp[2], -(s * lp[0]));
                    float wx = fmaf(sc_, rx, px), wy = fmaf(sc_, lp[1], py), wz = fmaf(sc_, rz, pz);
                    h[k] = xform(&cam, wx, wy, wz);
                    float n[3] = {fmaf(c, ln[0], s * ln[2]) / sc_, ln[1] / sc_, fmaf(c, ln[2], -(s * ln[0])) / sc_};
                    light(&cam, n, &m->rgb[(t * 3 + k) * 3], vcol[k]);
                    uv[k][0] = m->uv[(t * 3 + k) * 2];
                    uv[k][1] = m->uv[(t * 3 + k) * 2 + 1];
                }
                prim p;
                if (setup_prim(sc, h, 3, (const float (*)[2])uv, (const float (*)[3])vcol, 1, m->tex, &p))
                    draw_prim(sc, &tg, &p, draw);
            }
            if (m->ntris > 0) memcpy(stale_n, &m->nrm[((size_t)(m->ntris - 1) * 3 + 2) * 3], sizeof stale_n);
        }
    }
    if (sc->render_agent) {
        /* Agent.render (entity.py:518-539): a red triangle at the top of the agent's cylinder, drawn
         * without any glNormal3f => lit with the normal the previous draw left behind, under the
         * camera modelview only (no renormalisation, no model scale).  Pinned untextured. */
        double sd, cd;
        mwo_sincos(sc->agent_dir, &sd, &cd);
        double rad = sc->agent_radius, hgt = sc->agent_height;
        double p[3] = {sc->agent_pos[0] + 0 * hgt, sc->agent_pos[1] + 1 * hgt, sc->agent_pos[2] + 0 * hgt};
        double dv[3] = {cd * rad, 0 * rad, -sd * rad}, rv[3] = {sd * rad, 0 * rad, cd * rad};
        double p0[3], p1[3], p2[3];
        for (int i = 0; i < 3; ++i) {
            p0[i] = p[i] + dv[i];
            p1[i] = p[i] + 0.75 * (rv[i] - dv[i]);
            p2[i] = p[i] + 0.75 * (-rv[i] - dv[i]);
        }
        float v[4][3] = {{(float)p0[0], (float)p0[1], (float)p0[2]}, {(float)p2[0], (float)p2[1], (float)p2[2]},
                         {(float)p1[0], (float)p1[1], (float)p1[2]}, {0, 0, 0}};
        static const float red[3] = {1.0f, 0.0f, 0.0f};
        draw_poly(sc, &cam, &tg, (const float (*)[3])v, NULL, stale_n, red, 3, -1, draw);
        ++draw;
    }
    /* R12: resolve (opengl.py:339-398): mean of the samples in float; the samples of one
     * primitive carry one colour, and the groups are accumulated in ascending draw index,
     * uncovered (sky) samples last; 8-bit conversion round-half-up. */
    for (int py = 0; py < tg.H; ++py)
        for (int px = 0; px < tg.W; ++px) {
            int64_t base = ((int64_t)py * tg.W + px) * tg.S;
            float acc[3] = {0.0f, 0.0f, 0.0f};
            int done[MAXS] = {0};
            for (;;) {
                int best = -1;
                for (int s = 0; s < tg.S; ++s) {
                    if (done[s]) continue;
                    int64_t ks = tg.ibuf[base + s] < 0 ? 0x7fffffff : tg.ibuf[base + s];
                    int64_t kb = best < 0 ? -1 : (tg.ibuf[base + best] < 0 ? 0x7fffffff : tg.ibuf[base + best]);
                    if (best < 0 || ks < kb) best = s;
                }
                if (best < 0) break;
                int cnt = 0;
                for (int r = 0; r < tg.S; ++r)
                    if (!done[r] && tg.ibuf[base + r] == tg.ibuf[base + best]) { done[r] = 1; ++cnt; }
                for (int c = 0; c < 3; ++c) acc[c] = fmaf((float)cnt, tg.cbuf[(base + best) * 3 + c], acc[c]);
            }
            float inv = 1.0f / (float)tg.S;
            for (int c = 0; c < 3; ++c) {
                float v = acc[c] * inv;
                v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                rgb[((int64_t)py * tg.W + px) * 3 + c] = (uint8_t)(int)fmaf(v, 255.0f, 0.5f);
            }
            /* R13: depth resolve = sample 0 (GL_NEAREST blit, opengl.py:361-372) */
            uint16_t z = tg.zbuf[base];
            if (z16out) z16out[(int64_t)py * tg.W + px] = z;
            if (depth) {
                /* R14: get_depth_map (opengl.py:426-431) in float32, as numpy evaluates it */
                float d = (float)z / 65535.0f;
                float clip = (d - 0.5f) * 2.0f;
                float zfar = 100.0f, znear = 0.04f;
                float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
                (void)zfar; (void)znear;
                depth[(int64_t)py * tg.W + px] = (float)(-2.0 * 100.0 * 0.04) / den;
            }
            if (primout)
                for (int s = 0; s < tg.S; ++s) primout[base + s] = tg.ibuf[base + s];
        }
    free(tg.zbuf); free(tg.ibuf); free(tg.cbuf);
    return 0;
}

/* MiniWorldEnv.get_visible_ents (miniworld.py:1238-1333): the rooms are drawn untextured into the
 * cleared obs frame buffer (same camera as render_obs, :1263-1288; Room._render :1291-1293), then,
 * per entity in the order of self.entities, an axis-aligned 0.2 m proxy box at ent.pos is drawn
 * inside a GL_ANY_SAMPLES_PASSED query (:1296-1313): depth-tested (GL_LESS) AND depth-written, so
 * an earlier proxy can hide a later one; back faces are culled (:512) and produce no samples.
 * sc->ents must be in self.entities order here (not draw order); vis[i] = query result != 0. */
int mwo_visible_ents(const mwo_scene *sc, uint8_t *vis)
{
    target tg;
    tg.W = sc->width; tg.H = sc->height; tg.S = sc->nsamples;
    switch (sc->nsamples) {
    case 1: tg.pat = PAT1; break;
    case 4: tg.pat = PAT4; break;
    case 8: tg.pat = PAT8; break;
    case 16: tg.pat = PAT16; break;
    default: return -1;
    }
    int64_t ns = (int64_t)tg.W * tg.H * tg.S;
    tg.zbuf = (uint16_t *)malloc(ns * sizeof(uint16_t));
    tg.ibuf = (int32_t *)malloc(ns * sizeof(int32_t));
    tg.cbuf = (float *)malloc(ns * 3 * sizeof(float));
    if (!tg.zbuf || !tg.ibuf || !tg.cbuf) return -2;
    camera cam;
    build_camera(sc, &cam);
    for (int64_t i = 0; i < ns; ++i) { tg.zbuf[i] = 65535; tg.ibuf[i] = -1; }
    memset(tg.cbuf, 0, ns * 3 * sizeof(float));
    static const float white[3] = {1.0f, 1.0f, 1.0f};
    int draw = 0;
    for (int i = 0; i < sc->n_polys; ++i, ++draw) {
        const mwo_poly *q = &sc->polys[i];
        if (q->nv & MWO_POLY_ENTITY) continue;          /* only room._render() is drawn (:1291-1293) */
        draw_poly(sc, &cam, &tg, q->v, q->uv, q->n, white, q->nv, -1, draw);        /* glDisable(GL_TEXTURE_2D) */
    }
    for (int e = 0; e < sc->n_ents; ++e) {
        const mwo_ent *en = &sc->ents[e];
        vis[e] = 0;
        if (en->kind == MWO_ENT_NONE) continue;
        /* drawBox arguments are computed in double (python floats) and reach GL through glVertex3f */
        float lo[3] = {(float)(en->pos[0] - 0.1), (float)en->pos[1], (float)(en->pos[2] - 0.1)};
        float hi[3] = {(float)(en->pos[0] + 0.1), (float)(en->pos[1] + 0.2), (float)(en->pos[2] + 0.1)};
        int passed = 0;
        for (int f = 0; f < 6; ++f, ++draw) {
            float v[4][3];
            for (int k = 0; k < 4; ++k) {
                v[k][0] = BOXV[f][k][0] ? hi[0] : lo[0];
                v[k][1] = BOXV[f][k][1] ? hi[1] : lo[1];
                v[k][2] = BOXV[f][k][2] ? hi[2] : lo[2];
            }
            passed += draw_poly(sc, &cam, &tg, (const float (*)[3])v, NULL, BOXN[f], white, 4, -1, draw);
        }
        vis[e] = passed != 0;
    }
    free(tg.zbuf); free(tg.ibuf); free(tg.cbuf);
    return 0;
}
