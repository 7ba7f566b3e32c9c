/* ORACLE — TEST INFRASTRUCTURE ONLY.  Raster half of the pixel oracle (see mwo_gl.h for the vertex half).
 *
 * CPU restatement of what the reference's frame goes through after the vertex stage:
 *   FrameBuffer (multisampled RGBA32F + DEPTH16, resolve, readback, flip)  miniworld/opengl.py:202-398
 *   FrameBuffer.get_depth_map                                                miniworld/opengl.py:400-435
 *   Texture.load (GL_RGB, glGenerateMipmap, GL_LINEAR_MIPMAP_LINEAR, REPEAT)  miniworld/opengl.py:148-184
 *   glEnable(GL_DEPTH_TEST), glEnable(GL_CULL_FACE)                          miniworld/miniworld.py:511-512
 *   get_visible_ents (GL_ANY_SAMPLES_PASSED queries)                         miniworld/miniworld.py:1238-1333
 *
 * PARITY: PINNED on the driver the reference's own CI uses — Mesa 23.2.1 llvmpipe (LLVM 15, x86-64 with FMA), where
 * the reference asks for 8 / 16 samples and gets GL_MAX_SAMPLES = 4 (opengl.py:229-231).  tools/refshim_gl.py runs
 * /root/reference/miniworld unmodified on that driver; tests/golden/gl_*.npz are its frames;
 * tests/test_oracle_vs_reference_gl.py compares this file with them (RGB, 16-bit depth, depth map, top view,
 * visible entities).  The arithmetic below restates llvmpipe (third-party, not in /root/reference):
 *   lp_setup_tri.c        24.8 fixed-point vertex snap, integer edge functions, fill rule, face culling on the
 *                         snapped area, front faces set up in the order (v1, v0, v2)
 *   lp_state_setup.c      plane coefficients a0, dadx, dady of z, 1/w and the perspective attributes
 *   lp_bld_interp.c       a = fma(dady, y, fma(dadx, x, a0)), times 1 / (1/w); z at the sample position
 *   lp_bld_depth.c        z -> 16-bit: round-to-nearest-even of fl(z * 65535/65536) * 65536, GL_LESS
 *   lp_bld_sample*.c      per-quad rho^2 from pixel differences, lod = 0.5 * (exponent + mantissa - 1),
 *                         8.8 fixed-point texel coordinates, 8-bit lerp weights with +128 rounding
 *   u_blitter / u_simple_shaders  resolve = ((s0 + s1) + s2 + s3) * (1/n), sample 0 for depth
 *   lp_bld_conv.c         float -> unorm8 = round-to-nearest-even of fl(c * 255)
 * With 8 or 16 samples (no llvmpipe counterpart) the same rules run on the D3D standard patterns.
 *
 * Compile with -ffp-contract=off: every fused multiply-add below is an explicit fmaf().
 */
#include "mwo_gl.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXS 16
#define FIXED_ONE 256

/* sample positions in 1/16 pixel inside the pixel, GL frame-buffer space (x right, y UP): what
 * glGetMultisamplefv(GL_SAMPLE_POSITION) reports for the 4-sample FBO; 8 / 16: D3D standard patterns */
static const int PAT1[1][2] = {{8, 8}};
static const int PAT4[4][2] = {{6, 2}, {14, 6}, {2, 10}, {10, 14}};
static const int PAT8[8][2] = {{9, 5}, {7, 11}, {13, 9}, {5, 3}, {3, 13}, {1, 7}, {11, 15}, {15, 1}};
static const int PAT16[16][2] = {{9, 9}, {7, 5}, {5, 10}, {12, 7}, {3, 6}, {10, 13}, {13, 11}, {11, 3},
                                 {6, 14}, {8, 1}, {4, 2}, {2, 12}, {0, 8}, {15, 4}, {14, 15}, {1, 0}};

typedef struct {
    int W, H, S;
    const int (*pat)[2];
    uint16_t *zbuf;     /* [H][W][S], row 0 = GL y 0 (bottom) */
    int32_t *ibuf;      /* draw index, -1 = clear */
    float *cbuf;        /* [H][W][S][3] */
} target;

typedef struct { float a0, dadx, dady; } plane;

typedef struct {
    int32_t x[3], y[3];             /* snapped vertices, 24.8 (see pixel_offset in setup)  */
    float eval_off;                 /* pixel centre in the coordinates the planes are set up in */
    int32_t dcdx[3], dcdy[3];
    int64_t c[3];
    plane z, w, s, t, col[3];       /* s, t, col hold attribute / w                        */
    int tex, draw;
    int minx, maxx, miny, maxy;     /* pixel bounds                                         */
} setup_tri;

static inline int32_t iround_even(float f) { return (int32_t)lrintf(f); }

/* lp_state_setup.c calc_coef4 */
static inline void coef(plane *p, float a0, float a1, float a2, float dy20_ooa, float dy01_ooa, float dx20_ooa,
                        float dx01_ooa, float x0c, float y0c)
{
    float da01 = a0 - a1, da20 = a2 - a0;
    p->dadx = da01 * dy20_ooa - da20 * dy01_ooa;
    p->dady = da20 * dx01_ooa - da01 * dx20_ooa;
    p->a0 = a0 - (p->dadx * x0c + p->dady * y0c);
}


static int setup(const target *tg, const mwo_tri *tri, setup_tri *s)
{
    const mwo_vert *v0 = &tri->v[0], *v1 = &tri->v[1], *v2 = &tri->v[2];
    int32_t fx[3], fy[3];
    /* lp_setup: a single-sampled target shifts the vertices by half a pixel (pixel_offset 0.5: integer coordinates
     * are pixel centres); a multisampled one does not (integer coordinates are pixel corners, the samples sit at
     * their positions inside the pixel and attributes are evaluated at x + 0.5) */
    const float pixel_offset = tg->S > 1 ? 0.0f : 0.5f;
    s->eval_off = 0.5f - pixel_offset;
    for (int i = 0; i < 3; ++i) {
        fx[i] = iround_even((tri->v[i].win[0] - pixel_offset) * (float)FIXED_ONE);
        fy[i] = iround_even((tri->v[i].win[1] - pixel_offset) * (float)FIXED_ONE);
    }
    int64_t dx01 = fx[0] - fx[1], dy01 = fy[0] - fy[1], dx20 = fx[2] - fx[0], dy20 = fy[2] - fy[0];
    int64_t area = dx01 * dy20 - dx20 * dy01;
    /* GL_CULL_FACE, GL_BACK, front = counter-clockwise in window space (y up): area < 0 here; zero area is culled */
    if (area >= 0) return 0;
    /* triangle_cw -> rotate_fixed_position_01 -> do_triangle_ccw(v1, v0, v2) */
    const mwo_vert *t = v0; v0 = v1; v1 = t;
    int32_t ti = fx[0]; fx[0] = fx[1]; fx[1] = ti;
    ti = fy[0]; fy[0] = fy[1]; fy[1] = ti;
    for (int i = 0; i < 3; ++i) { s->x[i] = fx[i]; s->y[i] = fy[i]; }
    for (int i = 0; i < 3; ++i) {
        int j = (i + 1) % 3;
        s->dcdy[i] = fx[i] - fx[j];
        s->dcdx[i] = fy[i] - fy[j];
        s->c[i] = (int64_t)s->dcdx[i] * fx[i] - (int64_t)s->dcdy[i] * fy[i];
        /* fill rule: an edge owns the samples exactly on it when it is a left edge, or a horizontal edge on the side
         * the frame buffer's bottom_edge_rule selects */
        if (s->dcdx[i] < 0) s->c[i]++;
        else if (s->dcdx[i] == 0 && s->dcdy[i] > 0) s->c[i]++;
    }
    int minx = fx[0] < fx[1] ? fx[0] : fx[1]; if (fx[2] < minx) minx = fx[2];
    int maxx = fx[0] > fx[1] ? fx[0] : fx[1]; if (fx[2] > maxx) maxx = fx[2];
    int miny = fy[0] < fy[1] ? fy[0] : fy[1]; if (fy[2] < miny) miny = fy[2];
    int maxy = fy[0] > fy[1] ? fy[0] : fy[1]; if (fy[2] > maxy) maxy = fy[2];
    /* conservative pixel range */
    s->minx = (minx >> 8) - 1; s->maxx = (maxx >> 8) + 1;
    s->miny = (miny >> 8) - 1; s->maxy = (maxy >> 8) + 1;
    if (s->minx < 0) s->minx = 0;
    if (s->miny < 0) s->miny = 0;
    if (s->maxx > tg->W - 1) s->maxx = tg->W - 1;
    if (s->maxy > tg->H - 1) s->maxy = tg->H - 1;
    if (s->minx > s->maxx || s->miny > s->maxy) return 0;
    /* lp_state_setup.c init_args, on the unsnapped float window coordinates */
    float fdx01 = v0->win[0] - v1->win[0], fdy01 = v0->win[1] - v1->win[1];
    float fdx20 = v2->win[0] - v0->win[0], fdy20 = v2->win[1] - v0->win[1];
    float ooa = 1.0f / (fdx01 * fdy20 - fdx20 * fdy01);
    float dy20_ooa = fdy20 * ooa, dy01_ooa = fdy01 * ooa, dx20_ooa = fdx20 * ooa, dx01_ooa = fdx01 * ooa;
    float x0c = v0->win[0] - pixel_offset, y0c = v0->win[1] - pixel_offset;
#define COEF(p, a, b, c) coef(p, a, b, c, dy20_ooa, dy01_ooa, dx20_ooa, dx01_ooa, x0c, y0c)
    COEF(&s->z, v0->win[2], v1->win[2], v2->win[2]);
    COEF(&s->w, v0->win[3], v1->win[3], v2->win[3]);
    COEF(&s->s, v0->st[0] * v0->win[3], v1->st[0] * v1->win[3], v2->st[0] * v2->win[3]);
    COEF(&s->t, v0->st[1] * v0->win[3], v1->st[1] * v1->win[3], v2->st[1] * v2->win[3]);
    for (int k = 0; k < 3; ++k)
        COEF(&s->col[k], v0->col[k] * v0->win[3], v1->col[k] * v1->win[3], v2->col[k] * v2->win[3]);
#undef COEF
    s->tex = tri->tex;
    s->draw = tri->draw;
    return 1;
}

static inline float interp(const plane *p, float x, float y) { return fmaf(p->dady, y, fmaf(p->dadx, x, p->a0)); }

/* ------------------------------------------------------------------ texture sampler */

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

static inline int lerp8(int a, int b, int w) { return a + ((w * (b - a) + 128) >> 8); }

/* one axis of lp_build_sample_image_linear (AoS path): texel pair and 8-bit weight, GL_REPEAT */
static inline void linear_coord(float coord, int size, int *i0, int *i1, int *wt)
{
    if ((size & (size - 1)) == 0) {
        int32_t f = iround_even(coord * (float)size * 256.0f) - 128;
        *wt = f & 255;
        *i0 = (f >> 8) & (size - 1);
        *i1 = (*i0 + 1) & (size - 1);
    } else {
        /* lp_build_coord_repeat_npot_linear_int: fract, scale, then the half-texel shift */
        float fr = coord - floorf(coord);
        int32_t f = iround_even(fr * (float)size * 256.0f) - 128;
        *wt = f & 255;
        int32_t i = f >> 8;
        if (i < 0) i = size - 1;
        if (i > size - 1) i = size - 1;
        *i0 = i;
        *i1 = (i == size - 1) ? 0 : i + 1;
    }
}

static void bilinear8(const mwo_tex *t, int level, float s, float tt, int out[3])
{
    int w, h, i0, i1, j0, j1, wx, wy;
    const uint8_t *px = tex_level(t, level, &w, &h);
    linear_coord(s, w, &i0, &i1, &wx);
    linear_coord(tt, h, &j0, &j1, &wy);
    const uint8_t *t00 = px + ((int64_t)j0 * w + i0) * 3, *t10 = px + ((int64_t)j0 * w + i1) * 3;
    const uint8_t *t01 = px + ((int64_t)j1 * w + i0) * 3, *t11 = px + ((int64_t)j1 * w + i1) * 3;
    for (int c = 0; c < 3; ++c) out[c] = lerp8(lerp8(t00[c], t10[c], wx), lerp8(t01[c], t11[c], wx), wy);
}

/* fixed-function texturing is projective: the lookup coordinate is (s, t) * (1 / q) with q the interpolated fourth
 * texture coordinate — 1 at every vertex, so its plane IS the 1/w plane and q = fl(W' * fl(1 / W')), one ulp around 1 */
static inline void tex_coords(const setup_tri *p, float x, float y, float *s, float *t)
{
    float wv = interp(&p->w, x, y);
    float oow = 1.0f / wv;
    float invq = 1.0f / (wv * oow);
    *s = (interp(&p->s, x, y) * oow) * invq;
    *t = (interp(&p->t, x, y) * oow) * invq;
}

/* fragment colour of pixel (px, py) (GL coordinates); the lod comes from the pixel's 2x2 quad */
static void shade(const mwo_scene *sc, const setup_tri *p, int px, int py, float out[3])
{
    float x = (float)px + p->eval_off, y = (float)py + p->eval_off;
    float oow = 1.0f / interp(&p->w, x, y);
    float col[3];
    for (int k = 0; k < 3; ++k) col[k] = interp(&p->col[k], x, y) * oow;
    if (p->tex < 0) { memcpy(out, col, sizeof col); return; }
    const mwo_tex *tx = &sc->tex[p->tex];
    float s, t;
    tex_coords(p, x, y, &s, &t);
    /* lp_build_rho: differences over the quad's first row / first column */
    float qx = (float)(px & ~1) + p->eval_off, qy = (float)(py & ~1) + p->eval_off;
    float s00, t00, s10, t10, s01, t01;
    tex_coords(p, qx, qy, &s00, &t00);
    tex_coords(p, qx + 1.0f, qy, &s10, &t10);
    tex_coords(p, qx, qy + 1.0f, &s01, &t01);
    float fw = (float)tx->w, fh = (float)tx->h;
    float dsdx = (s10 - s00) * fw, dsdy = (s01 - s00) * fw, dtdx = (t10 - t00) * fh, dtdy = (t01 - t00) * fh;
    float rx = dsdx * dsdx + dtdx * dtdx, ry = dsdy * dsdy + dtdy * dtdy;
    float rho2 = rx > ry ? rx : ry;
    /* lp_build_lod_selector: lod = 0.5 * fast_log2(rho^2), fast_log2(x) = exponent + (mantissa - 1) */
    uint32_t b;
    memcpy(&b, &rho2, 4);
    int e = (int)((b >> 23) & 255u) - 127;
    uint32_t mb = (b & 0x7fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mb, 4);
    float lod = ((float)e + (m - 1.0f)) * 0.5f;
    float fl = floorf(lod);
    int ip = (int)fl;
    float fp = lod - fl;
    int last = tx->nlevels - 1, l0 = ip, w8;
    if (!(rho2 > 0.0f) || ip < 0) { l0 = 0; fp = 0.0f; }         /* magnification (and rho = 0: exponent -127) */
    else if (ip >= last) { l0 = last; fp = 0.0f; }
    w8 = (int)(fp * 256.0f);
    int c0[3], c1[3];
    bilinear8(tx, l0, s, t, c0);
    if (w8 > 0) {
        bilinear8(tx, l0 + 1 > last ? last : l0 + 1, s, t, c1);
        for (int k = 0; k < 3; ++k) c0[k] = lerp8(c0[k], c1[k], w8);
    }
    for (int k = 0; k < 3; ++k) out[k] = ((float)c0[k] * (1.0f / 255.0f)) * col[k];      /* GL_MODULATE */
}

/* ------------------------------------------------------------------ raster */

static inline uint16_t z_to_unorm16(float z)
{
    /* lp_build_clamped_float_to_unsigned_norm(16): z in [0, 1] */
    if (!(z > 0.0f)) z = 0.0f;
    if (z > 1.0f) z = 1.0f;
    float r = z * (65535.0f / 65536.0f) + 128.0f;
    uint32_t u;
    memcpy(&u, &r, 4);
    return (uint16_t)(u & 0xffffu);
}

/* returns the number of samples that passed the depth test */
static int draw_tri(const mwo_scene *sc, target *tg, const setup_tri *p)
{
    int passed = 0;
    for (int py = p->miny; py <= p->maxy; ++py)
        for (int px = p->minx; px <= p->maxx; ++px) {
            int64_t base = ((int64_t)py * tg->W + px) * tg->S;
            int shaded = 0;
            float col[3];
            for (int s = 0; s < tg->S; ++s) {
                /* sample position in the snapped space */
                int32_t fx, fy;
                float xs, ys;
                if (tg->S > 1) {
                    fx = px * FIXED_ONE + tg->pat[s][0] * 16; fy = py * FIXED_ONE + tg->pat[s][1] * 16;
                    xs = (float)px + (float)tg->pat[s][0] * 0.0625f; ys = (float)py + (float)tg->pat[s][1] * 0.0625f;
                } else {
                    fx = px * FIXED_ONE; fy = py * FIXED_ONE;
                    xs = (float)px; ys = (float)py;
                }
                int in = 1;
                for (int k = 0; k < 3; ++k)
                    in &= (p->c[k] + (int64_t)p->dcdy[k] * fy - (int64_t)p->dcdx[k] * fx) > 0;
                if (!in) continue;
                uint16_t z16 = z_to_unorm16(interp(&p->z, xs, ys));
                if (!(z16 < tg->zbuf[base + s])) continue;        /* GL_LESS */
                if (!shaded) { shade(sc, p, px, py, col); shaded = 1; }
                ++passed;
                tg->zbuf[base + s] = z16;
                tg->ibuf[base + s] = p->draw;
                memcpy(&tg->cbuf[(base + s) * 3], col, sizeof col);
            }
        }
    return passed;
}

static int target_init(target *tg, const mwo_scene *sc)
{
    tg->W = sc->width; tg->H = sc->height; tg->S = sc->nsamples;
    switch (sc->nsamples) {
    case 1: tg->pat = PAT1; break;
    case 4: tg->pat = PAT4; break;
    case 8: tg->pat = PAT8; break;
    case 16: tg->pat = PAT16; break;
    default: return -1;
    }
    int64_t ns = (int64_t)tg->W * tg->H * tg->S;
    tg->zbuf = (uint16_t *)malloc(ns * sizeof(uint16_t));
    tg->ibuf = (int32_t *)malloc(ns * sizeof(int32_t));
    tg->cbuf = (float *)malloc(ns * 3 * sizeof(float));
    if (!tg->zbuf || !tg->ibuf || !tg->cbuf) return -2;
    return 0;
}

static void target_free(target *tg) { free(tg->zbuf); free(tg->ibuf); free(tg->cbuf); }

static inline uint8_t float_to_unorm8(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)lrintf(v * 255.0f);
}

int mwo_render_obs(const mwo_scene *sc, uint8_t *rgb, uint16_t *z16out, float *depth, int32_t *primout)
{
    target tg;
    int rc = target_init(&tg, sc);
    if (rc) return rc;
    mwo_trilist tris = {0};
    rc = mwo_geometry(sc, 0, &tris, NULL);
    if (rc) { target_free(&tg); mwo_trilist_free(&tris); return rc; }
    int64_t ns = (int64_t)tg.W * tg.H * tg.S;
    /* glClear (miniworld.py:1193-1195) */
    float sky[3] = {(float)sc->sky[0], (float)sc->sky[1], (float)sc->sky[2]};
    for (int64_t i = 0; i < ns; ++i) {
        tg.zbuf[i] = 65535;
        tg.ibuf[i] = -1;
        tg.cbuf[i * 3 + 0] = sky[0]; tg.cbuf[i * 3 + 1] = sky[1]; tg.cbuf[i * 3 + 2] = sky[2];
    }
    for (int i = 0; i < tris.n; ++i) {
        setup_tri st;
        if (setup(&tg, &tris.tris[i], &st)) draw_tri(sc, &tg, &st);
    }
    mwo_trilist_free(&tris);
    /* resolve (opengl.py:339-398): blit = sum of the samples in index order times 1/n, float -> unorm8; the numpy flip
     * turns GL's bottom-up rows into image rows */
    float inv = 1.0f / (float)tg.S;
    for (int gy = 0; gy < tg.H; ++gy)
        for (int px = 0; px < tg.W; ++px) {
            int64_t base = ((int64_t)gy * tg.W + px) * tg.S;
            int64_t o = (int64_t)(tg.H - 1 - gy) * tg.W + px;
            for (int c = 0; c < 3; ++c) {
                float acc = tg.cbuf[base * 3 + c];
                for (int s = 1; s < tg.S; ++s) acc = acc + tg.cbuf[(base + s) * 3 + c];
                rgb[o * 3 + c] = float_to_unorm8(acc * inv);
            }
            /* depth resolve = sample 0 (GL_NEAREST blit, opengl.py:361-372) */
            uint16_t z = tg.zbuf[base];
            if (z16out) z16out[o] = z;
            if (depth) {
                /* get_depth_map (opengl.py:426-431) in float32, as numpy evaluates it */
                float d = (float)z / 65535.0f;
                float clip = (d - 0.5f) * 2.0f;
                float den = clip * (float)(100.0 - 0.04) - (float)(100.0 + 0.04);
                depth[o] = (float)(-2.0 * 100.0 * 0.04) / den;
            }
            if (primout)
                for (int s = 0; s < tg.S; ++s) primout[o * tg.S + s] = tg.ibuf[base + s];
        }
    target_free(&tg);
    return 0;
}

/* MiniWorldEnv.get_visible_ents (miniworld.py:1238-1333): the rooms are drawn untextured into the cleared obs frame
 * buffer (same camera as render_obs; Room._render :1291-1293), then, per entity in the order of self.entities, an
 * axis-aligned 0.2 m proxy box at ent.pos inside a GL_ANY_SAMPLES_PASSED query (:1296-1313): depth-tested (GL_LESS)
 * AND depth-written, so an earlier proxy can hide a later one; back faces are culled (:512).
 * sc->ents must be in self.entities order here (not draw order); vis[i] = query result != 0. */
int mwo_visible_ents(const mwo_scene *sc, uint8_t *vis)
{
    target tg;
    int rc = target_init(&tg, sc);
    if (rc) return rc;
    mwo_trilist tris = {0};
    int *first = (int *)malloc((size_t)(sc->n_ents + 1) * sizeof(int));
    rc = first ? mwo_geometry(sc, 1, &tris, first) : -2;
    if (rc) { target_free(&tg); mwo_trilist_free(&tris); free(first); return rc; }
    int64_t ns = (int64_t)tg.W * tg.H * tg.S;
    for (int64_t i = 0; i < ns; ++i) { tg.zbuf[i] = 65535; tg.ibuf[i] = -1; }
    memset(tg.cbuf, 0, ns * 3 * sizeof(float));
    int e = 0;
    for (int i = 0; i < sc->n_ents; ++i) vis[i] = 0;
    for (int i = 0; i < tris.n; ++i) {
        while (e < sc->n_ents && i >= first[e + 1]) ++e;
        setup_tri st;
        int passed = setup(&tg, &tris.tris[i], &st) ? draw_tri(sc, &tg, &st) : 0;
        if (i >= first[0] && e < sc->n_ents && i >= first[e] && passed) vis[e] = 1;
    }
    target_free(&tg);
    mwo_trilist_free(&tris);
    free(first);
    return 0;
}
