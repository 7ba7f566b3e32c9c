/* ORACLE — TEST INFRASTRUCTURE ONLY.  Vertex half of the pixel oracle (see mwo_gl.h).
 *
 * What the reference's GL calls do to a vertex, restated from Mesa 23.2.1 as it runs on llvmpipe:
 *   render_obs            miniworld/miniworld.py:1177-1221   gluPerspective, gluLookAt
 *   render_top_view       miniworld/miniworld.py:1088-1175   glOrtho, glLoadMatrixf
 *   _render_static        miniworld/miniworld.py:1019-1062   light, colour material, display list 1
 *   Room._render          miniworld/miniworld.py:401-434     GL_POLYGON floor / ceiling, GL_QUADS walls
 *   _render_world         miniworld/miniworld.py:1064-1086   draw order
 *   Box.render / drawBox  miniworld/entity.py:409-432, miniworld/opengl.py:460-503
 *   MeshEnt.render        miniworld/entity.py:150-161, miniworld/objmesh.py:280-292
 *   ImageFrame / TextFrame.render miniworld/entity.py:193-259, 303-383
 *   Agent.render          miniworld/entity.py:518-539
 *   get_visible_ents      miniworld/miniworld.py:1238-1333
 * Third-party arithmetic restated here (none of it is in /root/reference; pinned versions: Mesa 23.2.1,
 * libGLU 9.0 (SGI libutil/project.c), glibc 2.35):
 *   libGLU gluPerspective / gluLookAt; Mesa src/mesa/math/m_matrix.c (matmul4, translate, scale, rotate, ortho),
 *   src/mesa/main/light.c (light position -> eye space, _VP_inf_norm), src/mesa/main/ffvertex_prog.c (position by the
 *   MVP matrix, per-vertex lighting with GL_COLOR_MATERIAL), gallium/auxiliary/draw (primitive decomposition,
 *   draw_pipe_clip.c, viewport transform in the vertex shader and in the clipper).
 * Every formula below was checked bit for bit against the driver through GL feedback mode
 * (tools/gl_feedback_check.py): window coordinates, 1/w, colours and texture coordinates of every clipped triangle.
 *
 * Compile with -ffp-contract=off: fused multiply-adds are explicit fmaf() / fma().
 */
#include "mwo_gl.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ glibc sinf / cosf */

typedef struct { double sign[4], hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincosf_tab;
static const sincosf_tab SCT[2] = {
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2,
     0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
    {{1.0, -1.0, -1.0, 1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2,
     -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3,
     0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};

static inline float sincosf_poly(double x, double x2, const sincosf_tab *p, int n)
{
    if ((n & 1) == 0) {
        double x3 = x * x2, s1 = fma(x2, p->s3, p->s2), x7 = x3 * x2, s = fma(x3, p->s1, x);
        return (float)fma(x7, s1, s);
    }
    double x4 = x2 * x2, c2 = fma(x2, p->c4, p->c3), c1 = fma(x2, p->c1, p->c0), x6 = x4 * x2, c = fma(x4, p->c2, c1);
    return (float)fma(x6, c2, c);
}

static inline uint32_t abstop12(float x) { uint32_t u; memcpy(&u, &x, 4); return (u >> 20) & 0x7ff; }

static inline double reduce_fast(double x, const sincosf_tab *p, int *np)
{
    double r = x * p->hpi_inv;
    int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return fma(-(double)n, p->hpi, x);
}

/* valid for |x| < 120 (the reference's angles are a few turns at most); beyond that libm's own */
float mwo_sinf(float y)
{
    double x = y;
    int n;
    const sincosf_tab *p = &SCT[0];
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return y;
        return sincosf_poly(x, x * x, p, 0);
    }
    if (!(abstop12(y) < abstop12(120.0f))) return sinf(y);
    x = reduce_fast(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &SCT[1];
    return sincosf_poly(x * s, x * x, p, n);
}

float mwo_cosf(float y)
{
    double x = y;
    int n;
    const sincosf_tab *p = &SCT[0];
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f;
        return sincosf_poly(x, x * x, p, 1);
    }
    if (!(abstop12(y) < abstop12(120.0f))) return cosf(y);
    x = reduce_fast(x, p, &n);
    double s = p->sign[n & 3];
    if (n & 2) p = &SCT[1];
    return sincosf_poly(x * s, x * x, p, n ^ 1);
}

/* ------------------------------------------------------------------ Mesa m_matrix.c */

#define A(row, col) a[((col) << 2) + (row)]
#define B(row, col) b[((col) << 2) + (row)]
#define P(row, col) p[((col) << 2) + (row)]

static void mat_identity(mwo_mat4 *m)
{
    memset(m, 0, sizeof *m);
    m->m[0] = m->m[5] = m->m[10] = m->m[15] = 1.0f;
}

/* matmul4: every element ((ai0*b0j + ai1*b1j) + ai2*b2j) + ai3*b3j, separately rounded */
static void matmul4(float *p, const float *a, const float *b)
{
    float out[16];
    for (int i = 0; i < 4; ++i) {
        const float ai0 = A(i, 0), ai1 = A(i, 1), ai2 = A(i, 2), ai3 = A(i, 3);
        for (int j = 0; j < 4; ++j)
            out[(j << 2) + i] = ((ai0 * B(0, j) + ai1 * B(1, j)) + ai2 * B(2, j)) + ai3 * B(3, j);
    }
    memcpy(p, out, sizeof out);
}

/* _math_matrix_translate */
static void mat_translate(mwo_mat4 *mat, float x, float y, float z)
{
    float *m = mat->m;
    m[12] = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
    m[13] = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
    m[14] = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
    m[15] = ((m[3] * x + m[7] * y) + m[11] * z) + m[15];
}

/* _math_matrix_scale */
static void mat_scale(mwo_mat4 *mat, float x, float y, float z)
{
    float *m = mat->m;
    m[0] *= x; m[4] *= y; m[8] *= z;
    m[1] *= x; m[5] *= y; m[9] *= z;
    m[2] *= x; m[6] *= y; m[10] *= z;
    m[3] *= x; m[7] *= y; m[11] *= z;
}

/* _math_matrix_rotate(mat, angle, 0, 1, 0): the y-axis special case builds c, s into an identity and multiplies */
static void mat_rotate_y(mwo_mat4 *mat, float angle)
{
    float arg = (float)((double)angle * 3.14159265358979323846 / 180.0);
    float s = mwo_sinf(arg), c = mwo_cosf(arg);
    mwo_mat4 r;
    mat_identity(&r);
    r.m[0] = c; r.m[10] = c;            /* M(0,0), M(2,2) */
    r.m[8] = s; r.m[2] = -s;            /* M(0,2) = s, M(2,0) = -s */
    matmul4(mat->m, mat->m, r.m);
}

/* ------------------------------------------------------------------ per-frame GL state */

typedef struct {
    mwo_mat4 proj, view;        /* GL_PROJECTION, GL_MODELVIEW as render_obs / render_top_view leave them */
    float vp_scale[3], vp_trans[3];
    float light_eye[3];         /* EyePosition of GL_LIGHT0 (w = 0): modelview * position at glLightfv time */
    float l_amb[3], l_dif[3];   /* GL_AMBIENT, GL_DIFFUSE of the light                                  */
    float sky[3];
    int lighting;               /* 0 inside get_visible_ents (no display list => state of the last frame) */
} glstate;

static void build_view(const mwo_scene *sc, glstate *st)
{
    mat_identity(&st->proj);
    mat_identity(&st->view);
    if (sc->view == 1) {
        /* render_top_view (miniworld.py:1108-1160), python doubles */
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
        /* glOrtho(min_x, max_x, -max_z, -min_z, -100, 100) -> _math_matrix_ortho on floats */
        float l = (float)min_x, r = (float)max_x, b = (float)-max_z, t = (float)-min_z, n = -100.0f, f = 100.0f;
        mwo_mat4 o;
        mat_identity(&o);
        o.m[0] = 2.0f / (r - l);  o.m[12] = -(r + l) / (r - l);
        o.m[5] = 2.0f / (t - b);  o.m[13] = -(t + b) / (t - b);
        o.m[10] = -2.0f / (f - n); o.m[14] = -(f + n) / (f - n);
        matmul4(st->proj.m, st->proj.m, o.m);
        /* glLoadMatrixf: (x, y, z) -> (x, -z, y) */
        static const float M[16] = {1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1};
        memcpy(st->view.m, M, sizeof M);
        return;
    }
    /* Agent.cam_pos / cam_dir via gen_rot_matrix (math.py:11-27, entity.py:476-503), evaluated in double exactly
     * as numpy does (terms multiplied by exact zeros dropped) */
    double sh, ch, sp, cp;
    mwo_sincos(sc->agent_dir / 2.0, &sh, &ch);
    double a = ch, c = -1.0 * sh;
    double ry00 = a * a - c * c, ry02 = 2.0 * (a * c), ry11 = a * a + c * c;
    double pitch = sc->cam_pitch * 3.14159265358979323846 / 180.0;
    mwo_sincos(pitch / 2.0, &sp, &cp);
    double az = cp, dz = -1.0 * sp;
    double rz00 = az * az - dz * dz, rz01 = 2.0 * (0.0 - az * dz);
    double eye[3], dir[3];
    eye[0] = sc->agent_pos[0] + sc->cam_fwd_disp * ry00;
    eye[1] = sc->agent_pos[1] + sc->cam_height * ry11;
    eye[2] = sc->agent_pos[2] + sc->cam_fwd_disp * ry02;
    dir[0] = rz00 * ry00; dir[1] = rz01 * ry11; dir[2] = rz00 * ry02;
    double at[3] = {eye[0] + dir[0], eye[1] + dir[1], eye[2] + dir[2]};

    /* gluPerspective(fovy, W / H, 0.04, 100): doubles, glMultMatrixd rounds the matrix to float */
    double radians = sc->cam_fov_y / 2 * 3.14159265358979323846 / 180;
    double sine, cosine;
    mwo_sincos(radians, &sine, &cosine);
    double cot = cosine / sine, aspect = (double)sc->width / (double)sc->height, zn = 0.04, zf = 100.0, dzz = zf - zn;
    memset(&st->proj, 0, sizeof st->proj);
    st->proj.m[0] = (float)(cot / aspect);
    st->proj.m[5] = (float)cot;
    st->proj.m[10] = (float)(-(zf + zn) / dzz);
    st->proj.m[11] = -1.0f;
    st->proj.m[14] = (float)(-2 * zn * zf / dzz);

    /* gluLookAt: float vectors (libutil/project.c), glMultMatrixf, glTranslated(-eye) */
    float fw[3] = {(float)(at[0] - eye[0]), (float)(at[1] - eye[1]), (float)(at[2] - eye[2])};
    float r = (float)sqrt((double)((fw[0] * fw[0] + fw[1] * fw[1]) + fw[2] * fw[2]));
    if (r != 0.0f) { fw[0] /= r; fw[1] /= r; fw[2] /= r; }
    /* side = forward x (0, 1, 0) */
    float side[3] = {fw[1] * 0.0f - fw[2] * 1.0f, fw[2] * 0.0f - fw[0] * 0.0f, fw[0] * 1.0f - fw[1] * 0.0f};
    r = (float)sqrt((double)((side[0] * side[0] + side[1] * side[1]) + side[2] * side[2]));
    if (r != 0.0f) { side[0] /= r; side[1] /= r; side[2] /= r; }
    float up[3] = {side[1] * fw[2] - side[2] * fw[1], side[2] * fw[0] - side[0] * fw[2], side[0] * fw[1] - side[1] * fw[0]};
    float *m = st->view.m;
    m[0] = side[0]; m[4] = side[1]; m[8] = side[2];
    m[1] = up[0];   m[5] = up[1];   m[9] = up[2];
    m[2] = -fw[0];  m[6] = -fw[1];  m[10] = -fw[2];
    mat_translate(&st->view, (float)-eye[0], (float)-eye[1], (float)-eye[2]);
}

static void build_state(const mwo_scene *sc, glstate *st)
{
    build_view(sc, st);
    /* glViewport(0, 0, W, H), depth range [0, 1]; an FBO is not flipped */
    st->vp_scale[0] = (float)sc->width * 0.5f;  st->vp_trans[0] = (float)sc->width * 0.5f;
    st->vp_scale[1] = (float)sc->height * 0.5f; st->vp_trans[1] = (float)sc->height * 0.5f;
    st->vp_scale[2] = 0.5f; st->vp_trans[2] = 0.5f;
    /* glLightfv(GL_LIGHT0, GL_POSITION, (GLfloat*4)(*light_pos + [1])) (miniworld.py:1031): ndarray + [1] adds 1 to
     * every component and leaves w = 0: a directional light.  Executed by glCallList under the camera's modelview:
     * EyePosition = M * p (light.c TRANSFORM_POINT), _VP_inf_norm = normalised (1 / sqrtf) */
    float lp[4] = {(float)(sc->light_pos[0] + 1.0), (float)(sc->light_pos[1] + 1.0), (float)(sc->light_pos[2] + 1.0), 0.0f};
    const float *M = st->view.m;
    float e[3];
    for (int i = 0; i < 3; ++i) e[i] = ((M[i] * lp[0] + M[4 + i] * lp[1]) + M[8 + i] * lp[2]) + M[12 + i] * lp[3];
    for (int i = 0; i < 3; ++i) {
        st->light_eye[i] = e[i];
        st->l_amb[i] = (float)sc->light_ambient[i];
        st->l_dif[i] = (float)sc->light_color[i];
        st->sky[i] = (float)sc->sky[i];
    }
    st->lighting = 1;
}

/* ------------------------------------------------------------------ vertex program + draw module */

/* Mesa keeps geometry flags per matrix (m_matrix.c); they select how the inverse is computed */
enum { MF_ROTATION = 1, MF_TRANSLATION = 2, MF_UNIFORM_SCALE = 4, MF_GENERAL = 8 };

typedef struct {
    mwo_mat4 mv, mvp;
    unsigned flags;
    float light[3];             /* STATE_LIGHT_POSITION_NORMALIZED in OBJECT space (see make_xform) */
    float nscale;               /* STATE_NORMAL_SCALE: _ModelViewInvScale                            */
} xform;

#define SQf(x) ((x) * (x))

/* analyse_from_scratch for a matrix of the MATRIX_3D kind (last row 0 0 0 1) */
static unsigned analyse_from_scratch(const mwo_mat4 *mat)
{
    const float *m = mat->m;
    unsigned flags = 0;
    if (m[12] != 0.0f || m[13] != 0.0f || m[14] != 0.0f) flags |= MF_TRANSLATION;
    float c1 = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2];
    float c2 = (m[4] * m[4] + m[5] * m[5]) + m[6] * m[6];
    float c3 = (m[8] * m[8] + m[9] * m[9]) + m[10] * m[10];
    float d1 = (m[0] * m[4] + m[1] * m[5]) + m[2] * m[6];
    if (SQf(c1 - c2) < SQf(1e-6f) && SQf(c1 - c3) < SQf(1e-6f)) {
        if (SQf(c1 - 1.0f) > SQf(1e-6f)) flags |= MF_UNIFORM_SCALE;
    } else {
        flags |= MF_GENERAL;
    }
    if (SQf(d1) < SQf(1e-6f)) {
        float cp[3] = {m[1] * m[6] - m[2] * m[5], m[2] * m[4] - m[0] * m[6], m[0] * m[5] - m[1] * m[4]};
        cp[0] -= m[8]; cp[1] -= m[9]; cp[2] -= m[10];
        if ((cp[0] * cp[0] + cp[1] * cp[1]) + cp[2] * cp[2] < SQf(1e-6f)) flags |= MF_ROTATION;
        else flags |= MF_GENERAL;
    } else {
        flags |= MF_GENERAL;
    }
    return flags;
}

/* upper-left 3x3 of the inverse, column-major 4x4 layout (m_matrix.c invert_matrix_3d / _general) */
static void invert3(const mwo_mat4 *mv, unsigned flags, float out[16])
{
    const float *in = mv->m;
#define MAT(m, r, c) (m)[(c) * 4 + (r)]
    memset(out, 0, 16 * sizeof(float));
    if (flags & MF_GENERAL) {
        float pos = 0.0f, neg = 0.0f, t;
        t = MAT(in, 0, 0) * MAT(in, 1, 1) * MAT(in, 2, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = MAT(in, 1, 0) * MAT(in, 2, 1) * MAT(in, 0, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = MAT(in, 2, 0) * MAT(in, 0, 1) * MAT(in, 1, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = -MAT(in, 2, 0) * MAT(in, 1, 1) * MAT(in, 0, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = -MAT(in, 1, 0) * MAT(in, 0, 1) * MAT(in, 2, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = -MAT(in, 0, 0) * MAT(in, 2, 1) * MAT(in, 1, 2); if (t >= 0.0f) pos += t; else neg += t;
        float det = pos + neg;
        if (fabsf(det) < 1e-25f) return;
        det = 1.0f / det;
        MAT(out, 0, 0) = (MAT(in, 1, 1) * MAT(in, 2, 2) - MAT(in, 2, 1) * MAT(in, 1, 2)) * det;
        MAT(out, 0, 1) = -(MAT(in, 0, 1) * MAT(in, 2, 2) - MAT(in, 2, 1) * MAT(in, 0, 2)) * det;
        MAT(out, 0, 2) = (MAT(in, 0, 1) * MAT(in, 1, 2) - MAT(in, 1, 1) * MAT(in, 0, 2)) * det;
        MAT(out, 1, 0) = -(MAT(in, 1, 0) * MAT(in, 2, 2) - MAT(in, 2, 0) * MAT(in, 1, 2)) * det;
        MAT(out, 1, 1) = (MAT(in, 0, 0) * MAT(in, 2, 2) - MAT(in, 2, 0) * MAT(in, 0, 2)) * det;
        MAT(out, 1, 2) = -(MAT(in, 0, 0) * MAT(in, 1, 2) - MAT(in, 1, 0) * MAT(in, 0, 2)) * det;
        MAT(out, 2, 0) = (MAT(in, 1, 0) * MAT(in, 2, 1) - MAT(in, 2, 0) * MAT(in, 1, 1)) * det;
        MAT(out, 2, 1) = -(MAT(in, 0, 0) * MAT(in, 2, 1) - MAT(in, 2, 0) * MAT(in, 0, 1)) * det;
        MAT(out, 2, 2) = (MAT(in, 0, 0) * MAT(in, 1, 1) - MAT(in, 1, 0) * MAT(in, 0, 1)) * det;
        return;
    }
    float scale = 1.0f;
    if (flags & MF_UNIFORM_SCALE) {
        scale = (MAT(in, 0, 0) * MAT(in, 0, 0) + MAT(in, 0, 1) * MAT(in, 0, 1)) + MAT(in, 0, 2) * MAT(in, 0, 2);
        if (scale == 0.0f) return;
        scale = 1.0f / scale;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) MAT(out, r, c) = scale * MAT(in, c, r);
    } else {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) MAT(out, r, c) = MAT(in, c, r);
    }
#undef MAT
}

/* With one directional light, no local viewer and no eye-space texgen Mesa lights in OBJECT space
 * (ctx->_NeedEyeCoords false, light.c compute_light_positions): _Position = inverse(modelview) * EyePosition, the
 * vertex program dots it — normalised (prog_statevars.c STATE_LIGHT_POSITION_NORMALIZED: 1 / sqrtf) — with the raw
 * glNormal3f.  A uniformly scaled mesh is therefore lit as if unscaled. */
static void make_xform(const glstate *st, const mwo_mat4 *mv, unsigned flags, xform *x)
{
    x->mv = *mv;
    x->flags = flags;
    matmul4(x->mvp.m, st->proj.m, mv->m);
    float inv[16];
    invert3(mv, flags, inv);
    const float *e = st->light_eye;
    float q[3];
    for (int i = 0; i < 3; ++i) q[i] = ((inv[i] * e[0] + inv[4 + i] * e[1]) + inv[8 + i] * e[2]) + inv[12 + i] * 0.0f;
    float len = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    if (len != 0.0f) {
        len = 1.0f / sqrtf(len);
        q[0] *= len; q[1] *= len; q[2] *= len;
    }
    memcpy(x->light, q, sizeof q);
    /* update_modelview_scale (light.c): a modelview that is not length preserving rescales the normals in the vertex
     * program (ffvertex_prog.c get_transformed_normal: MUL normal, STATE_NORMAL_SCALE) by the length of the
     * inverse's third row — 1 / scale for MeshEnt's glScalef, what the inverse transpose does to a normal */
    x->nscale = 1.0f;
    if (flags & (MF_UNIFORM_SCALE | MF_GENERAL)) {
        float f = (inv[2] * inv[2] + inv[6] * inv[6]) + inv[10] * inv[10];
        if (f < 1e-12f) f = 1.0f;
        x->nscale = sqrtf(f);
    }
}

/* one vertex through the fixed-function vertex program and the shader's viewport code */
static void shade_vertex(const glstate *st, const xform *x, const float p[3], const float n[3], const float c[3],
                         const float uv[2], mwo_vert *v)
{
    const float *m = x->mvp.m;
    /* position: MUL, MAD, MAD, MAD by the columns of the MVP matrix, unfused */
    for (int i = 0; i < 4; ++i) v->clip[i] = ((p[0] * m[i] + p[1] * m[4 + i]) + p[2] * m[8 + i]) + m[12 + i];
    if (st->lighting) {
        float ns[3] = {n[0] * x->nscale, n[1] * x->nscale, n[2] * x->nscale};
        float dot = (ns[0] * x->light[0] + ns[1] * x->light[1]) + ns[2] * x->light[2];
        float d = dot > 0.0f ? dot : 0.0f;
        for (int i = 0; i < 3; ++i) {
            float scene = 0.2f * c[i];                       /* GL_LIGHT_MODEL_AMBIENT * material ambient */
            float acc = st->l_amb[i] * c[i] + scene;
            acc = d * (st->l_dif[i] * c[i]) + acc;
            v->col[i] = acc < 0.0f ? 0.0f : (acc > 1.0f ? 1.0f : acc);
        }
    } else {
        for (int i = 0; i < 3; ++i) v->col[i] = c[i];
    }
    v->col[3] = 1.0f;
    v->st[0] = uv ? uv[0] : 0.0f;
    v->st[1] = uv ? uv[1] : 0.0f;
    /* clip test against the frustum (draw_llvm.c generate_clipmask) */
    unsigned mask = 0;
    float w = v->clip[3];
    if (v->clip[0] > w) mask |= 1u << 0;
    if (v->clip[0] + w < 0.0f) mask |= 1u << 1;
    if (v->clip[1] > w) mask |= 1u << 2;
    if (v->clip[1] + w < 0.0f) mask |= 1u << 3;
    if (v->clip[2] + w < 0.0f) mask |= 1u << 4;     /* plane 4: (0, 0, 1, 1) */
    if (v->clip[2] > w) mask |= 1u << 5;            /* plane 5: (0, 0, -1, 1) */
    v->clipmask = mask;
    /* viewport (draw_llvm.c generate_viewport): 1/w by division, x * (1/w), then one fused multiply-add */
    float oow = 1.0f / w;
    v->win[0] = fmaf(v->clip[0] * oow, st->vp_scale[0], st->vp_trans[0]);
    v->win[1] = fmaf(v->clip[1] * oow, st->vp_scale[1], st->vp_trans[1]);
    v->win[2] = fmaf(v->clip[2] * oow, st->vp_scale[2], st->vp_trans[2]);
    v->win[3] = oow;
}

static int push_tri(mwo_trilist *l, const mwo_vert *a, const mwo_vert *b, const mwo_vert *c, int tex, int draw)
{
    if (l->n == l->cap) {
        int cap = l->cap ? l->cap * 2 : 256;
        mwo_tri *t = (mwo_tri *)realloc(l->tris, (size_t)cap * sizeof(mwo_tri));
        if (!t) return -1;
        l->tris = t; l->cap = cap;
    }
    mwo_tri *t = &l->tris[l->n++];
    t->v[0] = *a; t->v[1] = *b; t->v[2] = *c;
    t->tex = tex; t->draw = draw;
    return 0;
}

void mwo_trilist_free(mwo_trilist *l) { free(l->tris); l->tris = NULL; l->n = l->cap = 0; }

static const float CLIP_PLANES[6][4] = {
    {-1, 0, 0, 1}, {1, 0, 0, 1}, {0, -1, 0, 1}, {0, 1, 0, 1}, {0, 0, 1, 1}, {0, 0, -1, 1}};

/* draw_pipe_clip.c interp(): dst = out + t * (in - out) on clip position and attributes, then the projective
 * divide and the viewport transform in plain C (two roundings, no fused multiply-add) */
static void clip_interp(const glstate *st, mwo_vert *dst, float t, const mwo_vert *out, const mwo_vert *in)
{
    for (int i = 0; i < 4; ++i) dst->clip[i] = out->clip[i] + t * (in->clip[i] - out->clip[i]);
    for (int i = 0; i < 4; ++i) dst->col[i] = out->col[i] + t * (in->col[i] - out->col[i]);
    for (int i = 0; i < 2; ++i) dst->st[i] = out->st[i] + t * (in->st[i] - out->st[i]);
    float oow = 1.0f / dst->clip[3];
    dst->win[0] = dst->clip[0] * oow * st->vp_scale[0] + st->vp_trans[0];
    dst->win[1] = dst->clip[1] * oow * st->vp_scale[1] + st->vp_trans[1];
    dst->win[2] = dst->clip[2] * oow * st->vp_scale[2] + st->vp_trans[2];
    dst->win[3] = oow;
    dst->clipmask = 0;
}

static inline int different_signs(float a, float b) { return !((a >= 0.0f) == (b >= 0.0f)) ; }

/* clip_tri + do_clip_tri + emit_poly */
static int clip_and_emit(const glstate *st, mwo_trilist *l, const mwo_vert *v0, const mwo_vert *v1, const mwo_vert *v2,
                         int tex, int draw)
{
    unsigned clipmask = v0->clipmask | v1->clipmask | v2->clipmask;
    if (clipmask == 0) return push_tri(l, v0, v1, v2, tex, draw);
    if (v0->clipmask & v1->clipmask & v2->clipmask) return 0;
    mwo_vert store[32];
    int nstore = 0;
    const mwo_vert *a[16], *b[16];
    const mwo_vert **inlist = a, **outlist = b;
    int n = 3;
    inlist[0] = v0; inlist[1] = v1; inlist[2] = v2;
    while (clipmask && n >= 3) {
        int plane_idx = __builtin_ffs((int)clipmask) - 1;
        const float *plane = CLIP_PLANES[plane_idx];
        clipmask &= ~(1u << plane_idx);
        const mwo_vert *vert_prev = inlist[0];
        float dp_prev = ((vert_prev->clip[0] * plane[0] + vert_prev->clip[1] * plane[1]) + vert_prev->clip[2] * plane[2]) +
                        vert_prev->clip[3] * plane[3];
        int outcount = 0;
        inlist[n] = inlist[0];
        for (int i = 1; i <= n; ++i) {
            const mwo_vert *vert = inlist[i];
            float dp = ((vert->clip[0] * plane[0] + vert->clip[1] * plane[1]) + vert->clip[2] * plane[2]) + vert->clip[3] * plane[3];
            if (isnan(dp) || isinf(dp)) return 0;
            if (dp_prev >= 0.0f) outlist[outcount++] = vert_prev;
            if (different_signs(dp, dp_prev)) {
                mwo_vert *nv = &store[nstore++];
                outlist[outcount++] = nv;
                /* the new vertex is interpolated from the endpoint that is closer to the plane (smaller |dp|), whichever
                 * way the edge is traversed: both triangles that share an edge get the same vertex */
                if (fabsf(dp) < fabsf(dp_prev)) {
                    float t = dp / (dp - dp_prev);
                    clip_interp(st, nv, t, vert, vert_prev);
                } else {
                    float t = dp_prev / (dp_prev - dp);
                    clip_interp(st, nv, t, vert_prev, vert);
                }
            }
            vert_prev = vert;
            dp_prev = dp;
        }
        const mwo_vert **tmp = inlist; inlist = outlist; outlist = tmp;
        n = outcount;
    }
    /* emit_poly: a fan that keeps the provoking (last) vertex in v[2] */
    if (n >= 3)
        for (int i = 2; i < n; ++i)
            if (push_tri(l, inlist[i - 1], inlist[i], inlist[0], tex, draw)) return -1;
    return 0;
}

/* GL primitive -> triangles.  A polygon is always the fan (1,2,0) (2,3,0).  A quad is split in one of two ways:
 *   (0,1,3) (1,2,3)   inside display list 1 (vbo_save converts the list to indexed triangles: rooms, frames, static
 *                     entities), and for an immediate-mode draw call (one glBegin / glEnd: a drawBox) of which ANY vertex
 *                     lies outside the view frustum — the whole call then runs through the draw module's pipeline
 *                     (draw_pipe.c decomposes, clips);
 *   (0,1,2) (0,2,3)   for an immediate-mode draw call without a clipped vertex (llvmpipe's own vertex-buffer path,
 *                     lp_setup_vbuf.c).
 * The union is the same quad; the halves differ, and each half has its own plane coefficients. */
enum { PRIM_TRIANGLES, PRIM_QUADS, PRIM_POLYGON, PRIM_IMMEDIATE = 8 };

static int draw_prim(const glstate *st, const xform *x, mwo_trilist *l, int mode, int nv, const float (*pos)[3],
                     const float (*nrm)[3], const float (*col)[3], const float (*uv)[2], int tex, int draw)
{
    mwo_vert v[4];
    if (nv > 4) return -3;
    for (int k = 0; k < nv; ++k) shade_vertex(st, x, pos[k], nrm[k], col[k], uv ? uv[k] : NULL, &v[k]);
    if ((mode & 7) == PRIM_TRIANGLES) return clip_and_emit(st, l, &v[0], &v[1], &v[2], tex, draw);
    if (nv == 3) return clip_and_emit(st, l, &v[1], &v[2], &v[0], tex, draw);       /* GL_POLYGON of three vertices */
    if ((mode & 7) == PRIM_QUADS && (mode & PRIM_IMMEDIATE)) {
        if (clip_and_emit(st, l, &v[0], &v[1], &v[2], tex, draw)) return -1;
        return clip_and_emit(st, l, &v[0], &v[2], &v[3], tex, draw);
    }
    if (mode == PRIM_QUADS) {
        if (clip_and_emit(st, l, &v[0], &v[1], &v[3], tex, draw)) return -1;
        return clip_and_emit(st, l, &v[1], &v[2], &v[3], tex, draw);
    }
    /* GL_POLYGON: (1, 2, 0), (2, 3, 0) */
    if (clip_and_emit(st, l, &v[1], &v[2], &v[0], tex, draw)) return -1;
    return clip_and_emit(st, l, &v[2], &v[3], &v[0], tex, draw);
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

static int draw_box(const glstate *st, const xform *x, mwo_trilist *l, const float lo[3], const float hi[3],
                    const float col[3], int *draw, int immediate)
{
    if (immediate) {
        /* one glBegin / glEnd = one draw call: a clipped vertex anywhere sends all six faces through the pipeline */
        for (int f = 0; f < 6 && immediate; ++f)
            for (int k = 0; k < 4; ++k) {
                float p[3] = {BOXV[f][k][0] ? hi[0] : lo[0], BOXV[f][k][1] ? hi[1] : lo[1], BOXV[f][k][2] ? hi[2] : lo[2]};
                mwo_vert tmp;
                shade_vertex(st, x, p, BOXN[f], col, NULL, &tmp);
                if (tmp.clipmask) immediate = 0;
            }
    }
    for (int f = 0; f < 6; ++f, ++*draw) {
        float v[4][3], n[4][3], c[4][3];
        for (int k = 0; k < 4; ++k) {
            v[k][0] = BOXV[f][k][0] ? hi[0] : lo[0];
            v[k][1] = BOXV[f][k][1] ? hi[1] : lo[1];
            v[k][2] = BOXV[f][k][2] ? hi[2] : lo[2];
            memcpy(n[k], BOXN[f], sizeof n[k]);
            memcpy(c[k], col, sizeof c[k]);
        }
        int rc = draw_prim(st, x, l, PRIM_QUADS | (immediate ? PRIM_IMMEDIATE : 0), 4, (const float (*)[3])v,
                           (const float (*)[3])n, (const float (*)[3])c, NULL, -1, *draw);
        if (rc) return rc;
    }
    return 0;
}

int mwo_geometry(const mwo_scene *sc, int proxies, mwo_trilist *out, int *ent_first)
{
    glstate st;
    build_state(sc, &st);
    if (proxies) st.lighting = 1;    /* GL_LIGHTING stays enabled from the last frame's display list */
    xform cam;
    unsigned cam_flags = analyse_from_scratch(&st.view);
    make_xform(&st, &st.view, cam_flags, &cam);
    int draw = 0;
    /* the GL "current normal": what the last glNormal3f — immediate mode, or the end of the display list — left */
    float stale_n[3] = {0.0f, 1.0f, 0.0f};
    static const float white[3] = {1.0f, 1.0f, 1.0f};
    /* display list 1: rooms (miniworld.py:1053-1055), then static entities' quads */
    for (int i = 0; i < sc->n_polys; ++i, ++draw) {
        const mwo_poly *q = &sc->polys[i];
        int nv = q->nv & 0xFF;
        if (proxies && (q->nv & MWO_POLY_ENTITY)) continue;      /* only room._render() (:1291-1293) */
        float n[4][3], c[4][3];
        for (int k = 0; k < nv; ++k) { memcpy(n[k], q->n, sizeof n[k]); memcpy(c[k], proxies ? white : q->rgb, sizeof c[k]); }
        const xform *x = &cam;
        xform ex;
        if (q->nv & MWO_POLY_XF) {
            /* glPushMatrix; glTranslatef(*pos); glRotatef(dir * (180 / pi), 0, 1, 0) (entity.py:205-207) */
            mwo_mat4 mv = st.view;
            mat_translate(&mv, q->xf[0], q->xf[1], q->xf[2]);
            mat_rotate_y(&mv, q->xf[3]);
            make_xform(&st, &mv, cam_flags | MF_TRANSLATION | MF_ROTATION, &ex);
            x = &ex;
        }
        /* floor and ceiling are GL_POLYGON, walls and frame quads GL_QUADS; a 3-vertex polygon is one triangle */
        int mode = (q->nv & MWO_POLY_QUAD) ? PRIM_QUADS : PRIM_POLYGON;
        int rc = draw_prim(&st, x, out, mode, nv, q->v, (const float (*)[3])n, (const float (*)[3])c,
                           (proxies || q->tex < 0) ? NULL : q->uv, proxies ? -1 : q->tex, draw);
        if (rc) return rc;
        memcpy(stale_n, q->n, sizeof stale_n);
    }
    for (int e = 0; e < sc->n_ents; ++e) {
        const mwo_ent *en = &sc->ents[e];
        if (ent_first) ent_first[e] = out->n;
        if (en->kind == MWO_ENT_NONE) continue;
        if (proxies) {
            /* drawBox arguments are python doubles and reach GL through glVertex3f (miniworld.py:1303-1311) */
            float lo[3] = {(float)(en->pos[0] - 0.1), (float)en->pos[1], (float)(en->pos[2] - 0.1)};
            float hi[3] = {(float)(en->pos[0] + 0.1), (float)(en->pos[1] + 0.2), (float)(en->pos[2] + 0.1)};
            int rc = draw_box(&st, &cam, out, lo, hi, white, &draw, 1);
            if (rc) return rc;
            continue;
        }
        if (en->kind == MWO_ENT_BOX) {
            /* Box.render (entity.py:409-432) */
            mwo_mat4 mv = st.view;
            mat_translate(&mv, (float)en->pos[0], (float)en->pos[1], (float)en->pos[2]);
            mat_rotate_y(&mv, (float)(en->dir * (180 / 3.14159265358979323846)));
            xform ex;
            make_xform(&st, &mv, cam_flags | MF_TRANSLATION | MF_ROTATION, &ex);
            float lo[3] = {(float)(-en->size[0] / 2), 0.0f, (float)(-en->size[2] / 2)};
            float hi[3] = {(float)(en->size[0] / 2), (float)en->size[1], (float)(en->size[2] / 2)};
            float col[3] = {(float)en->color[0], (float)en->color[1], (float)en->color[2]};
            int rc = draw_box(&st, &ex, out, lo, hi, col, &draw, !en->is_static);
            if (rc) return rc;
            stale_n[0] = 0.0f; stale_n[1] = -1.0f; stale_n[2] = 0.0f;
        } else if (en->kind == MWO_ENT_MESH) {
            /* MeshEnt.render (entity.py:150-161): translate, scale, rotate; ObjMesh.render draws vertex lists */
            const mwo_mesh *m = &sc->meshes[en->mesh];
            mwo_mat4 mv = st.view;
            float sc_ = (float)en->scale;
            mat_translate(&mv, (float)en->pos[0], (float)en->pos[1], (float)en->pos[2]);
            mat_scale(&mv, sc_, sc_, sc_);
            mat_rotate_y(&mv, (float)(en->dir * 180 / 3.14159265358979323846));
            xform ex;
            make_xform(&st, &mv, cam_flags | MF_TRANSLATION | MF_ROTATION | MF_UNIFORM_SCALE, &ex);
            for (int t = 0; t < m->ntris; ++t, ++draw) {
                const float (*p)[3] = (const float (*)[3])&m->pos[(size_t)t * 9];
                const float (*n)[3] = (const float (*)[3])&m->nrm[(size_t)t * 9];
                const float (*c)[3] = (const float (*)[3])&m->rgb[(size_t)t * 9];
                const float (*uv)[2] = (const float (*)[2])&m->uv[(size_t)t * 6];
                int rc = draw_prim(&st, &ex, out, PRIM_TRIANGLES, 3, p, n, c, m->tex >= 0 ? uv : NULL, m->tex, draw);
                if (rc) return rc;
            }
            /* glDrawArrays with a normal array leaves the current normal alone */
        }
    }
    if (ent_first) ent_first[sc->n_ents] = out->n;
    if (sc->render_agent && !proxies) {
        /* Agent.render (entity.py:518-539): a red triangle at the top of the agent's cylinder, drawn without any
         * glNormal3f => lit with the normal the previous draw left current */
        double sd, cd;
        mwo_sincos(sc->agent_dir, &sd, &cd);
        double rad = sc->agent_radius, hgt = sc->agent_height;
        double p[3] = {sc->agent_pos[0] + 0 * hgt, sc->agent_pos[1] + 1 * hgt, sc->agent_pos[2] + 0 * hgt};
        double dv[3] = {cd * rad, 0 * rad, -sd * rad}, rv[3] = {sd * rad, 0 * rad, cd * rad};
        float v[3][3], n[3][3], c[3][3];
        for (int i = 0; i < 3; ++i) {
            v[0][i] = (float)(p[i] + dv[i]);
            v[2][i] = (float)(p[i] + 0.75 * (rv[i] - dv[i]));
            v[1][i] = (float)(p[i] + 0.75 * (-rv[i] - dv[i]));
        }
        for (int k = 0; k < 3; ++k) { memcpy(n[k], stale_n, sizeof n[k]); c[k][0] = 1.0f; c[k][1] = 0.0f; c[k][2] = 0.0f; }
        int rc = draw_prim(&st, &cam, out, PRIM_TRIANGLES, 3, (const float (*)[3])v, (const float (*)[3])n,
                           (const float (*)[3])c, NULL, -1, draw);
        if (rc) return rc;
    }
    return 0;
}

/* debug / test hook: the triangle stream as flat floats, 3 x (win[4], col[4], st[2]) + tex + draw per triangle */
int mwo_debug_geometry(const mwo_scene *sc, int proxies, float *buf, int max_tris)
{
    mwo_trilist l = {0};
    int rc = mwo_geometry(sc, proxies, &l, NULL);
    if (rc) { mwo_trilist_free(&l); return rc; }
    int n = l.n < max_tris ? l.n : max_tris;
    for (int i = 0; i < n; ++i) {
        float *o = buf + (size_t)i * 32;
        for (int k = 0; k < 3; ++k) {
            memcpy(o + k * 10, l.tris[i].v[k].win, 16);
            memcpy(o + k * 10 + 4, l.tris[i].v[k].col, 16);
            memcpy(o + k * 10 + 8, l.tris[i].v[k].st, 8);
        }
        o[30] = (float)l.tris[i].tex;
        o[31] = (float)l.tris[i].draw;
    }
    int total = l.n;
    mwo_trilist_free(&l);
    return total;
}
