// The vertex half of the frame, as the reference's GL driver computes it (host + device).
//
// What MiniWorldEnv.render_obs / render_top_view / _render_static / Room._render / Box.render / MeshEnt.render /
// ImageFrame.render / Agent.render (miniworld.py:401-434, 1019-1221; entity.py:150-161, 193-259, 409-432, 518-539;
// opengl.py:460-503) hand to OpenGL, followed through the fixed-function vertex stage exactly as Mesa 23.2.1 / llvmpipe —
// the driver the reference's frames were captured on (tests/golden/gl_*.npz) — evaluates it:
//   libGLU gluPerspective (double, rounded by glMultMatrixd) and gluLookAt (float vectors, glMultMatrixf, glTranslated);
//   Mesa's matrix stack (m_matrix.c: matmul4, translate, scale, rotate with glibc's sinf / cosf, the geometry flags that
//   select how the inverse is taken); the light in OBJECT space (light.c compute_light_positions, ffvertex_prog.c:
//   one directional light, colour material, normals rescaled by the inverse's third row); position by the MVP matrix;
//   the draw module's primitive decomposition, frustum clipping (draw_pipe_clip.c: new vertices interpolated from the
//   endpoint nearer to the plane) and the two viewport transforms (fused in the vertex shader, unfused in the clipper).
// DESIGN.md section 3 lists the rules; every one was measured against the driver (GL feedback mode, float frame buffers).
// Everything here is plain IEEE float / double arithmetic in a fixed order: compile with -ffp-contract=off; fused
// multiply-adds are explicit.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#ifdef __HIPCC__
#define MW_HD __host__ __device__ inline
#else
#define MW_HD inline
#endif

namespace mwgl {

// ---------------------------------------------------------------- glibc 2.35 sinf / cosf (FMA multiarch variant)
// Mesa's _math_matrix_rotate calls sinf / cosf: restated (sysdeps/ieee754/flt-32/s_sinf.c, sincosf.h) so that the
// device produces glibc's bits; checked against libm over all floats of |x| < 120 (tests/test_host_logic_cpu.py).
MW_HD float sincosf_poly(double x, double x2, int tab, int n)
{
    // tab 0 / 1: the table and its negated-cosine twin (quadrants 2, 3)
    const double c0 = tab ? -0x1p0 : 0x1p0, c1 = tab ? 0x1.ffffffd0c621cp-2 : -0x1.ffffffd0c621cp-2;
    const double c2 = tab ? -0x1.55553e1068f19p-5 : 0x1.55553e1068f19p-5, c3 = tab ? 0x1.6c087e89a359dp-10 : -0x1.6c087e89a359dp-10;
    const double c4 = tab ? -0x1.99343027bf8c3p-16 : 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2, t1 = fma(x2, s3, s2), x7 = x3 * x2, s = fma(x3, s1, x);
        return (float)fma(x7, t1, s);
    }
    const double x4 = x2 * x2, t2 = fma(x2, c4, c3), t1 = fma(x2, c1, c0), x6 = x4 * x2, c = fma(x4, c2, t1);
    return (float)fma(x6, t2, c);
}

MW_HD uint32_t f2u(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
MW_HD float u2f(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }
MW_HD uint32_t abstop12(float x) { return (f2u(x) >> 20) & 0x7ffu; }

MW_HD void sincosf_glibc(float y, float &sn, float &cs)
{
    // valid for |y| < 120 (angles of a few turns); beyond that the reference's own libm takes another path
    double x = y;
    if (abstop12(y) < abstop12(0x1.921FB6p-1f)) {
        if (abstop12(y) < abstop12(0x1p-12f)) { sn = y; cs = 1.0f; return; }
        const double x2 = x * x;
        sn = sincosf_poly(x, x2, 0, 0);
        cs = sincosf_poly(x, x2, 0, 1);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = fma(-(double)n, 0x1.921FB54442D18p0, x);
    const double sg = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    const int tab = (n & 2) ? 1 : 0;
    sn = sincosf_poly(x * sg, x * x, tab, n);
    cs = sincosf_poly(x * sg, x * x, tab, n ^ 1);
}

// ---------------------------------------------------------------- Mesa m_matrix.c
struct Mat4 { float m[16]; };       // column-major like GL: m[col * 4 + row]

MW_HD void mat_identity(Mat4 &a)
{
    for (int i = 0; i < 16; ++i) a.m[i] = 0.0f;
    a.m[0] = a.m[5] = a.m[10] = a.m[15] = 1.0f;
}

// matmul4: every element ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j) + a_i3 b_3j, separately rounded
MW_HD void matmul4(Mat4 &p, const Mat4 &a, const Mat4 &b)
{
    Mat4 o;
    for (int i = 0; i < 4; ++i) {
        const float ai0 = a.m[i], ai1 = a.m[4 + i], ai2 = a.m[8 + i], ai3 = a.m[12 + i];
        for (int j = 0; j < 4; ++j)
            o.m[j * 4 + i] = ((ai0 * b.m[j * 4] + ai1 * b.m[j * 4 + 1]) + ai2 * b.m[j * 4 + 2]) + ai3 * b.m[j * 4 + 3];
    }
    p = o;
}

MW_HD void mat_translate(Mat4 &a, float x, float y, float z)
{
    float *m = a.m;
    m[12] = ((m[0] * x + m[4] * y) + m[8] * z) + m[12];
    m[13] = ((m[1] * x + m[5] * y) + m[9] * z) + m[13];
    m[14] = ((m[2] * x + m[6] * y) + m[10] * z) + m[14];
    m[15] = ((m[3] * x + m[7] * y) + m[11] * z) + m[15];
}

MW_HD void mat_scale(Mat4 &a, float x, float y, float z)
{
    float *m = a.m;
    m[0] *= x; m[4] *= y; m[8] *= z;
    m[1] *= x; m[5] *= y; m[9] *= z;
    m[2] *= x; m[6] *= y; m[10] *= z;
    m[3] *= x; m[7] *= y; m[11] *= z;
}

// _math_matrix_rotate(mat, angle, 0, 1, 0): c, s into an identity, then a full product
MW_HD void mat_rotate_y(Mat4 &a, float angle)
{
    const float arg = (float)((double)angle * 3.14159265358979323846 / 180.0);
    float s, c;
    sincosf_glibc(arg, s, c);
    Mat4 r;
    mat_identity(r);
    r.m[0] = c; r.m[10] = c;
    r.m[8] = s; r.m[2] = -s;
    matmul4(a, a, r);
}

enum { MF_ROTATION = 1, MF_TRANSLATION = 2, MF_UNIFORM_SCALE = 4, MF_GENERAL = 8 };

// analyse_from_scratch for a MATRIX_3D-shaped matrix (last row 0 0 0 1)
MW_HD unsigned analyse_from_scratch(const Mat4 &a)
{
    const float *m = a.m;
    unsigned flags = 0;
    if (m[12] != 0.0f || m[13] != 0.0f || m[14] != 0.0f) flags |= MF_TRANSLATION;
    const float c1 = (m[0] * m[0] + m[1] * m[1]) + m[2] * m[2];
    const float c2 = (m[4] * m[4] + m[5] * m[5]) + m[6] * m[6];
    const float c3 = (m[8] * m[8] + m[9] * m[9]) + m[10] * m[10];
    const float d1 = (m[0] * m[4] + m[1] * m[5]) + m[2] * m[6];
    const float e6 = 1e-6f * 1e-6f;
    if ((c1 - c2) * (c1 - c2) < e6 && (c1 - c3) * (c1 - c3) < e6) {
        if ((c1 - 1.0f) * (c1 - 1.0f) > e6) flags |= MF_UNIFORM_SCALE;
    } else {
        flags |= MF_GENERAL;
    }
    if (d1 * d1 < e6) {
        float cp[3] = {m[1] * m[6] - m[2] * m[5], m[2] * m[4] - m[0] * m[6], m[0] * m[5] - m[1] * m[4]};
        cp[0] -= m[8]; cp[1] -= m[9]; cp[2] -= m[10];
        if ((cp[0] * cp[0] + cp[1] * cp[1]) + cp[2] * cp[2] < e6) flags |= MF_ROTATION;
        else flags |= MF_GENERAL;
    } else {
        flags |= MF_GENERAL;
    }
    return flags;
}

// upper-left 3x3 of the inverse, column-major 3x3 (inv[col * 3 + row]) (invert_matrix_3d / _general)
MW_HD void invert3(const Mat4 &mv, unsigned flags, float inv[9])
{
    const float *in = mv.m;
#define MWGL_IN(r, c) in[(c) * 4 + (r)]
#define MWGL_OUT(r, c) inv[(c) * 3 + (r)]
    for (int i = 0; i < 9; ++i) inv[i] = 0.0f;
    if (flags & MF_GENERAL) {
        float pos = 0.0f, neg = 0.0f, t;
        t = MWGL_IN(0, 0) * MWGL_IN(1, 1) * MWGL_IN(2, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = MWGL_IN(1, 0) * MWGL_IN(2, 1) * MWGL_IN(0, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = MWGL_IN(2, 0) * MWGL_IN(0, 1) * MWGL_IN(1, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = -MWGL_IN(2, 0) * MWGL_IN(1, 1) * MWGL_IN(0, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = -MWGL_IN(1, 0) * MWGL_IN(0, 1) * MWGL_IN(2, 2); if (t >= 0.0f) pos += t; else neg += t;
        t = -MWGL_IN(0, 0) * MWGL_IN(2, 1) * MWGL_IN(1, 2); if (t >= 0.0f) pos += t; else neg += t;
        float det = pos + neg;
        if (fabsf(det) < 1e-25f) return;
        det = 1.0f / det;
        MWGL_OUT(0, 0) = (MWGL_IN(1, 1) * MWGL_IN(2, 2) - MWGL_IN(2, 1) * MWGL_IN(1, 2)) * det;
        MWGL_OUT(0, 1) = -(MWGL_IN(0, 1) * MWGL_IN(2, 2) - MWGL_IN(2, 1) * MWGL_IN(0, 2)) * det;
        MWGL_OUT(0, 2) = (MWGL_IN(0, 1) * MWGL_IN(1, 2) - MWGL_IN(1, 1) * MWGL_IN(0, 2)) * det;
        MWGL_OUT(1, 0) = -(MWGL_IN(1, 0) * MWGL_IN(2, 2) - MWGL_IN(2, 0) * MWGL_IN(1, 2)) * det;
        MWGL_OUT(1, 1) = (MWGL_IN(0, 0) * MWGL_IN(2, 2) - MWGL_IN(2, 0) * MWGL_IN(0, 2)) * det;
        MWGL_OUT(1, 2) = -(MWGL_IN(0, 0) * MWGL_IN(1, 2) - MWGL_IN(1, 0) * MWGL_IN(0, 2)) * det;
        MWGL_OUT(2, 0) = (MWGL_IN(1, 0) * MWGL_IN(2, 1) - MWGL_IN(2, 0) * MWGL_IN(1, 1)) * det;
        MWGL_OUT(2, 1) = -(MWGL_IN(0, 0) * MWGL_IN(2, 1) - MWGL_IN(2, 0) * MWGL_IN(0, 1)) * det;
        MWGL_OUT(2, 2) = (MWGL_IN(0, 0) * MWGL_IN(1, 1) - MWGL_IN(1, 0) * MWGL_IN(0, 1)) * det;
        return;
    }
    float scale = 1.0f;
    if (flags & MF_UNIFORM_SCALE) {
        scale = (MWGL_IN(0, 0) * MWGL_IN(0, 0) + MWGL_IN(0, 1) * MWGL_IN(0, 1)) + MWGL_IN(0, 2) * MWGL_IN(0, 2);
        if (scale == 0.0f) return;
        scale = 1.0f / scale;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) MWGL_OUT(r, c) = scale * MWGL_IN(c, r);
    } else {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) MWGL_OUT(r, c) = MWGL_IN(c, r);
    }
#undef MWGL_IN
#undef MWGL_OUT
}

// ---------------------------------------------------------------- per-frame GL state
struct Frame {
    Mat4 proj, view;
    unsigned view_flags;        // geometry flags of the camera modelview after Mesa's analysis
    float light_eye[3];         // EyePosition of GL_LIGHT0 (w = 0): modelview * (light_pos + 1) at glCallList time
    float l_amb[3], l_dif[3];
    float vp_scale[3], vp_trans[3];
};

// Transform of one draw: the MVP matrix, the light direction in the draw's OBJECT space, the normal rescale factor.
struct Xform {
    Mat4 mvp;
    float light[3];
    float nscale;
};

// camera modelview + projection.  eye / at: Agent.cam_pos and cam_pos + cam_dir as numpy evaluates them (doubles).
MW_HD void frame_perspective(Frame &f, const double eye[3], const double at[3], double cot, int W, int H)
{
    const double aspect = (double)W / (double)H, zn = 0.04, zf = 100.0, dz = zf - zn;
    for (int i = 0; i < 16; ++i) f.proj.m[i] = 0.0f;
    f.proj.m[0] = (float)(cot / aspect);
    f.proj.m[5] = (float)cot;
    f.proj.m[10] = (float)(-(zf + zn) / dz);
    f.proj.m[11] = -1.0f;
    f.proj.m[14] = (float)(-2 * zn * zf / dz);
    float fw[3] = {(float)(at[0] - eye[0]), (float)(at[1] - eye[1]), (float)(at[2] - eye[2])};
    float r = (float)sqrt((double)((fw[0] * fw[0] + fw[1] * fw[1]) + fw[2] * fw[2]));
    if (r != 0.0f) { fw[0] /= r; fw[1] /= r; fw[2] /= r; }
    float side[3] = {fw[1] * 0.0f - fw[2] * 1.0f, fw[2] * 0.0f - fw[0] * 0.0f, fw[0] * 1.0f - fw[1] * 0.0f};
    r = (float)sqrt((double)((side[0] * side[0] + side[1] * side[1]) + side[2] * side[2]));
    if (r != 0.0f) { side[0] /= r; side[1] /= r; side[2] /= r; }
    const float up[3] = {side[1] * fw[2] - side[2] * fw[1], side[2] * fw[0] - side[0] * fw[2], side[0] * fw[1] - side[1] * fw[0]};
    mat_identity(f.view);
    float *m = f.view.m;
    m[0] = side[0]; m[4] = side[1]; m[8] = side[2];
    m[1] = up[0];   m[5] = up[1];   m[9] = up[2];
    m[2] = -fw[0];  m[6] = -fw[1];  m[10] = -fw[2];
    mat_translate(f.view, (float)-eye[0], (float)-eye[1], (float)-eye[2]);
}

// render_top_view (miniworld.py:1108-1160): glOrtho on floats, glLoadMatrixf (x, y, z) -> (x, -z, y)
MW_HD void frame_top(Frame &f, double min_x, double max_x, double min_z, double max_z)
{
    const float l = (float)min_x, r = (float)max_x, b = (float)-max_z, t = (float)-min_z, n = -100.0f, fa = 100.0f;
    mat_identity(f.proj);
    f.proj.m[0] = 2.0f / (r - l);   f.proj.m[12] = -(r + l) / (r - l);
    f.proj.m[5] = 2.0f / (t - b);   f.proj.m[13] = -(t + b) / (t - b);
    f.proj.m[10] = -2.0f / (fa - n); f.proj.m[14] = -(fa + n) / (fa - n);
    const float M[16] = {1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 0, 0, 0, 1};
    for (int i = 0; i < 16; ++i) f.view.m[i] = M[i];
}

// viewport, light (miniworld.py:1031: (GLfloat*4)(*light_pos + [1]) adds 1 to every component and leaves w = 0)
MW_HD void frame_finish(Frame &f, int W, int H, const double light_pos[3], const double light_color[3], const double light_ambient[3])
{
    f.vp_scale[0] = (float)W * 0.5f; f.vp_trans[0] = (float)W * 0.5f;
    f.vp_scale[1] = (float)H * 0.5f; f.vp_trans[1] = (float)H * 0.5f;
    f.vp_scale[2] = 0.5f; f.vp_trans[2] = 0.5f;
    const float lp[3] = {(float)(light_pos[0] + 1.0), (float)(light_pos[1] + 1.0), (float)(light_pos[2] + 1.0)};
    const float *M = f.view.m;
    for (int i = 0; i < 3; ++i) {
        f.light_eye[i] = ((M[i] * lp[0] + M[4 + i] * lp[1]) + M[8 + i] * lp[2]) + M[12 + i] * 0.0f;
        f.l_amb[i] = (float)light_ambient[i];
        f.l_dif[i] = (float)light_color[i];
    }
    f.view_flags = analyse_from_scratch(f.view);
}

// One draw's transform: MVP = proj * mv; the light taken to the draw's object space through the inverse modelview and
// normalised (1 / sqrtf); a modelview that is not length preserving rescales the normals by the length of the inverse's
// third row (light.c update_modelview_scale, ffvertex_prog.c MUL normal, STATE_NORMAL_SCALE).
MW_HD void make_xform(const Frame &f, const Mat4 &mv, unsigned flags, Xform &x)
{
    matmul4(x.mvp, f.proj, mv);
    float inv[9];
    invert3(mv, flags, inv);
    const float *e = f.light_eye;
    float q[3];
    for (int i = 0; i < 3; ++i) q[i] = ((inv[i] * e[0] + inv[3 + i] * e[1]) + inv[6 + i] * e[2]) + 0.0f * 0.0f;
    float len = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    if (len != 0.0f) {
        len = 1.0f / sqrtf(len);
        q[0] *= len; q[1] *= len; q[2] *= len;
    }
    x.light[0] = q[0]; x.light[1] = q[1]; x.light[2] = q[2];
    x.nscale = 1.0f;
    if (flags & (MF_UNIFORM_SCALE | MF_GENERAL)) {
        float fl = (inv[2] * inv[2] + inv[5] * inv[5]) + inv[8] * inv[8];
        if (fl < 1e-12f) fl = 1.0f;
        x.nscale = sqrtf(fl);
    }
}

// Box.render / ImageFrame.render: glTranslatef(pos), glRotatef(angle, 0, 1, 0); MeshEnt.render: translate, scale, rotate
MW_HD void entity_xform(const Frame &f, const float pos[3], float angle_deg, float scale, bool scaled, Xform &x)
{
    Mat4 mv = f.view;
    mat_translate(mv, pos[0], pos[1], pos[2]);
    if (scaled) mat_scale(mv, scale, scale, scale);
    mat_rotate_y(mv, angle_deg);
    make_xform(f, mv, f.view_flags | MF_TRANSLATION | MF_ROTATION | (scaled ? MF_UNIFORM_SCALE : 0u), x);
}

// ---------------------------------------------------------------- vertex program
struct Vert {
    float clip[4];
    float win[4];       // window x, y (GL frame-buffer space, y up), z, 1 / w
    float st[2];
    float col[3];
    uint32_t clipmask;
};

MW_HD void light_vertex(const Frame &f, const Xform &x, const float n[3], const float c[3], float out[3])
{
    const float ns[3] = {n[0] * x.nscale, n[1] * x.nscale, n[2] * x.nscale};
    const float dot = (ns[0] * x.light[0] + ns[1] * x.light[1]) + ns[2] * x.light[2];
    const float d = dot > 0.0f ? dot : 0.0f;
    for (int i = 0; i < 3; ++i) {
        const float scene = 0.2f * c[i];
        float acc = f.l_amb[i] * c[i] + scene;
        acc = d * (f.l_dif[i] * c[i]) + acc;
        out[i] = acc < 0.0f ? 0.0f : (acc > 1.0f ? 1.0f : acc);
    }
}

MW_HD void transform_vertex(const Frame &f, const Xform &x, const float p[3], Vert &v)
{
    const float *m = x.mvp.m;
    for (int i = 0; i < 4; ++i) v.clip[i] = ((p[0] * m[i] + p[1] * m[4 + i]) + p[2] * m[8 + i]) + m[12 + i];
    const float w = v.clip[3];
    uint32_t mask = 0;
    if (v.clip[0] > w) mask |= 1u;
    if (v.clip[0] + w < 0.0f) mask |= 2u;
    if (v.clip[1] > w) mask |= 4u;
    if (v.clip[1] + w < 0.0f) mask |= 8u;
    if (v.clip[2] + w < 0.0f) mask |= 16u;
    if (v.clip[2] > w) mask |= 32u;
    v.clipmask = mask;
    const float oow = 1.0f / w;
    v.win[0] = fmaf(v.clip[0] * oow, f.vp_scale[0], f.vp_trans[0]);
    v.win[1] = fmaf(v.clip[1] * oow, f.vp_scale[1], f.vp_trans[1]);
    v.win[2] = fmaf(v.clip[2] * oow, f.vp_scale[2], f.vp_trans[2]);
    v.win[3] = oow;
}

// ---------------------------------------------------------------- boxes of polygons (big scenes' culling, mw_geom.hip)
// An axis-aligned box under the MVP matrix m: the frustum planes every corner lies outside of, whether it lies in front
// of the eye throughout (w >= 0.1), its extent in window x and its smallest w.
// "Outside a plane" carries a margin above the rounding of the sums below and of transform_vertex's (four roundings of half an
// ulp of at most mag[i] each, twice): every vertex inside the box is then outside that plane in transform_vertex's own
// arithmetic too, so a box may stand for its polygons in the frustum test (tests/test_engine_math_cpu.py).
struct BoxView { uint32_t all; bool front; float xmn, xmx, zq; };

MW_HD float rcp_estimate(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

MW_HD BoxView box_view(const float mn[3], const float mx[3], const float *m, float vp_scale_x, float vp_trans_x)
{
    const float R = fmaxf(fmaxf(fmaxf(fabsf(mn[0]), fabsf(mx[0])), fmaxf(fabsf(mn[1]), fabsf(mx[1]))), fmaxf(fabsf(mn[2]), fabsf(mx[2])));
    float mag[4];
    for (int i = 0; i < 4; ++i) mag[i] = ((fabsf(m[i]) + fabsf(m[4 + i])) + fabsf(m[8 + i])) * R + fabsf(m[12 + i]);
    const float ex = 4e-6f * (mag[0] + mag[3]), ey = 4e-6f * (mag[1] + mag[3]), ez = 4e-6f * (mag[2] + mag[3]);
    BoxView b = {0x3Fu, true, 1e30f, -1e30f, 1e30f};
    for (int k = 0; k < 8; ++k) {
        const float p[3] = {(k & 1) ? mx[0] : mn[0], (k & 2) ? mx[1] : mn[1], (k & 4) ? mx[2] : mn[2]};
        float cl[4];
        for (int i = 0; i < 4; ++i) cl[i] = ((p[0] * m[i] + p[1] * m[4 + i]) + p[2] * m[8 + i]) + m[12 + i];
        const float w = cl[3];
        uint32_t mask = 0;
        if (cl[0] - w > ex) mask |= 1u;
        if (cl[0] + w < -ex) mask |= 2u;
        if (cl[1] - w > ey) mask |= 4u;
        if (cl[1] + w < -ey) mask |= 8u;
        if (cl[2] + w < -ez) mask |= 16u;
        if (cl[2] - w > ez) mask |= 32u;
        b.all &= mask;
        b.front &= w >= 0.1f;
        const float wx = fmaf(cl[0] * rcp_estimate(w), vp_scale_x, vp_trans_x);
        b.xmn = fminf(b.xmn, wx); b.xmx = fmaxf(b.xmx, wx); b.zq = fminf(b.zq, w);
    }
    return b;
}

// ---------------------------------------------------------------- clipper (draw_pipe_clip.c)
// GOURAUD: the colour varies over the primitive (meshes) and is clipped like the texture coordinates; a flat
// primitive's colour is the same at every vertex and survives the interpolation o + t (c - c) unchanged.
#define MWGL_MAX_CLIP_VERTS 10

// the clipper's work-list vertex without the fields a flat-shaded primitive does not need there (colour: the same at every
// vertex; clip mask: zero for every vertex the clipper makes) — 40 bytes instead of 56 in the geometry kernel's LDS lists
struct alignas(16) ClipVert {       // (three 16-byte LDS accesses per copy)
    float clip[4];
    float win[4];
    float st[2];
    float pad[2];
};

template <class V>
MW_HD float clip_dist(const V &v, int plane)
{
    // dot4(clip, plane) with the planes (-1,0,0,1) (1,0,0,1) (0,-1,0,1) (0,1,0,1) (0,0,1,1) (0,0,-1,1)
    const float px = plane == 0 ? -1.0f : (plane == 1 ? 1.0f : 0.0f);
    const float py = plane == 2 ? -1.0f : (plane == 3 ? 1.0f : 0.0f);
    const float pz = plane == 4 ? 1.0f : (plane == 5 ? -1.0f : 0.0f);
    return ((v.clip[0] * px + v.clip[1] * py) + v.clip[2] * pz) + v.clip[3] * 1.0f;
}

MW_HD void clip_set_flat(Vert &d, const Vert &src) { for (int i = 0; i < 3; ++i) d.col[i] = src.col[i]; d.clipmask = 0; }
MW_HD void clip_set_flat(ClipVert &, const ClipVert &) {}
MW_HD void clip_copy_in(Vert &d, const Vert &s) { d = s; }
MW_HD void clip_copy_in(ClipVert &d, const Vert &s)
{
    for (int i = 0; i < 4; ++i) { d.clip[i] = s.clip[i]; d.win[i] = s.win[i]; }
    d.st[0] = s.st[0]; d.st[1] = s.st[1];
}

template <bool GOURAUD, class V>
MW_HD void clip_interp(const Frame &f, V &d, float t, const V &out, const V &in)
{
    for (int i = 0; i < 4; ++i) d.clip[i] = out.clip[i] + t * (in.clip[i] - out.clip[i]);
    for (int i = 0; i < 2; ++i) d.st[i] = out.st[i] + t * (in.st[i] - out.st[i]);
    if constexpr (GOURAUD) { for (int i = 0; i < 3; ++i) d.col[i] = out.col[i] + t * (in.col[i] - out.col[i]); d.clipmask = 0; }
    else clip_set_flat(d, out);
    const float oow = 1.0f / d.clip[3];
    d.win[0] = d.clip[0] * oow * f.vp_scale[0] + f.vp_trans[0];
    d.win[1] = d.clip[1] * oow * f.vp_scale[1] + f.vp_trans[1];
    d.win[2] = d.clip[2] * oow * f.vp_scale[2] + f.vp_trans[2];
    d.win[3] = oow;
}

// Clips the triangle (a, b, c) against the frustum planes named by the union of the vertices' clip masks, lowest plane
// first (do_clip_tri).  buf0 / buf1: two work lists of MWGL_MAX_CLIP_VERTS vertices (caller's storage: LDS on the
// device).  Returns the vertex count n of the result, left in *res (buf0 or buf1): the output triangles are
// (res[i-1], res[i], res[0]) for i = 2 .. n-1 (emit_poly: the provoking vertex stays last).  n = 0: nothing left.
template <bool GOURAUD, class V = Vert>
MW_HD int clip_triangle(const Frame &f, const Vert &a, const Vert &b, const Vert &c, V *buf0, V *buf1, V **res)
{
    uint32_t clipmask = a.clipmask | b.clipmask | c.clipmask;
    if (a.clipmask & b.clipmask & c.clipmask) { *res = buf0; return 0; }
    V *inl = buf0, *outl = buf1;
    clip_copy_in(inl[0], a); clip_copy_in(inl[1], b); clip_copy_in(inl[2], c);
    int n = 3;
    while (clipmask && n >= 3) {
        int plane = 0;
        while (!((clipmask >> plane) & 1u)) ++plane;
        clipmask &= ~(1u << plane);
        int oc = 0;
        int prev = 0;
        float dp_prev = clip_dist(inl[0], plane);
        for (int i = 1; i <= n; ++i) {
            const int cur = i == n ? 0 : i;
            const float dp = clip_dist(inl[cur], plane);
            if (!(dp == dp) || dp - dp != 0.0f) { *res = buf0; return 0; }      // NaN / Inf: the triangle is dropped
            if (dp_prev >= 0.0f) outl[oc++] = inl[prev];
            if ((dp >= 0.0f) != (dp_prev >= 0.0f)) {
                // the new vertex is interpolated from the endpoint that is closer to the plane, whichever way the edge
                // is traversed: both triangles sharing an edge get the same vertex
                // (one interpolation whichever way round: same arithmetic, half the code)
                const bool from_cur = fabsf(dp) < fabsf(dp_prev);
                const float t = from_cur ? dp / (dp - dp_prev) : dp_prev / (dp_prev - dp);
                clip_interp<GOURAUD>(f, outl[oc], t, inl[from_cur ? cur : prev], inl[from_cur ? prev : cur]);
                ++oc;
            }
            prev = cur;
            dp_prev = dp;
        }
        V *t = inl; inl = outl; outl = t;
        n = oc;
    }
    *res = inl;
    return n >= 3 ? n : 0;
}

// ---------------------------------------------------------------- triangle setup (llvmpipe lp_setup_tri.c, lp_state_setup.c)
struct Plane { float a0, dadx, dady; };

struct TriSetup {
    int32_t dcdx[3], dcdy[3];
    int64_t c[3];               // edge constants with the fill rule folded in: inside <=> c + dcdy * fy - dcdx * fx > 0
    int32_t minx, maxx, miny, maxy;     // snapped vertex bounds (24.8)
    Plane z, w, s, t, col[3];
};

MW_HD int32_t iround_even(float x) { return (int32_t)rintf(x); }

MW_HD void plane_coef(Plane &p, float a0, float a1, float a2, float dy20_ooa, float dy01_ooa, float dx20_ooa, float dx01_ooa,
                      float x0c, float y0c)
{
    const float da01 = a0 - a1, da20 = a2 - a0;
    p.dadx = da01 * dy20_ooa - da20 * dy01_ooa;
    p.dady = da20 * dx01_ooa - da01 * dx20_ooa;
    p.a0 = a0 - (p.dadx * x0c + p.dady * y0c);
}

// Would the setup's area test (on the snapped vertices) drop the triangle whatever the rounding of the snap does?  In units
// of 1/256 px the snapped coordinates differ from the exact ones by at most a half each, the edge differences by at most one, the area by at most the sum of the four
// differences plus two; the float products below add their own rounding (1e-6 of their size, generously).
MW_HD bool clearly_back(const float wa[4], const float wb[4], const float wc[4])
{
    const float dx01 = (wa[0] - wb[0]) * 256.0f, dy01 = (wa[1] - wb[1]) * 256.0f, dx20 = (wc[0] - wa[0]) * 256.0f, dy20 = (wc[1] - wa[1]) * 256.0f;
    const float p = dx01 * dy20, q = dx20 * dy01;
    const float margin = (fabsf(dx01) + fabsf(dy20)) + (fabsf(dx20) + fabsf(dy01)) + 4.0f + 1e-6f * (fabsf(p) + fabsf(q));
    return p - q > margin;
}

// The position-only part of the setup (edges + depth plane): what a kernel that only needs coverage and depth keys pays.
struct TriEdges {
    int32_t dcdx[3], dcdy[3];
    int64_t c[3];
    int32_t minx, maxx, miny, maxy;
    Plane z;
};

MW_HD bool setup_triangle_pos(const float wa[4], const float wb[4], const float wc[4], bool multisampled, TriEdges &s)
{
    const float off = multisampled ? 0.0f : 0.5f;
    const float *v[3] = {wa, wb, wc};
    int32_t fx[3], fy[3];
    for (int i = 0; i < 3; ++i) {
        fx[i] = iround_even((v[i][0] - off) * 256.0f);
        fy[i] = iround_even((v[i][1] - off) * 256.0f);
    }
    const int64_t dx01 = fx[0] - fx[1], dy01 = fy[0] - fy[1], dx20 = fx[2] - fx[0], dy20 = fy[2] - fy[0];
    if (dx01 * dy20 - dx20 * dy01 >= 0) return false;
    const float *t = v[0]; v[0] = v[1]; v[1] = t;
    int32_t ti = fx[0]; fx[0] = fx[1]; fx[1] = ti;
    ti = fy[0]; fy[0] = fy[1]; fy[1] = ti;
    for (int i = 0; i < 3; ++i) {
        const int j = i == 2 ? 0 : i + 1;
        s.dcdy[i] = fx[i] - fx[j];
        s.dcdx[i] = fy[i] - fy[j];
        s.c[i] = (int64_t)s.dcdx[i] * fx[i] - (int64_t)s.dcdy[i] * fy[i];
        if (s.dcdx[i] < 0) s.c[i]++;
        else if (s.dcdx[i] == 0 && s.dcdy[i] > 0) s.c[i]++;
    }
    s.minx = fx[0] < fx[1] ? fx[0] : fx[1]; if (fx[2] < s.minx) s.minx = fx[2];
    s.maxx = fx[0] > fx[1] ? fx[0] : fx[1]; if (fx[2] > s.maxx) s.maxx = fx[2];
    s.miny = fy[0] < fy[1] ? fy[0] : fy[1]; if (fy[2] < s.miny) s.miny = fy[2];
    s.maxy = fy[0] > fy[1] ? fy[0] : fy[1]; if (fy[2] > s.maxy) s.maxy = fy[2];
    const float fdx01 = v[0][0] - v[1][0], fdy01 = v[0][1] - v[1][1];
    const float fdx20 = v[2][0] - v[0][0], fdy20 = v[2][1] - v[0][1];
    const float ooa = 1.0f / (fdx01 * fdy20 - fdx20 * fdy01);
    plane_coef(s.z, v[0][2], v[1][2], v[2][2], fdy20 * ooa, fdy01 * ooa, fdx20 * ooa, fdx01 * ooa, v[0][0] - off, v[0][1] - off);
    return true;
}

// multisampled != 0: pixel_offset 0 (integer coordinates are pixel corners); else 0.5 (integer coordinates are centres).
// Returns false for a back-facing or zero-area triangle (culled on the snapped area).
MW_HD bool setup_triangle(const Vert &a, const Vert &b, const Vert &c, bool multisampled, bool textured, TriSetup &s)
{
    const float off = multisampled ? 0.0f : 0.5f;
    const Vert *v[3] = {&a, &b, &c};
    int32_t fx[3], fy[3];
    for (int i = 0; i < 3; ++i) {
        fx[i] = iround_even((v[i]->win[0] - off) * 256.0f);
        fy[i] = iround_even((v[i]->win[1] - off) * 256.0f);
    }
    const int64_t dx01 = fx[0] - fx[1], dy01 = fy[0] - fy[1], dx20 = fx[2] - fx[0], dy20 = fy[2] - fy[0];
    const int64_t area = dx01 * dy20 - dx20 * dy01;
    if (area >= 0) return false;        // GL_CULL_FACE: front = counter-clockwise in window space (y up)
    // front faces are set up in the order (v1, v0, v2)
    const Vert *t = v[0]; v[0] = v[1]; v[1] = t;
    int32_t ti = fx[0]; fx[0] = fx[1]; fx[1] = ti;
    ti = fy[0]; fy[0] = fy[1]; fy[1] = ti;
    for (int i = 0; i < 3; ++i) {
        const int j = i == 2 ? 0 : i + 1;
        s.dcdy[i] = fx[i] - fx[j];
        s.dcdx[i] = fy[i] - fy[j];
        s.c[i] = (int64_t)s.dcdx[i] * fx[i] - (int64_t)s.dcdy[i] * fy[i];
        if (s.dcdx[i] < 0) s.c[i]++;
        else if (s.dcdx[i] == 0 && s.dcdy[i] > 0) s.c[i]++;
    }
    s.minx = fx[0] < fx[1] ? fx[0] : fx[1]; if (fx[2] < s.minx) s.minx = fx[2];
    s.maxx = fx[0] > fx[1] ? fx[0] : fx[1]; if (fx[2] > s.maxx) s.maxx = fx[2];
    s.miny = fy[0] < fy[1] ? fy[0] : fy[1]; if (fy[2] < s.miny) s.miny = fy[2];
    s.maxy = fy[0] > fy[1] ? fy[0] : fy[1]; if (fy[2] > s.maxy) s.maxy = fy[2];
    const float fdx01 = v[0]->win[0] - v[1]->win[0], fdy01 = v[0]->win[1] - v[1]->win[1];
    const float fdx20 = v[2]->win[0] - v[0]->win[0], fdy20 = v[2]->win[1] - v[0]->win[1];
    const float ooa = 1.0f / (fdx01 * fdy20 - fdx20 * fdy01);
    const float dy20_ooa = fdy20 * ooa, dy01_ooa = fdy01 * ooa, dx20_ooa = fdx20 * ooa, dx01_ooa = fdx01 * ooa;
    const float x0c = v[0]->win[0] - off, y0c = v[0]->win[1] - off;
#define MWGL_COEF(p, q0, q1, q2) plane_coef(p, q0, q1, q2, dy20_ooa, dy01_ooa, dx20_ooa, dx01_ooa, x0c, y0c)
    MWGL_COEF(s.z, v[0]->win[2], v[1]->win[2], v[2]->win[2]);
    MWGL_COEF(s.w, v[0]->win[3], v[1]->win[3], v[2]->win[3]);
    if (textured) {
        MWGL_COEF(s.s, v[0]->st[0] * v[0]->win[3], v[1]->st[0] * v[1]->win[3], v[2]->st[0] * v[2]->win[3]);
        MWGL_COEF(s.t, v[0]->st[1] * v[0]->win[3], v[1]->st[1] * v[1]->win[3], v[2]->st[1] * v[2]->win[3]);
    } else {
        s.s.a0 = s.s.dadx = s.s.dady = 0.0f;
        s.t = s.s;
    }
    for (int k = 0; k < 3; ++k)
        MWGL_COEF(s.col[k], v[0]->col[k] * v[0]->win[3], v[1]->col[k] * v[1]->win[3], v[2]->col[k] * v[2]->win[3]);
#undef MWGL_COEF
    return true;
}

// ---------------------------------------------------------------- primitive -> triangles
// A polygon is always the fan (1,2,0) (2,3,0).  A quad is split (0,1,3) (1,2,3) inside display list 1 (rooms, frames,
// static entities: vbo_save converts the list to indexed triangles) and for an immediate-mode draw call with a clipped
// vertex (the call runs through the draw module's pipeline), and (0,1,2) (0,2,3) for an immediate-mode draw call without
// one (llvmpipe's own vertex-buffer path).  Each half has its own plane coefficients.
enum { SPLIT_POLYGON = 0, SPLIT_QUAD_LIST = 1, SPLIT_QUAD_DIRECT = 2, SPLIT_TRIANGLE = 3 };

// sink(const TriSetup &) is called for every front-facing triangle that survives clipping, in drawing order
template <bool GOURAUD, class Sink>
MW_HD void emit_triangle(const Frame &f, const Vert &a, const Vert &b, const Vert &c, Vert *buf0, Vert *buf1, bool multisampled,
                         bool textured, Sink &sink)
{
    TriSetup ts;
    if ((a.clipmask | b.clipmask | c.clipmask) == 0u) {
        if (setup_triangle(a, b, c, multisampled, textured, ts)) sink(ts);
        return;
    }
    Vert *r;
    const int n = clip_triangle<GOURAUD>(f, a, b, c, buf0, buf1, &r);
    for (int i = 2; i < n; ++i)
        if (setup_triangle(r[i - 1], r[i], r[0], multisampled, textured, ts)) sink(ts);
}

template <bool GOURAUD, class Sink>
MW_HD void emit_primitive(const Frame &f, const Vert v[4], int nv, int split, Vert *buf0, Vert *buf1, bool multisampled,
                          bool textured, Sink &sink)
{
    if (split == SPLIT_TRIANGLE) {
        emit_triangle<GOURAUD>(f, v[0], v[1], v[2], buf0, buf1, multisampled, textured, sink);
    } else if (nv == 3) {
        emit_triangle<GOURAUD>(f, v[1], v[2], v[0], buf0, buf1, multisampled, textured, sink);
    } else if (split == SPLIT_QUAD_DIRECT) {
        emit_triangle<GOURAUD>(f, v[0], v[1], v[2], buf0, buf1, multisampled, textured, sink);
        emit_triangle<GOURAUD>(f, v[0], v[2], v[3], buf0, buf1, multisampled, textured, sink);
    } else if (split == SPLIT_QUAD_LIST) {
        emit_triangle<GOURAUD>(f, v[0], v[1], v[3], buf0, buf1, multisampled, textured, sink);
        emit_triangle<GOURAUD>(f, v[1], v[2], v[3], buf0, buf1, multisampled, textured, sink);
    } else {
        emit_triangle<GOURAUD>(f, v[1], v[2], v[0], buf0, buf1, multisampled, textured, sink);
        emit_triangle<GOURAUD>(f, v[2], v[3], v[0], buf0, buf1, multisampled, textured, sink);
    }
}

// drawBox (opengl.py:460-503): vertex selectors (bit 0 x max, bit 1 y max, bit 2 z max) and normals of the six faces
MW_HD int box_sel(int face, int k)
{
    const unsigned char sel[6][4] = {{7, 6, 4, 5}, {2, 3, 1, 0}, {6, 2, 0, 4}, {3, 7, 5, 1}, {7, 3, 2, 6}, {1, 5, 4, 0}};
    return sel[face][k];
}
MW_HD void box_normal(int face, float n[3])
{
    const float nn[6][3] = {{0, 0, 1}, {0, 0, -1}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, -1, 0}};
    n[0] = nn[face][0]; n[1] = nn[face][1]; n[2] = nn[face][2];
}

}  // namespace mwgl
