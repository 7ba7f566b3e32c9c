// The fragment half of the frame, as the reference's GL driver computes it (host + device): what happens to a sample
// between llvmpipe's triangle setup (mw_glmath.h) and the bytes FrameBuffer.resolve() / get_depth_map() return
// (opengl.py:339-435).  Restated from llvmpipe (Mesa 23.2.1) and measured against it (tests/golden/gl_*.npz):
//   coverage     integer edge functions on the 24.8 snapped vertices, exact
//   depth        z = fma(dzdy, y, fma(dzdx, x, z0)) at the sample position; 16 bits = round-to-nearest-even of
//                fl(z * 65535 / 65536) * 65536 (lp_bld_conv.c's mantissa trick); GL_LESS
//   attributes   a = fma(dady, y, fma(dadx, x, a0)) at the pixel centre, times 1 / (1/w plane)
//   texturing    projective (s, t) * (1 / q), q = the interpolated fourth coordinate; per-quad rho^2 = max over x, y of
//                the squared texel-space derivative taken as pixel differences over the 2x2 quad; lod = 0.5 * (exponent +
//                mantissa - 1); GL_LINEAR_MIPMAP_LINEAR with 8.8 fixed-point texel coordinates (round to nearest even),
//                8-bit weights, lerp a + ((w (b - a) + 128) >> 8) — x then y then between levels; GL_REPEAT
//   GL_MODULATE  (texel * (1 / 255)) * colour
//   resolve      ((s0 + s1) + s2 + ...) * (1 / n) in sample order; unorm8 = round-to-nearest-even of fl(c * 255)
#pragma once
#include "mw_glmath.h"

namespace mwgl {

MW_HD float plane_at(const Plane &p, float x, float y) { return fmaf(p.dady, y, fmaf(p.dadx, x, p.a0)); }

MW_HD uint32_t z_to_unorm16(float z)
{
    if (!(z > 0.0f)) z = 0.0f;
    if (z > 1.0f) z = 1.0f;
    const float r = z * (65535.0f / 65536.0f) + 128.0f;
    return f2u(r) & 0xffffu;
}

MW_HD uint32_t float_to_unorm8(float v)
{
    if (!(v > 0.0f)) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint32_t)(int32_t)rintf(v * 255.0f);
}

// texture coordinates of the pixel whose centre is (x, y) in the planes' coordinates; also returns 1 / W' for the colours
MW_HD void tex_coords(const Plane &w, const Plane &s, const Plane &t, float x, float y, float &so, float &to, float &oow)
{
    const float wv = plane_at(w, x, y);
    oow = 1.0f / wv;
    const float invq = 1.0f / (wv * oow);
    so = (plane_at(s, x, y) * oow) * invq;
    to = (plane_at(t, x, y) * oow) * invq;
}

// level l0 and the 8-bit weight of level l0 + 1 (0: one level only) from rho^2, the squared texel-space footprint of the quad:
// lod = 0.5 * (exponent + mantissa - 1), floor / fraction, clamped to the pyramid
MW_HD void lod_from_rho2(float rho2, int nlevels, int &l0, int &w8)
{
    const uint32_t b = f2u(rho2);
    const int e = (int)((b >> 23) & 255u) - 127;
    const float m = u2f((b & 0x7fffffu) | 0x3f800000u);
    const float lod = ((float)e + (m - 1.0f)) * 0.5f;
    const float fl = floorf(lod);
    const int ip = (int)fl;
    float fp = lod - fl;
    const int last = nlevels - 1;
    l0 = ip;
    if (!(rho2 > 0.0f) || ip < 0) { l0 = 0; fp = 0.0f; }
    else if (ip >= last) { l0 = last; fp = 0.0f; }
    w8 = (int)(fp * 256.0f);
}

// The same function of rho2 on the float's bits (the quad kernel's form: 10 instructions instead of 22).  For rho2 >= 1,
// U = bits - bits(1.0f) is exponent.mantissa in 9.23 fixed point, i.e. U 2^-23 = e + (m - 1) exactly; the float sum above
// rounds that value once, to nearest even — and so does the conversion of the integer U to float.  T = RNE(U) is an integer
// again: lod = T 2^-24, its floor T >> 24, the weight's eight bits (T >> 16) & 255.  Below 1 (and for a NaN) the level is 0.
// (mw_selftest_lod compares the two for all 2^32 floats.)
MW_HD void lod_from_rho2_bits(float rho2, int nlevels, int &l0, int &w8)
{
    const uint32_t U = rho2 >= 1.0f ? f2u(rho2) - 0x3F800000u : 0u;
    const uint32_t T = (uint32_t)(float)U;
    const int ip = (int)(T >> 24), last = nlevels - 1;
    l0 = ip < last ? ip : last;
    w8 = ip < last ? (int)((T >> 16) & 255u) : 0;
}

// rho^2 from the quad's three corner coordinates: max over x, y of the squared texel-space derivative, pixel differences
MW_HD float lod_rho2(float s00, float t00, float s10, float t10, float s01, float t01, float fw, float fh)
{
    const float dsdx = (s10 - s00) * fw, dsdy = (s01 - s00) * fw, dtdx = (t10 - t00) * fh, dtdy = (t01 - t00) * fh;
    const float rx = dsdx * dsdx + dtdx * dtdx, ry = dsdy * dsdy + dtdy * dtdy;
    return rx > ry ? rx : ry;
}

// lod from the quad's three corner coordinates: level l0 and the 8-bit weight of level l0 + 1 (0: one level only)
MW_HD void lod_select(float s00, float t00, float s10, float t10, float s01, float t01, float fw, float fh, int nlevels,
                      int &l0, int &w8)
{
    lod_from_rho2(lod_rho2(s00, t00, s10, t10, s01, t01, fw, fh), nlevels, l0, w8);
}

// one axis of the GL_LINEAR lookup: first texel (wrapped, GL_REPEAT) and the 8-bit weight of its right / upper neighbour
MW_HD void linear_coord(float coord, int size, bool pot, int &i0, int &wt)
{
    if (pot) {
        const int32_t f = iround_even(coord * (float)size * 256.0f) - 128;
        wt = f & 255;
        i0 = (f >> 8) & (size - 1);
    } else {
        const float fr = coord - floorf(coord);
        const int32_t f = iround_even(fr * (float)size * 256.0f) - 128;
        wt = f & 255;
        int32_t i = f >> 8;
        if (i < 0) i = size - 1;
        if (i > size - 1) i = size - 1;
        i0 = i;
    }
}

MW_HD int lerp8(int a, int b, int w) { return a + ((w * (b - a) + 128) >> 8); }

// bilinear filter of one footprint (four RGBA8 texels: (i, j), (i+1, j), (i, j+1), (i+1, j+1)): x first, then y
MW_HD void bilerp_rgb(uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, int wx, int wy, int out[3])
{
    for (int c = 0; c < 3; ++c) {
        const int sh = 8 * c;
        const int a = (int)((t00 >> sh) & 255u), b = (int)((t10 >> sh) & 255u), cc = (int)((t01 >> sh) & 255u), d = (int)((t11 >> sh) & 255u);
        out[c] = lerp8(lerp8(a, b, wx), lerp8(cc, d, wx), wy);
    }
}

}  // namespace mwgl
