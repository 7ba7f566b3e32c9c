/* ORACLE — TEST INFRASTRUCTURE ONLY.
 *
 * Mip pyramid for Texture.load (miniworld/opengl.py:148-184: glTexImage2D(GL_RGB) +
 * glGenerateMipmap with GL_NICEST).  glGenerateMipmap's filter is implementation
 * defined; pinned here:
 *   - level k+1 has dims max(1, floor(n/2)) per axis;
 *   - even axis: the two source texels 2i, 2i+1 with weights 1,1;
 *   - odd axis n = 2d+1 (d >= 1): texels 2i, 2i+1, 2i+2 with weights d-i, d, i+1
 *     (the polyphase box filter of Guthe & Heckbert, "Non-power-of-two mipmap creation");
 *   - axis of length 1 stays (weight 1);
 *   - separable integer weights, exact integer sum, round half up.
 */
#include "mwo.h"
#include <math.h>
#include <string.h>

int64_t mwo_mip_bytes(int32_t w, int32_t h, int32_t *nlevels)
{
    int64_t total = 0;
    int n = 0;
    for (;;) {
        total += (int64_t)w * h * 3;
        ++n;
        if (w == 1 && h == 1) break;
        w = w > 1 ? w / 2 : 1;
        h = h > 1 ? h / 2 : 1;
    }
    if (nlevels) *nlevels = n;
    return total;
}

/* One axis of the minification blit: destination texel i of dn reads source texels i0, i1 with an 8-bit weight.
 * Mesa's glGenerateMipmap on llvmpipe is a GL_LINEAR blit of the previous level (util_gen_mipmap): the sampler works
 * on 24.8 fixed-point texel coordinates, fixed = iround(s * 256) - 128 with s = (i + 0.5) * n / dn, texel = fixed >> 8,
 * weight = fixed & 255, CLAMP_TO_EDGE.  For an even axis that is texels 2i, 2i+1 with weight 128. */
static void axis_taps(int n, int dn, int i, int *i0, int *i1, int *wt)
{
    if (n == 1) { *i0 = *i1 = 0; *wt = 0; return; }
    double s = ((double)i + 0.5) * (double)n / (double)dn * 256.0;
    long fixed = lrint(s) - 128;                 /* round half to even (cvtps2dq) */
    long ip = fixed >> 8;
    *wt = (int)(fixed & 255);
    *i0 = ip < 0 ? 0 : (ip > n - 1 ? n - 1 : (int)ip);
    *i1 = ip + 1 < 0 ? 0 : (ip + 1 > n - 1 ? n - 1 : (int)(ip + 1));
}

/* 8-bit lerp of the llvmpipe AoS sampler: a + ((w * (b - a) + 128) >> 8), arithmetic shift */
static inline int lerp8(int a, int b, int w) { return a + ((w * (b - a) + 128) >> 8); }

void mwo_build_mips(const uint8_t *rgb, int32_t w, int32_t h, uint8_t *out)
{
    memcpy(out, rgb, (size_t)w * h * 3);
    const uint8_t *src = out;
    while (!(w == 1 && h == 1)) {
        int nw = w > 1 ? w / 2 : 1, nh = h > 1 ? h / 2 : 1;
        uint8_t *dst = (uint8_t *)src + (size_t)w * h * 3;
        for (int j = 0; j < nh; ++j) {
            int j0, j1, wy;
            axis_taps(h, nh, j, &j0, &j1, &wy);
            for (int i = 0; i < nw; ++i) {
                int i0, i1, wx;
                axis_taps(w, nw, i, &i0, &i1, &wx);
                for (int c = 0; c < 3; ++c) {
                    int t0 = lerp8(src[((size_t)j0 * w + i0) * 3 + c], src[((size_t)j0 * w + i1) * 3 + c], wx);
                    int t1 = lerp8(src[((size_t)j1 * w + i0) * 3 + c], src[((size_t)j1 * w + i1) * 3 + c], wx);
                    dst[((size_t)j * nw + i) * 3 + c] = (uint8_t)lerp8(t0, t1, wy);
                }
            }
        }
        src = dst;
        w = nw; h = nh;
    }
}
