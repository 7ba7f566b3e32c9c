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

static int axis_taps(int n, int i, int idx[3], int wt[3], int *total)
{
    if (n == 1) { idx[0] = 0; wt[0] = 1; *total = 1; return 1; }
    if ((n & 1) == 0) { idx[0] = 2 * i; idx[1] = 2 * i + 1; wt[0] = wt[1] = 1; *total = 2; return 2; }
    int d = n / 2;
    idx[0] = 2 * i; idx[1] = 2 * i + 1; idx[2] = 2 * i + 2;
    wt[0] = d - i; wt[1] = d; wt[2] = i + 1;
    *total = n;
    return 3;
}

void mwo_build_mips(const uint8_t *rgb, int32_t w, int32_t h, uint8_t *out)
{
    memcpy(out, rgb, (size_t)w * h * 3);
    const uint8_t *src = out;
    while (!(w == 1 && h == 1)) {
        int nw = w > 1 ? w / 2 : 1, nh = h > 1 ? h / 2 : 1;
        uint8_t *dst = (uint8_t *)src + (size_t)w * h * 3;
        for (int j = 0; j < nh; ++j) {
            int jy[3], wy[3], ty;
            int ny = axis_taps(h, j, jy, wy, &ty);
            for (int i = 0; i < nw; ++i) {
                int ix[3], wx[3], tx;
                int nx = axis_taps(w, i, ix, wx, &tx);
                int64_t tot = (int64_t)tx * ty;
                for (int c = 0; c < 3; ++c) {
                    int64_t acc = 0;
                    for (int b = 0; b < ny; ++b)
                        for (int a = 0; a < nx; ++a)
                            acc += (int64_t)wx[a] * wy[b] * src[((size_t)jy[b] * w + ix[a]) * 3 + c];
                    dst[((size_t)j * nw + i) * 3 + c] = (uint8_t)((2 * acc + tot) / (2 * tot));
                }
            }
        }
        src = dst;
        w = nw; h = nh;
    }
}
