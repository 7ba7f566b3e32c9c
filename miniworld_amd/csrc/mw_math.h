// Deterministic double-precision sin/cos for the engine (device + host).
//
// The reference computes headings with libm (entity.py:95-113 math.cos/math.sin,
// math.py:18-19).  ROCm's ocml and glibc disagree by an ulp on some arguments, so the
// engine evaluates the classic published algorithm instead (Sun fdlibm: two-stage
// Cody-Waite reduction by pi/2 followed by the k_sin / k_cos minimax kernels), using only
// IEEE-754 + - * in a fixed order.  Compiled with -ffp-contract=off the result is a pure
// function of the input bits on any conforming target.  Accurate to < 1 ulp, |x| < 1e6.
#pragma once
#include <hip/hip_runtime.h>

namespace mw {

struct SinCos { double s, c; };

__host__ __device__ inline double kernel_sin(double x, double tail)
{
    const double z = x * x;
    const double v = z * x;
    const double r = 8.33333333332248946124e-03 +
                     z * (-1.98412698298579493134e-04 +
                          z * (2.75573137070700676789e-06 +
                               z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    return x - ((z * (0.5 * tail - v * r) - tail) - v * -1.66666666666666324348e-01);
}

__host__ __device__ inline double kernel_cos(double x, double tail)
{
    const double z = x * x;
    const double r = z * (4.16666666666666019037e-02 +
                          z * (-1.38888888888741095749e-03 +
                               z * (2.48015872894767294178e-05 +
                                    z * (-2.75573143513906633035e-07 +
                                         z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * tail));
}

__host__ __device__ inline SinCos sincos_det(double x)
{
    const double t = x * 6.36619772367581382433e-01;            // x * 2/pi
    const double fn = (t >= 0.0) ? (double)(long long)(t + 0.5) : -(double)(long long)(0.5 - t);
    const long long q = (long long)fn;
    const double head = x - fn * 1.57079632673412561417e+00;    // exact (33-bit constant)
    double w = fn * 6.07710050630396597660e-11;
    const double r = head - w;
    w = fn * 2.02226624879595063154e-21 - ((head - r) - w);
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double sn = kernel_sin(y0, y1), cs = kernel_cos(y0, y1);
    SinCos o;
    switch ((int)(q & 3)) {
    case 0: o.s = sn; o.c = cs; break;
    case 1: o.s = cs; o.c = -sn; break;
    case 2: o.s = -sn; o.c = -cs; break;
    default: o.s = -cs; o.c = sn; break;
    }
    return o;
}

}  // namespace mw
