/* ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into or called by the product.
 *
 * mwo_sincos(): deterministic double-precision sin/cos built from IEEE + - * only
 * (Cody-Waite 2-stage pi/2 reduction + the classic fdlibm minimax kernels; published
 * algorithm: Sun fdlibm k_sin.c / k_cos.c / e_rem_pio2.c medium-size path).
 *
 * Why it exists: the reference computes headings with libm (entity.py:101-102,
 * math.py:18-19 via math.cos/math.sin).  glibc and ROCm's ocml differ by <=1 ulp on
 * some arguments, which would make "HIP engine == oracle" only approximately true.
 * Both the oracle (this file) and the engine (csrc/mw_math.h, written separately)
 * implement this same published algorithm, so their results are bit-identical, and
 * tests/test_oracle_math.py pins this function to glibc within 1 ulp.
 *
 * Valid for |x| < ~1e6 (agent headings stay far below that: <= 1536 steps * 20 deg).
 */
#include "mwo.h"

static const double PIO2_1  = 1.57079632673412561417e+00; /* first 33 bits of pi/2 */
static const double PIO2_2  = 6.07710050630396597660e-11; /* second 33 bits */
static const double PIO2_2T = 2.02226624879595063154e-21;
static const double INVPIO2 = 6.36619772367581382433e-01;

static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                    S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                    S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                    C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                    C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;

/* sin on [-pi/4, pi/4] with tail y (x+y is the reduced argument) */
static double ksin(double x, double y)
{
    double z = x * x;
    double v = z * x;
    double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

/* cos on [-pi/4, pi/4] with tail y */
static double kcos(double x, double y)
{
    double z = x * x;
    double r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

void mwo_sincos(double x, double *s, double *c)
{
    /* k = nearest integer to x*2/pi (round-half-away is fine: |r| stays <= ~pi/4+eps) */
    double t = x * INVPIO2;
    double fn = (t >= 0.0) ? (double)(long long)(t + 0.5) : -(double)(long long)(0.5 - t);
    long long k = (long long)fn;
    /* two-stage Cody-Waite, both stages always run (branch-free => deterministic):
     * pi/2 = PIO2_1 + PIO2_2 + PIO2_2T carries ~119 bits, ample for |x| < 1e6 */
    double tt = x - fn * PIO2_1;   /* exact: PIO2_1 has 33 bits, |fn| < 2^20 */
    double w = fn * PIO2_2;
    double r = tt - w;
    w = fn * PIO2_2T - ((tt - r) - w);
    double y0 = r - w;
    double y1 = (r - y0) - w;
    double sn = ksin(y0, y1), cs = kcos(y0, y1);
    switch ((int)(k & 3)) {
    case 0: *s = sn;  *c = cs;  break;
    case 1: *s = cs;  *c = -sn; break;
    case 2: *s = -sn; *c = -cs; break;
    default: *s = -cs; *c = sn; break;
    }
}
