// The dense K1 compiled for the MW_RNG_PCG64 stream (numpy's PCG64 in the reference's call order, mw_rng.h):
// the same source as mw_setup_dense.hip with the other generator inlined.
#define MW_RNG_KIND 1
#define MW_DENSE_KERNEL_NAME mw_step_setup_dense_pcg_kernel
#include "mw_setup_dense.hip"
