// The fused step + geometry kernel compiled for the MW_RNG_PCG64 stream (numpy's PCG64 in the reference's call order,
// mw_rng.h): the same source as mw_geom.hip with the other generator inlined; only the fused kernel is emitted.
#define MW_RNG_KIND 1
#define MW_GEOM_STEP_ONLY 1
#define MW_GEOM_STEP_KERNEL_NAME mw_geom_step_pcg_kernel
#include "mw_geom.hip"
