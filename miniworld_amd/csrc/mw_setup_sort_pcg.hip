// mw_setup_sort.hip compiled for the MW_RNG_PCG64 stream.
#define MW_SORT_VIS 1
#define MW_RNG_KIND 1
#define MW_SETUP_KERNEL_NAME mw_step_setup_sort_pcg_kernel
#include "mw_setup.hip"
