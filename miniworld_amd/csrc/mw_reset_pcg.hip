// mw_reset_kernel compiled for the MW_RNG_PCG64 stream (see mw_setup_pcg.hip).
#define MW_RNG_KIND 1
#define MW_RESET_KERNEL_NAME mw_reset_pcg_kernel
#define MW_REFILL_KERNEL_NAME mw_refill_pcg_kernel
#define MW_RESPAWN_KERNEL_NAME mw_collect_respawn_pcg_kernel
#include "mw_reset.hip"
