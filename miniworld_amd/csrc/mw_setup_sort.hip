// K1 for scenes with more than 64 visible polygons per env (Maze): the same source as mw_setup.hip, plus the
// per-polygon depth bound and the depth-sorted visiting order consumed by mw_raster_big_kernel.
#define MW_SORT_VIS 1
#define MW_SETUP_KERNEL_NAME mw_step_setup_sort_kernel
#include "mw_setup.hip"
