// The dense K1 with the mesh-entity walks compiled in (mw_setup_dense.hip: MW_DENSE_MESH), Philox stream.
#define MW_DENSE_MESH 1
#define MW_DENSE_KERNEL_NAME mw_step_setup_dense_mesh_kernel
#include "mw_setup_dense.hip"
