// mw_reset_kernel — batched MiniWorldEnv.reset (miniworld.py:544-604): one thread per env, or one wavefront per
// env for the Maze generator (its 127 rooms are emitted one per lane; grid = N blocks then).
#include "mw_gen.h"

#ifndef MW_RESET_KERNEL_NAME
#define MW_RESET_KERNEL_NAME mw_reset_kernel
#endif
extern "C" __global__ __launch_bounds__(64) void MW_RESET_KERNEL_NAME(MwArgs a, const uint8_t *__restrict__ mask, int force_all, int mark_refill)
{
    __shared__ unsigned char ws[64][MW_GEN_WS_BYTES];
    const bool wave_per_env = a.generator == MW_GEN_MAZE;
    const int env = wave_per_env ? (int)blockIdx.x : (int)(blockIdx.x * 64 + threadIdx.x);
    if (env >= a.N) return;
    if (!force_all && !mask[env]) return;
    mw::generate_world(*a.gen_live, env, wave_per_env ? ws[0] : ws[threadIdx.x], wave_per_env ? (int)threadIdx.x : 0);
    if (mark_refill && (!wave_per_env || threadIdx.x == 0)) a.refill_mask[env] = 1u;      // spare mode: its spare is stale now
}

#ifndef MW_REFILL_KERNEL_NAME
#define MW_REFILL_KERNEL_NAME mw_refill_kernel
#endif
// spare mode: regenerate every consumed spare now (mw_reset needs them current); grid like K1's refill blocks
extern "C" __global__ __launch_bounds__(64) void MW_REFILL_KERNEL_NAME(MwArgs a)
{
    __shared__ unsigned char refill_ws[MW_GEN_WS_BYTES];
    mw::refill_spares(a, (int)blockIdx.x, (int)threadIdx.x, refill_ws);
}

#ifndef MW_RESPAWN_KERNEL_NAME
#define MW_RESPAWN_KERNEL_NAME mw_collect_respawn_kernel
#endif
// CollectHealth: the kit consumed by this step respawns after its frame was set up (collecthealth.py:86-90), with
// place_entity's draws from the env's stream; one thread per env, launched behind the geometry kernel
extern "C" __global__ __launch_bounds__(64) void MW_RESPAWN_KERNEL_NAME(MwArgs a)
{
    const int env = (int)(blockIdx.x * 64 + threadIdx.x);
    if (env >= a.N) return;
    const int rs = a.pending_remove[env];
    if (rs < 0) return;
    mw::collect_respawn(a, env, a.shared_geom ? 0 : env, rs, a.ax[env], a.az[env]);
    a.pending_remove[env] = -1;
}

#if MW_RNG_KIND == 0
// mw_reset without seeds in spare mode: the masked envs take their pre-generated world (one wavefront per env)
extern "C" __global__ __launch_bounds__(64) void mw_take_spare_kernel(MwArgs a, const uint8_t *__restrict__ mask, int force_all)
{
    const int env = blockIdx.x;
    if (env >= a.N) return;
    if (!force_all && !mask[env]) return;
    mw::take_spare(a, env, (int)threadIdx.x);
    if (threadIdx.x == 0) a.refill_mask[env] = 1u;
}
#endif
