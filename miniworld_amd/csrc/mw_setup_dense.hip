// K1, dense variant — the step of SMALL scenes, several envs per wavefront.
//
// Same work and same arithmetic as mw_setup.hip (which see for the reference file:line map), other lane mapping: the
// wave-per-env kernel spends a wavefront's 64 lanes on one env, whose scalar work (f64 physics) is repeated 64 times.
// Here an env owns L consecutive lanes (Hallway / OneRoom: 12, five envs per wave), every lane of an env evaluates the
// env's step itself, and the leading lane writes its state and flags and — on an episode's end — runs the generator.
// Not handled here (the engine launches mw_setup.hip instead): scenes whose L exceeds 32, MW_TASK_COLLECT.
#include "mw_setup_dense_body.h"

#ifndef MW_DENSE_KERNEL_NAME
#define MW_DENSE_KERNEL_NAME mw_step_setup_dense_kernel
#endif


extern "C" __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) void MW_DENSE_KERNEL_NAME(
    MwArgs a, int do_step, int lanes_per_env, const int32_t *__restrict__ actions, float *__restrict__ reward,
    uint8_t *__restrict__ term, uint8_t *__restrict__ trunc)
{
    __shared__ unsigned char gen_ws[MW_GEN_WS_BYTES];      // generator scratch (used by the Maze generator only)
    const int lane = threadIdx.x;
    const int L = lanes_per_env;
    const int epw = 64 / L;                                                   // envs per wavefront
    const int el = (int)(((uint32_t)lane * ((65536u + (uint32_t)L - 1u) / (uint32_t)L)) >> 16);     // lane / L, exact for lane < 64
    const int slot = lane - el * L;
    // spare mode: blocks appended to the grid regenerate the spare worlds consumed in earlier steps, beside the step
    const int env_blocks = (a.N + epw - 1) / epw;
    if ((int)blockIdx.x >= env_blocks) {
        mw::refill_spares(a, (int)blockIdx.x - env_blocks, lane, gen_ws);
        return;
    }
    const int env = (int)blockIdx.x * epw + el;
    if (el >= epw || env >= a.N) return;
    dense_step(a, do_step, env, lane, slot == 0, actions, reward, term, trunc, gen_ws);
}
