// Test hooks that need the GPU but no engine (tests/test_gpu_numerics.py).
//
// mw_selftest_rcp: the raster kernels take 1 / W of the perspective-correct interpolation with rcp_exact()
// (mw_raster_common.h: hardware reciprocal estimate + one fused Newton step) instead of the compiler's IEEE division
// sequence (11 instructions).  "Same result as the oracle's 1.0f / x" is a claim about every float: this kernel
// evaluates both for ALL 2^32 bit patterns and counts where they differ, per binade, so that the guard range inside
// rcp_exact() is measured, not assumed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mw_raster_common.h"

extern "C" __global__ void mw_selftest_rcp_kernel(unsigned long long *bad_per_exp, uint32_t *examples, unsigned int *n_examples)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const float want = 1.0f / x;                 // correctly rounded (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
        const float got = rcp_domain(x) ? rcp_exact(x) : want;      // the callers' guard
        const bool same = __float_as_uint(want) == __float_as_uint(got) || (want != want && got != got);
        if (!same) {
            atomicAdd(&bad_per_exp[(b >> 23) & 511u], 1ull);           // sign | exponent
            const unsigned int k = atomicAdd(n_examples, 1u);
            if (k < 64u) examples[k] = (uint32_t)b;
        }
    }
}

extern "C" int mw_selftest_rcp(unsigned long long *host_bad_per_exp /*[512]*/, uint32_t *host_examples /*[64]*/, uint32_t *host_n)
{
    unsigned long long *d_bad = nullptr;
    uint32_t *d_ex = nullptr;
    unsigned int *d_n = nullptr;
    if (hipMalloc((void **)&d_bad, 512 * 8) != hipSuccess || hipMalloc((void **)&d_ex, 64 * 4) != hipSuccess ||
        hipMalloc((void **)&d_n, 4) != hipSuccess) return -1;
    (void)hipMemset(d_bad, 0, 512 * 8); (void)hipMemset(d_ex, 0, 64 * 4); (void)hipMemset(d_n, 0, 4);
    hipLaunchKernelGGL(mw_selftest_rcp_kernel, dim3(256 * 32), dim3(256), 0, 0, d_bad, d_ex, d_n);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    (void)hipMemcpy(host_bad_per_exp, d_bad, 512 * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(host_examples, d_ex, 64 * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(host_n, d_n, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad); (void)hipFree(d_ex); (void)hipFree(d_n);
    return 0;
}

// mw_selftest_div: div_exact(a, rcp_exact(b), b) against a / b for 2^32 pseudo-random pairs of its domain
// (b in [1e-10, 1e10], |a| in [1e-25, 1e25] or zero; both signs of a), the mantissas drawn from a 64-bit mixer.
extern "C" __global__ void mw_selftest_div_kernel(unsigned long long *n_bad, uint32_t *examples)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        uint64_t z = i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;       // splitmix64
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        // exponents: b in 2^-33 .. 2^33 (inside [1e-10, 1e10]), a in 2^-83 .. 2^83 (inside [1e-25, 1e25])
        const uint32_t mb = (uint32_t)z & 0x7FFFFFu, ma = (uint32_t)(z >> 23) & 0x7FFFFFu;
        const uint32_t eb = 127u - 33u + (uint32_t)((z >> 46) % 67u), ea = 127u - 83u + (uint32_t)((z >> 53) % 167u);
        const float b = __uint_as_float((eb << 23) | mb);
        float a = __uint_as_float((((uint32_t)(z >> 63)) << 31) | (ea << 23) | ma);
        if ((i & 0xFFFFull) == 0ull) a = 0.0f;
        if (!div_domain(b)) continue;
        const float want = a / b;
        const float got = div_exact(a, rcp_exact(b), b);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            const unsigned long long k = atomicAdd(n_bad, 1ull);
            if (k < 32ull) { examples[2 * k] = __float_as_uint(a); examples[2 * k + 1] = __float_as_uint(b); }
        }
    }
}

extern "C" int mw_selftest_div(unsigned long long *host_n_bad, uint32_t *host_examples /*[64]*/)
{
    unsigned long long *d_n = nullptr;
    uint32_t *d_ex = nullptr;
    if (hipMalloc((void **)&d_n, 8) != hipSuccess || hipMalloc((void **)&d_ex, 64 * 4) != hipSuccess) return -1;
    (void)hipMemset(d_n, 0, 8); (void)hipMemset(d_ex, 0, 64 * 4);
    hipLaunchKernelGGL(mw_selftest_div_kernel, dim3(256 * 32), dim3(256), 0, 0, d_n, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    (void)hipMemcpy(host_n_bad, d_n, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(host_examples, d_ex, 64 * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_n); (void)hipFree(d_ex);
    return 0;
}

// mw_selftest_unorm8: the quad kernel converts a pixel's resolved sum with v_cvt_pk_u8_f32(acc * (255 / S)) instead of
// mwgl::float_to_unorm8(acc * (1 / S)) (clamp, * 255, rint, convert).  Compared here for ALL 2^32 floats, S = 4 and 8.
// mw_selftest_lod: mwgl::lod_from_rho2 (llvmpipe's lod arithmetic as the oracle states it) against lod_from_rho2_bits (the
// quad kernel's form on the float's bits) for all 2^32 floats and pyramids of 1 .. 12 levels.
extern "C" __global__ void mw_selftest_q_kernel(unsigned long long *n_bad /*[2]*/, uint32_t *examples /*[2][32]*/)
{
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (uint64_t b = tid; b < (1ull << 32); b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const bool bad_u8 = mwgl::float_to_unorm8(x * 0.125f) != __builtin_amdgcn_cvt_pk_u8_f32(x * (255.0f / 8), 0u, 0u) ||
                            mwgl::float_to_unorm8(x * 0.25f) != __builtin_amdgcn_cvt_pk_u8_f32(x * (255.0f / 4), 0u, 0u);
        if (bad_u8) { const unsigned long long k = atomicAdd(&n_bad[0], 1ull); if (k < 32ull) examples[k] = (uint32_t)b; }
        bool bad_lod = false;
        for (int nl = 1; nl <= 12; ++nl) {
            int l0, w8, l0b, w8b;
            mwgl::lod_from_rho2(x, nl, l0, w8);
            mwgl::lod_from_rho2_bits(x, nl, l0b, w8b);
            bad_lod |= l0 != l0b || w8 != w8b;
        }
        if (bad_lod) { const unsigned long long k = atomicAdd(&n_bad[1], 1ull); if (k < 32ull) examples[32 + k] = (uint32_t)b; }
    }
}

extern "C" int mw_selftest_q(unsigned long long *host_n_bad /*[2]*/, uint32_t *host_examples /*[64]*/)
{
    unsigned long long *d_n = nullptr;
    uint32_t *d_ex = nullptr;
    if (hipMalloc((void **)&d_n, 16) != hipSuccess || hipMalloc((void **)&d_ex, 64 * 4) != hipSuccess) return -1;
    (void)hipMemset(d_n, 0, 16); (void)hipMemset(d_ex, 0, 64 * 4);
    hipLaunchKernelGGL(mw_selftest_q_kernel, dim3(256 * 32), dim3(256), 0, 0, d_n, d_ex);
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    (void)hipMemcpy(host_n_bad, d_n, 16, hipMemcpyDeviceToHost);
    (void)hipMemcpy(host_examples, d_ex, 64 * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_n); (void)hipFree(d_ex);
    return 0;
}
