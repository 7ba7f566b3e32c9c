// Counter-based RNG for device-side domain randomisation and world generation.
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
// keyed by the env's 64-bit seed, indexed by a 64-bit draw counter.  Doubles are formed
// like numpy's Generator.uniform: lo + (hi - lo) * (u64 >> 11) * 2^-53 (the reference
// draws from np_random.uniform, params.py:99), i.e. same distribution, different stream
// (stream-exact PCG64 parity is host-side only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mw {

struct Rng { uint64_t seed, ctr; };

__host__ __device__ inline void philox_round(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__host__ __device__ inline uint64_t rng_next_u64(Rng &r)
{
    uint32_t c[4] = {(uint32_t)r.ctr, (uint32_t)(r.ctr >> 32), 0x6d77656eu, 0x67696e65u};
    uint32_t k0 = (uint32_t)r.seed, k1 = (uint32_t)(r.seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r.ctr += 1;
    return ((uint64_t)c[1] << 32) | c[0];
}

__host__ __device__ inline double rng_double(Rng &r)
{
    return (double)(rng_next_u64(r) >> 11) * (1.0 / 9007199254740992.0);
}

__host__ __device__ inline double rng_uniform(Rng &r, double lo, double hi)
{
    return lo + (hi - lo) * rng_double(r);
}

__host__ __device__ inline uint32_t rng_below(Rng &r, uint32_t n)   // unbiased enough for n << 2^32
{
    return (uint32_t)(((rng_next_u64(r) >> 32) * (uint64_t)n) >> 32);
}

__device__ inline Rng rng_load(const uint64_t *p, int N, int env) { return Rng{p[env], p[(size_t)N + env]}; }
__device__ inline void rng_store(uint64_t *p, int N, int env, const Rng &r) { p[(size_t)N + env] = r.ctr; }

}  // namespace mw
