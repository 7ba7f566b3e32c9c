// Counter-based RNG for device-side domain randomisation and world generation.
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
// keyed by the env's 64-bit seed, indexed by a 64-bit draw counter.  Doubles are formed
// like numpy's Generator.uniform: lo + (hi - lo) * (u64 >> 11) * 2^-53 (the reference
// draws from np_random.uniform, params.py:99), i.e. same distribution, different stream
// (MW_RNG_PCG64 reproduces numpy's own stream instead, for the generators that support it).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mw {

// Two streams share the interface.  kind 0: Philox (a = seed, b = draw counter).  kind 1: numpy's PCG64
// (XSL-RR 128/64, O'Neill 2014) exactly as numpy.random.Generator(PCG64(SeedSequence(seed))) runs it —
// (a, b) = 128-bit state, (c, d) = 128-bit increment, seeded on the host (mw_engine.hip) — so that a
// device reset draws the very numbers the reference's env.reset(seed=...) draws (miniworld.py:551).
struct Rng { uint64_t a, b, c, d; int kind; uint32_t has32, buf32; };   // has32 / buf32: PCG64's buffered upper half

// Device code is compiled once per stream (MW_RNG_KIND = 0 in mw_setup.hip / mw_reset.hip, 1 in their
// *_pcg.hip twins, which only re-include them): a kernel carries one generator, not a run-time switch
// at each of its dozens of inlined draw sites.  Host code (seeding, the test hook) looks at r.kind.
#ifndef MW_RNG_KIND
#define MW_RNG_KIND 0
#endif
__host__ __device__ inline bool rng_is_pcg(const Rng &r)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return MW_RNG_KIND == 1;
#else
    return r.kind == 1;
#endif
}

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

// state = state * 0x2360ED051FC65DA44385DF649FCCF645 + inc  (mod 2^128)
__host__ __device__ inline void pcg64_step(uint64_t &hi, uint64_t &lo, uint64_t inc_hi, uint64_t inc_lo)
{
    const uint64_t mh = 2549297995355413924ull, ml = 4865540595714422341ull;
    const unsigned __int128 p = (unsigned __int128)lo * ml;
    uint64_t nh = (uint64_t)(p >> 64) + lo * mh + hi * ml;
    uint64_t nl = (uint64_t)p;
    const uint64_t sl = nl + inc_lo;
    nh += inc_hi + (sl < nl ? 1ull : 0ull);
    hi = nh; lo = sl;
}

__host__ __device__ inline uint64_t rng_next_u64(Rng &r)
{
    if (rng_is_pcg(r)) {    // pcg64_next64: step, then XSL-RR of the new state
        pcg64_step(r.a, r.b, r.c, r.d);
        const uint64_t x = r.a ^ r.b;
        const unsigned rot = (unsigned)(r.a >> 58);
        return (x >> rot) | (x << ((64u - rot) & 63u));
    }
    uint32_t c[4] = {(uint32_t)r.b, (uint32_t)(r.b >> 32), 0x6d77656eu, 0x67696e65u};
    uint32_t k0 = (uint32_t)r.a, k1 = (uint32_t)(r.a >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r.b += 1;
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

// numpy's pcg64_next32: a 64-bit output serves two 32-bit draws, low half first; doubles never touch
// the buffered half
__host__ __device__ inline uint32_t rng_next_u32(Rng &r)
{
    if (r.has32) { r.has32 = 0; return r.buf32; }
    const uint64_t v = rng_next_u64(r);
    r.has32 = 1; r.buf32 = (uint32_t)(v >> 32);
    return (uint32_t)v;
}

// uniform integer in [0, n), n >= 1.  PCG64 stream: exactly Generator.integers(0, n) / Generator.choice(n)
// for n <= 2^32 (random_bounded_uint64 -> buffered_bounded_lemire_uint32, numpy/random/src/distributions):
// no draw at all for n == 1.  Philox stream: one 64-bit draw, multiply-shift.
__host__ __device__ inline uint32_t rng_below(Rng &r, uint32_t n)
{
    if (rng_is_pcg(r)) {
        if (n <= 1u) return 0u;
        const uint32_t rng = n - 1u, rng_excl = n;
        uint64_t m = (uint64_t)rng_next_u32(r) * rng_excl;
        uint32_t leftover = (uint32_t)m;
        if (leftover < rng_excl) {
            const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
            while (leftover < threshold) {
                m = (uint64_t)rng_next_u32(r) * rng_excl;
                leftover = (uint32_t)m;
            }
        }
        return (uint32_t)(m >> 32);
    }
    return (uint32_t)(((rng_next_u64(r) >> 32) * (uint64_t)n) >> 32);
}

// storage: uint64[5][N] (a, b, c, d, has32 << 32 | buf32); the kind is a property of the kernel (MW_RNG_KIND)
__device__ inline Rng rng_load(const uint64_t *p, int N, int env)
{
    const bool pcg = MW_RNG_KIND == 1;
    const uint64_t w4 = pcg ? p[(size_t)4 * N + env] : 0ull;
    return Rng{p[env], p[(size_t)N + env], pcg ? p[(size_t)2 * N + env] : 0ull, pcg ? p[(size_t)3 * N + env] : 0ull, MW_RNG_KIND,
               (uint32_t)(w4 >> 32), (uint32_t)w4};
}
__device__ inline void rng_store(uint64_t *p, int N, int env, const Rng &r)
{
    p[(size_t)N + env] = r.b;
    if (MW_RNG_KIND == 1) {
        p[env] = r.a;
        p[(size_t)4 * N + env] = ((uint64_t)r.has32 << 32) | r.buf32;
    }
}

}  // namespace mw
