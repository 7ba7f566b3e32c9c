// Micro-benchmark, round 6: the shader clock a VALU-bound kernel actually runs at, and with it the issue cost of the two VALU
// instruction classes in CYCLES.  Every wavefront reads s_memtime (shader-clock ticks) and s_memrealtime (the constant 100 MHz
// counter) around its loop; clock = d(memtime) / d(memrealtime) x 100 MHz, cycles per instruction per SIMD = the loop's shader
// cycles / the instructions the SIMD's waves issued in it.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC clock_probe.hip -o libclock_probe.so ; run: python tools/ubench/run.py clock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define F4(ins) REP16(asm volatile(ins : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ins : "+v"(a1) : "v"(b), "v"(c)); asm volatile(ins : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ins : "+v"(a3) : "v"(b), "v"(c));)
#define US4(ins) REP16(asm volatile(ins : "+v"(u0) : "v"(ub), "s"(m)); asm volatile(ins : "+v"(u1) : "v"(ub), "s"(m)); asm volatile(ins : "+v"(u2) : "v"(ub), "s"(m)); asm volatile(ins : "+v"(u3) : "v"(ub), "s"(m));)
#define U4(ins) REP16(asm volatile(ins : "+v"(u0) : "v"(ub), "v"(uc)); asm volatile(ins : "+v"(u1) : "v"(ub), "v"(uc)); asm volatile(ins : "+v"(u2) : "v"(ub), "v"(uc)); asm volatile(ins : "+v"(u3) : "v"(ub), "v"(uc));)

template <int OP>
__global__ void k(float *out, unsigned long long *stamps, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f, c = 0.5f;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, ub = 7, uc = 3;
    unsigned long long m = 0x5555555555555555ull;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { F4("v_fma_f32 %0, %0, %1, %2") }
        if (OP == 1) { US4("v_cndmask_b32_e64 %0, %0, %1, %2") }
        if (OP == 2) { U4("v_mad_i32_i24 %0, %0, %1, %2") }
        if (OP == 3) { F4("v_rcp_f32 %0, %0") }
        if (OP == 4) { U4("v_pk_mad_u16 %0, %0, %1, %2") }
        if (OP == 5) { U4("v_add_u32 %0, %0, %1") }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(u0 + u1 + u2 + u3) + (float)m;
}

template <int OP>
void run(const char *name, float *d, unsigned long long *ds, int waves_per_simd, int iters)
{
    const int blocks = 256 * 4 * waves_per_simd;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, ds, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, ds, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(2 * (size_t)blocks);
    hipMemcpy(h.data(), ds, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> mhz, cyc;
    for (int i = 0; i < blocks; ++i) { mhz.push_back((double)h[2 * i] / ((double)h[2 * i + 1] / 100.0)); cyc.push_back((double)h[2 * i]); }
    std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end());
    const double insts_per_simd = (double)iters * 64 * waves_per_simd;
    const double med_cyc = cyc[cyc.size() / 2];
    printf("%-22s waves/SIMD=%d  %.3f ns/inst/SIMD (HIP events, %.3f ms)   shader clock %.0f MHz (median of the waves; p5 %.0f, p95 %.0f)   %.2f cycles/inst/SIMD (a wave's loop: %.0f cycles)\n",
           name, waves_per_simd, ms * 1e6 / insts_per_simd, ms, mhz[mhz.size() / 2], mhz[mhz.size() / 20], mhz[mhz.size() * 19 / 20], med_cyc / insts_per_simd, med_cyc);
    fflush(stdout);
}

extern "C" int ubench_main()
{
    float *d; unsigned long long *ds;
    if (hipMalloc(&d, 256 * 4 * 8 * 64 * 4) != hipSuccess || hipMalloc(&ds, 256 * 4 * 8 * 2 * 8) != hipSuccess) return 1;
    // short (a kernel of ~60 us: clocks as a 230 us raster kernel finds them) and long (several ms: the sustained state)
    for (int iters : {50, 200, 4000}) {
        printf("-- %d iterations of 64 instructions per wave\n", iters);
        for (int w : {8, 6, 1}) {
            run<0>("v_fma_f32", d, ds, w, iters); run<5>("v_add_u32", d, ds, w, iters); run<1>("v_cndmask_b32(sgpr)", d, ds, w, iters); run<2>("v_mad_i32_i24", d, ds, w, iters);
            run<4>("v_pk_mad_u16", d, ds, w, iters); run<3>("v_rcp_f32", d, ds, w, iters);
        }
    }
    return 0;
}
