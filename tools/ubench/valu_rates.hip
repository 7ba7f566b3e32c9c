// Micro-benchmark: issue cost (cycles per wave64 instruction per SIMD) of the VALU / SALU
// instructions the raster kernel is made of.  Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3;
    float b = 1.0001f, c = 0.5f;
    unsigned long long m = 0;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(b), "v"(c)); asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(b), "v"(c));) }
        if (OP == 1) { REP16(asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(a0), "v"(b)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(a1), "v"(b)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(a2), "v"(b)); asm volatile("v_cmp_gt_f32 %0, %1, %2" : "=s"(m) : "v"(a3), "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a0) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a1) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a2) : "v"(b)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a3) : "v"(b));) }
        if (OP == 3) { REP16(asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a0) : "v"(u0)); asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a1) : "v"(u1)); asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a2) : "v"(u2)); asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a3) : "v"(u3));) }
        if (OP == 4) { REP16(asm volatile("v_min_u32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u1) : "v"(u2)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u2) : "v"(u3)); asm volatile("v_min_u32 %0, %0, %1" : "+v"(u3) : "v"(u0));) }
        if (OP == 5) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&a2)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&a6)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a0) : "v"(*(double*)&a2)); asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&a4) : "v"(*(double*)&a6));) }
        if (OP == 6) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u0) : "v"(u1), "v"(u2)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u1) : "v"(u2), "v"(u3)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u2) : "v"(u3), "v"(u0)); asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u3) : "v"(u0), "v"(u1));) }
        if (OP == 7) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u1) : "v"(u2)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u2) : "v"(u3)); asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u3) : "v"(u0));) }
        if (OP == 8) { REP16(asm volatile("s_and_b64 %0, %0, exec" : "+s"(m)); asm volatile("s_and_b64 %0, %0, exec" : "+s"(m)); asm volatile("s_and_b64 %0, %0, exec" : "+s"(m)); asm volatile("s_and_b64 %0, %0, exec" : "+s"(m));) }
        if (OP == 9) { REP16(asm volatile("v_floor_f32 %0, %0" : "+v"(a0)); asm volatile("v_floor_f32 %0, %0" : "+v"(a1)); asm volatile("v_floor_f32 %0, %0" : "+v"(a2)); asm volatile("v_floor_f32 %0, %0" : "+v"(a3));) }
        if (OP == 10) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a1) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(a3) : "v"(b));) }
        if (OP == 11) { REP16(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u0) : "v"(a0)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u1) : "v"(a1)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u2) : "v"(a2)); asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u3) : "v"(a3));) }
        if (OP == 12) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u0) : "v"(u1)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u1) : "v"(u2)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u2) : "v"(u3)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u3) : "v"(u0));) }
        if (OP == 13) { REP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(a0)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a1)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a2)); asm volatile("v_rcp_f32 %0, %0" : "+v"(a3));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3) + (float)m;
}

template <int OP>
double run(const char *name, float *d, int waves_per_simd)
{
    const int iters = 2000;
    const int blocks = 256 * 4 * waves_per_simd;   // one wave per block
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double insts_per_simd = (double)iters * 64 * waves_per_simd;
    // cycles at an assumed 2.4 GHz; also report ns per instruction
    const double ns = ms * 1e6 / insts_per_simd;
    fflush(stdout);
    printf("%-16s waves/SIMD=%d  %.3f ns/inst/SIMD  (%.2f cyc @2.4GHz, %.2f @2.1GHz)\n", name, waves_per_simd, ns, ns * 2.4, ns * 2.1);
    fflush(stdout);
    return ns;
}

extern "C" int ubench_main()
{
    float *d; hipError_t st = hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    printf("malloc: %d\n", (int)st); fflush(stdout);
    for (int w : {8}) {
        run<0>("v_fma_f32", d, w); run<10>("v_add_f32", d, w); run<5>("v_pk_fma_f32", d, w); run<1>("v_cmp_gt_f32->s", d, w);
        run<2>("v_cndmask_b32", d, w); run<3>("v_cvt_f32_ubyte", d, w); run<11>("v_cvt_i32_f32", d, w); run<9>("v_floor_f32", d, w);
        run<4>("v_min_u32", d, w); run<6>("v_mad_u32_u24", d, w); run<12>("v_lshl_add_u32", d, w); run<7>("v_mul_lo_u32", d, w);
        run<13>("v_rcp_f32", d, w); run<8>("s_and_b64", d, w);
    }
    return 0;
}
