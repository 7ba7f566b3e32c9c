// Micro-benchmark, round 4: issue cost of the VALU instructions the raster kernels are made of, each as four
// independent dependency chains per wave at 8 waves per SIMD (ns per wave64 instruction per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC valu_rates2.hip -o libvalu_rates2.so ; run: python tools/ubench/run.py 2
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
// four chains on a0..a3 (float) or u0..u3 (uint)
#define F4(ins) REP16(asm volatile(ins : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ins : "+v"(a1) : "v"(b), "v"(c)); asm volatile(ins : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ins : "+v"(a3) : "v"(b), "v"(c));)
#define U4(ins) REP16(asm volatile(ins : "+v"(u0) : "v"(ub), "v"(uc)); asm volatile(ins : "+v"(u1) : "v"(ub), "v"(uc)); asm volatile(ins : "+v"(u2) : "v"(ub), "v"(uc)); asm volatile(ins : "+v"(u3) : "v"(ub), "v"(uc));)
#define US4(ins) REP16(asm volatile(ins : "+v"(u0) : "v"(ub), "s"(m)); asm volatile(ins : "+v"(u1) : "v"(ub), "s"(m)); asm volatile(ins : "+v"(u2) : "v"(ub), "s"(m)); asm volatile(ins : "+v"(u3) : "v"(ub), "s"(m));)
#define C4(ins) REP16(asm volatile(ins : "=s"(m) : "v"(u0), "v"(ub)); asm volatile(ins : "=s"(m2) : "v"(u1), "v"(ub)); asm volatile(ins : "=s"(m) : "v"(u2), "v"(ub)); asm volatile(ins : "=s"(m2) : "v"(u3), "v"(ub));)
#define D4(ins) REP16(asm volatile(ins : "+v"(d0) : "v"(d2)); asm volatile(ins : "+v"(d1) : "v"(d2)); asm volatile(ins : "+v"(d0) : "v"(d3)); asm volatile(ins : "+v"(d1) : "v"(d3));)

template <int OP>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned u0 = threadIdx.x, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, ub = 7, uc = 3;
    double d0 = a0, d1 = a1, d2 = 1.0001, d3 = 0.999;
    float b = 1.0001f, c = 0.5f;
    unsigned long long m = 0x5555555555555555ull, m2 = 0;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { F4("v_fma_f32 %0, %0, %1, %2") }
        if (OP == 1) { F4("v_mul_f32 %0, %0, %1") }
        if (OP == 2) { F4("v_add_f32 %0, %0, %1") }
        if (OP == 3) { F4("v_max_f32 %0, %0, %1") }
        if (OP == 4) { F4("v_med3_f32 %0, %0, %1, %2") }
        if (OP == 5) { US4("v_cndmask_b32_e64 %0, %0, %1, %2") }
        if (OP == 6) { U4("v_mov_b32 %0, %1") }
        if (OP == 7) { U4("v_mov_b32_dpp %0, %1 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf") }
        if (OP == 8) { U4("v_and_b32 %0, %0, %1") }
        if (OP == 9) { U4("v_add_u32 %0, %0, %1") }
        if (OP == 10) { U4("v_lshlrev_b32 %0, 3, %0") }
        if (OP == 11) { U4("v_bfe_u32 %0, %0, %1, %2") }
        if (OP == 12) { U4("v_perm_b32 %0, %0, %1, %2") }
        if (OP == 13) { U4("v_mad_i32_i24 %0, %0, %1, %2") }
        if (OP == 14) { U4("v_mul_i32_i24 %0, %0, %1") }
        if (OP == 15) { U4("v_pk_mad_u16 %0, %0, %1, %2") }
        if (OP == 16) { U4("v_pk_lshrrev_b16 %0, 8, %0") }
        if (OP == 17) { U4("v_pk_add_u16 %0, %0, %1") }
        if (OP == 18) { C4("v_cmp_gt_i32_e64 %0, %1, %2") }
        if (OP == 19) { C4("v_cmp_gt_f32_e64 %0, %1, %2") }
        if (OP == 20) { US4("v_addc_co_u32_e64 %0, vcc, %0, %0, %2") }
        if (OP == 21) { F4("v_cvt_f32_i32 %0, %0") }
        if (OP == 22) { F4("v_cvt_i32_f32 %0, %0") }
        if (OP == 23) { F4("v_rndne_f32 %0, %0") }
        if (OP == 24) { F4("v_floor_f32 %0, %0") }
        if (OP == 25) { F4("v_ldexp_f32 %0, %0, 3") }
        if (OP == 26) { F4("v_rcp_f32 %0, %0") }
        if (OP == 27) { D4("v_pk_fma_f32 %0, %0, %1, %1") }
        if (OP == 28) { D4("v_pk_mul_f32 %0, %0, %1") }
        if (OP == 29) { D4("v_pk_add_f32 %0, %0, %1") }
        if (OP == 30) { U4("v_min_u32 %0, %0, %1") }
        if (OP == 31) { U4("v_lshl_add_u32 %0, %0, 2, %1") }
        if (OP == 32) { U4("v_lshl_or_b32 %0, %0, 16, %1") }
        if (OP == 33) { U4("v_and_or_b32 %0, %0, %1, %2") }
        if (OP == 34) { U4("v_mul_lo_u32 %0, %0, %1") }
        if (OP == 35) { U4("v_cvt_f32_ubyte0 %0, %0") }
        if (OP == 36) { U4("v_readlane_b32 s20, %0, 3") }
        if (OP == 37) { F4("v_fmac_f32 %0, %1, %2") }
        if (OP == 38) { F4("v_sub_f32 %0, %0, %1") }
        if (OP == 39) { U4("v_min3_u32 %0, %0, %1, %2") }
        if (OP == 40) { U4("v_bfi_b32 %0, %1, %2, %0") }
        if (OP == 41) { U4("v_cvt_pk_u8_f32 %0, %1, 0, %0") }
        if (OP == 42) { F4("v_frexp_mant_f32 %0, %0") }
        if (OP == 43) { U4("v_ashrrev_i32 %0, 8, %0") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(u0 + u1 + u2 + u3) + (float)(m + m2) + (float)(d0 + d1);
}

template <int OP>
void run(const char *name, float *d, int waves_per_simd)
{
    const int iters = 1000;
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
    const double ns = ms * 1e6 / insts_per_simd;
    printf("%-22s waves/SIMD=%d  %.3f ns/inst/SIMD\n", name, waves_per_simd, ns);
    fflush(stdout);
}

extern "C" int ubench_main()
{
    float *d; hipError_t st = hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
    printf("malloc: %d\n", (int)st); fflush(stdout);
    for (int w : {8, 2}) {
        run<0>("v_fma_f32", d, w); run<37>("v_fmac_f32", d, w); run<1>("v_mul_f32", d, w); run<2>("v_add_f32", d, w); run<38>("v_sub_f32", d, w); run<3>("v_max_f32", d, w); run<4>("v_med3_f32", d, w);
        run<5>("v_cndmask_b32(sgpr)", d, w); run<6>("v_mov_b32", d, w); run<7>("v_mov_b32_dpp", d, w); run<8>("v_and_b32", d, w); run<9>("v_add_u32", d, w);
        run<10>("v_lshlrev_b32", d, w); run<43>("v_ashrrev_i32", d, w); run<11>("v_bfe_u32", d, w); run<12>("v_perm_b32", d, w); run<40>("v_bfi_b32", d, w); run<13>("v_mad_i32_i24", d, w); run<14>("v_mul_i32_i24", d, w);
        run<15>("v_pk_mad_u16", d, w); run<16>("v_pk_lshrrev_b16", d, w); run<17>("v_pk_add_u16", d, w); run<18>("v_cmp_gt_i32->s", d, w); run<19>("v_cmp_gt_f32->s", d, w);
        run<20>("v_addc_co_u32", d, w); run<21>("v_cvt_f32_i32", d, w); run<22>("v_cvt_i32_f32", d, w); run<23>("v_rndne_f32", d, w); run<24>("v_floor_f32", d, w);
        run<25>("v_ldexp_f32", d, w); run<42>("v_frexp_mant_f32", d, w); run<26>("v_rcp_f32", d, w); run<27>("v_pk_fma_f32", d, w); run<28>("v_pk_mul_f32", d, w); run<29>("v_pk_add_f32", d, w);
        run<30>("v_min_u32", d, w); run<39>("v_min3_u32", d, w); run<31>("v_lshl_add_u32", d, w); run<32>("v_lshl_or_b32", d, w); run<33>("v_and_or_b32", d, w); run<34>("v_mul_lo_u32", d, w);
        run<35>("v_cvt_f32_ubyte0", d, w); run<41>("v_cvt_pk_u8_f32", d, w); run<36>("v_readlane_b32", d, w);
    }
    return 0;
}
