// Which XCD does workgroup b of a launch run on?  HW_REG_XCC_ID of 64 workgroups (tools/ubench: hipcc xcc_probe.hip -o /tmp/xcc && /tmp/xcc)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
int main() {
    unsigned *d, h[4096];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int i = 0; i < 32; ++i) printf("%x ", h[i]);
    int bad = 0; for (int i = 0; i < 4096; ++i) bad += (h[i] & 15u) != (unsigned)(i & 7);
    printf("\nraw[0]=%08x; workgroups whose XCC_ID & 15 differs from b %% 8: %d of 4096\n", h[0], bad);
    return 0;
}
