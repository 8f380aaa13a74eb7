// Which SIMD do the eight waves of a 512-thread work-group land on?  (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13])
//   hipcc --offload-arch=gfx950 -O2 scripts/hwid_probe.hip -o scripts/hwid_probe && scripts/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d;
    hipMalloc(&d, 8 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(8), dim3(512), 0, 0, d);
    unsigned h[64];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; b++) {
        printf("block %d:", b);
        for (int w = 0; w < 8; w++) printf("  w%d simd %u slot %u cu %u", w, (h[b * 8 + w] >> 4) & 3, h[b * 8 + w] & 15, (h[b * 8 + w] >> 8) & 15);
        printf("\n");
    }
    return 0;
}
