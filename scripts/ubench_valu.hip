// Issue rates of the vector instructions the weight kernel's exponential can be built from (gfx950), at 1 / 2 / 3 / 4 waves
// per SIMD: cycles per wave-instruction seen by one SIMD.  Guided the f32 exponent path of k_kde_split.
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// 16 independent instructions per iteration on registers the asm owns
#define KERNEL32(name, INSTR)                                                                    \
    __global__ __launch_bounds__(256) void name(int iters, float* out) {                         \
        float a[16];                                                                             \
        for (int i = 0; i < 16; i++) a[i] = 1.0f + threadIdx.x * 1e-3f + i;                      \
        float b = 0.999f + threadIdx.x * 1e-6f, c = 1e-3f;                                       \
        for (int it = 0; it < iters; it++) {                                                     \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(INSTR : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                        \
        float s = 0;                                                                             \
        for (int i = 0; i < 16; i++) s += a[i];                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                                 \
    }
#define KERNEL64(name, INSTR)                                                                    \
    __global__ __launch_bounds__(256) void name(int iters, float* out) {                         \
        double a[16];                                                                            \
        for (int i = 0; i < 16; i++) a[i] = 1.0 + threadIdx.x * 1e-3 + i;                        \
        double b = 0.999 + threadIdx.x * 1e-6, c = 1e-3;                                         \
        int e = (int)(threadIdx.x & 1);                                                          \
        for (int it = 0; it < iters; it++) {                                                     \
            _Pragma("unroll") for (int i = 0; i < 16; i++) asm volatile(INSTR : "+v"(a[i]) : "v"(b), "v"(c), "v"(e)); \
        }                                                                                        \
        double s = 0;                                                                            \
        for (int i = 0; i < 16; i++) s += a[i];                                                  \
        out[blockIdx.x * 256 + threadIdx.x] = (float)s;                                          \
    }

KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %2")
KERNEL32(k_fract_f32, "v_fract_f32 %0, %0")
KERNEL32(k_floor_f32, "v_floor_f32 %0, %0")
KERNEL32(k_cvt_flr, "v_cvt_flr_i32_f32 %0, %0")
KERNEL32(k_cvt_i32, "v_cvt_i32_f32 %0, %0")
KERNEL32(k_rndne_f32, "v_rndne_f32 %0, %0")
KERNEL32(k_ldexp_f32, "v_ldexp_f32 %0, %0, %2")
KERNEL32(k_lshl_add, "v_lshl_add_u32 %0, %0, 1, %1")
KERNEL32(k_exp_f32, "v_exp_f32 %0, %0")
KERNEL32(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL64(k_ldexp_f64, "v_ldexp_f64 %0, %0, %3")
// conversions change the register width: dedicated kernels
__global__ __launch_bounds__(256) void k_cvt_f64_f32(int iters, float* out) {
    float a[16];
    double d[16];
    for (int i = 0; i < 16; i++) a[i] = 1.0f + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
    }
    double s = 0;
    for (int i = 0; i < 16; i++) s += d[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
// the candidate per-pair sequence: floor / fract / add / 6 fma (f32) + cvt + ldexp + add (f64); 4 pairs per iteration
__global__ __launch_bounds__(256) void k_seq_f32path(int iters, float* out) {
    float X[4], Y[4];
    double s = 0.0;
    for (int i = 0; i < 4; i++) { X[i] = -3.25f - threadIdx.x * 0.0625f - i; Y[i] = 0.01f * i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int n; float fr, g, p; double pd;
            asm volatile("v_cvt_flr_i32_f32 %0, %1" : "=v"(n) : "v"(X[i]));
            asm volatile("v_fract_f32 %0, %1" : "=v"(fr) : "v"(X[i]));
            asm volatile("v_add_f32 %0, %1, %2" : "=v"(g) : "v"(fr), "v"(Y[i]));
            p = 1.5e-4f;
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(g), "v"(1.3e-3f));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(g), "v"(9.6e-3f));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(g), "v"(5.5e-2f));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(g), "v"(0.24f));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(g), "v"(0.69f));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(g), "v"(1.0f));
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(pd) : "v"(p));
            asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(pd) : "v"(n));
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(pd));
            asm volatile("" : "+v"(X[i]), "+v"(Y[i]));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
// today's per-pair sequence: 2 cvt, 3 add, 6 fma, ldexp, add -- all f64
__global__ __launch_bounds__(256) void k_seq_f64path(int iters, float* out) {
    float X[4], Y[4];
    double s = 0.0;
    for (int i = 0; i < 4; i++) { X[i] = -3.25f - threadIdx.x * 0.0625f - i; Y[i] = 0.01f * i; }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double x = (double)X[i] + (double)Y[i];
            const double tm = x + 6755399441055744.0;
            const double f = x - (tm - 6755399441055744.0);
            double p = 0x1.41d333a1fbff9p-13;
            p = fma(p, f, 0x1.5f456a867c735p-10);
            p = fma(p, f, 0x1.3b2dbbc0aa7a3p-7);
            p = fma(p, f, 0x1.c6aed4b95c606p-5);
            p = fma(p, f, 0x1.ebfbdadcb136fp-3);
            p = fma(p, f, 0x1.62e430c7e91afp-1);
            p = fma(p, f, 0x1.00000002614ffp+0);
            s += ldexp(p, __double2loint(tm));
            asm volatile("" : "+v"(X[i]), "+v"(Y[i]));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

template <typename F>
static int run(const char* name, F kern, int per_iter, float* out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    int clk_khz = 0;
    CK(hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0));
    printf("%-16s", name);
    for (int wps = 1; wps <= 4; wps++) {
        const int iters = 20000;
        // 256 CUs x wps work-groups of 4 waves: wps waves per SIMD
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, 100, out);
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(256 * wps), dim3(256), 0, 0, iters, out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, a, b));
        // wave-instructions per SIMD: wps waves x iters x per_iter
        const double ns_per = 1e6 * ms / ((double)wps * iters * per_iter);
        printf("  %dw/SIMD %6.3f ns (%5.2f cyc @2.4GHz)", wps, ns_per, ns_per * 2.4);
    }
    printf("\n");
    return 0;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 256 * 4 * 256 * sizeof(float)));
#define R(k) if (run(#k, k, 16, out)) return 1
    R(k_fma_f32); R(k_add_f32); R(k_fract_f32); R(k_floor_f32); R(k_cvt_flr); R(k_cvt_i32); R(k_rndne_f32); R(k_ldexp_f32);
    R(k_lshl_add); R(k_exp_f32); R(k_max_f32); R(k_pk_fma_f32); R(k_pk_add_f32); R(k_fma_f64); R(k_add_f64); R(k_ldexp_f64);
    R(k_cvt_f64_f32);
    if (run("seq_f32path/pair", k_seq_f32path, 4, out)) return 1;
    if (run("seq_f64path/pair", k_seq_f64path, 4, out)) return 1;
    CK(hipDeviceSynchronize());
    return 0;
}
