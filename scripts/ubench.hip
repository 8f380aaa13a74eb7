// Micro-benchmarks that guided the k_gram / k_project design (run on the MI355X box):
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// (1) contiguous 16-B/lane read, grid-stride
__global__ void k_read_linear(const d2* __restrict__ p, size_t n2, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) { d2 v = p[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
// (2) column-major tile pattern: C columns of N rows; a WG reads tiles of TR rows x C columns (1 KiB per column per wave-instr)
template <int C, bool CONTIG>
__global__ __launch_bounds__(256) void k_read_tiles(const double* __restrict__ X, size_t N, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ntiles = N / 128;
    const size_t per = (ntiles + gridDim.x - 1) / gridDim.x;
    double s = 0;
    for (size_t k = 0; k < per; k++) {
        const size_t tile = CONTIG ? blockIdx.x * per + k : k * gridDim.x + blockIdx.x;
        if (tile >= ntiles) break;
        const size_t r = tile * 128 + 2 * lane;
#pragma unroll
        for (int i = 0; i < C / 4; i++) { d2 v = *(const d2*)(X + (size_t)(wave + 4 * i) * N + r); s += v.x + v.y; }
    }
    if (s == 1.2345e300) out[0] = s;
}
// (3) MFMA f64 issue rate: NACC independent accumulators, back to back
template <int NACC>
__global__ __launch_bounds__(256) void k_mfma_rate(int iters, double* out) {
    d4 acc[NACC];
    for (int a = 0; a < NACC; a++) acc[a] = (d4){0, 0, 0, 0};
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
    }
    double s = 0;
    for (int a = 0; a < NACC; a++) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// (4) f64 VALU FMA rate: 16 independent accumulators per lane
__global__ __launch_bounds__(256) void k_fma_rate(int iters, double* out) {
    double acc[16];
    for (int a = 0; a < 16; a++) acc[a] = threadIdx.x * 1e-9 + a;
    const double x = 1.0 + threadIdx.x * 1e-12, y = 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int a = 0; a < 16; a++) acc[a] = fma(acc[a], x, y);
    }
    double s = 0;
    for (int a = 0; a < 16; a++) s += acc[a];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// (5) both pipes: waves 0-1 of each 512-thread WG issue MFMA, waves... (even waves MFMA, odd waves VALU), 2 waves per SIMD
__global__ __launch_bounds__(512) void k_mixed_rate(int iters, double* out) {
    const int wave = threadIdx.x >> 6;
    double s = 0;
    if (wave < 4) {
        d4 acc[6];
        for (int a = 0; a < 6; a++) acc[a] = (d4){0, 0, 0, 0};
        double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int a = 0; a < 6; a++) acc[a] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[a], 0, 0, 0);
        }
        for (int a = 0; a < 6; a++) s += acc[a][0] + acc[a][3];
    } else {
        double acc[16];
        for (int a = 0; a < 16; a++) acc[a] = threadIdx.x * 1e-9 + a;
        const double x = 1.0 + threadIdx.x * 1e-12, y = 1e-9 * threadIdx.x;
        for (int it = 0; it < iters * 12; it++) {
#pragma unroll
            for (int a = 0; a < 16; a++) acc[a] = fma(acc[a], x, y);
        }
        for (int a = 0; a < 16; a++) s += acc[a];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    const size_t N = 1 << 20, C = 48;
    double* X; double* out;
    CK(hipMalloc(&X, N * C * 8)); CK(hipMalloc(&out, 1 << 22));
    CK(hipMemset(X, 0, N * C * 8));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time = [&](auto f, const char* name, double bytes) {
        for (int i = 0; i < 3; i++) f();
        hipEventRecord(a); for (int i = 0; i < 10; i++) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        printf("%-44s %8.1f us  %8.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6);
    };
    const double bytes = (double)N * C * 8;
    for (int blocks : {512, 1024, 2048, 4096})
        time([&] { hipLaunchKernelGGL(k_read_linear, dim3(blocks), dim3(256), 0, 0, (const d2*)X, N * C / 2, out); },
             (std::string("linear 16B/lane blocks=") + std::to_string(blocks)).c_str(), bytes);
    for (int blocks : {256, 512, 768, 1024, 2048}) {
        time([&] { hipLaunchKernelGGL((k_read_tiles<48, false>), dim3(blocks), dim3(256), 0, 0, X, N, out); },
             (std::string("tiles 48col strided  blocks=") + std::to_string(blocks)).c_str(), bytes);
        time([&] { hipLaunchKernelGGL((k_read_tiles<48, true>), dim3(blocks), dim3(256), 0, 0, X, N, out); },
             (std::string("tiles 48col contig   blocks=") + std::to_string(blocks)).c_str(), bytes);
    }
    // MFMA rate: 256 CUs x 4 waves, 1 wave / SIMD
    for (int nacc : {1, 2, 6}) {
        const int iters = 2000;
        auto f = [&] {
            if (nacc == 1) hipLaunchKernelGGL(k_mfma_rate<1>, dim3(256), dim3(256), 0, 0, iters, out);
            if (nacc == 2) hipLaunchKernelGGL(k_mfma_rate<2>, dim3(256), dim3(256), 0, 0, iters, out);
            if (nacc == 6) hipLaunchKernelGGL(k_mfma_rate<6>, dim3(256), dim3(256), 0, 0, iters, out);
        };
        for (int i = 0; i < 2; i++) f();
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double nm = (double)iters * nacc;   // MFMAs per wave (1 wave per SIMD)
        printf("mfma_f64_16x16x4 nacc=%d: %.1f us -> %.1f ns per MFMA per SIMD (= %.0f cycles @2.4GHz), %.1f TFLOP/s\n", nacc,
               ms * 1e3, ms * 1e6 / nm, ms * 1e6 / nm * 2.4, nm * 1024 * 2048 / ms / 1e9);
    }
    // MFMA rate vs waves per SIMD (independent accumulators, 6 per wave)
    for (int wg : {256, 512, 1024, 2048}) {
        const int iters = 1000;
        auto f = [&] { hipLaunchKernelGGL(k_mfma_rate<6>, dim3(wg), dim3(256), 0, 0, iters, out); };
        for (int i = 0; i < 2; i++) f();
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double fl = (double)wg * 4 * iters * 6 * 2048;
        printf("mfma_f64_16x16x4, %d waves/SIMD: %.1f us -> %.1f TFLOP/s\n", wg / 256, ms * 1e3, fl / ms / 1e9);
    }
    for (int wg : {256, 512, 1024}) {   // 1, 2, 4 waves per SIMD
        const int iters = 4000;
        auto f = [&] { hipLaunchKernelGGL(k_fma_rate, dim3(wg), dim3(256), 0, 0, iters, out); };
        for (int i = 0; i < 2; i++) f();
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double fl = (double)wg * 256 * iters * 16 * 2;
        printf("v_fma_f64 %d WGs: %.1f us -> %.1f TFLOP/s\n", wg, ms * 1e3, fl / ms / 1e9);
    }
    {
        const int iters = 1000;
        auto f = [&] { hipLaunchKernelGGL(k_mixed_rate, dim3(256), dim3(512), 0, 0, iters, out); };
        for (int i = 0; i < 2; i++) f();
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        const double flm = 256.0 * 4 * iters * 6 * 2048, flv = 256.0 * 256 * iters * 12 * 16 * 2;
        printf("mixed (1 MFMA wave + 1 VALU wave per SIMD): %.1f us -> MFMA %.1f + VALU %.1f TFLOP/s (if fully overlapped)\n", ms * 1e3,
               flm / ms / 1e9, flv / ms / 1e9);
    }
    return 0;
}
