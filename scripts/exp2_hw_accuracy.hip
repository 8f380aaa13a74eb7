// Accuracy of the hardware 2^x (v_exp_f32) on the argument range of the split-operand weight kernel, g = fract(X) + Y in
// [-0.3, 1.3], against exp2 in double: every float in the range in steps of `stride` ulps.  Decides whether one
// transcendental can replace the six-FMA polynomial of ks_exp2_f32 (abcsmc_amd/csrc/weights.hip).
//   hipcc --offload-arch=gfx950 -O3 scripts/exp2_hw_accuracy.hip -o /tmp/exp2_hw && /tmp/exp2_hw
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float* y, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = __builtin_amdgcn_exp2f(x[i]);
}
int main() {
    std::vector<float> xs;
    for (float v = -0.3f; v <= 1.3f; v = std::nextafterf(v + 3e-7f, 2.0f)) xs.push_back(v);
    for (int i = 0; i <= 4096; i++) xs.push_back((float)i / 4096.0f);       // exact fractions with Y = 0
    const size_t n = xs.size();
    float *dx, *dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4);
    hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dy, n);
    std::vector<float> ys(n);
    hipMemcpy(ys.data(), dy, n * 4, hipMemcpyDeviceToHost);
    double mx = 0, sq = 0, mean = 0; float worst = 0;
    for (size_t i = 0; i < n; i++) {
        const double ref = std::exp2((double)xs[i]);
        const double e = (double)ys[i] / ref - 1.0;
        sq += e * e; mean += e;
        if (std::fabs(e) > mx) { mx = std::fabs(e); worst = xs[i]; }
    }
    printf("v_exp_f32 on [-0.3, 1.3], %zu points: max rel err %.3e (at %.7f), rms %.3e, mean %.3e\n", n, mx, worst, std::sqrt(sq / n), mean / n);
    return 0;
}
