// Probe (not the product): is v_mfma_f64_16x16x4_f64, chained through its accumulator over K, bit for bit the sequential chain
// s = fma(a[k], b[k], s), k ascending?  (The projection's distances are compared bit for bit with the oracle's fma chain: only then
// could the score contraction of many-component models move to the matrix pipe.)
//   hipcc --offload-arch=gfx950 -O2 -o scripts/mfma_f64_probe scripts/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /* 16 x K row-major */, const double* B /* K x 16 row-major */, int K, double* D /* 16 x 16 */) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += 4) {
        const double a = A[r * K + k0 + q], b = B[(k0 + q) * 16 + r];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 4; i++) D[(q + 4 * i) * 16 + r] = acc[i];
}
int main() {
    const int K = 128;
    double *A, *B, *D;
    hipMallocManaged(&A, 16 * K * 8); hipMallocManaged(&B, K * 16 * 8); hipMallocManaged(&D, 256 * 8);
    srand(3);
    long bad = 0, tot = 0; double worst = 0;
    for (int trial = 0; trial < 200; trial++) {
        for (int i = 0; i < 16 * K; i++) {
            const double m = (rand() / (double)RAND_MAX - 0.5) * 2.0;
            A[i] = m * pow(2.0, (trial % 5 == 0) ? (rand() % 40 - 20) : (rand() % 4 - 2));
            B[i] = (rand() / (double)RAND_MAX - 0.5) * pow(2.0, (trial % 7 == 0) ? (rand() % 30 - 15) : 0);
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, K, D);
        hipDeviceSynchronize();
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 16; c++) {
                double s = 0.0;
                for (int kk = 0; kk < K; kk++) s = fma(A[r * K + kk], B[kk * 16 + c], s);
                tot++;
                if (s != D[r * 16 + c]) { bad++; const double e = fabs(s - D[r * 16 + c]) / fmax(fabs(s), 1e-300); if (e > worst) worst = e; }
            }
    }
    printf("v_mfma_f64_16x16x4_f64 chained over K = %d against the ascending fma chain: %ld of %ld results differ (worst relative difference %.3g)\n", K, bad, tot, worst);
    return 0;
}
