// Probe (not the product): does v_mfma_f32_32x32x16_f16 (a) take f16 SUBNORMAL inputs at their exact value and (b) return an
// exact f32 sum when every product is a multiple of 2^-12 and all partial sums stay below 2^11?  Both are what an f16 limb
// split of the weight kernel's pair dot products needs (weights.hip).  Build: hipcc --offload-arch=gfx950 -O2 -o probe this.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A: 32 x 16 (row r, k), B: 16 x 32 (k, col c) given as Bt[c][k]; D[r][c] = sum_k A[r][k] B[k][c], three chained MFMAs into one acc
__global__ void k(const _Float16* A0, const _Float16* B0, const _Float16* A1, const _Float16* B1, float* D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    h8 a0, b0, a1, b1;
    for (int j = 0; j < 8; j++) {
        a0[j] = A0[r * 16 + 8 * h + j]; b0[j] = B0[r * 16 + 8 * h + j];
        a1[j] = A1[r * 16 + 8 * h + j]; b1[j] = B1[r * 16 + 8 * h + j];
    }
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
    // D layout of 32x32: lane (col = lane & 31, half h); register i -> row 8 (i / 4) + 4 h ... : write via the generic rule
    for (int i = 0; i < 16; i++) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h, col = r;
        D[row * 32 + col] = acc[i];
    }
}

int main() {
    const int n = 32 * 16;
    _Float16 *A0, *B0, *A1, *B1; float* D;
    hipMallocManaged(&A0, n * 2); hipMallocManaged(&B0, n * 2); hipMallocManaged(&A1, n * 2); hipMallocManaged(&B1, n * 2);
    hipMallocManaged(&D, 32 * 32 * 4);
    srand(7);
    int bad_exact = 0, bad_sub = 0;
    double maxerr = 0;
    for (int trial = 0; trial < 200; trial++) {
        // pair 0: h0 x h0' (multiples of 2^-6 up to 10): exact sum expected.  pair 1: subnormal h1 (multiples of 2^-17 up to 2^-7)
        // against h0' : products multiples of 2^-23
        double a0[n], b0[n], a1[n], b1[n];
        for (int i = 0; i < n; i++) {
            a0[i] = (rand() % 1281 - 640) / 64.0; b0[i] = (rand() % 1281 - 640) / 64.0;
            const int m = (trial & 1) ? (rand() % 15 - 7) : (rand() % 2049 - 1024);      // odd trials: |m| < 8 -> subnormal f16 only
            a1[i] = m / 131072.0; b1[i] = (rand() % 1281 - 640) / 64.0;
            A0[i] = (_Float16)a0[i]; B0[i] = (_Float16)b0[i]; A1[i] = (_Float16)a1[i]; B1[i] = (_Float16)b1[i];
            if ((double)A1[i] != a1[i] || (double)A0[i] != a0[i]) { printf("host conversion inexact\n"); return 1; }
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A0, B0, A1, B1, D);
        hipDeviceSynchronize();
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                double e0 = 0, e1 = 0;
                for (int kk = 0; kk < 16; kk++) { e0 += a0[r * 16 + kk] * b0[c * 16 + kk]; e1 += a1[r * 16 + kk] * b1[c * 16 + kk]; }
                const double ex = e0 + e1;                       // exact in double
                const double got = D[r * 32 + c];
                const double err = fabs(got - ex);
                // the f32 result can hold e0 exactly (multiple of 2^-12 below 2^11) but e0 + e1 needs more bits: compare with the
                // correctly rounded f32 of the exact value
                if ((float)ex != D[r * 32 + c]) bad_exact++;
                if (err > maxerr) maxerr = err;
                if (e1 != 0 && fabs(got - e0) < 1e-12 && fabs(e1) > 1e-4) bad_sub++;      // the subnormal operand was flushed
            }
    }
    printf("results != correctly rounded f32 of the exact sum: %d of %d; max |err| %.3e; sums where the subnormal limb vanished: %d\n",
           bad_exact, 200 * 1024, maxerr, bad_sub);
    // pure exactness test: only pair 0 (second pair zero)
    int bad0 = 0;
    for (int trial = 0; trial < 100; trial++) {
        double a0[n], b0[n];
        for (int i = 0; i < n; i++) {
            a0[i] = (rand() % 1281 - 640) / 64.0; b0[i] = (rand() % 1281 - 640) / 64.0;
            A0[i] = (_Float16)a0[i]; B0[i] = (_Float16)b0[i]; A1[i] = (_Float16)0.0; B1[i] = (_Float16)0.0;
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A0, B0, A1, B1, D);
        hipDeviceSynchronize();
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                double e0 = 0;
                for (int kk = 0; kk < 16; kk++) e0 += a0[r * 16 + kk] * b0[c * 16 + kk];
                if ((double)D[r * 32 + c] != e0) bad0++;
            }
    }
    printf("h0 x h0' sums (multiples of 2^-12, |sum| <= 1600) not exact: %d of %d\n", bad0, 100 * 1024);
    return 0;
}
