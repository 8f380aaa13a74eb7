// Probe (not the product): can the pair-sum kernel subtract the norm top AND the batch reference n in ONE bf16 MFMA step?
// Accumulator in: X' = h0.h0' (a multiple of 2^-14, |X'| <= 400, exact).  Products: -top (three bf16 pieces against -1, top a
// multiple of 2^-14 in [-100, 1100]) and -n (two bf16 pieces of the integer n = floor(max(X' - top)) against -1 entries, routed
// per half).  All addends are multiples of 2^-14 below 2^11 and the result X' - top - n is small: is it EXACT whatever the
// instruction's internal alignment does?   hipcc --offload-arch=gfx950 -O2 -o scripts/mfma_topn_probe scripts/mfma_topn_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    union { bf16x8 v; unsigned short s[8]; } a, b;
    for (int j = 0; j < 8; j++) { a.s[j] = A[r * 16 + 8 * h + j]; b.s[j] = B[r * 16 + 8 * h + j]; }
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    for (int i = 0; i < 16; i++) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}
static unsigned short bf16_rne(double v, double* back) {
    float f = (float)v; unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u;
    memcpy(&f, &u, 4); *back = f; return (unsigned short)(u >> 16);
}
int main() {
    unsigned short *A, *B; float *C, *D;
    hipMallocManaged(&A, 32 * 16 * 2); hipMallocManaged(&B, 32 * 16 * 2); hipMallocManaged(&C, 32 * 32 * 4); hipMallocManaged(&D, 32 * 32 * 4);
    srand(5);
    long inexact = 0, cnt = 0, near = 0; double worst = 0;
    for (int trial = 0; trial < 2000; trial++) {
        double X[32][32], top[32], nn[32][2];
        memset(A, 0, 32 * 16 * 2); memset(B, 0, 32 * 16 * 2);
        const double tbase = (trial % 4 == 0) ? 1100.0 : (rand() % 601 - 100);          // hbTop in [-100, 500], or the padding value
        for (int r = 0; r < 32; r++) {
            top[r] = rint((tbase == 1100.0 && (r & 1) ? 1100.0 : (rand() % 6001 - 1000) / 10.0 + (rand() % 16384) / 16384.0) * 16384.0) / 16384.0;
            for (int c = 0; c < 32; c++) X[r][c] = rint(((rand() % 8001 - 4000) / 10.0) * 16384.0 + rand() % 16384) / 16384.0, C[r * 32 + c] = (float)X[r][c];
        }
        for (int r = 0; r < 32; r++) for (int c = 0; c < 32; c++) if ((double)C[r * 32 + c] != X[r][c]) { printf("X not exact in f32\n"); return 1; }
        for (int c = 0; c < 32; c++)
            for (int h = 0; h < 2; h++) {
                double m = -1e30;
                for (int r = 0; r < 32; r++) if (((r >> 2) & 1) == h && X[r][c] - top[r] > m) m = X[r][c] - top[r];
                nn[c][h] = floor(m);
            }
        for (int r = 0; r < 32; r++) {                                       // A: previous particle r: top pieces in slots 0..2, -1 in 6,7 / 14,15
            double rem = top[r], back;
            for (int kk = 0; kk < 3; kk++) { A[r * 16 + kk] = bf16_rne(rem, &back); rem -= back; }
            if (rem != 0.0) { printf("top does not fit three pieces\n"); return 1; }
            const int s0 = ((r >> 2) & 1) ? 14 : 6;
            A[r * 16 + s0] = 0xBF80; A[r * 16 + s0 + 1] = 0xBF80;
        }
        for (int c = 0; c < 32; c++) {                                       // B: new particle c: -1 against the top pieces, n pieces
            for (int kk = 0; kk < 3; kk++) B[c * 16 + kk] = 0xBF80;
            for (int h = 0; h < 2; h++) {
                const int s0 = h ? 14 : 6;
                double back, t = nn[c][h];
                B[c * 16 + s0] = bf16_rne(t, &back);
                const double rest = t - back;
                B[c * 16 + s0 + 1] = bf16_rne(rest, &back);
                if (rest != back) { printf("n = %g does not fit two pieces\n", t); return 1; }
            }
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, D);
        hipDeviceSynchronize();
        for (int r = 0; r < 32; r++)
            for (int c = 0; c < 32; c++) {
                const int h = (r >> 2) & 1;
                const double ex = X[r][c] - top[r] - nn[c][h];               // a multiple of 2^-14, <= 1
                cnt++;
                if ((double)D[r * 32 + c] != ex) {
                    // exact is only required where the result can be represented: |ex| < 2^10
                    if (fabs(ex) < 1024.0) { inexact++; const double e = fabs(D[r * 32 + c] - ex); if (e > worst) worst = e; }
                    if (fabs(ex) < 128.0) near++;                            // the terms that carry a sum lie within 2^-126 of its largest
                }
            }
    }
    printf("X' - top - n in one bf16 MFMA: %ld of %ld results (|result| < 2^10) not exact, %ld of them with |result| < 128; largest deviation %.3g\n", inexact, cnt, near, worst);
    return 0;
}
