// Probe (not the product): can the pair-sum kernel's "-n" step and its "norm low" step share ONE v_mfma_f32_32x32x16_bf16?
// The accumulator holds X (a multiple of 2^-14, |X| < 2^10: exact); the step adds -n (an integer near X, as two bf16 pieces
// against ones) AND -low (three bf16 pieces of a value below 2^-14 against -1).  Whether X - n - low comes out rounded ONCE at the
// small magnitude of the result depends on how the instruction orders / widens its internal sum: measured here for several
// K-slot placements.  Reports, per placement, the largest error in units of the result's ulp for results below 4, against the
// exactly computed value (double).   Build: hipcc --offload-arch=gfx950 -O2 -o probe scripts/mfma_merge_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: 32 rows x 16 K (bf16 bits), B: 32 columns x 16 K (bf16 bits), C/D: 32 x 32 f32 (row-major [row][col])
__global__ void k(const unsigned short* A, const unsigned short* B, const float* C, float* D, int two_steps,
                  const unsigned short* B2) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    union { bf16x8 v; unsigned short s[8]; } a, b, b2;
    for (int j = 0; j < 8; j++) { a.s[j] = A[r * 16 + 8 * h + j]; b.s[j] = B[r * 16 + 8 * h + j]; b2.s[j] = B2[r * 16 + 8 * h + j]; }
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = C[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
    if (two_steps) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b2.v, acc, 0, 0, 0);
    for (int i = 0; i < 16; i++) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
}

static unsigned short bf16_rne(double v, double* back) {
    float f = (float)v; unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u;
    memcpy(&f, &u, 4); *back = f; return (unsigned short)(u >> 16);
}

int main() {
    unsigned short *A, *B, *B2; float *C, *D;
    hipMallocManaged(&A, 32 * 16 * 2); hipMallocManaged(&B, 32 * 16 * 2); hipMallocManaged(&B2, 32 * 16 * 2);
    hipMallocManaged(&C, 32 * 32 * 4); hipMallocManaged(&D, 32 * 32 * 4);
    // placements: K-slots of (ones for rows with bit 2 clear: two slots), (ones for rows with bit 2 set), (the three low pieces)
    struct Pl { const char* name; int n0[2], n1[2], low[3]; } pls[] = {
        {"current slots: low 3,4,5  n 6,7 | 8,9      ", {6, 7}, {8, 9}, {3, 4, 5}},
        {"n first:       n 3,4 | 8,9  low 10,11,12   ", {3, 4}, {8, 9}, {10, 11, 12}},
        {"n first, same half: n 0,1 | 2,3  low 5,6,7 ", {0, 1}, {2, 3}, {5, 6, 7}},
        {"low first, other half: low 0,1,2 n 8,9|10,11", {8, 9}, {10, 11}, {0, 1, 2}},
        {"n last:        low 0,1,2  n 12,13 | 14,15  ", {12, 13}, {14, 15}, {0, 1, 2}},
        {"interleaved:   n 0,8 | 1,9  low 4,5,12     ", {0, 8}, {1, 9}, {4, 5, 12}},
    };
    srand(11);
    for (auto& pl : pls) {
        for (int two = 0; two < 2; two++) {
            double worst = 0, worst_big = 0; long inexact = 0, cnt = 0;
            for (int trial = 0; trial < 300; trial++) {
                double X[32][32], low[32], nn[32][2];
                memset(A, 0, 32 * 16 * 2); memset(B, 0, 32 * 16 * 2); memset(B2, 0, 32 * 16 * 2);
                const double base = (rand() % 1601 - 800);                        // where the batch sits
                for (int r = 0; r < 32; r++) {
                    low[r] = ((rand() % 2000001) - 1000000) * 1e-6 * 0x1p-15;      // |low| <= 2^-15
                    if (trial % 3 == 0) low[r] *= 3.0;                             // and a bit beyond
                    for (int c = 0; c < 32; c++) {
                        const double spread = (trial & 1) ? 40.0 : 3.0;
                        X[r][c] = rint((base - spread * (rand() % 10001) / 10000.0) * 16384.0) / 16384.0;
                        if (fabs(X[r][c]) >= 1000.0) X[r][c] = copysign(999.0, X[r][c]);
                        C[r * 32 + c] = (float)X[r][c];
                    }
                }
                for (int c = 0; c < 32; c++)
                    for (int h = 0; h < 2; h++) {
                        double m = -1e30;
                        for (int r = 0; r < 32; r++) if (((r >> 2) & 1) == h && X[r][c] > m) m = X[r][c];
                        nn[c][h] = floor(m);
                    }
                for (int r = 0; r < 32; r++) {                                       // A: previous particle r
                    const int* ones = ((r >> 2) & 1) ? pl.n1 : pl.n0;
                    A[r * 16 + ones[0]] = 0x3F80; A[r * 16 + ones[1]] = 0x3F80;
                    double rem = low[r], back;
                    for (int k = 0; k < 3; k++) { A[r * 16 + pl.low[k]] = bf16_rne(rem, &back); rem -= back; }
                    low[r] -= rem;                                                  // what the three pieces carry
                }
                for (int c = 0; c < 32; c++) {                                       // B: new particle c
                    unsigned short* Bn = two ? B : B;                               // step 1 (or the merged step): -n
                    unsigned short* Bl = two ? B2 : B;                              // low: the second step, or the same one
                    for (int h = 0; h < 2; h++) {
                        const int* sl = h ? pl.n1 : pl.n0;
                        double back, t = -nn[c][h];
                        Bn[c * 16 + sl[0]] = bf16_rne(t, &back);
                        // first piece: the leading eight bits (truncation as the kernel does would do as well); second: the rest
                        double rest = t - back;
                        Bn[c * 16 + sl[1]] = bf16_rne(rest, &back);
                        if (rest != back) { printf("n does not fit two pieces\n"); return 1; }
                    }
                    for (int k = 0; k < 3; k++) Bl[c * 16 + pl.low[k]] = 0xBF80;
                }
                hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, C, D, two, B2);
                hipDeviceSynchronize();
                for (int r = 0; r < 32; r++)
                    for (int c = 0; c < 32; c++) {
                        const int h = (r >> 2) & 1;
                        const double ex = X[r][c] - nn[c][h] - low[r];
                        const double got = D[r * 32 + c];
                        const double ulp = ldexp(1.0, ilogb(fmax(fabs(ex), 0x1p-126)) - 23);
                        const double e = fabs(got - ex) / ulp;
                        if (fabs(ex) < 4.0) { if (e > worst) worst = e; cnt++; if ((float)ex != D[r * 32 + c]) inexact++; }
                        else if (e > worst_big) worst_big = e;
                    }
            }
            printf("%s %s: |result| < 4: worst %.3f ulp, %ld of %ld not correctly rounded; larger results: worst %.3f ulp\n",
                   pl.name, two ? "TWO steps" : "ONE step ", worst, inexact, cnt, worst_big);
        }
    }
    return 0;
}
