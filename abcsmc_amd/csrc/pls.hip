// PLS model from sufficient statistics: z-score moments, improved-kernel PLS2 deflation loop
// (Dayal & MacGregor "type 2": everything in M x M / M x P space, no further pass over the particles),
// PRESS on the validation statistics, component choice and observed scores.
// Replaces PLS::Model ctor, cv_NEW_DATA, optimal_num_components (reference call sites
// AbcUtil.cpp:432-449, 453).
//
// Latency-bound, tiny matrices: k_zstats is a grid of up to 128 256-thread work-groups (one for the simple model); k_pls_fit is ONE wavefront
// (no inter-wave barriers): XY and the P x P eigen work matrices live in LDS, XX stays in L2.
#include <stdlib.h>

#include <hip/hip_ext.h>
#include "abc_internal.h"

namespace {

// Wave reductions on the DPP path (no LDS crossbar: hipcc turns every __shfl_xor into two ds_bpermute_b32, ~100 cycles each
// on a dependent chain, and these kernels are nothing but dependent chains).  Inside a row of 16 lanes: quad_perm [1,0,3,2] and
// [2,3,0,1], then row_half_mirror and row_mirror (every lane of a quad / half row already holds the same partial result, so the
// mirrors act as xor 4 / xor 8); across the four rows: readlane of lanes 0, 16, 32, 48.  Additions commute, so every lane ends
// with the same bits.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    union { double d; int i[2]; } u, r;
    u.d = v;
    r.i[0] = dpp_i32<CTRL>(u.i[0]);
    r.i[1] = dpp_i32<CTRL>(u.i[1]);
    return r.d;
}
__device__ __forceinline__ double lane_f64(double v, int src_lane) {         // wave-uniform broadcast of one lane's value
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], src_lane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], src_lane);
    return u.d;
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_f64<0xB1>(v);
    v += dpp_f64<0x4E>(v);
    v += dpp_f64<0x141>(v);
    v += dpp_f64<0x140>(v);
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}
// arg-max over the wave of (key, index) with ties -> lowest index; `pay` travels with the winner
__device__ __forceinline__ void argmax_step(double& k, int& i, double& p, double ok, int oi, double op) {
    const bool take = (ok > k) | ((ok == k) & (oi < i));
    k = take ? ok : k; i = take ? oi : i; p = take ? op : p;
}
__device__ __forceinline__ void wave_argmax(double& key, int& idx, double& pay) {
    argmax_step(key, idx, pay, dpp_f64<0xB1>(key), dpp_i32<0xB1>(idx), dpp_f64<0xB1>(pay));
    argmax_step(key, idx, pay, dpp_f64<0x4E>(key), dpp_i32<0x4E>(idx), dpp_f64<0x4E>(pay));
    argmax_step(key, idx, pay, dpp_f64<0x141>(key), dpp_i32<0x141>(idx), dpp_f64<0x141>(pay));
    argmax_step(key, idx, pay, dpp_f64<0x140>(key), dpp_i32<0x140>(idx), dpp_f64<0x140>(pay));
    double k = lane_f64(key, 0), p = lane_f64(pay, 0);
    int i = __builtin_amdgcn_readlane(idx, 0);
#pragma unroll
    for (int row = 1; row < 4; row++)
        argmax_step(k, i, p, lane_f64(key, 16 * row), __builtin_amdgcn_readlane(idx, 16 * row), lane_f64(pay, 16 * row));
    key = k; idx = i; pay = p;
}

// zwork layout: [ XYtr M*P | XXtr M*M | XYte M*P | XXte M*M | YYte P ]
struct ZLayout {
    size_t off_XY[2], off_XX[2], off_YY, len;
};
__host__ __device__ static inline ZLayout z_layout(size_t M, size_t P) {
    ZLayout z;
    z.off_XY[0] = 0;
    z.off_XX[0] = M * P;
    z.off_XY[1] = z.off_XX[0] + M * M;
    z.off_XX[1] = z.off_XY[1] + M * P;
    z.off_YY = z.off_XX[1] + M * M;
    z.len = z.off_YY + P;
    return z;
}

// mean / n-1 stdev of every column, then the z-scored cross-products of both partitions.  A grid of work-groups: every one
// derives the column moments itself (a few loads per column), the M (M + P) cross-product entries of each partition are dealt
// out over the grid (one work-group took 69 us at 128 metrics: 144 dependent global loads per thread); block 0 writes the model.
// (a device function: small sets run it as the prologue of k_pls_fit16 instead of as a launch of its own)
__device__ __forceinline__ void zstats_body(const double* __restrict__ stats, int M, int P, int A, double* __restrict__ model,
                                            double* __restrict__ zwork, double* __restrict__ zsh /* 2 (M + P) doubles of LDS */,
                                            int t, int nt, int blk, int nblk) {
    const StatsLayout L = stats_layout(M, P);
    const ModelLayout ML = model_layout(M, P, A);
    const ZLayout Z = z_layout(M, P);
    const int C = M + P;
    const bool first = blk == 0;
    double* delta = zsh;
    double* sd = zsh + C;
    const double n0 = stats[L.off_n], n1 = stats[L.off_n + 1];
    const double n = n0 + n1;
    for (int c = t; c < C; c += nt) {
        const double s = stats[L.off_sum[0] + c] + stats[L.off_sum[1] + c];
        const double d = (n > 0) ? s / n : 0.0;  // mean - shift
        const double g = stats[L.off_G[0] + c + L.C16 * c] + stats[L.off_G[1] + c + L.C16 * c];
        double ss = g - n * d * d;               // centred sum of squares
        if (ss < 0.0) ss = 0.0;
        const double sdv = (n >= 2) ? sqrt(ss / (n - 1.0)) : 0.0;
        delta[c] = d;
        sd[c] = sdv;
        if (first) {
            model[ML.off_mean + c] = stats[L.off_shift + c] + d;
            model[ML.off_sd + c] = sdv;
        }
    }
    if (first && t == 0) { model[ML.off_hdr + 1] = (double)A; model[ML.off_hdr + 2] = n; model[ML.off_hdr + 3] = 0.0; }
    __syncthreads();
    if (zwork == nullptr) return;
    const int stride = nt * nblk;
    for (int part = 0; part < 2; part++) {
        const double np = part ? n1 : n0;
        const double* G = stats + L.off_G[part];
        const double* S = stats + L.off_sum[part];
        // XX (M x M) and XY (M x P) in one sweep over columns b of [X|Y]
        for (int e = blk * nt + t; e < M * C; e += stride) {
            const int a = e % M, b = e / M;
            const double cross = G[a + L.C16 * b] - delta[a] * S[b] - delta[b] * S[a] + np * delta[a] * delta[b];
            const double den = sd[a] * sd[b];
            const double zv = (den > 0.0) ? cross / den : 0.0;
            if (b < M) zwork[Z.off_XX[part] + a + (size_t)M * b] = zv;
            else zwork[Z.off_XY[part] + a + (size_t)M * (b - M)] = zv;
        }
    }
    if (first)
        for (int j = t; j < P; j += nt) {
            const int c = M + j;
            const double cross = stats[L.off_G[1] + c + L.C16 * c] - 2.0 * delta[c] * stats[L.off_sum[1] + c] +
                                 n1 * delta[c] * delta[c];
            const double den = sd[c] * sd[c];
            zwork[Z.off_YY + j] = (den > 0.0) ? cross / den : 0.0;
        }
}
__global__ __launch_bounds__(256) void k_zstats(const double* __restrict__ stats, int M, int P, int A,
                                                double* __restrict__ model, double* __restrict__ zwork) {
    extern __shared__ double zsh[];          // 2 (M + P) doubles
    zstats_body(stats, M, P, A, model, zwork, zsh, threadIdx.x, 256, blockIdx.x, gridDim.x);
}

typedef double d4 __attribute__((ext_vector_type(4)));

// trace of a matrix held as NB x NB blocks in the C/D layout: diagonal element 16 I + c sits in register c>>2 of
// lane c + 16 (c & 3)
template <int NB>
__device__ __forceinline__ double trace_cd(const d4 (&D)[NB][NB]) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    double t = 0.0;
    if ((c & 3) == q) {                  // this lane holds diagonal entries 16 I + c, in register c >> 2
#pragma unroll
        for (int I = 0; I < NB; I++) {
            const double lo = (c & 4) ? D[I][I][1] : D[I][I][0], hi = (c & 4) ? D[I][I][3] : D[I][I][2];
            t += (c & 8) ? hi : lo;
        }
    }
    return wave_sum(t);
}

// Dominant-eigenvector helper: repeated squaring of B = S / trace(S) (S symmetric PSD, n <= 16*NB)
// entirely in registers with v_mfma_f64_16x16x4_f64.  A 16x16 block of a SYMMETRIC matrix held in
// the MFMA C/D layout (lane l, reg r <-> [l&15][(l>>4) + 4r], using symmetry) is bit-for-bit the
// A-operand layout of k-step r AND (again by symmetry) the B-operand layout, so
//   C_IJ = sum_K sum_r mfma(D_IK[r], D_JK[r])        needs no data movement between squarings.
// The error is squared every step once the spectral gap opens: "changed by < 1e-9", then one more.
template <int NB>
__device__ double eig_square(const double* __restrict__ XY, int M, int n, double* __restrict__ S,
                             double* __restrict__ Bout, double* __restrict__ grp_out = nullptr,
                             const double* __restrict__ Sp = nullptr, int nparts = 0) {
    const int lane = threadIdx.x & 63, c = lane & 15, q = lane >> 4;
    // S = XY' XY straight into the C/D register layout with MFMA: block (I,J) = sum over 4-row slabs of
    // mfma(a_J, a_I), a_I(lane) = XY[4s + q][16 I + c]  (zero beyond M rows / n columns)
    d4 D[NB][NB];
#pragma unroll
    for (int I = 0; I < NB; I++)
#pragma unroll
        for (int J = 0; J < NB; J++) D[I][J] = (d4){0.0, 0.0, 0.0, 0.0};
    if (NB == 1 && Sp) {
        // the work-group's waves have each contracted their share of the slabs (k_pls_fit): block = sum of the partial blocks
#pragma unroll
        for (int r = 0; r < 4; r++) {
            double v = 0.0;
            for (int w = 0; w < nparts; w++) v += Sp[(4 * w + r) * 64 + lane];
            D[0][0][r] = v;
        }
    } else
    for (int m0 = 0; m0 < M; m0 += 4) {
        double a[NB];
#pragma unroll
        for (int I = 0; I < NB; I++) {
            const int m = m0 + q, col = 16 * I + c;
            a[I] = (m < M && col < n) ? XY[m + M * col] : 0.0;
        }
#pragma unroll
        for (int I = 0; I < NB; I++)
#pragma unroll
            for (int J = 0; J < NB; J++) D[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[J], a[I], D[I][J], 0, 0, 0);
    }
    const double tr = trace_cd<NB>(D);
    const double itr = (tr > 0.0) ? __builtin_amdgcn_rcp(tr) : 0.0;     // (the scale of B is immaterial: every group renormalises by its own trace)
#pragma unroll
    for (int I = 0; I < NB; I++)
#pragma unroll
        for (int J = 0; J < NB; J++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * I + c, j = 16 * J + q + 4 * r;
                if (i < n && j < n) S[i + n * j] = D[I][J][r];          // kept in LDS for the power step
                D[I][J][r] = (tr > 0.0) ? D[I][J][r] * itr : ((i == 0 && j == 0) ? 1.0 : 0.0);
            }
    // Groups of 3 un-normalised squarings of the trace-1 matrix (B -> B^8, entries >= n^-8), then one trace
    // normalisation.  With eigenvalues lambda_i (sum 1), t = trace(B^8) = sum lambda_i^8.  t > 0.95 forces
    // lambda_1 > 0.9936, i.e. every other eigenvalue of the group's INPUT was < 6.4e-3 of it, so its OUTPUT has
    // them below (6.4e-3)^8 = 3e-18: converged to rounding, stop.
    int grp = 0;
    for (; grp < 24; grp++) {
#pragma unroll 1
        for (int sq = 0; sq < 3; sq++) {
            d4 T[NB][NB];
#pragma unroll
            for (int I = 0; I < NB; I++)
#pragma unroll
                for (int J = 0; J < NB; J++) {
                    d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int K = 0; K < NB; K++) {   // operands (J,K),(I,K): the MFMA output row/col map gives block (I,J)
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(D[J][K][0], D[I][K][0], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(D[J][K][1], D[I][K][1], acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(D[J][K][2], D[I][K][2], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(D[J][K][3], D[I][K][3], acc1, 0, 0, 0);
                    }
                    T[I][J] = acc0 + acc1;
                }
#pragma unroll
            for (int I = 0; I < NB; I++)
#pragma unroll
                for (int J = 0; J < NB; J++) D[I][J] = T[I][J];
        }
        const double t = trace_cd<NB>(D);
        const double inv = __builtin_amdgcn_rcp(t);      // v_rcp_f64: the scale only keeps the entries in range
#pragma unroll
        for (int I = 0; I < NB; I++)
#pragma unroll
            for (int J = 0; J < NB; J++)
#pragma unroll
                for (int r = 0; r < 4; r++) D[I][J][r] *= inv;
        if (t > 0.95) break;
    }
    if (grp_out && lane == 0) *grp_out = (double)grp;
#pragma unroll
    for (int I = 0; I < NB; I++)
#pragma unroll
        for (int J = 0; J < NB; J++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 16 * I + c, j = 16 * J + q + 4 * r;
                if (i < n && j < n) Bout[i + n * j] = D[I][J][r];
            }
    return tr;
}

// More than 64 responses (the reference has no limit; the register-resident squaring above stops at 64 x 64): the same
// algorithm -- S = XY'XY, B = S / trace, groups of three squarings with a trace normalisation until trace(B^8) > 0.95, the column
// of the largest diagonal entry, one power step with S, unit norm, largest |component| positive -- by the whole work-group on
// matrices in memory (S, B, T: n x n each; wide sets always run the global-memory mode).  qv receives the eigenvector.
template <int NT>
__device__ void eig_generic(const double* __restrict__ XY, int M, int n, double* __restrict__ S, double* __restrict__ Bm,
                            double* __restrict__ T, double* __restrict__ qv, double* __restrict__ red /* >= 8 */) {
    const int t = threadIdx.x;
    for (int e = t; e < n * n; e += NT) {
        const int i = e % n, j = e / n;
        double s = 0.0;
        for (int m = 0; m < M; m++) s = fma(XY[m + M * i], XY[m + M * j], s);
        S[e] = s;
    }
    __threadfence_block(); __syncthreads();
    if (t == 0) { double tr = 0.0; for (int i = 0; i < n; i++) tr += S[i + n * i]; red[0] = tr; }
    __threadfence_block(); __syncthreads();
    const double tr = red[0];
    for (int e = t; e < n * n; e += NT) Bm[e] = (tr > 0.0) ? S[e] / tr : ((e == 0) ? 1.0 : 0.0);
    __threadfence_block(); __syncthreads();
    for (int grp = 0; grp < 24; grp++) {
        for (int sq = 0; sq < 3; sq++) {
            for (int e = t; e < n * n; e += NT) {
                const int i = e % n, j = e / n;
                double s = 0.0;
                for (int k = 0; k < n; k++) s = fma(Bm[i + n * k], Bm[k + n * j], s);
                T[e] = s;
            }
            __threadfence_block(); __syncthreads();
            for (int e = t; e < n * n; e += NT) Bm[e] = T[e];
            __threadfence_block(); __syncthreads();
        }
        if (t == 0) { double tt = 0.0; for (int i = 0; i < n; i++) tt += Bm[i + n * i]; red[1] = tt; }
        __threadfence_block(); __syncthreads();
        const double tt = red[1];
        for (int e = t; e < n * n; e += NT) Bm[e] = Bm[e] / tt;
        __threadfence_block(); __syncthreads();
        if (tt > 0.95) break;
    }
    if (t == 0) {
        int best = 0;
        for (int i = 1; i < n; i++) if (Bm[i + n * i] > Bm[best + n * best]) best = i;       // ties -> lowest index
        red[2] = (double)best;
    }
    __threadfence_block(); __syncthreads();
    const int best = (int)red[2];
    for (int i = t; i < n; i += NT) T[i] = Bm[i + n * best];
    __threadfence_block(); __syncthreads();
    if (tr > 0.0) {
        for (int i = t; i < n; i += NT) {
            double v = 0.0;
            for (int k = 0; k < n; k++) v = fma(S[i + n * k], T[k], v);
            T[n + i] = v;
        }
        __threadfence_block(); __syncthreads();
        for (int i = t; i < n; i += NT) T[i] = T[n + i];
        __threadfence_block(); __syncthreads();
    }
    if (t == 0) {
        double nn = 0.0, am = -1.0, sv = 1.0;
        for (int i = 0; i < n; i++) { nn = fma(T[i], T[i], nn); if (fabs(T[i]) > am) { am = fabs(T[i]); sv = T[i]; } }
        red[3] = sqrt(nn);
        red[4] = (sv < 0.0) ? -1.0 : 1.0;
    }
    __threadfence_block(); __syncthreads();
    for (int i = t; i < n; i += NT) qv[i] = red[4] * T[i] / red[3];
    __threadfence_block(); __syncthreads();
}

// Small GEMMs of the PRESS statistics on the fp64 matrix pipe: C[i + ldc j] = sum_k A(i,k) B(k,j), i < I, j < J, k < Kd, with
// A(i,k) = pa[i sai + k sak], B(k,j) = pb[k sbk + j sbj]; the 16 x 16 output blocks are dealt out to the NW waves, every block
// v_mfma_f64_16x16x4_f64 over Kd / 4 steps (A operand: lane (c, q) holds A[row c][k q]; B: B[k q][col c]; D: [q + 4 r][c]).
// (One thread per output element walking Kd dependent fma's with L2 loads was 22 % of the fit at 32 components.)
template <int NW>
__device__ __forceinline__ void pls_gemm(const double* __restrict__ pa, size_t sai, size_t sak, const double* __restrict__ pb, size_t sbk,
                                         size_t sbj, int I, int J, int Kd, double* __restrict__ Cm, size_t ldc, int wrot = 0) {
    const int lane = threadIdx.x & 63, wave = ((threadIdx.x >> 6) + NW - wrot) % NW, c = lane & 15, q = lane >> 4;
    const int nbi = (I + 15) / 16, nbj = (J + 15) / 16;
    for (int b = wave; b < nbi * nbj; b += NW) {
        const int bi = b % nbi, bj = b / nbi;
        const int i = 16 * bi + c, j = 16 * bj + c;
        const int ic = (i < I) ? i : I - 1, jc = (j < J) ? j : J - 1;       // in-range addresses, masked values
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        constexpr int U = 8;                       // k-steps whose operands are fetched together (the loads were waited for one by one)
        for (int k0 = 0; k0 < Kd; k0 += 4 * U) {
            double av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 + 4 * u + q, kc = (k < Kd) ? k : Kd - 1;
                av[u] = pa[(size_t)ic * sai + (size_t)kc * sak];
                bv[u] = pb[(size_t)kc * sbk + (size_t)jc * sbj];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 + 4 * u + q;
                if (k0 + 4 * u < Kd)               // (wave-uniform)
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64((i < I && k < Kd) ? av[u] : 0.0, (j < J && k < Kd) ? bv[u] : 0.0, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 16 * bi + q + 4 * r;
            if (row < I && j < J) Cm[row + ldc * (size_t)j] = acc[r];
        }
    }
}

// sum over the work-group (NW waves); NW == 1: the wave sum
template <int NW>
__device__ __forceinline__ double pls_sum(double v, double* red) {
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    __threadfence_block();
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __threadfence_block();
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; w++) s += red[w];
    return s;
}

// NW == 1: ONE wavefront (M <= 64: everything is tiny and latency-bound, no inter-wave barriers).  NW == 8 (M > 64, e.g.
// BASELINE configs[4]: 128 metrics, 32 components): the eigenvector work stays on wave 0 (register-resident MFMA), every
// vector phase is spread over the work-group -- X'X r with each row's dot product split four ways, the r-update with one
// wave per earlier component, the PRESS contractions one entry per thread.  LDS: XY (M*P), S and V (np*np each), vectors.
// NBT: the register-resident eigen-squaring is compiled for 16 NBT x 16 NBT blocks (1: up to 16 responses, 2: up to 32); 0: 33..64
// responses (4 x 4 blocks, an out-of-line call) or the memory-resident one beyond -- one kernel per case, so that the 16-response
// kernel does not carry the registers of the 64-response one.
template <int NW, bool GMEM, int NBT>
__global__ __launch_bounds__(64 * NW) void k_pls_fit(const double* __restrict__ zwork, const double* __restrict__ obs,
                                                int M, int P, int A, double* __restrict__ model,
                                                double* __restrict__ scratch /* A*M + A*A + P*A */, int xx_in_lds,
                                                double* __restrict__ gbase /* NULL: the work arrays live in LDS */) {
    extern __shared__ double lds_[];
    // wide sets (M P + 2 M A + ... beyond the 160 KB of LDS): the same arrays in global memory; every barrier is then preceded
    // by a work-group fence so the stores are visible to the other waves of the (single) work-group
    // (a compile-time switch: with a run-time choice of base every access to the work arrays is a flat load -- the LDS through the
    // flat aperture, each waited for with vmcnt(0) AND lgkmcnt(0))
    double* const lds = GMEM ? gbase : lds_;
#define PLS_SYNC() do { if constexpr (GMEM) __threadfence_block(); __syncthreads(); } while (0)
#define PLS_WSYNC() do { if constexpr (GMEM) __threadfence_block(); __builtin_amdgcn_wave_barrier(); } while (0)
#ifdef PLS_STAMPS
    long long st_last = __builtin_readcyclecounter();
    double* st_out = scratch + (size_t)A * M + (size_t)A * A + 2 * (size_t)P * A;     // 16 doubles of diagnostics
    if (threadIdx.x < 16) st_out[threadIdx.x] = 0.0;
#define STAMP(id) do { const long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) st_out[id] += (double)(now_ - st_last); st_last = now_; } while (0)
#else
#define STAMP(id)
#endif
    const ModelLayout ML = model_layout(M, P, A);
    const ZLayout Z = z_layout(M, P);
    constexpr int NT = 64 * NW;
    const int lane = threadIdx.x;            // thread in the work-group (NW == 1: the lane)
    const int wave = threadIdx.x >> 6;
    const int np = P;
    double* XY = lds;                   // M*P
    double* S = XY + (size_t)M * P;     // np*np
    double* V = S + np * np;            // np*np: converged power of S
    double* qv = V + np * np;           // np
    double* wv = qv + np;               // M
    double* rv = wv + M;                // M
    double* pv = rv + M;                // M
    double* xr = pv + M;                // M
    double* Pl = xr + M;                // M*A: loadings P (LDS copy, read by the deflation of later components)
    double* Rl = Pl + (size_t)M * A;    // M*A: rotations R
    double* XXl = Rl + (size_t)M * A;   // M*M copy of X'X (training) when it fits (xx_in_lds)
    double* red = XXl + (xx_in_lds ? (size_t)M * M : 0);   // 8: work-group sums
    double* pwv = red + 8;              // A: projections of w on the earlier loadings (NW > 1)
    double* xp = pwv + A;               // 4*M: partial dot products of X'X r (NW > 1)
    double* Tg = xp + 4 * (size_t)M;    // np*np + np: third matrix of the memory-resident eigen-squaring (P > 64 only)
    double* Sp = Tg + (P > 64 ? (size_t)np * np + np : 0);   // NW*256: the waves' partial XY'XY blocks (NW > 1, P <= 16)

    const double* XXtr = zwork + Z.off_XX[0];
    if (xx_in_lds) {
        for (int e = lane; e < M * M; e += NT) XXl[e] = XXtr[e];
        XXtr = XXl;
    }
    double* Rm = model + ML.off_R;
    double* Qm = model + ML.off_Q;
    double* Wm = model + ML.off_W;
    double* Pm = model + ML.off_P;

    for (int e = lane; e < M * P; e += NT) XY[e] = zwork[Z.off_XY[0] + e];
    // 65..128 metrics on eight waves: X'X (up to 128 KB, beside 64 KB of loadings: too much for the LDS) lives in REGISTERS, thread
    // (row a, quarter part) keeping the <= 32 entries of its quarter row; X'X r is then 32 fma's on wave-uniform LDS reads of r
    constexpr bool XXREG_OK = (NW == 8) && !GMEM;
    const bool xx_in_reg = XXREG_OK && M <= 128;
    double xxq[XXREG_OK ? 32 : 1];
    if constexpr (XXREG_OK) {
        if (xx_in_reg) {
            const int qb = (M + 3) / 4, a = lane % M, part = lane / M, b0 = part * qb;
#pragma unroll
            for (int i = 0; i < 32; i++) xxq[i] = (part < 4 && i < qb && b0 + i < M) ? XXtr[a + (size_t)M * (b0 + i)] : 0.0;
        }
    }
    PLS_SYNC();

    STAMP(9);
    for (int comp = 0; comp < A; comp++) {
        if (P == 1) {
            for (int m = lane; m < M; m += NT) wv[m] = XY[m];
        } else {
            // S = XY' XY (symmetric PSD, P x P) and its dominant eigenvector: MFMA cross-product into registers,
            // repeated squaring there (eig_square), one power step with S itself.  (The oracle uses a full
            // Jacobi eigen-solve; both deliver the dominant eigenvector of the same symmetric matrix.)
            const int n = P;
            double* Bc = V;
            STAMP(1);
            if constexpr (NW > 1 && NBT == 1) {
                // XY'XY: every wave contracts every NW-th slab of four rows on the matrix pipe (one wave: M/4 dependent MFMAs, the
                // longest single item of a component at 128 metrics), wave 0 adds the partial blocks
                const int l = lane & 63, c = l & 15, q = l >> 4;
                d4 Dp = (d4){0.0, 0.0, 0.0, 0.0};
                for (int m0 = 4 * wave; m0 < M; m0 += 4 * NW) {
                    const int m = m0 + q;
                    const double a = (m < M && c < n) ? XY[m + M * c] : 0.0;
                    Dp = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, Dp, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) Sp[(4 * wave + r) * 64 + l] = Dp[r];
                PLS_SYNC();
            }
            if (NBT == 0 && n > 64) {
                eig_generic<NT>(XY, M, n, S, Bc, Tg, qv, red);
            } else
            if (NW == 1 || wave == 0) {          // the whole eigenvector step is one wave's work (wave-uniform branch)
            double tr;
#ifdef PLS_STAMPS
            double* const grp_out = st_out + 16 + (comp < 48 ? comp : 47);
#else
            double* const grp_out = nullptr;
#endif
            if constexpr (NBT == 1) tr = eig_square<1>(XY, M, n, S, Bc, grp_out, (NW > 1) ? Sp : nullptr, NW);
            else if constexpr (NBT == 2) tr = eig_square<2>(XY, M, n, S, Bc, grp_out);
            else tr = eig_square<4>(XY, M, n, S, Bc, grp_out);
            if constexpr (NW == 1) PLS_SYNC(); else PLS_WSYNC();
            STAMP(2);
            // column of the converged power with the largest diagonal entry (wave arg-max, ties -> lowest index)
            double dg = (lane < n) ? Bc[lane + n * lane] : -1.0, dpay = 0.0;
            int bi = lane;
            wave_argmax(dg, bi, dpay);
            const int best = bi;
            // one power step with S itself washes out the rounding of the squarings; then unit norm and
            // the sign convention (largest |component| positive; ties -> lowest index), all in registers
            double qi = (lane < n) ? Bc[lane + n * best] : 0.0;
            if (tr > 0.0) {
                if (lane < n) qv[lane] = qi;
                if constexpr (NW == 1) PLS_SYNC(); else PLS_WSYNC();
                double v = 0.0;
                if (lane < n) {
#pragma unroll 8
                    for (int k = 0; k < n; k++) v = fma(S[lane + n * k], qv[k], v);
                }
                qi = v;
                if constexpr (NW == 1) PLS_SYNC(); else PLS_WSYNC();
            }
            const double nrm = sqrt(wave_sum(qi * qi));
            double am = fabs(qi), sv = qi;
            int ai = lane;
            wave_argmax(am, ai, sv);
            const double sgn = (sv < 0.0) ? -1.0 : 1.0;
            if (lane < n) qv[lane] = sgn * qi / nrm;
            }
            PLS_SYNC();
            for (int m = lane; m < M; m += NT) {
                double s = 0.0;
                _Pragma("unroll 8") for (int j = 0; j < P; j++) s = fma(XY[m + M * j], qv[j], s);
                wv[m] = s;
            }
        }
        PLS_SYNC();
        STAMP(3);
        double ww = 0.0;
        for (int m = lane; m < M; m += NT) ww = fma(wv[m], wv[m], ww);
        ww = sqrt(pls_sum<NW>(ww, red));
        for (int m = lane; m < M; m += NT) { const double x = wv[m] / ww; wv[m] = x; rv[m] = x; }
        PLS_SYNC();
        if constexpr (NW == 1) {
            for (int j = 0; j < comp; j++) {
                double pw = 0.0;
                for (int m = lane; m < M; m += NT) pw = fma(Pl[m + (size_t)M * j], wv[m], pw);
                pw = wave_sum(pw);
                for (int m = lane; m < M; m += NT) rv[m] -= pw * Rl[m + (size_t)M * j];
            }
        } else {
            // the projections p_j'w only involve w: one wave per earlier component (same lane -> row map and wave sum as the
            // one-wave code, so the same values), then every row subtracts them in component order
            for (int j = wave; j < comp; j += NW) {
                double pw = 0.0;
                for (int m = lane & 63; m < M; m += 64) pw = fma(Pl[m + (size_t)M * j], wv[m], pw);
                pw = wave_sum(pw);
                if ((lane & 63) == 0) pwv[j] = pw;
            }
            PLS_SYNC();
            for (int m = lane; m < M; m += NT) {
                double r = rv[m];
                for (int j = 0; j < comp; j++) r -= pwv[j] * Rl[m + (size_t)M * j];
                rv[m] = r;
            }
        }
        PLS_SYNC();
        STAMP(4);
        // type 2: xr = XX r ; tt = r' xr ; p = xr / tt
        double tt = 0.0;
        if constexpr (NW == 1) {
            for (int a = lane; a < M; a += NT) {
                double s = 0.0;
                _Pragma("unroll 8") for (int b = 0; b < M; b++) s = fma(XXtr[a + (size_t)M * b], rv[b], s);
                xr[a] = s;
                tt = fma(rv[a], s, tt);
            }
            tt = wave_sum(tt);
        } else {
            // four threads per row, a quarter of the columns each (four independent chains of M/4 instead of one of M; as a
            // one-column GEMM on the fp64 matrix pipe the same product was twice as slow: 32 dependent steps with L2 loads)
            const int qb = (M + 3) / 4;
            bool done = false;
            if constexpr (XXREG_OK) {
                if (xx_in_reg) {
                    if (lane < 4 * M) {
                        const int b0 = (lane / M) * qb;
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < 32; i++) { const int b = (b0 + i < M) ? b0 + i : M - 1; s = fma(xxq[i], rv[b], s); }
                        xp[lane] = s;        // (entries past the quarter are zeros: same chain as the loop below)
                    }
                    done = true;
                }
            }
            if (!done)
            for (int e = lane; e < 4 * M; e += NT) {
                const int a = e % M, part = e / M;
                const int b0 = part * qb, b1 = (b0 + qb < M) ? b0 + qb : M;
                double s = 0.0;
                _Pragma("unroll 8") for (int b = b0; b < b1; b++) s = fma(XXtr[a + (size_t)M * b], rv[b], s);
                xp[e] = s;
            }
            PLS_SYNC();
            for (int a = lane; a < M; a += NT) {
                const double s = (xp[a] + xp[M + a]) + (xp[2 * M + a] + xp[3 * M + a]);
                xr[a] = s;
                tt = fma(rv[a], s, tt);
            }
            tt = pls_sum<NW>(tt, red);
        }
        for (int m = lane; m < M; m += NT) {
            const double pm = xr[m] / tt;
            pv[m] = pm;
            Pm[m + (size_t)M * comp] = pm;
            Pl[m + (size_t)M * comp] = pm;
            Wm[m + (size_t)M * comp] = wv[m];
            Rm[m + (size_t)M * comp] = rv[m];
            Rl[m + (size_t)M * comp] = rv[m];
        }
        PLS_SYNC();
        STAMP(5);
        if (NW > 1 && M > 64 && 4 * M >= 8 * P) {
            // q_j = XY[:, j]' r / tt: eight threads per response, an eighth of the rows each, partial sums through xp (4 M doubles)
            const int qb = (M + 7) / 8;
            for (int e = lane; e < 8 * P; e += NT) {
                const int j = e % P, part = e / P;
                const int m0 = part * qb, m1 = (m0 + qb < M) ? m0 + qb : M;
                double s = 0.0;
                _Pragma("unroll 8") for (int m = m0; m < m1; m++) s = fma(XY[m + M * j], rv[m], s);
                xp[e] = s;
            }
            PLS_SYNC();
            for (int j = lane; j < P; j += NT) {
                double s = 0.0;
                for (int part = 0; part < 8; part++) s += xp[part * P + j];
                s /= tt;
                qv[j] = s;
                Qm[j + (size_t)P * comp] = s;
            }
        } else {
        for (int j = lane; j < P; j += NT) {
            double s = 0.0;
            _Pragma("unroll 8") for (int m = 0; m < M; m++) s = fma(XY[m + M * j], rv[m], s);
            s /= tt;
            qv[j] = s;
            Qm[j + (size_t)P * comp] = s;
        }
        }
        PLS_SYNC();
        for (int e = lane; e < M * P; e += NT) {
            const int m = e % M, j = e / M;
            XY[e] -= tt * (pv[m] * qv[j]);
        }
        PLS_SYNC();
        STAMP(6);
    }

    STAMP(7);
    // ---- PRESS on the validation statistics -------------------------------------------------
    // c_jk = r_k' XYte[:,j]; v_k = XXte r_k; H_kl = r_k' v_l;
    // PRESS_j(a) = YY_jj - 2 sum_{k<a} q_jk c_jk + sum_{k,l<a} q_jk q_jl H_kl
    const double* XYte = zwork + Z.off_XY[1];
    const double* XXte = zwork + Z.off_XX[1];
    const double* YYte = zwork + Z.off_YY;
    double* vk = scratch;               // A*M
    double* H = vk + (size_t)A * M;     // A*A
    double* cm = H + (size_t)A * A;     // P*A
    if (A * M >= 1024) {           // (below that one thread per element is as fast: 256 outputs at 32 metrics, 8 components)
        pls_gemm<NW>(XXte, 1, (size_t)M, Rm, 1, (size_t)M, M, A, M, vk, (size_t)M);            // vk = XXte R        (M x A)
        PLS_SYNC();
        pls_gemm<NW>(Rm, (size_t)M, 1, vk, 1, (size_t)M, A, A, M, H, (size_t)A);                 // H = R' vk          (A x A)
        pls_gemm<NW>(XYte, (size_t)M, 1, Rm, 1, (size_t)M, P, A, M, cm, (size_t)P);              // c = XYte' R        (P x A)
    } else {
        for (int e = lane; e < A * M; e += NT) {
            const int m = e % M, k = e / M;
            double s = 0.0;
            _Pragma("unroll 8") for (int b = 0; b < M; b++) s = fma(XXte[m + (size_t)M * b], Rm[b + (size_t)M * k], s);
            vk[e] = s;
        }
        PLS_SYNC();
        for (int e = lane; e < A * A; e += NT) {
            const int k = e % A, l = e / A;
            double s = 0.0;
            _Pragma("unroll 8") for (int m = 0; m < M; m++) s = fma(Rm[m + (size_t)M * k], vk[m + (size_t)M * l], s);
            H[e] = s;
        }
        for (int e = lane; e < P * A; e += NT) {
            const int j = e % P, k = e / P;
            double s = 0.0;
            _Pragma("unroll 8") for (int m = 0; m < M; m++) s = fma(Rm[m + (size_t)M * k], XYte[m + (size_t)M * j], s);
            cm[e] = s;
        }
    }
    PLS_SYNC();
    for (int e = lane; e < A * A; e += NT) model[ML.off_H + e] = H[e];          // (the Wilcoxon reduction's scales)
    double* press = model + ML.off_press;   // A x P, column-major
    for (int j = lane; j < P; j += NT) {
        double lin = 0.0, quad = 0.0;
        double best = 0.0; int besta = 0;
        for (int a = 0; a < A; a++) {
            const double qa = Qm[j + (size_t)P * a];
            lin = fma(qa, cm[j + (size_t)P * a], lin);
            // add row/column a of the quadratic form
            double add = 0.0;
            _Pragma("unroll 8") for (int l = 0; l < a; l++) add = fma(Qm[j + (size_t)P * l], H[a + (size_t)A * l], add);
            quad += 2.0 * qa * add + qa * qa * H[a + (size_t)A * a];
            const double pr = YYte[j] - 2.0 * lin + quad;
            press[a + (size_t)A * j] = pr;
            if (a == 0 || pr < best) { best = pr; besta = a; }
        }
        model[ML.off_per + j] = (double)(besta + 1);
    }
    PLS_SYNC();
    STAMP(8);
    // ncomp = max over responses; observed z-scores and scores (m-ascending fma chain per component)
    int ncomp = 1;
    for (int j = 0; j < P; j++) { const int v = (int)model[ML.off_per + j]; if (v > ncomp) ncomp = v; }
    if (lane == 0) model[ML.off_hdr] = (double)ncomp;
    for (int m = lane; m < M; m += NT) {
        const double sdv = model[ML.off_sd + m];
        model[ML.off_zobs + m] = (sdv == 0.0) ? 0.0 : (obs[m] - model[ML.off_mean + m]) / sdv;
    }
    PLS_SYNC();
    for (int k = lane; k < A; k += NT) {
        double s = 0.0;
        _Pragma("unroll 8") for (int m = 0; m < M; m++) s = fma(model[ML.off_zobs + m], Rm[m + (size_t)M * k], s);
        model[ML.off_oscore + k] = s;
    }
}


// s + sum_{i < cnt} a[i sa] b[i sb] as ONE fma chain in index order, the operands of U steps loaded ahead of their arithmetic.
// (Left to itself hipcc emits load, s_waitcnt lgkmcnt(0), fma per element: an LDS round trip of ~120 cycles on every step of
// chains that are 16 to 32 steps long -- most of what a component cost.)
template <int U>
__device__ __forceinline__ double dot_ahead(const double* __restrict__ a, int sa, const double* __restrict__ b, int sb, int cnt,
                                            double s) {
    int i = 0;
    for (; i + U <= cnt; i += U) {
        double av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { av[u] = a[(size_t)(i + u) * sa]; bv[u] = b[(size_t)(i + u) * sb]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; u++) s = fma(av[u], bv[u], s);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (U > 1) {                  // the rest in batches of U/2, U/4, ... (no masked lanes, no wasted loads)
        if (i < cnt) s = dot_ahead<U / 2>(a + (size_t)i * sa, sa, b + (size_t)i * sb, sb, cnt - i, s);
    }
    return s;
}

// ---- up to 16 responses, more than one wave: the latency-tuned fit --------------------------------------------------------
// Same algorithm and the same formulas as k_pls_fit; what differs is who does what between barriers, because a component is a
// chain of short dependent phases and a work-group barrier per phase was a third of its time (15 per component there, 5 here):
//   (1) every wave finishes the PREVIOUS component for the slabs of four rows it owns -- tt = r'X'X r and q recomputed by every
//       wave from the partial sums (no cross-wave reduction), XY deflated in place -- and contracts those slabs into its partial
//       XY'XY on the matrix pipe in the same pass;
//   (2) wave 0: dominant eigenvector (eig_square's scheme; the closing power step with S is one more MFMA product, row `best` of
//       B S, so nothing leaves the registers), then w = XY q;
//   (3) every wave: |w| (itself), the projections p_j'w of its share of the earlier components;
//   (4) r = w - sum_j (p_j'w) r_j, one row per thread;
//   (5) partial sums of X'X r (four threads per row) and of XY'r (eight per response).
// LDS: XY, the loadings and rotations, the partial blocks; X'X in LDS up to 64 metrics, in registers up to 128 on eight waves.
// NB: 16 x 16 blocks per side of XY'XY (1: up to 16 responses, 2: up to 32)
template <int NW, int NB>
__global__ __launch_bounds__(64 * NW) void k_pls_fit16(const double* zwork, const double* __restrict__ obs,
                                                      int M, int P, int A, double* __restrict__ model,
                                                      double* __restrict__ scratch /* A*M + A*A + P*A */, int xx_in_lds,
                                                      const double* __restrict__ stats /* non-NULL: run k_zstats' work first */,
                                                      double* __restrict__ zwork_w) {
    extern __shared__ double lds_[];
    const ModelLayout ML = model_layout(M, P, A);
    const ZLayout Z = z_layout(M, P);
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x, l = tid & 63, wave = tid >> 6, c = l & 15, q4 = l >> 4;
    if (stats) {
        // small sets: the z-scored cross-products are this work-group's own prologue (a launch and its dependency gap less)
        zstats_body(stats, M, P, A, model, zwork_w, lds_, tid, NT, 0, 1);
        __threadfence_block();
        __syncthreads();
    }
    const int n = P;
    // XY column-major with the leading dimension padded to 2 (mod 4): the slab pass reads / writes four consecutive rows of all 16
    // columns with one instruction, 16-way bank-conflicted at a leading dimension of 32 or 128 doubles, conflict-free at 34 / 130
    const int LX = M + ((6 - (M & 3)) & 3);
    double* XY = lds_;                        // LX*P
    double* qv = XY + (size_t)LX * P;         // 16 NB
    double* wv = qv + 16 * NB;                // M: w before normalisation
    double* wn = wv + M;                      // M: w / |w|
    double* rv = wn + M;                      // M + 4 (zero tail: the quarter rows of step (5) may read up to 4 ceil(M/4) entries)
    double* xp = rv + M + 4;                  // 4*M: partial sums of X'X r
    double* qp = xp + 4 * (size_t)M;          // 8 * 16 NB: partial sums of XY'r
    double* pwv = qp + 128 * NB;              // A: p_j'w
    double* Pl = pwv + A;                     // M*A
    double* Rl = Pl + (size_t)M * A;          // M*A
    double* Sp = Rl + (size_t)M * A;          // NW * NB^2 * 256: partial XY'XY blocks, [wave][block][reg][lane]
    double* XXl = Sp + NW * NB * NB * 256;    // M*M (xx_in_lds)
    double* Hl = XXl + (xx_in_lds ? (size_t)M * M : 0);   // A*A   } PRESS statistics (vk = XXte R overlays Pl, which is dead by then)
    double* cml = Hl + (size_t)A * A;         // P*A   }
    double* addl = cml + (size_t)P * A;       // P*A   }
    double* Ql = addl + (size_t)P * A;        // P*A: Q (LDS copy)
    double* Ex = Ql + (size_t)P * A;          // NB == 2: two exchange buffers of the distributed squaring, NB^2 * 256 each
    const double* XXtr = zwork + Z.off_XX[0];
    if (xx_in_lds)
        for (int e = tid; e < M * M; e += NT) XXl[e] = XXtr[e];
    double* Rm = model + ML.off_R;
    double* Qm = model + ML.off_Q;
    double* Wm = model + ML.off_W;
    double* Pm = model + ML.off_P;
    for (int e = tid; e < M * P; e += NT) XY[e % M + LX * (e / M)] = zwork[Z.off_XY[0] + e];
    if (tid < 4) rv[M + tid] = 0.0;
    constexpr bool XXREG_OK = (NW == 8) && (NB == 1);      // (two blocks per side: the eigen matrices take the registers)
    const bool xx_in_reg = XXREG_OK && !xx_in_lds && M <= 128;
    const int qb = (M + 3) / 4, qb8 = (M + 7) / 8;
    double xxq[XXREG_OK ? 32 : 1];
    if constexpr (XXREG_OK) {
        if (xx_in_reg) {
            const int a = tid % M, part = tid / M, b0 = part * qb;
#pragma unroll
            for (int i = 0; i < 32; i++) xxq[i] = (part < 4 && i < qb && b0 + i < M) ? XXtr[a + (size_t)M * (b0 + i)] : 0.0;
        }
    }
    __syncthreads();
#ifdef PLS_STAMPS
    long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_readcyclecounter();
#define STAMP16(id) do { const long long now_ = __builtin_readcyclecounter(); st_acc[id] += now_ - st_last; st_last = now_; } while (0)
#else
#define STAMP16(id)
#endif

    for (int comp = 0; comp <= A; comp++) {
        // ---- (1) close component comp - 1 (tt, p, q, stores, deflation) and contract the slabs into the partial S -------------
        double tt = 0.0, itt = 0.0, qc[NB];
#pragma unroll
        for (int b = 0; b < NB; b++) qc[b] = 0.0;
        if (comp > 0) {
            const int k = comp - 1;
            double t = 0.0;
            for (int a = l; a < M; a += 64) {
                const double xr = (xp[a] + xp[M + a]) + (xp[2 * M + a] + xp[3 * M + a]);
                t = fma(rv[a], xr, t);
            }
            tt = wave_sum(t);
            STAMP16(8);
            itt = 1.0 / tt;
            STAMP16(9);
#pragma unroll
            for (int b = 0; b < NB; b++) {
                const int col = c + 16 * b;
                if (col < n) {
                    double qs = 0.0;
#pragma unroll
                    for (int part = 0; part < 8; part++) qs += qp[part * P + col];
                    qc[b] = qs * itt;
                }
            }
            if (tid < n) {                       // (lanes 0 .. n - 1 of wave 0: q4 = tid >> 4 is the column block, c = tid & 15)
                const double qj = (NB == 1 || q4 == 0) ? qc[0] : qc[NB - 1];
                Qm[tid + (size_t)P * k] = qj;
                Ql[tid + (size_t)P * k] = qj;
            }
            for (int m = tid; m < M; m += NT) {
                const double xr = (xp[m] + xp[M + m]) + (xp[2 * M + m] + xp[3 * M + m]);
                const double pm = xr * itt, rm = rv[m];
                Pl[m + (size_t)M * k] = pm;
                Rl[m + (size_t)M * k] = rm;
                Pm[m + (size_t)M * k] = pm;
                Rm[m + (size_t)M * k] = rm;
                Wm[m + (size_t)M * k] = wn[m];
            }
            if (comp == A) break;
        }
        STAMP16(10);
        {
            d4 Dp[NB][NB];
#pragma unroll
            for (int I = 0; I < NB; I++)
#pragma unroll
                for (int J = 0; J < NB; J++) Dp[I][J] = (d4){0.0, 0.0, 0.0, 0.0};
            constexpr int SB = 2;                                  // slabs whose operands are fetched together
            for (int base = 4 * wave; base < M; base += 4 * NW * SB) {
                double av[SB][NB], xr[SB];
#pragma unroll
                for (int sI = 0; sI < SB; sI++) {
                    const int m = base + 4 * NW * sI + q4, mm = (m < M) ? m : 0;
#pragma unroll
                    for (int b = 0; b < NB; b++) av[sI][b] = XY[mm + LX * ((c + 16 * b < n) ? c + 16 * b : 0)];
                    xr[sI] = (comp > 0) ? (xp[mm] + xp[M + mm]) + (xp[2 * M + mm] + xp[3 * M + mm]) : 0.0;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sI = 0; sI < SB; sI++) {
                    const int m0 = base + 4 * NW * sI, m = m0 + q4;
                    if (m0 < M) {                                  // (wave-uniform)
                        double a[NB];
#pragma unroll
                        for (int b = 0; b < NB; b++) {
                            const int col = c + 16 * b;
                            const bool ok = m < M && col < n;
                            a[b] = ok ? av[sI][b] : 0.0;
                            if (comp > 0 && ok) {
                                a[b] -= tt * ((xr[sI] * itt) * qc[b]);
                                XY[m + LX * col] = a[b];
                            }
                        }
#pragma unroll
                        for (int I = 0; I < NB; I++)
#pragma unroll
                            for (int J = 0; J < NB; J++)
                                Dp[I][J] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[J], a[I], Dp[I][J], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int I = 0; I < NB; I++)
#pragma unroll
                for (int J = 0; J < NB; J++)
#pragma unroll
                    for (int r = 0; r < 4; r++) Sp[(((wave * NB + I) * NB + J) * 4 + r) * 64 + l] = Dp[I][J][r];
        }
        STAMP16(11);
        __syncthreads();
        STAMP16(1);
        // ---- (2) wave 0: dominant eigenvector of S = XY'XY, then w = XY q ----------------------------------------------------
        // Two blocks per side: the four blocks of a squaring are dealt out to waves 0..3 (8 dependent MFMAs each instead of 32 on
        // one wave), exchanged through LDS (two buffers in turn: one barrier per squaring); every wave reads the product back and
        // takes the trace itself, so the loop's control flow is uniform over the work-group without a flag.
        constexpr bool SPLITSQ = (NB == 2);
        d4 Sr[NB][NB], D[NB][NB];
        double tr = 0.0, itr = 0.0;
        const bool on_diag = (c & 3) == q4;                   // diagonal entry c of a block sits in register c >> 2 of lane c + 16 (c & 3)
        auto diag_of = [&](const d4& X) {
            const double lo = (c & 4) ? X[1] : X[0], hi = (c & 4) ? X[3] : X[2];
            return (c & 8) ? hi : lo;
        };
        auto trace_of = [&](const d4 (&X)[NB][NB]) {
            double t = 0.0;
            if (on_diag) {
#pragma unroll
                for (int I = 0; I < NB; I++) t += diag_of(X[I][I]);
            }
            return wave_sum(t);
        };
        if (SPLITSQ || wave == 0) {
#pragma unroll
            for (int I = 0; I < NB; I++)
#pragma unroll
                for (int J = 0; J < NB; J++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        double v = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; w++) v += Sp[(((w * NB + I) * NB + J) * 4 + r) * 64 + l];
                        Sr[I][J][r] = v;
                    }
            tr = trace_of(Sr);
            itr = (tr > 0.0) ? __builtin_amdgcn_rcp(tr) : 0.0;
#pragma unroll
            for (int I = 0; I < NB; I++)
#pragma unroll
                for (int J = 0; J < NB; J++)
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        D[I][J][r] = (tr > 0.0) ? Sr[I][J][r] * itr : ((l == 0 && r == 0 && I == 0 && J == 0) ? 1.0 : 0.0);
        }
        STAMP16(12);
        if constexpr (SPLITSQ) {
            const int bI = (wave >> 1) & 1, bJ = wave & 1;
            int buf = 0;
            for (int grp = 0; grp < 24; grp++) {
#pragma unroll 1
                for (int sq = 0; sq < 3; sq++) {
                    double* ex = Ex + buf * (NB * NB * 256);
                    if (wave < 4) {
                        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int K2 = 0; K2 < NB; K2++)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const double aop = bJ ? D[1][K2][r] : D[0][K2][r], bop = bI ? D[1][K2][r] : D[0][K2][r];
                                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, bop, acc, 0, 0, 0);
                            }
#pragma unroll
                        for (int r = 0; r < 4; r++) ex[((bI * NB + bJ) * 4 + r) * 64 + l] = acc[r];
                    }
                    __syncthreads();
#pragma unroll
                    for (int I = 0; I < NB; I++)
#pragma unroll
                        for (int J = 0; J < NB; J++)
#pragma unroll
                            for (int r = 0; r < 4; r++) D[I][J][r] = ex[((I * NB + J) * 4 + r) * 64 + l];
                    buf ^= 1;
                }
                const double t = trace_of(D);
                const double inv = __builtin_amdgcn_rcp(t);
#pragma unroll
                for (int I = 0; I < NB; I++)
#pragma unroll
                    for (int J = 0; J < NB; J++)
#pragma unroll
                        for (int r = 0; r < 4; r++) D[I][J][r] *= inv;
                if (t > 0.95) break;
            }
        } else if (wave == 0) {
            // groups of three squarings, trace normalisation, stop at trace(B^8) > 0.95 (see eig_square)
            for (int grp = 0; grp < 24; grp++) {
#pragma unroll 1
                for (int sq = 0; sq < 3; sq++) {
                    // (one accumulator per block: back-to-back dependent MFMAs forward it; two chains cost four adds and a wait)
                    d4 T2[NB][NB];
#pragma unroll
                    for (int I = 0; I < NB; I++)
#pragma unroll
                        for (int J = 0; J < NB; J++) {
                            d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                            for (int K2 = 0; K2 < NB; K2++)
#pragma unroll
                                for (int r = 0; r < 4; r++)
                                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(D[J][K2][r], D[I][K2][r], acc, 0, 0, 0);
                            T2[I][J] = acc;
                        }
#pragma unroll
                    for (int I = 0; I < NB; I++)
#pragma unroll
                        for (int J = 0; J < NB; J++) D[I][J] = T2[I][J];
                }
                const double t = trace_of(D);
                const double inv = __builtin_amdgcn_rcp(t);
#pragma unroll
                for (int I = 0; I < NB; I++)
#pragma unroll
                    for (int J = 0; J < NB; J++)
#pragma unroll
                        for (int r = 0; r < 4; r++) D[I][J][r] *= inv;
                if (t > 0.95) break;
            }
        }
        STAMP16(13);
        if (wave == 0) {
            // column of the converged power with the largest diagonal entry (ties -> lowest index) ...
            double dg = -1.0, dpay = 0.0;
            int best = c;
            if (on_diag) {
#pragma unroll
                for (int I = 0; I < NB; I++) {
                    const double v = diag_of(D[I][I]);
                    if (16 * I + c < n && v > dg) { dg = v; best = 16 * I + c; }
                }
            }
            wave_argmax(dg, best, dpay);
            // ... and one power step with S itself: (S B)[:, best] = row `best` of B S; the product of block row best >> 4 of B
            // with block column J of S leaves entries 16 J + c of that row in register (best & 15) >> 2 of the lanes
            // (c, best & 3)
            // (with the trace-1 copy of S, so that the entries stay of order one: q is NOT brought to unit length -- only its
            // direction and sign enter w = XY q, which is normalised in (3) -- that saves a reduction, a square root and a division)
            const int Ib = best >> 4, rb = best & 15;              // wave-uniform
            double yv[NB];
#pragma unroll
            for (int J = 0; J < NB; J++) {
                d4 T;
                if (tr > 0.0) {
                    d4 t0 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int K2 = 0; K2 < NB; K2++)
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const double bop = (NB == 1 || Ib == 0) ? D[0][K2][r] : D[NB - 1][K2][r];
                            t0 = __builtin_amdgcn_mfma_f64_16x16x4f64(bop, Sr[J][K2][r] * itr, t0, 0, 0, 0);
                        }
                    T = t0;
                } else {
                    T = (NB == 1 || Ib == 0) ? D[0][J] : D[NB - 1][J];
                }
                const int br = rb >> 2;
                yv[J] = (br == 0) ? T[0] : (br == 1) ? T[1] : (br == 2) ? T[2] : T[3];
            }
            const bool myrow = q4 == (rb & 3);
            double am = -1.0, sv = 0.0;
            int ai = c;
#pragma unroll
            for (int J = 0; J < NB; J++) {
                const bool mine = myrow && 16 * J + c < n;
                if (mine && fabs(yv[J]) > am) { am = fabs(yv[J]); ai = 16 * J + c; sv = yv[J]; }
            }
            wave_argmax(am, ai, sv);                               // largest |component| positive (ties -> lowest index)
#pragma unroll
            for (int J = 0; J < NB; J++)
                if (myrow && 16 * J + c < n) qv[16 * J + c] = (sv < 0.0) ? -yv[J] : yv[J];
            STAMP16(14);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int m = l; m < M; m += 64) wv[m] = dot_ahead<16>(XY + m, LX, qv, 1, P, 0.0);
        }
        STAMP16(2);
        __syncthreads();
        // ---- (3) |w| by every wave; projections on the earlier loadings, one wave per earlier component --------------------------
        double iww;
        {
            double ss = 0.0;
            for (int m = l; m < M; m += 64) ss = fma(wv[m], wv[m], ss);
            iww = 1.0 / sqrt(wave_sum(ss));
        }
        for (int m = tid; m < M; m += NT) wn[m] = wv[m] * iww;
        for (int j = wave; j < comp; j += NW) {
            double pw = 0.0;
            for (int m = l; m < M; m += 64) pw = fma(Pl[m + (size_t)M * j], wv[m], pw);
            pw = wave_sum(pw) * iww;
            if (l == 0) pwv[j] = pw;
        }
        __syncthreads();
        STAMP16(3);
        // ---- (4) r = w - sum_j (p_j'w) r_j --------------------------------------------------------------------------------------
        for (int m = tid; m < M; m += NT) rv[m] = wn[m] - dot_ahead<16>(pwv, 1, Rl + m, M, comp, 0.0);
        __syncthreads();
        STAMP16(4);
        // ---- (5) partial sums of X'X r (four threads per row) and of XY'r (eight per response, the last threads) ---------------
        {
            if (xx_in_reg) {
                if constexpr (XXREG_OK) {
                    if (tid < 4 * M) {
                        const double* rb = rv + (tid / M) * qb;              // (rv is padded with zeros up to 4 qb entries)
                        double rr[32];
#pragma unroll
                        for (int i = 0; i < 32; i++) rr[i] = rb[(i < qb) ? i : 0];
                        __builtin_amdgcn_sched_barrier(0);
                        double s = 0.0;
#pragma unroll
                        for (int i = 0; i < 32; i++) s = fma(xxq[i], rr[i], s);  // (entries past the quarter are zeros)
                        xp[tid] = s;
                    }
                }
            } else if (xx_in_lds) {
                for (int e = tid; e < 4 * M; e += NT) {
                    const int a = e % M, part = e / M;
                    const int b0 = part * qb, b1 = (b0 + qb < M) ? b0 + qb : M;
                    xp[e] = dot_ahead<16>(XXl + a + (size_t)M * b0, M, rv + b0, 1, b1 - b0, 0.0);
                }
            } else {
                const double* XXg = zwork + Z.off_XX[0];
                for (int e = tid; e < 4 * M; e += NT) {
                    const int a = e % M, part = e / M;
                    const int b0 = part * qb, b1 = (b0 + qb < M) ? b0 + qb : M;
                    xp[e] = dot_ahead<16>(XXg + a + (size_t)M * b0, M, rv + b0, 1, b1 - b0, 0.0);
                }
            }
            const int e = NT - 1 - tid;
            if (e < 8 * P) {
                const int j = e % P, part = e / P;
                const int m0 = part * qb8, m1 = (m0 + qb8 < M) ? m0 + qb8 : M;
                qp[e] = dot_ahead<16>(XY + m0 + LX * j, 1, rv + m0, 1, m1 - m0, 0.0);
            }
        }
        __syncthreads();
        STAMP16(5);
    }
    __syncthreads();
    STAMP16(6);

    // ---- PRESS on the validation statistics (as in k_pls_fit; the inner sums of the quadratic form one (response, component)
    // pair per thread, every chain with its operands fetched ahead) ------------------------------------------------------------
    const double* XYte = zwork + Z.off_XY[1];
    const double* XXte = zwork + Z.off_XX[1];
    const double* YYte = zwork + Z.off_YY;
    double* vk = Pl;                    // A*M (LDS: every operand of the phase but the validation statistics themselves is)
    double* H = Hl;
    double* cm = cml;
    double* addm = addl;                // P*A: sum_{l < a} q_jl H_al
    if (A * M >= 1024) {
        pls_gemm<NW>(XXte, 1, (size_t)M, Rl, 1, (size_t)M, M, A, M, vk, (size_t)M);            // vk = XXte R        (M x A)
        __syncthreads();
        pls_gemm<NW>(Rl, (size_t)M, 1, vk, 1, (size_t)M, A, A, M, H, (size_t)A);                 // H = R' vk          (A x A)
        pls_gemm<NW>(XYte, (size_t)M, 1, Rl, 1, (size_t)M, P, A, M, cm, (size_t)P, NW / 2);      // c = XYte' R        (P x A)
    } else {
        for (int e = tid; e < A * M; e += NT) {
            const int m = e % M, k = e / M;
            vk[e] = dot_ahead<16>(XXte + m, M, Rl + (size_t)M * k, 1, M, 0.0);
        }
        __syncthreads();
        for (int e = tid; e < A * A + P * A; e += NT) {
            if (e < A * A) {
                const int k = e % A, l2 = e / A;
                H[e] = dot_ahead<16>(Rl + (size_t)M * k, 1, vk + (size_t)M * l2, 1, M, 0.0);
            } else {
                const int e2 = e - A * A, j = e2 % P, k = e2 / P;
                cm[e2] = dot_ahead<16>(Rl + (size_t)M * k, 1, XYte + (size_t)M * j, 1, M, 0.0);
            }
        }
    }
    for (int m = tid; m < M; m += NT) {          // observed z-scores (kept in LDS for the scores below)
        const double sdv = model[ML.off_sd + m];
        const double z = (sdv == 0.0) ? 0.0 : (obs[m] - model[ML.off_mean + m]) / sdv;
        model[ML.off_zobs + m] = z;
        wv[m] = z;
    }
    __syncthreads();
    for (int e = tid; e < P * A; e += NT) {
        const int j = e % P, a = e / P;
        addm[e] = dot_ahead<16>(Ql + j, P, H + a, A, a, 0.0);
    }
    for (int k = tid; k < A; k += NT) model[ML.off_oscore + k] = dot_ahead<16>(wv, 1, Rl + (size_t)M * k, 1, M, 0.0);
    for (int e = tid; e < A * A; e += NT) model[ML.off_H + e] = H[e];           // (the Wilcoxon reduction's scales)
    __syncthreads();
    double* press = model + ML.off_press;   // A x P, column-major
    double* perl = qp;                      // P: per-response component counts
    for (int j = tid; j < P; j += NT) {
        double lin = 0.0, quad = 0.0;
        double best = 0.0; int besta = 0;
        const double yy = YYte[j];
        for (int a = 0; a < A; a++) {
            const double qa = Ql[j + (size_t)P * a];
            lin = fma(qa, cm[j + (size_t)P * a], lin);
            quad += 2.0 * qa * addm[j + (size_t)P * a] + qa * qa * H[a + (size_t)A * a];
            const double pr = yy - 2.0 * lin + quad;
            press[a + (size_t)A * j] = pr;
            if (a == 0 || pr < best) { best = pr; besta = a; }
        }
        model[ML.off_per + j] = (double)(besta + 1);
        perl[j] = (double)(besta + 1);
    }
    __syncthreads();
    if (tid == 0) {
        int ncomp = 1;
        for (int j = 0; j < P; j++) { const int v = (int)perl[j]; if (v > ncomp) ncomp = v; }
        model[ML.off_hdr] = (double)ncomp;
    }
#ifdef PLS_STAMPS
    STAMP16(7);
    if (tid == 0) {
        double* st_out = scratch + (size_t)A * M + (size_t)A * A + 2 * (size_t)P * A;
        for (int i = 0; i < 16; i++) st_out[i] = (double)st_acc[i];
    }
#endif
}

__global__ __launch_bounds__(64) void k_simple_obs(const double* __restrict__ obs, int M, int P,
                                                   double* __restrict__ model) {
    const ModelLayout ML = model_layout(M, P, 0);
    for (int m = threadIdx.x; m < M; m += 64) {
        const double sdv = model[ML.off_sd + m];
        model[ML.off_zobs + m] = (sdv == 0.0) ? 0.0 : (obs[m] - model[ML.off_mean + m]) / sdv;
    }
    if (threadIdx.x == 0) model[ML.off_hdr] = 0.0;
}

}  // namespace

#ifdef PLS_STAMPS
static double g_pls_stamps[64];
#endif

static int zstats_lds(abc_ctx* ctx, size_t M, size_t P, size_t* bytes);
// done (optional): an event behind the fit -- bound to the fit kernel's own completion signal where that kernel is the 16-lane
// one (a hipEventRecord is one more packet in front of the projection: ~7 us of the critical path), recorded otherwise
int launch_pls_model(abc_ctx* ctx, const double* stats, const double* obs, size_t M, size_t P, size_t A, int rule,
                     double* model, hipEvent_t done) {
    if (A < 1 || A > M) ABC_FAIL(ctx, ABC_ERR_INVALID, "pls: components A=%zu must be in [1, M=%zu]", A, M);
    StageTimer tm(ctx, ST_PLS_MODEL);
    const ZLayout Z = z_layout(M, P);
    double* zwork = (double*)abc_ws_alloc(ctx, Z.len * sizeof(double));
    double* scratch = (double*)abc_ws_alloc(ctx, (A * M + A * A + 2 * P * A + 64) * sizeof(double));
    if (!zwork || !scratch) ABC_FAIL(ctx, ABC_ERR_NOMEM, "pls: workspace exhausted");
    // (the z-scored cross-products: a launch of their own, or -- small sets on the latency-tuned fit -- that kernel's prologue)
    ABC_HIP(ctx, hipGetLastError());
    const size_t np = P;
    const int xx_in_lds = M <= 64;
    const size_t lds_d = M * P + 2 * np * np + np + 4 * M + 2 * M * A + (xx_in_lds ? M * M : 0) + 8 + (8 + A + 4 * M) +
                         (P > 64 ? np * np + np : 0) + (P <= 16 ? 8 * 256 : 0);
    size_t lds_bytes = lds_d * sizeof(double);
    // beyond the LDS (about 500 metrics at 16 parameters and 8 components) the same work arrays live in global memory: the
    // reference has no size limit here (PLS::Model on Eigen matrices); slower, one work-group either way
    double* gbase = nullptr;
    if (lds_bytes > 160 * 1024) {
        gbase = (double*)abc_ws_alloc(ctx, lds_bytes);
        if (!gbase) ABC_FAIL(ctx, ABC_ERR_NOMEM, "pls: workspace exhausted (%zu B of work arrays)", lds_bytes);
        lds_bytes = 64;
    }
    // up to 16 metrics ONE wavefront (no work-group barriers at all), four up to 64 (measured: 0.119 -> 0.105 ms at M = 32,
    // P = 16, A = 8; 0.308 -> 0.262 ms at M = 64, P = 32), eight beyond
#define PLS_LAUNCH(NW_, GM_, NB_) do { \
        ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_pls_fit<NW_, GM_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes)); \
        hipLaunchKernelGGL((k_pls_fit<NW_, GM_, NB_>), dim3(1), dim3(64 * NW_), lds_bytes, ctx->stream, zwork, obs, (int)M, (int)P, (int)A, \
                           model, scratch, xx_in_lds, gbase); } while (0)
#define PLS_LAUNCH_NB(NW_) do { if (P <= 16) PLS_LAUNCH(NW_, false, 1); else if (P <= 32) PLS_LAUNCH(NW_, false, 2); \
                                else PLS_LAUNCH(NW_, false, 0); } while (0)
    // 2..16 responses on more than one wave: the latency-tuned kernel, when its arrays fit the LDS
    const size_t nb16 = P <= 16 ? 1 : 2;
    const size_t lds16_d = (M + 3) * P + 16 * nb16 + 3 * M + 4 + 4 * M + 128 * nb16 + A + 2 * M * A + (M > 64 ? 8 : 4) * 256 * nb16 * nb16 + (xx_in_lds ? M * M : 0) + A * A + 3 * P * A + (nb16 == 2 ? 2048 : 0);
    const bool fit16 = P >= 2 && P <= 32 && M > 16 && lds16_d * sizeof(double) <= 160 * 1024;
    const bool fold_z = fit16 && M * (M + P) <= 4096;
    const double* stats_in = fold_z ? stats : nullptr;
    if (!fold_z) {
        const unsigned zblocks = (unsigned)((M * (M + P) + 255) / 256 > 128 ? 128 : (M * (M + P) + 255) / 256);      // one entry per thread and partition
        size_t zb = 0;
        ABC_TRY(zstats_lds(ctx, M, P, &zb));
        hipLaunchKernelGGL(k_zstats, dim3(zblocks), dim3(256), zb, ctx->stream, stats, (int)M, (int)P, (int)A, model, zwork);
        ABC_HIP(ctx, hipGetLastError());
    }
    if (fit16) {
        const int lb = (int)(lds16_d * sizeof(double));
#define FIT16_LAUNCH(NW_, NB_) do { \
            ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_pls_fit16<NW_, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, lb)); \
            hipExtLaunchKernelGGL((k_pls_fit16<NW_, NB_>), dim3(1), dim3(64 * NW_), lb, ctx->stream, nullptr, done, 0, (const double*)zwork, obs, (int)M, \
                                  (int)P, (int)A, model, scratch, xx_in_lds, stats_in, zwork); done = nullptr; } while (0)
        if (M > 64) { if (nb16 == 1) FIT16_LAUNCH(8, 1); else FIT16_LAUNCH(8, 2); }
        else { if (nb16 == 1) FIT16_LAUNCH(4, 1); else FIT16_LAUNCH(4, 2); }
#undef FIT16_LAUNCH
    } else
    if (gbase) {                                    // wide sets: eight waves in every case
        if (P <= 16) PLS_LAUNCH(8, true, 1); else PLS_LAUNCH(8, true, 0);
    } else if (M > 64) PLS_LAUNCH_NB(8);            // eight waves for the vector phases
    else if (M > 16) PLS_LAUNCH_NB(4);
    else PLS_LAUNCH_NB(1);                          // one wavefront
#undef PLS_LAUNCH_NB
#undef PLS_LAUNCH
    ABC_HIP(ctx, hipGetLastError());
    if (done) ABC_HIP(ctx, hipEventRecord(done, ctx->stream));          // (the other fit kernels: a record behind them)
#ifdef PLS_STAMPS
    ABC_HIP(ctx, hipMemcpyAsync(g_pls_stamps, scratch + A * M + A * A + 2 * P * A, 64 * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
#endif
    (void)rule;
    return ABC_OK;
}
#ifdef PLS_STAMPS
extern "C" void abc_debug_pls_stamps(double* out64) { for (int i = 0; i < 64; i++) out64[i] = g_pls_stamps[i]; }
#endif

// k_zstats keeps the two moment vectors of all M + P columns in dynamic LDS: beyond 64 KB (4096 columns) the launch needs the
// attribute, beyond the 160 KB of a CU (10240 columns) the kernel does not apply
static int zstats_lds(abc_ctx* ctx, size_t M, size_t P, size_t* bytes) {
    *bytes = 2 * (M + P) * sizeof(double);
    if (*bytes > 160 * 1024)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "z-score moments: %zu columns exceed the 10240 this kernel holds in LDS", M + P);
    if (*bytes > 64 * 1024)
        ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_zstats, hipFuncAttributeMaxDynamicSharedMemorySize, (int)*bytes));
    return ABC_OK;
}

int launch_simple_model(abc_ctx* ctx, const double* stats, const double* obs, size_t M, size_t P, double* model) {
    size_t zb = 0;
    ABC_TRY(zstats_lds(ctx, M, P, &zb));
    hipLaunchKernelGGL(k_zstats, dim3(1), dim3(256), zb, ctx->stream, stats, (int)M, (int)P, 0, model, (double*)nullptr);
    ABC_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_simple_obs, dim3(1), dim3(64), 0, ctx->stream, obs, (int)M, (int)P, model);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
