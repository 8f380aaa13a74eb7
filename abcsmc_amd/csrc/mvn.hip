// Proposal covariance of the selected particles and its Cholesky factor.
// Replaces ABC::setup_mvn_sampler (AbcUtil.cpp:462-488): [GSL] gsl_ran_multivariate_gaussian_vcov
// (n-1 covariance), diagonal doubled (:475-479), [GSL] gsl_linalg_cholesky_decomp1 (lower factor in
// place, strict upper triangle keeps the covariance entries).
// The K x P cross-products come from the same one-pass MFMA Gram kernel as the PLS statistics
// (gram.hip, theta treated as a P-column matrix); the P x P factorisation is one wavefront in LDS.
#include <stdlib.h>

#include "abc_internal.h"

namespace {

// covariance (n - 1), diagonal doubled, and its lower Cholesky factor from a statistics record: ONE wavefront; A: P x P
// work matrix, delta: P doubles (LDS, or global memory behind fences).  SYNC orders the wave's own memory operations.
#define CH_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
__device__ __forceinline__ void cov_chol_wave(const double* __restrict__ stats, int P, double* A, double* delta,
                                              double* __restrict__ Lout, int* __restrict__ status, int lane) {
    const StatsLayout SL = stats_layout(P, 0);
    const double n = stats[SL.off_n] + stats[SL.off_n + 1];
    for (int c = lane; c < P; c += 64) delta[c] = (stats[SL.off_sum[0] + c] + stats[SL.off_sum[1] + c]) / n;
    CH_SYNC();
    for (int e = lane; e < P * P; e += 64) {
        const int a = e % P, b = e / P;
        const double g = stats[SL.off_G[0] + a + SL.C16 * b] + stats[SL.off_G[1] + a + SL.C16 * b];
        double c = (g - n * delta[a] * delta[b]) / (n - 1.0);
        if (a == b) c = 2.0 * c;                       // AbcUtil.cpp:475-479
        A[e] = c;
    }
    CH_SYNC();
    int ok = 1;
    for (int j = 0; j < P; j++) {
        for (int i = j + lane; i < P; i += 64) {
            double temp = 0.0;
            for (int k = 0; k < j; k++) temp += A[j + P * k] * A[i + P * k];
            A[i + P * j] += -1.0 * temp;
        }
        CH_SYNC();
        const double ajj = A[j + P * j];
        if (!(ajj > 0.0)) { ok = 0; break; }           // GSL_EDOM in the reference (process abort)
        const double inv = 1.0 / sqrt(ajj);
        CH_SYNC();
        for (int i = j + lane; i < P; i += 64) A[i + P * j] *= inv;
        CH_SYNC();
    }
    for (int e = lane; e < P * P; e += 64) Lout[e] = A[e];
    if (lane == 0) *status = ok ? 0 : -1;
    CH_SYNC();
}

// The same factorisation with the matrix in REGISTERS (up to 32 parameters): lane i holds row i, the entries of row j that
// column j's update needs come through v_readlane as scalar operands -- no LDS round trip per product, no barrier per column
// (the LDS version: ~120 dependent load-multiply-add steps and 48 wave barriers at 16 parameters, most of k_post_tail's 17 us on
// the critical path of a first set).  Same operations in the same order as above (GSL's cholesky_decomp1: temp += a * b with
// separate multiply and add, A_ij -= temp, the column scaled by 1 / sqrt(A_jj)).  Rows and columns beyond P: identity.
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <int PP>
__device__ __forceinline__ void cov_chol_regs(const double* __restrict__ stats, int P, double* __restrict__ Lout, int* __restrict__ status,
                                              double* __restrict__ Lpad, int lane) {
    const StatsLayout SL = stats_layout(P, 0);
    const double n = stats[SL.off_n] + stats[SL.off_n + 1];
    const bool row = lane < P;
    const double di = row ? (stats[SL.off_sum[0] + lane] + stats[SL.off_sum[1] + lane]) / n : 0.0;
    double a[PP];
#pragma unroll
    for (int b = 0; b < PP; b++) {
        const double db = lane_bcast(di, b);                       // (lanes >= P hold 0; columns >= P are not used below)
        double c = (lane == b) ? 1.0 : 0.0;
        if (row && b < P) {
            const double g = stats[SL.off_G[0] + lane + SL.C16 * b] + stats[SL.off_G[1] + lane + SL.C16 * b];
            c = (g - n * di * db) / (n - 1.0);
            if (lane == b) c = 2.0 * c;                            // AbcUtil.cpp:475-479
        }
        a[b] = c;
    }
    int ok = 1;
#pragma unroll
    for (int j = 0; j < PP; j++) {
        double temp = 0.0;
#pragma unroll
        for (int k = 0; k < j; k++) temp += lane_bcast(a[k], j) * a[k];
        if (lane >= j) a[j] += -1.0 * temp;
        const double ajj = lane_bcast(a[j], j);
        if (j < P && !(ajj > 0.0)) { ok = 0; break; }              // GSL_EDOM in the reference (process abort); wave-uniform
        const double inv = 1.0 / sqrt(ajj);
        if (lane >= j) a[j] *= inv;
    }
    if (row) {
#pragma unroll
        for (int b = 0; b < PP; b++) if (b < P) Lout[lane + P * b] = a[b];
    }
    if (Lpad && lane < PP) {
#pragma unroll
        for (int b = 0; b < PP; b++) Lpad[lane + PP * b] = (row && b < P && b <= lane) ? a[b] : 0.0;
    }
    if (lane == 0) *status = ok ? 0 : -1;
}

// gA != NULL (more than 128 parameters: P x P doubles exceed the LDS): the work matrix lives there instead
__global__ __launch_bounds__(64) void k_cov_chol(const double* __restrict__ stats, int P, double* __restrict__ Lout,
                                                 int* __restrict__ status, double* __restrict__ gA) {
    extern __shared__ double Ash[];   // P x P column-major (or just delta when gA is given), then delta[P]
    double* A = gA ? gA : Ash;
    double* delta = gA ? Ash : Ash + (size_t)P * P;
    cov_chol_wave(stats, P, A, delta, Lout, status, (int)threadIdx.x);
}

__device__ __forceinline__ double dv_of_stats(const double* __restrict__ stats, int P, int p) {
    const StatsLayout SL = stats_layout(P, 0);
    const double n = stats[SL.off_n] + stats[SL.off_n + 1];
    const double d = (stats[SL.off_sum[0] + p] + stats[SL.off_sum[1] + p]) / n;
    double ss = stats[SL.off_G[0] + p + SL.C16 * p] + stats[SL.off_G[1] + p + SL.C16 * p] - n * d * d;
    if (ss < 0.0) ss = 0.0;
    return (n > 1.0) ? 2.0 * (ss / (n - 1.0)) : 0.0;
}

// ---- everything that follows from the posterior's statistics record, ONE launch ------------------------------------------------
// Round 2 ran three dependent launches behind the record (doubled variance, covariance + Cholesky, row-major copy + padded
// factor: 27 us of the 49 us the posterior's moments took, on the critical path of set 0).  Here work-group 0 computes the
// doubled variance (AbcUtil.cpp:528-537), the proposal factor (AbcUtil.cpp:462-488), its status and its padded copy, while
// work-groups 1.. write the padded row-major copy of the rows (k_perturb's parent table), which does not depend on the record.
// (Also measured: the moments themselves by a direct kernel -- 128 groups, wave-level cross-product columns -- instead of pilot
// shift + MFMA Gram + reduce: 47 + 34 us against 22, too few waves in flight; and any "last group finishes" ticket costs a
// device-scope fence per group, i.e. an L2 write-back: 131 us.)
template <int PP>
__global__ __launch_bounds__(256) void k_post_tail(const double* __restrict__ theta, size_t K, int P, const double* __restrict__ stats,
                                                   double* __restrict__ dv, double* __restrict__ Lout, int* __restrict__ spd,
                                                   double* __restrict__ rows, double* __restrict__ Lpad,
                                                   const double* __restrict__ model_hdr, double* __restrict__ hdr_pin, int* __restrict__ spd_pin) {
    __shared__ double t[PP][65];                    // group 0: the P x P work matrix + P means; others: the transposed tile
    if (blockIdx.x == 0) {
        double* sA = &t[0][0];
        if (dv) for (int p = threadIdx.x; p < P; p += 256) dv[p] = dv_of_stats(stats, P, p);
        if (model_hdr && hdr_pin && threadIdx.x >= 64 && threadIdx.x < 68) hdr_pin[threadIdx.x - 64] = model_hdr[threadIdx.x - 64];
        if (Lout && threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if constexpr (PP <= 32) {
                cov_chol_regs<PP>(stats, P, Lout, spd, Lpad, lane);
            } else {
                cov_chol_wave(stats, P, sA, sA + P * P, Lout, spd, lane);
                if (Lpad) {
                    for (int e = lane; e < PP * PP; e += 64) {
                        const int a = e % PP, b = e / PP;
                        Lpad[e] = (a < P && b < P && b <= a) ? sA[a + P * b] : 0.0;
                    }
                }
            }
            if (spd_pin && lane == 0) *spd_pin = *spd;          // (lane 0 wrote it itself)
        }
        return;
    }
    if (!rows) return;
    const size_t k0 = (size_t)(blockIdx.x - 1) * 64;
    for (int e = threadIdx.x; e < PP * 64; e += 256) {
        const int p = e >> 6, r = e & 63;
        t[p][r] = (p < P && k0 + r < K) ? theta[k0 + r + K * (size_t)p] : 0.0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PP * 64; e += 256) {
        const int r = e / PP, p = e % PP;
        if (k0 + r < K) rows[(k0 + r) * PP + p] = t[p][r];
    }
}

// dv_p = 2 * Var_p (n-1 denominator) from the same statistics record (AbcUtil.cpp:528-537)
__global__ void k_dv_from_stats(const double* __restrict__ stats, int P, double* __restrict__ dv) {
    const int p = threadIdx.x;
    if (p >= P) return;
    dv[p] = dv_of_stats(stats, P, p);
}

}  // namespace

// one pass over the K x P posterior: pilot shift + Gram (k_gram<.,0>) -> statistics record in the arena
int launch_theta_stats(abc_ctx* ctx, const double* theta, size_t K, size_t P, double** stats_out) {
    const StatsLayout SL = stats_layout(P, 0);
    double* stats = (double*)abc_ws_alloc(ctx, SL.len * sizeof(double));
    if (!stats) ABC_FAIL(ctx, ABC_ERR_NOMEM, "theta stats: workspace exhausted");
    ABC_TRY(launch_stats_shift(ctx, theta, theta, K, K, K, P, 0, stats));
    ctx->in_mvn = true;
    const int rc_acc = launch_stats_accumulate(ctx, theta, theta, K, K, K, P, 0, 0, K, stats);
    ctx->in_mvn = false;
    ABC_TRY(rc_acc);
    *stats_out = stats;
    return ABC_OK;
}

// what follows from the record, one launch (P <= 64): every output optional; rows: K x PP row-major, Lpad: PP x PP (abc_perturb_pp)
int launch_post_tail(abc_ctx* ctx, const double* theta, size_t K, size_t P, const double* stats, const abc_theta_fused* f) {
    if (P > 64) ABC_FAIL(ctx, ABC_ERR_INVALID, "post tail: more than 64 parameters");
    const int PP = abc_perturb_pp(P);
    const unsigned grid = 1u + (f->rows ? (unsigned)((K + 63) / 64) : 0u);
#define PT_LAUNCH(PPV) hipLaunchKernelGGL(k_post_tail<PPV>, dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, stats, f->dv, f->L, f->spd, \
                                          f->rows, f->Lpad, f->model_hdr, f->hdr_pin, f->spd_pin)
    switch (PP) {
        case 2: PT_LAUNCH(2); break;
        case 4: PT_LAUNCH(4); break;
        case 8: PT_LAUNCH(8); break;
        case 16: PT_LAUNCH(16); break;
        case 32: PT_LAUNCH(32); break;
        default: PT_LAUNCH(64); break;
    }
#undef PT_LAUNCH
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_dv_from_stats(abc_ctx* ctx, const double* stats, size_t P, double* dv) {
    hipLaunchKernelGGL(k_dv_from_stats, dim3(1), dim3(256), 0, ctx->stream, stats, (int)P, dv);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_mvn_from_stats(abc_ctx* ctx, const double* stats, size_t P, double* L, int* status_dev) {
    double* gA = nullptr;
    size_t lds = (P * P + P) * sizeof(double);
    if (lds > 150 * 1024) {            // beyond the LDS: the work matrix in the arena
        gA = (double*)abc_ws_alloc(ctx, P * P * sizeof(double));
        if (!gA) ABC_FAIL(ctx, ABC_ERR_NOMEM, "mvn: workspace exhausted");
        lds = P * sizeof(double);
    }
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_cov_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_cov_chol, dim3(1), dim3(64), lds, ctx->stream, stats, (int)P, L, status_dev, gA);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_mvn_setup(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* L, int* status_host,
                     int* status_dev) {
    if (K < 2) ABC_FAIL(ctx, ABC_ERR_INVALID, "mvn: need at least 2 particles (K=%zu)", K);
    int* status = status_dev ? status_dev : (int*)abc_ws_alloc(ctx, sizeof(int));
    if (!status) ABC_FAIL(ctx, ABC_ERR_NOMEM, "mvn: workspace exhausted");
    StageTimer tm(ctx, ST_MVN);
    double* stats = nullptr;
    ABC_TRY(launch_theta_stats(ctx, theta, K, P, &stats));
    ABC_TRY(launch_mvn_from_stats(ctx, stats, P, L, status));
    if (status_host) {
        ABC_HIP(ctx, hipMemcpyAsync(status_host, status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return ABC_OK;
}
