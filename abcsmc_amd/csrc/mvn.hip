// Proposal covariance of the selected particles and its Cholesky factor.
// Replaces ABC::setup_mvn_sampler (AbcUtil.cpp:462-488): [GSL] gsl_ran_multivariate_gaussian_vcov
// (n-1 covariance), diagonal doubled (:475-479), [GSL] gsl_linalg_cholesky_decomp1 (lower factor in
// place, strict upper triangle keeps the covariance entries).
// The K x P cross-products come from the same one-pass MFMA Gram kernel as the PLS statistics
// (gram.hip, theta treated as a P-column matrix); the P x P factorisation is one wavefront in LDS.
#include "abc_internal.h"

namespace {

// gA != NULL (more than 128 parameters: P x P doubles exceed the LDS): the work matrix lives there instead
__global__ __launch_bounds__(64) void k_cov_chol(const double* __restrict__ stats, int P, double* __restrict__ Lout,
                                                 int* __restrict__ status, double* __restrict__ gA) {
    extern __shared__ double Ash[];   // P x P column-major (or just delta when gA is given), then delta[P]
    double* A = gA ? gA : Ash;
    double* delta = gA ? Ash : Ash + (size_t)P * P;
#define CH_SYNC() do { if (gA) __threadfence_block(); __syncthreads(); } while (0)
    const StatsLayout SL = stats_layout(P, 0);
    const int lane = threadIdx.x;
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
#undef CH_SYNC
}

// dv_p = 2 * Var_p (n-1 denominator) from the same statistics record (AbcUtil.cpp:528-537)
__global__ void k_dv_from_stats(const double* __restrict__ stats, int P, double* __restrict__ dv) {
    const StatsLayout SL = stats_layout(P, 0);
    const int p = threadIdx.x;
    if (p >= P) return;
    const double n = stats[SL.off_n] + stats[SL.off_n + 1];
    const double d = (stats[SL.off_sum[0] + p] + stats[SL.off_sum[1] + p]) / n;
    double ss = stats[SL.off_G[0] + p + SL.C16 * p] + stats[SL.off_G[1] + p + SL.C16 * p] - n * d * d;
    if (ss < 0.0) ss = 0.0;
    dv[p] = (n > 1.0) ? 2.0 * (ss / (n - 1.0)) : 0.0;
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
