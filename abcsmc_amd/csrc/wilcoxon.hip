// Wilcoxon signed-rank reduction of the number of PLS components ([PLS] optimal_num_components, called at
// AbcUtil.cpp:447-449): for every response j whose PRESS optimum is a*_j > 1, the
// smallest a' < a*_j whose absolute validation errors are not significantly different (two-sided, normal
// approximation, alpha = 0.1, "average" tie ranks as lib/ranker.h:66-76, zero differences dropped) replaces it.
//
// All (j, a') tests are batched: one pass over the validation rows writes, per test ("segment"), the key
// |d_i| = ||e_a*(i,j)| - |e_a'(i,j)|| (IEEE bit pattern; zero differences get the maximal key) and a payload
// (segment, sign); one stable LSD radix sort by key then by segment groups every segment in ascending |d|;
// rank sums are sums of half-integers (< 2^52), hence exact and order-independent, so integer/double atomics
// stay bit-reproducible.  Residuals use the same fixed fma order as the oracle.
#include "abc_internal.h"

namespace {

constexpr size_t MAXSEG = 65535;      // (response, candidate) tests of one call: the segment id travels in two payload bytes of the
                                      // batched sort and in grid.y; P (A - 1) beyond that is ABC_ERR_UNSUPPORTED

struct WxPlan {            // built on the device from the model record; the arrays live in the workspace (P (A - 1), P entries)
    int nseg;
    int pad_;
    int* seg_j;            // response
    int* seg_a;            // candidate a' (1-based)
    int* astar;            // PRESS optimum per response
};

__global__ void k_wx_plan(const double* __restrict__ model, int M, int P, int A, WxPlan* __restrict__ plan,
                          int* __restrict__ seg_j, int* __restrict__ seg_a, int* __restrict__ astar, int nseg_max,
                          unsigned long long* __restrict__ nz, double* __restrict__ W) {
    for (int s = threadIdx.x; s < nseg_max; s += blockDim.x) { nz[s] = 0; W[s] = 0.0; }
    if (threadIdx.x != 0) return;
    const ModelLayout ML = model_layout(M, P, A);
    int ns = 0;
    for (int j = 0; j < P; j++) {
        const int as = (int)model[ML.off_per + j];
        astar[j] = as;
        for (int a = 1; a < as && ns < nseg_max; a++) { seg_j[ns] = j; seg_a[ns] = a; ns++; }
    }
    plan->nseg = ns;
    plan->pad_ = 0;
    plan->seg_j = seg_j;
    plan->seg_a = seg_a;
    plan->astar = astar;
}

// scores of the validation rows: S[i + nt*k] = sum_m z(x_im) R[m,k]  (m ascending fma chain, as the oracle)
template <int KC>
__global__ __launch_bounds__(256) void k_wx_scores(const double* __restrict__ X, size_t ldx, size_t row_test, size_t nt,
                                                   int M, int P, int A, const double* __restrict__ model,
                                                   double* __restrict__ S) {
    const ModelLayout ML = model_layout(M, P, A);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    double s[KC];
#pragma unroll
    for (int k = 0; k < KC; k++) s[k] = 0.0;
    for (int m = 0; m < M; m++) {
        const double sd = model[ML.off_sd + m];
        const double z = (sd == 0.0) ? 0.0 : (X[row_test + i + ldx * m] - model[ML.off_mean + m]) / sd;
#pragma unroll
        for (int k = 0; k < KC; k++)
            if (k < A) s[k] = fma(z, model[ML.off_R + m + (size_t)M * k], s[k]);
    }
#pragma unroll
    for (int k = 0; k < KC; k++)
        if (k < A) S[i + nt * k] = s[k];
}

// more than 32 components: chunks of 32 (the row's metrics re-read per chunk; same fma chain per component)
__global__ __launch_bounds__(256) void k_wx_scores_wide(const double* __restrict__ X, size_t ldx, size_t row_test, size_t nt,
                                                        int M, int P, int A, const double* __restrict__ model,
                                                        double* __restrict__ S) {
    const ModelLayout ML = model_layout(M, P, A);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    for (int k0 = 0; k0 < A; k0 += 32) {
        double s[32];
#pragma unroll
        for (int k = 0; k < 32; k++) s[k] = 0.0;
        for (int m = 0; m < M; m++) {
            const double sd = model[ML.off_sd + m];
            const double z = (sd == 0.0) ? 0.0 : (X[row_test + i + ldx * m] - model[ML.off_mean + m]) / sd;
#pragma unroll
            for (int k = 0; k < 32; k++)
                if (k0 + k < A) s[k] = fma(z, model[ML.off_R + m + (size_t)M * (k0 + k)], s[k]);
        }
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (k0 + k < A) S[i + nt * (k0 + k)] = s[k];
    }
}

// one thread per (validation row, segment): key / payload of the paired difference
__global__ __launch_bounds__(256) void k_wx_diffs(const double* __restrict__ Y, size_t ldy, size_t row_test, size_t nt,
                                                  int M, int P, int A, const double* __restrict__ model,
                                                  const double* __restrict__ S, const WxPlan* __restrict__ plan,
                                                  unsigned long long* __restrict__ key,
                                                  unsigned long long* __restrict__ val,
                                                  unsigned long long* __restrict__ nz) {
    const ModelLayout ML = model_layout(M, P, A);
    const int seg = blockIdx.y;
    if (seg >= plan->nseg) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int j = plan->seg_j[seg], a1 = plan->seg_a[seg], as = plan->astar[j];
    unsigned int nonzero = 0;
    if (i < nt) {
        const double sdy = model[ML.off_sd + M + j];
        const double zy = (sdy == 0.0) ? 0.0 : (Y[row_test + i + ldy * j] - model[ML.off_mean + M + j]) / sdy;
        double pred = 0.0, e_small = 0.0;
        for (int k = 0; k < as; k++) {
            pred = fma(S[i + nt * k], model[ML.off_Q + j + (size_t)P * k], pred);
            if (k + 1 == a1) e_small = zy - pred;
        }
        const double e_star = zy - pred;
        const double d = fabs(e_star) - fabs(e_small);
        const unsigned long long sign = d > 0.0 ? 1ull : 0ull;
        nonzero = d != 0.0;
        key[(size_t)seg * nt + i] = nonzero ? (unsigned long long)__double_as_longlong(fabs(d)) : ~0ull;
        val[(size_t)seg * nt + i] = ((unsigned long long)seg << 32) | (sign << 31);
    }
    const unsigned long long m = __ballot(nonzero);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&nz[seg], (unsigned long long)__popcll(m));
}

// after sorting by (segment, |d|): signed sum of average ranks of the non-zero differences of every segment
__global__ __launch_bounds__(256) void k_wx_ranksum(const unsigned long long* __restrict__ key,
                                                    const unsigned long long* __restrict__ val, size_t nt,
                                                    const WxPlan* __restrict__ plan,
                                                    const unsigned long long* __restrict__ nz, double* __restrict__ W) {
    const int seg = blockIdx.y;
    if (seg >= plan->nseg) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)nz[seg];
    double w = 0.0;
    if (i < n) {
        const unsigned long long* k = key + (size_t)seg * nt;
        const unsigned long long me = k[i];
        size_t lo = i, hi = i;                       // tie run [lo, hi] (ties of doubles are rare: short scans)
        while (lo > 0 && k[lo - 1] == me) lo--;
        while (hi + 1 < n && k[hi + 1] == me) hi++;
        const double rank = (double)(lo + hi) / 2.0 + 1.0;             // ranker.h:74-75 "average"
        const bool pos = (val[(size_t)seg * nt + i] >> 31) & 1ull;
        w = pos ? rank : -rank;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o, 64);       // exact: multiples of 1/2 below 2^52
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(&W[seg], w);
}

__device__ double normalcdf_poly(double z) {        // [PLS] normalcdf, Abramowitz & Stegun 26.2.18
    const double c1 = 0.196854, c2 = 0.115194, c3 = 0.000344, c4 = 0.019527;
    const double x = fabs(z);
    const double d = 1.0 + c1 * x + c2 * x * x + c3 * x * x * x + c4 * x * x * x * x;
    const double tail = 0.5 / (d * d * d * d);
    return (z >= 0.0) ? 1.0 - tail : tail;
}

__global__ void k_wx_decide(double* __restrict__ model, int M, int P, int A, const WxPlan* __restrict__ plan,
                            const unsigned long long* __restrict__ nz, const double* __restrict__ W) {
    if (threadIdx.x != 0) return;
    const ModelLayout ML = model_layout(M, P, A);
    int ncomp = 1, s = 0;
    for (int j = 0; j < P; j++) {
        int best = plan->astar[j];
        bool found = false;
        for (int a = 1; a < plan->astar[j]; a++, s++) {
            if (found || s >= plan->nseg) continue;
            const double m = (double)nz[s];
            double p = 1.0;
            if (m > 0.0) {
                const double sigma = sqrt(m * (m + 1.0) * (2.0 * m + 1.0) / 6.0);
                p = 2.0 * (1.0 - normalcdf_poly(fabs(W[s] / sigma)));
            }
            if (p > 0.1) { best = a; found = true; }
        }
        model[ML.off_per + j] = (double)best;
        if (best > ncomp) ncomp = best;
    }
    model[ML.off_hdr] = (double)ncomp;
}

// observed scores do not depend on ncomp (all A are stored), nothing else to refresh

}  // namespace

int launch_wilcoxon(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                    size_t P, size_t A, size_t row_test, double* model) {
    if (row_test >= n) return ABC_OK;                    // empty validation set: nothing to reduce
    if (P * (A - 1) > MAXSEG)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "wilcoxon: P (A - 1) = %zu tests, more than %zu", P * (A - 1), MAXSEG);
    StageTimer tm(ctx, ST_PLS_MODEL);
    const size_t nt = n - row_test;
    const size_t nseg_max = P * (A - 1);
    if (nseg_max == 0) return ABC_OK;
    WxPlan* plan = (WxPlan*)abc_ws_alloc(ctx, sizeof(WxPlan));
    int* seg_j = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    int* seg_a = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    int* astar = (int*)abc_ws_alloc(ctx, P * sizeof(int));
    unsigned long long* nz = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * 8);
    double* W = (double*)abc_ws_alloc(ctx, nseg_max * 8);
    double* S = (double*)abc_ws_alloc(ctx, nt * A * 8);
    unsigned long long* key0 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    unsigned long long* val0 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    unsigned long long* key1 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    unsigned long long* val1 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    if (!plan || !seg_j || !seg_a || !astar || !nz || !W || !S || !key0 || !val0 || !key1 || !val1)
        ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%zu segments x %zu rows)", nseg_max, nt);
    hipLaunchKernelGGL(k_wx_plan, dim3(1), dim3(256), 0, ctx->stream, model, (int)M, (int)P, (int)A, plan, seg_j, seg_a, astar,
                       (int)nseg_max, nz, W);
    // the number of segments actually needed lives on the device; size the grid for the maximum, idle blocks exit
    int nseg_host = 0;
    ABC_HIP(ctx, hipMemcpyAsync(&nseg_host, &plan->nseg, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nseg_host == 0) return ABC_OK;
    const unsigned rb = (unsigned)((nt + 255) / 256);
    int KC = 1;
    while (KC < (int)A) KC *= 2;
    if (A > 32) {
        hipLaunchKernelGGL(k_wx_scores_wide, dim3(rb), dim3(256), 0, ctx->stream, X, ldx, row_test, nt, (int)M, (int)P, (int)A, model, S);
        KC = 0;
    }
#define LAUNCH_SC(KCV) hipLaunchKernelGGL(k_wx_scores<KCV>, dim3(rb), dim3(256), 0, ctx->stream, X, ldx, row_test, nt, \
                                          (int)M, (int)P, (int)A, model, S)
    switch (KC) {
        case 0: break;
        case 1: LAUNCH_SC(1); break;
        case 2: LAUNCH_SC(2); break;
        case 4: LAUNCH_SC(4); break;
        case 8: LAUNCH_SC(8); break;
        case 16: LAUNCH_SC(16); break;
        default: LAUNCH_SC(32); break;
    }
#undef LAUNCH_SC
    hipLaunchKernelGGL(k_wx_diffs, dim3(rb, nseg_host), dim3(256), 0, ctx->stream, Y, ldy, row_test, nt, (int)M, (int)P,
                       (int)A, model, S, plan, key0, val0, nz);
    ABC_HIP(ctx, hipGetLastError());
    const size_t tot = (size_t)nseg_host * nt;
    // ascending |d| (8 byte passes), then stable by segment (payload bytes 4-5) -> segment-major, |d| ascending
    ABC_TRY(abc_sort_u64_bytes(ctx, key0, val0, key1, val1, tot, 0, 8));
    ABC_TRY(abc_sort_u64_bytes(ctx, val0, key0, val1, key1, tot, 4, 6));
    hipLaunchKernelGGL(k_wx_ranksum, dim3(rb, nseg_host), dim3(256), 0, ctx->stream, key0, val0, nt, plan, nz, W);
    hipLaunchKernelGGL(k_wx_decide, dim3(1), dim3(64), 0, ctx->stream, model, (int)M, (int)P, (int)A, plan, nz, W);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
