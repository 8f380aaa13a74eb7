// Posterior bookkeeping of one SMC set: gather of the K selected parameter rows, doubled variance,
// Gaussian-kernel importance weights and their L2 normalisation.
// Replaces ABC::calculate_doubled_variance (AbcUtil.cpp:528-537, RunningStat.h:16-46),
// ABC::weight_predictive_prior (AbcUtil.cpp:539-586), Parameter::likelihood (Priors.h:54-56, 76-78,
// 102-104) and the Eigen row gathers of AbcSmc.cpp:1045-1060.
//
// The weight kernel is the compute-bound stage of a generation, O(K * K' * P): per pair the P
// per-parameter Gaussian factors of the reference are fused into ONE exponential,
//   prod_p pdf(t_ip - t'_jp; sqrt(dv_p)) = C * exp(-1/2 sum_p ((t_ip - t'_jp)/sigma_p)^2),
// with both parameter sets centred on one previous particle and pre-scaled by 1/sigma_p, and the squared
// distance expanded as |a|^2 + |b|^2 - 2 a.b so a pair costs P FMAs + one exp:
//   w'_j * exp(-1/2 |a_i - b_j|^2) = exp(a_i.b_j - 1/2|a_i|^2 - (1/2|b_j|^2 - ln w'_j)).
// Both sets are additionally scaled by sqrt(log2 e), so the exponent comes out in base 2 and the exponential is
//   2^x = 2^n * 2^f,  n = round(x) by the 1.5*2^52 addition (its low dword IS n), f = x - n in [-1/2, 1/2],
// 2^f a degree-8 minimax polynomial (|rel err| < 7.8e-13, scripts/exp2_minimax.py; far inside the 1e-6 budget):
// 13 instructions for the exponential, 30 per pair at P = 16, instead of ~40 for the library exp alone.
// One new particle per lane; previous-set rows are wave-uniform and stream through the scalar cache.
// fp64 VALU throughout.
#include "abc_internal.h"

namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ double block_sum_256(double v, double* sm /* >= 4 */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ __launch_bounds__(256) void k_gather_rows(const double* __restrict__ Y, size_t n_local, size_t ldy,
                                                     int P, const unsigned long long* __restrict__ idx, size_t K,
                                                     unsigned long long idx_base, double* __restrict__ theta,
                                                     size_t ldt) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= K * (size_t)P) return;
    const size_t i = e % K, p = e / K;
    const unsigned long long g = idx[i];
    if (g >= idx_base && g - idx_base < n_local) theta[i + ldt * p] = Y[(g - idx_base) + ldy * p];
}

// one work-group per parameter: dv_p = 2 * sum (x - mean)^2 / (K - 1)
__global__ __launch_bounds__(256) void k_doubled_variance(const double* __restrict__ theta, size_t K,
                                                          double* __restrict__ dv) {
    __shared__ double sm[4];
    const double* col = theta + K * (size_t)blockIdx.x;
    double s = 0.0;
    for (size_t i = threadIdx.x; i < K; i += 256) s += col[i];
    const double mean = block_sum_256(s, sm) / (double)K;
    double ss = 0.0;
    for (size_t i = threadIdx.x; i < K; i += 256) { const double d = col[i] - mean; ss = fma(d, d, ss); }
    ss = block_sum_256(ss, sm);
    if (threadIdx.x == 0) dv[blockIdx.x] = (K > 1) ? 2.0 * (ss / (double)(K - 1)) : 0.0;
}

// ---- weights --------------------------------------------------------------------------------------
struct WConst {           // per-parameter constants, built on the device by k_wprep
    double scale[64];     // sqrt(log2 e)/sqrt(dv_p), or 0 when dv_p == 0
    double logC;          // unused
    double C;             // prod over dv_p != 0 of 1/(sqrt(2 pi) sqrt(dv_p))
    int nzero;            // number of parameters with dv_p == 0
    int zero_idx[64];
    int far;              // set by k_wscale when a scaled coordinate is so large that exponents may leave int32
};

constexpr double W_SQRT_LOG2E = 1.2011224087864497825;     // sqrt(log2 e): a.b then comes out in base 2
constexpr double W_COORD_BOUND = 2000.0;                   // |x| <= 2 PP B^2 + 1e8 < 2^31 for PP <= 64
constexpr double W_HB_MAX = 1.0e8;                         // stands for "weight 0": 2^-1e8 == 0

__global__ void k_wprep(const double* __restrict__ dv_prev, int P, WConst* __restrict__ wc) {
    if (threadIdx.x != 0) return;
    double C = 1.0; int nz = 0;
    for (int p = 0; p < 64; p++) {
        double sc = 0.0;
        if (p < P) {
            const double dv = dv_prev[p];
            if (dv != 0.0) { const double sg = sqrt(dv); sc = 1.0 / sg; C *= 1.0 / (sqrt(2.0 * M_PI) * sg); }
            else wc->zero_idx[nz++] = p;
        }
        wc->scale[p] = sc * W_SQRT_LOG2E;
    }
    wc->C = C; wc->nzero = nz; wc->logC = 0.0; wc->far = 0;
}

// scaled copies: out[row*PP + p] = (in[row + ld*p] - centre[p]) * scale[p]   (row-major, zero padded to PP);
// centre = first previous particle (differences are unchanged, magnitudes stay O(few sigma)).
// If hb != NULL also hb[row] = 1/2 |out[row,:]|^2 - log2(w[row])  (the per-column part of the base-2 exponent),
// with -log2 w capped at W_HB_MAX (w = 0 -> the term vanishes).  Rows further than W_COORD_BOUND from the centre raise wc->far.
__global__ __launch_bounds__(256) void k_wscale(const double* __restrict__ in, size_t rows, size_t ld, int P, int PP,
                                                WConst* __restrict__ wc, const double* __restrict__ centre,
                                                size_t ldc, const double* __restrict__ w, double* __restrict__ out,
                                                double* __restrict__ hb) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    double nn = 0.0;
    bool far = false;
    for (int p = 0; p < PP; p++) {
        const double v = (p < P) ? (in[r + ld * p] - centre[ldc * p]) * wc->scale[p] : 0.0;
        out[r * PP + p] = v;
        nn = fma(v, v, nn);
        far = far || (fabs(v) > W_COORD_BOUND);
    }
    if (hb) {
        const double lw = -log2(w[r]);                   // w = 0 -> +inf: capped so the term is 2^-1e8 == 0, not NaN
        hb[r] = 0.5 * nn + ((lw > W_HB_MAX) ? W_HB_MAX : lw);      // (only the weight part: a far row keeps its |b|^2)
    }
    if (far) atomicOr(&wc->far, 1);
}

// 2^x for x <= ~0.  SAFE = false requires |x| < 2^31 (guaranteed when wc->far == 0); SAFE = true clamps first.
template <bool SAFE>
__device__ __forceinline__ double exp2_neg(double x) {
    if (SAFE) x = fmax(x, -1100.0);                       // 2^-1100 == 0 in double
    const double tm = x + 6755399441055744.0;             // 1.5 * 2^52: rounds x to an integer in the low mantissa bits
    const double f = x - (tm - 6755399441055744.0);       // [-1/2, 1/2], exact
    double p = 0x1.61afced541895p-20;                     // scripts/exp2_minimax.py 8
    p = fma(p, f, 0x1.00dad250bededp-16);
    p = fma(p, f, 0x1.430acca32fd76p-13);
    p = fma(p, f, 0x1.5d87483855455p-10);
    p = fma(p, f, 0x1.3b2ab5c529311p-7);
    p = fma(p, f, 0x1.c6b08dd46d38fp-5);
    p = fma(p, f, 0x1.ebfbdff9319e1p-3);
    p = fma(p, f, 0x1.62e42fef8615ep-1);
    p = fma(p, f, 0x1.ffffffffff7a3p-1);
    return ldexp(p, __double2loint(tm));                  // v_ldexp_f64: subnormal / zero results handled
}

// partial denominators: part[slice*kn + i] = sum_{j in slice} exp(a_i.b_j - 1/2|a_i|^2 - hb_j) [* zero-dv mask]
template <int PP>
__global__ __launch_bounds__(256) void k_kde(const double* __restrict__ a /* kn x PP scaled rows */, size_t kn,
                                             const double* __restrict__ b /* Kp x PP scaled rows */, size_t Kp,
                                             const double* __restrict__ hb /* Kp */, const WConst* __restrict__ wc,
                                             const double* __restrict__ theta_raw, size_t K, size_t k0,
                                             const double* __restrict__ prev_raw, double* __restrict__ part) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t slices = gridDim.y, sl = blockIdx.y;
    // wave-uniform bounds in SGPRs (Kp < 2^32, checked by the launcher): the loop test stays off the vector pipe
    const unsigned j0 = __builtin_amdgcn_readfirstlane((unsigned)(Kp * sl / slices));
    const unsigned j1 = __builtin_amdgcn_readfirstlane((unsigned)(Kp * (sl + 1) / slices));
    const bool active = i < kn;
    double ai[PP];
    double ha = 0.0;
#pragma unroll
    for (int p = 0; p < PP; p++) { ai[p] = active ? a[i * PP + p] : 0.0; ha = fma(ai[p], ai[p], ha); }
    ha *= 0.5;
    const int nzero = wc->nzero;
    double acc = 0.0;
    if (nzero == 0 && wc->far == 0) {          // the common case: nothing but the 30-instruction body
        for (unsigned j = j0; j < j1; j++) {
            const double* bj = b + (size_t)j * PP;
            double bc[PP];
#pragma unroll
            for (int p = 0; p < PP; p++) bc[p] = bj[p];
            const double hc = hb[j];
            __builtin_amdgcn_sched_barrier(0);      // all scalar loads of the row are in flight before the first FMA waits
            double e = -(ha + hc);
#pragma unroll
            for (int p = 0; p < PP; p++) e = fma(ai[p], bc[p], e);
            acc += exp2_neg<false>(e);
        }
    } else {
        for (unsigned j = j0; j < j1; j++) {
            const double* bj = b + (size_t)j * PP;
            double e = -(ha + hb[j]);
#pragma unroll
            for (int p = 0; p < PP; p++) e = fma(ai[p], bj[p], e);
            double term = exp2_neg<true>(e);
            // converged parameters: factor 1 if equal (AbcUtil.cpp:573), else 0 (declared)
            for (int z = 0; z < nzero; z++) {
                const int p = wc->zero_idx[z];
                if (active && theta_raw[(k0 + i) + K * (size_t)p] != prev_raw[j + Kp * (size_t)p]) term = 0.0;
            }
            acc += term;
        }
    }
    if (active) part[sl * kn + i] = acc;
}

__device__ __forceinline__ double gaussian_pdf(double x, double sigma) {   // [GSL] gsl_ran_gaussian_pdf
    const double u = x / fabs(sigma);
    return (1.0 / (sqrt(2.0 * M_PI) * fabs(sigma))) * exp(-u * u / 2.0);
}

__device__ __forceinline__ double prior_likelihood(const abc_prior& pr, double v) {
    if (pr.kind == ABC_PRIOR_GAUSS) return gaussian_pdf(v - pr.a, pr.b);
    if (pr.kind == ABC_PRIOR_UNIF_INT)
        return ((v == round(v)) && (pr.a <= v) && (v <= pr.b)) ? 1.0 / (pr.b - pr.a + 1.0) : 0.0;
    return ((pr.a <= v) && (v <= pr.b)) ? 1.0 / (pr.b - pr.a) : 0.0;
}

// w_raw[i] = prod_p likelihood_p(theta_ip) / (C * sum over slices)   (AbcUtil.cpp:557-580)
__global__ __launch_bounds__(256) void k_wfinish(const abc_prior* __restrict__ priors, const double* __restrict__ theta,
                                                 size_t K, int P, size_t k0, size_t kn,
                                                 const double* __restrict__ part, int slices,
                                                 const WConst* __restrict__ wc, double* __restrict__ w_raw) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= kn) return;
    double num = 1.0;
    for (int p = 0; p < P; p++) num *= prior_likelihood(priors[p], theta[(k0 + i) + K * (size_t)p]);
    double den = 0.0;
    for (int s = 0; s < slices; s++) den += part[(size_t)s * kn + i];
    w_raw[i] = num / (wc->C * den);
}

__global__ __launch_bounds__(256) void k_fill(double* __restrict__ w, size_t K, double v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < K) w[i] = v;
}

__global__ __launch_bounds__(256) void k_sumsq_partial(const double* __restrict__ w, size_t K,
                                                       double* __restrict__ part) {
    __shared__ double sm[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < K; i += (size_t)gridDim.x * 256) s = fma(w[i], w[i], s);
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_div_norm(double* __restrict__ w, size_t K, const double* __restrict__ part,
                                                  int nparts) {
    double sq = 0.0;
    for (int b = 0; b < nparts; b++) sq += part[b];
    if (!(sq > 0.0)) return;                     // Eigen normalize(): only if squaredNorm > 0
    const double nrm = sqrt(sq);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < K) w[i] = w[i] / nrm;
}

}  // namespace

int launch_gather_rows(abc_ctx* ctx, const double* Y, size_t n_local, size_t ldy, size_t P, const uint64_t* idx,
                       size_t K, uint64_t idx_base, double* theta, size_t ldt) {
    const size_t tot = K * P;
    if (!tot) return ABC_OK;
    StageTimer tm(ctx, ST_GATHER_DV);
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, Y, n_local, ldy,
                       (int)P, (const unsigned long long*)idx, K, (unsigned long long)idx_base, theta, ldt);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_doubled_variance(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* dv) {
    if (!P) return ABC_OK;
    StageTimer tm(ctx, ST_GATHER_DV);
    if (P > 64 || K < 2) {       // outside the Gram path: direct two-pass kernel, one work-group per parameter
        hipLaunchKernelGGL(k_doubled_variance, dim3((unsigned)P), dim3(256), 0, ctx->stream, theta, K, dv);
        ABC_HIP(ctx, hipGetLastError());
        return ABC_OK;
    }
    double* stats = nullptr;
    ABC_TRY(launch_theta_stats(ctx, theta, K, P, &stats));
    return launch_dv_from_stats(ctx, stats, P, dv);
}

int launch_weights_raw(abc_ctx* ctx, const abc_prior* priors, const double* theta, size_t K, size_t P, size_t k0,
                       size_t kn, const double* theta_prev, size_t Kp, const double* w_prev, const double* dv_prev,
                       double* w_raw) {
    if (P > 64) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "weights: P = %zu > 64 parameters", P);
    if (kn == 0) return ABC_OK;
    if (Kp > 0xffffffffull) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "weights: K' = %zu >= 2^32", Kp);
    if (k0 + kn > K) ABC_FAIL(ctx, ABC_ERR_INVALID, "weights: row range [%zu,%zu) outside K=%zu", k0, k0 + kn, K);
    int PP = 2;
    while (PP < (int)P) PP *= 2;
    const size_t rb = (kn + 255) / 256;
    // Column slices: many short ones.  Work-groups are dispatched slice by slice, so the ~2000 resident groups all
    // stream the same few hundred KB of the previous set through the scalar cache / L2; with 6 slices of 2 MB each
    // (one residency) the same kernel was 25 % slower, waiting on scalar loads (measured: 6 -> 12.0 ms, 16 -> 10.0,
    // 64 -> 9.1, 128 -> 9.05 at K = K' = 1e5).  Bounded by the partial-sum buffer (slices x kn doubles <= 64 MB).
    size_t slices = abc_kde_slices(kn, Kp, PP);
    if (slices > Kp / 64) slices = Kp / 64;
    if (slices < 1) slices = 1;
    if (slices > 1024) slices = 1024;
    WConst* wc = (WConst*)abc_ws_alloc(ctx, sizeof(WConst));
    double* a = (double*)abc_ws_alloc(ctx, kn * PP * sizeof(double));
    double* b = (double*)abc_ws_alloc(ctx, Kp * PP * sizeof(double));
    double* part = (double*)abc_ws_alloc(ctx, slices * kn * sizeof(double));
    double* hb = (double*)abc_ws_alloc(ctx, Kp * sizeof(double));
    if (!wc || !a || !b || !part || !hb) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
    StageTimer tm(ctx, ST_WEIGHTS_MISC);
    hipLaunchKernelGGL(k_wprep, dim3(1), dim3(64), 0, ctx->stream, dv_prev, (int)P, wc);
    hipLaunchKernelGGL(k_wscale, dim3((unsigned)((kn + 255) / 256)), dim3(256), 0, ctx->stream, theta + k0, kn, K,
                       (int)P, PP, wc, theta_prev, Kp, (const double*)nullptr, a, (double*)nullptr);
    hipLaunchKernelGGL(k_wscale, dim3((unsigned)((Kp + 255) / 256)), dim3(256), 0, ctx->stream, theta_prev, Kp, Kp,
                       (int)P, PP, wc, theta_prev, Kp, w_prev, b, hb);
#define LAUNCH_KDE(PPV)                                                                                        \
    hipLaunchKernelGGL(k_kde<PPV>, dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream, a, kn, b, \
                       Kp, hb, wc, theta, K, k0, theta_prev, part)
    {
        StageTimer tk(ctx, ST_KDE);
        switch (PP) {
            case 2: LAUNCH_KDE(2); break;
            case 4: LAUNCH_KDE(4); break;
            case 8: LAUNCH_KDE(8); break;
            case 16: LAUNCH_KDE(16); break;
            case 32: LAUNCH_KDE(32); break;
            default: LAUNCH_KDE(64); break;
        }
    }
#undef LAUNCH_KDE
    hipLaunchKernelGGL(k_wfinish, dim3((unsigned)rb), dim3(256), 0, ctx->stream, priors, theta, K, (int)P, k0, kn, part,
                       (int)slices, wc, w_raw);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_fill(abc_ctx* ctx, double* w, size_t K, double v) {
    if (!K) return ABC_OK;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, w, K, v);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_normalize_l2(abc_ctx* ctx, double* w, size_t K) {
    if (!K) return ABC_OK;
    int nparts = (int)((K + 255) / 256);
    if (nparts > 256) nparts = 256;
    double* part = (double*)abc_ws_alloc(ctx, nparts * sizeof(double));
    if (!part) ABC_FAIL(ctx, ABC_ERR_NOMEM, "normalize: workspace exhausted");
    StageTimer tm(ctx, ST_WEIGHTS_MISC);
    hipLaunchKernelGGL(k_sumsq_partial, dim3(nparts), dim3(256), 0, ctx->stream, w, K, part);
    hipLaunchKernelGGL(k_div_norm, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, w, K, part, nparts);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
