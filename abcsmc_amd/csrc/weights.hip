// Posterior bookkeeping of one SMC set: gather of the K selected parameter rows, doubled variance,
// Gaussian-kernel importance weights and their L2 normalisation.
// Replaces ABC::calculate_doubled_variance (AbcUtil.cpp:528-537, RunningStat.h:16-46),
// ABC::weight_predictive_prior (AbcUtil.cpp:539-586), Parameter::likelihood (Priors.h:54-56, 76-78,
// 102-104) and the Eigen row gathers of AbcSmc.cpp:1045-1060.
//
// The weight kernel is the compute-bound stage of a generation, O(K * K' * P): per pair the P
// per-parameter Gaussian factors of the reference are fused into ONE exponential,
//   prod_p pdf(t_ip - t'_jp; sqrt(dv_p)) = C * exp(-1/2 sum_p ((t_ip - t'_jp)/sigma_p)^2),
// with both parameter sets centred on a robust centre of the previous set (k_wcentre) and pre-scaled by 1/sigma_p, and the squared
// distance expanded as |a|^2 + |b|^2 - 2 a.b so a pair costs P FMAs + one exp:
//   w'_j * exp(-1/2 |a_i - b_j|^2) = exp(a_i.b_j - 1/2|a_i|^2 - (1/2|b_j|^2 - ln w'_j)).
// Both sets are additionally scaled by sqrt(log2 e), so the exponent comes out in base 2 and the exponential is
//   2^x = 2^n * 2^f,  n = round(x) by the 1.5*2^52 addition (its low dword IS n), f = x - n in [-1/2, 1/2],
// 2^f a degree-8 minimax polynomial (|rel err| < 7.8e-13, scripts/exp2_minimax.py; far inside the 1e-6 budget):
// 13 instructions for the exponential, 30 per pair at P = 16, instead of ~40 for the library exp alone.
// Two kernels evaluate the pair sums (include/abcsmc_hip.h: abc_ctx_set_kde_mode):
//   k_kde        fp64 vector kernel: one new particle per lane, previous-set rows wave-uniform through the scalar cache;
//   k_kde_split  (default for 5..32 parameters) the dot products a_i.b_j on the f16 matrix pipe from limb products,
//                the vector pipe left with convert + add + 2^x: see the section "split-operand weight kernel" below,
// plus two fp64 fix-up kernels for the rows the split kernel cannot represent exactly.
#include <stdlib.h>

#include "abc_internal.h"
#include <hip/hip_ext.h>

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
                                                     size_t ldt, const int* __restrict__ sel_fail, int* __restrict__ sel_fail_pin,
                                                     unsigned long long* __restrict__ giveups) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    // (fused drivers) the first kernel behind the selection: the selection's give-up flag goes to the pinned status block
    // now, so the host sees it at its next synchronisation -- before it builds an alias table of placeholder weights --, and the
    // give-up counter of the proposals is snapshotted (a repeated generation restores it)
    if (e == 0) { if (sel_fail) *sel_fail_pin = *sel_fail; if (giveups) giveups[1] = giveups[0]; }
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
constexpr int W_MAXP = 1024;   // parameters the weight stage takes (arrays of WConst; beyond 64 the generic fp64 kernels run)
struct WConst {           // per-parameter constants, built on the device by k_wprep
    double scale[W_MAXP]; // sqrt(log2 e)/sqrt(dv_p), or 0 when dv_p == 0
    double logC;          // unused
    double C;             // prod over dv_p != 0 of 1/(sqrt(2 pi) sqrt(dv_p))
    int nzero;            // number of parameters with dv_p == 0
    int zero_idx[W_MAXP];
    int far;              // set by k_wscale when a scaled coordinate is so large that exponents may leave int32
    int nfar_i, nfar_j;   // rows of the new / previous set outside the range the split-operand kernel is exact on (k_wrows)
    int lim_i;            // more far new rows than this: the fp64 kernel takes the whole call
    double centre[W_MAXP]; // robust column centre of the previous set (k_wcentre): both sets are centred here
    unsigned long long hb_lo, hb_hi;   // KS_TOPN: smallest / largest norm hb of the previous rows that take part, as order-preserving integers (k_whb)
};

constexpr double W_SQRT_LOG2E = 1.2011224087864497825;     // sqrt(log2 e): a.b then comes out in base 2
constexpr double W_COORD_BOUND = 2000.0;                   // |x| <= 2 PP B^2 + 1e8 < 2^31 for PP <= 64
constexpr double W_HB_MAX = 1.0e8;                         // stands for "weight 0": 2^-1e8 == 0

constexpr int KS_MAX_FAR_J = 1024;     // far previous rows the column fix-up takes; more: the fp64 kernel takes the whole call
// the split-operand kernel (+ its far-row fix-ups) handles this call; otherwise the fp64 kernel does
__device__ __forceinline__ bool ks_split_on(const WConst* wc) {
    return !(wc->nzero | wc->far) && wc->nfar_i <= wc->lim_i && wc->nfar_j <= KS_MAX_FAR_J;
}

// (one wave: the square roots and divisions of the parameters side by side, then lane 0 alone takes the product and the list of
// converged parameters in parameter order -- the same operations in the same order as a single thread's loop, which took 34 us)
__global__ __launch_bounds__(64) void k_wprep(const double* __restrict__ dv_prev, int P, WConst* __restrict__ wc, int lim_i) {
    __shared__ double fac[64];
    __shared__ int live[64];
    const int pmax = (P > 64) ? ((P + 63) / 64) * 64 : 64;
    double C = 1.0; int nz = 0;
    for (int p0 = 0; p0 < pmax; p0 += 64) {
        const int p = p0 + (int)threadIdx.x;
        double sc = 0.0, f = 1.0;
        int lv = 0;                                  // 0: converged (dv = 0) or padding
        if (p < P) {
            const double dv = dv_prev[p];
            if (dv != 0.0) { const double sg = sqrt(dv); sc = 1.0 / sg; f = 1.0 / (sqrt(2.0 * M_PI) * sg); lv = 1; }
        }
        wc->scale[p] = sc * W_SQRT_LOG2E;
        fac[threadIdx.x] = f; live[threadIdx.x] = lv;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int q = 0; q < 64 && p0 + q < P; q++) {
                if (live[q]) C *= fac[q]; else wc->zero_idx[nz++] = p0 + q;
            }
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    wc->C = C; wc->nzero = nz; wc->logC = 0.0; wc->far = 0; wc->nfar_i = 0; wc->nfar_j = 0; wc->lim_i = lim_i;
    wc->hb_lo = ~0ull; wc->hb_hi = 0ull;
}

// scaled copies: out[row*PP + p] = (in[row + ld*p] - centre[p]) * scale[p]   (row-major, zero padded to PP);
// centre = k_wcentre's robust column centre of the previous set: differences are unchanged, magnitudes stay O(few sigma).
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
__device__ __forceinline__ void kde_body(unsigned bx, unsigned by, unsigned ny, const double* __restrict__ a /* kn x PP scaled rows */, size_t kn,
                                         const double* __restrict__ b /* Kp x PP scaled rows */, size_t Kp,
                                         const double* __restrict__ hb /* Kp */, const WConst* __restrict__ wc,
                                         const double* __restrict__ theta_raw, size_t K, size_t k0,
                                         const double* __restrict__ prev_raw, double* __restrict__ part,
                                         int fallback_of_split) {
    // launched behind k_kde_split: runs only when that kernel declined (converged parameters, too many far rows)
    if (fallback_of_split && ks_split_on(wc)) return;
    const size_t i = (size_t)bx * 256 + threadIdx.x;
    const size_t slices = ny, sl = by;
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
template <int PP>
__global__ __launch_bounds__(256) void k_kde(const double* __restrict__ a, size_t kn, const double* __restrict__ b, size_t Kp,
                                             const double* __restrict__ hb, const WConst* __restrict__ wc,
                                             const double* __restrict__ theta_raw, size_t K, size_t k0,
                                             const double* __restrict__ prev_raw, double* __restrict__ part, int fallback_of_split) {
    kde_body<PP>(blockIdx.x, blockIdx.y, gridDim.y, a, kn, b, Kp, hb, wc, theta_raw, K, k0, prev_raw, part, fallback_of_split);
}

// ---- optional Epanechnikov kernel (ABC_WEIGHT_EPANECHNIKOV; an extension, see include/abcsmc_hip.h) -------------------------
// part[slice*kn + i] = sum_{j in slice} w'_j max(0, 1 - |a_i - b_j|^2 c),  a, b the scaled rows of k_wscale (scale
// sqrt(log2 e) / sqrt(dv_p), 0 for dv_p = 0) and c = 1 / (log2 e (P' + 4)).  One new particle per lane, previous rows wave-uniform.
template <int PP>
__global__ __launch_bounds__(256) void k_epan(const double* __restrict__ a, size_t kn, const double* __restrict__ b, size_t Kp,
                                              const double* __restrict__ w_prev, const WConst* __restrict__ wc, int P,
                                              double* __restrict__ part) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t slices = gridDim.y, sl = blockIdx.y;
    const unsigned j0 = __builtin_amdgcn_readfirstlane((unsigned)(Kp * sl / slices));
    const unsigned j1 = __builtin_amdgcn_readfirstlane((unsigned)(Kp * (sl + 1) / slices));
    const bool active = i < kn;
    double ai[PP];
#pragma unroll
    for (int p = 0; p < PP; p++) ai[p] = active ? a[i * PP + p] : 0.0;
    const double c = 1.0 / (1.4426950408889634 * (double)(P - wc->nzero + 4));
    double acc = 0.0;
    for (unsigned j = j0; j < j1; j++) {
        const double* bj = b + (size_t)j * PP;
        double d2 = 0.0;
#pragma unroll
        for (int p = 0; p < PP; p++) { const double d = ai[p] - bj[p]; d2 = fma(d, d, d2); }
        const double k = 1.0 - d2 * c;
        acc += (k > 0.0) ? w_prev[j] * k : 0.0;
    }
    if (active) part[sl * kn + i] = acc;
}

// ---- more than 64 parameters: the same sums without register-resident rows (the reference's loops have no size limit) ----------
// one new particle per thread, its scaled row re-read from global memory (L1), the previous row through the scalar cache; the
// guarded exponential and the converged-parameter rule of k_kde's general branch
__global__ __launch_bounds__(256) void k_kde_gen(const double* __restrict__ a, size_t kn, const double* __restrict__ b, size_t Kp,
                                                 const double* __restrict__ hb, const WConst* __restrict__ wc,
                                                 const double* __restrict__ theta_raw, size_t K, size_t k0,
                                                 const double* __restrict__ prev_raw, double* __restrict__ part, int PP, int epan, int P,
                                                 const double* __restrict__ w_prev) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t slices = gridDim.y, sl = blockIdx.y;
    const unsigned j0 = __builtin_amdgcn_readfirstlane((unsigned)(Kp * sl / slices));
    const unsigned j1 = __builtin_amdgcn_readfirstlane((unsigned)(Kp * (sl + 1) / slices));
    if (i >= kn) return;
    const double* ai = a + i * (size_t)PP;
    double ha = 0.0;
    for (int p = 0; p < PP; p++) ha = fma(ai[p], ai[p], ha);
    ha *= 0.5;
    const int nzero = wc->nzero;
    const double c = 1.0 / (1.4426950408889634 * (double)(P - nzero + 4));
    double acc = 0.0;
    for (unsigned j = j0; j < j1; j++) {
        const double* bj = b + (size_t)j * PP;
        if (epan) {
            double d2 = 0.0;
            for (int p = 0; p < PP; p++) { const double d = ai[p] - bj[p]; d2 = fma(d, d, d2); }
            const double k = 1.0 - d2 * c;
            acc += (k > 0.0) ? w_prev[j] * k : 0.0;
        } else {
            double e = -(ha + hb[j]);
            for (int p = 0; p < PP; p++) e = fma(ai[p], bj[p], e);
            double term = exp2_neg<true>(e);
            for (int z = 0; z < nzero; z++) {
                const int p = wc->zero_idx[z];
                if (theta_raw[(k0 + i) + K * (size_t)p] != prev_raw[j + Kp * (size_t)p]) term = 0.0;
            }
            acc += term;
        }
    }
    part[sl * kn + i] = acc;
}

// ---- split-operand weight kernel: the pair dot products on the f16 matrix pipe ------------------------------------
// The fp64 body above spends 17 of its 30 vector instructions per pair on the dot product a_i.b_j.  Here every
// scaled coordinate is written as a sum of three "limbs" that f16 operands carry without loss,
//   v = h0 + h1 + r2,   h0 = rint(128 v)/128 (|v| <= 8: at most 1024 units),  h1 = rint(2^18 (v - h0))/2^18 (at most 1024 units;
//   below 2^-14 an f16 SUBNORMAL, which v_mfma_f32_32x32x16_f16 takes at its value: scripts/mfma_f16_probe.hip),
//   r2 = v - h0 - h1, |r2| <= 2^-19, entering as (h0 2^-11).(r2' 2^11) -- both factors f16-representable, the product unscaled,
// and the dot product as the sum of the limb products, which the MFMA evaluates with the parameter index as its K
// dimension (P <= 16: one chunk; P <= 32: two; P <= 64: four), 32 previous x 32 new particles per instruction.  One f32 accumulator, in this order:
//   X = h0.h0' - hbTop_j          every term a multiple of 2^-14, every partial sum below 2^10: EXACT in f32 whatever the
//                                 order of accumulation (bound: KS_NORM2 below; probe: 0 inexact sums of 102400)
//   - n                           n = floor(max X) of the 16 values a lane owns: still exact, and now small for the terms that matter
//   + Y = h0.h1' + h1.h0' + h1.h1' + h0.r2' + r2.h0' - hbLow_j      small terms into a small accumulator: roundings ~1e-8 |Z|
// where hb_j = 1/2|b_j|^2 - log2 w'_j (= Top, a multiple of 2^-14, + Low) enters through (bf16) K-steps against constant -1
// operands.  6 + 3 MFMAs per 16 parameters and 1024 pairs; round 1 / the first half of round 2 used four bf16 limbs, 13 + 2
// (bf16 carries 8 significant bits: 15 -> 10 MFMAs alone took that kernel from 3.5 to 2.5 ms, diagnostic build).  1/2|a_i|^2 is the same in every term of row i: it stays out of the sums (integer part: subtracted from every batch's
// power of two; fraction: k_wfinish, fp64), which also halves the range X has to be exact on.  Left out: h1.r2' + r2.h1'
// (1.4e-8 rms / 7e-8 max on the exponent at 16 parameters, 2e-8 / 9e-8 at 32: scripts/split_precision.py) and r2.r2'.
// Z is the base-2 exponent of the term up to the row's factor and the batch's 2^n, and the vector pipe only exponentiates and adds:
//   terms of a batch = 2^n * sum of 2^Z      3 issue slots per pair instead of 30 (kz_slots: v_exp_f32, f32 add; one fp64 scaling
//   and add per 16 pairs).
// Error of a batch sum (16 terms) with the f32 evaluation: 5e-8 rms, 2e-7 max (+ one ulp of v_exp_f32).  Measured error of a
// weight against the oracle: tests/test_gpu_parity.py::test_weight_split_kernel_accuracy_and_zero_weights (bound 2.5e-7 up to 16 parameters), budget 1e-6.
// Rows outside the exact range (a |coordinate| > 8, |row|^2 > 400, a weight outside {0} U [2^-300, 2^100]) are "far": k_wrows
// gives them all-zero limbs (a far previous row then contributes exactly 0 here), flags / lists them, and two fp64 fix-up
// kernels add their pairs (k_kde_fixups: a far new particle against the whole previous set; the far previous
// particles against every new one), k_wfinish picking per row.  With converged parameters, coordinates beyond the int32
// exponent range, or too many far rows (> K/16 + 32 new, > 1024 previous) the fp64 kernel above takes the whole call: every
// kernel is always launched and those whose turn it is not return at once, so no flag travels to the host.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int KS_NL = 4;                       // f16 operands per 16-parameter chunk and side
constexpr int KS_NP = 6;                       // limb products per chunk
// Exact range of the split kernel (rows outside are "far": fp64 fix-ups).  X = sum h0.h0' - hbTop counts in units of 2^-14 and
// must stay below 2^10 in every partial sum: with |a|^2, |b|^2 <= KS_NORM2 the products of any number of chunks sum to at most
// |a||b| <= 400 in absolute value (Cauchy-Schwarz), hbTop = 1/2|b|^2 - log2 w' lies in [-100, 500]: at most 900.
constexpr double KS_BOUND = 8.0;               // |scaled coordinate| (h0 <= 1024 units of 2^-7: one f16)
constexpr double KS_NORM2 = 400.0;             // |scaled row|^2  (typical: 0.7 x parameters)
constexpr double KS_LW_CAP = 300.0, KS_LW_MIN = -100.0;   // -log2 w' of a non-zero weight (w' = 0: an all-zero row with hb = KS_HB_ZERO)
constexpr double KS_HB_ZERO = 1100.0;          // 2^-1100 == 0 in double
constexpr double KS_U0 = 128.0;                // h0 = rint(v U0) / U0
constexpr double KS_UH = KS_U0 * 2048.0;       // h1 = rint((v - h0) UH) / UH
constexpr double KS_XUNIT_INV = KS_U0 * KS_U0; // X counts in units of 2^-14

// Centre of both sets -> wc->centre (one work-group per parameter): the first previous particle plus the mean offset
// of the previous set from it, each offset clipped at +-16 proposal sigmas.  Without outliers this is the column
// mean; a particle 1e7 sigmas away (tests) moves it by at most 16/K' sigmas, so the bulk keeps small coordinates --
// which the expanded distance |a|^2 + |b|^2 - 2 a.b of both kernels needs (cancellation), and the split kernel's range.
// two stages, fixed summation order (bit-reproducible): WC_NB row blocks per parameter, then one thread per parameter
constexpr int WC_NB = 64;
__global__ __launch_bounds__(256) void k_wcentre(const double* __restrict__ prev, size_t Kp, const WConst* __restrict__ wc,
                                                 double* __restrict__ partial /* P x WC_NB */) {
    __shared__ double sm[4];
    const double* col = prev + Kp * (size_t)blockIdx.x;
    const double c0 = col[0], sc = wc->scale[blockIdx.x];
    const size_t i0 = Kp * blockIdx.y / WC_NB, i1 = Kp * (blockIdx.y + 1) / WC_NB;
    double s = 0.0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const double d = (col[i] - c0) * sc;
        s += fmin(fmax(d, -16.0), 16.0);             // NaN -> -16: harmless, the NaN row poisons the weights anyway
    }
    s = block_sum_256(s, sm);
    if (threadIdx.x == 0) partial[blockIdx.x * WC_NB + blockIdx.y] = s;
}
__global__ void k_wcentre_finish(const double* __restrict__ prev, size_t Kp, int P, const double* __restrict__ partial,
                                 WConst* __restrict__ wc) {
    const int p = threadIdx.x;
    if (p >= P) return;
    double s = 0.0;
    for (int b = 0; b < WC_NB; b++) s += partial[p * WC_NB + b];
    const double c0 = prev[Kp * (size_t)p], sc = wc->scale[p];
    wc->centre[p] = (sc != 0.0) ? c0 + (s / (double)Kp) / sc : c0;
}

__device__ __forceinline__ unsigned bf16_bits(double v, double* back) {      // round to nearest even, value returned too
    unsigned u = __float_as_uint((float)v);
    u += 0x7fffu + ((u >> 16) & 1u);
    u &= 0xffff0000u;
    *back = (double)__uint_as_float(u);
    return u >> 16;
}
// IEEE binary16 bits of x (|x| < 2^15), round to nearest even, subnormals kept (v_mfma_f32_32x32x16_f16 takes them at
// their value: scripts/mfma_f16_probe.hip); by integer arithmetic, so the result does not depend on conversion modes
__device__ __forceinline__ unsigned f16_bits(double x) {
    const unsigned sg = (x < 0.0) ? 0x8000u : 0u;
    const double ax = fabs(x);
    if (ax == 0.0) return sg;
    int e = ilogb(ax);
    if (e < -14) return sg | (unsigned)rint(ldexp(ax, 24));            // subnormal grid 2^-24 (1024 = the smallest normal)
    unsigned m = (unsigned)rint(ldexp(ax, 10 - e));                     // [1024, 2048]
    if (m == 2048u) { m = 1024u; e++; }
    return sg | ((unsigned)(e + 15) << 10) | (m - 1024u);
}
// value of IEEE binary16 bits (finite)
__device__ __forceinline__ double f16_value(unsigned b) {
    const int e = (int)((b >> 10) & 31u), m = (int)(b & 0x3ffu);
    const double v = e ? ldexp((double)(1024 + m), e - 25) : ldexp((double)m, -24);
    return (b & 0x8000u) ? -v : v;
}
// KS_FOLD: h = top + low as f16 pieces for the spare K-slots of the limb operands -- top (a multiple of 2^-14 below 2^11: 11 + 11 + 3
// bits, exact) and low scaled by 2^24 (|low| <= 2^-15: three pieces carry it to 2^-48)
__device__ __forceinline__ void ks_pieces_f16(double h, double unit_inv, unsigned top_pc[3], unsigned low_pc[3]) {
    const double top = rint(h * unit_inv) / unit_inv;
    double rem = top;
    for (int k = 0; k < 3; k++) { top_pc[k] = f16_bits(rem); rem -= f16_value(top_pc[k]); }
    rem = (h - top) * 0x1p24;
    for (int k = 0; k < 3; k++) { low_pc[k] = f16_bits(rem); rem -= f16_value(low_pc[k]); }
}
// h = top (a multiple of 1/unit_inv = 2^-14, <= 24 significant bits: three bf16 pieces hold it exactly) + low (three more pieces)
__device__ __forceinline__ void ks_pieces(double h, double unit_inv, unsigned pc[6]) {
    const double top = rint(h * unit_inv) / unit_inv;
    double back, rem = top;
    for (int k = 0; k < 3; k++) { pc[k] = bf16_bits(rem, &back); rem -= back; }
    rem = h - top;
    for (int k = 3; k < 6; k++) { pc[k] = bf16_bits(rem, &back); rem -= back; }
}
constexpr unsigned KS_MONE = 0xBF80u;                       // bf16 -1

// Limb tiles of one set (written by k_wrows below).
// Output, per tile of 32 rows: `ops` operands of 1 KiB in MFMA fragment order [half h][row r][8 x 16 bit] (lane 32h + r
// reads its 16 bytes at 16*(32h + r)): operand (c*KS_NL + k) = f16 operand k of parameters 16c..16c+15
//   0: h0    1: h1    2: h0 2^-11    3: r2 2^11        (v = h0 + h1 + r2; operands 2 and 3 meet in h0.r2' = (h0 2^-11).(r2' 2^11):
//                                                        r2 itself, below 2^-19, would fall on f16's subnormal grid)
// and, for the previous set (w != NULL), ONE more operand (bf16) for the norm step: K-slots 0..2 hbTop pieces, 3..5 hbLow pieces,
// hb = 1/2|b|^2 - log2 w' = Top (a multiple of 2^-14) + Low; the kernel multiplies it with constant -1 patterns.
// The new set's 1/2|a_i|^2 is common to every term of row i and stays OUT of the sums: its integer part goes to ha_int
// (subtracted from the exponent of every batch's power of two), its fraction to ha_frac (k_wfinish, fp64).
// Rows >= rows are padding: zero limbs; a padded previous row has hb = KS_HB_ZERO (a term of exactly 0).

// f16 bits of n 2^-s for an integer |n| <= 1024 whose value lies on f16's grid (the exact limbs h0 = n0 2^-7, h1 = n1 2^-18,
// h0 2^-11 = n0 2^-18): integer arithmetic only -- f16_bits does the same through ilogb / ldexp / rint in fp64
__device__ __forceinline__ unsigned f16_of_int(int n, int s) {
    const unsigned sg = (n < 0) ? 0x8000u : 0u;
    const unsigned a = (unsigned)(n < 0 ? -n : n);
    if (a == 0u) return sg;
    const int e = 31 - __clz((int)a), E = e - s;
    if (E >= -14) return sg | ((unsigned)(E + 15) << 10) | ((a << (10 - e)) & 0x3ffu);
    return sg | (a << (24 - s));                                   // subnormal: units of 2^-24
}

// Scaled copy and limb tiles in ONE pass over a set (round 3: as two kernels, k_wscale + a per-row split kernel, they were 37 us of the new set's prologue, in front of the
// pair sums, one thread per row with ~2000 dependent instructions each): a row is dealt out to 2 NCH lanes of one wave, eight
// parameters each -- lane g RW + rr handles parameters 8g .. 8g + 7 of row rr of the wave's RW = 64 / (2 NCH) rows -- which is
// also the tile layout: the lane's four 16-byte limb operands go straight to [half][row][8 x 16 bit].  Row quantities (norm,
// far flags) are combined across the row's lanes in fixed order.  Same outputs as the two kernels: the scaled fp64 row-major
// copy (+ hb for the previous set), the limb tiles, far flags / list, ha_int / ha_frac.
// (NST < NCH -- 33..48 parameters, round 6: the lanes are dealt out as for four chunks, three are stored; the fourth chunk's
// parameters do not exist, its limbs would be zeros that the pair sums' matrix steps multiply for nothing)
template <int NCH, int NST = NCH>
__global__ __launch_bounds__(256) void k_wrows(const double* __restrict__ in, size_t rows, size_t ld, int P, int PP, size_t rows_pad,
                                               WConst* __restrict__ wc, const double* __restrict__ w, int is_prev,
                                               double* __restrict__ out, double* __restrict__ hb,
                                               unsigned short* __restrict__ tiles, int ops, unsigned char* __restrict__ far_flag,
                                               unsigned* __restrict__ far_list, int* __restrict__ ha_int, double* __restrict__ ha_frac,
                                               int fold /* KS_FOLD: norm pieces in K-slots 13..15 of the limb operands, no norm operand */,
                                               const unsigned* __restrict__ rank /* KS_TOPN: the tile position of every row (its rank by norm top), or NULL */,
                                               float* __restrict__ topf /* ... and the tops as f32, by tile position */) {
    constexpr int G = 2 * NCH, RW = 64 / G;
    const int lane = threadIdx.x & 63, g = lane / RW, rr = lane % RW;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave * RW >= rows_pad) return;                              // wave-uniform
    const size_t r = wave * RW + rr;
    const bool inrange = r < rows;
    double v[8], nn = 0.0;
    bool far2k = false, far8 = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int p = 8 * g + j;
        v[j] = (inrange && p < P) ? (in[r + ld * (size_t)p] - wc->centre[p]) * wc->scale[p] : 0.0;
        nn = fma(v[j], v[j], nn);
        far2k = far2k || (fabs(v[j]) > W_COORD_BOUND);
        far8 = far8 || !(fabs(v[j]) <= KS_BOUND);
    }
    if (inrange) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const int p = 8 * g + j;
            if (p < PP) *reinterpret_cast<double2*>(out + r * (size_t)PP + p) = make_double2(v[j], v[j + 1]);
        }
    }
    // the row's norm: its lanes' partial sums in ascending parameter order (every lane of the row gets the same value)
    double nrow = __shfl(nn, rr, 64);
#pragma unroll
    for (int gg = 1; gg < G; gg++) nrow += __shfl(nn, gg * RW + rr, 64);
    bool rfar2k = far2k, rfar8 = far8;
#pragma unroll
    for (int o = RW; o < 64; o <<= 1) {
        // (the shuffles unconditionally, by every lane: behind `flag || shuffle` the lanes whose flag is set skip the exchange, and a
        // lane that skips it delivers nothing -- the row's other lanes then never learn of a far coordinate in this lane's eight)
        const int o2k = __shfl_xor((int)rfar2k, o, 64), o8 = __shfl_xor((int)rfar8, o, 64);
        rfar2k = rfar2k | (o2k != 0);
        rfar8 = rfar8 | (o8 != 0);
    }
    bool valid = inrange;
    double lw = 0.0;
    bool far = false;
    if (is_prev) {
        const double wr = inrange ? w[r] : 0.0;
        const double lwr = -log2(wr);                                // w = 0 -> +inf
        if (inrange && g == 0) hb[r] = 0.5 * nrow + ((lwr > W_HB_MAX) ? W_HB_MAX : lwr);     // fp64 kernels: capped, not NaN
        if (wr == 0.0) valid = false;                                // weight 0: contributes exactly nothing, like padding
        else {
            lw = lwr;
            if (!(lw >= KS_LW_MIN && lw <= KS_LW_CAP)) far = true;   // w' > 2^100, < 2^-300, negative or NaN
        }
    }
    if (rfar2k && inrange && g == 0) atomicOr(&wc->far, 1);
    if (valid) far = far || rfar8 || !(nrow <= KS_NORM2);
    // a far row takes no part in the matrix work (all-zero limbs, and hb = KS_HB_ZERO on the previous side): the fp64
    // fix-up launch (k_kde_fixups) adds its pairs
    if (far) {
        if (g == 0) {
            if (is_prev) { const int pos = atomicAdd(&wc->nfar_j, 1); if (pos < KS_MAX_FAR_J) far_list[pos] = (unsigned)r; }
            else atomicAdd(&wc->nfar_i, 1);
        }
        valid = false;
    }
    if (!is_prev && g == 0) far_flag[r] = far ? 1 : 0;
    const size_t spos = (rank && inrange) ? (size_t)rank[r] : r;      // where the row sits in the tiles (padding rows: behind all others either way)
    unsigned short* tb = tiles + (spos >> 5) * (size_t)ops * 512;
    const unsigned r32 = (unsigned)(spos & 31);
    const int c = g >> 1, h = g & 1;
    unsigned pk[KS_NL][4];
    unsigned top_pc[3] = {0u, 0u, 0u}, low_pc[3] = {0u, 0u, 0u};
    const bool fold_lane = fold && g == 2 * NST - 1;                  // the lane that holds K-slots 8..15 of its row's LAST (stored) chunk
    if (fold_lane && is_prev) ks_pieces_f16(valid ? 0.5 * nrow + lw : KS_HB_ZERO, KS_XUNIT_INV, top_pc, low_pc);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const double x = valid ? v[j] : 0.0;
        const int n0 = (int)rint(x * KS_U0);                         // h0 = n0 2^-7, |n0| <= 1024
        const double r1 = x - (double)n0 * (1.0 / KS_U0);            // |r1| <= 2^-8, exact
        const int n1 = (int)rint(r1 * KS_UH);                        // h1 = n1 2^-18, |n1| <= 1024
        const double r2 = r1 - (double)n1 * (1.0 / KS_UH);           // |r2| <= 2^-19, exact
        unsigned b[KS_NL];
        b[0] = f16_of_int(n0, 7);
        b[1] = f16_of_int(n1, 18);
        b[2] = f16_of_int(n0, 18);                                   // h0 2^-11
        b[3] = f16_bits(r2 * 0x1p11);                                // |.| <= 2^-8, rounded to eleven significant bits
        if (fold_lane && j >= 5) {                                   // K-slots 13..15 of the last chunk (its parameters 13..15 do not exist)
            if (is_prev) { b[0] = top_pc[j - 5]; b[2] = low_pc[j - 5]; }      // against -1 in the new set's h0 operand, -2^-24 in its r2 operand
            else { b[0] = 0xBC00u; b[3] = 0x8001u; }
        }
#pragma unroll
        for (int k = 0; k < KS_NL; k++) {
            if (j & 1) pk[k][j >> 1] |= b[k] << 16; else pk[k][j >> 1] = b[k];
        }
    }
#pragma unroll
    for (int k = 0; k < KS_NL; k++)
        if (NST == NCH || c < NST)
            *(uint4*)(tb + (size_t)(c * KS_NL + k) * 512 + (h * 32 + r32) * 8) = make_uint4(pk[k][0], pk[k][1], pk[k][2], pk[k][3]);
    if (g == 0) {
        if (is_prev && !fold) {
            unsigned pc[6];
            unsigned short* ob = tb + (size_t)(NST * KS_NL) * 512;
            const double hbv = valid ? 0.5 * nrow + lw : KS_HB_ZERO;
            ks_pieces(hbv, KS_XUNIT_INV, pc);
            if (topf) topf[spos] = (float)(rint(hbv * KS_XUNIT_INV) / KS_XUNIT_INV);     // (the top ks_pieces splits off)
            // K-slots 6,7 = -1 for the rows an MFMA result holds in its lanes 0..31 (bit 2 of the row clear), 14,15 = -1 for the others:
            // against a B operand whose slots 6,7 / 14,15 carry the pieces of n -- the LAST four bytes of every lane's sixteen, so the
            // kernel builds that operand without a lane select --, the lane's own batch reference is subtracted from exactly its rows
            const unsigned ones = KS_MONE | (KS_MONE << 16);            // (-1: the kernel hands over +n, not -n: one negation less per batch)
            *(uint4*)(ob + r32 * 8) = make_uint4(pc[0] | (pc[1] << 16), pc[2] | (pc[3] << 16), pc[4] | (pc[5] << 16), (r32 & 4) ? 0u : ones);
            *(uint4*)(ob + (32 + r32) * 8) = make_uint4(0u, 0u, 0u, (r32 & 4) ? ones : 0u);
        } else if (!is_prev) {
            const double ha = valid ? 0.5 * nrow : 0.0, hi = floor(ha);          // 0 for a far / padded row
            ha_int[r] = (int)hi;
            ha_frac[r] = ha - hi;
        }
    }
}

// ---- far rows: pairs the split-operand kernel leaves out, in fp64 -----------------------------------------------------
// the base-2 exponent of one pair from the scaled fp64 rows, as k_kde's guarded loop computes it
template <int PP>
__device__ __forceinline__ double ks_pair_term(const double (&ai)[PP], double ha, const double* __restrict__ bj, double hbj) {
    double e = -(ha + hbj);
#pragma unroll
    for (int p = 0; p < PP; p++) e = fma(ai[p], bj[p], e);
    return exp2_neg<true>(e);
}

// The pairs of far rows, ONE launch (it returns at once when there are none): work-groups [0, nat) take the far NEW particles
// -- their whole sum over the previous set, one work-group per tile of 32 new rows, tiles without a far row return --,
// work-groups [nat, nat + rb) the far PREVIOUS particles: their terms for every new particle (one per lane), in ascending row
// order (the atomically appended list is sorted in LDS by every work-group: a fixed order makes the sums bit-reproducible).
template <int PP>
__device__ __forceinline__ void far_body(unsigned bx, const double* __restrict__ a, size_t kn, const double* __restrict__ b,
                                         size_t Kp, const double* __restrict__ hb, const WConst* __restrict__ wc,
                                         const unsigned char* __restrict__ far_flag, const unsigned* __restrict__ list,
                                         unsigned nat, double* __restrict__ fix_i, double* __restrict__ fix_j) {
    if (!ks_split_on(wc)) return;
    __shared__ double sm[4];
    __shared__ unsigned sl[KS_MAX_FAR_J];
    if (bx < nat) {
        if (wc->nfar_i == 0) return;
        const size_t i0 = (size_t)bx * 32;
        for (int r = 0; r < 32; r++) {
            const size_t i = i0 + r;
            if (i >= kn || !far_flag[i]) continue;               // work-group uniform
            double ai[PP], ha = 0.0;
#pragma unroll
            for (int p = 0; p < PP; p++) { ai[p] = a[i * PP + p]; ha = fma(ai[p], ai[p], ha); }
            ha *= 0.5;
            double s = 0.0;
            for (size_t j = threadIdx.x; j < Kp; j += 256) s += ks_pair_term<PP>(ai, ha, b + j * PP, hb[j]);
            s = block_sum_256(s, sm);
            if (threadIdx.x == 0) fix_i[i] = s;
            __syncthreads();
        }
        return;
    }
    const int n = wc->nfar_j;
    if (n == 0) return;
    const unsigned t = threadIdx.x;
    for (unsigned e = t; e < KS_MAX_FAR_J; e += 256) sl[e] = ((int)e < n) ? list[e] : 0xffffffffu;
    for (unsigned k = 2; k <= KS_MAX_FAR_J; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (unsigned p = t; p < KS_MAX_FAR_J / 2; p += 256) {
                const unsigned i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const unsigned x = sl[i], y = sl[i + j];
                if ((x > y) == ((i & k) == 0)) { sl[i] = y; sl[i + j] = x; }
            }
        }
    __syncthreads();
    const size_t i = (size_t)(bx - nat) * 256 + t;
    if (i >= kn) return;
    double ai[PP], ha = 0.0;
#pragma unroll
    for (int p = 0; p < PP; p++) { ai[p] = a[i * PP + p]; ha = fma(ai[p], ai[p], ha); }
    ha *= 0.5;
    double s = 0.0;
    for (int q = 0; q < n; q++) {
        const size_t j = sl[q];
        s += ks_pair_term<PP>(ai, ha, b + j * PP, hb[j]);
    }
    fix_j[i] = s;
}
// ONE launch behind k_kde_split for everything that kernel may have left undone (round 2: two, 10 us of empty launches on the
// critical path of every weighted generation): work-groups [0, nat + rb) the far rows' fix-ups, the rest the fp64 kernel's grid
// for the case that the split kernel declined the whole call.  Whose turn it is not returns at once.
template <int PP>
__global__ __launch_bounds__(256) void k_kde_fixups(const double* __restrict__ a, size_t kn, const double* __restrict__ b, size_t Kp,
                                                    const double* __restrict__ hb, const WConst* __restrict__ wc,
                                                    const unsigned char* __restrict__ far_flag, const unsigned* __restrict__ list,
                                                    unsigned nat, unsigned rb, unsigned slices, double* __restrict__ fix_i,
                                                    double* __restrict__ fix_j, const double* __restrict__ theta_raw, size_t K, size_t k0,
                                                    const double* __restrict__ prev_raw, double* __restrict__ part) {
    if (blockIdx.x < nat + rb) { far_body<PP>(blockIdx.x, a, kn, b, Kp, hb, wc, far_flag, list, nat, fix_i, fix_j); return; }
    const unsigned bid = blockIdx.x - (nat + rb);
    kde_body<PP>(bid % rb, bid / rb, slices, a, kn, b, Kp, hb, wc, theta_raw, K, k0, prev_raw, part, 1);
}

// The limb products of one 16-parameter chunk: which operand of the previous (A) and of the new (B) particle.  Product 0 is the
// exact h0.h0' (multiples of 2^-14); then h0.h1', h1.h0', h1.h1', h0.r2', r2.h0'.  Left out: h1.r2' + r2.h1' (each factor pair
// below 2^-8 x 2^-19: 1.4e-8 rms on the exponent at 16 parameters, scripts/split_precision.py) and r2.r2'.
constexpr signed char KS_LA[KS_NP] = {0, 0, 1, 1, 2, 3};
constexpr signed char KS_LB[KS_NP] = {0, 1, 0, 1, 3, 2};
__device__ __forceinline__ float ks_max3(float a, float b, float c) {
    float r;
    asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));   // (fmaxf would canonicalise its inputs first)
    return r;
}
struct KsRef { float p0, p1, p2, p3; };        // the four running f32 sums of a batch
// ---- one accumulator per batch ----------------------------------------------------------------------------------------------
// Z = X - n + Y in ONE f32 accumulator.  The chain starts with the exact part X (norm top, h0.h0'); the vector pipe takes
// n = floor(max X) over the 16 values a lane owns of the 32 x 32 block (8 v_max3_f32) and hands n back to the matrix pipe as
// the B operand of one more bf16 step -- its two bf16 pieces in K-slots 6,7 of lanes 0..31 and 14,15 of lanes 32..63, against the
// -1 entries k_wrows put in the previous set's norm operand, so each lane's -n lands on exactly the rows that lane holds -- which is
// still exact (multiples of 2^-14 below 2^11); then -hbLow and the five small products follow into the SAME accumulator, which
// by then is small (< 1.3 for the terms that carry the sum), so their roundings are 2^-25 |Z| each.  The vector pipe is left with
// v_exp_f32 (measured on gfx950 over every float of [-0.3, 1.3], scripts/exp2_hw_accuracy.hip: max 8.2e-8, rms 2.6e-8 relative)
// and the running sums (four f32 chains of four, a tree, ONE convert / v_ldexp_f64 / fp64 add per batch; terms more than 2^126
// below their batch's largest flush to zero): 3 issue slots per pair + 1.1 of batch overhead (66 per 16 pairs).  History of the
// vector side: 14.5 fp64 instructions per pair (round 1) -> 8 slots (floor / fract split of an exact X, per-pair ldexp) -> 5.75
// (one power of two per batch, two accumulators: subtract, add, exp, add) -> 4.1, for one MFMA more (9 per 1024 pairs at 16
// parameters).  Error of a batch sum (emulation, scripts/split_precision.py): rms 4.8e-8 / max 2.0e-7 (4.6e-8 / 2.0e-7 with two
// accumulators).  Measured at 1e10 pairs, P = 16: 2.38 -> 2.24 ms -- the chip clocks 6 % lower under the ninth MFMA (1.78 against
// 1.90 GHz, PMC), which eats most of what the 26 % shorter instruction stream buys.
// The kernel's variants, by their first template argument: 1, 2, 4 = 16-parameter chunks; KS_FOLD + 1, 2, 4 = as many chunks
// holding at most 13 / 29 / 61 parameters with the norm pieces in the three spare K-slots of the LAST chunk's limb operands -- hbTop (three f16 pieces against -1) in the
// exact h0.h0' step, hbLow (scaled by 2^24, three f16 pieces against -2^-24: f16 has no exponent for it otherwise) in the
// (h0 2^-11).(r2' 2^11) step, which only ever meets its own partner operand: 7 / 13 / 25 MFMAs per 1024 pairs instead of 9 / 15 / 27 (the -n
// step stays an instruction of its own: scripts/mfma_merge_probe.hip), and no norm operand to stream (the -1 entries the -n step
// needs are a per-lane constant).  The matrix pipe's energy is what bounds this kernel.
constexpr int KS_FOLD = 100;               // variant tag KS_FOLD + chunks: 101, 102, 104
// KS_TOPN + 1, 2, 4 = full chunks (14..16, 30..32, 62..64 parameters: no spare K-slot) with ONE MFMA less than the plain variants:
// the norm top is not subtracted by a step of its own in front of the batch reference but TOGETHER with the reference, in one
// bf16 step (-1 against the top pieces, the pieces of n against the -1 entries): X' - top - n is exact for every result within
// 2^-128 of the batch's largest (scripts/mfma_topn_probe.hip: 0 of 2 048 000 such results inexact), the others flush to zero
// anyway.  The reference has to come from X' = h0.h0' alone then: n = floor(max X' - tmin) with tmin the smallest top of the
// tile, which is only close to the largest X' - top if the tile's tops are close to one another -- so the previous set's tiles
// are filled in the ORDER OF THE ROWS' TOPS (launch_weights_prev: k_whb, a stable sort, k_wrows writing every row to its rank;
// pair sums do not care about the order of their terms): the spread inside a tile is the set's range / its tiles.  Taking the
// maximum of X' - top row by row instead (16 subtractions per lane and tile) cost more than the MFMA saved (+5.7 % at 16
// parameters); this way the vector pipe gets one subtraction per batch.
constexpr int KS_TOPN = 200;
template <int V> __host__ __device__ constexpr int kz_nch() { return V > KS_TOPN ? V - KS_TOPN : V > KS_FOLD ? V - KS_FOLD : V; }
template <int V> __host__ __device__ constexpr bool kz_fold() { return V > KS_FOLD && V < KS_TOPN; }
template <int V> __host__ __device__ constexpr bool kz_topn() { return V > KS_TOPN; }
template <int V> __host__ __device__ constexpr int kz_nexact() { return V > KS_FOLD ? kz_nch<V>() : 1 + V; }
template <int NCH>
__host__ __device__ constexpr int kz_nsteps() { return kz_fold<NCH>() ? 1 + 6 * kz_nch<NCH>() : kz_topn<NCH>() ? 2 + 6 * kz_nch<NCH>() : 3 + 6 * NCH; }
// step S of a batch: 0 norm top; 1..NCH h0.h0'; [vector: n]; NCH+1: -n; NCH+2: norm low; then 5 products per chunk
// (folded: 0..NCH-1 h0.h0', the last chunk's with the norm top; [vector: n]; NCH: -n; then 5 products per chunk, the last chunk's
// (h0 2^-11).(r2' 2^11) with the norm low)
// (AT: the previous tile's operands -- a register array (uint4*) or, in k_kde_split_lds, KsLdsOps: the same indices into a tile staged in LDS)
template <int NCH, int S, typename AT>
__device__ __forceinline__ void kz_mfma(const AT A, const uint4* B, const uint4 (&NB)[2], const uint4& BN, f32x16& Z) {
    const f32x16 Z0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (kz_fold<NCH>()) {
        constexpr int C = kz_nch<NCH>();
        if constexpr (S == 0) {
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[0]), __builtin_bit_cast(f16x8, B[0]), Z0, 0, 0, 0);
        } else if constexpr (S < C) {
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[S * KS_NL]), __builtin_bit_cast(f16x8, B[S * KS_NL]), Z, 0, 0, 0);
        } else if constexpr (S == C) {
            Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, NB[0]), __builtin_bit_cast(bf16x8, BN), Z, 0, 0, 0);
        } else {
            constexpr int q = S - (C + 1), c = q / 5, l = 1 + q % 5;             // products 1..5 of KS_LA / KS_LB
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c * KS_NL + KS_LA[l]]),
                                                       __builtin_bit_cast(f16x8, B[c * KS_NL + KS_LB[l]]), Z, 0, 0, 0);
        }
    } else if constexpr (kz_topn<NCH>()) {
        constexpr int C = kz_nch<NCH>();
        if constexpr (S == 0) {
            // a WIDE tile (its tops spread over more than KS_TOPN_SPREAD: the sparse tail of the set) takes the plain variant's
            // route -- the norm top in a step of its own, in front of everything -- so that its reference is the maximum of X' - top
            // itself; the tile's info word says so (mask 0), and the reference step below then leaves the top out of its operand
            if (__builtin_amdgcn_readfirstlane(A[C * KS_NL + 1].y) == 0u) {
                Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[C * KS_NL]), __builtin_bit_cast(bf16x8, NB[0]), Z0, 0, 0, 0);
                Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[0]), __builtin_bit_cast(f16x8, B[0]), Z, 0, 0, 0);
            } else {
                Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[0]), __builtin_bit_cast(f16x8, B[0]), Z0, 0, 0, 0);
            }
        } else if constexpr (S < C) {
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[S * KS_NL]), __builtin_bit_cast(f16x8, B[S * KS_NL]), Z, 0, 0, 0);
        } else if constexpr (S == C) {          // - top - n
            Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[C * KS_NL]), __builtin_bit_cast(bf16x8, BN), Z, 0, 0, 0);
        } else if constexpr (S == C + 1) {      // - low
            Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[C * KS_NL]), __builtin_bit_cast(bf16x8, NB[1]), Z, 0, 0, 0);
        } else {
            constexpr int q = S - (C + 2), c = q / 5, l = 1 + q % 5;
            Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c * KS_NL + KS_LA[l]]),
                                                       __builtin_bit_cast(f16x8, B[c * KS_NL + KS_LB[l]]), Z, 0, 0, 0);
        }
    } else if constexpr (S == 0) {
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[NCH * KS_NL]), __builtin_bit_cast(bf16x8, NB[0]), Z0, 0, 0, 0);
    } else if constexpr (S <= NCH) {
        constexpr int c = S - 1;
        Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c * KS_NL]), __builtin_bit_cast(f16x8, B[c * KS_NL]), Z, 0, 0, 0);
    } else if constexpr (S == NCH + 1) {
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[NCH * KS_NL]), __builtin_bit_cast(bf16x8, BN), Z, 0, 0, 0);
    } else if constexpr (S == NCH + 2) {
        Z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[NCH * KS_NL]), __builtin_bit_cast(bf16x8, NB[1]), Z, 0, 0, 0);
    } else {
        constexpr int q = S - (NCH + 3), c = q / 5, l = 1 + q % 5;          // products 1..5 of KS_LA / KS_LB
        Z = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A[c * KS_NL + KS_LA[l]]),
                                                   __builtin_bit_cast(f16x8, B[c * KS_NL + KS_LB[l]]), Z, 0, 0, 0);
    }
}
template <int NCH, int S0, int S1, typename AT>
__device__ __forceinline__ void kz_mfma_range(const AT A, const uint4* B, const uint4 (&NB)[2], const uint4& BN, f32x16& Z) {
    if constexpr (S0 < S1) {
        kz_mfma<NCH, S0>(A, B, NB, BN, Z);
        kz_mfma_range<NCH, S0 + 1, S1>(A, B, NB, BN, Z);
    }
}
// n = floor(max of the lane's 16 values) and the B operand that subtracts it (see above); |n| < 2048: two bf16 pieces
template <int V, typename AT>
__device__ __forceinline__ void kz_reference(const f32x16& Z, const AT A, const uint4& NB0, unsigned lane, int& n, uint4& BN) {
    // The first read of the freshly written accumulator is an instruction the compiler knows (it places the wait states a VALU
    // read of an MFMA result needs; it does not look inside inline assembly): ONE v_max_f32 of Z[0] with itself (fmaxf of two
    // accumulator values is three instructions: each input is quieted first); the v_max3_f32 tree follows.
    const float f0 = __builtin_canonicalizef(Z[0]);
    __builtin_amdgcn_sched_barrier(0);
    const float m0 = ks_max3(f0, Z[1], Z[2]), m1 = ks_max3(Z[3], Z[4], Z[5]), m2 = ks_max3(Z[6], Z[7], Z[8]),
                m3 = ks_max3(Z[9], Z[10], Z[11]), m4 = ks_max3(Z[12], Z[13], Z[14]);
    const float ma = ks_max3(m0, m1, m2), mb = ks_max3(m3, m4, Z[15]);
    float m = ks_max3(ma, mb, mb);
    if constexpr (kz_topn<V>()) {
        // the tile's info words (behind its operands): its smallest top and an all-ones mask, or 0 and 0 for a wide tile (whose
        // top is in the accumulator already: kz_mfma, step 0)
        const uint4 ti = A[kz_nch<V>() * KS_NL + 1];
        m -= __uint_as_float(ti.x);
        BN.x = NB0.x & ti.y;
        BN.y = NB0.y & ti.y;
    }
    const float nf = __builtin_floorf(m);
    n = (int)nf;
    // n = p1 + p2, two bf16 pieces against the -1 entries of the previous set's norm operand: p1 = n rounded to bf16 (eight
    // significant bits), p2 = n - p1 (|n| < 2048: at most three bits, exact); v_cvt_pk_bf16_f32 packs both
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const bf16x2_ h1 = __builtin_convertvector((f32x2_){nf, 0.f}, bf16x2_);
    const float p1 = __uint_as_float(__builtin_bit_cast(unsigned, h1) << 16);
    const bf16x2_ pk = __builtin_convertvector((f32x2_){nf, nf - p1}, bf16x2_);
    (void)lane;
    BN.w = __builtin_bit_cast(unsigned, pk);      // K-slots 6,7 (lanes 0..31) / 14,15 (lanes 32..63), the rest stays zero: see k_wrows
}
// One step of a wave: the vector work of the finished batch (Zc, reference nc) interleaved with the whole chain of the next one
// (An x Bn -> Zn, reference nn).  Eight slots of two exponentials; the next batch's exact part goes first, its reference is taken
// in slot 3 (its two MFMAs have ~100 cycles behind them by then), the rest of the chain follows.
template <int NCH, int R, typename AT>
__device__ __forceinline__ void kz_slots(const AT An, const uint4* Bn, const uint4 (&NB)[2], f32x16& Zn, int& nn, unsigned lane,
                                         const f32x16& Zc, int nc, int hsub, double& s, KsRef& q, uint4& BN) {
    if constexpr (R < 8) {
        constexpr int NS = kz_nsteps<NCH>(), NA = kz_nexact<NCH>();   // NA exact steps, then the reference, then NS - NA more
        if constexpr (R == 0) kz_mfma_range<NCH, 0, NA>(An, Bn, NB, BN, Zn);
        if constexpr (R == 3) {
            kz_reference<NCH>(Zn, An, NB[0], lane, nn, BN);
        }
        if constexpr (R >= 3 && R < 7) {
            constexpr int r = R - 3, M = NS - NA;                        // spread the remaining M steps over slots 3..6
            constexpr int s0 = NA + (r * M) / 4, s1 = NA + ((r + 1) * M) / 4;
            if constexpr (r > 0) asm volatile("" : "+v"(Zn));
            kz_mfma_range<NCH, s0, s1>(An, Bn, NB, BN, Zn);
        }
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int i = R * 2 + g;
            const float e = __builtin_amdgcn_exp2f(Zc[i]);
            float& p = (i & 3) == 0 ? q.p0 : (i & 3) == 1 ? q.p1 : (i & 3) == 2 ? q.p2 : q.p3;
            if (i < 4) p = e; else p += e;
        }
        if constexpr (R == 7) {
            float t = q.p0 + q.p1;
            asm volatile("" : "+v"(t));
            t += q.p2 + q.p3;
            s += ldexp((double)t, nc - hsub);
            asm volatile("" : "+v"(s));
        } else {
            asm volatile("" : "+v"(q.p0), "+v"(q.p1), "+v"(q.p2), "+v"(q.p3));
        }
        __builtin_amdgcn_sched_barrier(0);
        kz_slots<NCH, R + 1>(An, Bn, NB, Zn, nn, lane, Zc, nc, hsub, s, q, BN);
    }
}

// The same step of a wave in SIXTEEN slots of one exponential: with few waves per SIMD (two at 32 parameters, one at 64) the
// instruction stream of a wave is what fills the matrix pipe -- a wave issues in order, so vector work placed behind a run of
// dependent MFMAs does not overlap with them; here at most ceil(steps / slots) MFMAs stand between two exponentials.  The exact
// steps go first (one per slot), the reference one slot behind them, the remaining steps evenly over the slots that are left.
template <int NCH, int RO, int R, typename AT>
__device__ __forceinline__ void kz_fine(const AT An, const uint4* Bn, const uint4 (&NB)[2], f32x16& Zn, int& nn, unsigned lane,
                                        const f32x16& Zc, int nc, int hsub, double& s, KsRef& q, uint4& BN) {
    if constexpr (R < 16) {
        constexpr int NS = kz_nsteps<NCH>(), NA = kz_nexact<NCH>(), R0 = NA + RO, M = NS - NA, L = 16 - R0;
        if constexpr (R < NA) kz_mfma_range<NCH, R, R + 1>(An, Bn, NB, BN, Zn);
        if constexpr (R == R0) kz_reference<NCH>(Zn, An, NB[0], lane, nn, BN);
        if constexpr (R >= R0) {
            constexpr int r = R - R0;
            constexpr int s0 = NA + (r * M) / L, s1 = NA + ((r + 1) * M) / L;
            if constexpr (r > 0) asm volatile("" : "+v"(Zn));
            kz_mfma_range<NCH, s0, s1>(An, Bn, NB, BN, Zn);
        }
        {
            const float e = __builtin_amdgcn_exp2f(Zc[R]);
            float& p = (R & 3) == 0 ? q.p0 : (R & 3) == 1 ? q.p1 : (R & 3) == 2 ? q.p2 : q.p3;
            if (R < 4) p = e; else p += e;
        }
        if constexpr (R == 15) {
            float t = q.p0 + q.p1;
            asm volatile("" : "+v"(t));
            t += q.p2 + q.p3;
            s += ldexp((double)t, nc - hsub);
            asm volatile("" : "+v"(s));
        } else {
            asm volatile("" : "+v"(q.p0), "+v"(q.p1), "+v"(q.p2), "+v"(q.p3));
        }
        __builtin_amdgcn_sched_barrier(0);
        kz_fine<NCH, RO, R + 1>(An, Bn, NB, Zn, nn, lane, Zc, nc, hsub, s, q, BN);
    }
}
template <int NCH, int FINE, typename AT>
__device__ __forceinline__ void kz_step(const AT An, const uint4* Bn, const uint4 (&NB)[2], f32x16& Zn, int& nn, unsigned lane,
                                        const f32x16& Zc, int nc, int hsub, double& s, KsRef& q, uint4& BN) {
    if constexpr (FINE > 0) kz_fine<NCH, FINE, 0>(An, Bn, NB, Zn, nn, lane, Zc, nc, hsub, s, q, BN);
    else kz_slots<NCH, 0>(An, Bn, NB, Zn, nn, lane, Zc, nc, hsub, s, q, BN);
}

template <int NCH, int WPS, bool PING = (NCH >= 4), int FINE = 0>
__global__ __launch_bounds__(256, WPS) void k_kde_split(const uint4* __restrict__ at, size_t kn,
                                                      const uint4* __restrict__ bt, unsigned nbt,
                                                      const WConst* __restrict__ wc, const int* __restrict__ ha_int,
                                                      double* __restrict__ part, const uint2* __restrict__ tmin /* KS_TOPN: two info words per tile */) {
    if (!ks_split_on(wc)) return;                             // the fp64 kernel's turn
    constexpr int OPA = kz_nch<NCH>() * KS_NL, OPB = OPA + (kz_fold<NCH>() ? 0 : 1);
    constexpr int OPT = kz_topn<NCH>() ? 1 : 0;            // one more register behind a tile's operands: its smallest top (KS_TOPN)
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t it0 = ((size_t)blockIdx.x * 4 + wv) * 2;
    const unsigned slices = gridDim.y, sl = blockIdx.y;
    const unsigned t0 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)nbt * sl / slices));
    const unsigned t1 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)nbt * (sl + 1) / slices));
    uint4 B0[OPA], B1[OPA];
#pragma unroll
    for (int q = 0; q < OPA; q++) {
        B0[q] = at[((it0 + 0) * OPA + q) * 64 + lane];
        B1[q] = at[((it0 + 1) * OPA + q) * 64 + lane];
    }
    uint4 NB[2];
    NB[0] = (lane < 32) ? make_uint4(KS_MONE | (KS_MONE << 16), KS_MONE, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    NB[1] = (lane < 32) ? make_uint4(0u, KS_MONE << 16, KS_MONE | (KS_MONE << 16), 0u) : make_uint4(0u, 0u, 0u, 0u);
    if constexpr (kz_fold<NCH>()) {
        // the A operand of the -n step: -1 in K-slots 6,7 for the rows an MFMA result holds in lanes 0..31 (bit 2 of the row clear),
        // in 14,15 for the others (what k_wrows writes into the norm operand of the other variants), nothing else
        const bool mine = (lane < 32) == (((lane & 31u) & 4u) == 0u);
        NB[0] = make_uint4(0u, 0u, 0u, mine ? (KS_MONE | (KS_MONE << 16)) : 0u);
    }
    const int hs0 = ha_int[(it0 + 0) * 32 + (lane & 31)], hs1 = ha_int[(it0 + 1) * 32 + (lane & 31)];
    double acc0 = 0.0, acc1 = 0.0;
    if (t0 < t1) {
        uint4 A[OPB + OPT];
#pragma unroll
        for (int q = 0; q < OPB; q++) A[q] = bt[((size_t)t0 * OPB + q) * 64 + lane];
        if constexpr (OPT) { const uint2 ti = tmin[t0]; A[OPB] = make_uint4(ti.x, ti.y, 0u, 0u); }
        f32x16 Z0, Z1;
        int n0 = 0, n1 = 0;
        uint4 BN = make_uint4(0u, 0u, 0u, 0u);                   // (kz_reference only ever writes its last component)
        {   // (t0, columns 0): the whole chain up front
            constexpr int NA = kz_nexact<NCH>();
            kz_mfma_range<NCH, 0, NA>(A, B0, NB, BN, Z0);
            kz_reference<NCH>(Z0, A, NB[0], lane, n0, BN);
            kz_mfma_range<NCH, NA, kz_nsteps<NCH>()>(A, B0, NB, BN, Z0);
        }
        if constexpr (!PING) {
        for (unsigned t = t0; t < t1; t++) {
            uint4 An[OPB + OPT];
            const unsigned tn = (t + 1 < t1) ? t + 1 : t;      // the last pass re-reads its own tile (no branch); unused
#pragma unroll
            for (int q = 0; q < OPB; q++) An[q] = bt[((size_t)tn * OPB + q) * 64 + lane];
            if constexpr (OPT) { const uint2 ti = tmin[tn]; An[OPB] = make_uint4(ti.x, ti.y, 0u, 0u); }
            __builtin_amdgcn_sched_barrier(0);
            KsRef q;
            kz_step<NCH, FINE>(A, B1, NB, Z1, n1, lane, Z0, n0, hs0, acc0, q, BN);      // matrix: (t, columns 1); vector: (t, columns 0)
            kz_step<NCH, FINE>(An, B0, NB, Z0, n0, lane, Z1, n1, hs1, acc1, q, BN);     // matrix: (t+1, columns 0); vector: (t, columns 1)
#pragma unroll
            for (int q2 = 0; q2 < OPB; q2++) A[q2] = An[q2];
            if constexpr (OPT) { A[OPB].x = An[OPB].x; A[OPB].y = An[OPB].y; }
        }
        } else {
        // 33..64 parameters: the two operand sets of the previous tiles (17 x 16 bytes per lane each) trade places every pass
        // instead of being copied (at 16 and 32 parameters the same loop was measured 1-3 % SLOWER than the copying one: PING stays off there) -- 68 register moves per 2048 pairs otherwise, on top of the accumulation-register traffic of a
        // kernel that needs more than 256 registers (one wave per SIMD; the compiler parks operands in accumulation registers)
        uint4 An[OPB + OPT];
        for (unsigned t = t0; t < t1; t += 2) {
            {
                const unsigned tn = (t + 1 < t1) ? t + 1 : t;
#pragma unroll
                for (int q = 0; q < OPB; q++) An[q] = bt[((size_t)tn * OPB + q) * 64 + lane];
                if constexpr (OPT) { const uint2 ti = tmin[tn]; An[OPB] = make_uint4(ti.x, ti.y, 0u, 0u); }
            }
            __builtin_amdgcn_sched_barrier(0);
            KsRef q;
            kz_step<NCH, FINE>(A, B1, NB, Z1, n1, lane, Z0, n0, hs0, acc0, q, BN);      // matrix: (t, columns 1); vector: (t, columns 0)
            kz_step<NCH, FINE>(An, B0, NB, Z0, n0, lane, Z1, n1, hs1, acc1, q, BN);     // matrix: (t+1, columns 0); vector: (t, columns 1)
            if (t + 1 >= t1) break;                                                   // (wave-uniform)
            {
                const unsigned tn = (t + 2 < t1) ? t + 2 : t + 1;
#pragma unroll
                for (int q2 = 0; q2 < OPB; q2++) A[q2] = bt[((size_t)tn * OPB + q2) * 64 + lane];
                if constexpr (OPT) { const uint2 ti = tmin[tn]; A[OPB] = make_uint4(ti.x, ti.y, 0u, 0u); }
            }
            __builtin_amdgcn_sched_barrier(0);
            kz_step<NCH, FINE>(An, B1, NB, Z1, n1, lane, Z0, n0, hs0, acc0, q, BN);     // matrix: (t+1, columns 1); vector: (t+1, columns 0)
            kz_step<NCH, FINE>(A, B0, NB, Z0, n0, lane, Z1, n1, hs1, acc1, q, BN);      // matrix: (t+2, columns 0); vector: (t+1, columns 1)
        }
        }
    }
    {
        const double v0 = acc0 + __shfl_xor(acc0, 32, 64);          // the two halves hold different previous rows
        const double v1 = acc1 + __shfl_xor(acc1, 32, 64);
        const size_t i0 = (it0 + 0) * 32 + (lane & 31), i1 = (it0 + 1) * 32 + (lane & 31);
        if (lane < 32 && i0 < kn) part[(size_t)sl * kn + i0] = v0;
        if (lane < 32 && i1 < kn) part[(size_t)sl * kn + i1] = v1;
    }
}

// 33..64 parameters with the previous tiles STAGED IN LDS (round 6).  k_kde_split keeps two operand sets of the previous tiles in
// registers (2 x 17 x 4) beside the two resident column sets (2 x 64): 354 registers, one wave per SIMD, the compiler parks operands
// in accumulation registers -- its MFMA costs 25 % more time than at 32 parameters.  Here a work-group's four waves share ONE copy of
// the tile in LDS (they all walk the same tiles), double-buffered; an MFMA's A operand is a ds_read_b128 in front of it, and the
// kernel fits two waves per SIMD, so one wave's LDS latency and vector work sit under the other's MFMAs.  The chain, its order and
// every operand are k_kde_split's: the sums are bit-identical.  One barrier per tile: the buffer written in pass t (behind the first
// step) was last read in the first step of pass t - 1, which every wave has left when any wave is past pass t - 1's barrier.
// Three chunks (33..48 parameters; k_wrows<4, 3>): 163-168 registers, THREE waves per SIMD (KDE_LDS_W3; two: + 3 %).  Measured per
// 1e10 pairs: 64 parameters 7.82 -> 6.35 ms, 48: 7.21 -> 4.92, 33: 7.17 -> 4.16 (profiles/r06_kde_by_parameters.txt); the same
// staging at 16 / 32 parameters, where the register kernel already holds three / two waves, is 7-12 % / 3.5-6 % slower and is not built;
// the sixteen-slot interleave (KDE_LDS_F4 = 1) costs 2 % here.
// (INFO: the index of the tile's info word -- KS_TOPN: its smallest top and the mask -- or -1.  That word is wave-uniform and is
// needed at the head of a chain, in front of a branch: it travels in scalar registers, fetched from the info array itself, instead of
// through LDS like the operands -- the chain then starts without a wait on LDS)
template <int INFO>
struct KsLdsOps {
    const uint4* p;                                              // the tile's base in LDS + lane
    uint4 info;
    __device__ __forceinline__ const uint4& operator[](int i) const { return (INFO >= 0 && i == INFO) ? info : p[i * 64]; }
};
#ifndef KDE_LDS_F4
#define KDE_LDS_F4 0
#endif
#ifndef KDE_LDS_W3
#define KDE_LDS_W3 3
#endif
template <int NCH, int WPS, int FINE>
__global__ __launch_bounds__(256, WPS) void k_kde_split_lds(const uint4* __restrict__ at, size_t kn,
                                                          const uint4* __restrict__ bt, unsigned nbt,
                                                          const WConst* __restrict__ wc, const int* __restrict__ ha_int,
                                                          double* __restrict__ part, const uint2* __restrict__ tmin) {
    if (!ks_split_on(wc)) return;                             // the fp64 kernel's turn
    constexpr int OPA = kz_nch<NCH>() * KS_NL, OPB = OPA + (kz_fold<NCH>() ? 0 : 1);
    constexpr int OPT = kz_topn<NCH>() ? 1 : 0;
    constexpr int TILE = OPB * 64, NR = OPA / 4;      // NR rounds of 256 words: the limb operands
    __shared__ uint4 sA[2 * TILE];
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t it0 = ((size_t)blockIdx.x * 4 + wv) * 2;
    const unsigned slices = gridDim.y, sl = blockIdx.y;
    const unsigned t0 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)nbt * sl / slices));
    const unsigned t1 = __builtin_amdgcn_readfirstlane((unsigned)((size_t)nbt * (sl + 1) / slices));
    uint4 B0[OPA], B1[OPA];
#pragma unroll
    for (int q = 0; q < OPA; q++) {
        B0[q] = at[((it0 + 0) * OPA + q) * 64 + lane];
        B1[q] = at[((it0 + 1) * OPA + q) * 64 + lane];
    }
    uint4 NB[2];
    NB[0] = (lane < 32) ? make_uint4(KS_MONE | (KS_MONE << 16), KS_MONE, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    NB[1] = (lane < 32) ? make_uint4(0u, KS_MONE << 16, KS_MONE | (KS_MONE << 16), 0u) : make_uint4(0u, 0u, 0u, 0u);
    if constexpr (kz_fold<NCH>()) {
        const bool mine = (lane < 32) == (((lane & 31u) & 4u) == 0u);
        NB[0] = make_uint4(0u, 0u, 0u, mine ? (KS_MONE | (KS_MONE << 16)) : 0u);
    }
    const int hs0 = ha_int[(it0 + 0) * 32 + (lane & 31)], hs1 = ha_int[(it0 + 1) * 32 + (lane & 31)];
    double acc0 = 0.0, acc1 = 0.0;
    if (t0 < t1) {                                              // (uniform over the work-group: the barriers below are too)
        uint4 r[NR], r4 = make_uint4(0u, 0u, 0u, 0u);
        // a tile is OPB x 64 consecutive 16-byte words; 256 threads fetch it in rounds and write it where it came from
        // (the limb operands are NR whole rounds of the work-group; the norm operand -- plain and KS_TOPN variants -- is wave 0's extra word)
        static_assert(OPA * 64 == NR * 256, "a tile's limb operands are whole rounds of the work-group");
#define KZ_TILE_FETCH(T)                                                                                            \
        {                                                                                                           \
            const uint4* src = bt + (size_t)(T) * (OPB * 64) + threadIdx.x;                                         \
            _Pragma("unroll") for (int j = 0; j < NR; j++) r[j] = src[256 * j];                                     \
            if constexpr (OPB > OPA) { if (threadIdx.x < 64) r4 = src[256 * NR]; }                                  \
        }
#define KZ_TILE_STORE(BUF)                                                                                          \
        {                                                                                                           \
            uint4* dst = sA + (BUF) * TILE + threadIdx.x;                                                           \
            _Pragma("unroll") for (int j = 0; j < NR; j++) dst[256 * j] = r[j];                                     \
            if constexpr (OPB > OPA) { if (threadIdx.x < 64) dst[256 * NR] = r4; }                                  \
        }
        KZ_TILE_FETCH(t0)
        KZ_TILE_STORE(0)
        __syncthreads();
        f32x16 Z0, Z1;
        int n0 = 0, n1 = 0;
        uint4 BN = make_uint4(0u, 0u, 0u, 0u);
        unsigned cur = 0;
        {   // (t0, columns 0): the whole chain up front
            constexpr int NA = kz_nexact<NCH>();
            KsLdsOps<OPT ? OPB : -1> A{sA + lane, make_uint4(0u, 0u, 0u, 0u)};
            if constexpr (OPT) { const uint2 ti = tmin[t0]; A.info = make_uint4(ti.x, ti.y, 0u, 0u); }
            kz_mfma_range<NCH, 0, NA>(A, B0, NB, BN, Z0);
            kz_reference<NCH>(Z0, A, NB[0], lane, n0, BN);
            kz_mfma_range<NCH, NA, kz_nsteps<NCH>()>(A, B0, NB, BN, Z0);
        }
        for (unsigned t = t0; t < t1; t++) {
            const unsigned tn = (t + 1 < t1) ? t + 1 : t;      // the last pass stages its own tile once more (no branch); unused
            KZ_TILE_FETCH(tn)
            __builtin_amdgcn_sched_barrier(0);
            KsLdsOps<OPT ? OPB : -1> A{sA + cur * TILE + lane, make_uint4(0u, 0u, 0u, 0u)}, An{sA + (cur ^ 1u) * TILE + lane, make_uint4(0u, 0u, 0u, 0u)};
            if constexpr (OPT) {
                const uint2 ti = tmin[t], tj = tmin[tn];       // (wave-uniform addresses: scalar loads)
                A.info = make_uint4(ti.x, ti.y, 0u, 0u);
                An.info = make_uint4(tj.x, tj.y, 0u, 0u);
            }
            KsRef q;
            kz_step<NCH, FINE>(A, B1, NB, Z1, n1, lane, Z0, n0, hs0, acc0, q, BN);      // matrix: (t, columns 1); vector: (t, columns 0)
            KZ_TILE_STORE(cur ^ 1u)
            __syncthreads();
            kz_step<NCH, FINE>(An, B0, NB, Z0, n0, lane, Z1, n1, hs1, acc1, q, BN);     // matrix: (t+1, columns 0); vector: (t, columns 1)
            cur ^= 1u;
        }
#undef KZ_TILE_FETCH
#undef KZ_TILE_STORE
    }
    {
        const double v0 = acc0 + __shfl_xor(acc0, 32, 64);
        const double v1 = acc1 + __shfl_xor(acc1, 32, 64);
        const size_t i0 = (it0 + 0) * 32 + (lane & 31), i1 = (it0 + 1) * 32 + (lane & 31);
        if (lane < 32 && i0 < kn) part[(size_t)sl * kn + i0] = v0;
        if (lane < 32 && i1 < kn) part[(size_t)sl * kn + i1] = v1;
    }
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

// Sum of squares of the weights (the L2 normalisation, AbcUtil.cpp:583) in a fixed, launch-independent order, so that every
// path -- k_wfinish below, the stand-alone k_sumsq_partial -- produces the same bits: work-groups of 64 consecutive rows
// (thread r < 64 of a 256-thread group holds row r's square, the other threads 0; block_sum_256's tree) write one partial each;
// the total is the partials added in index order by ONE work-group's worth of threads -- thread t takes t, t + 256, ..., then
// the same tree (sumsq_total) -- in the prologue of the normalisation kernel (every work-group repeats it: the same bits
// everywhere) or, beyond SQ_INLINE_MAX partials, by a one-group launch in front of it.  (A "last group finishes" ticket inside
// the producing kernel was measured first: the device-scope fence it needs writes back the XCD's L2 per work-group -- k_wfinish
// went from 19 to 46 us.  Kernel boundaries are the cheap coherence point on this chip.)
constexpr unsigned SQ_INLINE_MAX = 4096;
__device__ __forceinline__ double sumsq_total(const double* __restrict__ sq_part, unsigned nparts, double* sm /* >= 4 */) {
    double t = 0.0;
    for (unsigned j = threadIdx.x; j < nparts; j += 256) t += sq_part[j];
    return block_sum_256(t, sm);
}
__global__ __launch_bounds__(256) void k_sumsq_total(const double* __restrict__ sq_part, unsigned nparts, double* __restrict__ sq_total) {
    __shared__ double sm[4];
    const double t = sumsq_total(sq_part, nparts, sm);
    if (threadIdx.x == 0) *sq_total = t;
}

// w_raw[i] = prod_p likelihood_p(theta_ip) / (C * sum over slices)   (AbcUtil.cpp:557-580)
// FOUR waves per 64 rows (round 3: one thread per row walked 16 likelihoods -- a division, an exponential, a square root each --
// and ~30 slice partials as ONE dependent chain, 19 us at K = 1e5 on a quarter-filled chip): wave q takes the parameters and the
// slices = q (mod 4) -- the parameter stays wave-uniform: its prior comes through the scalar cache and only its kind's code runs --,
// the four partial products / sums meet in LDS and wave 0 finishes the row.
// sq_part != NULL: the 64-row partials of the raw weights' sum of squares come out of the same launch (see sumsq_total).
__global__ __launch_bounds__(256) void k_wfinish(const abc_prior* __restrict__ priors, const double* __restrict__ theta,
                                                 size_t K, int P, size_t k0, size_t kn,
                                                 const double* __restrict__ part, int slices,
                                                 const WConst* __restrict__ wc, double* __restrict__ w_raw,
                                                 int split_launched, int epan, int* __restrict__ which,
                                                 const unsigned char* __restrict__ far_flag, const double* __restrict__ fix_i,
                                                 const double* __restrict__ fix_j, const double* __restrict__ ha_frac,
                                                 double* __restrict__ sq_part) {
    __shared__ double sm[4];
    __shared__ double sn[4][64], sd[4][64];
    const int wq = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool split_on = split_launched && ks_split_on(wc);
    if (blockIdx.x == 0 && threadIdx.x == 0) *which = split_on ? ABC_KDE_RAN_SPLIT : ABC_KDE_RAN_FP64;       // abc_kde_last_kernel
    const bool active = i < kn;
    double num = 1.0, den = 0.0;
    const bool far_row = active && split_on && far_flag[i];
    if (active) {
        for (int p = wq; p < P; p += 4) num *= prior_likelihood(priors[p], theta[(k0 + i) + K * (size_t)p]);
        if (!far_row) for (int s = wq; s < slices; s += 4) den += part[(size_t)s * kn + i];
    }
    sn[wq][lane] = num;
    sd[wq][lane] = den;
    __syncthreads();
    double wv = 0.0;
    if (wq == 0 && active) {
        num = (sn[0][lane] * sn[1][lane]) * (sn[2][lane] * sn[3][lane]);
        if (far_row) {
            den = fix_i[i];                                      // a far new particle: summed in fp64 by k_kde_fixups
        } else {
            den = (sd[0][lane] + sd[1][lane]) + (sd[2][lane] + sd[3][lane]);
            if (split_on) den *= exp2(-ha_frac[i]);              // the fraction of 1/2|a_i|^2 the split kernel left out (k_wrows)
            if (split_on && wc->nfar_j > 0) den += fix_j[i];     // + the far previous particles (k_kde_fixups)
        }
        if (epan) wv = (den > 0.0) ? num / den : 0.0;            // compact support: a particle nothing supports gets weight 0
        else wv = num / (wc->C * den);
        w_raw[i] = wv;
    }
    if (sq_part) {
        const double bs = block_sum_256(wv * wv, sm);
        if (threadIdx.x == 0) sq_part[blockIdx.x] = bs;
    }
}

__global__ __launch_bounds__(256) void k_fill(double* __restrict__ w, size_t K, double v) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < K) w[i] = v;
}

// the same partials for weights that are already in memory (stage-level callers, the sharded driver's gathered slices)
__global__ __launch_bounds__(256) void k_sumsq_partial(const double* __restrict__ w, size_t K, double* __restrict__ sq_part) {
    __shared__ double sm[4];
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    double wv = 0.0;
    if (threadIdx.x < 64 && i < K) wv = w[i];
    const double bs = block_sum_256(wv * wv, sm);
    if (threadIdx.x == 0) sq_part[blockIdx.x] = bs;
}

// host_mirror (optional): a pinned, device-visible buffer that receives the normalised weights as they are written -- the
// host's alias build needs them next, and a store over PCIe from here saves the blit copy and its launch gap behind this kernel
// nparts != 0: sq points at the partials, summed here (sumsq_total); nparts == 0: at the finished total
__global__ __launch_bounds__(256) void k_div_norm(double* __restrict__ w, size_t K, const double* __restrict__ sq_in, unsigned nparts,
                                                  double* __restrict__ host_mirror) {
    __shared__ double sm[4];
    const double sq = nparts ? sumsq_total(sq_in, nparts, sm) : *sq_in;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    double v = w[i];
    if (sq > 0.0) {                              // Eigen normalize(): only if squaredNorm > 0
        v = v / sqrt(sq);
        w[i] = v;
    }
    if (host_mirror) host_mirror[i] = v;
}

}  // namespace

int abc_kde_words(abc_ctx* ctx);

int launch_gather_rows(abc_ctx* ctx, const double* Y, size_t n_local, size_t ldy, size_t P, const uint64_t* idx,
                       size_t K, uint64_t idx_base, double* theta, size_t ldt, const int* sel_fail, int* sel_fail_pin,
                       hipEvent_t done) {
    const size_t tot = K * P;
    if (!tot) { if (done) ABC_HIP(ctx, hipEventRecord(done, ctx->stream)); return ABC_OK; }
    StageTimer tm(ctx, ST_GATHER_DV);
    if (done)
        hipExtLaunchKernelGGL(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, nullptr, done, 0, Y, n_local, ldy,
                              (int)P, (const unsigned long long*)idx, K, (unsigned long long)idx_base, theta, ldt, sel_fail, sel_fail_pin,
                              ctx->giveups_dev);
    else
        hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, Y, n_local, ldy,
                           (int)P, (const unsigned long long*)idx, K, (unsigned long long)idx_base, theta, ldt, sel_fail, sel_fail_pin,
                           ctx->giveups_dev);
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

// Everything of the weight stage that needs the PREVIOUS set only: scales and constants (k_wprep), robust centre, the scaled
// row-major copy b with hb, the limb tiles of the split kernel.  The fused drivers queue it on the side stream at the start of a
// generation (it runs beside the ranking); st == NULL: the context's stream.  kn_max: the most rows a later launch_weights_raw
// will handle (its far-row budget).
// ---- KS_TOPN: the previous set's tiles in the order of the rows' norm tops ------------------------------------------------
// grouping key of a previous row, step 1: its hb = 1/2 |b|^2 - log2 w' as k_wrows will see it (KS_HB_ZERO for a row that takes no
// part in the matrix work: weight 0, far), and the smallest / largest hb of the rows that do (exact, order-independent: atomics on
// order-preserving integers)
__device__ __forceinline__ unsigned long long ks_okey(double x) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ks_okey_inv(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__global__ __launch_bounds__(256) void k_whb(const double* __restrict__ prev, size_t Kp, int P, WConst* __restrict__ wc,
                                             const double* __restrict__ w, double* __restrict__ hbk) {
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= Kp) return;
    double nn = 0.0;
    bool far8 = false;
    for (int p = 0; p < P; p++) {
        const double v = (prev[r + Kp * (size_t)p] - wc->centre[p]) * wc->scale[p];
        nn = fma(v, v, nn);
        far8 = far8 || !(fabs(v) <= KS_BOUND);
    }
    const double wr = w[r];
    double k = KS_HB_ZERO;
    if (wr != 0.0) {
        const double lw = -log2(wr);
        if ((lw >= KS_LW_MIN && lw <= KS_LW_CAP) && !far8 && (nn <= KS_NORM2)) k = 0.5 * nn + lw;
    }
    hbk[r] = k;
    if (k < KS_HB_ZERO) { const unsigned long long o = ks_okey(k); atomicMin(&wc->hb_lo, o); atomicMax(&wc->hb_hi, o); }
}
// ... step 2: 255 equal bins between the two (bin 255: the rows that take no part -- they go behind all others) and ONE stable
// counting pass on that bin (per-block histograms, a scan, ranks by wave ballots: the scheme of select.hip's radix passes, here with
// the bin computed on the fly and the row's RANK as the only output).  Tiles only need rows of SIMILAR tops -- a tile's spread is a
// bin or two, 1/255 of the set's range (0.14 at the bench's sets) --, and where the set is sparse the kernel handles a tile the
// plain way (k_tile_tmin).  (A full 64-bit sort of 1e5 keys took 0.12 ms on the side stream, two generic 8-bit passes 0.11.)
constexpr int WQ_ITEMS = 8, WQ_CHUNK = 256 * WQ_ITEMS;      // rows per work-group; each wave owns WQ_CHUNK / 4 of them, in order
__device__ __forceinline__ unsigned ks_qbin(double k, double lo, double hi) {
    if (!(k < KS_HB_ZERO)) return 255u;
    const double t = (hi > lo) ? (k - lo) / (hi - lo) * 255.0 : 0.0;
    return (t >= 254.0) ? 254u : (t <= 0.0 ? 0u : (unsigned)t);
}
__global__ __launch_bounds__(256) void k_wq_hist(const double* __restrict__ hbk, size_t Kp, const WConst* __restrict__ wc,
                                                 unsigned* __restrict__ bh /* [256][nb] */, int nb) {
    __shared__ unsigned lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const double lo = ks_okey_inv(wc->hb_lo), hi = ks_okey_inv(wc->hb_hi);
    const size_t base = (size_t)blockIdx.x * WQ_CHUNK;
#pragma unroll
    for (int j = 0; j < WQ_ITEMS; j++) {
        const size_t i = base + (size_t)j * 256 + threadIdx.x;
        if (i < Kp) atomicAdd(&lh[ks_qbin(hbk[i], lo, hi)], 1u);
    }
    __syncthreads();
    bh[(size_t)threadIdx.x * nb + blockIdx.x] = lh[threadIdx.x];
}
// exclusive scan of the bin-major [256][nb] histogram = of the array as it lies in memory; one work-group, through LDS
__global__ __launch_bounds__(1024) void k_wq_scan(unsigned* __restrict__ bh, int total) {
    __shared__ unsigned wsum[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (total + 1023) / 1024, i0 = t * per;
    unsigned sum = 0;
    for (int i = i0; i < i0 + per && i < total; i++) sum += bh[i];
    unsigned inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int w = 0; w < wave; w++) run += wsum[w];
    for (int i = i0; i < i0 + per && i < total; i++) { const unsigned v = bh[i]; bh[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_wq_rank(const double* __restrict__ hbk, size_t Kp, const WConst* __restrict__ wc,
                                                 const unsigned* __restrict__ bh, int nb, unsigned* __restrict__ rank) {
    __shared__ unsigned whist[4][256];
    __shared__ volatile unsigned woff[4][256];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) whist[w][t] = 0;
    __syncthreads();
    const double lo = ks_okey_inv(wc->hb_lo), hi = ks_okey_inv(wc->hb_hi);
    const size_t seg = (size_t)blockIdx.x * WQ_CHUNK + (size_t)wave * (WQ_CHUNK / 4);
    unsigned q[WQ_ITEMS];
#pragma unroll
    for (int j = 0; j < WQ_ITEMS; j++) {
        const size_t i = seg + (size_t)j * 64 + lane;
        q[j] = (i < Kp) ? ks_qbin(hbk[i], lo, hi) : 0u;
        if (i < Kp) atomicAdd(&whist[wave][q[j]], 1u);
    }
    __syncthreads();
    {
        unsigned run = bh[(size_t)t * nb + blockIdx.x];          // scanned: the first position of (bin t, this block)
#pragma unroll
        for (int w = 0; w < 4; w++) { woff[w][t] = run; run += whist[w][t]; }
    }
    __syncthreads();
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < WQ_ITEMS; j++) {
        const size_t i = seg + (size_t)j * 64 + lane;
        const bool valid = i < Kp;
        const unsigned d = q[j];
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        if (valid) rank[i] = woff[wave][d] + (unsigned)__popcll(peers & lt_mask);
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0) woff[wave][d] += (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
}
// a tile's info words: (smallest top, all ones), or (0, 0) if the tops of its rows that take part (below KS_HB_ZERO) spread over
// more than KS_TOPN_SPREAD -- then the kernel handles the tile as the plain variant would
constexpr float KS_TOPN_SPREAD = 0.75f;
__global__ __launch_bounds__(256) void k_tile_tmin(const float* __restrict__ topf, size_t ntiles, uint2* __restrict__ tinfo) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int j = 0; j < 32; j++) {
        const float v = topf[t * 32 + j];
        if (v < (float)KS_HB_ZERO) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    }
    if (mx < mn) { mn = (float)KS_HB_ZERO; mx = mn; }              // no row of the tile takes part
    const bool wide = !(mx - mn <= KS_TOPN_SPREAD);
    tinfo[t] = wide ? make_uint2(0u, 0u) : make_uint2(__float_as_uint(mn), 0xffffffffu);
}
// full chunks (no spare K-slot) and enough pairs for the sort to pay (it runs on the side stream of the fused drivers, in front of a
// stage-level call): the variants with one MFMA less
static bool ks_topn_on(size_t P, bool split, bool fold, size_t pairs) {
    if (!split || fold) return false;
    const char* e = abc_diag_env("ABC_KDE_TOPN_MIN_PAIRS");                       // (tests: 0 = always; A/B: a huge number = never)
    const double minp = e ? atof(e) : 4.0e9;
    return (double)pairs >= minp;
}

// 33..64 parameters: the previous tiles staged in LDS (k_kde_split_lds; ABC_KDE_LDS=0 under ABC_DIAG: the register-resident kernel of
// rounds 3-5), and with it three stored chunks instead of four at 33..48 parameters (ABC_KDE_CHUNKS3=0: four)
static bool ks_lds_on() {
    static const bool on = [] { const char* e = abc_diag_env("ABC_KDE_LDS"); return !e || atoi(e) != 0; }();
    return on;
}
static int ks_chunks(size_t P) {
    static const bool three = [] { const char* e = abc_diag_env("ABC_KDE_CHUNKS3"); return !e || atoi(e) != 0; }();
    return (P <= 16) ? 1 : (P <= 32) ? 2 : (P <= 48 && three && ks_lds_on()) ? 3 : 4;
}
// up to 13 / 17..29 / 33..45 / 49..61 parameters: the variants of the split kernel with two MFMAs fewer (norm pieces in the spare K-slots: KS_FOLD)
static bool ks_fold_on(size_t P, bool split) {
    static const bool off = abc_diag_env("ABC_KDE_NOFOLD") != nullptr;              // A/B switch for measurements
    return split && P + 3 <= (size_t)16 * ks_chunks(P) && !off;               // three spare K-slots in the last chunk
}

int launch_weights_prev(abc_ctx* ctx, size_t P, size_t kn_max, const double* theta_prev, size_t Kp, const double* w_prev,
                        const double* dv_prev, abc_wprev* out, hipStream_t st) {
    memset(out, 0, sizeof(*out));
    if (P > (size_t)W_MAXP) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "weights: P = %zu > %d parameters", P, W_MAXP);
    if (Kp > 0xffffffffull) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "weights: K' = %zu >= 2^32", Kp);
    hipStream_t s = st ? st : ctx->stream;
    int PP = 2;
    while (PP < (int)P) PP *= 2;
    if (P > 64) PP = (int)((P + 63) / 64 * 64);
    const int NCH = ks_chunks(P);                                   // (3: dealt out as four, three stored)
    const bool epan = ctx->weight_kernel == ABC_WEIGHT_EPANECHNIKOV;
    const bool split = (PP >= 8 && P <= 64 && ctx->kde_mode != ABC_KDE_FP64 && !epan);
    const size_t nbt = (Kp + 31) / 32;
    const bool fold = ks_fold_on(P, split);
    const int opb = NCH * KS_NL + (fold ? 0 : 1);
    WConst* wc = (WConst*)abc_ws_alloc(ctx, sizeof(WConst));
    double* b = (double*)abc_ws_alloc(ctx, Kp * PP * sizeof(double));
    double* hb = (double*)abc_ws_alloc(ctx, Kp * sizeof(double));
    double* cpart = (double*)abc_ws_alloc(ctx, (P > 64 ? P : 64) * WC_NB * sizeof(double));
    unsigned short* bt = nullptr;
    unsigned* far_list = nullptr;
    const bool topn = ks_topn_on(P, split, fold, Kp * kn_max);
    unsigned* rank = nullptr;
    uint2* tmin = nullptr;
    float* topf = nullptr;
    double* hbk = nullptr;
    unsigned* qh = nullptr;
    if (split) {
        bt = (unsigned short*)abc_ws_alloc(ctx, nbt * opb * 1024);
        far_list = (unsigned*)abc_ws_alloc(ctx, KS_MAX_FAR_J * sizeof(unsigned));
        if (!bt || !far_list) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
        if (topn) {
            rank = (unsigned*)abc_ws_alloc(ctx, Kp * sizeof(unsigned));
            tmin = (uint2*)abc_ws_alloc(ctx, nbt * sizeof(uint2));
            topf = (float*)abc_ws_alloc(ctx, nbt * 32 * sizeof(float));
            hbk = (double*)abc_ws_alloc(ctx, Kp * sizeof(double));
            qh = (unsigned*)abc_ws_alloc(ctx, (size_t)256 * ((Kp + WQ_CHUNK - 1) / WQ_CHUNK) * sizeof(unsigned));
            if (!rank || !tmin || !topf || !hbk || !qh) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
        }
    }
    if (!wc || !b || !hb || !cpart) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
    const double* centre = (const double*)((const char*)wc + offsetof(WConst, centre));
    hipLaunchKernelGGL(k_wprep, dim3(1), dim3(64), 0, s, dv_prev, (int)P, wc, (int)(kn_max / 16 + 32));
    hipLaunchKernelGGL(k_wcentre, dim3((unsigned)P, WC_NB), dim3(256), 0, s, theta_prev, Kp, wc, cpart);
    hipLaunchKernelGGL(k_wcentre_finish, dim3(1), dim3(P > 64 ? 1024 : 64), 0, s, theta_prev, Kp, (int)P, cpart, wc);
    if (topn) {          // the rows' ranks by norm top (KS_TOPN): keys, a stable sort, the inverse permutation
        const int nbq = (int)((Kp + WQ_CHUNK - 1) / WQ_CHUNK);
        hipLaunchKernelGGL(k_whb, dim3((unsigned)((Kp + 255) / 256)), dim3(256), 0, s, theta_prev, Kp, (int)P, wc, w_prev, hbk);
        hipLaunchKernelGGL(k_wq_hist, dim3(nbq), dim3(256), 0, s, (const double*)hbk, Kp, (const WConst*)wc, qh, nbq);
        hipLaunchKernelGGL(k_wq_scan, dim3(1), dim3(1024), 0, s, qh, 256 * nbq);
        hipLaunchKernelGGL(k_wq_rank, dim3(nbq), dim3(256), 0, s, (const double*)hbk, Kp, (const WConst*)wc, (const unsigned*)qh, nbq, rank);
        ABC_HIP(ctx, hipGetLastError());
    }
    if (split) {         // scaled copy, hb and the limb tiles in one pass (k_wrows)
        const size_t rbp = nbt * 32;
        if (NCH == 1)
            hipLaunchKernelGGL(k_wrows<1>, dim3((unsigned)((rbp / 32 + 3) / 4)), dim3(256), 0, s, theta_prev, Kp, Kp, (int)P, PP, rbp, wc,
                               w_prev, 1, b, hb, bt, opb, (unsigned char*)nullptr, far_list, (int*)nullptr, (double*)nullptr, fold ? 1 : 0, (const unsigned*)rank, topf);
        else if (NCH == 2)
            hipLaunchKernelGGL(k_wrows<2>, dim3((unsigned)((rbp / 16 + 3) / 4)), dim3(256), 0, s, theta_prev, Kp, Kp, (int)P, PP, rbp, wc,
                               w_prev, 1, b, hb, bt, opb, (unsigned char*)nullptr, far_list, (int*)nullptr, (double*)nullptr, fold ? 1 : 0, (const unsigned*)rank, topf);
        else if (NCH == 3)
            hipLaunchKernelGGL((k_wrows<4, 3>), dim3((unsigned)((rbp / 8 + 3) / 4)), dim3(256), 0, s, theta_prev, Kp, Kp, (int)P, PP, rbp, wc,
                               w_prev, 1, b, hb, bt, opb, (unsigned char*)nullptr, far_list, (int*)nullptr, (double*)nullptr, fold ? 1 : 0, (const unsigned*)rank, topf);
        else
            hipLaunchKernelGGL(k_wrows<4>, dim3((unsigned)((rbp / 8 + 3) / 4)), dim3(256), 0, s, theta_prev, Kp, Kp, (int)P, PP, rbp, wc,
                               w_prev, 1, b, hb, bt, opb, (unsigned char*)nullptr, far_list, (int*)nullptr, (double*)nullptr, fold ? 1 : 0, (const unsigned*)rank, topf);
    } else {
        hipLaunchKernelGGL(k_wscale, dim3((unsigned)((Kp + 255) / 256)), dim3(256), 0, s, theta_prev, Kp, Kp,
                           (int)P, PP, wc, centre, (size_t)1, w_prev, b, hb);
    }
    if (topn) hipLaunchKernelGGL(k_tile_tmin, dim3((unsigned)((nbt + 255) / 256)), dim3(256), 0, s, (const float*)topf, nbt, tmin);
    ABC_HIP(ctx, hipGetLastError());
    out->wc = wc; out->b = b; out->hb = hb; out->bt = bt; out->far_list = far_list; out->tmin = (unsigned*)tmin;
    out->Kp = Kp; out->P = P; out->kn_max = kn_max; out->split = split ? 1 : 0; out->ready = 1;
    return ABC_OK;
}

int launch_weights_raw(abc_ctx* ctx, const abc_prior* priors, const double* theta, size_t K, size_t P, size_t k0,
                       size_t kn, const double* theta_prev, size_t Kp, const double* w_prev, const double* dv_prev,
                       double* w_raw, const abc_wprev* prev, const double** sumsq_out) {
    if (sumsq_out) *sumsq_out = nullptr;
    if (P > (size_t)W_MAXP) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "weights: P = %zu > %d parameters", P, W_MAXP);
    if (kn == 0) return ABC_OK;
    ABC_TRY(abc_kde_words(ctx));
    if (Kp > 0xffffffffull) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "weights: K' = %zu >= 2^32", Kp);
    if (k0 + kn > K) ABC_FAIL(ctx, ABC_ERR_INVALID, "weights: row range [%zu,%zu) outside K=%zu", k0, k0 + kn, K);
    int PP = 2;
    while (PP < (int)P) PP *= 2;
    if (P > 64) PP = (int)((P + 63) / 64 * 64);
    const size_t rb = (kn + 255) / 256;
    // Column slices: many short ones.  Work-groups are dispatched slice by slice, so the ~2000 resident groups all
    // stream the same few hundred KB of the previous set through the scalar cache / L2; with 6 slices of 2 MB each
    // (one residency) the same kernel was 25 % slower, waiting on scalar loads (measured: 6 -> 12.0 ms, 16 -> 10.0,
    // 64 -> 9.1, 128 -> 9.05 at K = K' = 1e5).  Bounded by the partial-sum buffer (slices x kn doubles <= 64 MB).
    size_t slices = abc_kde_slices(kn, Kp, PP);
    if (slices > Kp / 64) slices = Kp / 64;
    if (slices < 1) slices = 1;
    if (slices > 1024) slices = 1024;
    // split-operand kernel: 5 <= P <= 64 parameters (padded width 8, 16, 32 or 64: one, two or four 16-parameter chunks; below
    // that the fp64 body is as short), unless the caller asked for fp64
    const int NCH = ks_chunks(P);
    const bool epan = ctx->weight_kernel == ABC_WEIGHT_EPANECHNIKOV;
    const bool split = (PP >= 8 && P <= 64 && ctx->kde_mode != ABC_KDE_FP64 && !epan);
    const size_t nbt = (Kp + 31) / 32, nat = rb * 8;
    const int opa = NCH * KS_NL;
    const bool fold = ks_fold_on(P, split);
    if (split) {
        // Every work-group of the split kernel loads its 32-64 KB of resident operands once per slice: with the 65 slices
        // the fp64 kernel likes that was 1.2 GB of fabric traffic per launch at K = K' = 1e5 (PMC).  Its time is flat between
        // 24 and 130 slices, so it gets about 12k work-groups (16 residencies of the chip) and no more.
        size_t cap = 12288 / rb;
        if (cap < 8) cap = 8;
        if (slices > cap) slices = cap;
    }
    // the previous set's share: prepared by the caller (side stream, already joined) or here
    abc_wprev own;
    if (!prev || !prev->ready) {
        ABC_TRY(launch_weights_prev(ctx, P, kn, theta_prev, Kp, w_prev, dv_prev, &own, nullptr));
        prev = &own;
    }
    if (prev->Kp != Kp || prev->P != P || prev->kn_max < kn || (prev->split != 0) != split)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "weights: the prepared previous set does not match this call");
    WConst* wc = (WConst*)prev->wc;
    double* b = prev->b;
    double* hb = prev->hb;
    unsigned short* bt = prev->bt;
    unsigned* far_list = prev->far_list;
    double* a = (double*)abc_ws_alloc(ctx, kn * PP * sizeof(double));
    double* part = (double*)abc_ws_alloc(ctx, slices * kn * sizeof(double));
    unsigned short* at = nullptr;
    unsigned char* far_flag = nullptr;
    double *fix_i = nullptr, *fix_j = nullptr, *ha_frac = nullptr;
    int* ha_int = nullptr;
    if (split) {
        at = (unsigned short*)abc_ws_alloc(ctx, nat * opa * 1024);
        far_flag = (unsigned char*)abc_ws_alloc(ctx, nat * 32);
        fix_i = (double*)abc_ws_alloc(ctx, kn * sizeof(double));
        fix_j = (double*)abc_ws_alloc(ctx, kn * sizeof(double));
        ha_frac = (double*)abc_ws_alloc(ctx, nat * 32 * sizeof(double));
        ha_int = (int*)abc_ws_alloc(ctx, nat * 32 * sizeof(int));
        if (!at || !far_flag || !fix_i || !fix_j || !ha_frac || !ha_int) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
    }
    if (!a || !part) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
    const double* centre = (const double*)((const char*)wc + offsetof(WConst, centre));
    StageTimer tm(ctx, ST_WEIGHTS_MISC);
    if (split) {
        const size_t ra = nat * 32;
        if (NCH == 1)
            hipLaunchKernelGGL(k_wrows<1>, dim3((unsigned)((ra / 32 + 3) / 4)), dim3(256), 0, ctx->stream, theta + k0, kn, K, (int)P, PP, ra,
                               wc, (const double*)nullptr, 0, a, (double*)nullptr, at, opa, far_flag, far_list, ha_int, ha_frac, fold ? 1 : 0, (const unsigned*)nullptr, (float*)nullptr);
        else if (NCH == 2)
            hipLaunchKernelGGL(k_wrows<2>, dim3((unsigned)((ra / 16 + 3) / 4)), dim3(256), 0, ctx->stream, theta + k0, kn, K, (int)P, PP, ra,
                               wc, (const double*)nullptr, 0, a, (double*)nullptr, at, opa, far_flag, far_list, ha_int, ha_frac, fold ? 1 : 0, (const unsigned*)nullptr, (float*)nullptr);
        else if (NCH == 3)
            hipLaunchKernelGGL((k_wrows<4, 3>), dim3((unsigned)((ra / 8 + 3) / 4)), dim3(256), 0, ctx->stream, theta + k0, kn, K, (int)P, PP, ra,
                               wc, (const double*)nullptr, 0, a, (double*)nullptr, at, opa, far_flag, far_list, ha_int, ha_frac, fold ? 1 : 0, (const unsigned*)nullptr, (float*)nullptr);
        else
            hipLaunchKernelGGL(k_wrows<4>, dim3((unsigned)((ra / 8 + 3) / 4)), dim3(256), 0, ctx->stream, theta + k0, kn, K, (int)P, PP, ra,
                               wc, (const double*)nullptr, 0, a, (double*)nullptr, at, opa, far_flag, far_list, ha_int, ha_frac, fold ? 1 : 0, (const unsigned*)nullptr, (float*)nullptr);
    } else {
        hipLaunchKernelGGL(k_wscale, dim3((unsigned)((kn + 255) / 256)), dim3(256), 0, ctx->stream, theta + k0, kn, K,
                           (int)P, PP, wc, centre, (size_t)1, (const double*)nullptr, a, (double*)nullptr);
    }
#define LAUNCH_KDE(PPV)                                                                                        \
    hipLaunchKernelGGL(k_kde<PPV>, dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream, a, kn, b, \
                       Kp, hb, wc, theta, K, k0, theta_prev, part, split ? 1 : 0)
    if (epan) {
        StageTimer tk(ctx, ST_KDE);
#define LAUNCH_EPAN(PPV) hipLaunchKernelGGL(k_epan<PPV>, dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream, a, kn, b, Kp, w_prev, wc, (int)P, part)
        if (PP > 64)
            hipLaunchKernelGGL(k_kde_gen, dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream, a, kn, b, Kp, hb, wc, theta, K, k0,
                               theta_prev, part, PP, 1, (int)P, w_prev);
        else
        switch (PP) {
            case 2: LAUNCH_EPAN(2); break;
            case 4: LAUNCH_EPAN(4); break;
            case 8: LAUNCH_EPAN(8); break;
            case 16: LAUNCH_EPAN(16); break;
            case 32: LAUNCH_EPAN(32); break;
            default: LAUNCH_EPAN(64); break;
        }
#undef LAUNCH_EPAN
    } else {
        StageTimer tk(ctx, ST_KDE);
        const bool kde_lds = ks_lds_on();
        if (split) {
            // three waves per SIMD at 16 parameters (129 VGPRs; four, with two spills: no faster), two at 32 (192)
            // ... one at 64 on this kernel (more than 256 registers: the two resident column operand sets alone are 128) -- which is
            // why 33..64 parameters go to k_kde_split_lds since round 6: the previous tiles in LDS, two waves per SIMD at four chunks
            // (194-205 registers), three at three chunks (up to 48 parameters; 163-168): 7.2-7.9 -> 4.2-6.3 ms per 1e10 pairs
            // (16 parameters: the sixteen-slot interleave, reference one slot behind the exact steps: 2.244 -> 2.220 ms per 1e10 pairs,
            // and ON TOP of it the operand sets trading places instead of being copied: -> 2.13-2.17 ms (without the finer interleave
            // the same loop was 2 % slower than the copying one, rounds 2 and 3); at 32 parameters either change and both together are
            // 0.7-2 % slower than the eight-slot copying loop, at 64 -- one wave per SIMD -- the finer interleave costs 50 %: they keep
            // kz_slots; four waves per SIMD, the reference two slots behind: +0.5 %)
            if (fold && NCH == 1)           // (up to 13 parameters: seven MFMAs per 1024 pairs, no norm operand)
                hipLaunchKernelGGL((k_kde_split<KS_FOLD + 1, 3, true, 1>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (fold && NCH == 2)      // (17..29: thirteen instead of fifteen)
                hipLaunchKernelGGL((k_kde_split<KS_FOLD + 2, 2, false, 0>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (fold && NCH == 3)      // (33..45: three chunks, nineteen MFMAs per 1024 pairs; previous tiles in LDS, two waves per SIMD)
                hipLaunchKernelGGL((k_kde_split_lds<KS_FOLD + 3, KDE_LDS_W3, KDE_LDS_F4>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (fold && kde_lds)       // (49..61 with the previous tiles in LDS: twenty-five)
                hipLaunchKernelGGL((k_kde_split_lds<KS_FOLD + 4, 2, KDE_LDS_F4>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (fold)                  // (33..61: twenty-five instead of twenty-seven)
                hipLaunchKernelGGL((k_kde_split<KS_FOLD + 4, 1, true, 0>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (prev->tmin && NCH == 1)      // (14..16 parameters, tiles in the order of the norm tops: eight MFMAs instead of nine)
                hipLaunchKernelGGL((k_kde_split<KS_TOPN + 1, 3, true, 1>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (prev->tmin && NCH == 2)      // (30..32: fourteen instead of fifteen)
                hipLaunchKernelGGL((k_kde_split<KS_TOPN + 2, 2, false, 0>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (prev->tmin && NCH == 3)      // (46..48: twenty)
                hipLaunchKernelGGL((k_kde_split_lds<KS_TOPN + 3, KDE_LDS_W3, KDE_LDS_F4>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (prev->tmin && kde_lds)       // (62..64: twenty-six)
                hipLaunchKernelGGL((k_kde_split_lds<KS_TOPN + 4, 2, KDE_LDS_F4>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (prev->tmin)                  // (62..64: twenty-six instead of twenty-seven)
                hipLaunchKernelGGL((k_kde_split<KS_TOPN + 4, 1, true, 0>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (NCH == 1)
                hipLaunchKernelGGL((k_kde_split<1, 3, true, 1>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (NCH == 2)
                hipLaunchKernelGGL((k_kde_split<2, 2>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (NCH == 3)
                hipLaunchKernelGGL((k_kde_split_lds<3, KDE_LDS_W3, KDE_LDS_F4>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else if (kde_lds)
                hipLaunchKernelGGL((k_kde_split_lds<4, 2, KDE_LDS_F4>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
            else
                hipLaunchKernelGGL((k_kde_split<4, 1>), dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream,
                                   (const uint4*)at, kn, (const uint4*)bt, (unsigned)nbt, wc, (const int*)ha_int, part, (const uint2*)prev->tmin);
        }
        if (PP > 64)
            hipLaunchKernelGGL(k_kde_gen, dim3((unsigned)rb, (unsigned)slices), dim3(256), 0, ctx->stream, a, kn, b, Kp, hb, wc, theta, K, k0,
                               theta_prev, part, PP, 0, (int)P, w_prev);
        else if (split) {       // far rows + the fp64 kernel as the split kernel's stand-in: one launch, returns at once when not needed
#define LAUNCH_FIX(PPV)                                                                                                       \
    hipLaunchKernelGGL(k_kde_fixups<PPV>, dim3((unsigned)(nat + rb + rb * slices)), dim3(256), 0, ctx->stream, a, kn, b, Kp, hb, wc, \
                       far_flag, far_list, (unsigned)nat, (unsigned)rb, (unsigned)slices, fix_i, fix_j, theta, K, k0, theta_prev, part)
            switch (PP) {
                case 8: LAUNCH_FIX(8); break;
                case 16: LAUNCH_FIX(16); break;
                case 32: LAUNCH_FIX(32); break;
                default: LAUNCH_FIX(64); break;
            }
#undef LAUNCH_FIX
        } else
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
    // the whole set in one call (k0 = 0, kn = K) and a caller that normalises next: the sum of squares comes out of k_wfinish
    const unsigned fb = (unsigned)((kn + 63) / 64);
    double* sq_part = nullptr;
    if (sumsq_out && k0 == 0 && kn == K) {
        sq_part = (double*)abc_ws_alloc(ctx, ((size_t)fb + 1) * sizeof(double));
        if (!sq_part) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
        *sumsq_out = sq_part;                 // the partials (launch_normalize_l2 sums them; slot fb is scratch for the total)
    }
    hipLaunchKernelGGL(k_wfinish, dim3(fb), dim3(256), 0, ctx->stream, priors, theta, K, (int)P, k0, kn, part,
                       (int)slices, wc, w_raw, split ? 1 : 0, epan ? 1 : 0, ctx->kde_which, far_flag, fix_i, fix_j, ha_frac, sq_part);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// the context's device word: which weight kernel ran last (abc_kde_last_kernel)
int abc_kde_words(abc_ctx* ctx) {
    if (ctx->kde_which) return ABC_OK;
    ABC_HIP(ctx, hipMalloc((void**)&ctx->kde_which, 2 * sizeof(int)));
    ABC_HIP(ctx, hipMemsetAsync(ctx->kde_which, 0, 2 * sizeof(int), ctx->stream));
    return ABC_OK;
}

int launch_fill(abc_ctx* ctx, double* w, size_t K, double v, hipStream_t st) {
    if (!K) return ABC_OK;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, st ? st : ctx->stream, w, K, v);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_normalize_l2(abc_ctx* ctx, double* w, size_t K, double* host_mirror, const double* sumsq_parts) {
    if (!K) return ABC_OK;
    StageTimer tm(ctx, ST_WEIGHTS_MISC);
    const unsigned fb = (unsigned)((K + 63) / 64);
    double* sq_part = const_cast<double*>(sumsq_parts);      // (k_wfinish's: fb partials + one scratch slot)
    if (!sq_part) {
        sq_part = (double*)abc_ws_alloc(ctx, ((size_t)fb + 1) * sizeof(double));
        if (!sq_part) ABC_FAIL(ctx, ABC_ERR_NOMEM, "normalize: workspace exhausted");
        hipLaunchKernelGGL(k_sumsq_partial, dim3(fb), dim3(256), 0, ctx->stream, w, K, sq_part);
    }
    const unsigned db = (unsigned)((K + 255) / 256);
    if (fb <= SQ_INLINE_MAX) {
        hipLaunchKernelGGL(k_div_norm, dim3(db), dim3(256), 0, ctx->stream, w, K, sq_part, fb, host_mirror);
    } else {
        hipLaunchKernelGGL(k_sumsq_total, dim3(1), dim3(256), 0, ctx->stream, sq_part, fb, sq_part + fb);
        hipLaunchKernelGGL(k_div_norm, dim3(db), dim3(256), 0, ctx->stream, w, K, sq_part + fb, 0u, host_mirror);
    }
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
