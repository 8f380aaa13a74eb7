// Resampling of parent particles and perturbation into the next set.
// Replaces ABC::gsl_rng_nonuniform_int / sample_posterior (AbcUtil.cpp:111-120, 366-375),
// sample_mvn_predictive_priors + gsl_ran_trunc_mv_normal (AbcUtil.cpp:391-404, 122-143),
// sample_predictive_priors + gsl_ran_trunc_normal + Prior::noise (AbcUtil.cpp:377-389, 145-158;
// Priors.h:19-43) and the per-particle seed draw (AbcSmc.cpp:535).
//
// Parent indices are BIT-EXACT with the reference stream: the reference consumes exactly one
// gsl_rng_taus2 output per draw (gsl_ran_discrete), and taus2 is linear over GF(2), so draw i's
// output is obtained by an O(log i) jump-ahead (32x32 bit-matrix powers); every lane regenerates a
// 64-output run of the sequential stream.  The Walker alias table is built on the host exactly as
// gsl_ran_discrete_preproc does (a data-dependent LIFO-stack algorithm, inherently serial).
// The Gaussian noise uses a counter-based Philox4x32-10 stream keyed by (rng state, draw index,
// attempt): same distribution as the reference's polar Box-Muller on taus2, not the same numbers
// (the reference's rejection loops consume a data-dependent number of outputs per particle).
#include <chrono>
#include <vector>

#include "abc_internal.h"
#include "refstream_host.h"

// ------------------------------------------------------------------------------------------------
// taus2 (host): [GSL] rng/taus.c
// ------------------------------------------------------------------------------------------------
static inline uint32_t taus_c1(uint32_t s) { return ((s & 4294967294u) << 12) ^ (((s << 13) ^ s) >> 19); }
static inline uint32_t taus_c2(uint32_t s) { return ((s & 4294967288u) << 4) ^ (((s << 2) ^ s) >> 25); }
static inline uint32_t taus_c3(uint32_t s) { return ((s & 4294967280u) << 17) ^ (((s << 3) ^ s) >> 11); }

uint32_t taus2_get(abc_rng* r) {
    r->s1 = taus_c1(r->s1); r->s2 = taus_c2(r->s2); r->s3 = taus_c3(r->s3);
    return r->s1 ^ r->s2 ^ r->s3;
}
void taus2_set(abc_rng* r, unsigned long seed) {
    uint32_t s = (uint32_t)(seed & 0xffffffffUL);
    if (s == 0) s = 1;
    r->s1 = 69069u * s;     if (r->s1 < 2) r->s1 += 2;
    r->s2 = 69069u * r->s1; if (r->s2 < 8) r->s2 += 8;
    r->s3 = 69069u * r->s2; if (r->s3 < 16) r->s3 += 16;
    for (int i = 0; i < 6; i++) taus2_get(r);
}

// 32x32 bit matrices over GF(2), stored as 32 column words: M * e_i = col[i]
struct BitMat { uint32_t col[32]; };
static inline uint32_t bm_apply(const BitMat& m, uint32_t v) {
    uint32_t y = 0;
    for (int i = 0; i < 32; i++) y ^= (uint32_t)(-(int32_t)((v >> i) & 1u)) & m.col[i];
    return y;
}
static inline BitMat bm_mul(const BitMat& a, const BitMat& b) {   // a*b
    BitMat c;
    for (int i = 0; i < 32; i++) c.col[i] = bm_apply(a, b.col[i]);
    return c;
}
static void taus_step_mats(BitMat m[3]) {
    for (int i = 0; i < 32; i++) {
        m[0].col[i] = taus_c1(1u << i); m[1].col[i] = taus_c2(1u << i); m[2].col[i] = taus_c3(1u << i);
    }
}
// pow2[k][c] = T_c^(2^k), k = 0..63 (built once; initialisation of a function-local static is thread-safe)
struct TausPow2 {
    BitMat tab[64][3];
    TausPow2() {
        taus_step_mats(tab[0]);
        for (int k = 1; k < 64; k++)
            for (int c = 0; c < 3; c++) tab[k][c] = bm_mul(tab[k - 1][c], tab[k - 1][c]);
    }
};
static const BitMat (*taus_pow2())[3] {
    static const TausPow2 t;
    return t.tab;
}
void taus2_jump(abc_rng* r, uint64_t n) {
    const BitMat(*tab)[3] = taus_pow2();
    for (int k = 0; k < 64; k++)
        if ((n >> k) & 1ull) {
            r->s1 = bm_apply(tab[k][0], r->s1); r->s2 = bm_apply(tab[k][1], r->s2); r->s3 = bm_apply(tab[k][2], r->s3);
        }
}

namespace {

constexpr int RUN = 64;        // consecutive stream outputs regenerated per lane
constexpr int RUN_LOG2 = 6;

// device table: jt[k][c][32] = T_c^(RUN * 2^k), k = 0..31
__device__ __forceinline__ uint32_t d_apply(const uint32_t* __restrict__ col, uint32_t v) {
    uint32_t y = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) y ^= (uint32_t)(-(int32_t)((v >> i) & 1u)) & col[i];
    return y;
}

// out[i] = output #(i) of the taus2 stream starting at state `base` (i.e. the value the (i+1)-th
// gsl_rng_get would return), for i in [0, n)
__global__ __launch_bounds__(256) void k_taus_stream(abc_rng base, size_t n, const uint32_t* __restrict__ jt,
                                                     uint32_t* __restrict__ out) {
    __shared__ uint32_t sjt[24 * 96];          // T^(RUN*2^k), k < 24 (16 M lanes x 64 outputs = 2^30 draws)
    for (int e = threadIdx.x; e < 24 * 96; e += blockDim.x) sjt[e] = jt[e];
    __syncthreads();
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t first = t * RUN;
    if (first >= n) return;
    uint32_t s1 = base.s1, s2 = base.s2, s3 = base.s3;
    for (int k = 0; k < 32; k++)
        if ((t >> k) & 1) {
            const uint32_t* m = (k < 24) ? sjt + k * 96 : jt + (size_t)k * 96;
            s1 = d_apply(m, s1); s2 = d_apply(m + 32, s2); s3 = d_apply(m + 64, s3);
        }
    const size_t last = (first + RUN < n) ? first + RUN : n;
    for (size_t i = first; i < last; i++) {
        s1 = ((s1 & 4294967294u) << 12) ^ (((s1 << 13) ^ s1) >> 19);
        s2 = ((s2 & 4294967288u) << 4) ^ (((s2 << 2) ^ s2) >> 25);
        s3 = ((s3 & 4294967280u) << 17) ^ (((s3 << 3) ^ s3) >> 11);
        out[i] = s1 ^ s2 ^ s3;
    }
}

// [GSL] gsl_ran_discrete (KNUTH_CONVENTION): u = get/2^32; c = floor(u*K); F[c]==1 ? c : (u<F[c] ? c : A[c]), with F[k] the
// preprocessed (F[k] + k) / K: the host leaves that last pass out (alias_preproc, knuth = false) and the two operations run
// here on the one entry a draw reads -- the same IEEE addition and division, hence the same bits.
__global__ __launch_bounds__(256) void k_alias_draw(const uint32_t* __restrict__ raw, size_t n,
                                                    const double* __restrict__ F, const uint32_t* __restrict__ A,
                                                    size_t K, unsigned long long* __restrict__ parent,
                                                    const int* __restrict__ verdict_src = nullptr, int* __restrict__ verdict_dev = nullptr,
                                                    int* __restrict__ verdict_pin = nullptr) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && verdict_src) { const int v = *verdict_src; *verdict_dev = v; *verdict_pin = v; }      // the device build's verdict (alias_dev.hip)
    if (i >= n) return;
    const double u = (double)raw[i] / 4294967296.0;
    const size_t c = (size_t)(u * (double)K);
    const double f = (F[c] + (double)c) / (double)K;
    parent[i] = (f == 1.0) ? c : ((u < f) ? c : (size_t)A[c]);
}

__global__ __launch_bounds__(256) void k_widen(const uint32_t* __restrict__ raw, size_t n,
                                               unsigned long long* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = raw[i];
}

// ---- Philox4x32-10 --------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c;
}
// FOUR independent N(0,1) from one Philox block: two Box-Muller pairs on the f32 transcendental hardware (v_log_f32,
// v_sqrt_f32, v_sin_f32 / v_cos_f32, 8 issue cycles each) instead of double-precision library calls -- the fp64 log, sqrt and
// sincospi of round 1 were ~300 vector instructions per pair and made the noise kernels compute-bound (71 us for 1.6e7
// deviates); this is ~20 per pair plus half a Philox block.  The radius comes from all 32 bits of its word,
//   -2 ln u = -2 ln 2 (log2(r + 1/2) - 32),   u in [2^-33, 1):  |z| <= 6.76,
// (the conversion of r to f32 rounds at 6e-8 relative: 9e-8 absolute in the logarithm), the angle from the top 24 bits of
// its word, in revolutions (what v_sin_f32 / v_cos_f32 take).  The deviates carry f32 rounding (~1e-7 relative): the device
// noise stream is distributional by contract (declared deviations); the reference-stream mode is untouched.
__device__ __forceinline__ void normal4(U4 r, double (&z)[4]) {
    const float NEG2LN2 = -1.3862943611198906f;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const uint32_t ru = h ? r.z : r.x, ra = h ? r.w : r.y;
        const float lg = __builtin_amdgcn_logf((float)ru + 0.5f) - 32.0f;           // log2 u, in [-33, 0)
        const float rad = __builtin_amdgcn_sqrtf(__builtin_fmaxf(NEG2LN2 * lg, 0.0f));   // (v_log_f32 may return 32 + 1 ulp at the top)
        const float ang = (float)(ra >> 8) * 5.9604644775390625e-08f;             // [0, 1) revolutions, exact
        z[2 * h] = (double)(rad * __builtin_amdgcn_cosf(ang));
        z[2 * h + 1] = (double)(rad * __builtin_amdgcn_sinf(ang));
    }
}

__device__ __forceinline__ double d_recast(const abc_prior& pr, double v) {          // Priors.h:58,80,106
    return (pr.kind == ABC_PRIOR_UNIF_INT) ? round(v) : v;
}
__device__ __forceinline__ bool d_valid(const abc_prior& pr, double v) {             // Parameter.h:77
    if (pr.kind == ABC_PRIOR_GAUSS) {
        const double u = (v - pr.a) / fabs(pr.b);
        // likelihood != 0.  For u^2 < 1400 and |sigma| < 1e10 the product is >= 4e-11 * exp(-700) = 3.9e-315 > 0 for
        // certain, so the exponential is only evaluated near the underflow edge (|u| > 37) and for NaN / huge sigma
        if (u * u < 1400.0 && fabs(pr.b) < 1e10) return true;
        return (1.0 / (sqrt(2.0 * M_PI) * fabs(pr.b))) * exp(-u * u / 2.0) != 0.0;
    }
    if (pr.kind == ABC_PRIOR_UNIF_INT) return (v == round(v)) && (pr.a <= v) && (v <= pr.b);
    return (pr.a <= v) && (v <= pr.b);
}
// recast + support test of one coordinate in the proposal loop.  The prior of a coordinate is the same in every lane: its
// kind goes through an SGPR, so only the code of that kind runs (scalar branches), and the Gaussian support test
// (likelihood != 0, i.e. |u| below ~38.6) multiplies by 1 / |sigma| first: t^2 < 1399 implies u^2 < 1400 for u = (v - a) / |sigma|
// (the two differ by rounding only), which d_valid accepts without evaluating anything; the exact test runs at the edge.
__device__ __forceinline__ double recast_valid(const abc_prior& pr, double inv_sigma, double v, bool& ok) {
    const int kind = __builtin_amdgcn_readfirstlane(pr.kind);
    if (kind == ABC_PRIOR_GAUSS) {
        const double t = (v - pr.a) * inv_sigma;
        if (!(t * t < 1399.0 && inv_sigma != 0.0)) ok = ok && d_valid(pr, v);
        return v;
    }
    if (kind == ABC_PRIOR_UNIF_INT) v = round(v);               // Priors.h:80; v == round(v) then holds by construction
    ok = ok && (pr.a <= v) && (v <= pr.b);
    return v;
}
__device__ __forceinline__ double d_prior_mean(const abc_prior& pr) {                 // Priors.h:35
    return (pr.kind == ABC_PRIOR_GAUSS) ? pr.a : (pr.b + pr.a) / 2.0;
}

// a proposal the perturbation gives up on: counted on the device, and -- rare path -- a flag word in the context's pinned block is
// raised (its address travels in slot 2 of the counter array), so that the host learns of give-ups at its next synchronisation
// without a kernel that copies the counter behind every generation
__device__ __forceinline__ void note_giveup(unsigned long long* __restrict__ g) {
    atomicAdd(g, 1ull);
    unsigned* f = (unsigned*)(size_t)g[2];
    if (f) *f = 1u;
}

constexpr unsigned MVN_MAX_TRIES = 1u << 14;   // the reference retries for ever (AbcUtil.cpp:132); bounded here

// theta (K x P column-major) -> row-major K x PP, zero padded: a parent row is then one contiguous PP*8-byte line
// instead of P strided 8-byte reads that each pull a whole 64-byte sector (PMC: 1.2 GB fetched for 128 MB used).
// Block 0 also pads the Cholesky factor (P x P, lower) to Lpad (PP x PP, column-major, zero above the diagonal and in the
// padding): k_perturb streams it through the scalar cache.
template <int PP>
__global__ __launch_bounds__(256) void k_theta_rows(const double* __restrict__ theta, size_t K, int P,
                                                    double* __restrict__ rows, const double* __restrict__ L,
                                                    double* __restrict__ Lpad) {
    __shared__ double t[PP][65];
    if (blockIdx.x == 0 && L) {
        for (int e = threadIdx.x; e < PP * PP; e += 256) {
            const int a = e % PP, b = e / PP;
            Lpad[e] = (a < P && b < P && b <= a) ? L[a + (size_t)P * b] : 0.0;
        }
    }
    const size_t k0 = (size_t)blockIdx.x * 64;
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

// The Gaussian noise of one proposal attempt, L z (multivariate) -- one Philox block -> four normals -> four columns of L.
// L is lower triangular: the columns of its right half are zero in rows < PP/2, so their blocks only touch the lower half
// of x (a quarter of the FMAs).  sL: the padded factor (k_theta_rows), wave-uniform addresses: in k_perturb it lives in
// global memory and arrives through the scalar cache as SGPR operands of the FMAs (from LDS every FMA needed its own
// broadcast read: 192 LDS instructions per attempt at 16 parameters).
template <int PP, int A0>
__device__ __forceinline__ void mv_cols4(const double* __restrict__ l, const double (&z)[4], double (&x)[PP]) {
#pragma unroll
    for (int a = A0; a < PP; a++) {
        x[a] = fma(l[a], z[0], x[a]); x[a] = fma(l[PP + a], z[1], x[a]);
        x[a] = fma(l[2 * PP + a], z[2], x[a]); x[a] = fma(l[3 * PP + a], z[3], x[a]);
    }
}
template <int PP>
__device__ __forceinline__ void mv_noise(const double* __restrict__ sL, unsigned long long gi, unsigned attempt, uint32_t k0,
                                         uint32_t k1, double (&x)[PP], int P) {
    const int nq = (P + 3) / 4;                                  // column quads that hold anything
#pragma unroll
    for (int a = 0; a < PP; a++) x[a] = 0.0;
    if constexpr (PP < 4) {
        U4 c; c.x = (uint32_t)gi; c.y = (uint32_t)(gi >> 32); c.z = attempt; c.w = 0u;
        double z[4];
        normal4(philox(c, k0, k1), z);
#pragma unroll
        for (int b = 0; b < PP; b++)
#pragma unroll
            for (int a = 0; a < PP; a++) x[a] = fma(sL[PP * b + a], z[b], x[a]);
    } else {
        // columns 4 qd .. 4 qd + 3 of L per Philox block.  L is lower triangular: a column block only touches the rows from its own
        // first row on -- taken in FOUR row groups from 17 parameters (round 6; two until then, and still at up to 16, where four
        // cost a wave per SIMD in registers: 5/8 instead of 3/4 of the square's FMAs), the left-out products are exact zeros -- and the padded columns (from P on) hold nothing: their blocks are not drawn
#pragma unroll 1
        for (int qd = 0; qd < nq; qd++) {
            U4 c; c.x = (uint32_t)gi; c.y = (uint32_t)(gi >> 32); c.z = attempt; c.w = (uint32_t)qd;
            double z[4];
            normal4(philox(c, k0, k1), z);
            const double* l = sL + PP * (4 * qd);
            if constexpr (PP >= 32) {
                const int g = (4 * qd) / (PP / 4);                   // (wave-uniform)
                if (g == 0) mv_cols4<PP, 0>(l, z, x);
                else if (g == 1) mv_cols4<PP, PP / 4>(l, z, x);
                else if (g == 2) mv_cols4<PP, PP / 2>(l, z, x);
                else mv_cols4<PP, 3 * (PP / 4)>(l, z, x);
            } else {
                if (qd < (PP / 4 + 1) / 2) mv_cols4<PP, 0>(l, z, x);
                else mv_cols4<PP, PP / 2>(l, z, x);
            }
        }
    }
}
// independent noise of coordinate p: sqrt(dv_p) z
__device__ __forceinline__ double indep_noise(double sigma, unsigned long long gi, unsigned attempt, int p, uint32_t k0, uint32_t k1) {
    U4 c; c.x = (uint32_t)gi; c.y = (uint32_t)(gi >> 32); c.z = attempt; c.w = 0x80000000u | (uint32_t)p;
    double z[4];
    normal4(philox(c, k0, k1), z);
    return sigma * z[0];
}
// one new particle per lane
template <int PP, bool MV>
__global__ __launch_bounds__(256, (MV && PP <= 16) ? 3 : (PP <= 16 ? 2 : 1)) void k_perturb(abc_rng key, const double* __restrict__ theta, size_t K, int P,
                                                 const abc_prior* __restrict__ priors,
                                                 const unsigned long long* __restrict__ parent,
                                                 unsigned long long i0, size_t n,
                                                 const double* __restrict__ L_or_dv /* MV: the padded factor */, double* __restrict__ out,
                                                 unsigned long long* __restrict__ giveups) {
    __shared__ double sS[PP];          // independent mode: sqrt(dv) (AbcUtil.cpp:150)
    __shared__ abc_prior sp[PP];
    __shared__ double sinv[PP];        // Gaussian priors: 1 / |sigma| for the fast side of the support test (0: always the exact test)
    if (!MV) for (int p = threadIdx.x; p < PP; p += 256) sS[p] = (p < P) ? sqrt(L_or_dv[p]) : 0.0;
    for (int p = threadIdx.x; p < PP; p += 256) {
        abc_prior q; q.kind = ABC_PRIOR_UNIF_REAL; q.pad_ = 0; q.a = -1e300; q.b = 1e300;
        sp[p] = (p < P) ? priors[p] : q;
        sinv[p] = (p < P && priors[p].kind == ABC_PRIOR_GAUSS && fabs(priors[p].b) < 1e10 && fabs(priors[p].b) > 1e-300) ? 1.0 / fabs(priors[p].b) : 0.0;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    {
    const unsigned long long gi = i0 + i;
    const double* mrow = theta + (size_t)parent[i] * PP;   // theta: row-major K x PP (k_theta_rows), read with 16-byte loads
    const uint32_t k0 = key.s1 ^ 0x5bd1e995u, k1 = key.s2 ^ (key.s3 * 0x9E3779B1u);
    // Only the noise / proposal vector x[PP] lives in registers across the attempt loop: the parent row is (re-)read where it is
    // added (a cached 128-byte line), so the kernel stays near 110 VGPRs (four waves per SIMD) instead of 260 with the row,
    // the noise and the candidate all resident (one wave per SIMD: 141 us at N+ = 1e6, latency-bound on the parent gather).
    double x[PP];
    if (MV) {
        // AbcUtil.cpp:132-139: draw the whole vector x = mu + L z, accept iff every coordinate is valid
        for (unsigned attempt = 0; attempt < MVN_MAX_TRIES; attempt++) {
            mv_noise<PP>(L_or_dv, gi, attempt, k0, k1, x, P);
            bool ok = true;
            asm volatile("" ::: "memory");      // the prior table is re-read from LDS here (hoisted out of the loop it costs 96 VGPRs)
#pragma unroll
            for (int a = 0; a < PP; a += 2) {
                if (PP > 32 && a >= P) continue;      // (33..64 parameters share the 64-wide kernel: nothing to add, test or store from P on)
                const double2 m = *reinterpret_cast<const double2*>(mrow + a);
                x[a] = recast_valid(sp[a], sinv[a], x[a] + m.x, ok);
                x[a + 1] = recast_valid(sp[a + 1], sinv[a + 1], x[a + 1] + m.y, ok);
            }
            if (ok) break;
            if (attempt + 1 == MVN_MAX_TRIES) {
#pragma unroll
                for (int a = 0; a < PP; a++) x[a] = mrow[a];   // give up: keep the (valid) parent, and say so (abc_perturb_giveups)
                note_giveup(giveups);
            }
        }
    } else {
        // Priors.h:19-33: per coordinate, up to 1000 tries, then the prior mean
#pragma unroll
        for (int p = 0; p < PP; p++) {
            x[p] = 0.0;
            if (p < P) {
                const double m = mrow[p];
                double v = 0.0; bool ok = false;
                asm volatile("" ::: "memory");          // (as above: this coordinate's prior is read from LDS here, not hoisted)
                for (unsigned attempt = 0; attempt < 1000 && !ok; attempt++) {
                    v = d_recast(sp[p], indep_noise(sS[p], gi, attempt, p, k0, k1) + m);
                    ok = d_valid(sp[p], v);
                }
                x[p] = ok ? v : d_prior_mean(sp[p]);
                if (!ok) note_giveup(giveups);       // the reference prints an error line per fallback (Priors.h:27-29)
            }
        }
    }
#pragma unroll
    for (int p = 0; p < PP; p++)
        if (p < P) __builtin_nontemporal_store(x[p], &out[i + n * (size_t)p]);   // read next by the host / simulators, not by a kernel:
                                                                                  // keep the 8 P N bytes out of L2 / Infinity Cache
    }
}

// 32 < P <= 64, INDEPENDENT noise (multivariate noise runs on k_perturb<64> since round 6: this kernel's multivariate branch kept the
// factor in LDS -- one broadcast read per FMA, the full 64 x 64 square, 255 registers, one wave per SIMD -- and took 1.1-2.5 ms for
// 1e6 proposals at 40-64 parameters where k_perturb<64> takes 0.39-1.05; the draws, their order and the acceptance rule are the same,
// the proposals bit-identical): same draws and acceptance rule as k_perturb; the parent row is re-read (one cached line per 16
// coordinates) and accepted coordinates are stored as they are produced -- a rejected attempt is simply overwritten by the next one.
template <int PP, bool MV>
__global__ __launch_bounds__(256) void k_perturb_stream(abc_rng key, const double* __restrict__ theta, size_t K, int P,
                                                        const abc_prior* __restrict__ priors,
                                                        const unsigned long long* __restrict__ parent,
                                                        unsigned long long i0, size_t n,
                                                        const double* __restrict__ L_or_dv, double* __restrict__ out,
                                                        unsigned long long* __restrict__ giveups) {
    extern __shared__ double smem[];
    double* sL = smem;                                  // MV: PP x PP, column-major, zero above the diagonal
    abc_prior* sp = reinterpret_cast<abc_prior*>(smem + (MV ? PP * PP : PP));
    if (MV) {
        for (int e = threadIdx.x; e < PP * PP; e += 256) {
            const int a = e % PP, b = e / PP;
            sL[e] = (a < P && b < P && b <= a) ? L_or_dv[a + (size_t)P * b] : 0.0;
        }
    } else {
        for (int e = threadIdx.x; e < PP; e += 256) sL[e] = (e < P) ? sqrt(L_or_dv[e]) : 0.0;
    }
    for (int p = threadIdx.x; p < PP; p += 256) {
        abc_prior q; q.kind = ABC_PRIOR_UNIF_REAL; q.pad_ = 0; q.a = -1e300; q.b = 1e300;
        sp[p] = (p < P) ? priors[p] : q;
    }
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long gi = i0 + i;
    const double* mu = theta + (size_t)parent[i] * PP;
    const uint32_t k0 = key.s1 ^ 0x5bd1e995u, k1 = key.s2 ^ (key.s3 * 0x9E3779B1u);
    if (MV) {
        bool ok = false;
        for (unsigned attempt = 0; attempt < MVN_MAX_TRIES && !ok; attempt++) {
            double x[PP];
#pragma unroll
            for (int a = 0; a < PP; a++) x[a] = 0.0;
#pragma unroll 1
            for (int qd = 0; qd < PP / 4; qd++) {
                U4 c; c.x = (uint32_t)gi; c.y = (uint32_t)(gi >> 32); c.z = attempt; c.w = (uint32_t)qd;
                double z[4];
                normal4(philox(c, k0, k1), z);
                const double* l = sL + PP * (4 * qd);
#pragma unroll
                for (int a = 0; a < PP; a++) {
                    x[a] = fma(l[a], z[0], x[a]); x[a] = fma(l[PP + a], z[1], x[a]);
                    x[a] = fma(l[2 * PP + a], z[2], x[a]); x[a] = fma(l[3 * PP + a], z[3], x[a]);
                }
            }
            ok = true;
#pragma unroll
            for (int a = 0; a < PP; a++) {
                if (a < P) {
                    const double v = d_recast(sp[a], x[a] + mu[a]);
                    out[i + n * (size_t)a] = v;
                    ok = ok && d_valid(sp[a], v);
                }
            }
        }
        if (!ok) {
            for (int p = 0; p < P; p++) out[i + n * (size_t)p] = mu[p];      // give up: keep the (valid) parent
            note_giveup(giveups);
        }
    } else {
        for (int p = 0; p < P; p++) {
            const double m = mu[p];
            double v = 0.0; bool ok = false;
            for (unsigned attempt = 0; attempt < 1000 && !ok; attempt++) {
                U4 c; c.x = (uint32_t)gi; c.y = (uint32_t)(gi >> 32); c.z = attempt; c.w = 0x80000000u | (uint32_t)p;
                double z[4];
                normal4(philox(c, k0, k1), z);
                v = d_recast(sp[p], sL[p] * z[0] + m);
                ok = d_valid(sp[p], v);
            }
            out[i + n * (size_t)p] = ok ? v : d_prior_mean(sp[p]);
            if (!ok) note_giveup(giveups);
        }
    }
}

// ---- more than 64 parameters (the reference's loops have no size limit): runtime width, nothing resident across chunks -------
// theta -> row-major K x PP (PP a multiple of 64) and, block 0, the padded factor
__global__ __launch_bounds__(256) void k_theta_rows_gen(const double* __restrict__ theta, size_t K, int P, int PP,
                                                        double* __restrict__ rows, const double* __restrict__ L,
                                                        double* __restrict__ Lpad) {
    if (blockIdx.x == 0 && L) {
        for (size_t e = threadIdx.x; e < (size_t)PP * PP; e += 256) {
            const int a = (int)(e % PP), b = (int)(e / PP);
            Lpad[e] = (a < P && b < P && b <= a) ? L[a + (size_t)P * b] : 0.0;
        }
    }
    const size_t tot = K * (size_t)PP;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (size_t)gridDim.x * 256) {
        const size_t r = e / PP;
        const int p = (int)(e % PP);
        rows[e] = (p < P) ? theta[r + K * (size_t)p] : 0.0;
    }
}
// Same draws and acceptance rule as k_perturb (counter (row, attempt, column quad) -> four normals), in chunks of 64
// coordinates: x = L z for rows 64 c .. 64 c + 63 needs the normals of columns 0 .. 64 c + 63, which are REGENERATED per chunk
// (they are a pure function of the counter); accepted coordinates are stored as they are produced, a rejected attempt is
// overwritten by the next one.  The factor comes through the scalar cache.
template <bool MV>
__global__ __launch_bounds__(256) void k_perturb_gen(abc_rng key, const double* __restrict__ rows, int P, int PP,
                                                     const abc_prior* __restrict__ priors,
                                                     const unsigned long long* __restrict__ parent,
                                                     unsigned long long i0, size_t n, const double* __restrict__ L_or_dv,
                                                     double* __restrict__ out, unsigned long long* __restrict__ giveups) {
    extern __shared__ abc_prior spg[];          // P priors
    for (int p = threadIdx.x; p < P; p += 256) spg[p] = priors[p];
    __syncthreads();
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long gi = i0 + i;
    const double* mu = rows + (size_t)parent[i] * PP;
    const uint32_t k0 = key.s1 ^ 0x5bd1e995u, k1 = key.s2 ^ (key.s3 * 0x9E3779B1u);
    if (MV) {
        bool ok = false;
        for (unsigned attempt = 0; attempt < MVN_MAX_TRIES && !ok; attempt++) {
            ok = true;
            for (int c = 0; c < PP / 64; c++) {
                double x[64];
#pragma unroll
                for (int a = 0; a < 64; a++) x[a] = 0.0;
                for (int qd = 0; qd < 16 * (c + 1); qd++) {          // columns 4 qd .. 4 qd + 3 <= the chunk's last row
                    U4 cn; cn.x = (uint32_t)gi; cn.y = (uint32_t)(gi >> 32); cn.z = attempt; cn.w = (uint32_t)qd;
                    double z[4];
                    normal4(philox(cn, k0, k1), z);
                    const double* l = L_or_dv + (size_t)PP * (4 * qd) + 64 * c;
#pragma unroll
                    for (int a = 0; a < 64; a++) {
                        x[a] = fma(l[a], z[0], x[a]); x[a] = fma(l[PP + a], z[1], x[a]);
                        x[a] = fma(l[2 * (size_t)PP + a], z[2], x[a]); x[a] = fma(l[3 * (size_t)PP + a], z[3], x[a]);
                    }
                }
#pragma unroll
                for (int a = 0; a < 64; a++) {
                    const int p = 64 * c + a;
                    if (p < P) {
                        const double v = d_recast(spg[p], x[a] + mu[p]);
                        out[i + n * (size_t)p] = v;
                        ok = ok && d_valid(spg[p], v);
                    }
                }
            }
        }
        if (!ok) {
            for (int p = 0; p < P; p++) out[i + n * (size_t)p] = mu[p];      // give up: keep the (valid) parent
            note_giveup(giveups);
        }
    } else {
        for (int p = 0; p < P; p++) {
            const double m = mu[p], sg = sqrt(L_or_dv[p]);
            double v = 0.0; bool ok = false;
            for (unsigned attempt = 0; attempt < 1000 && !ok; attempt++) {
                v = d_recast(spg[p], indep_noise(sg, gi, attempt, p, k0, k1) + m);
                ok = d_valid(spg[p], v);
            }
            out[i + n * (size_t)p] = ok ? v : d_prior_mean(spg[p]);
            if (!ok) note_giveup(giveups);
        }
    }
}

int ensure_jump_tab(abc_ctx* ctx) {
    if (ctx->jump_tab) return ABC_OK;
    const BitMat(*tab)[3] = taus_pow2();
    std::vector<uint32_t> h(32 * 96);
    for (int k = 0; k < 32; k++)
        for (int c = 0; c < 3; c++)
            for (int i = 0; i < 32; i++) h[(size_t)k * 96 + c * 32 + i] = tab[k + RUN_LOG2][c].col[i];
    ABC_HIP(ctx, hipMalloc((void**)&ctx->jump_tab, h.size() * sizeof(uint32_t)));
    ABC_HIP(ctx, hipMemcpy(ctx->jump_tab, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return ABC_OK;
}

int taus_stream(abc_ctx* ctx, abc_rng base, size_t n, uint32_t* out, hipStream_t st = nullptr) {
    ABC_TRY(ensure_jump_tab(ctx));
    const size_t threads = (n + RUN - 1) / RUN;
    // every thread pays ~3000 instructions (its jump ahead, then 64 outputs): work-groups of ONE wave while they do not fill the
    // chip four waves each (1e6 outputs: 62 groups of four waves on 62 of 256 CUs took 27 us; 245 single waves take a third)
    const unsigned bs = (threads <= (size_t)256 * 64 * 4) ? 64u : 256u;
    hipLaunchKernelGGL(k_taus_stream, dim3((unsigned)((threads + bs - 1) / bs)), dim3(bs), 0, st ? st : ctx->stream, base, n,
                       ctx->jump_tab, out);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// the serial Walker build lives in alias_host.{h,cpp} (plain C++, also compiled and checked by tests/cxx/alias_probe.cpp)

}  // namespace

int abc_uniform_alias(abc_ctx* ctx, size_t K) {
    if (K == 0 || K > 0xffffffffull) ABC_FAIL(ctx, ABC_ERR_INVALID, "resample: K = %zu", K);
    if (ctx->ualias_K == K) return ABC_OK;
    ctx->ualias_K = 0;
    if (ctx->ualias_cap < K) {
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ualias_F) { (void)hipFree(ctx->ualias_F); ctx->ualias_F = nullptr; }
        if (ctx->ualias_pin) { (void)hipHostFree(ctx->ualias_pin); ctx->ualias_pin = nullptr; }
        ctx->ualias_cap = 0;
        ABC_HIP(ctx, hipMalloc((void**)&ctx->ualias_F, K * (sizeof(double) + sizeof(uint32_t))));
        ABC_HIP(ctx, hipHostMalloc((void**)&ctx->ualias_pin, K * (sizeof(double) + sizeof(uint32_t)), hipHostMallocDefault));
        ctx->ualias_cap = K;
    }
    ABC_TRY(abc_pin_reserve(ctx, K * (sizeof(double) * 2 + sizeof(uint32_t) * 2) + 2 * sizeof(uint32_t)));
    double* hw = (double*)ctx->pin;                        // host-only scratch of the build
    double* hE = hw + K;
    uint32_t* hS = (uint32_t*)(hE + K);
    uint32_t* hB = hS + K + 1;
    double* hF = (double*)ctx->ualias_pin;                 // the tables, F[K] then A[K]: uploaded from their own staging buffer
    uint32_t* hA = (uint32_t*)(hF + K);
    const double v = 1.0 / (double)K;                      // what launch_fill writes (AbcUtil.cpp:543-544)
    for (size_t k = 0; k < K; k++) hw[k] = v;
    const auto t0 = std::chrono::steady_clock::now();
    abc_alias_preproc(K, hw, hF, hA, hE, hS, hB, /*knuth=*/false);
    if (ctx->timing) {
        ctx->stage_host_ms[ST_ALIAS_HOST] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        ctx->stage_cnt[ST_ALIAS_HOST] += 1;
    }
    ctx->ualias_A = (uint32_t*)(ctx->ualias_F + K);
    ABC_HIP(ctx, hipMemcpyAsync(ctx->ualias_F, hF, K * (sizeof(double) + sizeof(uint32_t)), hipMemcpyHostToDevice, ctx->stream));
    ctx->ualias_K = K;
    return ABC_OK;
}

static int launch_seeds(abc_ctx* ctx, const abc_rng* rng, uint64_t i0, size_t n, uint64_t* seeds, uint64_t seed_stream_offset,
                        hipStream_t st);

// The point of the main stream the side stream's work of this generation starts behind (the end of the Gram kernel): recorded
// once; the early launchers below find it and do not record their own.  The fused drivers record it, queue the model fit on
// the main stream and only then spend host time on the side stream's launches, which catch up beside the (long, one-work-group) fit.
int abc_side_fork(abc_ctx* ctx) {
    if (!ctx->side) {
        ABC_HIP(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, abc_xstream_event_flags()));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_side, abc_xstream_event_flags()));
    }
    // The side stream must start behind everything queued on the main one so far.  When the main stream is idle -- the usual
    // case at the start of a generation: the previous call ended with a synchronisation -- that holds without an event, and the
    // record + wait pair (~15 us of host time in front of the generation's first launch, the GPU idle meanwhile) is skipped.
    static const int always = abc_diag_env("ABC_FORK_ALWAYS") ? 1 : 0;           // A/B switch for measurements
    if (always || hipStreamQuery(ctx->stream) != hipSuccess) {
        (void)hipGetLastError();                                           // (hipErrorNotReady is not an error here)
        ABC_HIP(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
        ABC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_fork, 0));
    }
    ctx->side_forked = true;
    return ABC_OK;
}

int abc_rng_streams_early(abc_ctx* ctx, const abc_rng* rng, uint64_t i0, size_t n, uint64_t* seeds, uint64_t seed_stream_offset,
                          uint32_t** raw_out, uint64_t* parent_uniform, size_t K_uniform) {
    *raw_out = nullptr;
    if (n == 0) return ABC_OK;
    uint32_t* raw = (uint32_t*)abc_ws_alloc(ctx, n * sizeof(uint32_t));
    if (!raw) ABC_FAIL(ctx, ABC_ERR_NOMEM, "resample: workspace exhausted");
    ABC_TRY(ensure_jump_tab(ctx));                       // (its first-use upload is synchronous)
    // the side stream starts behind everything queued on the main one up to the fork (earlier users of the seed buffer)
    if (!ctx->side_forked) ABC_TRY(abc_side_fork(ctx));
    abc_rng base = *rng;
    taus2_jump(&base, i0);
    ABC_TRY(taus_stream(ctx, base, n, raw, ctx->side));
    if (parent_uniform && ctx->ualias_K == K_uniform && K_uniform) {
        hipLaunchKernelGGL(k_alias_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->side, raw, n, ctx->ualias_F,
                           ctx->ualias_A, K_uniform, (unsigned long long*)parent_uniform);
        ABC_HIP(ctx, hipGetLastError());
    }
    if (seeds) ABC_TRY(launch_seeds(ctx, rng, i0, n, seeds, seed_stream_offset, ctx->side));
    ABC_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->side));
    *raw_out = raw;
    return ABC_OK;
}

// the simulator seeds alone, on the side stream BEHIND whatever is queued there (the fused driver queues them behind the previous
// set's prologue, which the main stream needs much earlier); ev_side is recorded again behind them
int abc_rng_seeds_early(abc_ctx* ctx, const abc_rng* rng, uint64_t i0, size_t n, uint64_t* seeds, uint64_t seed_stream_offset) {
    if (n == 0 || !seeds) return ABC_OK;
    if (!ctx->side_forked) ABC_TRY(abc_side_fork(ctx));
    ABC_TRY(launch_seeds(ctx, rng, i0, n, seeds, seed_stream_offset, ctx->side));
    ABC_HIP(ctx, hipEventRecord(ctx->ev_side, ctx->side));
    return ABC_OK;
}

// the previous set's share of the weight stage on the side stream (behind whatever abc_rng_streams_early queued there, or
// forked here); the main stream waits for ev_prev before launch_weights_raw
int abc_weights_prev_early(abc_ctx* ctx, size_t P, size_t kn_max, const double* theta_prev, size_t Kp, const double* w_prev,
                           const double* dv_prev, abc_wprev* out) {
    if (!ctx->side_forked) ABC_TRY(abc_side_fork(ctx));
    if (!ctx->ev_prev) ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_prev, abc_xstream_event_flags()));
    ABC_TRY(launch_weights_prev(ctx, P, kn_max, theta_prev, Kp, w_prev, dv_prev, out, ctx->side));
    ABC_HIP(ctx, hipEventRecord(ctx->ev_prev, ctx->side));
    return ABC_OK;
}

int launch_resample(abc_ctx* ctx, const abc_rng* rng, const double* w, size_t K, uint64_t i0, size_t n,
                    uint64_t* parent, int (*while_host_builds)(void*), void* hook_arg, bool uniform_weights,
                    const uint32_t* raw_ready, bool weights_on_host, const volatile int* abort_flag, bool parents_ready,
                    int* alias_check_deferred) {
    if (alias_check_deferred) *alias_check_deferred = 0;
    bool wait_side = raw_ready != nullptr;               // raw_ready comes from the side stream (abc_rng_streams_early)
    if (n == 0) return ABC_OK;
    if (K == 0 || K > 0xffffffffull) ABC_FAIL(ctx, ABC_ERR_INVALID, "resample: K = %zu", K);
    if (uniform_weights) {
        ABC_TRY(abc_uniform_alias(ctx, K));               // (already built by the fused drivers; here for any other caller)
        uint32_t* raw = const_cast<uint32_t*>(raw_ready);
        if (!raw) raw = (uint32_t*)abc_ws_alloc(ctx, n * sizeof(uint32_t));
        if (!raw) ABC_FAIL(ctx, ABC_ERR_NOMEM, "resample: workspace exhausted");
        abc_rng base = *rng;
        taus2_jump(&base, i0);
        StageTimer tm(ctx, ST_RESAMPLE);
        if (raw_ready) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        else ABC_TRY(taus_stream(ctx, base, n, raw));
        if (while_host_builds) ABC_TRY(while_host_builds(hook_arg));
        if (!(parents_ready && raw_ready)) {
            hipLaunchKernelGGL(k_alias_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, raw, n, ctx->ualias_F,
                               ctx->ualias_A, K, (unsigned long long*)parent);
            ABC_HIP(ctx, hipGetLastError());
        }
        return ABC_OK;
    }
    if (ctx->alias_K < K) {       // the table's home in HBM: F[K] then A[K], one allocation
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->alias_F) { (void)hipFree(ctx->alias_F); ctx->alias_F = nullptr; ctx->alias_A = nullptr; }
        ABC_HIP(ctx, hipMalloc((void**)&ctx->alias_F, K * (sizeof(double) + sizeof(uint32_t))));
        ctx->alias_K = K;
    }
    ctx->alias_A = (uint32_t*)(ctx->alias_F + K);
    // ---- the table built on the device (alias_dev.hip): no copy of the weights to the host, no host wait ------------------------
    if (ctx->alias_mode == ABC_ALIAS_DEVICE && K >= ABC_ALIAS_DEV_MIN_K && K <= ABC_ALIAS_DEV_MAX_K && ctx->ws_off + abc_alias_dev_need(K) <= ctx->ws_bytes) {
        if (!ctx->alias_fail_dev) ABC_HIP(ctx, hipMalloc((void**)&ctx->alias_fail_dev, sizeof(int)));
        int* fail_pin = (int*)(ctx->status_pin + 44);
        *fail_pin = 0;
        const int* verdict = nullptr;
        uint32_t* raw = const_cast<uint32_t*>(raw_ready);
        if (!raw) raw = (uint32_t*)abc_ws_alloc(ctx, n * sizeof(uint32_t));
        if (!raw) ABC_FAIL(ctx, ABC_ERR_NOMEM, "resample: workspace exhausted");
        abc_rng base = *rng;
        taus2_jump(&base, i0);
        if (raw_ready) {
            if (!ctx->side_early_waited) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
        } else {
            StageTimer tm(ctx, ST_RESAMPLE);
            ABC_TRY(taus_stream(ctx, base, n, raw));
        }
        {
            StageTimer tm(ctx, ST_ALIAS_HOST);          // (the stage keeps its name: device time of the build in this mode)
            ABC_TRY(launch_alias_build_dev(ctx, w, K, ctx->alias_F, ctx->alias_A, ctx->alias_fail_dev, fail_pin, &verdict));
        }
        ctx->alias_dev_builds++;
        if (while_host_builds) ABC_TRY(while_host_builds(hook_arg));      // (nothing waits here: the caller's table-independent work simply follows)
        {
            StageTimer tm(ctx, ST_RESAMPLE);
            hipLaunchKernelGGL(k_alias_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, raw, n, ctx->alias_F,
                               ctx->alias_A, K, (unsigned long long*)parent, verdict, ctx->alias_fail_dev, fail_pin);
            ABC_HIP(ctx, hipGetLastError());
        }
        if (alias_check_deferred) { *alias_check_deferred = 1; return ABC_OK; }      // the caller reads the pinned flag at its next wait
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!*fail_pin) return ABC_OK;
        ctx->alias_dev_fallbacks++;
        while_host_builds = nullptr;                      // (already run)
        raw_ready = raw;                                  // the draws' taus2 outputs exist: only the table is rebuilt
        wait_side = false;
        weights_on_host = false;
    }
    // alias table: weights to the host, serial Walker build, tables back to HBM
    ABC_TRY(abc_pin_reserve(ctx, abc_alias_pin_bytes(K)));
    double* hw = (double*)ctx->pin;
    double* hF = hw + K;                                    // F[K] and A[K] adjacent: ONE host-to-device copy, nothing to move
    uint32_t* hA = (uint32_t*)(hF + K);
    double* hE = (double*)(((uintptr_t)(hA + K) + 7) & ~(uintptr_t)7);
    uint32_t* hS = (uint32_t*)(hE + K);
    uint32_t* hB = hS + K + 1;
    if (!weights_on_host)         // (else already in hw: stored there by the kernel that normalised them)
        ABC_HIP(ctx, hipMemcpyAsync(hw, w, K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (!ctx->ev_copy) ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_copy, hipEventDisableTiming));
    ABC_HIP(ctx, hipEventRecord(ctx->ev_copy, ctx->stream));
    // the raw taus2 outputs of the draws do not depend on the table: queued behind the copy, generated while the host builds it
    // (or long since, on the side stream: raw_ready)
    uint32_t* raw = const_cast<uint32_t*>(raw_ready);
    if (!raw) raw = (uint32_t*)abc_ws_alloc(ctx, n * sizeof(uint32_t));
    if (!raw) ABC_FAIL(ctx, ABC_ERR_NOMEM, "resample: workspace exhausted");
    abc_rng base = *rng;
    taus2_jump(&base, i0);
    if (raw_ready) {
        if (wait_side && !ctx->side_early_waited) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));
    } else {
        StageTimer tm(ctx, ST_RESAMPLE);
        ABC_TRY(taus_stream(ctx, base, n, raw));
    }
    if (while_host_builds) ABC_TRY(while_host_builds(hook_arg));      // more table-independent GPU work of the caller
    ABC_HIP(ctx, hipEventSynchronize(ctx->ev_copy));
    if (abort_flag && *abort_flag) return ABC_INTERNAL_RETRY;
    {
        const auto t0 = std::chrono::steady_clock::now();
        abc_alias_preproc(K, hw, hF, hA, hE, hS, hB, /*knuth=*/false);
        if (ctx->timing) {
            ctx->stage_host_ms[ST_ALIAS_HOST] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            ctx->stage_cnt[ST_ALIAS_HOST] += 1;
        }
    }
    StageTimer tm(ctx, ST_RESAMPLE);
    ABC_HIP(ctx, hipMemcpyAsync(ctx->alias_F, hF, K * (sizeof(double) + sizeof(uint32_t)), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_alias_draw, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, raw, n, ctx->alias_F,
                       ctx->alias_A, K, (unsigned long long*)parent);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// The parts of the perturbation that need neither the parents nor the alias table: the row-major copy of the posterior the
// kernel gathers parents from, and the simulator seeds (taus2 outputs seed_stream_offset + i0 + i).  launch_perturb does them
// itself unless the caller already has (the fused driver runs them while the host builds the alias table).
static int launch_theta_rows(abc_ctx* ctx, const double* theta, size_t K, size_t P, int PP, double* rows, const double* L,
                             double* Lpad) {
    const unsigned grid = (unsigned)((K + 63) / 64);
    if (PP > 64) {
        size_t gb = (K * (size_t)PP + 255) / 256;
        if (gb > 4096) gb = 4096;
        hipLaunchKernelGGL(k_theta_rows_gen, dim3((unsigned)gb), dim3(256), 0, ctx->stream, theta, K, (int)P, PP, rows, L, Lpad);
        ABC_HIP(ctx, hipGetLastError());
        return ABC_OK;
    }
    switch (PP) {
        case 2: hipLaunchKernelGGL((k_theta_rows<2>), dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, rows, L, Lpad); break;
        case 4: hipLaunchKernelGGL((k_theta_rows<4>), dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, rows, L, Lpad); break;
        case 8: hipLaunchKernelGGL((k_theta_rows<8>), dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, rows, L, Lpad); break;
        case 16: hipLaunchKernelGGL((k_theta_rows<16>), dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, rows, L, Lpad); break;
        case 32: hipLaunchKernelGGL((k_theta_rows<32>), dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, rows, L, Lpad); break;
        default: hipLaunchKernelGGL((k_theta_rows<64>), dim3(grid), dim3(256), 0, ctx->stream, theta, K, (int)P, rows, L, Lpad); break;
    }
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
static int launch_seeds(abc_ctx* ctx, const abc_rng* rng, uint64_t i0, size_t n, uint64_t* seeds, uint64_t seed_stream_offset,
                        hipStream_t st) {
    // AbcSmc.cpp:535: one gsl_rng_get per new particle; here taken from the taus2 stream at
    // position seed_stream_offset + i0 + i (after the resampling draws)
    uint32_t* raw = (uint32_t*)abc_ws_alloc(ctx, n * sizeof(uint32_t));
    if (!raw) ABC_FAIL(ctx, ABC_ERR_NOMEM, "perturb: workspace exhausted");
    abc_rng base = *rng;
    taus2_jump(&base, seed_stream_offset + i0);
    ABC_TRY(taus_stream(ctx, base, n, raw, st));
    hipLaunchKernelGGL(k_widen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st ? st : ctx->stream, raw, n, (unsigned long long*)seeds);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
int launch_perturb_prepare(abc_ctx* ctx, const abc_rng* rng, const double* theta, size_t K, size_t P, uint64_t i0, size_t n,
                           uint64_t* seeds, uint64_t seed_stream_offset, abc_perturb_prep* prep, int multivariate,
                           const double* L_or_dv) {
    if (n == 0) { prep->rows = nullptr; prep->Lpad = nullptr; return ABC_OK; }
    const int PP = abc_perturb_pp(P);
    StageTimer tm(ctx, ST_PERTURB);
    if (!prep->rows) {        // (preset: the caller's k_theta_moments launch has written the copy and the padded factor already)
        double* rows = (double*)abc_ws_alloc(ctx, K * (size_t)PP * sizeof(double));
        double* Lpad = (multivariate && L_or_dv) ? (double*)abc_ws_alloc(ctx, (size_t)PP * PP * sizeof(double)) : nullptr;
        if (!rows || (multivariate && L_or_dv && !Lpad)) ABC_FAIL(ctx, ABC_ERR_NOMEM, "perturb: workspace exhausted");
        ABC_TRY(launch_theta_rows(ctx, theta, K, P, PP, rows, Lpad ? L_or_dv : nullptr, Lpad));
        prep->rows = rows;
        prep->Lpad = Lpad;
    }
    if (seeds && !prep->seeds_done) { ABC_TRY(launch_seeds(ctx, rng, i0, n, seeds, seed_stream_offset, nullptr)); prep->seeds_done = 1; }
    return ABC_OK;
}

// the proposals' give-up counter: [0] the counter, [1] its snapshot (taken by the gather in front of a generation's weight stage:
// a repeated perturbation starts from it), [2] the address of the pinned flag word (note_giveup)
int abc_giveups_ensure(abc_ctx* ctx) {
    if (ctx->giveups_dev) return ABC_OK;
    ABC_HIP(ctx, hipMalloc((void**)&ctx->giveups_dev, 3 * sizeof(unsigned long long)));
    const unsigned long long init[3] = {0ull, 0ull, (unsigned long long)(size_t)(ctx->status_pin + 56)};
    ABC_HIP(ctx, hipMemcpy(ctx->giveups_dev, init, sizeof(init), hipMemcpyHostToDevice));
    return ABC_OK;
}

int launch_perturb(abc_ctx* ctx, const abc_rng* rng, const double* theta, size_t K, size_t P, const abc_prior* priors,
                   const uint64_t* parent, uint64_t i0, size_t n, int multivariate, const double* L_or_dv, double* out,
                   uint64_t* seeds, uint64_t seed_stream_offset, const abc_perturb_prep* prep) {
    if (n == 0) return ABC_OK;
    int PP = 2;
    while (PP < (int)P) PP *= 2;
    if (P > 64) PP = (int)((P + 63) / 64 * 64);
    ABC_TRY(abc_giveups_ensure(ctx));
    StageTimer tm(ctx, ST_PERTURB);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    double* rows = (prep && prep->rows) ? prep->rows : (double*)abc_ws_alloc(ctx, K * (size_t)PP * sizeof(double));
    if (!rows) ABC_FAIL(ctx, ABC_ERR_NOMEM, "perturb: workspace exhausted");
    const double* Lpad = (prep && prep->rows) ? prep->Lpad : nullptr;
    if (!(prep && prep->rows)) {
        const bool need_lp = multivariate != 0;
        double* lp = need_lp ? (double*)abc_ws_alloc(ctx, (size_t)PP * PP * sizeof(double)) : nullptr;
        if (need_lp && !lp) ABC_FAIL(ctx, ABC_ERR_NOMEM, "perturb: workspace exhausted");
        ABC_TRY(launch_theta_rows(ctx, theta, K, P, PP, rows, lp ? L_or_dv : nullptr, lp));
        Lpad = lp;
    }
    theta = rows;
    if (multivariate && !Lpad) ABC_FAIL(ctx, ABC_ERR_INVALID, "perturb: the padded factor was not prepared");
    if (PP > 64) {
        const size_t lds = P * sizeof(abc_prior);
        if (multivariate)
            hipLaunchKernelGGL((k_perturb_gen<true>), dim3(blocks), dim3(256), lds, ctx->stream, *rng, theta, (int)P, PP, priors,
                               (const unsigned long long*)parent, (unsigned long long)i0, n, Lpad, out, ctx->giveups_dev);
        else
            hipLaunchKernelGGL((k_perturb_gen<false>), dim3(blocks), dim3(256), lds, ctx->stream, *rng, theta, (int)P, PP, priors,
                               (const unsigned long long*)parent, (unsigned long long)i0, n, L_or_dv, out, ctx->giveups_dev);
        ABC_HIP(ctx, hipGetLastError());
        if (seeds && !(prep && prep->seeds_done)) ABC_TRY(launch_seeds(ctx, rng, i0, n, seeds, seed_stream_offset, nullptr));
        return ABC_OK;
    }
#define LAUNCH_PT(PPV)                                                                                                 \
    do {                                                                                                               \
        if (multivariate)                                                                                              \
            hipLaunchKernelGGL((k_perturb<PPV, true>), dim3(blocks), dim3(256), 0, ctx->stream, *rng, theta, K, (int)P, \
                               priors, (const unsigned long long*)parent, (unsigned long long)i0, n, Lpad, out,       \
                               ctx->giveups_dev);                                                                     \
        else                                                                                                           \
            hipLaunchKernelGGL((k_perturb<PPV, false>), dim3(blocks), dim3(256), 0, ctx->stream, *rng, theta, K, (int)P, \
                               priors, (const unsigned long long*)parent, (unsigned long long)i0, n, L_or_dv, out,    \
                               ctx->giveups_dev);                                                                     \
    } while (0)
    switch (PP) {
        case 2: LAUNCH_PT(2); break;
        case 4: LAUNCH_PT(4); break;
        case 8: LAUNCH_PT(8); break;
        case 16: LAUNCH_PT(16); break;
        case 32: LAUNCH_PT(32); break;
        default:                               // 33..64 parameters
            if (multivariate) {                // (until round 6: k_perturb_stream, the factor in LDS, one read per FMA, one wave per SIMD)
                LAUNCH_PT(64);
            } else {
                const size_t lds = 64 * sizeof(double) + 64 * sizeof(abc_prior);
                hipLaunchKernelGGL((k_perturb_stream<64, false>), dim3(blocks), dim3(256), lds, ctx->stream, *rng, theta, K,
                                   (int)P, priors, (const unsigned long long*)parent, (unsigned long long)i0, n, L_or_dv, out,
                                   ctx->giveups_dev);
            }
            break;
    }
#undef LAUNCH_PT
    ABC_HIP(ctx, hipGetLastError());
    if (seeds && !(prep && prep->seeds_done)) ABC_TRY(launch_seeds(ctx, rng, i0, n, seeds, seed_stream_offset, nullptr));
    return ABC_OK;
}

// ---- reference-stream proposals (abc_ctx_set_noise_mode): everything the host loop needs comes down, the proposals go back up
int launch_perturb_reference(abc_ctx* ctx, abc_rng* rng_after_draws, const double* theta, size_t K, size_t P,
                             const abc_prior* priors, const uint64_t* parent, size_t n, int multivariate, const double* L_or_dv,
                             double* out, uint64_t* seeds) {
    if (n == 0) return ABC_OK;
    StageTimer tm(ctx, ST_PERTURB);
    std::vector<double> hth(K * P), hl(multivariate ? P * P : P), hout(n * P);
    std::vector<uint64_t> hpar(n), hseeds(seeds ? n : 0);
    std::vector<abc_prior> hpr(P);
    ABC_HIP(ctx, hipMemcpyAsync(hth.data(), theta, hth.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipMemcpyAsync(hl.data(), L_or_dv, hl.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipMemcpyAsync(hpar.data(), parent, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipMemcpyAsync(hpr.data(), priors, P * sizeof(abc_prior), hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t0 = std::chrono::steady_clock::now();
    if (multivariate) {
        for (size_t a = 0; a < P; a++)           // the factor as gsl_linalg_cholesky_decomp1 leaves it has covariance entries above
            for (size_t b = a + 1; b < P; b++) hl[a + P * b] = 0.0;      // the diagonal: dtrmv(Lower) never reads them; neither do we
        ctx->giveups_host += abc_ref_perturb_mvn(rng_after_draws, n, K, P, hth.data(), hpar.data(), hl.data(), hpr.data(),
                                                 MVN_MAX_TRIES, hout.data());
    } else {
        ctx->giveups_host += abc_ref_perturb_indep(rng_after_draws, n, K, P, hth.data(), hpar.data(), hl.data(), hpr.data(),
                                                   hout.data());
    }
    if (seeds) abc_ref_seeds(rng_after_draws, n, hseeds.data());
    ctx->stage_host_ms[ST_PERTURB] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ABC_HIP(ctx, hipMemcpyAsync(out, hout.data(), hout.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (seeds) ABC_HIP(ctx, hipMemcpyAsync(seeds, hseeds.data(), n * 8, hipMemcpyHostToDevice, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));       // the host vectors go out of scope
    return ABC_OK;
}
