// Host side of the resampling table: GSL's gsl_ran_discrete_preproc restated so that every floating-point result is
// bit-identical with the sequential reference algorithm (that is what makes the resampled parents bit-exact), but
// without its serial dependencies where they can be removed exactly.  Plain C++ (no HIP): included by resample.hip and
// compiled on its own by tests/cxx/alias_probe.cpp, which checks it against the naive loops.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

// fl(...fl(fl(w0 + w1) + w2)... + w[K-1]): the value of the loop `s = 0; for k: s += w[k]`, bit for bit, without its
// K-long chain of dependent additions (1.0 of the 2.6 ms of the build at K = 8e5).  While the running sum s stays inside
// one binade, with ulp u, a rounded addition of w >= 0 is s + RN(w/u) u -- an INTEGER increment that does not depend on
// s except at an exact tie -- so a block of 256 additions is the exact integer sum of its increments (independent
// vector accumulators).  Blocks containing a tie (|RN(t) - t| == 1/2), a negative, huge or non-finite element, or
// crossing into the next binade are redone by the plain loop; so are the first elements (s == 0).
#if defined(__x86_64__)
#define ALIAS_TARGET_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define ALIAS_TARGET_CLONES
#endif
typedef double alias_v8d __attribute__((vector_size(64)));
typedef long long alias_v8i __attribute__((vector_size(64)));
// Explicit 8-wide vectors (two in flight): one AVX-512 instruction each in that clone, two AVX2 / four SSE2 ones in the others.
ALIAS_TARGET_CLONES static bool alias_sum_block(const double* w, double inv_u, double* inc_sum) {
    const double Ms = 4503599627370496.0;                    // 2^52: (t + M) - M = RN(t) for 0 <= t < 2^51
    const double Ls = 17592186044416.0;                      // 2^44: 256 increments below it sum exactly (< 2^52)
    alias_v8d M, IU, H, Z, LIM;
    for (int i = 0; i < 8; i++) { M[i] = Ms; IU[i] = inv_u; H[i] = 0.5; Z[i] = 0.0; LIM[i] = Ls; }
    alias_v8d acc0 = Z, acc1 = Z;
    alias_v8i bad = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 256; i += 16) {
        alias_v8d a, b;
        __builtin_memcpy(&a, w + i, 64);
        __builtin_memcpy(&b, w + i + 8, 64);
        const alias_v8d ta = a * IU, tb = b * IU;            // exact (power of two); overflow -> inf -> rejected
        const alias_v8d ra = (ta + M) - M, rb = (tb + M) - M;
        const alias_v8d da = ra - ta, db = rb - tb;
        acc0 += ra;                                          // integers: exact while the block total stays below 2^53
        acc1 += rb;
        // reject: a tie (the increment would depend on the parity of the running sum), negative, >= 2^44, NaN
        bad |= (da == H) | (da == -H) | (ta < Z) | !(ta < LIM);
        bad |= (db == H) | (db == -H) | (tb < Z) | !(tb < LIM);
    }
    const alias_v8d acc = acc0 + acc1;
    *inc_sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    long long anybad = 0;
    for (int i = 0; i < 8; i++) anybad |= bad[i];
    return anybad == 0;
}
inline double alias_sequential_sum(const double* w, size_t K) {
    double s = 0.0;
    size_t k = 0;
    while (k < K) {
        const size_t nb = (K - k < 256) ? K - k : 256;
        bool done = false;
        if (nb == 256 && s >= 1e-290 && s <= 1e290) {
            int e;
            (void)frexp(s, &e);                              // s in [2^(e-1), 2^e): ulp 2^(e-53)
            const double inv_u = ldexp(1.0, 53 - e), u = ldexp(1.0, e - 53);
            double inc;
            if (alias_sum_block(w + k, inv_u, &inc)) {
                const double S = s * inv_u + inc;            // integers below 2^53 + 2^52: exact if the result is < 2^53
                if (S < 9007199254740992.0) { s = S * u; k += 256; done = true; }
            }
        }
        if (!done) { for (size_t i = 0; i < nb; i++) s += w[k + i]; k += nb; }
    }
    return s;
}

// E[k] = w[k] / total: IEEE divisions, throughput-bound -- one clone per vector width the host may have (the baseline build
// divides two at a time: 40 us of the build at K = 1e5 on the GPU box's Zen 5 host, which has 512-bit units)
#if defined(__x86_64__)
#define ALIAS_WIDE_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define ALIAS_WIDE_CLONES
#endif
ALIAS_WIDE_CLONES static void alias_divide(const double* w, double total, double* E, size_t K) {
    for (size_t k = 0; k < K; k++) E[k] = w[k] / total;
}

// smalls / bigs: the indices with E[k] < mean / the others, both in ascending order (the stacks GSL fills by pushing k = 0, 1, ...).
// Branch-free scalar loop (for random weights a conditional push mispredicts every other element: 2.8 -> 0.7 ms at K = 8e5);
// with AVX-512 eight comparisons and two compress-stores per step (scratch holds K + 1 entries either way).
static inline void alias_classify_scalar(const double* E, double mean, size_t k0, size_t K, uint32_t* smalls, uint32_t* bigs,
                                         size_t& ns, size_t& nb) {
    for (size_t k = k0; k < K; k++) {
        const bool sm = E[k] < mean;
        smalls[ns] = (uint32_t)k;
        bigs[nb] = (uint32_t)k;
        ns += sm;
        nb += !sm;
    }
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx512f,avx512vl"))) static size_t alias_classify_avx512(const double* E, double mean, size_t K,
                                                                                uint32_t* smalls, uint32_t* bigs, size_t& ns,
                                                                                size_t& nb) {
    const __m512d vm = _mm512_set1_pd(mean);
    __m256i idx = _mm256_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7);
    const __m256i eight = _mm256_set1_epi32(8);
    size_t k = 0;
    for (; k + 8 <= K; k += 8) {
        const __mmask8 m = _mm512_cmp_pd_mask(_mm512_loadu_pd(E + k), vm, _CMP_LT_OQ);      // false for NaN, as `<`
        // compress in a register, store the whole vector (compress-stores to memory are microcoded on Zen 4 / 5): the lanes past
        // the count land on entries that later steps overwrite or that stay unused (ns, nb <= k, so ns + 8 <= K)
        _mm256_storeu_si256((__m256i*)(smalls + ns), _mm256_maskz_compress_epi32(m, idx));
        _mm256_storeu_si256((__m256i*)(bigs + nb), _mm256_maskz_compress_epi32((__mmask8)~m, idx));
        const size_t c = (size_t)__builtin_popcount((unsigned)m);
        ns += c;
        nb += 8 - c;
        idx = _mm256_add_epi32(idx, eight);
    }
    return k;
}
#endif
static inline void alias_classify(const double* E, double mean, size_t K, uint32_t* smalls, uint32_t* bigs, size_t& ns, size_t& nb) {
    size_t k0 = 0;
#if defined(__x86_64__)
    if (K < 0xFFFFFFF0u && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl"))
        k0 = alias_classify_avx512(E, mean, K, smalls, bigs, ns, nb);
#endif
    alias_classify_scalar(E, mean, k0, K, smalls, bigs, ns, nb);
}

// [GSL] gsl_ran_discrete_preproc (randist/discrete.c): Walker alias with two LIFO stacks.
// scratch: K doubles (E) + 2 (K + 1) uint32 (the stacks), caller-provided so the hot loop never allocates.
// knuth = false: F is left as the cut-off fractions in [0, 1]; the caller applies GSL's KNUTH_CONVENTION map (F[k] + k) / K where
// the table is read (k_alias_draw does, with the same two IEEE operations), which takes the last pass over K off the host.
inline void alias_preproc(size_t K, const double* w, double* F, uint32_t* A, double* E, uint32_t* smalls, uint32_t* bigs,
                          bool knuth = true) {
    // Same sequence of floating-point operations as GSL's loop (sequential total, E = w / total, one subtraction
    // per small from the big on top of the stack); only the bookkeeping differs: a big that stays big after serving
    // a small is pushed and popped again at once upstream, here it simply stays in registers.  (Threading the
    // order-free passes was measured on the GPU box, scripts/alias_bench.cpp: at K = 8e5 the serving loop and the
    // sequential total are 60-75 % of the time and thread start-up eats the rest of the gain.)
    const double total = alias_sequential_sum(w, K);          // == the loop `total += w[k]`, bit for bit
    const double mean = 1.0 / (double)K, dK = (double)K;
    alias_divide(w, total, E, K);
    size_t ns = 0, nb = 0;
    alias_classify(E, mean, K, smalls, bigs, ns, nb);
    // The serving loop.  GSL pops a small, lets the big on top of the other stack give it what it lacks, and pushes that big onto
    // the small stack once its residual falls below the mean -- where it is popped again at once, as the next small, for the next
    // big.  Here a demoted big never travels through the stack: it is served in an inner loop by the next big(s), with the same
    // operations on the same values in the same order (E[cb] = eb is what the pop would have read back).
    bool have = false;
    uint32_t cb = 0;
    double eb = 0.0;
    while (ns) {
        const uint32_t s = smalls[--ns];
        if (!have) {
            if (!nb) { A[s] = s; F[s] = 1.0; continue; }
            cb = bigs[--nb];
            eb = E[cb];
            have = true;
        }
        const double es = E[s];
        A[s] = cb;
        F[s] = dK * es;
        eb -= mean - es;
        while (eb < mean) {                     // demoted: cb is the next small
            if (!nb) { A[cb] = cb; F[cb] = 1.0; have = false; break; }      // no big left to serve it
            const uint32_t nbig = bigs[--nb];
            double enb = E[nbig];
            A[cb] = nbig;
            F[cb] = dK * eb;
            enb -= mean - eb;
            cb = nbig;
            eb = enb;
        }
        if (have && !(eb > mean) && !(eb < mean)) { A[cb] = cb; F[cb] = 1.0; have = false; }     // exactly full (or NaN)
    }
    if (have) { A[cb] = cb; F[cb] = 1.0; }
    while (nb) { const uint32_t b = bigs[--nb]; A[b] = b; F[b] = 1.0; }
    if (knuth) for (size_t k = 0; k < K; k++) F[k] = (F[k] + (double)k) / dK;            // KNUTH_CONVENTION
}
