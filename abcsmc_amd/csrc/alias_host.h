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
#define ALIAS_TARGET_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define ALIAS_TARGET_CLONES
#endif
typedef double alias_v4d __attribute__((vector_size(32)));
typedef long long alias_v4i __attribute__((vector_size(32)));
// Explicit 4-wide vectors (two in flight): one AVX2 instruction each in the avx2 clone, two SSE2 ones in the default.
ALIAS_TARGET_CLONES static bool alias_sum_block(const double* w, double inv_u, double* inc_sum) {
    const double Ms = 4503599627370496.0;                    // 2^52: (t + M) - M = RN(t) for 0 <= t < 2^51
    const alias_v4d M = {Ms, Ms, Ms, Ms}, IU = {inv_u, inv_u, inv_u, inv_u}, H = {0.5, 0.5, 0.5, 0.5}, Z = {0, 0, 0, 0};
    const double Ls = 17592186044416.0;                      // 2^44: 256 increments below it sum exactly (< 2^52)
    const alias_v4d LIM = {Ls, Ls, Ls, Ls};
    alias_v4d acc0 = Z, acc1 = Z;
    alias_v4i bad = {0, 0, 0, 0};
    for (int i = 0; i < 256; i += 8) {
        alias_v4d a, b;
        __builtin_memcpy(&a, w + i, 32);
        __builtin_memcpy(&b, w + i + 4, 32);
        const alias_v4d ta = a * IU, tb = b * IU;            // exact (power of two); overflow -> inf -> rejected
        const alias_v4d ra = (ta + M) - M, rb = (tb + M) - M;
        const alias_v4d da = ra - ta, db = rb - tb;
        acc0 += ra;                                          // integers: exact while the block total stays below 2^53
        acc1 += rb;
        // reject: a tie (the increment would depend on the parity of the running sum), negative, >= 2^44, NaN
        bad |= (da == H) | (da == -H) | (ta < Z) | !(ta < LIM);
        bad |= (db == H) | (db == -H) | (tb < Z) | !(tb < LIM);
    }
    const alias_v4d acc = acc0 + acc1;
    *inc_sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    return (bad[0] | bad[1] | bad[2] | bad[3]) == 0;
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
    for (size_t k = 0; k < K; k++) E[k] = w[k] / total;      // vectorised by the host compiler
    size_t ns = 0, nb = 0;
    for (size_t k = 0; k < K; k++) {          // branch-free: for random weights a conditional push mispredicts every
        const bool sm = E[k] < mean;          // other element (2.8 -> 0.7 ms at K = 8e5); scratch holds K + 1 entries
        smalls[ns] = (uint32_t)k;
        bigs[nb] = (uint32_t)k;
        ns += sm;
        nb += !sm;
    }
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
        if (eb < mean) { E[cb] = eb; smalls[ns++] = cb; have = false; }       // demoted: it is served next
        else if (!(eb > mean)) { A[cb] = cb; F[cb] = 1.0; have = false; }     // exactly full
    }
    if (have) { A[cb] = cb; F[cb] = 1.0; }
    while (nb) { const uint32_t b = bigs[--nb]; A[b] = b; F[b] = 1.0; }
    if (knuth) for (size_t k = 0; k < K; k++) F[k] = (F[k] + (double)k) / dK;            // KNUTH_CONVENTION
}
