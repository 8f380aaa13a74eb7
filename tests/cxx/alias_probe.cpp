// CPU check of abcsmc_amd/csrc/alias_host.h (the host side of the resampling table): alias_sequential_sum must equal
// the naive loop `s += w[k]` BIT FOR BIT on every input, and alias_preproc the naive restatement of GSL's
// gsl_ran_discrete_preproc (sequential total, two LIFO stacks).  Prints "ok <cases>" or the first mismatch; `time K`
// prints the milliseconds of both sums and of the whole build.   g++ -O2 -ffp-contract=off tests/cxx/alias_probe.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../abcsmc_amd/csrc/alias_host.h"

static double naive_sum(const double* w, size_t K) {
    volatile double s = 0.0;                       // volatile: no vectorisation / reassociation whatever the flags
    for (size_t k = 0; k < K; k++) s = s + w[k];
    return s;
}
static bool same_bits(double a, double b) { return memcmp(&a, &b, 8) == 0; }

// [GSL] randist/discrete.c, Knuth convention; push order and pops as upstream
static void naive_preproc(size_t K, const double* w, std::vector<double>& F, std::vector<uint32_t>& A) {
    std::vector<double> E(K);
    std::vector<size_t> S, B;
    double total = naive_sum(w, K);
    const double mean = 1.0 / (double)K;
    for (size_t k = 0; k < K; k++) E[k] = w[k] / total;
    for (size_t k = 0; k < K; k++) (E[k] < mean ? S : B).push_back(k);
    while (!S.empty()) {
        const size_t s = S.back(); S.pop_back();
        if (B.empty()) { A[s] = (uint32_t)s; F[s] = 1.0; continue; }
        const size_t b = B.back(); B.pop_back();
        A[s] = (uint32_t)b;
        F[s] = (double)K * E[s];
        const double d = mean - E[s];
        E[s] += d;
        E[b] -= d;
        if (E[b] < mean) S.push_back(b);
        else if (E[b] > mean) B.push_back(b);
        else { A[b] = (uint32_t)b; F[b] = 1.0; }
    }
    while (!B.empty()) { const size_t b = B.back(); B.pop_back(); A[b] = (uint32_t)b; F[b] = 1.0; }
    for (size_t k = 0; k < K; k++) F[k] = (F[k] + (double)k) / (double)K;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    std::mt19937_64 g(12345);
    std::normal_distribution<double> nd(0.0, 1.0);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    if (argc > 2 && !strcmp(argv[1], "time")) {
        const size_t K = (size_t)atol(argv[2]);
        std::vector<double> w(K), F(K), E(K);
        std::vector<uint32_t> A(K), S(K + 1), B(K + 1);
        for (auto& x : w) x = exp(1.5 * nd(g));
        double t0 = now_ms(), a = 0, b = 0;
        for (int r = 0; r < 5; r++) a = naive_sum(w.data(), K);
        double t1 = now_ms();
        for (int r = 0; r < 5; r++) b = alias_sequential_sum(w.data(), K);
        double t2 = now_ms();
        for (int r = 0; r < 5; r++) alias_preproc(K, w.data(), F.data(), A.data(), E.data(), S.data(), B.data());
        double t3 = now_ms();
        printf("K=%zu  naive sum %.3f ms  exact blocked sum %.3f ms (%s)  whole build %.3f ms\n", K, (t1 - t0) / 5, (t2 - t1) / 5,
               same_bits(a, b) ? "same bits" : "DIFFERENT", (t3 - t2) / 5);
        return same_bits(a, b) ? 0 : 1;
    }
    long cases = 0;
    for (int rep = 0; rep < 400; rep++) {
        const int kind = rep % 10;
        size_t K = (rep < 40) ? (size_t)rep * 37 : (size_t)(1 + g() % 20000);
        if (rep % 97 == 0) K = 300000 + g() % 1000;
        std::vector<double> w(K);
        for (size_t k = 0; k < K; k++) {
            double x;
            switch (kind) {
                case 0: x = ud(g); break;                                          // uniform
                case 1: x = exp(3.0 * nd(g)); break;                               // heavy tailed
                case 2: x = ldexp(1.0, -(int)(g() % 60)); break;                   // powers of two: exact ties everywhere
                case 3: x = (double)(g() % 1000) * 0.125; break;                   // small multiples of 1/8, zeros
                case 4: x = ud(g) * ((g() % 50 == 0) ? 1e12 : 1.0); break;         // rare huge elements (binade jumps)
                case 5: x = ud(g) * 1e-310; break;                                 // denormals
                case 6: x = (k % 1000 == 999) ? -ud(g) : ud(g); break;             // a few negatives
                case 7: x = 1e-5 + 1e-12 * ud(g); break;                           // nearly equal (a posterior's weights)
                case 8: x = ud(g) * 1e300; break;                                  // overflow to inf on the way
                default: x = (k == K / 2 && rep % 20 == 9) ? NAN : exp(nd(g)); break;
            }
            w[k] = x;
        }
        const double a = naive_sum(w.data(), K), b = alias_sequential_sum(w.data(), K);
        if (!same_bits(a, b) && !(a != a && b != b)) { printf("sum mismatch rep %d kind %d K %zu: %a vs %a\n", rep, kind, K, a, b); return 1; }
        cases++;
        if (K >= 2 && (kind == 0 || kind == 1 || kind == 3 || kind == 7)) {        // valid weight vectors: the whole table
            bool anypos = false;
            for (double x : w) anypos = anypos || x > 0;
            if (!anypos) continue;
            std::vector<double> F0(K), F1(K), E(K);
            std::vector<uint32_t> A0(K), A1(K), S(K + 1), B(K + 1);
            naive_preproc(K, w.data(), F0, A0);
            alias_preproc(K, w.data(), F1.data(), A1.data(), E.data(), S.data(), B.data());
            if (memcmp(F0.data(), F1.data(), K * 8) || memcmp(A0.data(), A1.data(), K * 4)) {
                printf("table mismatch rep %d kind %d K %zu\n", rep, kind, K);
                return 1;
            }
            cases++;
        }
    }
    printf("ok %ld\n", cases);
    return 0;
}
