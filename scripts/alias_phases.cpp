// Host timing of the phases of the Walker alias build (abcsmc_amd/csrc/alias_host.h) on one core: exact blocked sum, division,
// classification, serving loop, Knuth transform.   clang++ -O3 -ffp-contract=off scripts/alias_phases.cpp -o alias_phases && ./alias_phases 800000
#include <chrono>
#include <cstdio>
#include <random>
#include <vector>
#include "../abcsmc_amd/csrc/alias_host.h"
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const size_t K = atol(argv[1]);
    std::mt19937_64 g(1); std::normal_distribution<double> nd(0, 1);
    std::vector<double> w(K), F(K), E(K); std::vector<uint32_t> A(K), S(K + 1), B(K + 1);
    for (auto& x : w) x = exp(1.5 * nd(g));
    double ph[6] = {0};
    for (int rep = 0; rep < 6; rep++) {
        double t0 = now_ms();
        const double total = alias_sequential_sum(w.data(), K);
        double t1 = now_ms();
        const double mean = 1.0 / (double)K, dK = (double)K;
        alias_divide(w.data(), total, E.data(), K);
        double t2 = now_ms();
        size_t ns = 0, nb = 0;
        uint32_t* smalls = S.data(); uint32_t* bigs = B.data();
        alias_classify(E.data(), mean, K, smalls, bigs, ns, nb);
        double t3 = now_ms();
        bool have = false; uint32_t cb = 0; double eb = 0.0;
        while (ns) {
            const uint32_t s = smalls[--ns];
            if (!have) { if (!nb) { A[s] = s; F[s] = 1.0; continue; } cb = bigs[--nb]; eb = E[cb]; have = true; }
            const double es = E[s]; A[s] = cb; F[s] = dK * es; eb -= mean - es;
            while (eb < mean) {
                if (!nb) { A[cb] = cb; F[cb] = 1.0; have = false; break; }
                const uint32_t nbig = bigs[--nb]; double enb = E[nbig];
                A[cb] = nbig; F[cb] = dK * eb; enb -= mean - eb; cb = nbig; eb = enb;
            }
            if (have && !(eb > mean) && !(eb < mean)) { A[cb] = cb; F[cb] = 1.0; have = false; }
        }
        if (have) { A[cb] = cb; F[cb] = 1.0; }
        while (nb) { const uint32_t b = bigs[--nb]; A[b] = b; F[b] = 1.0; }
        double t4 = now_ms();
        for (size_t k = 0; k < K; k++) F[k] = (F[k] + (double)k) / dK;
        double t5 = now_ms();
        if (rep) { ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; ph[4] += t5 - t4; }
    }
    printf("K=%zu sum %.3f divide %.3f classify %.3f serve %.3f knuth %.3f ms\n", K, ph[0]/5, ph[1]/5, ph[2]/5, ph[3]/5, ph[4]/5);
}
