// Host micro-benchmark of the Walker alias build phases (sequential total, division+classification, serving loop,
// Knuth transform) with 1..N threads on the order-free passes.   g++ -O3 -pthread scripts/alias_bench.cpp -o alias_bench
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename Fn> void par(size_t K, int nt, Fn f) {
    if (nt <= 1) { f(0, (size_t)0, K); return; }
    std::vector<std::thread> th;
    const size_t chunk = (K + nt - 1) / nt;
    for (int t = 1; t < nt; t++) th.emplace_back([=] { f(t, std::min(K, chunk * t), std::min(K, chunk * (t + 1))); });
    f(0, (size_t)0, std::min(K, chunk));
    for (auto& x : th) x.join();
}
int main(int argc, char** argv) {
    const size_t K = argc > 1 ? atol(argv[1]) : 800000;
    std::vector<double> w(K), E(K), F(K);
    std::vector<uint32_t> A(K), S(K + 1), B(K + 1);
    srand(1);
    for (auto& x : w) { const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); x = std::exp(1.5 * std::sqrt(-2 * std::log(u)) * std::cos(6.283185307 * v)); }
    for (int nt : {1, 2, 4, 8, 16}) {
        double ph[5] = {0};
        for (int rep = 0; rep < 5; rep++) {
            double t0 = now(), total = 0;
            for (size_t k = 0; k < K; k++) total += w[k];
            double t1 = now();
            const double mean = 1.0 / K, dK = (double)K;
            size_t cs[17] = {0}, cb[17] = {0};
            const size_t chunk = (K + nt - 1) / nt;
            par(K, nt, [&](int t, size_t lo, size_t hi) {
                size_t s = lo, b = lo;
                for (size_t k = lo; k < hi; k++) { const double e = w[k] / total; E[k] = e; if (e < mean) S[s++] = k; else B[b++] = k; }
                cs[t + 1] = s - lo; cb[t + 1] = b - lo;
            });
            size_t ns = cs[1], nb = cb[1];
            for (int t = 1; t < nt; t++) { const size_t lo = std::min(K, chunk * t); memmove(&S[ns], &S[lo], cs[t + 1] * 4); memmove(&B[nb], &B[lo], cb[t + 1] * 4); ns += cs[t + 1]; nb += cb[t + 1]; }
            double t2 = now();
            bool have = false; uint32_t cbig = 0; double eb = 0;
            while (ns) {
                const uint32_t s = S[--ns];
                if (!have) { if (!nb) { A[s] = s; F[s] = 1; continue; } cbig = B[--nb]; eb = E[cbig]; have = true; }
                const double es = E[s];
                A[s] = cbig; F[s] = dK * es; eb -= mean - es;
                if (eb < mean) { E[cbig] = eb; S[ns++] = cbig; have = false; }
                else if (!(eb > mean)) { A[cbig] = cbig; F[cbig] = 1; have = false; }
            }
            if (have) { A[cbig] = cbig; F[cbig] = 1; }
            while (nb) { const uint32_t b = B[--nb]; A[b] = b; F[b] = 1; }
            double t3 = now();
            par(K, nt, [&](int, size_t lo, size_t hi) { for (size_t k = lo; k < hi; k++) F[k] = (F[k] + (double)k) / dK; });
            double t4 = now();
            if (rep) { ph[0] += t1 - t0; ph[1] += t2 - t1; ph[2] += t3 - t2; ph[3] += t4 - t3; }
        }
        printf("K=%zu threads=%2d  total %.3f  divide+classify %.3f  loop %.3f  knuth %.3f  (ms, mean of 4)\n", K, nt, ph[0] / 4, ph[1] / 4, ph[2] / 4, ph[3] / 4);
    }
}
