// Host-only translation unit of libabcsmc_hip.so: see refstream_host.h.
#include "refstream_host.h"

#include <cmath>
#include <vector>

// taus2 (defined in resample.hip)
uint32_t taus2_get(abc_rng* r);

namespace {

inline double uniform_pos(abc_rng* r) {                 // [GSL] gsl_rng_uniform_pos: a zero output is drawn again
    double x;
    do { x = taus2_get(r) / 4294967296.0; } while (x == 0.0);
    return x;
}
inline double ran_gaussian(abc_rng* r, double sigma) {  // [GSL] randist/gauss.c gsl_ran_gaussian (polar Box-Muller)
    double x, y, r2;
    do {
        x = -1.0 + 2.0 * uniform_pos(r);
        y = -1.0 + 2.0 * uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    return sigma * y * std::sqrt(-2.0 * std::log(r2) / r2);
}
inline double likelihood(const abc_prior& pr, double v) {        // Priors.h:54-56, 76-78, 102-104
    if (pr.kind == ABC_PRIOR_GAUSS) {
        const double u = (v - pr.a) / std::fabs(pr.b);
        return (1.0 / (std::sqrt(2.0 * M_PI) * std::fabs(pr.b))) * std::exp(-u * u / 2.0);
    }
    if (pr.kind == ABC_PRIOR_UNIF_INT) return ((v == std::round(v)) && (pr.a <= v) && (v <= pr.b)) ? 1.0 / (pr.b - pr.a + 1.0) : 0.0;
    return ((pr.a <= v) && (v <= pr.b)) ? 1.0 / (pr.b - pr.a) : 0.0;
}
inline double recast(const abc_prior& pr, double v) { return (pr.kind == ABC_PRIOR_UNIF_INT) ? std::round(v) : v; }
inline bool valid(const abc_prior& pr, double v) { return likelihood(pr, v) != 0.0; }                  // Parameter.h:77
inline double prior_mean(const abc_prior& pr) { return (pr.kind == ABC_PRIOR_GAUSS) ? pr.a : (pr.b + pr.a) / 2.0; }

}  // namespace

size_t abc_ref_perturb_mvn(abc_rng* rng, size_t n, size_t K, size_t P, const double* theta, const uint64_t* parent,
                           const double* L, const abc_prior* priors, size_t max_tries, double* out) {
    std::vector<double> x(P), vals(P);
    size_t given_up = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t par = (size_t)parent[i];
        bool success = false;
        size_t tries = 0;
        while (!success) {                                                   // AbcUtil.cpp:132
            success = true;
            for (size_t p = 0; p < P; p++) x[p] = ran_gaussian(rng, 1.0);     // [GSL] multivariate_gaussian: ugaussian, in order
            for (size_t a = P; a > 0 && a--;) {                              // [GSL] dtrmv(Lower, NoTrans, NonUnit), i = P-1 .. 0
                double temp = 0.0;
                for (size_t b = 0; b < a; b++) temp += x[b] * L[a + P * b];
                x[a] = temp + x[a] * L[a + P * a];
            }
            for (size_t p = 0; p < P; p++) x[p] += theta[par + K * p];
            for (size_t p = 0; success && p < P; p++) {                       // :135-138, short-circuit
                vals[p] = recast(priors[p], x[p]);
                success = valid(priors[p], vals[p]);
            }
            if (!success && max_tries && ++tries >= max_tries) {              // declared bound (the reference retries for ever)
                for (size_t p = 0; p < P; p++) vals[p] = theta[par + K * p];
                given_up++;
                break;
            }
        }
        for (size_t p = 0; p < P; p++) out[i + n * p] = vals[p];
    }
    return given_up;
}

size_t abc_ref_perturb_indep(abc_rng* rng, size_t n, size_t K, size_t P, const double* theta, const uint64_t* parent,
                             const double* dv, const abc_prior* priors, double* out) {
    std::vector<double> sigma(P);
    for (size_t p = 0; p < P; p++) sigma[p] = std::sqrt(dv[p]);               // AbcUtil.cpp:150
    size_t fallbacks = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t par = (size_t)parent[i];
        for (size_t p = 0; p < P; p++) {
            const double mu = theta[par + K * p];
            size_t attempts = 1;                                             // Priors.h:23-26
            double dev = recast(priors[p], ran_gaussian(rng, sigma[p]) + mu);
            while (!valid(priors[p], dev) && (attempts++ < 1000)) dev = recast(priors[p], ran_gaussian(rng, sigma[p]) + mu);
            if (!valid(priors[p], dev)) { dev = prior_mean(priors[p]); fallbacks++; }
            out[i + n * p] = dev;
        }
    }
    return fallbacks;
}

void abc_ref_seeds(abc_rng* rng, size_t n, uint64_t* seeds) {                // AbcSmc.cpp:535
    for (size_t i = 0; i < n; i++) seeds[i] = taus2_get(rng);
}
