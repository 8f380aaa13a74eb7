// AbcUtilHip.hpp -- C++ host facade with the reference's own signatures for the hot path
// (/root/reference/include/AbcSmc/AbcUtil.h:78-172), header-only over the C ABI (include/abcsmc_hip.h).
//
// Drop-in intent: a translation unit that used `ABC::particle_ranking_PLS(...)` etc. from AbcUtil.h includes
// this header instead and links libabcsmc_hip.so.  Eigen and GSL are not required: Mat2D / Row / Col are the
// minimal column-major containers below (Eigen's default storage order, so `Eigen::Map` over `.data()` is a
// zero-copy view either way), `gsl_rng*` becomes `ABC::RNG*` (taus2, bit-compatible stream), and the
// `vector<const Parameter*>` argument keeps its meaning through the three concrete priors of Priors.h:46-110.
// Errors: the reference asserts / exits / lets GSL abort; here every failure throws ABC::HipError.
#ifndef ABCSMC_AMD_ABCUTILHIP_HPP
#define ABCSMC_AMD_ABCUTILHIP_HPP

#include <cmath>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/abcsmc_hip.h"

namespace ABC {

typedef double float_type;
typedef std::vector<float_type> Row;
typedef std::vector<float_type> Col;

struct Mat2D {                                   // column-major, like Eigen::Matrix<double,Dynamic,Dynamic>
    size_t r, c;
    std::vector<float_type> v;
    Mat2D() : r(0), c(0) {}
    Mat2D(size_t rows_, size_t cols_) : r(rows_), c(cols_), v(rows_ * cols_, 0.0) {}
    size_t rows() const { return r; }
    size_t cols() const { return c; }
    float_type& operator()(size_t i, size_t j) { return v[i + r * j]; }
    float_type operator()(size_t i, size_t j) const { return v[i + r * j]; }
    const float_type* data() const { return v.data(); }
    float_type* data() { return v.data(); }
};

struct HipError : std::runtime_error {
    int code;
    HipError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- parameters (Parameter.h:36-87, Priors.h:46-110): only what the hot path evaluates -------------------
struct Parameter {
    virtual ~Parameter() {}
    virtual abc_prior pod() const = 0;
};
struct GaussianPrior : Parameter {
    float_type mean, sd;
    GaussianPrior(float_type mn, float_type s) : mean(mn), sd(s) {}
    abc_prior pod() const override { return abc_prior{ABC_PRIOR_GAUSS, 0, mean, sd}; }
};
struct DiscreteUniformPrior : Parameter {
    long minval, maxval;
    DiscreteUniformPrior(long mn, long mx) : minval(mn), maxval(mx) {}
    abc_prior pod() const override { return abc_prior{ABC_PRIOR_UNIF_INT, 0, (double)minval, (double)maxval}; }
};
struct ContinuousUniformPrior : Parameter {
    float_type minval, maxval;
    ContinuousUniformPrior(float_type mn, float_type mx) : minval(mn), maxval(mx) {}
    abc_prior pod() const override { return abc_prior{ABC_PRIOR_UNIF_REAL, 0, minval, maxval}; }
};

// ---- RNG: stands in for `const gsl_rng*` of type gsl_rng_taus2 (examples/include/examples.h:10) ----------
struct RNG {
    mutable abc_rng state;
    explicit RNG(unsigned long seed = 0) { abc_rng_set(&state, seed); }
};
inline void rng_set(const RNG* r, unsigned long seed) { abc_rng_set(&r->state, seed); }     // gsl_rng_set
inline unsigned long rng_get(const RNG* r) { return abc_rng_get(&r->state); }                // gsl_rng_get

// ---- context (one per host thread, device 0 unless set before first use) -------------------------------
inline int& default_device() { static int d = 0; return d; }
inline abc_ctx* context() {
    static thread_local abc_ctx* ctx = nullptr;
    if (!ctx) {
        const int rc = abc_ctx_create(default_device(), &ctx);
        if (rc != ABC_OK) throw HipError(rc, "abc_ctx_create failed: no usable MI355X (there is no CPU fallback)");
    }
    return ctx;
}
inline void check(int rc) { if (rc != ABC_OK) throw HipError(rc, abc_last_error(context())); }
inline std::vector<abc_prior> to_pod(const std::vector<const Parameter*>& pars) {
    std::vector<abc_prior> p;
    for (const Parameter* q : pars) p.push_back(q->pod());
    return p;
}

// ---- AbcUtil.h:144-153 ----------------------------------------------------------------------------------
inline std::vector<size_t> particle_ranking_PLS(const Mat2D& X_orig, const Mat2D& Y_orig, const Row& target_values,
                                                const float_type training_fraction) {
    if (!((0 < training_fraction) && (training_fraction <= 1))) throw HipError(ABC_ERR_INVALID, "training_fraction");
    const size_t N = X_orig.rows();
    std::vector<uint64_t> idx(N);
    check(abc_particle_ranking_pls(context(), X_orig.data(), Y_orig.data(), target_values.data(), N, X_orig.cols(),
                                   Y_orig.cols(), training_fraction, 0, ABC_RULE_MIN_PRESS, N, idx.data(), nullptr,
                                   nullptr, nullptr, nullptr, nullptr));
    return std::vector<size_t>(idx.begin(), idx.end());
}
inline std::vector<size_t> particle_ranking_simple(const Mat2D& X_orig, const Mat2D& /* Y_orig */,
                                                   const Row& target_values) {
    const size_t N = X_orig.rows();
    std::vector<uint64_t> idx(N);
    check(abc_particle_ranking_simple(context(), X_orig.data(), target_values.data(), N, X_orig.cols(), N, idx.data(),
                                      nullptr));
    return std::vector<size_t>(idx.begin(), idx.end());
}

// ---- AbcUtil.h:155-170 ----------------------------------------------------------------------------------
inline Row calculate_doubled_variance(const Mat2D& params) {
    Row dv(params.cols());
    check(abc_calculate_doubled_variance(context(), params.data(), params.rows(), params.cols(), dv.data()));
    return dv;
}
inline Row weight_predictive_prior(const std::vector<const Parameter*>& /* mpars */, const Mat2D& params) {
    Row w(params.rows());
    check(abc_weight_predictive_prior_uniform(context(), params.rows(), w.data()));
    return w;
}
inline Row weight_predictive_prior(const std::vector<const Parameter*>& mpars, const Mat2D& params,
                                   const Mat2D& prev_params, const Row& prev_weights,
                                   const Row& prev_doubled_variance) {
    Row w(params.rows());
    const std::vector<abc_prior> pr = to_pod(mpars);
    check(abc_weight_predictive_prior(context(), pr.data(), params.data(), params.rows(), params.cols(),
                                      prev_params.data(), prev_params.rows(), prev_weights.data(),
                                      prev_doubled_variance.data(), w.data()));
    return w;
}

// ---- AbcUtil.h:78, 110-137 ------------------------------------------------------------------------------
// setup_mvn_sampler returns an owning gsl_matrix* in the reference (freed by the caller, AbcSmc.cpp:492,502);
// here a value: P x P, lower triangle + diagonal = L.
inline Mat2D setup_mvn_sampler(const Mat2D& params) {
    Mat2D L(params.cols(), params.cols());
    check(abc_setup_mvn_sampler(context(), params.data(), params.rows(), params.cols(), L.data()));
    return L;
}
inline std::vector<size_t> gsl_rng_nonuniform_int(const RNG* rng, const size_t num_samples, const Col& weights) {
    std::vector<uint64_t> idx(num_samples);
    check(abc_sample_posterior(context(), &rng->state, weights.data(), weights.size(), num_samples, idx.data()));
    return std::vector<size_t>(idx.begin(), idx.end());
}
inline Mat2D sample_posterior(const RNG* rng, const size_t num_samples, const Col& weights, const Mat2D& posterior) {
    const std::vector<size_t> rows = gsl_rng_nonuniform_int(rng, num_samples, weights);
    Mat2D out(num_samples, posterior.cols());
    for (size_t j = 0; j < posterior.cols(); j++)
        for (size_t i = 0; i < num_samples; i++) out(i, j) = posterior(rows[i], j);
    return out;
}
inline Mat2D sample_mvn_predictive_priors(const RNG* rng, const size_t num_samples, const Col& weights,
                                          const Mat2D& parameter_prior, const std::vector<const Parameter*>& pars,
                                          const Mat2D& L) {
    Mat2D out(num_samples, parameter_prior.cols());
    const std::vector<abc_prior> pr = to_pod(pars);
    check(abc_sample_mvn_predictive_priors(context(), &rng->state, num_samples, weights.data(), parameter_prior.data(),
                                           parameter_prior.rows(), parameter_prior.cols(), pr.data(), L.data(),
                                           out.data(), nullptr, nullptr));
    return out;
}
inline Mat2D sample_predictive_priors(const RNG* rng, const size_t num_samples, const Col& weights,
                                      const Mat2D& parameter_prior, const std::vector<const Parameter*>& pars,
                                      const Row& doubled_variance) {
    Mat2D out(num_samples, parameter_prior.cols());
    const std::vector<abc_prior> pr = to_pod(pars);
    check(abc_sample_predictive_priors(context(), &rng->state, num_samples, weights.data(), parameter_prior.data(),
                                       parameter_prior.rows(), parameter_prior.cols(), pr.data(),
                                       doubled_variance.data(), out.data(), nullptr, nullptr));
    return out;
}

// AbcUtil.cpp:320-324 (host helper kept for completeness; the device fuses it into the projection)
inline Col euclidean(const Mat2D& sims, const Row& ref) {
    Col d(sims.rows());
    for (size_t i = 0; i < sims.rows(); i++) {
        double s = 0;
        for (size_t k = 0; k < sims.cols(); k++) { const double t = sims(i, k) - ref[k]; s = std::fma(t, t, s); }
        d[i] = std::sqrt(s);
    }
    return d;
}

}  // namespace ABC
#endif
