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

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
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

// ---- RNG: stands in for `const gsl_rng*` of type gsl_rng_taus2 (examples/include/examples.h:10) ----------
struct RNG {
    mutable abc_rng state;
    explicit RNG(unsigned long seed = 0) { abc_rng_set(&state, seed); }
};
inline void rng_set(const RNG* r, unsigned long seed) { abc_rng_set(&r->state, seed); }     // gsl_rng_set
inline unsigned long rng_get(const RNG* r) { return abc_rng_get(&r->state); }                // gsl_rng_get
// the three GSL draws Priors.h uses for set 0 (host side, one-off per fit): gsl_rng_uniform = get / 2^32;
// gsl_rng_uniform_pos redraws a zero; gsl_rng_uniform_int rejects above n * (0xffffffff / n); gsl_ran_gaussian is the
// polar Box-Muller on uniform_pos draws mapped to (-1, 1) [published GSL 2.x algorithms; GSL itself is not in this image]
inline double rng_uniform(const RNG* r) { return abc_rng_get(&r->state) / 4294967296.0; }
inline double rng_uniform_pos(const RNG* r) {
    double x;
    do { x = rng_uniform(r); } while (x == 0.0);
    return x;
}
inline unsigned long rng_uniform_int(const RNG* r, unsigned long n) {
    const unsigned long scale = 0xffffffffUL / n;
    unsigned long k;
    do { k = abc_rng_get(&r->state) / scale; } while (k >= n);
    return k;
}
inline double ran_gaussian(const RNG* r, double sigma) {
    double x, y, r2;
    do {
        x = -1.0 + 2.0 * rng_uniform_pos(r);      // gauss.c: gsl_rng_uniform_pos, a zero output is drawn again
        y = -1.0 + 2.0 * rng_uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    return sigma * y * std::sqrt(-2.0 * std::log(r2) / r2);
}

// ---- ParRNG (ParRNG.h:19-81): the RNG handed to Parameter::sample, plus the odometer over PSEUDO states and
// the cursor over POSTERIOR rows.  One pseudo parameter advances per unlock(); a parameter at its last state
// wraps to 0 and lets the next one advance.
struct Parameter;
struct ParRNG {
    ParRNG(const RNG* rng, const std::vector<const Parameter*>& mpars, size_t posterior_size);
    const RNG* rng() const { return rng_; }
    void unlock() { lock_ = false; }
    size_t pseudo(const Parameter* p) {
        std::pair<size_t, size_t>* st = nullptr;
        for (auto& e : pseudo_) if (e.first == p) st = &e.second;
        if (!st) throw std::logic_error("ParRNG::pseudo: parameter was not registered");
        const size_t ret = st->first;
        if (!lock_) {
            if (st->first < st->second) { st->first++; lock_ = true; } else { st->first = 0; }
        }
        return ret;
    }
    size_t posterior() {
        const size_t ret = post_idx_;
        if (!lock_) post_idx_ = (post_idx_ < post_max_) ? post_idx_ + 1 : 0;
        return ret;
    }

   private:
    const RNG* rng_;
    std::vector<std::pair<const Parameter*, std::pair<size_t, size_t>>> pseudo_;   // (state, last state)
    bool lock_ = false;
    size_t post_idx_ = 0, post_max_;
};
typedef ParRNG PRNG;

// ---- parameters (Parameter.h:36-87, Priors.h:9-110, IndexedPars.h:9-55) ------------------------------------
// The device evaluates likelihood / recast / valid / noise from the POD (`pod()`); the host keeps the names and
// the set-0 `sample`.  The two-argument constructors (no names) are a convenience the reference does not have.
struct Parameter {
    Parameter(const std::string& s = "", const std::string& ss = "", size_t states = 0)
        : name(s), short_name(ss), state_size_(states) {}
    virtual ~Parameter() {}
    std::string get_name() const { return name; }
    std::string get_short_name() const { return short_name; }
    virtual abc_prior pod() const = 0;
    virtual float_type sample(PRNG& prng) const = 0;
    virtual float_type likelihood(float_type pval) const = 0;
    virtual float_type recast(float_type pval) const = 0;
    virtual float_type get_mean() const { return std::nan(""); }
    virtual float_type get_sd() const { return std::nan(""); }
    virtual bool isPosterior() const { return false; }
    bool valid(float_type pval) const { return likelihood(pval) != 0.0; }
    // Parameter.h:52-77 / Priors.h:19-43: recast(N(mu, sigma)) from the shared taus2 stream until it is valid, at most MAX_ATTEMPTS
    // times; then the prior's mean, with the reference's message on stderr (host side: one value at a time, as upstream)
    float_type noise(const RNG* rng, const float_type mu, const float_type sigma, const size_t MAX_ATTEMPTS = 1000) const {
        size_t attempts = 1;
        float_type dev = recast(ran_gaussian(rng, sigma) + mu);
        while (!valid(dev) && (attempts++ < MAX_ATTEMPTS)) dev = recast(ran_gaussian(rng, sigma) + mu);
        if (!valid(dev)) {
            std::fprintf(stderr, "ERROR: failed to draw valid noise from prior %s - returning mean value.\n", get_name().c_str());
            return get_mean();
        }
        return dev;
    }
    size_t state_size() const { return state_size_; }

   private:
    std::string name, short_name;
    size_t state_size_;
};
inline ParRNG::ParRNG(const RNG* rng, const std::vector<const Parameter*>& mpars, size_t posterior_size)
    : rng_(rng), post_max_(posterior_size - 1) {
    for (const Parameter* p : mpars)
        if (!p->isPosterior() && p->state_size() != 0) pseudo_.push_back({p, {0, p->state_size() - 1}});
}

struct GaussianPrior : Parameter {
    float_type mean, sd;
    GaussianPrior(const std::string& nm, const std::string& snm, float_type mn, float_type s)
        : Parameter(nm, snm), mean(mn), sd(s) {}
    GaussianPrior(float_type mn, float_type s) : mean(mn), sd(s) {}
    abc_prior pod() const override { return abc_prior{ABC_PRIOR_GAUSS, 0, mean, sd}; }
    float_type sample(PRNG& prng) const override { return ran_gaussian(prng.rng(), sd) + mean; }
    float_type likelihood(float_type pval) const override {
        const float_type u = (pval - mean) / std::fabs(sd);
        return (1.0 / (std::sqrt(2.0 * M_PI) * std::fabs(sd))) * std::exp(-u * u / 2.0);
    }
    float_type recast(float_type pval) const override { return pval; }
    float_type get_mean() const override { return mean; }
    float_type get_sd() const override { return sd; }
};
struct DiscreteUniformPrior : Parameter {
    long minval, maxval;
    DiscreteUniformPrior(const std::string& nm, const std::string& snm, long mn, long mx)
        : Parameter(nm, snm), minval(mn), maxval(mx) {}
    DiscreteUniformPrior(long mn, long mx) : minval(mn), maxval(mx) {}
    abc_prior pod() const override { return abc_prior{ABC_PRIOR_UNIF_INT, 0, (double)minval, (double)maxval}; }
    float_type sample(PRNG& prng) const override {
        return (float_type)((long)rng_uniform_int(prng.rng(), (unsigned long)(maxval - minval + 1)) + minval);
    }
    float_type likelihood(float_type pval) const override {
        return (pval == recast(pval) && minval <= pval && pval <= maxval) ? 1.0 / (maxval - minval + 1) : 0.0;
    }
    float_type recast(float_type pval) const override { return std::round(pval); }
    float_type get_mean() const override { return (float_type)(maxval + minval) / 2.0; }
    float_type get_sd() const override { return (float_type)(maxval - minval) / std::sqrt(12.0); }
};
struct ContinuousUniformPrior : Parameter {
    float_type minval, maxval;
    ContinuousUniformPrior(const std::string& nm, const std::string& snm, float_type mn, float_type mx)
        : Parameter(nm, snm), minval(mn), maxval(mx) {}
    ContinuousUniformPrior(float_type mn, float_type mx) : minval(mn), maxval(mx) {}
    abc_prior pod() const override { return abc_prior{ABC_PRIOR_UNIF_REAL, 0, minval, maxval}; }
    float_type sample(PRNG& prng) const override { return rng_uniform(prng.rng()) * (maxval - minval) + minval; }
    float_type likelihood(float_type pval) const override {
        return (minval <= pval && pval <= maxval) ? 1.0 / (maxval - minval) : 0.0;
    }
    float_type recast(float_type pval) const override { return pval; }
    float_type get_mean() const override { return (maxval + minval) / 2.0; }
    float_type get_sd() const override { return (maxval - minval) / std::sqrt(12.0); }
};
// PSEUDO / POSTERIOR parameters only occur in projection mode (one set, no weights, no perturbation): asking them
// for a density is an error upstream too (IndexedPars.h:20-28)
struct IndexedPar : Parameter {
    IndexedPar(const std::string& s, const std::string& ss, size_t size) : Parameter(s, ss, size) {
        if (size == 0) throw std::invalid_argument("IndexedPar: empty state set");
    }
    abc_prior pod() const override { throw std::logic_error("IndexedPar " + get_name() + " has no prior density"); }
    float_type likelihood(float_type) const override { throw std::logic_error("likelihood asked of IndexedPar " + get_name()); }
    float_type recast(float_type) const override { throw std::logic_error("recast asked of IndexedPar " + get_name()); }
};
struct PseudoPar : IndexedPar {
    std::vector<float_type> states;
    PseudoPar(const std::string& s, const std::string& ss, const std::vector<float_type>& vals)
        : IndexedPar(s, ss, vals.size()), states(vals) {}
    float_type sample(PRNG& prng) const override { return states[prng.pseudo(this)]; }
};
struct PosteriorPar : IndexedPar {
    PosteriorPar(const std::string& s, const std::string& ss, size_t size) : IndexedPar(s, ss, size) {}
    float_type sample(PRNG& prng) const override { return (float_type)prng.posterior(); }
    bool isPosterior() const override { return true; }
};

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
// Proposals from the reference's own sequential taus2 stream (abc_ctx_set_noise_mode): values, seeds and the final RNG state
// of sample_*_predictive_priors then equal a CPU run of the reference bit for bit; default off (device Philox stream).
inline void set_reference_stream(bool on) { check(abc_ctx_set_noise_mode(context(), on ? ABC_NOISE_REFERENCE_STREAM : ABC_NOISE_DEVICE)); }
inline uint64_t perturb_giveups(bool reset = false) {
    uint64_t n = 0;
    check(abc_perturb_giveups(context(), &n, reset ? 1 : 0));
    return n;
}
// What the speculation on the component count has cost this context (abc_generation_repeats): whole generations under the Wilcoxon
// rule rank on the count the fit wrote while the reduction runs beside them; .first = how often the reduction lowered the largest
// count and the ranking was repeated, .second = how often a generation started over (degenerate selection, a cascade that gave up)
inline std::pair<uint64_t, uint64_t> generation_repeats(bool reset = false) {
    uint64_t r = 0, g = 0;
    check(abc_generation_repeats(context(), &r, &g, reset ? 1 : 0));
    return {r, g};
}
// How many PLS components particle_ranking_PLS keeps (AbcUtil.cpp:447-449: `PLS::optimal_num_components(em).maxCoeff()`; the PLS
// library is not in the reference tree).  SURVEY A.2, the only specification of it at hand, describes upstream as: per response
// the component count of least PRESS, REDUCED to the smallest count whose validation errors a two-sided Wilcoxon signed-rank
// test (alpha = 0.1) cannot tell from it.  That is the drop-in default here (ABC_RULE_WILCOXON); ABC_RULE_MIN_PRESS keeps the
// plain argmin (about 1 ms less per million particles; never fewer components).  `max_components` = 0 means min(#metrics,
// #parameters), the unverifiable default of `PLS::Model plsm(X, Y)` (AbcUtil.cpp:443).
inline int& component_rule_ref() { static int r = ABC_RULE_WILCOXON; return r; }
inline int& max_components_ref() { static int a = 0; return a; }
inline void set_component_rule(int rule) {
    if (rule != ABC_RULE_MIN_PRESS && rule != ABC_RULE_WILCOXON) throw HipError(ABC_ERR_INVALID, "set_component_rule: unknown rule");
    component_rule_ref() = rule;
}
inline int component_rule() { return component_rule_ref(); }
inline void set_max_components(int a) { max_components_ref() = a < 0 ? 0 : a; }
inline std::vector<abc_prior> to_pod(const std::vector<const Parameter*>& pars) {
    std::vector<abc_prior> p;
    for (const Parameter* q : pars) p.push_back(q->pod());
    return p;
}

// ---- several GPUs of one node (include/abcsmc_hip.h, "Multi-GPU"; the reference has no multi-device path) ----------
// use_devices({0, 1, ...}) joins the listed GPUs with RCCL communicators (abc_ctx_create_multi); rank_and_weight() then
// runs the rank + truncate + doubled variance + weights of one finished set (AbcSmc.cpp:634-664, 1041-1066) with the set's
// rows sharded over them (abc_generation_multi).  Without it every call below runs on the one default device.
inline std::vector<abc_ctx*>& multi_contexts() { static std::vector<abc_ctx*> v; return v; }
inline void use_devices(const std::vector<int>& devices) {
    for (abc_ctx* c : multi_contexts()) abc_ctx_destroy(c);
    multi_contexts().clear();
    if (devices.empty()) return;
    std::vector<abc_ctx*> v(devices.size(), nullptr);
    const int rc = abc_ctx_create_multi(devices.data(), (int)devices.size(), v.data());
    if (rc != ABC_OK) throw HipError(rc, "abc_ctx_create_multi failed (RCCL not loadable, or a listed GPU is not usable)");
    multi_contexts() = v;
}
inline bool multi_device() { return !multi_contexts().empty(); }
struct RankedSet {
    std::vector<size_t> idx;      // the K best particles, ascending distance
    Row dist;                     // their distances
    Mat2D theta;                  // their parameter rows (K x P)
    Row weights, doubled_variance;
    int ncomp = 0;
};
inline RankedSet rank_and_weight(const Mat2D& X, const Mat2D& Y, const Row& obs, float_type training_fraction, size_t K,
                                 const std::vector<const Parameter*>& mpars, const Mat2D* prev_params = nullptr,
                                 const Row* prev_weights = nullptr, const Row* prev_doubled_variance = nullptr,
                                 int rule = -1 /* -1: component_rule() */, int max_components = -1 /* -1: the process-wide setting */) {
    if (!multi_device()) throw HipError(ABC_ERR_INVALID, "rank_and_weight: call use_devices() first");
    const size_t N = X.rows(), M = X.cols(), P = Y.cols();
    RankedSet out;
    std::vector<uint64_t> idx(K);
    out.dist.assign(K, 0.0); out.theta = Mat2D(K, P); out.weights.assign(K, 0.0); out.doubled_variance.assign(P, 0.0);
    const std::vector<abc_prior> pr = to_pod(mpars);
    abc_generation_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.N = N; cfg.M = M; cfg.P = P; cfg.K = K; cfg.Kp = prev_params ? prev_params->rows() : 0; cfg.Nnext = 0;
    cfg.train_frac = training_fraction; cfg.max_comp = max_components < 0 ? max_components_ref() : max_components;
    cfg.rule = rule < 0 ? component_rule() : rule; cfg.multivariate = 0;
    abc_generation_io io;
    memset(&io, 0, sizeof(io));
    io.X = X.data(); io.Y = Y.data(); io.obs = obs.data(); io.priors = pr.data();
    if (prev_params) { io.theta_prev = prev_params->data(); io.w_prev = prev_weights->data(); io.dv_prev = prev_doubled_variance->data(); }
    io.idx = idx.data(); io.dist = out.dist.data(); io.theta = out.theta.data(); io.w = out.weights.data();
    io.dv = out.doubled_variance.data();
    abc_rng unused = {0, 0, 0};
    int32_t nc = 0;
    const int rc = abc_generation_multi(multi_contexts().data(), (int)multi_contexts().size(), &cfg, &io, &unused, &nc);
    if (rc != ABC_OK) throw HipError(rc, abc_last_error(multi_contexts()[0]));
    out.idx.assign(idx.begin(), idx.end());
    out.ncomp = nc;
    return out;
}

// ---- AbcUtil.h:144-153 ----------------------------------------------------------------------------------
// (rule / max_components: what an AbcSmc object was configured with -- the process-wide set_component_rule / set_max_components are
// only the defaults of the four-argument form the reference has)
inline std::vector<size_t> particle_ranking_PLS(const Mat2D& X_orig, const Mat2D& Y_orig, const Row& target_values,
                                                const float_type training_fraction, int rule, int max_components) {
    if (!((0 < training_fraction) && (training_fraction <= 1))) throw HipError(ABC_ERR_INVALID, "training_fraction");
    const size_t N = X_orig.rows();
    std::vector<uint64_t> idx(N);
    check(abc_particle_ranking_pls(context(), X_orig.data(), Y_orig.data(), target_values.data(), N, X_orig.cols(),
                                   Y_orig.cols(), training_fraction, max_components, rule, N, idx.data(), nullptr,
                                   nullptr, nullptr, nullptr, nullptr));
    return std::vector<size_t>(idx.begin(), idx.end());
}
inline std::vector<size_t> particle_ranking_PLS(const Mat2D& X_orig, const Mat2D& Y_orig, const Row& target_values,
                                                const float_type training_fraction) {
    return particle_ranking_PLS(X_orig, Y_orig, target_values, training_fraction, component_rule(), max_components_ref());
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
// `seeds` (optional, not in the reference signature): the per-particle simulator seeds AbcSmc.cpp:535 draws with
// gsl_rng_get after the proposals; when asked for they are produced by the same call and the stream advances
// past them (DESIGN.md "Declared deviations": the Gaussian noise does not consume the taus2 stream).
inline Mat2D sample_mvn_predictive_priors(const RNG* rng, const size_t num_samples, const Col& weights,
                                          const Mat2D& parameter_prior, const std::vector<const Parameter*>& pars,
                                          const Mat2D& L, std::vector<unsigned long>* seeds = nullptr) {
    Mat2D out(num_samples, parameter_prior.cols());
    const std::vector<abc_prior> pr = to_pod(pars);
    std::vector<uint64_t> sd(seeds ? num_samples : 0);
    check(abc_sample_mvn_predictive_priors(context(), &rng->state, num_samples, weights.data(), parameter_prior.data(),
                                           parameter_prior.rows(), parameter_prior.cols(), pr.data(), L.data(),
                                           out.data(), nullptr, seeds ? sd.data() : nullptr));
    if (seeds) seeds->assign(sd.begin(), sd.end());
    return out;
}
inline Mat2D sample_predictive_priors(const RNG* rng, const size_t num_samples, const Col& weights,
                                      const Mat2D& parameter_prior, const std::vector<const Parameter*>& pars,
                                      const Row& doubled_variance, std::vector<unsigned long>* seeds = nullptr) {
    Mat2D out(num_samples, parameter_prior.cols());
    const std::vector<abc_prior> pr = to_pod(pars);
    std::vector<uint64_t> sd(seeds ? num_samples : 0);
    check(abc_sample_predictive_priors(context(), &rng->state, num_samples, weights.data(), parameter_prior.data(),
                                       parameter_prior.rows(), parameter_prior.cols(), pr.data(),
                                       doubled_variance.data(), out.data(), nullptr, seeds ? sd.data() : nullptr));
    if (seeds) seeds->assign(sd.begin(), sd.end());
    return out;
}

// ---- AbcUtil.h:80-91: the per-row samplers behind sample_predictive_priors / sample_mvn_predictive_priors ------------------------
// The reference only calls them from those two (AbcUtil.cpp:386, 401); a drop-in header carries every declaration of
// AbcUtil.h:78-172, so here they are, with the reference's semantics on the reference's stream: ONE row from the shared taus2
// stream on the host (the device path draws whole sets: abc_sample_*).  Bit for bit the oracle's rows (tests/test_gpu_parity.py).
// AbcUtil.cpp:145-158: per coordinate Parameter::noise (<= 1000 tries, then the prior's mean)
inline Row gsl_ran_trunc_normal(const RNG* rng, const std::vector<const Parameter*> _model_pars, const Row& mu,
                                const Row& sigma_squared) {
    Row res(sigma_squared.size(), 0.0);
    for (size_t parIdx = 0; parIdx < sigma_squared.size(); parIdx++)
        res[parIdx] = _model_pars[parIdx]->noise(rng, mu[parIdx], std::sqrt(sigma_squared[parIdx]));
    return res;
}
// AbcUtil.cpp:122-143: x = mu + L z (z iid N(0,1) in coordinate order: gsl_ran_multivariate_gaussian), every coordinate recast,
// the whole draw again unless all are valid.  L: P x P, the lower triangle + diagonal of setup_mvn_sampler's factor.  The
// reference retries without bound; here ABC_MVN_MAX_TRIES draws, then HipError (the device path falls back to the parent and
// counts it, abc_perturb_giveups)
constexpr size_t ABC_MVN_MAX_TRIES = 16384;
inline Row gsl_ran_trunc_mv_normal(const RNG* rng, const std::vector<const Parameter*> _model_pars, const Row& mu, const Mat2D& L) {
    const size_t npar = _model_pars.size();
    Row par_values(npar, 0.0), z(npar, 0.0);
    for (size_t tries = 0; tries < ABC_MVN_MAX_TRIES; tries++) {
        bool success = true;
        for (size_t i = 0; i < npar; i++) z[i] = ran_gaussian(rng, 1.0);                     // gsl_ran_ugaussian
        for (size_t parIdx = 0; success && (parIdx < npar); parIdx++) {
            double x = 0.0;                                                                  // dtrmv, lower, non-unit: row parIdx of L times z
            for (size_t k = 0; k <= parIdx; k++) x += L(parIdx, k) * z[k];
            par_values[parIdx] = _model_pars[parIdx]->recast(x + mu[parIdx]);
            success = _model_pars[parIdx]->valid(par_values[parIdx]);
        }
        if (success) return par_values;
    }
    throw HipError(ABC_ERR_INVALID, "gsl_ran_trunc_mv_normal: no valid draw in 16384 tries");
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

// ---- host-side helpers of the shell (one-off or O(K) work; not on the per-generation device path) --------
// AbcUtil.cpp:490-526: set 0 (and projection mode): every non-posterior parameter is sampled in order, then the
// posterior cursor; the posterior columns are filled from the looked-up rows afterwards.
inline Mat2D sample_priors(const RNG* rng, const size_t num_samples, const Mat2D& posterior,
                           const std::vector<const Parameter*>& mpars, std::vector<size_t>& post_ranks) {
    ParRNG par_rng(rng, mpars, posterior.rows());
    Mat2D par_samples(num_samples, mpars.size());
    std::vector<size_t> nonpost, post;
    for (size_t j = 0; j < mpars.size(); j++) (mpars[j]->isPosterior() ? post : nonpost).push_back(j);
    if (post.size() != posterior.cols()) throw std::invalid_argument("sample_priors: posterior columns != POSTERIOR parameters");
    if (!post.empty()) post_ranks.resize(num_samples);
    for (size_t i = 0; i < num_samples; i++) {
        par_rng.unlock();
        for (size_t j : nonpost) par_samples(i, j) = mpars[j]->sample(par_rng);
        if (!post.empty()) post_ranks[i] = (size_t)mpars[post[0]]->sample(par_rng);
    }
    for (size_t c = 0; c < post.size(); c++)
        for (size_t i = 0; i < num_samples; i++) par_samples(i, post[c]) = posterior(post_ranks[i], c);
    return par_samples;
}
inline Mat2D select_rows(const Mat2D& m, const std::vector<size_t>& rows) {     // Eigen's m(rows, all)
    Mat2D out(rows.size(), m.cols());
    for (size_t j = 0; j < m.cols(); j++)
        for (size_t i = 0; i < rows.size(); i++) out(i, j) = m(rows[i], j);
    return out;
}
inline Row col_means(const Mat2D& m) {
    Row mu(m.cols(), 0.0);
    for (size_t j = 0; j < m.cols(); j++) {
        double s = 0;
        for (size_t i = 0; i < m.rows(); i++) s += m(i, j);
        mu[j] = m.rows() ? s / (double)m.rows() : 0.0;
    }
    return mu;
}
inline float_type median(Col data) {                                            // AbcUtil.cpp:46-62
    if (data.empty()) throw std::invalid_argument("median of an empty column");
    std::sort(data.begin(), data.end());
    const size_t n = data.size();
    return (n % 2 == 0) ? (data[n / 2 - 1] + data[n / 2]) / 2 : data[n / 2];
}
inline float_type calculate_nrmse(const Mat2D& posterior_mets, const Row& observed) {   // AbcUtil.cpp:326-345
    const Row sim = col_means(posterior_mets);
    double acc = 0;
    for (size_t i = 0; i < sim.size(); i++) {
        double expected = (std::fabs(observed[i]) + std::fabs(sim[i])) / 2.0;
        if (sim[i] == observed[i]) expected = 1;
        const double d = (sim[i] - observed[i]) / expected;
        acc += d * d;
    }
    return std::sqrt(acc / (double)sim.size());
}
inline float_type logistic(float_type t) { return 1.0 / (1.0 + std::exp(-t)); }

}  // namespace ABC
#endif
