/*
 * abc_oracle.h -- CPU ORACLE for the AbcSmc per-generation numerical hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path (abcsmc_amd/, the HIP
 * library) never links, imports or calls anything in oracle/.
 *
 * It is a single-threaded C++17 restatement (no Eigen, no GSL, no third-party code) of
 *   /root/reference/src/AbcUtil.cpp:111-158, 320-324, 366-458, 462-488, 528-586
 *   /root/reference/include/AbcSmc/RunningStat.h:16-46
 *   /root/reference/include/AbcSmc/Priors.h:19-110, Parameter.h:52-77
 *   /root/reference/lib/ranker.h:46-53, 66-76 (order / "average" ranks)
 * plus the two third-party dependencies that are ABSENT from /root/reference:
 *   - tjhladish/PLS (git submodule lib/PLS, empty, version unpinned): restated from the
 *     Dayal & MacGregor (1997) improved-kernel PLS algorithm and the 7 call sites
 *     AbcUtil.cpp:413-457;
 *   - GSL >= 2.2 (system package, not installed): taus2, uniform(_pos/_int), polar
 *     Box-Muller gaussian, gaussian_pdf, Walker-alias discrete, multivariate_gaussian
 *     (+_vcov), cholesky_decomp1 restated from GSL's documented algorithms.
 *
 * PINNING STATUS
 *   pinned by reference-owned known answers: ordered() (tests/pls.cpp:15-23),
 *     colwise_z_scores / n-1 stdev (tests/abcutil.cpp:11-21), euclidean (tests/abcutil.cpp:29-38),
 *     dice identities (examples/README.md:29-34).
 *   pinned by published third-party known answers / independent implementations:
 *     taus (GSL manual: seed 123 -> first value 2720986350), PLS vs scikit-learn
 *     PLSRegression(scale=False), covariance/Cholesky vs numpy, gaussian pdf vs scipy.
 *   PARITY UNPINNED by any reference-owned test: PLS::Model / cv_NEW_DATA /
 *     optimal_num_components (incl. the Wilcoxon reduction), weights, covariance,
 *     resampling, perturbation.  See DESIGN.md "Oracle".
 *
 * Conventions: all matrices column-major double, leading dimension = number of rows
 * (Eigen's default for the reference's Mat2D).  Fixed operation order everywhere
 * (explicit fma chains, compiled with -ffp-contract=off) so the HIP kernels can be
 * compared bit-for-bit stage by stage.
 */
#ifndef ABC_ORACLE_H
#define ABC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- prior table (POD form of Priors.h:46-110) -------------------------------- */
enum { ORC_PRIOR_GAUSS = 0, ORC_PRIOR_UNIF_INT = 1, ORC_PRIOR_UNIF_REAL = 2 };
typedef struct {
    int32_t kind;   /* ORC_PRIOR_*                                          */
    int32_t pad_;
    double  a;      /* GAUSS: mean ; UNIF_*: min                            */
    double  b;      /* GAUSS: sd   ; UNIF_*: max                            */
} orc_prior_t;

/* ---- gsl_rng_taus2 state -------------------------------------------------------- */
typedef struct { uint32_t s1, s2, s3; } orc_rng_t;

/* component-selection rule for optimal_num_components */
enum { ORC_RULE_MIN_PRESS = 0, ORC_RULE_WILCOXON = 1 };

/* ---- z-scores (PLS lib; tests/abcutil.cpp:6-21) -------------------------------- */
void orc_col_means(const double* X, size_t n, size_t c, double* mean);
void orc_colwise_stdev(const double* X, size_t n, size_t c, const double* mean, double* sd);
void orc_colwise_z_scores(const double* X, size_t n, size_t c, const double* mean,
                          const double* sd, double* Z);
void orc_z_scores(const double* row, size_t c, const double* mean, const double* sd, double* out);

/* ---- euclidean (AbcUtil.cpp:320-324) and ordered (tests/pls.cpp:6-24) ---------- */
void orc_euclidean(const double* S, size_t n, size_t a, const double* ref, double* dist);
void orc_ordered(const double* v, size_t n, uint64_t* idx);

/* ---- kernel PLS2 (PLS::Model, call site AbcUtil.cpp:443) ------------------------ */
/* method 1 = KERNEL_TYPE1 (t = X r), 2 = KERNEL_TYPE2 (X'X). W,Pm,R are M x A, Q is P x A. */
int orc_pls_fit(const double* X, const double* Y, size_t n, size_t M, size_t P, size_t A,
                int method, double* W, double* Pm, double* Q, double* R);
/* scores = Xnew * R[:, :a]  (n x a) */
void orc_pls_scores(const double* Xnew, size_t n, size_t M, const double* R, size_t a, double* S);
/* PRESS[a-1 + A*j] over new data, a = 1..A (cv_NEW_DATA, AbcUtil.cpp:446) */
void orc_pls_press(const double* Xt, const double* Yt, size_t nt, size_t M, size_t P, size_t A,
                   const double* R, const double* Q, double* press);
/* optimal_num_components (AbcUtil.cpp:447): per-response optimum, returns max over responses */
int orc_pls_optimal_components(const double* Xt, const double* Yt, size_t nt, size_t M, size_t P,
                               size_t A, const double* R, const double* Q, int rule,
                               int32_t* per_response);
double orc_wilcoxon_p(const double* e1, const double* e2, size_t n);
double orc_normalcdf(double z);

/* staged projection+distance with the fixed operation order shared with the HIP kernel */
void orc_project_distance(const double* X, size_t n, size_t M, const double* mean, const double* sd,
                          const double* R, size_t a, const double* obs_scores, double* dist);

/* ---- particle_ranking_PLS / _simple (AbcUtil.cpp:408-458) ---------------------- */
/* max_comp <= 0 -> min(M,P).  Outputs (any may be NULL): idx[N], dist[N], ncomp, R[M*A],
 * Q[P*A], mean[M], sd[M], press[A*P]. Returns 0 on success. */
int orc_particle_ranking_pls(const double* X, const double* Y, const double* obs,
                             size_t N, size_t M, size_t P, double train_frac, int max_comp,
                             int rule, uint64_t* idx, double* dist, int32_t* ncomp,
                             double* R, double* Q, double* mean, double* sd, double* press);
int orc_particle_ranking_simple(const double* X, const double* obs, size_t N, size_t M,
                                uint64_t* idx, double* dist);

/* ---- weights, variances (AbcUtil.cpp:528-586, RunningStat.h) -------------------- */
void   orc_doubled_variance(const double* theta, size_t K, size_t P, double* dv);
double orc_prior_likelihood(const orc_prior_t* pr, double v);
double orc_prior_recast(const orc_prior_t* pr, double v);
int    orc_prior_valid(const orc_prior_t* pr, double v);
double orc_prior_mean(const orc_prior_t* pr);
void   orc_weights_uniform(size_t K, double* w);
/* zero_dv_policy: 0 = declared deviation (factor 0 when dv==0 and values differ),
 *                 1 = reference-literal gsl_ran_gaussian_pdf(x,0) (NaN poison) */
/* extension (no reference counterpart): radial Epanechnikov kernel of the same covariance, see abc_oracle.cpp */
void   orc_weights_epanechnikov(const orc_prior_t* priors, const double* theta, size_t K, const double* theta_prev, size_t Kp,
                                const double* w_prev, const double* dv_prev, size_t P, double* w);
void   orc_weights_importance(const orc_prior_t* priors, const double* theta, size_t K,
                              const double* theta_prev, size_t Kp, const double* w_prev,
                              const double* dv_prev, size_t P, int zero_dv_policy, double* w);

/* ---- MVN sampler setup (AbcUtil.cpp:462-488) ----------------------------------- */
/* L is P x P column-major; lower triangle+diag = Cholesky factor, strict upper = covariance
 * entries (as gsl_linalg_cholesky_decomp1 leaves them). cov_out (optional) = doubled-diagonal
 * covariance before factorisation. Returns 0, or -1 if not positive definite. */
int orc_mvn_setup(const double* theta, size_t K, size_t P, double* L, double* cov_out);

/* ---- GSL restatements -------------------------------------------------------- */
void     orc_rng_set(orc_rng_t* r, unsigned long seed);
uint32_t orc_rng_get(orc_rng_t* r);
double   orc_rng_uniform(orc_rng_t* r);
double   orc_rng_uniform_pos(orc_rng_t* r);
unsigned long orc_rng_uniform_int(orc_rng_t* r, unsigned long n);
double   orc_ran_gaussian(orc_rng_t* r, double sigma);
double   orc_ran_gaussian_pdf(double x, double sigma);
void     orc_discrete_preproc(size_t K, const double* w, double* F, uint64_t* A);
uint64_t orc_discrete_draw(orc_rng_t* r, size_t K, const double* F, const uint64_t* A);

/* ---- resample + perturb (AbcUtil.cpp:111-158, 366-404; Priors.h:19-33) ------- */
void orc_resample(orc_rng_t* r, const double* w, size_t K, size_t n, uint64_t* idx);
/* returns number of per-coordinate fallbacks to the prior mean (INDEPENDENT mode) */
size_t orc_sample_predictive_priors(orc_rng_t* r, size_t n, const double* w, const double* theta,
                                    size_t K, size_t P, const orc_prior_t* priors,
                                    const double* dv, double* out, uint64_t* parent_idx);
/* max_tries == 0 -> unbounded like the reference; returns total number of rejected proposals */
size_t orc_sample_mvn_predictive_priors(orc_rng_t* r, size_t n, const double* w, const double* theta,
                                        size_t K, size_t P, const orc_prior_t* priors,
                                        const double* L, size_t max_tries, double* out,
                                        uint64_t* parent_idx);

/* ---- one whole generation turn-over (AbcSmc.cpp:634-664, 1041-1066, 490-518) -- */
typedef struct {
    size_t N, M, P;          /* particles, metrics, parameters of this set          */
    size_t K, Kp, Nnext;     /* pred-prior size, previous pred-prior size, next set  */
    double train_frac;
    int    max_comp, rule, multivariate, zero_dv_policy;
} orc_generation_cfg_t;
/* outputs: idx[K], w[K], dv[P], L[P*P] (if multivariate), next[Nnext*P], parent[Nnext],
 * seeds[Nnext] (gsl_rng_get per new particle, AbcSmc.cpp:535). theta_prev may be NULL (set 0). */
int orc_generation(const orc_generation_cfg_t* cfg, const double* X, const double* Y,
                   const double* obs, const orc_prior_t* priors, const double* theta_prev,
                   const double* w_prev, const double* dv_prev, orc_rng_t* rng,
                   uint64_t* idx, double* w, double* dv, double* L, double* next,
                   uint64_t* parent, uint64_t* seeds, int32_t* ncomp);

#ifdef __cplusplus
}
#endif
#endif /* ABC_ORACLE_H */
