/*
 * abcsmc_hip.h -- C ABI of the MI355X (gfx950) implementation of AbcSmc's per-generation
 * numerical hot path.  Plain pointers and sizes only; no C++/torch types cross this line.
 *
 * The reference has no FFI for its numerics: the boundary this library replaces is the set of
 * C++ free functions in namespace ABC declared in /root/reference/include/AbcSmc/AbcUtil.h:78-172
 * and called from /root/reference/src/AbcSmc.cpp:490-518, 634-640, 1041-1066.  Each entry point
 * below names the declaration it stands in for.  A C++ facade with the reference's own
 * signatures sits on top (abcsmc_amd/cxx/AbcUtilHip.hpp); INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - all matrices are double, COLUMN-MAJOR with leading dimension = number of rows (Eigen's
 *     default layout for the reference's Mat2D): each metric / parameter is one contiguous
 *     particle-major vector.
 *   - functions WITHOUT the _dev suffix take HOST pointers (drop-in for the reference call
 *     sites); functions WITH _dev take DEVICE pointers (HBM-resident data, used by bench.py and
 *     the multi-GPU driver) and run asynchronously on the context's stream.
 *   - every function returns ABC_OK (0) or a negative abc_status; abc_last_error() gives text; there is no positive status
 *     (`if (rc)` is a valid failure test).  A generation whose outputs are complete and valid but in which the perturbation gave
 *     up on some proposals -- they are their (valid) parents or a prior mean; the reference would still be retrying,
 *     AbcUtil.cpp:132 -- returns ABC_OK and says so through abc_generation_giveups / abc_perturb_giveups.
 *     Nothing here calls exit() or throws (the reference exits/aborts).
 *   - one context per GPU and per host thread; calls on one context are serialised.
 *   - Diagnostic environment switches: the library reads NO environment variable unless ABC_DIAG=1 is set; beside it, the
 *     test / A-B switches listed in INTEGRATION.md section 6 act (ABC_WS_POISON: workspace pre-filled with a byte;
 *     ABC_ALIAS_FORCE_FAIL: the device alias build reports failure; kernel-variant and stream-orchestration A/B switches).
 *     None changes a result except by forcing a documented fall-back path.
 */
#ifndef ABCSMC_HIP_H
#define ABCSMC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct abc_ctx abc_ctx;

typedef enum {
    ABC_OK = 0,
    ABC_ERR_INVALID = -1,       /* bad argument (reference: assert / exit)                   */
    ABC_ERR_HIP = -2,           /* HIP runtime failure                                        */
    ABC_ERR_NOT_SPD = -3,       /* covariance not positive definite (reference: GSL abort)    */
    ABC_ERR_UNSUPPORTED = -4,   /* size outside what the kernels are built for                */
    ABC_ERR_NOMEM = -5,
    ABC_ERR_COMM = -6           /* RCCL / caller-supplied collective failed, or librccl is missing */
} abc_status;

/* POD form of the concrete priors in Priors.h:46-110 (likelihood / recast / valid / mean). */
enum { ABC_PRIOR_GAUSS = 0, ABC_PRIOR_UNIF_INT = 1, ABC_PRIOR_UNIF_REAL = 2 };
typedef struct {
    int32_t kind;     /* ABC_PRIOR_*                               */
    int32_t pad_;
    double  a;        /* GAUSS: mean ; UNIF_*: min                 */
    double  b;        /* GAUSS: sd   ; UNIF_*: max                 */
} abc_prior;

/* gsl_rng_taus2 state (examples/include/examples.h:10); abc_rng_set == gsl_rng_set. */
typedef struct { uint32_t s1, s2, s3; } abc_rng;

/* PLS component-selection rule ([PLS] optimal_num_components, AbcUtil.cpp:447-449). */
enum { ABC_RULE_MIN_PRESS = 0, ABC_RULE_WILCOXON = 1 };

/* ---- context ----------------------------------------------------------------------- */
int  abc_ctx_create(int device, abc_ctx** out);
void abc_ctx_destroy(abc_ctx* ctx);
const char* abc_last_error(const abc_ctx* ctx);
/* Run on an existing hipStream_t (e.g. torch's current stream; NULL = HIP's default stream).
 * A new context runs on a private non-blocking stream until this is called;
 * abc_ctx_use_own_stream switches back to it. Both synchronise the stream being left. */
int  abc_ctx_set_stream(abc_ctx* ctx, void* hip_stream);
int  abc_ctx_use_own_stream(abc_ctx* ctx);
int  abc_ctx_synchronize(abc_ctx* ctx);
int  abc_version(void);
/* Which kernel evaluates the O(K K' P) pair sums of weight_predictive_prior (AbcUtil.cpp:556-581).
 * ABC_KDE_AUTO (default): 5 <= P <= 64 parameters run the split-operand kernel (pair dot products on the f16 matrix
 * pipe from three limbs per coordinate, 1.4e-8 rms / 7e-8 max (P <= 16), 2e-8 / 9e-8 (P <= 32), ~3e-8 / 1.3e-7 (P <= 64) absolute
 * error in the base-2 exponent of a term; the terms are evaluated in f32, sixteen at a time, and summed in fp64: 5e-8 rms / 2e-7
 * max relative on such a partial sum; error budget of a weight that one term dominates (all of it at its worst): 5e-7 up to 16 parameters, 5.5e-7 at 17..32, 8e-7 at 33..64; the fixed-seed tests hold 2.5e-7 / 3e-7 / 7e-7 (largest
 * error over ~15 000 weights per parameter count, every count from 5 to 64: 2.2e-7 / 2.4e-7 / 4.1e-7, profiles/history/r03_kde_accuracy.json; in 980 whole generations at random shapes 3.1e-7 / 3.1e-7 / 5.2e-7, profiles/history/r03_generation_fuzz.json),
 * budget 1e-6; rows it cannot represent exactly are summed in fp64, sets it cannot take fall back by themselves); P < 5 and
 * 64 < P run the fp64 kernel (at 64 parameters it is 10 times slower: 63.5 against 6.35 ms per 1e10 pairs; round 6: above 32
 * parameters the previous tiles are staged in LDS -- the same matrix steps on the same operands, the same sums -- and 33..48
 * parameters take three 16-parameter chunks instead of four).
 * ABC_KDE_FP64: always the fp64 vector kernel (<= 1e-12 relative). */
enum { ABC_KDE_AUTO = 0, ABC_KDE_FP64 = 1 };
int  abc_ctx_set_kde_mode(abc_ctx* ctx, int mode);
/* Which kernel takes the sufficient statistics (column sums, Gram blocks) of WIDE sets: 97..160 columns (metrics + parameters), and,
 * from 2 000 000 rows, 81..96 columns whose last one or two 16-column blocks hold parameters only (80 + 16, 64 + 32: configs[3]).
 * The byte-limb kernel (csrc/gram.hip: k_gram_i8, the i8 matrix pipe) rounds every value to a 32-bit fixed-point grid per column.
 * Row counts, column sums and the Gram DIAGONAL stay exact; an off-diagonal entry obeys ONE error model (tests/_gram_model.py, the
 * bound the fixed tests and tests/fuzz/wide_gram_fuzz.py assert per entry):
 *     |G_ab - exact| <= 2^-32 (4 range_a range_b sqrt(rows) + range_a |S_b| + range_b |S_a|),
 * range_c = the column's grid (4 x a robust size of |x - shift| among 4096 sampled rows, rounded up to a power of two: 10 .. 19
 * sigma for Gaussian-like columns), S_c = the partition's sum of x - shift_c: 2e-10 of sqrt(G_aa G_bb) at 2e5 rows for Gaussian-like
 * columns (measured: 0.06 .. 0.1 of the bound), 1.5e-9 beside a column whose mass sits in one value.  That noise is harmless at the
 * Gram's scale but is AMPLIFIED in the loadings of components that fit noise (cross products sqrt(rows) below that scale, close
 * eigenvalues): tests/fuzz/wide_model_fuzz.py holds every USED loading column against the oracle's fit and found 4.3e-6 (fp64
 * kernels: 4e-10) at 66 000 training rows x 29 responses x 30 components, against the 1e-6 of BASELINE.json; with 450 000 rows and
 * more in each partition the worst of 64 fuzzed sets is 2.6e-7 (profiles/r06_wide_model_fuzz*.json).  Hence:
 * ABC_GRAM_AUTO (default): the byte-limb kernel only where EVERY non-empty partition (training rows, validation rows) OF THE WHOLE
 *   SET has at least 400 000 rows -- the sharded generation decides from N_total and the training fraction, so every rank and the
 *   unsharded run of the same set take the same kernel; a rank whose own shard the kernel cannot take (odd row count, columns not
 *   16-byte aligned, fewer than 4096 rows) accumulates ITS rows in fp64 --, the fp64 kernels everywhere else.  Rows outside a
 *   column's grid ("far" rows) are summed in fp64 by a serial side kernel: sets with heavy tails in many rows should use ABC_GRAM_FP64.
 * ABC_GRAM_FP64: the fp64 kernels always (products on the fp64 matrix pipe, ~1e-15 of sqrt(G_aa G_bb)); 1.5x the time of the i8
 *   kernel at 1e6 rows x 144 columns.  With it the sharded generation's statistics equal the unsharded ones to rounding of the order
 *   of summation (distances within 1e-12); under the byte-limb kernel the two differ by the fixed-point noise above (each rank
 *   rounds on its own grid), selection indices still agree up to near-ties.
 * ABC_GRAM_I8: the byte-limb kernel from 200 000 rows in the whole set (round 5's default; A/B runs, the kernel's own tests): Gram
 *   entries within the model above, loadings NOT held to 1e-6. */
enum { ABC_GRAM_AUTO = 0, ABC_GRAM_FP64 = 1, ABC_GRAM_I8 = 2 };
int  abc_ctx_set_gram_mode(abc_ctx* ctx, int mode);
/* Which of the two kernels produced the pair sums of the most recent weight call on this context (synchronises). */
enum { ABC_KDE_RAN_NONE = 0, ABC_KDE_RAN_FP64 = 1, ABC_KDE_RAN_SPLIT = 2 };
int  abc_kde_last_kernel(abc_ctx* ctx, int* which);
/* Kernel of the importance weights (weight_predictive_prior, set > 0).  ABC_WEIGHT_GAUSSIAN (default) is the reference's product
 * of Gaussian factors (AbcUtil.cpp:572-576).  ABC_WEIGHT_EPANECHNIKOV is an EXTENSION with no reference counterpart (the
 * reference only mentions the name in a comment, AbcUtil.cpp:476; BASELINE.json's north_star asks for it): the radial
 * Epanechnikov kernel of the same covariance, K = max(0, 1 - r2 / (P' + 4)), r2 = sum_p (theta_ip - theta'_jp)^2 / dv'_p over the
 * P' parameters with dv'_p != 0; a particle without support among the previous ones gets weight 0.  fp64 vector kernel. */
enum { ABC_WEIGHT_GAUSSIAN = 0, ABC_WEIGHT_EPANECHNIKOV = 1 };
int  abc_ctx_set_weight_kernel(abc_ctx* ctx, int kernel);
/* Which stream the Gaussian noise of the proposals comes from (sample_*_predictive_priors, abc_generation_dev).
 * ABC_NOISE_DEVICE (default): counter-based Philox stream keyed by (rng state, draw, attempt), evaluated on the device -- same
 *   distribution as the reference, different numbers; the simulator seeds are the taus2 outputs right after the resampling draws.
 * ABC_NOISE_REFERENCE_STREAM: the shared taus2 stream consumed EXACTLY as the reference does (AbcUtil.cpp:122-158,
 *   Priors.h:19-43: polar Box-Muller on gsl_rng_uniform_pos per coordinate, whole-vector / per-coordinate rejection, then one
 *   gsl_rng_get per row for the seeds, AbcSmc.cpp:535): proposals, seeds and the final rng state equal a CPU run of the
 *   reference bit for bit.  Inherently sequential -- a host loop inside the library, ~50 ns per normal; not available to the
 *   row-sliced entry points (abc_perturb_dev, abc_generation_sharded_dev). */
enum { ABC_NOISE_DEVICE = 0, ABC_NOISE_REFERENCE_STREAM = 1 };
int  abc_ctx_set_noise_mode(abc_ctx* ctx, int mode);
/* Where the Walker alias table of the resampling step (gsl_ran_discrete_preproc, AbcUtil.cpp:111-120) is built.
 * ABC_ALIAS_DEVICE (default): on the GPU, as two verified prefix scans (csrc/alias_dev.hip) -- the same table bit for bit; when
 *   its verification does not hold (or the weights are outside its grid: negative, non-finite, spread over more than 2^44) the
 *   host builds the table instead, and abc_alias_stats counts it.
 * ABC_ALIAS_HOST: always on the host (the sequential algorithm, the GPU idle meanwhile): round 2's path, kept for A/B runs. */
enum { ABC_ALIAS_DEVICE = 0, ABC_ALIAS_HOST = 1 };
int  abc_ctx_set_alias_mode(abc_ctx* ctx, int mode);
/* device builds queued / of those found unusable (rebuilt on the host) since the context was created or the last reset */
int  abc_alias_stats(abc_ctx* ctx, uint64_t* device_builds, uint64_t* host_fallbacks, int reset);
/* The table itself, for inspection: F (K doubles, GSL's KNUTH_CONVENTION applied: (F[k] + k) / K) and A (K uint64) from weights in
 * host memory, built as the context's alias mode says; *on_device = 1 when the device build was used (0: host, incl. fallback). */
int  abc_alias_table(abc_ctx* ctx, const double* w, size_t K, double* F, uint64_t* A, int* on_device);
/* Proposals the perturbation gave up on since the context was created (or since the last reset): multivariate rows whose
 * 16384 whole-vector draws were all rejected (the valid parent is emitted; the reference would retry for ever,
 * AbcUtil.cpp:132) plus independent-noise coordinates that fell back to the prior mean after 1000 tries (the reference prints
 * an error line per fallback, Priors.h:27-29).  Synchronises. */
int  abc_perturb_giveups(abc_ctx* ctx, uint64_t* count, int reset);
/* ... and how many of them the most recent abc_generation_dev / abc_generation_sharded_dev call on this context added (the sharded
 * call: those of THIS rank's slice of the proposals; 0 after a clean generation; the value the host already holds at the call's
 * end: no synchronisation). */
int  abc_generation_giveups(const abc_ctx* ctx, uint64_t* count);
/* What the speculation on the component count cost since the context was created (or the last reset).  A whole generation under
 * ABC_RULE_WILCOXON ranks on the count the fit wrote while the reduction of AbcUtil.cpp:447-449 runs beside it;
 * when the reduction lowers the LARGEST per-response count -- the one the distances use, AbcUtil.cpp:449 -- the projection, the
 * selection and the gather run once more with it (*ranking_repeats), and when the reduction itself gives up on its fast path, or a
 * degenerate selection has to be redone by radix select, the generation starts over (*generation_repeats).  Both 0 on clean
 * responses; neither changes a result.  No synchronisation. */
int  abc_generation_repeats(abc_ctx* ctx, uint64_t* ranking_repeats, uint64_t* generation_repeats, int reset);
/* Optional per-stage timing: HIP events recorded on the context's stream around each stage
 * (and around the k_gram / k_kde kernels alone).  abc_timing_read synchronises, then returns the
 * number of stages; names[i] is a static string, ms[i] the accumulated device time, host_ms[i]
 * accumulated host-side time (alias-table build), count[i] the number of launches; reset != 0 clears.
 * on: 0 = off, 1 = every stage (an event pair per stage: ~10 us of host / dispatch gap each, i.e. ~0.15 ms per generation),
 *     2 = only the brackets around the k_gram and k_kde kernels (what bench.py's roofline needs inside its timed region). */
int  abc_timing_enable(abc_ctx* ctx, int on);
int  abc_timing_read(abc_ctx* ctx, const char** names, double* ms, double* host_ms, long long* count,
                     int max_stages, int reset);
/* What an event pair around a single kernel reports beyond the kernel's own execution time (dispatch and
 * end-of-kernel release latencies), measured with empty kernels on the context's stream.  bench.py reports the
 * k_gram duration both raw and with this subtracted; rocprofv3's kernel duration is the arbiter. */
int  abc_timing_overhead(abc_ctx* ctx, int reps, double* overhead_ms);

/* ---- RNG (gsl_rng_set / gsl_rng_get on taus2) ------------------------------------------- */
void     abc_rng_set(abc_rng* r, unsigned long seed);
uint32_t abc_rng_get(abc_rng* r);
/* advance the state by n outputs in O(log n) (taus2 is GF(2)-linear) */
void     abc_rng_jump(abc_rng* r, uint64_t n);

/* ======================================================================================== */
/* HOST-pointer entry points (drop-in for the AbcUtil.h free functions)                     */
/* ======================================================================================== */

/* ABC::particle_ranking_PLS (AbcUtil.h:149-153, AbcUtil.cpp:423-458).
 * X: N x M metrics, Y: N x P parameters, obs: M observed metrics.  Returns the first K entries
 * of the ascending-distance ordering (the caller keeps only those: AbcSmc.cpp:645-646); K = N
 * gives the whole vector the reference returns.  max_comp <= 0 -> min(M,P).
 * Optional outputs (NULL to skip): dist[K], ncomp, R[M*A], mean[M], sd[M]. */
int abc_particle_ranking_pls(abc_ctx* ctx, const double* X, const double* Y, const double* obs,
                             size_t N, size_t M, size_t P, double train_frac, int max_comp,
                             int rule, size_t K, uint64_t* idx, double* dist, int32_t* ncomp,
                             double* R, double* mean, double* sd);

/* ABC::particle_ranking_simple (AbcUtil.h:144-147, AbcUtil.cpp:408-421) */
int abc_particle_ranking_simple(abc_ctx* ctx, const double* X, const double* obs, size_t N,
                                size_t M, size_t K, uint64_t* idx, double* dist);

/* ABC::calculate_doubled_variance (AbcUtil.h:168-170, AbcUtil.cpp:528-537); theta K x P */
int abc_calculate_doubled_variance(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* dv);

/* ABC::weight_predictive_prior, set 0 (AbcUtil.h:155-158, AbcUtil.cpp:539-545) */
int abc_weight_predictive_prior_uniform(abc_ctx* ctx, size_t K, double* w);

/* ABC::weight_predictive_prior, set > 0 (AbcUtil.h:160-166, AbcUtil.cpp:547-586): Gaussian-kernel
 * importance weights, L2-normalised.  theta K x P, theta_prev Kp x P, w_prev[Kp], dv_prev[P]. */
int abc_weight_predictive_prior(abc_ctx* ctx, const abc_prior* priors, const double* theta, size_t K,
                                size_t P, const double* theta_prev, size_t Kp, const double* w_prev,
                                const double* dv_prev, double* w);

/* ABC::setup_mvn_sampler (AbcUtil.h:128-130, AbcUtil.cpp:462-488): L is P x P column-major, lower
 * triangle + diagonal = Cholesky factor of the doubled-diagonal covariance, strict upper triangle =
 * covariance entries (as gsl_linalg_cholesky_decomp1 leaves them). */
int abc_setup_mvn_sampler(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* L);

/* ABC::gsl_rng_nonuniform_int / sample_posterior (AbcUtil.h:78, 110-114; AbcUtil.cpp:111-120,
 * 366-375): n weighted draws of parent rows; consumes exactly n outputs of rng (bit-exact with
 * gsl_ran_discrete on taus2) and advances it. */
int abc_sample_posterior(abc_ctx* ctx, abc_rng* rng, const double* w, size_t K, size_t n, uint64_t* idx);

/* ABC::sample_mvn_predictive_priors (AbcUtil.h:132-137, AbcUtil.cpp:391-404, 122-143) and
 * ABC::sample_predictive_priors (AbcUtil.h:121-126, AbcUtil.cpp:377-389, 145-158).
 * out: n x P proposals; parent (optional): n parent rows; seeds (optional): n simulator seeds
 * (AbcSmc.cpp:535).  Parent indices are bit-exact with the reference stream; the Gaussian noise
 * comes from a counter-based generator keyed by (rng state, particle), i.e. it is distributed as
 * the reference's but is not the same stream (DESIGN.md "Declared deviations"). */
int abc_sample_mvn_predictive_priors(abc_ctx* ctx, abc_rng* rng, size_t n, const double* w,
                                     const double* theta, size_t K, size_t P, const abc_prior* priors,
                                     const double* L, double* out, uint64_t* parent, uint64_t* seeds);
int abc_sample_predictive_priors(abc_ctx* ctx, abc_rng* rng, size_t n, const double* w,
                                 const double* theta, size_t K, size_t P, const abc_prior* priors,
                                 const double* dv, double* out, uint64_t* parent, uint64_t* seeds);

/* ======================================================================================== */
/* DEVICE-pointer entry points                                                               */
/* ======================================================================================== */

/* One SMC generation turn-over with everything resident in HBM
 * (AbcSmc.cpp:634-664 rank+truncate, :1041-1066 dv+weights, :490-518 proposals, :535 seeds). */
typedef struct {
    size_t N, M, P;            /* this set: particles, metrics, parameters                   */
    size_t K, Kp, Nnext;       /* pred-prior size, previous pred-prior size (0 = set 0), next */
    double train_frac;
    int32_t max_comp, rule, multivariate, reserved;
} abc_generation_cfg;

typedef struct {               /* all DEVICE pointers; optional ones may be NULL              */
    const double* X;           /* N x M                                                       */
    const double* Y;           /* N x P                                                       */
    const double* obs;         /* M                                                           */
    const abc_prior* priors;   /* P                                                           */
    const double* theta_prev;  /* Kp x P  (NULL for set 0)                                    */
    const double* w_prev;      /* Kp                                                          */
    const double* dv_prev;     /* P                                                           */
    uint64_t* idx;             /* K   selected particle rows, ascending distance              */
    double*   dist;            /* K   their distances (optional)                              */
    double*   theta;           /* K x P gathered posterior (optional)                         */
    double*   w;               /* K   weights                                                 */
    double*   dv;              /* P   doubled variance                                        */
    double*   L;               /* P x P Cholesky factor (multivariate; optional)              */
    double*   next;            /* Nnext x P proposals                                         */
    uint64_t* parent;          /* Nnext parent rows (optional)                                */
    uint64_t* seeds;           /* Nnext simulator seeds (optional)                            */
} abc_generation_io;

int abc_generation_dev(abc_ctx* ctx, const abc_generation_cfg* cfg, const abc_generation_io* io,
                       abc_rng* rng, int32_t* ncomp_host);

/* ---- stage-level device entry points (used by the sharded multi-GPU driver) -- */

/* Doubles needed for one sufficient-statistics record for (M,P):
 *   [ n_train, n_test, shift[C16], sum_train[C16], sum_test[C16], G_train[C16*C16], G_test[C16*C16] ]
 * with C16 = 16*ceil((M+P)/16).  Records from different row shards that used the same shift are
 * combined by plain addition of everything after shift[]; records about different shifts are re-centred first
 * (abc_generation_sharded_dev all-gathers the ranks' records, each about its own pilot shift, and merges them). */
size_t abc_stats_len(size_t M, size_t P);
/* pilot shift (mean of the first min(n,256) local rows) -> stats record's shift[] */
int abc_stats_shift_dev(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx,
                        size_t ldy, size_t M, size_t P, double* stats);
/* one pass over the local rows: column sums + Gram of the shifted [X|Y], train rows =
 * global rows < n_train_global.  row0 = global index of local row 0. */
int abc_stats_accumulate_dev(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx,
                             size_t ldy, size_t M, size_t P, uint64_t row0, uint64_t n_train_global,
                             double* stats);
/* model record length in doubles, and the fit: z-score moments, kernel-PLS deflation (type 2),
 * PRESS on the test statistics, component choice, observed scores. */
size_t abc_model_len(size_t M, size_t P, size_t A);
int abc_pls_model_dev(abc_ctx* ctx, const double* stats, const double* obs, size_t M, size_t P,
                      size_t A, int rule, double* model);
/* optional second step of the fit for rule ABC_RULE_WILCOXON: reduces the per-response PRESS optima using the
 * validation rows [row_test, n) of THIS device (single-GPU sets only) and rewrites ncomp in the model record */
int abc_pls_wilcoxon_dev(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy,
                         size_t M, size_t P, size_t A, size_t row_test, double* model);
int abc_model_ncomp(abc_ctx* ctx, const double* model, size_t M, size_t P, size_t A, int32_t* ncomp);
/* "simple" model: only means / sds / z-scored obs (AbcUtil.cpp:412-416) */
int abc_simple_model_dev(abc_ctx* ctx, const double* stats, const double* obs, size_t M, size_t P,
                         double* model);
/* per-row distance to the observed scores (AbcUtil.cpp:453-455, or :419 when simple != 0) */
int abc_project_distance_dev(abc_ctx* ctx, const double* X, size_t n, size_t ldx, size_t M, size_t P,
                             size_t A, const double* model, int simple, double* dist);
/* K smallest of dist[n] in ascending (dist, index) order: idx[K] (local row + idx_base), dist[K] */
int abc_select_smallest_dev(abc_ctx* ctx, const double* dist, size_t n, size_t K, uint64_t idx_base,
                            uint64_t* idx, double* dist_out);
/* Distributed exact selection: every shard histograms its keys for pass p = 0..5 (digits of the
 * IEEE bit pattern, high to low), the caller all-reduces hist (2048 x int32) over the shards, then every shard
 * picks the same digit; after pass 5 `state` holds the global K-th smallest key.  count -> {#below, #equal} per
 * shard (the caller decides how many ties each shard takes, lowest global rows first); compact -> that shard's
 * winners (dist, row + idx_base) in row order.  state: 8 x int64, hist: 2048 x int32 (zeroed by begin / pick). */
int abc_select_begin_dev(abc_ctx* ctx, uint64_t K, int64_t* state, int32_t* hist);
int abc_select_hist_dev(abc_ctx* ctx, const double* dist, size_t n, const int64_t* state, int pass, int32_t* hist);
int abc_select_pick_dev(abc_ctx* ctx, int64_t* state, int pass, int32_t* hist, uint64_t K);
int abc_select_count_dev(abc_ctx* ctx, const double* dist, size_t n, const int64_t* state, int64_t* counts);
int abc_select_compact_dev(abc_ctx* ctx, const double* dist, size_t n, const int64_t* state, uint64_t n_less,
                           uint64_t ties_take, uint64_t idx_base, uint64_t* idx_out, double* dist_out);
/* sort n (key, idx) pairs by (key, idx); used to merge per-shard winners */
int abc_sort_pairs_dev(abc_ctx* ctx, double* key, uint64_t* idx, size_t n);
/* merge n_runs runs of run_len pairs, each sorted by (key, idx), laid out back to back, into one sorted
 * sequence (ties: lower run first = a stable sort of the concatenation).  Out-of-place. */
int abc_merge_sorted_runs_dev(abc_ctx* ctx, const double* key, const uint64_t* idx, int n_runs, size_t run_len,
                              double* key_out, uint64_t* idx_out);
/* theta[i, :] = Y[idx[i] - idx_base, :] for idx in [idx_base, idx_base + n_local), else untouched */
int abc_gather_rows_dev(abc_ctx* ctx, const double* Y, size_t n_local, size_t ldy, size_t P,
                        const uint64_t* idx, size_t K, uint64_t idx_base, double* theta, size_t ldt);
int abc_doubled_variance_dev(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* dv);
/* un-normalised importance weights for rows [k0, k0+kn) of theta (sharded KDE); w_raw[kn] */
int abc_weights_raw_dev(abc_ctx* ctx, const abc_prior* priors, const double* theta, size_t K, size_t P,
                        size_t k0, size_t kn, const double* theta_prev, size_t Kp, const double* w_prev,
                        const double* dv_prev, double* w_raw);
/* w /= ||w||_2 (AbcUtil.cpp:583) */
int abc_normalize_l2_dev(abc_ctx* ctx, double* w, size_t K);
int abc_setup_mvn_sampler_dev(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* L);
/* draws [i0, i0+n) of the reference's resampling stream: parent[i] for those draws; rng is the
 * state at draw 0 and is NOT advanced. w is a device pointer (the alias table is built on the
 * host exactly as gsl_ran_discrete_preproc, then cached in the context). */
int abc_resample_dev(abc_ctx* ctx, const abc_rng* rng, const double* w, size_t K, uint64_t i0,
                     size_t n, uint64_t* parent);
/* proposals for draws [i0, i0+n): out is n x P (ld = n) */
int abc_perturb_dev(abc_ctx* ctx, const abc_rng* rng, const double* theta, size_t K, size_t P,
                    const abc_prior* priors, const uint64_t* parent, uint64_t i0, size_t n,
                    int multivariate, const double* L_or_dv, double* out, uint64_t* seeds,
                    uint64_t seed_stream_offset);

/* ======================================================================================== */
/* Multi-GPU: rows (particles) sharded over several GPUs of one node             */
/* ======================================================================================== */
/* The reference has no multi-device path (its MPI farm distributes simulator calls, AbcMPI.cpp:28-143, and is compiled
 * out); this is the particle sharding BASELINE.json's north_star asks for.  One context per GPU; a communicator is attached
 * to each context and the sharded entry points below run the same protocol on every rank:
 *   all-gather of the ranks' sufficient-statistics records (<= 0.35 MB each, merged on every rank), all-gather of the ranks'
 *   sorted local-top lists with their parameter rows (merged on every rank: the K smallest of the whole set), all-gather of the
 *   per-rank weight slices; resampling / perturbation need no exchange.  Small sets, K > N / 2 and massively tied distances take
 *   the radix protocol instead of the lists: six all-reduces of a 2048-bin histogram (exact global K-th distance) and
 *   all-gathers of the per-rank winner lists and rows.
 * Communicators: RCCL over xGMI (one process per GPU: abc_comm_unique_id + abc_comm_init_rank; or one process driving
 * several GPUs: abc_ctx_create_multi), or collectives supplied by the caller (abc_comm_init_callbacks: any transport; used
 * by the tests to run two ranks over gloo on one GPU). */
#define ABC_COMM_ID_BYTES 128
enum { ABC_DT_F64 = 0, ABC_DT_I32 = 1, ABC_DT_I64 = 2 };
/* every callback works on DEVICE buffers, in stream order of `hip_stream`, and returns 0 on success */
typedef struct {
    int (*all_reduce_sum)(void* user, void* buf, size_t count, int dtype, void* hip_stream);
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank, void* hip_stream);
    int (*broadcast)(void* user, void* buf, size_t bytes, int root, void* hip_stream);
    void* user;
} abc_comm_callbacks;
/* rank 0 creates the id and hands its 128 bytes to the other ranks (any channel), then every rank calls init_rank */
int abc_comm_unique_id(void* id128);
int abc_comm_init_rank(abc_ctx* ctx, int world, int rank, const void* id128);
int abc_comm_init_callbacks(abc_ctx* ctx, int world, int rank, const abc_comm_callbacks* cb);
int abc_comm_destroy(abc_ctx* ctx);
/* 0 = none, 1 = RCCL, 2 = callbacks; world and rank of the attached communicator (1, 0 without one) */
int abc_comm_info(const abc_ctx* ctx, int* kind, int* world, int* rank);
/* one process, ndev GPUs: creates ndev contexts (out[0..ndev)) joined by RCCL communicators (ncclCommInitAll); each is then
 * driven from its own host thread (abc_generation_multi below does that); destroy every context with abc_ctx_destroy */
int abc_ctx_create_multi(const int* devices, int ndev, abc_ctx** out);

/* One generation turn-over with the rows of the set sharded over the ranks of ctx's communicator (every rank calls this with
 * its shard; the call is collective).  Global row g of the set lives on the rank with row0 <= g < row0 + n_local; the next
 * set's particles [next0, next0 + nnext_local) are proposed by this rank.  io: X, Y are the LOCAL rows (leading dimension
 * n_local); idx / dist / theta / w / dv / L are replicated outputs (global row numbers in idx); next / parent / seeds
 * hold this rank's nnext_local proposals (leading dimension nnext_local).  rng: the same state on every rank; advanced by
 * the 2 Nnext_total draws of the whole generation.  Results equal abc_generation_dev on the unsharded set bit for bit
 * (indices, parents, seeds) and to rounding of the reduction order (statistics -> model -> distances within 1e-12) -- for the wide
 * sets that take the byte-limb statistics kernel (abc_ctx_set_gram_mode: 97..160 columns, or 81..96 with parameters from 2 000 000
 * rows, and 400 000 rows in every partition of the whole set) under ABC_GRAM_FP64 only: that kernel rounds every rank's values on
 * the rank's own grid; indices then agree up to near-ties, distances to ~1e-7. */
typedef struct {
    size_t n_local, row0, N_total;        /* this rank's rows of the current set                      */
    size_t M, P;
    size_t K, Kp;                         /* pred-prior size, previous pred-prior size (0 = set 0)    */
    size_t nnext_local, next0, Nnext_total;
    double train_frac;
    int32_t max_comp, rule, multivariate, reserved;
} abc_sharded_cfg;
int abc_generation_sharded_dev(abc_ctx* ctx, const abc_sharded_cfg* cfg, const abc_generation_io* io, abc_rng* rng,
                               int32_t* ncomp_host);

/* HOST-pointer generation over the ndev contexts of abc_ctx_create_multi: splits the N rows into contiguous shards, uploads
 * them, runs abc_generation_sharded_dev on one host thread per GPU and collects the outputs.  Same argument meaning as
 * abc_generation_cfg / abc_generation_io with HOST pointers (X: N x M, Y: N x P, next: Nnext x P, column-major). */
int abc_generation_multi(abc_ctx* const* ctxs, int ndev, const abc_generation_cfg* cfg, const abc_generation_io* host_io,
                         abc_rng* rng, int32_t* ncomp);

#ifdef __cplusplus
}
#endif
#endif /* ABCSMC_HIP_H */
