// Internal declarations shared by the HIP translation units of libabcsmc_hip.so.
// gfx950 (MI355X, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/abcsmc_hip.h"

#define ABC_WAVE 64

// Diagnostic switches (include/abcsmc_hip.h, "Diagnostic environment switches"): an environment variable steers the library
// ONLY when ABC_DIAG=1 is set beside it -- a production process that happens to inherit ABC_WS_POISON or ABC_ALIAS_FORCE_FAIL
// from somebody's shell is not affected.  Every getenv of the library goes through here.
inline const char* abc_diag_env(const char* name) {
    static const bool on = [] { const char* d = getenv("ABC_DIAG"); return d && d[0] == '1'; }();
    return on ? getenv(name) : nullptr;
}

// timed stages (abc_timing_names in api.hip must match)
enum { ST_GRAM = 0, ST_STATS_REDUCE, ST_PLS_MODEL, ST_PROJECT, ST_SELECT, ST_SORT, ST_GATHER_DV, ST_KDE,
       ST_WEIGHTS_MISC, ST_MVN, ST_ALIAS_HOST, ST_RESAMPLE, ST_PERTURB, ST_COMM, ABC_NSTAGE };

struct abc_ctx {
    int device;
    hipStream_t stream;
    hipStream_t own_stream;
    char err[512];
    // device workspace arena (bump allocator, reset per API call)
    char* ws;
    size_t ws_bytes;
    size_t ws_off;
    // pinned host scratch
    char* pin;
    size_t pin_bytes;
    // 128 pinned bytes.  [0, 64): the status words a generation reads back at its end (component count, Cholesky status, selection
    // flag): into pinned memory the three small copies are queued back to back behind ONE synchronisation; into pageable
    // memory each one is a blocking round trip (17 + 50 us of gaps at the end of a generation, rocprofv3 timeline).
    // [64, 128): the words of the Wilcoxon cascade (tests left after a level, the largest validation-row count of a rank: wilcoxon.hip)
    char* status_pin;
    // cached alias table (device) for the last weights vector handed to abc_resample_dev
    double* alias_F;
    uint32_t* alias_A;
    size_t alias_K;
    // alias table of K equal weights 1/K (set 0, AbcUtil.cpp:539-545): built once per K, kept (abc_uniform_alias)
    double* ualias_F;
    uint32_t* ualias_A;
    size_t ualias_cap, ualias_K;
    char* ualias_pin;      // its own pinned staging (the upload is asynchronous: the shared scratch may be reused before it ran)
    hipEvent_t ev_copy;    // marks the end of the weights' device-to-host copy (the host waits on it, not on the stream)
    // side stream of the fused drivers: the two taus2 streams of a generation (draws, seeds) depend on the rng state alone and
    // run there from the first launch on, beside the ranking chain (abc_rng_streams_early); ev_fork / ev_side order them
    hipStream_t side;
    hipStream_t wx_stream;             // the Wilcoxon reduction of a fused generation, beside the ranking that speculates on its outcome
    hipEvent_t ev_wx_fork, ev_wx_done, ev_wx_scores;
    hipEvent_t ev_fork, ev_side, ev_prev;
    hipEvent_t ev_theta, ev_moments;   // the posterior's moments on the side stream: start (rows gathered) and end
    bool side_forked;      // ev_fork of the current generation is recorded (abc_side_fork); cleared when the generation ends
    bool side_early_waited; // the main stream already waits for everything queued early on the side stream (ev_prev implies ev_side)
    // jump-ahead matrices for taus2 (device), built once
    uint32_t* jump_tab;
    // optional per-stage timing with HIP events recorded on ctx->stream (abc_timing_*)
    int timing;            // 0 off, 1 every stage, 2 only the two kernel brackets bench.py's roofline needs (k_gram, k_kde)
    int kde_mode;  // ABC_KDE_AUTO / ABC_KDE_FP64
    int gram_mode; // ABC_GRAM_AUTO / ABC_GRAM_FP64 (abc_ctx_set_gram_mode)
    int noise_mode;  // ABC_NOISE_DEVICE / ABC_NOISE_REFERENCE_STREAM
    int weight_kernel;  // ABC_WEIGHT_GAUSSIAN / ABC_WEIGHT_EPANECHNIKOV
    int alias_mode;     // ABC_ALIAS_DEVICE (default) / ABC_ALIAS_HOST
    int* alias_fail_dev;               // device flag of the last device build
    unsigned long long alias_dev_builds, alias_dev_fallbacks;   // device builds queued / found unusable (abc_alias_stats)
    unsigned long long* giveups_dev;   // proposals the perturbation gave up on (device counter, abc_perturb_giveups)
    unsigned long long giveups_host;   // ... and in the reference-stream host loop
    unsigned long long giveups_seen;   // device counter as of the last generation's end
    unsigned long long giveups_last_call;   // what the most recent generation added to it (abc_generation_giveups)
    unsigned long long giveups_dev_known;   // the device counter as the host last read it (re-read only when the pinned flag word says it moved)
    int timers_open;                   // StageTimers between their two events (the ring is only drained when none is)
    unsigned long long timing_dropped; // samples that found the ring full while a timer was open (abc_timing_read reports them)
    int* kde_which;  // device: which weight kernel produced the last sums (ABC_KDE_RAN_*), written by k_wfinish
    int* sel_fail_dev;       // device: the sampled-range bin selection gave up (select.hip); read by abc_select_check
    bool sel_bins_ran;       // the last launch_select_smallest took the bin path and has not been checked yet
    bool sel_force_radix;    // set by a caller that repeats its work after a failed bin selection
    unsigned long long wx_moved_counts, generation_repeats;      // abc_generation_repeats
    bool wx_force_inline;    // set by a generation that repeats itself after its speculation on the component count failed: the Wilcoxon reduction in stream order
    bool wx_gather_rows;     // diagnostic (ABC_DIAG=1 ABC_WX_GATHER=1, set at context creation): the sharded generation's Wilcoxon rule by
                             // gathering the validation rows on every rank (rounds 1-4) instead of the sharded cascade
    bool in_mvn;   // the covariance pass reuses k_gram: keep it out of the k_gram stage timer
    int nev;
    struct { hipEvent_t a, b; int stage; } ev[256];
    // communicator of the row-sharded generation (sharded.hip): none, RCCL or caller-supplied collectives
    int comm_kind;            // 0 = none (world 1), 1 = RCCL, 2 = callbacks
    int comm_world, comm_rank;
    void* comm_nccl;          // ncclComm_t
    abc_comm_callbacks comm_cb;
    char* xbuf;               // exchange buffers of the sharded generation (winner lists, packed posterior rows), grown on demand
    size_t xbuf_bytes;
    double stage_ms[ABC_NSTAGE];
    double stage_host_ms[ABC_NSTAGE];
    long long stage_cnt[ABC_NSTAGE];
};

int abc_timing_flush(abc_ctx* ctx);
// Flags of the events that order the context's two streams: GPU-to-GPU dependencies on one device, so a device-scope release is
// enough (the default, system scope, also makes the writes visible to the host: more cache maintenance per record)
inline unsigned abc_xstream_event_flags() {
    return hipEventDisableTiming | hipEventReleaseToDevice;
}
// RAII marker: records an event pair around a stage when timing is on
struct StageTimer {
    abc_ctx* ctx; int slot;
    StageTimer(abc_ctx* c, int stage) : ctx(c), slot(-1) {
        if (!c->timing || stage < 0) return;
        if (c->timing == 2 && stage != ST_GRAM && stage != ST_KDE) return;
        // ring full: drain it (a stream synchronisation) unless an enclosing timer is still open -- then the sample is lost,
        // counted, and abc_timing_read says so instead of handing out a short sum
        if (c->nev >= 256) {
            if (c->timers_open == 0) (void)abc_timing_flush(c);
            if (c->nev >= 256) { c->timing_dropped++; return; }
        }
        c->timers_open++;
        slot = c->nev++;
        if (!c->ev[slot].a) { (void)hipEventCreate(&c->ev[slot].a); (void)hipEventCreate(&c->ev[slot].b); }
        c->ev[slot].stage = stage;
        (void)hipEventRecord(c->ev[slot].a, c->stream);
    }
    ~StageTimer() { if (slot >= 0) { (void)hipEventRecord(ctx->ev[slot].b, ctx->stream); ctx->timers_open--; } }
};

#define ABC_FAIL(ctx, code, ...)                                   \
    do {                                                           \
        snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__);     \
        return (code);                                             \
    } while (0)

#define ABC_HIP(ctx, call)                                                            \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            ABC_FAIL(ctx, ABC_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,   \
                     hipGetErrorString(e_));                                          \
    } while (0)

#define ABC_TRY(expr)                  \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != ABC_OK) return rc_; \
    } while (0)

// ---- workspace ------------------------------------------------------------------------
int abc_ws_reserve(abc_ctx* ctx, size_t bytes);           // may reallocate; resets the arena
void* abc_ws_alloc(abc_ctx* ctx, size_t bytes);           // 256-B aligned bump; NULL if exhausted
int abc_pin_reserve(abc_ctx* ctx, size_t bytes);

static inline size_t abc_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- stats record layout (see abc_stats_len in the public header) ----------------------
struct StatsLayout {
    size_t C16;        // padded column count (multiple of 16)
    size_t off_n;      // 2 doubles: n_train, n_test
    size_t off_shift;  // C16
    size_t off_sum[2]; // C16 each
    size_t off_G[2];   // C16*C16 each, column-major, symmetric
    size_t len;
};
__host__ __device__ static inline StatsLayout stats_layout(size_t M, size_t P) {
    StatsLayout s;
    s.C16 = (M + P + 15) / 16 * 16;
    s.off_n = 0;
    s.off_shift = 2;
    s.off_sum[0] = s.off_shift + s.C16;
    s.off_sum[1] = s.off_sum[0] + s.C16;
    s.off_G[0] = s.off_sum[1] + s.C16;
    s.off_G[1] = s.off_G[0] + s.C16 * s.C16;
    s.len = s.off_G[1] + s.C16 * s.C16;
    return s;
}

// ---- model record layout ---------------------------------------------------------------
// [ ncomp, A, n_total, pad, mean[M+P], sd[M+P], zobs[M], obs_scores[A], R[M*A], Q[P*A], W[M*A],
//   Pl[M*A], H[A*A], press[A*P], per_response[P] ]
// H = R' (X'X of the z-scored validation rows) R: the second moments of the validation scores, a by-product of PRESS (round 5: the
// Wilcoxon reduction takes the scale of every test's paired differences from it, wilcoxon.hip)
struct ModelLayout {
    size_t off_hdr, off_mean, off_sd, off_zobs, off_oscore, off_R, off_Q, off_W, off_P, off_H, off_press,
        off_per, len;
};
__host__ __device__ static inline ModelLayout model_layout(size_t M, size_t P, size_t A) {
    ModelLayout m;
    m.off_hdr = 0;
    m.off_mean = 4;
    m.off_sd = m.off_mean + M + P;
    m.off_zobs = m.off_sd + M + P;
    m.off_oscore = m.off_zobs + M;
    m.off_R = m.off_oscore + A;
    m.off_Q = m.off_R + M * A;
    m.off_W = m.off_Q + P * A;
    m.off_P = m.off_W + M * A;
    m.off_H = m.off_P + M * A;
    m.off_press = m.off_H + A * A;
    m.off_per = m.off_press + A * P;
    m.len = m.off_per + P;
    return m;
}

// ---- stage launchers (each in its own .hip file) ---------------------------------------
int launch_stats_shift(abc_ctx*, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy,
                       size_t M, size_t P, double* stats);
// n_set (0: n): the rows of the WHOLE set these n are a shard of -- the choice between the i8 and the fp64 kernel of wide sets is
// made from it, so that all ranks of a sharded generation and the unsharded run agree (abc_ctx_set_gram_mode)
bool abc_gram_takes_i8(const abc_ctx*, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P,
                       uint64_t n_train_global, size_t n_set);     // the byte-limb statistics kernel for this set? (gram.hip)
int launch_stats_accumulate(abc_ctx*, const double* X, const double* Y, size_t n, size_t ldx,
                            size_t ldy, size_t M, size_t P, uint64_t row0, uint64_t n_train_global,
                            double* stats, size_t n_set = 0);
int launch_pls_model(abc_ctx*, const double* stats, const double* obs, size_t M, size_t P, size_t A,
                     int rule, double* model, hipEvent_t done = nullptr);
int launch_simple_model(abc_ctx*, const double* stats, const double* obs, size_t M, size_t P,
                        double* model);
int launch_project_distance(abc_ctx*, const double* X, size_t n, size_t ldx, size_t M, size_t P,
                            size_t A, const double* model, int simple, double* dist);
// scores of n rows (all A components) by the projection kernels: S[i + n k]; returns the rows taken (an even count, 0: not their shape)
size_t launch_project_scores(abc_ctx*, const double* X, size_t n, size_t ldx, size_t M, size_t P, size_t A, const double* model, double* S);
// the ranking's projection and the validation scores (rows from row_test on: S[i - row_test + sld k]) in one pass over X;
// 0: queued, 1: not a shape for it, nothing queued
int launch_project_distance_scores(abc_ctx*, const double* X, size_t n, size_t ldx, size_t M, size_t P, size_t A, const double* model,
                                   double* dist, double* S, size_t sld, size_t row_test, hipEvent_t done);
// the distances again from the scores of all rows that pass has left (S[i + sld k], row_test = 0): a repeat of the ranking on a
// lowered component count (model[0]) without a second pass over X
int launch_distance_from_scores(abc_ctx*, const double* S, size_t n, size_t sld, size_t M, size_t P, size_t A, const double* model, double* dist);
int launch_select_smallest(abc_ctx*, const double* dist, size_t n, size_t K, uint64_t idx_base,
                           uint64_t* idx, double* dist_out, bool defer_check = false);
int abc_select_check(abc_ctx* ctx, int* failed);
// the same in two halves for a caller with its own synchronisation: queue the flag's copy into *slot (pinned), then, after the
// synchronisation, abc_select_check_done reads it
int abc_select_check_queue(abc_ctx* ctx, int* slot);
int abc_select_check_done(abc_ctx* ctx, const int* slot);
// column slices of the weight kernel (weights.hip) and the bytes of partial sums they need; shared with the
// workspace sizing in api.hip
inline size_t abc_kde_slices(size_t kn, size_t Kp, int PP) {
    if (kn == 0 || Kp == 0) return 1;
    size_t rows_per_slice = 24576 / (size_t)PP;            // ~190 KB of scaled previous-set rows per slice
    if (rows_per_slice < 256) rows_per_slice = 256;
    size_t s = (Kp + rows_per_slice - 1) / rows_per_slice;
    const size_t rb = (kn + 255) / 256;
    if (s * rb < 4096) s = (4096 + rb - 1) / rb;           // small sets: still a few work-groups per resident slot
    size_t cap = ((size_t)64 << 20) / (8 * kn);            // partial sums: slices x kn doubles <= 64 MB ...
    if (cap < 8) cap = 8;                                  // ... but at least 8 slices
    if (s > cap) s = cap;
    if (s > Kp / 64) s = Kp / 64;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return s;
}
// osrc (optional): for every output position the flat input position q * len + i it came from
int launch_merge_runs(abc_ctx* ctx, const double* key, const uint64_t* idx, int W, size_t len, double* okey, uint64_t* oidx,
                      uint64_t* osrc = nullptr);
int launch_sort_pairs(abc_ctx*, double* key, uint64_t* idx, size_t n);
// distributed radix select stages (state: 8 x int64, hist: 2048 x int32, all-reduced by the caller between hist and pick)
int launch_select_begin(abc_ctx*, uint64_t K, long long* state, int* hist);
int launch_select_hist(abc_ctx*, const double* dist, size_t n, const long long* state, int pass, int* hist);
int launch_select_pick(abc_ctx*, long long* state, int pass, int* hist, uint64_t K);
int launch_select_count(abc_ctx*, const double* dist, size_t n, const long long* state, long long* counts);
int launch_select_compact(abc_ctx*, const double* dist, size_t n, const long long* state, uint64_t n_less,
                          uint64_t ties_take, uint64_t idx_base, uint64_t* idx_out, double* dist_out);
int abc_sort_u64_bytes(abc_ctx*, unsigned long long* key0, unsigned long long* val0, unsigned long long* key1,
                       unsigned long long* val1, size_t n, int byte_lo, int byte_hi);
// Wilcoxon reduction of the per-response component counts (rule ABC_RULE_WILCOXON); test rows = [row_test, n).
// sh (row-sharded sets, sharded.hip): X / Y are THIS rank's rows; the counts of every level of the bounds cascade are all-reduced
// over the context's communicator, the keys of the tests the bounds leave undecided are all-gathered (wilcoxon.hip).  Returns
// ABC_INTERNAL_RETRY when the set is not one the cascade takes (abc_wx_cascade_applies) or a bin of its exact step outgrew LDS:
// the caller then gathers the validation rows and calls again without sh.
struct abc_wx_shard {
    size_t nv_total;          // validation rows of the whole set
    const double* nv_ranks;   // device: nv_ranks[q * nv_stride] = validation rows of rank q (the n_test word of its statistics record)
    size_t nv_stride;
};
bool abc_wx_cascade_applies(size_t nv_total, size_t P, size_t A);
// dec / changed_host (the fused generation's SPECULATIVE run, api.hip): with dec != NULL the cascade leaves the model record as the
// fit wrote it and puts its decision into dec (P per-response counts, then the largest); *changed_host = 0: the largest count is
// the fit's (everything ranked with it stands), 1: it differs (launch_wilcoxon_commit, rank again), 2: the reduction took a path
// that rewrote the model record itself (small sets, the sorted-path repeat): rank again.
int launch_wilcoxon(abc_ctx*, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P,
                    size_t A, size_t row_test, double* model, const abc_wx_shard* sh = nullptr, double* dec = nullptr,
                    int* changed_host = nullptr, int stop_at_max = 0);
// stop_at_max (the generations: their caller only uses the LARGEST per-response count, AbcUtil.cpp:449): the cascade ends as soon as
// that is certain; the per-response counts it leaves are then upper ends for the responses it did not finish
int launch_wilcoxon_commit(abc_ctx*, double* model, size_t M, size_t P, size_t A, const double* dec, int with_hdr);
// the cascade in two halves, for a caller with work to queue between them (the fused generation: api.hip)
struct abc_wx_run;
// scores: the caller has a pass over X of its own to queue (the ranking's projection) and lets it write the validation scores
// too: fn(arg, &S, &sld) queues that pass, makes the context's (= the cascade's) stream wait for it, says where the validation rows'
// scores are (S[i + sld k], row i of the validation rows: the caller's own buffer -- round 6: the generation keeps the scores of ALL
// rows, so that a moved count repeats the distances from them instead of from X) and returns 0 -- or returns 1 without queueing
// anything, and the cascade scores the rows itself into a buffer of its own
struct abc_wx_scores_hook { int (*fn)(void* arg, double** S, size_t* sld); void* arg; };
int launch_wilcoxon_begin(abc_ctx*, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P, size_t A,
                          size_t row_test, double* model, double* dec, int stop_at_max, abc_wx_run** out,
                          const abc_wx_scores_hook* scores = nullptr, int hold_level0 = 0, const abc_wx_shard* sh = nullptr);
// sh: the rows are a shard of a row-sharded set (the counts of every level are all-reduced over the context's communicator)
// hold_level0: begin stops in front of level 0's sweep; the caller queues it (on the cascade's stream) once its own launches are out
int launch_wilcoxon_level0(abc_ctx*, abc_wx_run* run);
int launch_wilcoxon_finish(abc_ctx*, abc_wx_run* run, int* changed_host);
void launch_wilcoxon_abandon(abc_ctx*, abc_wx_run* run, hipStream_t its_stream);
// collectives on the context's stream and the exchange buffer (sharded.hip)
int abc_comm_all_reduce(abc_ctx* ctx, void* buf, size_t count, int dtype);
int abc_comm_all_gather(abc_ctx* ctx, const void* send, void* recv, size_t bytes);
int abc_xbuf_reserve(abc_ctx* ctx, size_t bytes);
// sel_fail / sel_fail_pin (fused drivers): the bin selection's give-up flag is stored into the pinned status block by this
// kernel, and the proposals' give-up counter is snapshotted into its second slot.  done: an event bound to the kernel's OWN
// completion signal (hipExtLaunchKernelGGL's stop event) -- a hipEventRecord behind it is one more packet in the queue, and the
// next kernel waits for that packet: ~7 us of the main stream's critical path per record (rocprofv3 timeline)
int abc_giveups_ensure(abc_ctx* ctx);
int launch_gather_rows(abc_ctx*, const double* Y, size_t n_local, size_t ldy, size_t P,
                       const uint64_t* idx, size_t K, uint64_t idx_base, double* theta, size_t ldt,
                       const int* sel_fail = nullptr, int* sel_fail_pin = nullptr, hipEvent_t done = nullptr);
#define ABC_INTERNAL_RETRY (-2147483647)      // launch_resample: the caller's abort flag was set when the host looked (not an error code of the ABI)
int launch_doubled_variance(abc_ctx*, const double* theta, size_t K, size_t P, double* dv);
// K x P posterior moments computed once (Gram kernel) and shared by the doubled variance and the MVN factor
int launch_theta_stats(abc_ctx*, const double* theta, size_t K, size_t P, double** stats_out);
// everything that follows from the posterior's statistics record in ONE launch (mvn.hip: k_post_tail; P <= 64; fields optional)
struct abc_theta_fused {
    double* dv; double* L; int* spd; double* rows /* K x PP row-major, zero padded */; double* Lpad /* PP x PP */;
    // (fused drivers) the generation's status words straight into the pinned block, by the work-group that has the Cholesky status
    // anyway: the model header (component count; NULL: none) and the status -- no copy kernel behind the proposals then
    const double* model_hdr; double* hdr_pin; int* spd_pin;
};
int launch_post_tail(abc_ctx*, const double* theta, size_t K, size_t P, const double* stats, const abc_theta_fused* f);
int launch_dv_from_stats(abc_ctx*, const double* stats, size_t P, double* dv);
int launch_mvn_from_stats(abc_ctx*, const double* stats, size_t P, double* L, int* status_dev);
// the previous set's share of the weight stage (weights.hip: launch_weights_prev), prepared ahead of launch_weights_raw
struct abc_wprev {
    void* wc /* WConst */; double* b; double* hb; unsigned short* bt; unsigned* far_list;
    unsigned* tmin;        // KS_TOPN (tiles in the order of the rows' norm tops): the smallest top of every tile, f32 bits; else NULL
    size_t Kp, P, kn_max; int split, ready;
};
int launch_weights_prev(abc_ctx*, size_t P, size_t kn_max, const double* theta_prev, size_t Kp, const double* w_prev,
                        const double* dv_prev, abc_wprev* out, hipStream_t st);
int launch_weights_raw(abc_ctx*, const abc_prior* priors, const double* theta, size_t K, size_t P,
                       size_t k0, size_t kn, const double* theta_prev, size_t Kp,
                       const double* w_prev, const double* dv_prev, double* w_raw, const abc_wprev* prev = nullptr,
                       const double** sumsq_out = nullptr);
// sumsq_out (optional): when the call covers the whole set (k0 = 0, kn = K) the 64-row partials of the raw weights' sum of
// squares are computed by the same launch; *sumsq_out then points at them (device) for launch_normalize_l2's sumsq_parts, else NULL
// launch_weights_prev on the side stream (resample.hip); the caller makes its stream wait for ctx->ev_prev before the rest
int abc_weights_prev_early(abc_ctx* ctx, size_t P, size_t kn_max, const double* theta_prev, size_t Kp, const double* w_prev,
                           const double* dv_prev, abc_wprev* out);
int launch_fill(abc_ctx*, double* w, size_t K, double v, hipStream_t st = nullptr);     // st == NULL: the context's stream
int launch_normalize_l2(abc_ctx*, double* w, size_t K, double* host_mirror = nullptr, const double* sumsq_parts = nullptr);
// pinned scratch of the alias build over K weights: w | F | A | (pad) | E | the two index stacks (K + 1 entries each)
static inline size_t abc_alias_pin_bytes(size_t K) { return K * (sizeof(double) * 3 + sizeof(uint32_t) * 3) + 2 * sizeof(uint32_t) + 16; }
int launch_mvn_setup(abc_ctx*, const double* theta, size_t K, size_t P, double* L, int* status_host,
                     int* status_dev);
// while_host_builds (optional): called after the weights' copy to the host has been queued and before the host waits for it:
// GPU work launched there runs while the host builds the alias table
// uniform_weights: w is K copies of 1.0 / K (launch_fill): the table comes from abc_uniform_alias, no host round trip here
// raw_ready (optional): the n taus2 outputs of the draws, already queued on the side stream (abc_rng_streams_early)
int launch_resample(abc_ctx*, const abc_rng* rng, const double* w, size_t K, uint64_t i0, size_t n,
                    uint64_t* parent, int (*while_host_builds)(void*) = nullptr, void* hook_arg = nullptr,
                    bool uniform_weights = false, const uint32_t* raw_ready = nullptr, bool weights_on_host = false,
                    const volatile int* abort_flag = nullptr, bool parents_ready = false, int* alias_check_deferred = nullptr);
// alias_check_deferred (optional): with the table built on the device (ctx->alias_mode) nothing waits for the build's verdict
// here; *alias_check_deferred = 1 then tells the caller to read the pinned flag (ctx->status_pin + 44) at its next
// synchronisation and, if it is set, to call launch_resample again with ctx->alias_mode = ABC_ALIAS_HOST.  NULL: this
// function synchronises and falls back by itself.
// parents_ready (uniform weights only): abc_rng_streams_early has drawn the parents on the side stream already
// abort_flag (pinned, optional): read right after the host has waited for the weights; non-zero -> nothing more is queued and
// ABC_INTERNAL_RETRY is returned (the weights belong to a placeholder selection: the caller repeats its generation)
// weights_on_host: the kernel that normalised w already stored them at the start of the context's pinned scratch
// (launch_normalize_l2's host_mirror, reserved with the size used here): nothing to copy
// Queues on the context's side stream everything of the proposals that depends on the rng state alone: the taus2 outputs
// i0 .. i0 + n - 1 of the resampling draws (-> *raw) and, if seeds != NULL, the simulator seeds = outputs seed_stream_offset +
// i0 + i.  The main stream waits for them in launch_resample (raw_ready).  Call before the first kernel of the generation.
int abc_rng_streams_early(abc_ctx* ctx, const abc_rng* rng, uint64_t i0, size_t n, uint64_t* seeds, uint64_t seed_stream_offset,
                          uint32_t** raw, uint64_t* parent_uniform = nullptr, size_t K_uniform = 0);
// parent_uniform (set 0): the weights will be K_uniform copies of 1 / K_uniform whatever the ranking says, their alias table is
// on the device already (abc_uniform_alias, called BEFORE the fork), so the parents are drawn here too, beside the ranking
int abc_rng_seeds_early(abc_ctx* ctx, const abc_rng* rng, uint64_t i0, size_t n, uint64_t* seeds, uint64_t seed_stream_offset);
int abc_side_fork(abc_ctx* ctx);      // records where on the main stream the side stream's work of this generation may start
// Alias table of K equal weights: gsl_ran_discrete_preproc on K copies of 1.0 / K (bit-identical to the table of the filled
// weight vector), built on the host on first use for this K and kept in HBM.  The fused drivers call it right after queueing
// the ranking kernels, so the host builds the table while the GPU ranks.
int abc_uniform_alias(abc_ctx* ctx, size_t K);
// what launch_perturb_prepare has already done: row-major posterior copy, seeds, the padded Cholesky factor
struct abc_perturb_prep { double* rows; int seeds_done; double* Lpad; };   // seeds_done may be preset by the caller
// multivariate / L_or_dv (optional): with them the factor is padded for the perturbation kernel as well
int launch_perturb_prepare(abc_ctx*, const abc_rng* rng, const double* theta, size_t K, size_t P, uint64_t i0, size_t n,
                           uint64_t* seeds, uint64_t seed_stream_offset, abc_perturb_prep* prep, int multivariate = 0,
                           const double* L_or_dv = nullptr);
// padded width of the row-major posterior copy the perturbation kernels read (and of their padded factor)
static inline int abc_perturb_pp(size_t P) {
    int PP = 2;
    while (PP < (int)P) PP *= 2;
    return (P > 64) ? (int)((P + 63) / 64 * 64) : PP;
}
int launch_perturb(abc_ctx*, const abc_rng* rng, const double* theta, size_t K, size_t P,
                   const abc_prior* priors, const uint64_t* parent, uint64_t i0, size_t n,
                   int multivariate, const double* L_or_dv, double* out, uint64_t* seeds,
                   uint64_t seed_stream_offset, const abc_perturb_prep* prep = nullptr);

// reference-stream proposals (host loop, refstream_host.cpp): rng = state after the n resampling draws, advanced past all it consumes
int launch_perturb_reference(abc_ctx*, abc_rng* rng_after_draws, const double* theta, size_t K, size_t P, const abc_prior* priors,
                             const uint64_t* parent, size_t n, int multivariate, const double* L_or_dv, double* out,
                             uint64_t* seeds);

// [GSL] gsl_ran_discrete_preproc on the host (alias_host.cpp, a host-only translation unit built with the host compiler):
// scratch E: K doubles, smalls / bigs: K + 1 uint32 each
// knuth = false: without the final KNUTH_CONVENTION pass (k_alias_draw applies it when it reads the table)
void abc_alias_preproc(size_t K, const double* w, double* F, uint32_t* A, double* E, uint32_t* smalls, uint32_t* bigs, bool knuth = true);

// Walker alias table built on the device (alias_dev.hip): F without the KNUTH_CONVENTION map, as abc_alias_preproc(knuth = false);
// *fail_dev / *fail_pin (optional, pinned) = 1: the speculation did not verify (or the weights are outside its grid) -- the table
// must not be used, the caller builds it on the host
// verdict_src (optional): instead of a launch of its own that publishes the verdict, *verdict_src gets the device word that
// holds it; the caller's next kernel copies it to fail_dev / fail_pin (k_alias_draw does)
int launch_alias_build_dev(abc_ctx*, const double* w, size_t K, double* F, uint32_t* A, int* fail_dev, int* fail_pin,
                           const int** verdict_src = nullptr);
size_t abc_alias_dev_need(size_t K);       // arena bytes of one build
constexpr size_t ABC_ALIAS_DEV_MAX_K = (size_t)2 << 20;
// below this the ten launches of the device build (~90 us, latency-bound) lose to the host's round trip (K = 1e4: 26 us of host
// work + two PCIe hops); measured break-even between 1e4 and 1e5
constexpr size_t ABC_ALIAS_DEV_MIN_K = 20000;
constexpr size_t ABC_ALIAS_DEV_SMALL_K = 300000;   // up to here the build runs with two elements per thread, beyond with four (alias_dev.hip)

// arena bound shared by api.hip and sharded.hip
size_t abc_ws_need(size_t N, size_t M, size_t P, size_t A, size_t K, size_t Kp, size_t Nnext);
size_t abc_wx_need(size_t n_validation, size_t P, size_t A);     // arena of the Wilcoxon reduction (launch_wilcoxon)
int abc_timing_flush(abc_ctx* ctx);
void abc_comm_release(abc_ctx* ctx);     // sharded.hip: called by abc_ctx_destroy

// taus2 helpers shared by host code
void taus2_set(abc_rng* r, unsigned long seed);
uint32_t taus2_get(abc_rng* r);
void taus2_jump(abc_rng* r, uint64_t n);
