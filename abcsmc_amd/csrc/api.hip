// C ABI of libabcsmc_hip.so (see include/abcsmc_hip.h for the contract and reference citations).
#include <math.h>
#include <stdlib.h>

#include <new>
#include <vector>

#include "abc_internal.h"

// ---- workspace -------------------------------------------------------------------------------
int abc_ws_reserve(abc_ctx* ctx, size_t bytes) {
    bytes = abc_align(bytes, 1 << 20);
    if (bytes > ctx->ws_bytes) {
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->ws) { ABC_HIP(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_bytes = 0; }
        ABC_HIP(ctx, hipMalloc((void**)&ctx->ws, bytes));
        ctx->ws_bytes = bytes;
    }
    ctx->ws_off = 0;
    // ABC_WS_POISON=<byte> (debugging): every entry point starts from a workspace filled with that byte (ff: NaNs and huge
    // integers), so that a kernel which reads a word nobody wrote in THIS call shows up in the tests instead of inheriting
    // whatever the previous call left there
    static const char* poison = abc_diag_env("ABC_WS_POISON");
    if (poison && ctx->ws) ABC_HIP(ctx, hipMemsetAsync(ctx->ws, (int)strtol(poison, nullptr, 16), ctx->ws_bytes, ctx->stream));
    return ABC_OK;
}
void* abc_ws_alloc(abc_ctx* ctx, size_t bytes) {
    const size_t off = abc_align(ctx->ws_off, 256);
    if (off + bytes > ctx->ws_bytes) return nullptr;
    ctx->ws_off = off + bytes;
    return ctx->ws + off;
}
int abc_pin_reserve(abc_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pin_bytes) return ABC_OK;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pin) { ABC_HIP(ctx, hipHostFree(ctx->pin)); ctx->pin = nullptr; ctx->pin_bytes = 0; }
    bytes = abc_align(bytes, 1 << 16);
    ABC_HIP(ctx, hipHostMalloc((void**)&ctx->pin, bytes, hipHostMallocDefault));
    ctx->pin_bytes = bytes;
    return ABC_OK;
}

// generous upper bound of the arena needed by any single API call on these sizes
size_t abc_ws_need(size_t N, size_t M, size_t P, size_t A, size_t K, size_t Kp, size_t Nnext) {
    const size_t Call = (M + P + 15) / 16, C = Call > 6 ? 6 : Call;      // wider sets go through 96-column group pairs
    const size_t psz = (C * (C + 1) / 2) * 256 + 16 * C;
    size_t b = 0;
    b += 2 * 384 * psz * 8 * 2;                                   // gram partials (PLS stats + covariance)
    b += 4 * (stats_layout(M, P).len + model_layout(M, P, A ? A : 1).len) * 8;
    b += (2 * M * P + 2 * M * M + P + A * M + A * A + 2 * P * A + 64 + M * 40) * 8;
    b += (M * P + 3 * P * P + 2 * P + 8 * M + 2 * M * A + A + 64 + 2048) * 8 + 3 * 96 * 96 * 8 * 2;   // PLS work arrays beyond the LDS; group-pair records
    b += N * 8;                                                   // distances
    b += 4 * K * 8 + (N / 2048 + 2) * 8 + 2048 * 4 + 256 * (K / 2048 + 2) * 4 + 4096;   // select + sort
    b += 2 * (K + 1024) * 8 + 34 * 16384 * 4 + 4096;              // ... or the bin selection's pair buffer, counts and cursors
    b += 3 * N * 8 + 256 * (N / 2048 + 2) * 4;                    // full-sort case K == N
    const size_t PPw = P <= 64 ? 64 : (P + 63) / 64 * 64;         // padded row width of the row-major copies
    b += K * P * 8 + K * PPw * 8;                                 // theta, and its row-major copy for the perturb gather
    b += (K + Kp) * PPw * 8 + 1024 * 8 + 64 * PPw * 8 + 32768;    // weights: scaled copies of both sets, centre partials, constants
    b += (K + Kp + 512) * (9 * 32 + 8 + 12) + 8192;               // ... and their f16 limb tiles (<= 9 operands of 32 B a row), 1/2|a|^2 parts
    b += (K + 512) * 17 + 16384;                                   // far-row flags, list and fix-up sums
    b += (Kp + 64) * (8 + 4 + 4 + 1) + (Kp / 2048 + 2) * 256 * 4 + 4096;   // tiles in the order of the norm tops: keys, ranks, tops, tile info, bin counts
    if (K && Kp) b += ((size_t)64 << 20) + 64 * K + ((size_t)16 << 20);   // ... and the per-slice partial sums (abc_kde_slices)
    b += K * 8 + P * P * 8 + P * 8;
    b += Nnext * (8 + 8 + 4 + 4);                                 // parent, seeds, raw streams
    b += 2 * PPw * PPw * 8 + 4096;                                // padded Cholesky factor of the proposals, factorisation scratch
    if (K) b += abc_alias_dev_need(K);                            // the resampling table's device build (alias_dev.hip)
    b += 64 * 256;                                                // alignment slack
    return b + (4u << 20);
}

// ---- context ---------------------------------------------------------------------------------
extern "C" int abc_version(void) { return 100; }

extern "C" int abc_ctx_create(int device, abc_ctx** out) {
    if (!out) return ABC_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return ABC_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return ABC_ERR_HIP;
    abc_ctx* ctx = new (std::nothrow) abc_ctx();
    if (!ctx) return ABC_ERR_NOMEM;
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ABC_ERR_HIP; }
    if (hipHostMalloc((void**)&ctx->status_pin, 128, hipHostMallocDefault) != hipSuccess) { (void)hipStreamDestroy(ctx->own_stream); delete ctx; return ABC_ERR_HIP; }
    ctx->stream = ctx->own_stream;
    ctx->wx_gather_rows = abc_diag_env("ABC_WX_GATHER") != nullptr;
    ctx->gram_mode = abc_diag_env("ABC_GRAM_FP64") ? ABC_GRAM_FP64 : (abc_diag_env("ABC_GRAM_I8") ? ABC_GRAM_I8 : ABC_GRAM_AUTO);      // (the diagnostic switch of round 4: the initial mode)
    *out = ctx;
    return ABC_OK;
}

extern "C" void abc_ctx_destroy(abc_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->status_pin) (void)hipHostFree(ctx->status_pin);
    if (ctx->alias_F) (void)hipFree(ctx->alias_F);       // alias_A lives in the same allocation
    if (ctx->ualias_F) (void)hipFree(ctx->ualias_F);
    if (ctx->ualias_pin) (void)hipHostFree(ctx->ualias_pin);
    if (ctx->jump_tab) (void)hipFree(ctx->jump_tab);
    if (ctx->kde_which) (void)hipFree(ctx->kde_which);
    if (ctx->giveups_dev) (void)hipFree(ctx->giveups_dev);
    if (ctx->alias_fail_dev) (void)hipFree(ctx->alias_fail_dev);
    abc_comm_release(ctx);
    if (ctx->xbuf) (void)hipFree(ctx->xbuf);
    if (ctx->ev_copy) (void)hipEventDestroy(ctx->ev_copy);
    if (ctx->ev_theta) { (void)hipEventDestroy(ctx->ev_theta); (void)hipEventDestroy(ctx->ev_moments); }
    if (ctx->wx_stream) { (void)hipStreamSynchronize(ctx->wx_stream); (void)hipStreamDestroy(ctx->wx_stream); (void)hipEventDestroy(ctx->ev_wx_fork); (void)hipEventDestroy(ctx->ev_wx_done); (void)hipEventDestroy(ctx->ev_wx_scores); }
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_side); if (ctx->ev_prev) (void)hipEventDestroy(ctx->ev_prev); }
    for (int i = 0; i < 256; i++) if (ctx->ev[i].a) { (void)hipEventDestroy(ctx->ev[i].a); (void)hipEventDestroy(ctx->ev[i].b); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" const char* abc_last_error(const abc_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int abc_ctx_set_stream(abc_ctx* ctx, void* hip_stream) {
    if (!ctx) return ABC_ERR_INVALID;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    return ABC_OK;
}

extern "C" int abc_ctx_use_own_stream(abc_ctx* ctx) {
    if (!ctx) return ABC_ERR_INVALID;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return ABC_OK;
}

extern "C" int abc_ctx_set_kde_mode(abc_ctx* ctx, int mode) {
    if (!ctx) return ABC_ERR_INVALID;
    if (mode != ABC_KDE_AUTO && mode != ABC_KDE_FP64) ABC_FAIL(ctx, ABC_ERR_INVALID, "abc_ctx_set_kde_mode: unknown mode %d", mode);
    ctx->kde_mode = mode;
    return ABC_OK;
}

extern "C" int abc_ctx_set_gram_mode(abc_ctx* ctx, int mode) {
    if (!ctx) return ABC_ERR_INVALID;
    if (mode != ABC_GRAM_AUTO && mode != ABC_GRAM_FP64 && mode != ABC_GRAM_I8) ABC_FAIL(ctx, ABC_ERR_INVALID, "abc_ctx_set_gram_mode: unknown mode %d", mode);
    ctx->gram_mode = mode;
    return ABC_OK;
}

extern "C" int abc_ctx_set_weight_kernel(abc_ctx* ctx, int kernel) {
    if (!ctx) return ABC_ERR_INVALID;
    if (kernel != ABC_WEIGHT_GAUSSIAN && kernel != ABC_WEIGHT_EPANECHNIKOV)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "abc_ctx_set_weight_kernel: unknown kernel %d", kernel);
    ctx->weight_kernel = kernel;
    return ABC_OK;
}

extern "C" int abc_ctx_set_noise_mode(abc_ctx* ctx, int mode) {
    if (!ctx) return ABC_ERR_INVALID;
    if (mode != ABC_NOISE_DEVICE && mode != ABC_NOISE_REFERENCE_STREAM)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "abc_ctx_set_noise_mode: unknown mode %d", mode);
    ctx->noise_mode = mode;
    return ABC_OK;
}

extern "C" int abc_ctx_set_alias_mode(abc_ctx* ctx, int mode) {
    if (!ctx) return ABC_ERR_INVALID;
    if (mode != ABC_ALIAS_DEVICE && mode != ABC_ALIAS_HOST) ABC_FAIL(ctx, ABC_ERR_INVALID, "abc_ctx_set_alias_mode: unknown mode %d", mode);
    ctx->alias_mode = mode;
    return ABC_OK;
}

extern "C" int abc_alias_stats(abc_ctx* ctx, uint64_t* device_builds, uint64_t* host_fallbacks, int reset) {
    if (!ctx) return ABC_ERR_INVALID;
    if (device_builds) *device_builds = ctx->alias_dev_builds;
    if (host_fallbacks) *host_fallbacks = ctx->alias_dev_fallbacks;
    if (reset) { ctx->alias_dev_builds = 0; ctx->alias_dev_fallbacks = 0; }
    return ABC_OK;
}

extern "C" int abc_perturb_giveups(abc_ctx* ctx, uint64_t* count, int reset) {
    if (!ctx || !count) return ABC_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_HIP, "hipSetDevice failed");
    unsigned long long dev = 0;
    if (ctx->giveups_dev) {
        ABC_HIP(ctx, hipMemcpyAsync(&dev, ctx->giveups_dev, sizeof(dev), hipMemcpyDeviceToHost, ctx->stream));
        if (reset) ABC_HIP(ctx, hipMemsetAsync(ctx->giveups_dev, 0, sizeof(dev), ctx->stream));
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *count = (uint64_t)dev + (uint64_t)ctx->giveups_host;
    ctx->giveups_dev_known = reset ? 0ull : dev;
    ctx->giveups_seen = reset ? 0 : (unsigned long long)*count;
    if (reset) ctx->giveups_host = 0;
    return ABC_OK;
}

extern "C" int abc_generation_giveups(const abc_ctx* ctx, uint64_t* count) {
    if (!ctx || !count) return ABC_ERR_INVALID;
    *count = (uint64_t)ctx->giveups_last_call;
    return ABC_OK;
}

extern "C" int abc_generation_repeats(abc_ctx* ctx, uint64_t* ranking_repeats, uint64_t* generation_repeats, int reset) {
    if (!ctx) return ABC_ERR_INVALID;
    if (ranking_repeats) *ranking_repeats = (uint64_t)ctx->wx_moved_counts;
    if (generation_repeats) *generation_repeats = (uint64_t)ctx->generation_repeats;
    if (reset) { ctx->wx_moved_counts = 0; ctx->generation_repeats = 0; }
    return ABC_OK;
}

extern "C" int abc_kde_last_kernel(abc_ctx* ctx, int* which) {
    if (!ctx || !which) return ABC_ERR_INVALID;
    *which = ABC_KDE_RAN_NONE;
    if (!ctx->kde_which) return ABC_OK;
    ABC_HIP(ctx, hipMemcpyAsync(which, ctx->kde_which, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

extern "C" int abc_ctx_synchronize(abc_ctx* ctx) {
    if (!ctx) return ABC_ERR_INVALID;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

static const char* const kStageNames[ABC_NSTAGE] = {
    "k_gram", "stats_reduce", "pls_model", "project_distance", "select", "sort_winners", "gather_dv", "k_kde",
    "weights_misc", "mvn_setup", "alias_host", "resample", "perturb", "collectives"};

extern "C" int abc_timing_enable(abc_ctx* ctx, int on) {
    if (!ctx) return ABC_ERR_INVALID;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->timing = (on == 2) ? 2 : (on != 0);
    ctx->nev = 0;
    ctx->timers_open = 0;
    ctx->timing_dropped = 0;
    return ABC_OK;
}

int abc_timing_flush(abc_ctx* ctx) {
    if (!ctx->nev) return ABC_OK;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < ctx->nev; i++) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev[i].a, ctx->ev[i].b) == hipSuccess) {
            ctx->stage_ms[ctx->ev[i].stage] += ms;
            ctx->stage_cnt[ctx->ev[i].stage] += 1;
        }
    }
    ctx->nev = 0;
    return ABC_OK;
}

extern "C" int abc_timing_read(abc_ctx* ctx, const char** names, double* ms, double* host_ms, long long* count,
                               int max_stages, int reset) {
    if (!ctx) return ABC_ERR_INVALID;
    ABC_TRY(abc_timing_flush(ctx));
    if (ctx->timing_dropped) {
        const unsigned long long d = ctx->timing_dropped;
        ctx->timing_dropped = 0;
        ABC_FAIL(ctx, ABC_ERR_INVALID, "abc_timing_read: %llu stage samples were lost (event ring full inside an open stage); the sums are short", d);
    }
    for (int i = 0; i < ABC_NSTAGE && i < max_stages; i++) {
        if (names) names[i] = kStageNames[i];
        if (ms) ms[i] = ctx->stage_ms[i];
        if (host_ms) host_ms[i] = ctx->stage_host_ms[i];
        if (count) count[i] = ctx->stage_cnt[i];
    }
    if (reset) for (int i = 0; i < ABC_NSTAGE; i++) { ctx->stage_ms[i] = 0; ctx->stage_host_ms[i] = 0; ctx->stage_cnt[i] = 0; }
    return ABC_NSTAGE;
}

// Event-bracket overhead: what an event pair around ONE kernel reports beyond that kernel's execution (packet
// processing before the dispatch, the end-of-kernel release before the closing marker).  b1 = bracket around one
// empty kernel, b2 = around two; b2 - b1 is the marginal cost of an empty kernel, so overhead = b1 - (b2 - b1).
__global__ void k_noop() {}
extern "C" int abc_timing_overhead(abc_ctx* ctx, int reps, double* overhead_ms) {
    if (!ctx || !overhead_ms || reps < 1) return ABC_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_HIP, "hipSetDevice failed");
    hipEvent_t a, b;
    ABC_HIP(ctx, hipEventCreate(&a));
    ABC_HIP(ctx, hipEventCreate(&b));
    double acc[2] = {0, 0};
    for (int k = 0; k < 2; k++)
        for (int r = 0; r < reps + 2; r++) {
            hipLaunchKernelGGL(k_noop, dim3(256), dim3(256), 0, ctx->stream);      // the stream is busy, as in a step
            ABC_HIP(ctx, hipEventRecord(a, ctx->stream));
            for (int j = 0; j <= k; j++) hipLaunchKernelGGL(k_noop, dim3(256), dim3(256), 0, ctx->stream);
            ABC_HIP(ctx, hipEventRecord(b, ctx->stream));
            ABC_HIP(ctx, hipEventSynchronize(b));
            float ms = 0;
            ABC_HIP(ctx, hipEventElapsedTime(&ms, a, b));
            if (r >= 2) acc[k] += ms;                                              // two warm-up rounds
        }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    const double b1 = acc[0] / reps, b2 = acc[1] / reps;
    double ov = b1 - (b2 - b1);
    if (ov < 0) ov = 0;
    *overhead_ms = ov;
    return ABC_OK;
}

extern "C" void abc_rng_set(abc_rng* r, unsigned long seed) { taus2_set(r, seed); }
extern "C" uint32_t abc_rng_get(abc_rng* r) { return taus2_get(r); }
extern "C" void abc_rng_jump(abc_rng* r, uint64_t n) { taus2_jump(r, n); }

// Every public entry starts here.  With timing on, the event ring (256 pairs) is drained as soon as it is half full: no
// stage timer is open at an entry point, so every recorded pair is complete, and a long run of stage-level calls (the
// sharded driver opens ~30 timers per step and never reaches generation_core's flush) loses no sample.
#define CHECK_CTX(ctx)                         \
    do {                                       \
        if (!(ctx)) return ABC_ERR_INVALID;    \
        (ctx)->err[0] = 0;                     \
        if (hipSetDevice((ctx)->device) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_HIP, "hipSetDevice failed"); \
        if ((ctx)->timing && (ctx)->nev > 128) ABC_TRY(abc_timing_flush(ctx)); \
    } while (0)

static size_t default_A(size_t M, size_t P, int max_comp) {
    return (max_comp > 0) ? (size_t)max_comp : (M < P ? M : P);
}

// ---- stage-level device entry points -----------------------------------------------------------
extern "C" size_t abc_stats_len(size_t M, size_t P) { return stats_layout(M, P).len; }
extern "C" size_t abc_model_len(size_t M, size_t P, size_t A) { return model_layout(M, P, A).len; }

extern "C" int abc_stats_shift_dev(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy,
                                   size_t M, size_t P, double* stats) {
    CHECK_CTX(ctx);
    return launch_stats_shift(ctx, X, Y, n, ldx, ldy, M, P, stats);
}

extern "C" int abc_stats_accumulate_dev(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx,
                                        size_t ldy, size_t M, size_t P, uint64_t row0, uint64_t n_train_global,
                                        double* stats) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, M, P, 1, 0, 0, 0)));
    return launch_stats_accumulate(ctx, X, Y, n, ldx, ldy, M, P, row0, n_train_global, stats);
}

extern "C" int abc_pls_model_dev(abc_ctx* ctx, const double* stats, const double* obs, size_t M, size_t P, size_t A,
                                 int rule, double* model) {
    CHECK_CTX(ctx);
    if (rule != ABC_RULE_MIN_PRESS)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "abc_pls_model_dev fits from statistics only: call abc_pls_wilcoxon_dev "
                 "afterwards for rule %d (needs the validation rows)", rule);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, M, P, A, 0, 0, 0)));
    return launch_pls_model(ctx, stats, obs, M, P, A, rule, model);
}

extern "C" int abc_pls_wilcoxon_dev(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy,
                                    size_t M, size_t P, size_t A, size_t row_test, double* model) {
    CHECK_CTX(ctx);
    const size_t nt = row_test < n ? n - row_test : 0;
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, M, P, A, 0, 0, 0) + abc_wx_need(nt, P, A)));
    return launch_wilcoxon(ctx, X, Y, n, ldx, ldy, M, P, A, row_test, model);
}

extern "C" int abc_simple_model_dev(abc_ctx* ctx, const double* stats, const double* obs, size_t M, size_t P,
                                    double* model) {
    CHECK_CTX(ctx);
    return launch_simple_model(ctx, stats, obs, M, P, model);
}

extern "C" int abc_model_ncomp(abc_ctx* ctx, const double* model, size_t M, size_t P, size_t A, int32_t* ncomp) {
    CHECK_CTX(ctx);
    double hdr[4];
    ABC_HIP(ctx, hipMemcpyAsync(hdr, model, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ncomp) *ncomp = (int32_t)hdr[0];
    return ABC_OK;
}

extern "C" int abc_project_distance_dev(abc_ctx* ctx, const double* X, size_t n, size_t ldx, size_t M, size_t P,
                                        size_t A, const double* model, int simple, double* dist) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, M, P, A, 0, 0, 0)));
    return launch_project_distance(ctx, X, n, ldx, M, P, A, model, simple, dist);
}

extern "C" int abc_select_smallest_dev(abc_ctx* ctx, const double* dist, size_t n, size_t K, uint64_t idx_base,
                                       uint64_t* idx, double* dist_out) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(n, 1, 1, 1, K, 0, 0)));
    return launch_select_smallest(ctx, dist, n, K, idx_base, idx, dist_out);
}

extern "C" int abc_select_begin_dev(abc_ctx* ctx, uint64_t K, int64_t* state, int32_t* hist) {
    CHECK_CTX(ctx);
    return launch_select_begin(ctx, K, (long long*)state, (int*)hist);
}
extern "C" int abc_select_hist_dev(abc_ctx* ctx, const double* dist, size_t n, const int64_t* state, int pass,
                                   int32_t* hist) {
    CHECK_CTX(ctx);
    return launch_select_hist(ctx, dist, n, (const long long*)state, pass, (int*)hist);
}
extern "C" int abc_select_pick_dev(abc_ctx* ctx, int64_t* state, int pass, int32_t* hist, uint64_t K) {
    CHECK_CTX(ctx);
    return launch_select_pick(ctx, (long long*)state, pass, (int*)hist, K);
}
extern "C" int abc_select_count_dev(abc_ctx* ctx, const double* dist, size_t n, const int64_t* state, int64_t* counts) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(n, 1, 1, 1, 0, 0, 0)));
    return launch_select_count(ctx, dist, n, (const long long*)state, (long long*)counts);
}
extern "C" int abc_select_compact_dev(abc_ctx* ctx, const double* dist, size_t n, const int64_t* state, uint64_t n_less,
                                      uint64_t ties_take, uint64_t idx_base, uint64_t* idx_out, double* dist_out) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(n, 1, 1, 1, n_less + ties_take, 0, 0)));
    return launch_select_compact(ctx, dist, n, (const long long*)state, n_less, ties_take, idx_base, idx_out, dist_out);
}

extern "C" int abc_sort_pairs_dev(abc_ctx* ctx, double* key, uint64_t* idx, size_t n) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(n, 1, 1, 1, n, 0, 0)));
    return launch_sort_pairs(ctx, key, idx, n);
}

extern "C" int abc_merge_sorted_runs_dev(abc_ctx* ctx, const double* key, const uint64_t* idx, int n_runs, size_t run_len,
                                        double* key_out, uint64_t* idx_out) {
    CHECK_CTX(ctx);
    if (!key || !idx || !key_out || !idx_out || key == key_out || idx == idx_out)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "merge: null or aliased buffers");
    return launch_merge_runs(ctx, key, idx, n_runs, run_len, key_out, idx_out);
}

extern "C" int abc_gather_rows_dev(abc_ctx* ctx, const double* Y, size_t n_local, size_t ldy, size_t P,
                                   const uint64_t* idx, size_t K, uint64_t idx_base, double* theta, size_t ldt) {
    CHECK_CTX(ctx);
    return launch_gather_rows(ctx, Y, n_local, ldy, P, idx, K, idx_base, theta, ldt);
}

extern "C" int abc_doubled_variance_dev(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* dv) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, P, 0, 1, 0, 0, 0)));     // the moments go through the Gram kernel's partial records
    return launch_doubled_variance(ctx, theta, K, P, dv);
}

extern "C" int abc_weights_raw_dev(abc_ctx* ctx, const abc_prior* priors, const double* theta, size_t K, size_t P,
                                   size_t k0, size_t kn, const double* theta_prev, size_t Kp, const double* w_prev,
                                   const double* dv_prev, double* w_raw) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, 1, P, 1, kn, Kp, 0)));
    return launch_weights_raw(ctx, priors, theta, K, P, k0, kn, theta_prev, Kp, w_prev, dv_prev, w_raw);
}

extern "C" int abc_normalize_l2_dev(abc_ctx* ctx, double* w, size_t K) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, (K / 64 + 2) * sizeof(double) + (1 << 20)));
    return launch_normalize_l2(ctx, w, K);
}

extern "C" int abc_setup_mvn_sampler_dev(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* L) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, P, 0, 1, 0, 0, 0)));
    int st = 0;
    ABC_TRY(launch_mvn_setup(ctx, theta, K, P, L, &st, nullptr));
    if (st) ABC_FAIL(ctx, ABC_ERR_NOT_SPD, "covariance of the selected particles is not positive definite");
    return ABC_OK;
}

extern "C" int abc_resample_dev(abc_ctx* ctx, const abc_rng* rng, const double* w, size_t K, uint64_t i0, size_t n,
                                uint64_t* parent) {
    CHECK_CTX(ctx);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, 1, 1, 1, 0, 0, n) + abc_alias_dev_need(K)));
    return launch_resample(ctx, rng, w, K, i0, n, parent);
}

extern "C" int abc_perturb_dev(abc_ctx* ctx, const abc_rng* rng, const double* theta, size_t K, size_t P,
                               const abc_prior* priors, const uint64_t* parent, uint64_t i0, size_t n,
                               int multivariate, const double* L_or_dv, double* out, uint64_t* seeds,
                               uint64_t seed_stream_offset) {
    CHECK_CTX(ctx);
    if (ctx->noise_mode == ABC_NOISE_REFERENCE_STREAM)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "abc_perturb_dev works on row slices; the reference stream is sequential over the whole "
                 "set (use abc_sample_*_predictive_priors or abc_generation_dev)");
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, 1, P, 1, K, 0, n)));
    return launch_perturb(ctx, rng, theta, K, P, priors, parent, i0, n, multivariate, L_or_dv, out, seeds,
                          seed_stream_offset);
}

// ---- fused generation (device-resident) ----------------------------------------------------------
// does NOT reserve/reset the arena: the caller has done so (lets host wrappers keep their staging
// buffers in the same arena)
// the status words of a generation (model header with the component count, Cholesky status, selection flag) stored by the GPU
// straight into the context's pinned, device-visible block
__global__ void k_status_words(const double* __restrict__ model_hdr, const int* __restrict__ spd, const int* __restrict__ fail,
                               const unsigned long long* __restrict__ giveups, double* __restrict__ hdr_out, int* __restrict__ spd_out,
                               int* __restrict__ fail_out, unsigned long long* __restrict__ giveups_out) {
    const int t = threadIdx.x;
    if (model_hdr && t < 4) hdr_out[t] = model_hdr[t];
    if (spd && t == 4) *spd_out = *spd;
    if (fail && t == 5) *fail_out = *fail;
    if (t == 6) *giveups_out = giveups ? giveups[0] : 0ull;
}

static int generation_core(abc_ctx* ctx, const abc_generation_cfg* cfg, const abc_generation_io* io, abc_rng* rng,
                           int32_t* ncomp_host, int simple, const double** model_out = nullptr) {
    const size_t N = cfg->N, M = cfg->M, P = cfg->P, K = cfg->K, Kp = cfg->Kp, Nn = cfg->Nnext;
    if (!N || !M || K > N) ABC_FAIL(ctx, ABC_ERR_INVALID, "generation: bad sizes N=%zu M=%zu K=%zu", N, M, K);
    const size_t ws_entry = ctx->ws_off;              // a failed bin selection (select.hip) repeats the call from here
    ctx->side_early_waited = false;
    ctx->giveups_last_call = 0;
    if (Nn) ABC_TRY(abc_giveups_ensure(ctx));         // (the gather snapshots the counter: it has to exist before the first proposals)
    abc_rng rng_entry;
    if (rng) rng_entry = *rng;
    if (!simple && !(0.0 < cfg->train_frac && cfg->train_frac <= 1.0))      // AbcUtil.cpp:428
        ABC_FAIL(ctx, ABC_ERR_INVALID, "training fraction %g outside (0,1]", cfg->train_frac);
    if (!simple && cfg->rule != ABC_RULE_MIN_PRESS && cfg->rule != ABC_RULE_WILCOXON)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "unknown component rule %d", cfg->rule);
    const size_t A = simple ? 0 : default_A(M, P, cfg->max_comp);
    const StatsLayout SL = stats_layout(M, P);
    const ModelLayout ML = model_layout(M, P, A);
    double* stats = (double*)abc_ws_alloc(ctx, SL.len * 8);
    double* model = (double*)abc_ws_alloc(ctx, ML.len * 8);
    double* dist = (double*)abc_ws_alloc(ctx, N * 8);
    int* spd_dev = (int*)abc_ws_alloc(ctx, sizeof(int));
    if (!stats || !model || !dist || !spd_dev) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
    if (model_out) *model_out = model;
    const uint64_t ntrain = simple ? N : (uint64_t)llround((double)N * cfg->train_frac);   // AbcUtil.cpp:438
    const double* Yp = io->Y ? io->Y : io->X;
    const size_t Pstat = io->Y ? P : 0;
    // first set: the weights will be 1/K whatever the ranking says, so their alias table is built (once per K, then kept) before
    // anything is queued, and the parents are drawn on the side stream beside the ranking
    const bool uniform_w = io->w && K && (Kp == 0 || !io->theta_prev);
    const bool early = io->w && Nn && K && rng && ctx->noise_mode != ABC_NOISE_REFERENCE_STREAM;
    uint64_t* parent_early = nullptr;
    if (uniform_w && Nn) {
        ABC_TRY(abc_uniform_alias(ctx, K));
        if (early) {
            parent_early = io->parent ? io->parent : (uint64_t*)abc_ws_alloc(ctx, Nn * 8);
            if (!parent_early) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
        }
    }
    // The side stream's work (taus2 streams, the previous set's share of the weight stage) depends on nothing this call queues:
    // it is forked at the call's START -- the record is the first packet of an idle queue, processed while the host prepares the
    // first launch -- and runs beside the Gram kernel; forked behind that kernel (rounds 2 and 3) the record sat between the
    // reduce and the fit and cost the critical path ~6 us (rocprofv3 timeline).  Its launches still follow the fit's.
    // ... except beside the BYTE-LIMB statistics kernel of wide sets (round 6): that one keeps a 512-thread work-group with 150-160 KB
    // of LDS on every CU, and every side-stream kernel resident beside it takes CUs away from it for as long as it runs -- the
    // previous set's prologue alone is 0.74 ms there at configs[3] (0.2 ms on an idle chip): 2.34 ms for the statistics pass against
    // 1.6 stand-alone.  Forked behind it the side stream's work runs beside the model fit and the projection instead: everything
    // but the pair sums 6.80 -> 6.57 ms at configs[3], 1.408 -> 1.387 at configs[4].
    static const int fork_late_env = abc_diag_env("ABC_FORK_LATE") ? 1 : (abc_diag_env("ABC_FORK_EARLY") ? -1 : 0);     // A/B switches
    const int fork_late = fork_late_env > 0 || (fork_late_env == 0 && !simple && abc_gram_takes_i8(ctx, io->X, Yp, N, N, N, M, Pstat, ntrain, N));
    ctx->side_forked = false;
    if (!fork_late) ABC_TRY(abc_side_fork(ctx));
    ABC_TRY(launch_stats_shift(ctx, io->X, Yp, N, N, N, M, Pstat, stats));
    ABC_TRY(launch_stats_accumulate(ctx, io->X, Yp, N, N, N, M, Pstat, 0, ntrain, stats));
    if (fork_late) ABC_TRY(abc_side_fork(ctx));
    // (the cascade's stream is forked behind the fit: where the generation will speculate -- the condition is restated below --
    // the fork's event rides on the fit kernel's completion signal)
    static const int wx_inline = abc_diag_env("ABC_WX_INLINE") ? 1 : 0;                  // A/B switch for measurements
    const bool wx_spec = !simple && cfg->rule == ABC_RULE_WILCOXON && io->w && K && !wx_inline && !ctx->wx_force_inline &&
                         abc_wx_cascade_applies(N > (size_t)ntrain ? N - (size_t)ntrain : 0, P, A);
    if (wx_spec && !ctx->wx_stream) {
        // (stream priorities were tried for it, highest and lowest: no effect on any config beyond the runs' spread)
        ABC_HIP(ctx, hipStreamCreateWithFlags(&ctx->wx_stream, hipStreamNonBlocking));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_wx_fork, abc_xstream_event_flags()));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_wx_done, abc_xstream_event_flags()));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_wx_scores, abc_xstream_event_flags()));
    }
    if (simple) ABC_TRY(launch_simple_model(ctx, stats, io->obs, M, Pstat, model));
    else ABC_TRY(launch_pls_model(ctx, stats, io->obs, M, P, A, cfg->rule, model, wx_spec ? ctx->ev_wx_fork : nullptr));
    // the taus2 streams of the proposals (draws, seeds) need the rng state only: on the side stream, forked behind the Gram
    // kernel (which wants the whole memory system) and running beside the reduce / model fit that leave the chip empty
    uint32_t* raw_early = nullptr;
    bool filled_early = false;
    if (early && uniform_w) {         // the uniform weights themselves (AbcUtil.cpp:543-544): nothing on the main stream reads them
        ABC_TRY(launch_fill(ctx, io->w, K, 1.0 / (double)K, ctx->side));
        filled_early = true;
    }
    // ... and so does everything the weight stage needs of the PREVIOUS set (scales, centre, scaled copy, limb tiles -- in the
    // order of the rows' norm tops where that saves the pair sums an MFMA).  Order on the side stream: the draws' taus2 outputs (pure
    // arithmetic: they run beside the Gram kernel without touching its memory system), the previous set's prologue (the pair sums
    // need it ~0.3 ms into the generation), THEN the seeds (an output nobody reads before the call returns): the moments' event,
    // which the main stream waits for in front of the resampling table, is recorded behind all of them on the same stream
    const bool weighted = io->w && K && Kp && io->theta_prev;
    static const int moments_main = abc_diag_env("ABC_MOMENTS_MAIN") ? 1 : 0;             // A/B switches for measurements
    const bool moments_side_possible = Nn && P <= 64 && K >= 2 && !uniform_w && !moments_main && ctx->side;
    static const int seeds_first = abc_diag_env("ABC_SEEDS_FIRST") ? 1 : 0;              // A/B switch for measurements
    const bool seeds_late = early && weighted && !seeds_first && moments_side_possible;
    if (early) ABC_TRY(abc_rng_streams_early(ctx, rng, 0, Nn, seeds_late ? nullptr : io->seeds, Nn, &raw_early, parent_early, K));
    abc_wprev wprev;
    memset(&wprev, 0, sizeof(wprev));
    // (tried, round 6: the ranking's launches in front of the prologue's in this thread's order -- under rocprofv3 the selection's launches
    // arrive late behind the prologue's ten, 38 us of idle main stream; unprofiled the host is fast enough and the step is 11 us LONGER
    // that way, 0.609 against 0.597 ms outside the pair sums at configs[2]: the prologue then runs beside the selection instead of
    // beside the model fit's empty chip)
    if (weighted)
        ABC_TRY(abc_weights_prev_early(ctx, P, K, io->theta_prev, Kp, io->w_prev, io->dv_prev, &wprev));
    if (seeds_late) ABC_TRY(abc_rng_seeds_early(ctx, rng, 0, Nn, io->seeds, Nn));
    ctx->side_forked = false;
    // The Wilcoxon reduction of the component count goes BEHIND the side stream's launches in host order (round 5): the host looks
    // at the cascade's level counts between its launches, and whatever it has not queued by then waits for those looks -- queued
    // in front (rounds 1-4), the previous set's prologue and the taus2 streams started only after the reduction and the host's
    // ~90 us of enqueueing them showed as a bubble in front of the projection (rocprofv3 timeline, profiles/history/r05_timeline_*)
    // SPECULATION (round 5, second half): the ranking does not wait for the reduction.  The reduction can only LOWER the component
    // count of a response, and the count the distances use is the largest over the responses -- unchanged unless every response
    // that holds the maximum is reduced.  So projection, selection and gather are queued at once on the counts the fit wrote, the
    // cascade runs beside them on a stream of its own (it reads the model record, its decision goes to a buffer of its own), and
    // when the host has its result -- the ranking's kernels are long done by then -- either nothing has changed (the usual case:
    // the per-response counts are committed, the generation goes on) or the decision is committed and the three stages run again.
    // Whole generations on sets the cascade takes only (ranking-only calls and small sets: in stream order, as before).
    const bool wx_rule = !simple && cfg->rule == ABC_RULE_WILCOXON;
    const size_t nvalid = N > (size_t)ntrain ? N - (size_t)ntrain : 0;
    // ROUND 6: the host LOOKS at the reduction in front of the weight stage again (round 5's first form), because the look has become
    // cheap: the cascade takes the LARGEST COUNT FIRST (wilcoxon.hip, k_wx_plan) -- level 0 over the tests of two to four responses
    // that hold it instead of all P (A - 1) --, so its verdict is there before the selection and the gather beside it have ended, and
    // a moved count costs the cascade's second half and the three ranking stages once more instead of the whole generation
    // (round 5's default queued everything up to the proposals on the fit's count first and repeated the generation: ABC_WX_DEFER,
    // kept for A/B runs).
    static const int wx_defer_env = abc_diag_env("ABC_WX_DEFER") ? 1 : 0;
    double* wx_dec = nullptr;
    abc_wx_run* wx_run = nullptr;
    struct WxGuard {          // an error return between the cascade's halves: its kernels still write into this call's arena
        abc_ctx* c; abc_wx_run** r;
        ~WxGuard() { if (*r) { launch_wilcoxon_abandon(c, *r, c->wx_stream); *r = nullptr; } }
    } wx_guard = {ctx, &wx_run};
    bool projected = false;                              // the ranking's projection queued by the cascade's first half (below)
    const double* scores_all = nullptr;                  // ... which has then left the scores of all rows (N x A, leading dimension N)
    if (wx_spec) {
        wx_dec = (double*)abc_ws_alloc(ctx, (P + 1) * 8);
        if (!wx_dec) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
        ABC_HIP(ctx, hipStreamWaitEvent(ctx->wx_stream, ctx->ev_wx_fork, 0));           // (the fit's completion: launch_pls_model above)
        // its first half -- plan, scores, the level-0 sweep and bounds -- is queued BEFORE the ranking (the host needs ~60 us to queue
        // the ranking's eight launches: the cascade started that much late behind them, rocprofv3 timeline)
        // ONE pass over X for both: the ranking's projection (main stream) also writes the validation rows' scores, all A
        // components, and the cascade's sweeps wait for it -- as two launches the validation half of X was read twice and the
        // second pass (51 us at configs[2]) ran beside the selection's kernels
        hipStream_t main_stream = ctx->stream;
        // (round 6: the pass leaves the scores of EVERY row, not only the validation half -- N x A doubles of this call's arena: should
        // the reduction lower the largest count, the distances are taken again from them, N x count x 8 bytes instead of X once more)
        static const int scores_valid_only = abc_diag_env("ABC_SCORES_VALID_ONLY") ? 1 : 0;      // A/B switch: round 5's half
        struct ScoresArg { abc_ctx* ctx; hipStream_t main; const double* X; size_t N, M, P, A, ntrain; const double* model; double* dist; double* S_all; }
            sarg = {ctx, main_stream, io->X, N, M, P, A, (size_t)ntrain, model, dist, nullptr};
        if (!scores_valid_only && !(N & 1)) sarg.S_all = (double*)abc_ws_alloc(ctx, N * A * 8);
        abc_wx_scores_hook hook = {
            [](void* a, double** S, size_t* sld) -> int {
                ScoresArg* q = (ScoresArg*)a;
                abc_ctx* c = q->ctx;
                hipStream_t wx = c->stream;
                double* S_half = nullptr;
                if (!q->S_all) {
                    S_half = (double*)abc_ws_alloc(c, (q->N - q->ntrain) * q->A * 8);
                    if (!S_half) { snprintf(c->err, sizeof(c->err), "generation: workspace exhausted"); return ABC_ERR_NOMEM; }
                }
                c->stream = q->main;
                int rc = q->S_all ? launch_project_distance_scores(c, q->X, q->N, q->N, q->M, q->P, q->A, q->model, q->dist, q->S_all, q->N, 0, c->ev_wx_scores)
                                  : launch_project_distance_scores(c, q->X, q->N, q->N, q->M, q->P, q->A, q->model, q->dist, S_half, q->N - q->ntrain, q->ntrain,
                                                                   c->ev_wx_scores);
                c->stream = wx;
                if (rc == 0 && hipStreamWaitEvent(wx, c->ev_wx_scores, 0) != hipSuccess) rc = ABC_ERR_HIP;
                if (rc == 0) {
                    q->dist = nullptr;          // (taken)
                    *S = q->S_all ? q->S_all + q->ntrain : S_half;
                    *sld = q->S_all ? q->N : q->N - q->ntrain;
                } else if (rc == 1)
                    q->S_all = nullptr;         // (not a shape for the fused pass: no scores of all rows either)
                return rc;
            },
            &sarg};
        ctx->stream = ctx->wx_stream;
        // (level 0's sweep is held back until the selection and the gather are queued: the sweep cannot start before the projection
        // has ended anyway, and queued in front of them its three launches kept the selection from the main stream for ~65 us)
        const int rcb = launch_wilcoxon_begin(ctx, io->X, io->Y, N, N, N, M, P, A, (size_t)ntrain, model, wx_dec, /*stop_at_max=*/1, &wx_run, &hook,
                                              /*hold_level0=*/1);
        ctx->stream = main_stream;
        ABC_TRY(rcb);
        projected = sarg.dist == nullptr;
        scores_all = projected ? sarg.S_all : nullptr;
    } else if (wx_rule)
        ABC_TRY(launch_wilcoxon(ctx, io->X, io->Y, N, N, N, M, P, A, (size_t)ntrain, model));
    if (!projected) ABC_TRY(launch_project_distance(ctx, io->X, N, N, M, simple ? Pstat : P, A, model, simple, dist));
    if (K == 0) return ABC_OK;
    ABC_TRY(launch_select_smallest(ctx, dist, N, K, 0, io->idx, io->dist, /*defer_check=*/io->w != nullptr));
    // (tried, round 6: level 0 of the cascade held BEHIND the selection where the gather is long enough to hide it -- at configs[3] the
    // sweep beside the selection's histogram pass stretches that from 45 to 164 us, and the 0.6 ms gather behind it has room for the
    // sweep: everything but the pair sums 6.60 -> 6.69 ms, the gather loses more than the histogram gains.  Not kept.)
    if (!simple && ncomp_host && !io->w) {
        double hdr[4];
        ABC_HIP(ctx, hipMemcpyAsync(hdr, model, sizeof(hdr), hipMemcpyDeviceToHost, ctx->stream));
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *ncomp_host = (int32_t)hdr[0];
    }
    if (!io->w) return ABC_OK;   // ranking only

    double* theta = io->theta ? io->theta : (double*)abc_ws_alloc(ctx, K * P * 8);
    if (!theta) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
    // (the gather is the first kernel behind the selection: it also stores the selection's give-up flag into the pinned block)
    int* pfail_early = (int*)(ctx->status_pin + 40);
    *pfail_early = 0;
    volatile unsigned* pgaveup = (volatile unsigned*)(ctx->status_pin + 56);       // raised by a proposal kernel that gives up (note_giveup)
    *pgaveup = 0u;
    // the status words the host reads at the generation's end (model header, Cholesky status): zeroed here, written by the
    // posterior's k_post_tail where that kernel runs -- the copy kernel behind the proposals is then not launched
    double* const hdr_pin = (double*)ctx->status_pin;
    int* const spd_pin = (int*)(ctx->status_pin + 32);
    hdr_pin[0] = 0.0; *spd_pin = 0;
    bool status_early = false;
    static const int status_late = abc_diag_env("ABC_STATUS_KERNEL") ? 1 : 0;           // A/B switch for measurements
    bool bins_deferred = ctx->sel_bins_ran && ctx->sel_fail_dev && !ctx->sel_force_radix;
    // The main stream needs what the side stream queued early (the previous set's tiles, the taus2 outputs) only behind the
    // gather, and those kernels ended long ago: the wait goes IN FRONT of the gather, where the event is certain to have fired
    // (a wait that still has to be resolved between the gather and the new set's tiles cost ~12 us of the critical path there)
    static const int wait_late = abc_diag_env("ABC_WAIT_LATE") ? 1 : 0;
    ctx->side_early_waited = false;
    if (wprev.ready && !wait_late) {
        ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prev, 0));
        ctx->side_early_waited = true;                 // (ev_prev was recorded behind ev_side on the same stream)
    }
    // (weighted generations hand the gathered rows to the side stream, for the posterior's moments: the event that orders the two
    // is the gather's own completion signal, not a record behind it)
    static const int side_moments_on = abc_diag_env("ABC_MOMENTS_MAIN") ? 0 : 1;        // A/B switches for measurements
    static const int ev_marker = abc_diag_env("ABC_EV_MARKER") ? 1 : 0;
    const bool defer_moments = Nn && P <= 64 && K >= 2;
    const bool moments_side_planned = defer_moments && !uniform_w && side_moments_on && ctx->side && ctx->noise_mode != ABC_NOISE_REFERENCE_STREAM;
    if (moments_side_planned && !ctx->ev_theta) {
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_theta, abc_xstream_event_flags()));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_moments, abc_xstream_event_flags()));
    }
    const bool theta_ev_bound = moments_side_planned && !ev_marker;
    ABC_TRY(launch_gather_rows(ctx, io->Y, N, N, P, io->idx, K, 0, theta, K, bins_deferred ? ctx->sel_fail_dev : nullptr, pfail_early,
                               theta_ev_bound ? ctx->ev_theta : nullptr));
    if (wx_run) {                                        // level 0 of the cascade, behind the ranking's launches in host order
        hipStream_t main_stream = ctx->stream;
        // ... and, where its sweep is short enough to fit beside the resampling table and the proposals at the generation's end, behind
        // the gather ON THE DEVICE too: a level sweep fills every CU with a 1024-thread work-group, the selection's kernels beside it
        // waited for CUs (configs[3]: the 80 MB histogram pass took 1.4 ms beside a 1.4 ms sweep), and once the pair sums run the
        // sweep gets no CU until they end -- so a held-back sweep runs behind the pair sums.  Estimates: ~1e9 keys per ms at up to
        // 8 components (0.8e9 to 16, 0.5e9 to 32); the proposals ~16 P bytes per row at 4 TB/s behind ~0.15 ms of table build.
        // configs[3]: 359.9 -> 358.8 ms (everything but the pair sums 8.4 -> 7.1 ms), configs[2]: the same within the runs' spread,
        // configs[4] (a 0.5 ms sweep against 0.2 ms of proposals: not held back; forced, 3.71 -> 3.90 ms)
        static const int l0_force = abc_diag_env("ABC_WX_L0_AFTER_GATHER") ? atoi(abc_diag_env("ABC_WX_L0_AFTER_GATHER")) : -1;   // A/B switch: 0 / 1
        const double sweep_ms = (double)nvalid * (double)(P * (A - 1)) / (A <= 8 ? 1.0e9 : (A <= 16 ? 0.8e9 : 0.5e9));
        const double tail_ms = 0.15 + (double)Nn * (double)P * 16.0 / 4.0e9;
        // ... and only where there are pair sums to speak of (~4.8e9 pairs per ms): the cascade is a chain of short dependent steps
        // (sweep, totals, bounds, the host's look), and started behind the gather it outlasts a generation whose weight stage is
        // over in 0.04 ms (configs[1], K = K' = 1e4: 0.336 -> 0.361 ms held)
        const double pairs_ms = (double)K * (double)Kp / 4.8e9;
        const bool l0_after = (l0_force >= 0 ? l0_force != 0 : (sweep_ms <= tail_ms && pairs_ms >= 0.25)) && wx_defer_env;
        if (l0_after && wx_spec && weighted) {
            if (!theta_ev_bound) ABC_HIP(ctx, hipEventRecord(ctx->ev_wx_scores, ctx->stream));
            ABC_HIP(ctx, hipStreamWaitEvent(ctx->wx_stream, theta_ev_bound ? ctx->ev_theta : ctx->ev_wx_scores, 0));
        }
        ctx->stream = ctx->wx_stream;
        const int rc0 = launch_wilcoxon_level0(ctx, wx_run);
        ctx->stream = main_stream;
        ABC_TRY(rc0);
    }
    // Where the host looks at the cascade: HERE, in front of the weight stage (round 6; round 5's first form).  The verdict of the
    // cascade's first half -- the tests of a few responses that hold the largest count -- is there by the time the gather ends; a count
    // that stands costs nothing, a count that moved costs the cascade's second half and the ranking once more (distances from the
    // kept scores, selection, gather).  ABC_WX_DEFER: round 5's default -- the weight stage and the proposals queued first on the
    // fit's count, the looks beside the pair sums, a moved count throwing the generation away (kept for A/B runs).
    const bool wx_defer = wx_spec && wx_defer_env;
    bool wx_tail_pending = false;
    if (wx_spec && !wx_defer) {
        // the reduction itself, on its own stream, while the ranking queued above runs (the host's looks at the cascade's level
        // counts happen here, beside GPU work that does not depend on them)
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->wx_stream;
        int changed = 2;
        int rc = launch_wilcoxon_finish(ctx, wx_run, &changed);
        wx_run = nullptr;
        if (rc == ABC_OK && hipEventRecord(ctx->ev_wx_done, ctx->wx_stream) != hipSuccess) rc = ABC_ERR_HIP;
        ctx->stream = main_stream;
        if (rc == ABC_INTERNAL_RETRY) {          // a bin of its exact step outgrew LDS (massive ties): once more in stream order, on the sorted path
            ABC_HIP(ctx, hipStreamSynchronize(ctx->wx_stream));
            ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            rc = launch_wilcoxon(ctx, io->X, io->Y, N, N, N, M, P, A, (size_t)ntrain, model);
            changed = 2;
        }
        if (rc != ABC_OK) { if (!ctx->err[0]) snprintf(ctx->err, sizeof(ctx->err), "generation: the component rule's reduction failed"); return rc; }
        if (changed == 1) {                      // the counts AND the header, in front of the ranking's second run
            ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_wx_done, 0));
            ABC_TRY(launch_wilcoxon_commit(ctx, model, M, P, A, wx_dec, 1));
        } else if (changed == 0) {
            // the per-response counts into the model record on the cascade's own stream: nothing this generation still queues reads
            // them (the largest count, which everything used, is the fit's); the host waits for that stream at the generation's end
            ctx->stream = ctx->wx_stream;
            const int rcc = launch_wilcoxon_commit(ctx, model, M, P, A, wx_dec, 0);
            ctx->stream = main_stream;
            ABC_TRY(rcc);
            wx_tail_pending = true;
        }
        if (changed) {                           // the largest count moved: the ranking once more, with it
            ctx->wx_moved_counts++;
            // (the second selection's give-up flag goes to a word of its own in the pinned block: the first gather may still be
            // running -- resetting ITS word would need a wait for the stream here, ~15 us of idle GPU in front of the repeat)
            pfail_early = (int*)(ctx->status_pin + 96);
            *pfail_early = 0;
            if (scores_all) ABC_TRY(launch_distance_from_scores(ctx, scores_all, N, N, M, P, A, model, dist));
            else ABC_TRY(launch_project_distance(ctx, io->X, N, N, M, P, A, model, 0, dist));
            ABC_TRY(launch_select_smallest(ctx, dist, N, K, 0, io->idx, io->dist, /*defer_check=*/true));
            bins_deferred = ctx->sel_bins_ran && ctx->sel_fail_dev && !ctx->sel_force_radix;
            ABC_TRY(launch_gather_rows(ctx, io->Y, N, N, P, io->idx, K, 0, theta, K, bins_deferred ? ctx->sel_fail_dev : nullptr, pfail_early,
                                       theta_ev_bound ? ctx->ev_theta : nullptr));
        }
    }
    // A failed bin selection (degenerate distances) leaves placeholder winners: everything downstream of it is repeated with
    // the radix select.  Weighted generations learn of it at the host's wait for the weights (launch_resample's abort flag),
    // before the alias table, the draws and the proposals of the placeholder are queued; set 0 has no host wait before its
    // end and finds out there.  Either way the proposals' give-up counter is put back to its snapshot.
    auto repeat_generation = [&](bool radix, bool wx_in_order) -> int {
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->side) ABC_HIP(ctx, hipStreamSynchronize(ctx->side));
        if (ctx->wx_stream) ABC_HIP(ctx, hipStreamSynchronize(ctx->wx_stream));
        if (ctx->giveups_dev)
            ABC_HIP(ctx, hipMemcpyAsync(ctx->giveups_dev, ctx->giveups_dev + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
        ctx->sel_bins_ran = false;
        ctx->ws_off = ws_entry;
        if (rng) *rng = rng_entry;
        ctx->generation_repeats++;
        const bool radix0 = ctx->sel_force_radix, inline0 = ctx->wx_force_inline;      // (a repeat inside a repeat keeps the outer one's reason)
        ctx->sel_force_radix = radix0 || radix;
        ctx->wx_force_inline = inline0 || wx_in_order;
        const int rc = generation_core(ctx, cfg, io, rng, ncomp_host, simple, model_out);
        ctx->sel_force_radix = radix0;
        ctx->wx_force_inline = inline0;
        return rc;
    };
    auto repeat_with_radix = [&]() -> int { return repeat_generation(true, false); };
    double* dv = io->dv ? io->dv : (double*)abc_ws_alloc(ctx, P * 8);
    double* theta_stats = nullptr;        // moments of the posterior: shared by dv and the MVN factor
    // Weighted generations with proposals: the kernel density of the weights uses the PREVIOUS set's variance, so the new set's
    // moments (pilot shift, Gram, reduce, dv: 25 us of small launches) are not needed before the host's alias round trip --
    // they run in the GPU's idle time behind it (hook below) instead of in front of the pair sums.
    // (set 0 too: nothing in front of the proposals needs them, and up to 16 parameters ONE launch then delivers the moments,
    // the doubled variance, the proposal factor and the perturbation's row-major copy: k_theta_moments)
    if (defer_moments) {
    } else if (P <= 64 && K >= 2) {
        StageTimer tm(ctx, ST_GATHER_DV);
        ABC_TRY(launch_theta_stats(ctx, theta, K, P, &theta_stats));
        ABC_TRY(launch_dv_from_stats(ctx, theta_stats, P, dv));
    } else {
        ABC_TRY(launch_doubled_variance(ctx, theta, K, P, dv));
    }
    bool w_on_host = false;
    if (Kp == 0 || !io->theta_prev) {
        if (!filled_early) ABC_TRY(launch_fill(ctx, io->w, K, 1.0 / (double)K));                 // AbcUtil.cpp:543-544
    } else {
        if (wprev.ready && !ctx->side_early_waited) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prev, 0));
        const double* sumsq = nullptr;       // the normalisation's sum of squares comes out of the weight stage's last kernel
        ABC_TRY(launch_weights_raw(ctx, io->priors, theta, K, P, 0, K, io->theta_prev, Kp, io->w_prev, io->dv_prev,
                                   io->w, &wprev, &sumsq));
        // (with proposals to draw the host builds the alias table of these weights next: the normalisation kernel stores them
        // into the pinned scratch as it writes them -- launch_resample then has nothing to copy)
        double* mirror = nullptr;
        const bool alias_on_device = ctx->alias_mode == ABC_ALIAS_DEVICE && K >= ABC_ALIAS_DEV_MIN_K && K <= ABC_ALIAS_DEV_MAX_K;
        if (Nn && !alias_on_device) {          // (the device build reads the weights where they are)
            ABC_TRY(abc_pin_reserve(ctx, abc_alias_pin_bytes(K)));
            mirror = (double*)ctx->pin;
        }
        ABC_TRY(launch_normalize_l2(ctx, io->w, K, mirror, sumsq));           // AbcUtil.cpp:583
        w_on_host = mirror != nullptr;
    }
    // (queued BEHIND the weight stage's launches since round 5: with the ranking speculating beside the Wilcoxon cascade the host
    // arrives here late, and the side stream's four launches in front of them delayed the pair sums by their enqueue time)
    // The posterior's moments and what follows from them (doubled variance, proposal factor, the perturbation's row-major copy
    // and padded factor) need the gathered rows only: with the resampling table built on the device nothing waits for the host
    // any more, so they run on the SIDE stream from here on, beside the weight stage, and are long done when the proposals need
    // them (round 2 hid them behind the host's alias build).
    double* L_early = nullptr;
    abc_theta_fused side_out = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool moments_on_side = false, seeds_waited = false;
    // (set 0 has nothing to overlap them with: two cross-stream hand-overs for nothing, measured +35 us)
    if (moments_side_planned) {
        if (cfg->multivariate) {
            L_early = io->L ? io->L : (double*)abc_ws_alloc(ctx, P * P * 8);
            if (!L_early) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
        }
        const int PPr = abc_perturb_pp(P);
        side_out.dv = dv; side_out.L = L_early; side_out.spd = spd_dev;
        if (!status_late) { side_out.model_hdr = simple ? nullptr : model; side_out.hdr_pin = hdr_pin; side_out.spd_pin = L_early ? spd_pin : nullptr; status_early = true; }
        side_out.rows = (double*)abc_ws_alloc(ctx, K * (size_t)PPr * sizeof(double));
        if (L_early) side_out.Lpad = (double*)abc_ws_alloc(ctx, (size_t)PPr * PPr * sizeof(double));
        if (!side_out.rows || (L_early && !side_out.Lpad)) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
        if (!theta_ev_bound) ABC_HIP(ctx, hipEventRecord(ctx->ev_theta, ctx->stream));
        ABC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_theta, 0));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->side;                        // the launchers below queue on the context's stream
        double* st = nullptr;
        int rc = launch_theta_stats(ctx, theta, K, P, &st);
        if (rc == ABC_OK) rc = launch_post_tail(ctx, theta, K, P, st, &side_out);
        ctx->stream = main_stream;
        ABC_TRY(rc);
        ABC_HIP(ctx, hipEventRecord(ctx->ev_moments, ctx->side));
        theta_stats = st;
        moments_on_side = true;
    }
    int spd = 0;
    bool have_spd = false;
    int alias_deferred = 0;
    uint64_t* parent_used = nullptr;
    double* L_used = nullptr;
    abc_perturb_prep prep_used = {nullptr, 0, nullptr};
    if (Nn) {
        uint64_t* parent = parent_early ? parent_early : (io->parent ? io->parent : (uint64_t*)abc_ws_alloc(ctx, Nn * 8));
        if (!parent) ABC_FAIL(ctx, ABC_ERR_NOMEM, "generation: workspace exhausted");
        double* L = nullptr;
        if (cfg->multivariate) {
            L = L_early ? L_early : (io->L ? io->L : (double*)abc_ws_alloc(ctx, P * P * 8));
            have_spd = true;
        }
        // The alias-table host round trip sits inside launch_resample.  What does not depend on the weights runs on the GPU
        // meanwhile: the MVN factor (covariance + Cholesky), the row-major posterior copy and the seed stream of the
        // perturbation.
        abc_perturb_prep prep = {moments_on_side ? side_out.rows : nullptr, (early && io->seeds) ? 1 : 0, moments_on_side ? side_out.Lpad : nullptr};
        if (moments_on_side) { ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_moments, 0)); seeds_waited = true; }   // (recorded behind the seeds)
        struct PrepArg {
            abc_ctx* ctx; const abc_rng* rng; const double* theta; const double* theta_stats; size_t K, P, Nn;
            uint64_t* seeds; abc_perturb_prep* prep; double* L; int* spd_dev; double* dv; bool moments;
            const double* model_hdr; double* hdr_pin; int* spd_pin;
        };
        const bool hook_moments = defer_moments && !moments_on_side;
        PrepArg pa = {ctx, rng, theta, theta_stats, K, P, Nn, io->seeds, &prep, moments_on_side ? nullptr : L, spd_dev, dv,
                      hook_moments, nullptr, nullptr, nullptr};
        if (hook_moments && !status_late) {
            pa.model_hdr = simple ? nullptr : model; pa.hdr_pin = hdr_pin; pa.spd_pin = L ? spd_pin : nullptr;
            status_early = true;
        }
        auto hook = [](void* a) -> int {
            PrepArg* q = (PrepArg*)a;
            int fused_done = 0;
            if (q->moments) {
                StageTimer tm(q->ctx, ST_GATHER_DV);
                double* st = nullptr;
                ABC_TRY(launch_theta_stats(q->ctx, q->theta, q->K, q->P, &st));
                q->theta_stats = st;
                // doubled variance, proposal factor and the perturbation's inputs (row-major copy, padded factor) in ONE launch
                abc_theta_fused f = {q->dv, q->L, q->spd_dev, nullptr, nullptr, q->model_hdr, q->hdr_pin, q->spd_pin};
                if (q->ctx->noise_mode != ABC_NOISE_REFERENCE_STREAM) {
                    const int PP = abc_perturb_pp(q->P);
                    f.rows = (double*)abc_ws_alloc(q->ctx, q->K * (size_t)PP * sizeof(double));
                    if (q->L) f.Lpad = (double*)abc_ws_alloc(q->ctx, (size_t)PP * PP * sizeof(double));
                    if (!f.rows || (q->L && !f.Lpad)) { snprintf(q->ctx->err, sizeof(q->ctx->err), "generation: workspace exhausted"); return ABC_ERR_NOMEM; }
                }
                ABC_TRY(launch_post_tail(q->ctx, q->theta, q->K, q->P, st, &f));
                q->prep->rows = f.rows;
                q->prep->Lpad = f.Lpad;
                fused_done = 1;
            }
            if (q->L && !fused_done) {
                if (q->theta_stats) {
                    StageTimer tm(q->ctx, ST_MVN);
                    ABC_TRY(launch_mvn_from_stats(q->ctx, q->theta_stats, q->P, q->L, q->spd_dev));
                } else {
                    ABC_TRY(launch_mvn_setup(q->ctx, q->theta, q->K, q->P, q->L, nullptr, q->spd_dev));
                }
            }
            if (q->ctx->noise_mode == ABC_NOISE_REFERENCE_STREAM) return ABC_OK;     // nothing of the device stream is needed
            // ... and the row-major posterior copy, the padded factor and the seeds of the proposals
            return launch_perturb_prepare(q->ctx, q->rng, q->theta, q->K, q->P, 0, q->Nn, q->seeds, q->Nn, q->prep,
                                          q->L ? 1 : 0, q->L ? q->L : q->dv);
        };
        {
            const int rc = launch_resample(ctx, rng, io->w, K, 0, Nn, parent, hook, &pa, uniform_w, raw_early, w_on_host,
                                           bins_deferred ? pfail_early : nullptr, parent_early != nullptr,
                                           ctx->noise_mode == ABC_NOISE_REFERENCE_STREAM ? nullptr : &alias_deferred);
            if (rc == ABC_INTERNAL_RETRY) return repeat_with_radix();
            ABC_TRY(rc);
        }
        if (ctx->noise_mode == ABC_NOISE_REFERENCE_STREAM) {
            taus2_jump(rng, (uint64_t)Nn);   // the Nnext resampling draws; the host loop consumes the rest as the reference does
            ABC_TRY(launch_perturb_reference(ctx, rng, theta, K, P, io->priors, parent, Nn, cfg->multivariate,
                                             cfg->multivariate ? L : dv, io->next, io->seeds));
        } else {
            ABC_TRY(launch_perturb(ctx, rng, theta, K, P, io->priors, parent, 0, Nn, cfg->multivariate,
                                   cfg->multivariate ? L : dv, io->next, io->seeds, Nn, &prep));
            parent_used = parent; L_used = L; prep_used = prep;
            taus2_jump(rng, 2 * (uint64_t)Nn);   // Nnext resampling draws + Nnext seeds
        }
    }
    if (wx_defer) {
        // the cascade's second half (see above): EVERYTHING of this generation is queued by now, the pair sums run.  (Queued in
        // front of the resampling stage, first version, the host's waits for the cascade's levels -- whose kernels get hardly any
        // CU while the pair sums' work-groups hold them all -- kept the resampling table from being queued in time: 0.34 ms of
        // idle main stream behind the weights at configs[4], rocprofv3 timeline)
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->wx_stream;
        int changed = 2;
        int rc = launch_wilcoxon_finish(ctx, wx_run, &changed);
        wx_run = nullptr;
        if (rc == ABC_OK && hipEventRecord(ctx->ev_wx_done, ctx->wx_stream) != hipSuccess) rc = ABC_ERR_HIP;
        ctx->stream = main_stream;
        // the largest count moved (or a bin of the exact step outgrew LDS: massive ties): what was queued ranked on the wrong count
        if (rc == ABC_INTERNAL_RETRY || (rc == ABC_OK && changed)) return repeat_generation(false, true);
        if (rc != ABC_OK) { if (!ctx->err[0]) snprintf(ctx->err, sizeof(ctx->err), "generation: the component rule's reduction failed"); return rc; }
        // the per-response counts into the model record, on the cascade's own stream: nothing this generation still queues reads
        // them (the largest count, which everything used, is the fit's), and on the main stream the launch sat between the
        // normalised weights and the resampling table -- 16 us of the critical path (rocprofv3 timeline).  The host waits for that
        // stream at the generation's end.
        ctx->stream = ctx->wx_stream;
        rc = launch_wilcoxon_commit(ctx, model, M, P, A, wx_dec, 0);
        ctx->stream = main_stream;
        ABC_TRY(rc);
        wx_tail_pending = true;
    }
    {
        // status words into the pinned block: [0..31] model header (component count), [32] Cholesky status, [36] selection flag
        double* hdr = (double*)ctx->status_pin;
        int* pspd = (int*)(ctx->status_pin + 32);
        int* pfail = (int*)(ctx->status_pin + 36);
        if (!status_early) { hdr[0] = 0.0; *pspd = 0; }
        *pfail = 0;
        if (seeds_late && !seeds_waited) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_side, 0));     // (never in practice: see below)
        // ONE tiny kernel stores the three words into the (device-visible) pinned block: three copies were three blit launches
        const int* fail_dev = (ctx->sel_bins_ran && ctx->sel_fail_dev) ? (const int*)ctx->sel_fail_dev : nullptr;
        unsigned long long* pgive = (unsigned long long*)(ctx->status_pin + 48);
        *pgive = 0;
        if (!status_early) {
            hipLaunchKernelGGL(k_status_words, dim3(1), dim3(64), 0, ctx->stream, simple ? (const double*)nullptr : (const double*)model,
                               have_spd ? (const int*)spd_dev : (const int*)nullptr, fail_dev, (const unsigned long long*)ctx->giveups_dev, hdr, pspd,
                               pfail, pgive);
            ABC_HIP(ctx, hipGetLastError());
        }
        ctx->side_early_waited = false;
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (wx_tail_pending) ABC_HIP(ctx, hipStreamSynchronize(ctx->wx_stream));
        if (status_early) {
            // header and Cholesky status came from k_post_tail, the selection's flag from the gather (bins_deferred), the proposals'
            // give-up counter is read only when a proposal kernel raised the flag word
            *pfail = fail_dev ? *(volatile int*)pfail_early : 0;
        } else {
            ctx->giveups_dev_known = *(volatile unsigned long long*)pgive;
        }
        if (ncomp_host) *ncomp_host = (int32_t)hdr[0];
        spd = *pspd;
        // the sampled-range bin selection gave up (degenerate distances, an atypical sample): everything downstream of it
        // worked on a placeholder; once more, from the top, with the radix select
        const int failed = abc_select_check_done(ctx, pfail);
        if (failed && !ctx->sel_force_radix) return repeat_with_radix();
        // the device build of the resampling table did not verify: the draws and the proposals once more, with the table from the
        // host (the weights are final; only what depends on the table is repeated)
        if (alias_deferred && *(volatile int*)(ctx->status_pin + 44) && parent_used) {
            ctx->alias_dev_fallbacks++;
            const int mode = ctx->alias_mode;
            ctx->alias_mode = ABC_ALIAS_HOST;
            int rc = launch_resample(ctx, &rng_entry, io->w, K, 0, Nn, parent_used);
            ctx->alias_mode = mode;
            ABC_TRY(rc);
            prep_used.seeds_done = 1;
            // (the first pass's give-ups belong to proposals that are being replaced: counter back to the generation's snapshot)
            if (ctx->giveups_dev)
                ABC_HIP(ctx, hipMemcpyAsync(ctx->giveups_dev, ctx->giveups_dev + 1, sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
            ABC_TRY(launch_perturb(ctx, &rng_entry, theta, K, P, io->priors, parent_used, 0, Nn, cfg->multivariate,
                                   cfg->multivariate ? L_used : dv, io->next, nullptr, Nn, &prep_used));
            ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    if (ctx->timing && ctx->nev > 128) ABC_TRY(abc_timing_flush(ctx));
    if (spd) ABC_FAIL(ctx, ABC_ERR_NOT_SPD, "covariance of the selected particles is not positive definite");
    // proposals the perturbation gave up on during THIS call (the reference never returns in that case, AbcUtil.cpp:132): the
    // outputs are complete -- such a row is its (valid) parent, or the prior mean in INDEPENDENT mode -- and the caller is told
    {
        if (status_early) {
            if (*pgaveup && ctx->giveups_dev) {           // (rare) a proposal kernel gave up: fetch the counter
                unsigned long long now = 0;
                ABC_HIP(ctx, hipMemcpyAsync(&now, ctx->giveups_dev, sizeof(now), hipMemcpyDeviceToHost, ctx->stream));
                ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
                ctx->giveups_dev_known = now;
            }
            *(volatile unsigned long long*)(ctx->status_pin + 48) = ctx->giveups_dev_known;
        }
        const unsigned long long gv = *(volatile unsigned long long*)(ctx->status_pin + 48) + ctx->giveups_host;
        const unsigned long long before = ctx->giveups_seen;
        ctx->giveups_seen = gv;
        // (status stays ABC_OK: a C caller's `if (rc)` must not read a finished generation as a failure; abc_generation_giveups)
        ctx->giveups_last_call = (Nn && gv > before) ? gv - before : 0ull;
    }
    return ABC_OK;
}

extern "C" int abc_generation_dev(abc_ctx* ctx, const abc_generation_cfg* cfg, const abc_generation_io* io, abc_rng* rng,
                                  int32_t* ncomp_host) {
    CHECK_CTX(ctx);
    if (!cfg || !io || !io->X || !io->obs || !io->idx) ABC_FAIL(ctx, ABC_ERR_INVALID, "generation: null argument");
    const size_t A = default_A(cfg->M, cfg->P, cfg->max_comp);
    size_t need = abc_ws_need(cfg->N, cfg->M, cfg->P, A, cfg->K, cfg->Kp, cfg->Nnext);
    if (cfg->rule == ABC_RULE_WILCOXON) need += abc_wx_need(cfg->N, cfg->P, A) + cfg->N * A * 8 + 4096;      // (+ the scores of all rows)
    ABC_TRY(abc_ws_reserve(ctx, need));
    return generation_core(ctx, cfg, io, rng, ncomp_host, 0);
}

// ---- host-pointer entry points ---------------------------------------------------------------------
namespace {
struct Stage {   // host<->device staging inside the arena
    abc_ctx* ctx;
    template <typename T>
    T* up(const T* h, size_t n) {
        T* d = (T*)abc_ws_alloc(ctx, n * sizeof(T));
        if (d && h && n) (void)hipMemcpyAsync(d, h, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
        return d;
    }
    template <typename T>
    T* dev(size_t n) { return (T*)abc_ws_alloc(ctx, n * sizeof(T)); }
    template <typename T>
    void down(T* h, const T* d, size_t n) {
        if (h && d && n) (void)hipMemcpyAsync(h, d, n * sizeof(T), hipMemcpyDeviceToHost, ctx->stream);
    }
};
}  // namespace

static int ranking_host(abc_ctx* ctx, const double* X, const double* Y, const double* obs, size_t N, size_t M,
                        size_t P, double train_frac, int max_comp, int rule, size_t K, uint64_t* idx, double* dist,
                        int32_t* ncomp, double* R, double* mean, double* sd, int simple) {
    if (!X || !obs || !idx || (!simple && !Y)) ABC_FAIL(ctx, ABC_ERR_INVALID, "ranking: null argument");
    const size_t A = simple ? 0 : default_A(M, P, max_comp);
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(N, M, P, A, K, 0, 0) + (N * (M + P) + M + 2 * K) * 8 +
                                    ((!simple && rule == ABC_RULE_WILCOXON) ? abc_wx_need(N, P, A) : 0)));
    Stage s{ctx};
    abc_generation_io io;
    memset(&io, 0, sizeof(io));
    io.X = s.up(X, N * M);
    io.Y = simple ? nullptr : s.up(Y, N * P);
    io.obs = s.up(obs, M);
    io.idx = s.dev<uint64_t>(K);
    io.dist = s.dev<double>(K);
    if (!io.X || !io.obs || !io.idx || !io.dist) ABC_FAIL(ctx, ABC_ERR_NOMEM, "ranking: workspace exhausted");
    abc_generation_cfg cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.N = N; cfg.M = M; cfg.P = simple ? 0 : P; cfg.K = K; cfg.train_frac = train_frac;
    cfg.max_comp = max_comp; cfg.rule = rule;
    const double* model = nullptr;
    ABC_TRY(generation_core(ctx, &cfg, &io, nullptr, ncomp, simple, &model));
    s.down(idx, io.idx, K);
    s.down(dist, io.dist, K);
    if (R || mean || sd) {
        const ModelLayout ML = model_layout(M, simple ? 0 : P, A);
        if (R && !simple) s.down(R, model + ML.off_R, M * A);
        s.down(mean, model + ML.off_mean, M);
        s.down(sd, model + ML.off_sd, M);
    }
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

extern "C" int abc_particle_ranking_pls(abc_ctx* ctx, const double* X, const double* Y, const double* obs, size_t N,
                                        size_t M, size_t P, double train_frac, int max_comp, int rule, size_t K,
                                        uint64_t* idx, double* dist, int32_t* ncomp, double* R, double* mean,
                                        double* sd) {
    CHECK_CTX(ctx);
    return ranking_host(ctx, X, Y, obs, N, M, P, train_frac, max_comp, rule, K, idx, dist, ncomp, R, mean, sd, 0);
}

extern "C" int abc_particle_ranking_simple(abc_ctx* ctx, const double* X, const double* obs, size_t N, size_t M,
                                           size_t K, uint64_t* idx, double* dist) {
    CHECK_CTX(ctx);
    return ranking_host(ctx, X, nullptr, obs, N, M, 0, 1.0, 0, 0, K, idx, dist, nullptr, nullptr, nullptr, nullptr, 1);
}

extern "C" int abc_calculate_doubled_variance(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* dv) {
    CHECK_CTX(ctx);
    if (!theta || !dv) ABC_FAIL(ctx, ABC_ERR_INVALID, "doubled_variance: null argument");
    ABC_TRY(abc_ws_reserve(ctx, (K * P + P) * 8 + (1 << 20)));
    Stage s{ctx};
    double* dth = s.up(theta, K * P);
    double* ddv = s.dev<double>(P);
    ABC_TRY(launch_doubled_variance(ctx, dth, K, P, ddv));
    s.down(dv, ddv, P);
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

extern "C" int abc_weight_predictive_prior_uniform(abc_ctx* ctx, size_t K, double* w) {
    CHECK_CTX(ctx);
    if (!w || !K) ABC_FAIL(ctx, ABC_ERR_INVALID, "weights: null argument");
    ABC_TRY(abc_ws_reserve(ctx, K * 8 + (1 << 20)));
    Stage s{ctx};
    double* dw = s.dev<double>(K);
    ABC_TRY(launch_fill(ctx, dw, K, 1.0 / (double)K));
    s.down(w, dw, K);
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

extern "C" int abc_weight_predictive_prior(abc_ctx* ctx, const abc_prior* priors, const double* theta, size_t K,
                                           size_t P, const double* theta_prev, size_t Kp, const double* w_prev,
                                           const double* dv_prev, double* w) {
    CHECK_CTX(ctx);
    if (!priors || !theta || !theta_prev || !w_prev || !dv_prev || !w)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "weights: null argument");
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, 1, P, 1, K, Kp, 0) + ((K + Kp) * (P + 1) + 2 * P) * 8 + P * sizeof(abc_prior)));
    Stage s{ctx};
    abc_prior* dpr = s.up(priors, P);
    double* dth = s.up(theta, K * P);
    double* dtp = s.up(theta_prev, Kp * P);
    double* dwp = s.up(w_prev, Kp);
    double* ddv = s.up(dv_prev, P);
    double* dw = s.dev<double>(K);
    if (!dpr || !dth || !dtp || !dwp || !ddv || !dw) ABC_FAIL(ctx, ABC_ERR_NOMEM, "weights: workspace exhausted");
    ABC_TRY(launch_weights_raw(ctx, dpr, dth, K, P, 0, K, dtp, Kp, dwp, ddv, dw));
    ABC_TRY(launch_normalize_l2(ctx, dw, K));
    s.down(w, dw, K);
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

extern "C" int abc_setup_mvn_sampler(abc_ctx* ctx, const double* theta, size_t K, size_t P, double* L) {
    CHECK_CTX(ctx);
    if (!theta || !L) ABC_FAIL(ctx, ABC_ERR_INVALID, "mvn: null argument");
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, P, 0, 1, 0, 0, 0) + (K * P + P * P) * 8));
    Stage s{ctx};
    double* dth = s.up(theta, K * P);
    double* dL = s.dev<double>(P * P);
    int st = 0;
    ABC_TRY(launch_mvn_setup(ctx, dth, K, P, dL, &st, nullptr));
    if (st) ABC_FAIL(ctx, ABC_ERR_NOT_SPD, "covariance of the selected particles is not positive definite");
    s.down(L, dL, P * P);
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

extern "C" int abc_sample_posterior(abc_ctx* ctx, abc_rng* rng, const double* w, size_t K, size_t n, uint64_t* idx) {
    CHECK_CTX(ctx);
    if (!rng || !w || !idx) ABC_FAIL(ctx, ABC_ERR_INVALID, "sample_posterior: null argument");
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, 1, 1, 1, 0, 0, n) + K * 8 + n * 8 + abc_alias_dev_need(K)));
    Stage s{ctx};
    double* dw = s.up(w, K);
    uint64_t* dp = s.dev<uint64_t>(n);
    ABC_TRY(launch_resample(ctx, rng, dw, K, 0, n, dp));
    s.down(idx, dp, n);
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    taus2_jump(rng, n);
    return ABC_OK;
}

// the resampling table as this context builds it (for inspection / tests): F with GSL's KNUTH_CONVENTION applied
extern "C" int abc_alias_table(abc_ctx* ctx, const double* w, size_t K, double* F, uint64_t* A, int* on_device) {
    CHECK_CTX(ctx);
    if (!w || !F || !A || !K || K > 0xffffffffull) ABC_FAIL(ctx, ABC_ERR_INVALID, "alias table: null argument or K = %zu", K);
    if (on_device) *on_device = 0;
    std::vector<double> hF(K);
    std::vector<uint32_t> hA(K);
    bool done = false;
    if (ctx->alias_mode == ABC_ALIAS_DEVICE && K >= 2 && K <= ABC_ALIAS_DEV_MAX_K) {
        ABC_TRY(abc_ws_reserve(ctx, abc_alias_dev_need(K) + K * 20 + (1u << 20)));
        Stage s{ctx};
        double* dw = s.up(w, K);
        double* dF = s.dev<double>(K);
        uint32_t* dA = s.dev<uint32_t>(K);
        int* dfail = s.dev<int>(1);
        if (!dw || !dF || !dA || !dfail) ABC_FAIL(ctx, ABC_ERR_NOMEM, "alias table: workspace exhausted");
        ABC_TRY(launch_alias_build_dev(ctx, dw, K, dF, dA, dfail, nullptr));
        int fail = 0;
        ctx->alias_dev_builds++;
        s.down(hF.data(), (const double*)dF, K);
        s.down(hA.data(), (const uint32_t*)dA, K);
        s.down(&fail, (const int*)dfail, 1);
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (fail) ctx->alias_dev_fallbacks++; else done = true;
        if (done && on_device) *on_device = 1;
    }
    if (!done) {
        std::vector<double> E(K);
        std::vector<uint32_t> S(K + 1), B(K + 1);
        abc_alias_preproc(K, w, hF.data(), hA.data(), E.data(), S.data(), B.data(), /*knuth=*/false);
    }
    const double dK = (double)K;
    for (size_t k = 0; k < K; k++) { F[k] = (hF[k] + (double)k) / dK; A[k] = hA[k]; }      // KNUTH_CONVENTION, as k_alias_draw applies it
    return ABC_OK;
}

static int sample_host(abc_ctx* ctx, abc_rng* rng, size_t n, const double* w, const double* theta, size_t K, size_t P,
                       const abc_prior* priors, const double* L_or_dv, int multivariate, double* out, uint64_t* parent,
                       uint64_t* seeds) {
    if (!rng || !w || !theta || !priors || !L_or_dv || !out) ABC_FAIL(ctx, ABC_ERR_INVALID, "sample: null argument");
    ABC_TRY(abc_ws_reserve(ctx, abc_ws_need(0, 1, P, 1, K, 0, n) + (K * (P + 1) + P * P + n * (P + 2)) * 8 + P * sizeof(abc_prior)));
    Stage s{ctx};
    double* dw = s.up(w, K);
    double* dth = s.up(theta, K * P);
    abc_prior* dpr = s.up(priors, P);
    double* dl = s.up(L_or_dv, multivariate ? P * P : P);
    double* dout = s.dev<double>(n * P);
    uint64_t* dpar = s.dev<uint64_t>(n);
    uint64_t* dseed = seeds ? s.dev<uint64_t>(n) : nullptr;
    if (!dw || !dth || !dpr || !dl || !dout || !dpar) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sample: workspace exhausted");
    ABC_TRY(launch_resample(ctx, rng, dw, K, 0, n, dpar));
    if (ctx->noise_mode == ABC_NOISE_REFERENCE_STREAM) {
        taus2_jump(rng, (uint64_t)n);
        ABC_TRY(launch_perturb_reference(ctx, rng, dth, K, P, dpr, dpar, n, multivariate, dl, dout, dseed));
    } else {
        ABC_TRY(launch_perturb(ctx, rng, dth, K, P, dpr, dpar, 0, n, multivariate, dl, dout, dseed, n));
        taus2_jump(rng, seeds ? 2 * (uint64_t)n : (uint64_t)n);
    }
    s.down(out, dout, n * P);
    s.down(parent, dpar, n);
    s.down(seeds, dseed, n);
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ABC_OK;
}

extern "C" int abc_sample_mvn_predictive_priors(abc_ctx* ctx, abc_rng* rng, size_t n, const double* w,
                                                const double* theta, size_t K, size_t P, const abc_prior* priors,
                                                const double* L, double* out, uint64_t* parent, uint64_t* seeds) {
    CHECK_CTX(ctx);
    return sample_host(ctx, rng, n, w, theta, K, P, priors, L, 1, out, parent, seeds);
}

extern "C" int abc_sample_predictive_priors(abc_ctx* ctx, abc_rng* rng, size_t n, const double* w, const double* theta,
                                            size_t K, size_t P, const abc_prior* priors, const double* dv, double* out,
                                            uint64_t* parent, uint64_t* seeds) {
    CHECK_CTX(ctx);
    return sample_host(ctx, rng, n, w, theta, K, P, priors, dv, 0, out, parent, seeds);
}
