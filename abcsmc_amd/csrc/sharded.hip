// Row-sharded generation over several GPUs: communicators (RCCL over xGMI, or collectives supplied by the
// caller) and the protocol of one generation turn-over, in C++ behind the C ABI (include/abcsmc_hip.h, "Multi-GPU").
// The reference has no device or multi-device path; what is sharded is the per-generation numerical path of
// AbcSmc::process_database (AbcSmc.cpp:634-664 rank + truncate, :1041-1066 weights, :490-518 proposals, :535 seeds).
//
// Exchange steps of one generation (everything else is local to a rank):
//   1. broadcast of rank 0's pilot shift (16 ceil((M+P)/16) doubles)
//   2. ONE packed all-reduce of the sufficient-statistics record (counts, column sums, Gram blocks; <= 0.35 MB); the
//      model fit is then replicated (deterministic: identical on every rank)
//   3. exact distributed radix select: six all-reduces of a 2048-bin histogram give every rank the global K-th
//      distance; an all-gather of (#below, #equal) decides how many ties each rank keeps (lowest global rows first)
//   4. all-gather of the per-rank winner lists (dist, global row), each sorted, padded to the longest; every rank merges
//      the runs (stable: equal distances keep rank = row order) -> the K selected rows in ascending (distance, row) order
//   5. all-gather of the winners' parameter rows, packed per rank in its run order; the merge's source map puts them in
//      rank order (K x P posterior on every rank: the perturbation reads arbitrary parents)
//   6. all-gather of the per-rank slices of the raw importance weights (pair sums sharded K / G rows per rank)
//   7. none for resampling / perturbation / seeds: every rank regenerates its slice of the sequential taus2 stream by
//      jump-ahead.
// The only host round trip beyond abc_generation_dev's own (alias table, final status) is the 16 G bytes of step 3's
// counts: the lengths of the exchanged winner lists have to be known to the host that posts the all-gathers.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "abc_internal.h"
#include <hip/hip_ext.h>

namespace {

// ---- RCCL, loaded on first use (the library stays loadable, and single-GPU use needs nothing of it) -------------------
struct Rccl {
    void* h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
Rccl* rccl() {
    static Rccl R = [] {
        Rccl r;
        // an RCCL the process has already loaded comes first (PyTorch ships its own librccl.so: one library, one set of
        // proxy threads and IPC state for both its communicators and this one), then the system's
        for (const char* name : {"librccl.so", "librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (r.h) break;
        }
        if (!r.h)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.h) break;
            }
        if (!r.h) return r;
#define RCCL_SYM(f) r.f = (decltype(r.f))dlsym(r.h, "nccl" #f)
        RCCL_SYM(GetUniqueId); RCCL_SYM(CommInitRank); RCCL_SYM(CommInitAll); RCCL_SYM(CommDestroy); RCCL_SYM(CommAbort); RCCL_SYM(AllReduce);
        RCCL_SYM(AllGather); RCCL_SYM(Broadcast); RCCL_SYM(GetErrorString);
#undef RCCL_SYM
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.AllReduce && r.AllGather && r.Broadcast;
        return r;
    }();
    return &R;
}
static_assert(sizeof(ncclUniqueId) == ABC_COMM_ID_BYTES, "ABC_COMM_ID_BYTES must match ncclUniqueId");

#define ABC_NCCL(ctx, call)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess)                                                                                \
            ABC_FAIL(ctx, ABC_ERR_COMM, "%s:%d %s -> %s", __FILE__, __LINE__, #call,                          \
                     rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "RCCL error");                    \
    } while (0)

// ---- collectives on the context's stream ----------------------------------------------------------------------------------
int comm_all_reduce(abc_ctx* ctx, void* buf, size_t count, int dtype) {
    if (ctx->comm_kind == 0 || count == 0) return ABC_OK;
    StageTimer tm(ctx, ST_COMM);
    if (ctx->comm_kind == 1) {
        const ncclDataType_t dt = dtype == ABC_DT_F64 ? ncclFloat64 : dtype == ABC_DT_I32 ? ncclInt32 : ncclInt64;
        ABC_NCCL(ctx, rccl()->AllReduce(buf, buf, count, dt, ncclSum, (ncclComm_t)ctx->comm_nccl, ctx->stream));
        return ABC_OK;
    }
    if (ctx->comm_cb.all_reduce_sum(ctx->comm_cb.user, buf, count, dtype, (void*)ctx->stream))
        ABC_FAIL(ctx, ABC_ERR_COMM, "all_reduce callback failed");
    return ABC_OK;
}
int comm_all_gather(abc_ctx* ctx, const void* send, void* recv, size_t bytes) {
    if (bytes == 0) return ABC_OK;
    StageTimer tm(ctx, ST_COMM);
    if (ctx->comm_kind == 0) {
        if (send != recv) ABC_HIP(ctx, hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        return ABC_OK;
    }
    if (ctx->comm_kind == 1) {
        ABC_NCCL(ctx, rccl()->AllGather(send, recv, bytes, ncclUint8, (ncclComm_t)ctx->comm_nccl, ctx->stream));
        return ABC_OK;
    }
    if (ctx->comm_cb.all_gather(ctx->comm_cb.user, send, recv, bytes, (void*)ctx->stream))
        ABC_FAIL(ctx, ABC_ERR_COMM, "all_gather callback failed");
    return ABC_OK;
}
int comm_broadcast(abc_ctx* ctx, void* buf, size_t bytes, int root) {
    if (ctx->comm_kind == 0 || bytes == 0) return ABC_OK;
    StageTimer tm(ctx, ST_COMM);
    if (ctx->comm_kind == 1) {
        ABC_NCCL(ctx, rccl()->Broadcast(buf, buf, bytes, ncclUint8, root, (ncclComm_t)ctx->comm_nccl, ctx->stream));
        return ABC_OK;
    }
    if (ctx->comm_cb.broadcast(ctx->comm_cb.user, buf, bytes, root, (void*)ctx->stream))
        ABC_FAIL(ctx, ABC_ERR_COMM, "broadcast callback failed");
    return ABC_OK;
}

int xbuf_reserve(abc_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->xbuf_bytes) return ABC_OK;
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->xbuf) { ABC_HIP(ctx, hipFree(ctx->xbuf)); ctx->xbuf = nullptr; ctx->xbuf_bytes = 0; }
    bytes = abc_align(bytes + bytes / 8, 1 << 20);          // some head room: the longest winner list varies from set to set
    ABC_HIP(ctx, hipMalloc((void**)&ctx->xbuf, bytes));
    ctx->xbuf_bytes = bytes;
    return ABC_OK;
}

// ---- small kernels of the exchange steps ------------------------------------------------------------------------------------
// winner lists shorter than the longest one are padded with sentinels that sort last
__global__ __launch_bounds__(256) void k_pad_tail(double* __restrict__ key, unsigned long long* __restrict__ idx, size_t from, size_t to) {
    const size_t i = from + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < to) { key[i] = INFINITY; idx[i] = 1ull << 62; }
}
// theta[k, p] = rows[q][j, p] with src[k] = q * maxw + j: rows holds, per rank q, a column-major maxw x P block of that rank's
// winners in its run order
__global__ __launch_bounds__(256) void k_place_rows(const double* __restrict__ rows, const unsigned long long* __restrict__ src,
                                                    size_t K, int P, size_t maxw, double* __restrict__ theta) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= K * (size_t)P) return;
    const size_t k = e % K, p = e / K;
    const unsigned long long s = src[k];
    const size_t q = (size_t)(s / maxw), j = (size_t)(s % maxw);
    theta[k + K * p] = rows[(q * (size_t)P + p) * maxw + j];
}
// w[k0_q + i] = slices[q * kmax + i] for i < kn_q  (K = base * W + rem rows split as evenly as possible, low ranks first)
__global__ __launch_bounds__(256) void k_unpad_slices(const double* __restrict__ slices, size_t K, int W, size_t kmax,
                                                      double* __restrict__ w) {
    const size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const size_t base = K / (size_t)W, rem = K % (size_t)W;
    // rows [0, rem * (base + 1)) belong to the first `rem` ranks (base + 1 rows each), the rest to ranks of `base` rows
    size_t q, i;
    if (k < rem * (base + 1)) { q = k / (base + 1); i = k % (base + 1); }
    else { const size_t t = k - rem * (base + 1); q = rem + (base ? t / base : 0); i = base ? t % base : 0; }
    w[k] = slices[q * kmax + i];
}

// validation rows of the Wilcoxon rule: rows [v0, v0 + nv) of every column of a shard, packed column-major with leading
// dimension vmax (zero padded) for the all-gather ...
__global__ __launch_bounds__(256) void k_pack_valid(const double* __restrict__ src, size_t ld, size_t v0, size_t nv, size_t vmax,
                                                    int C, double* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= vmax * (size_t)C) return;
    const size_t j = e % vmax, c = e / vmax;
    dst[e] = (j < nv) ? src[v0 + j + ld * c] : 0.0;
}
// ... and, from the gathered blocks [rank q][column c][vmax], the validation rows of the whole set in global row order
// (rank q's rows start at voff[q]): dst is Nv x C, column-major
__global__ __launch_bounds__(256) void k_place_valid(const double* __restrict__ all, const long long* __restrict__ voff /* W + 1 */,
                                                     int W, size_t vmax, int C, size_t Nv, double* __restrict__ dst) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= Nv * (size_t)C) return;
    const size_t i = e % Nv, c = e / Nv;
    int q = 0;
    while (q + 1 < W && (long long)i >= voff[q + 1]) q++;
    dst[e] = all[((size_t)q * C + c) * vmax + (i - (size_t)voff[q])];
}

// candidate lists of all ranks, as gathered: rank q's record = [header 256 B][dist cap][global row cap][rows cap x P column-major]
struct CandRec {
    size_t cap, P, rec_bytes;
    __host__ __device__ const unsigned long long* hdr(const char* all, int q) const { return (const unsigned long long*)(all + (size_t)q * rec_bytes); }
    __host__ __device__ const double* dist(const char* all, int q) const { return (const double*)(all + (size_t)q * rec_bytes + 256); }
    __host__ __device__ const unsigned long long* idx(const char* all, int q) const { return (const unsigned long long*)(all + (size_t)q * rec_bytes + 256 + cap * 8); }
    __host__ __device__ const double* rows(const char* all, int q) const { return (const double*)(all + (size_t)q * rec_bytes + 256 + cap * 16); }
};
// The ranks' statistics records -- each about its own pilot shift t_r: counts n_r, sums s_r = sum (x - t_r), products
// G_r = sum (x - t_r)(x - t_r)' per partition -- as ONE record about rank 0's shift: with d_r = t_r - t_0,
//   sum (x - t_0) = s_r + n_r d_r,   sum (x - t_0)(x - t_0)' = G_r + d_r s_r' + s_r d_r' + n_r d_r d_r',
// added up in rank order (the same bits on every rank).  The shifts are all near the column means, so the corrections are small
// against G_r: nothing of the one-pass statistics' accuracy is lost.  Entries no rank has computed (the pure Y'Y blocks off their
// diagonal, padding) stay zero.
// Which entries the ranks have computed follows from the BLOCK STRUCTURE (a 16-column block pair is skipped only when neither
// block holds a metric column: the pure Y'Y blocks keep their diagonal alone), not from the values: a column that is constant
// within every shard but differs between them has G_r = 0 and s_r = 0 on every rank, and its re-centring terms are all there is
// (the values are still looked at for the blocks the structure does not vouch for).
__global__ __launch_bounds__(256) void k_stats_merge(const double* __restrict__ all, int W, StatsLayout L, int M, double* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x, C = L.C16;
    if (e < 2) {                                             // the counts
        double n = 0.0;
        for (int r = 0; r < W; r++) n += all[(size_t)r * L.len + L.off_n + e];
        out[L.off_n + e] = n;
        return;
    }
    size_t q = e - 2;
    if (q < C) { out[L.off_shift + q] = all[L.off_shift + q]; return; }          // rank 0's shift
    q -= C;
    if (q < 2 * C) {                                         // the column sums of a partition
        const int p = (int)(q / C);
        const size_t i = q % C;
        double acc = 0.0;
        for (int r = 0; r < W; r++) {
            const double* rec = all + (size_t)r * L.len;
            const double d = rec[L.off_shift + i] - all[L.off_shift + i];
            acc += fma(rec[L.off_n + p], d, rec[L.off_sum[p] + i]);
        }
        out[L.off_sum[p] + i] = acc;
        return;
    }
    q -= 2 * C;
    if (q >= 2 * C * C) return;
    const int p = (int)(q / (C * C));
    const size_t ij = q % (C * C), i = ij % C, j = ij / C;
    double acc = 0.0;
    bool any = false;
    for (int r = 0; r < W; r++) {
        const double* rec = all + (size_t)r * L.len;
        const double g = rec[L.off_G[p] + ij];
        any = any || g != 0.0;
        const double di = rec[L.off_shift + i] - all[L.off_shift + i], dj = rec[L.off_shift + j] - all[L.off_shift + j];
        const double si = rec[L.off_sum[p] + i], sj = rec[L.off_sum[p] + j], nr = rec[L.off_n + p];
        acc += g + fma(di, sj, fma(si, dj, nr * di * dj));
    }
    const size_t CX = ((size_t)M + 15) / 16;                 // blocks with at least one metric column
    const bool structural = i == j || i / 16 < CX || j / 16 < CX;
    out[L.off_G[p] + ij] = (any || structural) ? acc : 0.0;
}

// header of a rank's list: [0] entries, [1] its selection gave up (placeholder entries), [2] the rank's rows
__global__ void k_ls_header(unsigned long long* __restrict__ hdr, unsigned long long count, unsigned long long n_local, const int* __restrict__ sel_fail) {
    if (threadIdx.x == 0) { hdr[0] = count; hdr[1] = (sel_fail && *sel_fail) ? 1ull : 0ull; hdr[2] = n_local; }
}
// the gathered lists as W contiguous sorted runs (distance, global row); thread 0: did any rank's selection give up?
__global__ __launch_bounds__(256) void k_ls_unpack(const char* __restrict__ all, CandRec R, int W, double* __restrict__ cand_dist,
                                                   unsigned long long* __restrict__ cand_idx, int* __restrict__ fail, int* __restrict__ fail_pin) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e == 0) {
        int bad = 0;
        for (int q = 0; q < W; q++) {
            // a full list, or -- a shard with fewer rows than the list is long -- ALL of the rank's rows (exhaustive: k_ls_check)
            const unsigned long long cnt = R.hdr(all, q)[0], nloc = R.hdr(all, q)[2];
            const bool whole = cnt == (unsigned long long)R.cap || (nloc <= (unsigned long long)R.cap && cnt == nloc);
            if (R.hdr(all, q)[1] != 0ull || !whole) bad = 1;
        }
        *fail = bad;
        if (fail_pin) *fail_pin = bad;
    }
    if (e >= (size_t)W * R.cap) return;
    const int q = (int)(e / R.cap);
    const size_t j = e % R.cap;
    cand_dist[e] = R.dist(all, q)[j];
    cand_idx[e] = R.idx(all, q)[j];
}
// the rule of the local-top protocol: the first K of the merge are the K smallest of the whole set unless some rank may hold an
// unlisted key at or below the K-th -- i.e. unless its list is not exhaustive and its last key does not lie ABOVE the K-th (an
// unlisted key EQUAL to it could precede a chosen one of a higher rank in global row order).  NaN compares false: fails, safely.
__global__ void k_ls_check(const char* __restrict__ all, CandRec R, int W, const double* __restrict__ merged, size_t K, int* __restrict__ fail,
                           int* __restrict__ fail_pin) {
    const double kth = merged[K - 1];
    int bad = 0;
    for (int q = threadIdx.x; q < W; q += blockDim.x) {
        const bool exhaustive = R.hdr(all, q)[2] <= (unsigned long long)R.cap;
        if (!exhaustive && !(R.dist(all, q)[R.cap - 1] > kth)) bad = 1;       // (an exhaustive list: padded with +inf behind its rows)
    }
    if (__any(bad) && threadIdx.x == 0) { *fail = 1; if (fail_pin) *fail_pin = 1; }
}
// the K winners in ascending (distance, global row) order = the first K of the sorted candidates: their rows and parameters
__global__ __launch_bounds__(256) void k_ds_place(const char* __restrict__ all, CandRec R, const double* __restrict__ sdist,
                                                  const unsigned long long* __restrict__ spos, size_t K, const int* __restrict__ fail,
                                                  unsigned long long* __restrict__ idx, double* __restrict__ dist, double* __restrict__ theta) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= K * (R.P + 1)) return;
    const size_t k = e % K, p = e / K;          // p == P: the index / distance pair
    if (*fail) {                                // harmless, in-range placeholder; the generation repeats with the radix protocol
        if (p == R.P) { idx[k] = k; if (dist) dist[k] = 0.0; } else theta[k + K * p] = 0.0;
        return;
    }
    const unsigned long long pos = spos[k];
    const int q = (int)(pos / R.cap);
    const size_t j = (size_t)(pos % R.cap);
    if (p == R.P) { idx[k] = R.idx(all, q)[j]; if (dist) dist[k] = sdist[k]; }
    else theta[k + K * p] = R.rows(all, q)[j + R.cap * p];
}

size_t default_A(size_t M, size_t P, int max_comp) { return (max_comp > 0) ? (size_t)max_comp : (M < P ? M : P); }

}  // namespace

int abc_comm_all_reduce(abc_ctx* ctx, void* buf, size_t count, int dtype) { return comm_all_reduce(ctx, buf, count, dtype); }
int abc_comm_all_gather(abc_ctx* ctx, const void* send, void* recv, size_t bytes) { return comm_all_gather(ctx, send, recv, bytes); }
int abc_xbuf_reserve(abc_ctx* ctx, size_t bytes) { return xbuf_reserve(ctx, bytes); }

// ---- communicators ------------------------------------------------------------------------------------------------------------
extern "C" int abc_comm_unique_id(void* id128) {
    if (!id128) return ABC_ERR_INVALID;
    if (!rccl()->ok) return ABC_ERR_COMM;
    ncclUniqueId id;
    if (rccl()->GetUniqueId(&id) != ncclSuccess) return ABC_ERR_COMM;
    memcpy(id128, &id, sizeof(id));
    return ABC_OK;
}

void abc_comm_release(abc_ctx* ctx) {
    if (ctx->comm_kind == 1 && ctx->comm_nccl && rccl()->ok) (void)rccl()->CommDestroy((ncclComm_t)ctx->comm_nccl);
    ctx->comm_nccl = nullptr;
    ctx->comm_kind = 0;
    ctx->comm_world = 1;
    ctx->comm_rank = 0;
}

extern "C" int abc_comm_destroy(abc_ctx* ctx) {
    if (!ctx) return ABC_ERR_INVALID;
    if (hipSetDevice(ctx->device) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_HIP, "hipSetDevice failed");
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    abc_comm_release(ctx);
    return ABC_OK;
}

extern "C" int abc_comm_init_rank(abc_ctx* ctx, int world, int rank, const void* id128) {
    if (!ctx) return ABC_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world || !id128) ABC_FAIL(ctx, ABC_ERR_INVALID, "comm: world %d, rank %d", world, rank);
    if (!rccl()->ok) ABC_FAIL(ctx, ABC_ERR_COMM, "librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "missing symbols");
    if (hipSetDevice(ctx->device) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_HIP, "hipSetDevice failed");
    abc_comm_release(ctx);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ABC_NCCL(ctx, rccl()->CommInitRank(&c, world, id, rank));
    ctx->comm_nccl = c;
    ctx->comm_kind = 1;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return ABC_OK;
}

extern "C" int abc_comm_init_callbacks(abc_ctx* ctx, int world, int rank, const abc_comm_callbacks* cb) {
    if (!ctx) return ABC_ERR_INVALID;
    if (world < 1 || rank < 0 || rank >= world || !cb || !cb->all_reduce_sum || !cb->all_gather || !cb->broadcast)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "comm: world %d, rank %d, or a null callback", world, rank);
    abc_comm_release(ctx);
    ctx->comm_cb = *cb;
    ctx->comm_kind = 2;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return ABC_OK;
}

extern "C" int abc_comm_info(const abc_ctx* ctx, int* kind, int* world, int* rank) {
    if (!ctx) return ABC_ERR_INVALID;
    if (kind) *kind = ctx->comm_kind;
    if (world) *world = ctx->comm_kind ? ctx->comm_world : 1;
    if (rank) *rank = ctx->comm_kind ? ctx->comm_rank : 0;
    return ABC_OK;
}

extern "C" int abc_ctx_create_multi(const int* devices, int ndev, abc_ctx** out) {
    if (!devices || !out || ndev < 1) return ABC_ERR_INVALID;
    for (int i = 0; i < ndev; i++) out[i] = nullptr;
    if (!rccl()->ok) return ABC_ERR_COMM;
    for (int i = 0; i < ndev; i++) {
        const int rc = abc_ctx_create(devices[i], &out[i]);
        if (rc != ABC_OK) {
            for (int j = 0; j < i; j++) { abc_ctx_destroy(out[j]); out[j] = nullptr; }
            return rc;
        }
    }
    std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
    if (rccl()->CommInitAll(comms.data(), ndev, devices) != ncclSuccess) {
        for (int j = 0; j < ndev; j++) { abc_ctx_destroy(out[j]); out[j] = nullptr; }
        return ABC_ERR_COMM;
    }
    for (int i = 0; i < ndev; i++) {
        out[i]->comm_nccl = comms[(size_t)i];
        out[i]->comm_kind = 1;
        out[i]->comm_world = ndev;
        out[i]->comm_rank = i;
    }
    return ABC_OK;
}

// ---- one generation, rows sharded over the communicator ------------------------------------------------------------------------
static int sharded_core(abc_ctx* ctx, const abc_sharded_cfg* cfg, const abc_generation_io* io, abc_rng* rng, int32_t* ncomp_host);

// Argument checks are local and deterministic (the same call on every rank passes or fails them alike) and happen before
// the first collective.  Past them a failure on ONE rank only -- arena or exchange buffer exhausted, a HIP error, a failed
// collective -- would leave its peers waiting inside RCCL for ever: the communicator is then ABORTED (ncclCommAbort), which
// makes the peers' pending collectives return an error instead of hanging; the context is left without a communicator and
// says so.  A covariance that is not positive definite is computed from replicated data, i.e. reported by every rank: no abort.
extern "C" int abc_generation_sharded_dev(abc_ctx* ctx, const abc_sharded_cfg* cfg, const abc_generation_io* io, abc_rng* rng,
                                          int32_t* ncomp_host) {
    if (!ctx) return ABC_ERR_INVALID;
    ctx->err[0] = 0;
    if (hipSetDevice(ctx->device) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_HIP, "hipSetDevice failed");
    if (ctx->timing && ctx->nev > 96) ABC_TRY(abc_timing_flush(ctx));
    if (!cfg || !io || !io->X || !io->Y || !io->obs || !io->idx || !io->w || !rng)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "sharded generation: null argument");
    {
        const size_t n = cfg->n_local, N = cfg->N_total, M = cfg->M, P = cfg->P, K = cfg->K, Nn = cfg->nnext_local;
        const uint64_t row0 = cfg->row0;
        if (!N || !M || !P || K == 0 || K > N || row0 + n > N || cfg->next0 + Nn > cfg->Nnext_total)
            ABC_FAIL(ctx, ABC_ERR_INVALID, "sharded generation: bad sizes N=%zu n_local=%zu row0=%zu M=%zu P=%zu K=%zu", N, n, (size_t)row0, M, P, K);
        if (Nn && !io->next) ABC_FAIL(ctx, ABC_ERR_INVALID, "sharded generation: null proposal buffer");
    }
    if (!(0.0 < cfg->train_frac && cfg->train_frac <= 1.0))
        ABC_FAIL(ctx, ABC_ERR_INVALID, "training fraction %g outside (0,1]", cfg->train_frac);
    if (cfg->rule != ABC_RULE_MIN_PRESS && cfg->rule != ABC_RULE_WILCOXON)
        ABC_FAIL(ctx, ABC_ERR_INVALID, "unknown component rule %d", cfg->rule);
    if (ctx->noise_mode == ABC_NOISE_REFERENCE_STREAM)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "the reference noise stream is sequential over the whole set: not available to the sharded generation");
    const int rc = sharded_core(ctx, cfg, io, rng, ncomp_host);
    if (rc != ABC_OK && rc != ABC_ERR_NOT_SPD && ctx->comm_kind == 1 && ctx->comm_world > 1 && ctx->comm_nccl) {
        if (rccl()->CommAbort) (void)rccl()->CommAbort((ncclComm_t)ctx->comm_nccl);
        ctx->comm_nccl = nullptr;
        ctx->comm_kind = 0;
        ctx->comm_world = 1;
        ctx->comm_rank = 0;
        const size_t len = strlen(ctx->err);
        snprintf(ctx->err + len, sizeof(ctx->err) - len, " [rank-local failure: the RCCL communicator was aborted so that the other ranks do not hang; create a new one]");
    }
    return rc;
}

static int sharded_core(abc_ctx* ctx, const abc_sharded_cfg* cfg, const abc_generation_io* io, abc_rng* rng, int32_t* ncomp_host) {
    const int W = ctx->comm_kind ? ctx->comm_world : 1, r = ctx->comm_kind ? ctx->comm_rank : 0;
    const size_t n = cfg->n_local, N = cfg->N_total, M = cfg->M, P = cfg->P, K = cfg->K, Kp = cfg->Kp, Nn = cfg->nnext_local;
    const uint64_t row0 = cfg->row0;
    const size_t A = default_A(M, P, cfg->max_comp);
    const size_t kloc = K < n ? K : n;                       // most winners this rank can hold
    const size_t kbase = K / (size_t)W, krem = K % (size_t)W, kmax = kbase + (krem ? 1 : 0);
    const size_t k0 = (size_t)r * kbase + ((size_t)r < krem ? (size_t)r : krem), kn = kbase + ((size_t)r < krem ? 1 : 0);

    // Selection over the ranks (round 4, second half): every rank takes the ls_cap smallest of ITS OWN distances in ascending (distance,
    // row) order -- the single-GPU bin selection on the local shard --, ONE all-gather carries these lists with their parameter rows,
    // and every rank merges the W sorted runs: the first K of the merge are the K smallest of the whole set in the single-GPU order,
    // PROVIDED no rank holds an unlisted key below the K-th (every list's last key lies above it).  Rows are dealt to ranks without
    // regard to their distance, so a rank's share of the K smallest is Binomial(K, 1 / W): ls_cap = K / W + 8 sigma + 64 fails
    // once in ~1e15 sets on continuous data; massively tied distances (or a rank's bin selection giving up) fail the rule, every rank
    // sees that in the gathered lists alike, and the generation repeats itself with the radix protocol (six all-reduced histograms),
    // which also takes small sets (fewer than 4096 rows per rank) and K > N / 2.  (The first half of the round and round 3 gathered
    // a SAMPLE of the distances first, for a common bound: one collective and six launches more, and the candidates unsorted.)
    size_t ls_cap = 0;
    bool local_sel = W > 1 && !ctx->sel_force_radix && 2 * K <= N && N / (size_t)W >= 4096;
    if (local_sel) {
        const double mean = (double)K / (double)W, sd = sqrt(mean * (1.0 - 1.0 / (double)W));
        ls_cap = (size_t)(mean + 8.0 * sd + 64.0);
        ls_cap = (ls_cap + 31) / 32 * 32;
        if (ls_cap > N / (size_t)W) local_sel = false;            // (the smallest shard could not fill its list)
    }
    const size_t ds_cap = ls_cap;
    const abc_rng rng_entry = *rng;
    size_t need = abc_ws_need(n, M, P, A, K, Kp, Nn) + (size_t)W * kmax * 8 + 4 * kloc * 8 + (1u << 20) + 8 * (size_t)W * ds_cap * 8 +
                  (size_t)(W + 1) * stats_layout(M, P).len * 8;
    const uint64_t ntrain = (uint64_t)llround((double)N * cfg->train_frac);                  // AbcUtil.cpp:438, global rows
    if (cfg->rule == ABC_RULE_WILCOXON) {
        // the validation rows of the whole set are assembled on every rank (all-gather of the shards' validation rows)
        const size_t NvT = N > ntrain ? N - (size_t)ntrain : 0;
        need += 2 * abc_wx_need(NvT, P, A) + (2u << 20);  // (the cascade over the shards and, should its exact step give up, the
                                                          // reduction once more on the gathered rows; those live in the exchange buffer)
        need += n * A * 8 + 4096;                         // (the scores of all of this rank's rows: the projection's, read by the cascade)
    }
    ABC_TRY(abc_ws_reserve(ctx, need));
    const StatsLayout SL = stats_layout(M, P);
    const ModelLayout ML = model_layout(M, P, A);
    double* stats = (double*)abc_ws_alloc(ctx, SL.len * 8);
    double* model = (double*)abc_ws_alloc(ctx, ML.len * 8);
    double* dist = (double*)abc_ws_alloc(ctx, (n ? n : 1) * 8);
    int* spd_dev = (int*)abc_ws_alloc(ctx, sizeof(int));
    long long* sel_state = (long long*)abc_ws_alloc(ctx, 8 * sizeof(long long));
    int* sel_hist = (int*)abc_ws_alloc(ctx, 2048 * sizeof(int));
    long long* counts = (long long*)abc_ws_alloc(ctx, (size_t)(2 + 2 * W) * sizeof(long long));
    double* w_mine = (double*)abc_ws_alloc(ctx, (kmax ? kmax : 1) * 8);
    double* w_slices = (double*)abc_ws_alloc(ctx, (size_t)W * (kmax ? kmax : 1) * 8);
    if (!stats || !model || !dist || !spd_dev || !sel_state || !sel_hist || !counts || !w_mine || !w_slices)
        ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
    ABC_TRY(abc_pin_reserve(ctx, (size_t)(2 * W) * sizeof(long long) + 64));

    // ---- 1-2: sufficient statistics, replicated model fit -----------------------------------------------------------------
    // (the side stream is forked at the call's start, as in the single-GPU driver: a record between two kernels of the main
    // stream costs its critical path 6-7 us, the first packet of an idle queue nothing)
    ctx->side_early_waited = false;
    ctx->side_forked = false;
    *(volatile unsigned*)(ctx->status_pin + 56) = 0u;          // (raised by a proposal kernel that gives up: read at the call's end)
    // (... except beside the byte-limb statistics kernel of wide sets, which wants every CU to itself: forked behind it, api.hip)
    const bool fork_late = abc_gram_takes_i8(ctx, io->X, io->Y, n, n, n, M, P, ntrain, N);
    if (!fork_late) ABC_TRY(abc_side_fork(ctx));
    // Every rank takes its statistics about ITS OWN pilot shift (round 4: no broadcast of rank 0's in front of the pass over the
    // rows); the records are all-gathered and every rank re-centres them on rank 0's shift while it adds them up (k_stats_merge).
    ABC_TRY(launch_stats_shift(ctx, io->X, io->Y, n, n, n, M, P, stats));
    ABC_TRY(launch_stats_accumulate(ctx, io->X, io->Y, n, n, n, M, P, row0, ntrain, stats, N));
    if (fork_late) ABC_TRY(abc_side_fork(ctx));
    // the taus2 streams of this rank's proposals (draws, seeds) need the rng state only: on the side stream
    uint32_t* raw_early = nullptr;
    if (Nn) ABC_TRY(abc_rng_streams_early(ctx, rng, cfg->next0, Nn, io->seeds, cfg->Nnext_total, &raw_early));
    // ... and so does the previous set's share of the weight stage
    abc_wprev wprev;
    memset(&wprev, 0, sizeof(wprev));
    if (Kp && io->theta_prev && kn)
        ABC_TRY(abc_weights_prev_early(ctx, P, W == 1 ? K : kn, io->theta_prev, Kp, io->w_prev, io->dv_prev, &wprev));
    ctx->side_forked = false;
    double* stats_all = nullptr;           // the ranks' records as gathered (k_stats_merge leaves them as they came)
    if (W > 1) {
        stats_all = (double*)abc_ws_alloc(ctx, (size_t)W * SL.len * 8);
        if (!stats_all) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
        ABC_TRY(comm_all_gather(ctx, stats, stats_all, SL.len * 8));
        const size_t ne = 2 * SL.C16 * SL.C16 + 3 * SL.C16 + 2;
        hipLaunchKernelGGL(k_stats_merge, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)stats_all, W, SL, (int)M, stats);
        ABC_HIP(ctx, hipGetLastError());
    }
    ABC_TRY(launch_pls_model(ctx, stats, io->obs, M, P, A, cfg->rule, model));
    // The rank sums rank the paired differences of ALL validation rows (global rows >= ntrain, AbcUtil.cpp:438-446) together.
    // Round 5: nothing of the rows travels.  The reduction is a cascade of bounds on the rank sums from per-bin counts, and counts
    // are additive over rows: every rank sweeps ITS validation rows, the counts of a level are all-reduced (192 cells x 8 bytes a
    // test, then the fine bins of the tests still undecided), bounds and verdicts are computed from the same numbers on every
    // rank -- the (replicated) model stays identical everywhere without a broadcast; only the keys of the tests the bounds leave
    // undecided (8 bytes a validation row each) are all-gathered for their exact rank sums (wilcoxon.hip).
    // Round 6: ONE pass over this rank's rows for the ranking's projection AND the cascade's scores, as in the single-GPU driver -- the
    // projection runs first, on the count the fit wrote, and leaves the scores of all its rows; the cascade (the largest count first:
    // the caller only uses that one) reads the validation rows' scores from there, and should it lower the count, the distances are
    // taken again from the scores (k_dist_from_scores), not from X.  All on the main stream: the cascade's all-reduces and the
    // selection's all-gathers share one communicator, whose collectives every rank must issue in one order.
    int wx_rc = ABC_INTERNAL_RETRY;
    bool projected = false;
    const size_t wx_v0 = (ntrain > row0) ? (size_t)((ntrain - row0) < n ? (ntrain - row0) : n) : 0;
    if (cfg->rule == ABC_RULE_WILCOXON) {
        const size_t NvT = N > ntrain ? N - (size_t)ntrain : 0;
        if (NvT && abc_wx_cascade_applies(NvT, P, A) && !ctx->wx_gather_rows) {
            const abc_wx_shard sh = {NvT, W > 1 ? stats_all + SL.off_n + 1 : nullptr, SL.len};
            double* wx_dec = (double*)abc_ws_alloc(ctx, (P + 1) * 8);
            if (!wx_dec) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
            struct ScoresArg { abc_ctx* ctx; const double* X; size_t n, M, P, A, v0; const double* model; double* dist; double* S_all; }
                sarg = {ctx, io->X, n, M, P, A, wx_v0, model, dist, nullptr};
            if (!(n & 1)) sarg.S_all = (double*)abc_ws_alloc(ctx, n * A * 8);
            abc_wx_scores_hook hook = {
                [](void* a, double** S, size_t* sld) -> int {
                    ScoresArg* q = (ScoresArg*)a;
                    if (!q->S_all) return 1;
                    const int rc = launch_project_distance_scores(q->ctx, q->X, q->n, q->n, q->M, q->P, q->A, q->model, q->dist, q->S_all, q->n, 0, nullptr);
                    if (rc == 0) { *S = q->S_all + q->v0; *sld = q->n; q->dist = nullptr; }
                    else if (rc == 1) q->S_all = nullptr;
                    return rc;
                },
                &sarg};
            abc_wx_run* run = nullptr;
            wx_rc = launch_wilcoxon_begin(ctx, io->X, io->Y, n, n, n, M, P, A, wx_v0, model, wx_dec, /*stop_at_max=*/1, &run, &hook, 0, W > 1 ? &sh : nullptr);
            int changed = 2;
            if (wx_rc == ABC_OK) wx_rc = launch_wilcoxon_finish(ctx, run, &changed);
            if (wx_rc != ABC_OK && wx_rc != ABC_INTERNAL_RETRY) return wx_rc;
            if (wx_rc == ABC_OK) {
                ABC_TRY(launch_wilcoxon_commit(ctx, model, M, P, A, wx_dec, changed ? 1 : 0));
                projected = sarg.dist == nullptr;
                if (projected && changed) ABC_TRY(launch_distance_from_scores(ctx, sarg.S_all, n, n, M, P, A, model, dist));
            }
        }
        if (wx_rc == ABC_INTERNAL_RETRY && W == 1) {       // small sets, more than 32 components, the cascade's own give-up: in stream order
            ABC_TRY(launch_wilcoxon(ctx, io->X, io->Y, n, n, n, M, P, A, (size_t)ntrain, model));
            wx_rc = ABC_OK;
        }
    }
    if (cfg->rule == ABC_RULE_WILCOXON && W > 1 && wx_rc == ABC_INTERNAL_RETRY) {
        // Small sets (fewer than 16384 validation rows), more than 32 components, or a bin of the cascade's exact step that
        // outgrew LDS (massive ties; every rank sees it alike): every rank gathers the shards' validation rows, rebuilds the
        // validation block in global row order and runs the single-device reduction on it -- the same code on the same
        // numbers on every rank.
        const size_t v0 = wx_v0, nv = n - v0;
        long long* vcnt = (long long*)abc_ws_alloc(ctx, (size_t)(2 * W + 2) * sizeof(long long));
        if (!vcnt) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
        const long long mine = (long long)nv;
        ABC_HIP(ctx, hipMemcpyAsync(vcnt, &mine, sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        ABC_TRY(comm_all_gather(ctx, vcnt, vcnt + 1, sizeof(long long)));
        long long* hv = (long long*)ctx->pin;
        ABC_HIP(ctx, hipMemcpyAsync(hv, vcnt + 1, (size_t)W * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<long long> off((size_t)W + 1, 0);
        long long vmax = 0;
        for (int q = 0; q < W; q++) { off[(size_t)q + 1] = off[(size_t)q] + hv[q]; if (hv[q] > vmax) vmax = hv[q]; }
        const size_t Nv = (size_t)off[(size_t)W], vm = (size_t)vmax;
        if (Nv) {
            long long* voff = vcnt + 1 + W;
            // sized from the GATHERED counts (vmax over all ranks, Nv): the same bytes on every rank whatever its own shard
            // holds -- the arena is sized from the local row count and ran out on the smaller ranks of uneven shards only,
            // which then left their peers waiting in the all-gather below
            const size_t cb = (M + P) * 8;
            ABC_TRY(xbuf_reserve(ctx, ((size_t)(W + 1) * vm + Nv) * cb + 4 * 256));
            char* xq = ctx->xbuf;
            auto takeq = [&](size_t bytes) { char* p0 = xq; xq += abc_align(bytes, 256); return (double*)p0; };
            double* sendb = takeq(vm * cb);
            double* recvb = takeq((size_t)W * vm * cb);
            double* Xv = takeq(Nv * M * 8);
            double* Yv = takeq(Nv * P * 8);
            ABC_HIP(ctx, hipMemcpyAsync(voff, off.data(), (size_t)(W + 1) * sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
            // X and Y travel in one block: [M + P columns][vmax rows] per rank
            hipLaunchKernelGGL(k_pack_valid, dim3((unsigned)((vm * M + 255) / 256)), dim3(256), 0, ctx->stream, io->X, n, v0, nv, vm,
                               (int)M, sendb);
            hipLaunchKernelGGL(k_pack_valid, dim3((unsigned)((vm * P + 255) / 256)), dim3(256), 0, ctx->stream, io->Y, n, v0, nv, vm,
                               (int)P, sendb + vm * M);
            ABC_TRY(comm_all_gather(ctx, sendb, recvb, vm * (M + P) * 8));
            const int C = (int)(M + P);
            // place: columns 0..M-1 -> Xv, M..M+P-1 -> Yv (two launches over the same gathered blocks, column offset by pointer)
            hipLaunchKernelGGL(k_place_valid, dim3((unsigned)((Nv * M + 255) / 256)), dim3(256), 0, ctx->stream, recvb, voff, W, vm, C, Nv, Xv);
            ABC_HIP(ctx, hipGetLastError());
            // Y columns: the block of rank q starts M columns into its record
            hipLaunchKernelGGL(k_place_valid, dim3((unsigned)((Nv * P + 255) / 256)), dim3(256), 0, ctx->stream, recvb + vm * M, voff, W, vm, C, Nv, Yv);
            ABC_HIP(ctx, hipGetLastError());
            ABC_TRY(launch_wilcoxon(ctx, Xv, Yv, Nv, Nv, Nv, M, P, A, 0, model));
        }
    }
    if (!projected) ABC_TRY(launch_project_distance(ctx, io->X, n, n, M, P, A, model, 0, dist));
    // first set: uniform weights, their alias table is built by the host while the GPU ranks
    const bool uniform_w = (Kp == 0 || !io->theta_prev);
    if (uniform_w && Nn) ABC_TRY(abc_uniform_alias(ctx, K));

    // ---- 3-5: the K smallest distances of the whole set, their rows ---------------------------------------------------------
    double* theta = io->theta ? io->theta : (double*)abc_ws_alloc(ctx, K * P * 8);
    if (!theta) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
    // The posterior's moments go to the side stream beside the pair sums (below); the event that hands theta over is the completion
    // signal of the kernel that writes it, not a record behind it (a record between two kernels of the main stream cost ~35 us of
    // this driver's critical path in the world-1 timeline: scripts/trace_sharded.py)
    const bool moments_planned = Nn && !uniform_w && P <= 64 && K >= 2 && ctx->side;
    if (moments_planned && !ctx->ev_theta) {
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_theta, abc_xstream_event_flags()));
        ABC_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_moments, abc_xstream_event_flags()));
    }
    bool theta_ev_bound = false;
    int* pfail_early = nullptr;          // pinned: the selection's verdict, seen by the host at its wait for the weights / at the end
    int* ds_fail_dev = nullptr;
    if (W == 1) {
        // as abc_generation_dev: a bin selection that gave up (degenerate distances) is noticed at the generation's next host visit
        // and the generation repeats itself with the radix select -- no synchronisation in the middle of the ranking
        ABC_TRY(launch_select_smallest(ctx, dist, n, K, row0, io->idx, io->dist, true));
        const bool bins_deferred = ctx->sel_bins_ran && ctx->sel_fail_dev && !ctx->sel_force_radix;
        ctx->sel_bins_ran = false;
        pfail_early = (int*)(ctx->status_pin + 40);
        *pfail_early = 0;
        if (bins_deferred) ds_fail_dev = ctx->sel_fail_dev;
        ABC_TRY(launch_gather_rows(ctx, io->Y, n, n, P, io->idx, K, row0, theta, K, bins_deferred ? ctx->sel_fail_dev : nullptr, pfail_early,
                                   moments_planned ? ctx->ev_theta : nullptr));
        theta_ev_bound = moments_planned;
    } else if (local_sel) {
        CandRec R;
        R.cap = ls_cap; R.P = P; R.rec_bytes = abc_align(256 + ls_cap * (16 + 8 * P), 256);
        const size_t tot = (size_t)W * ls_cap;
        ABC_TRY(xbuf_reserve(ctx, (size_t)(W + 1) * R.rec_bytes + tot * 40 + 4096));
        char* rec_mine = ctx->xbuf;
        char* rec_all = rec_mine + R.rec_bytes;
        double* cand_dist = (double*)(rec_all + (size_t)W * R.rec_bytes);
        uint64_t* cand_idx = (uint64_t*)(cand_dist + tot);
        double* mrg_dist = (double*)(cand_idx + tot);
        uint64_t* mrg_idx = (uint64_t*)(mrg_dist + tot);
        uint64_t* mrg_src = mrg_idx + tot;
        int* ds_fail = (int*)abc_ws_alloc(ctx, sizeof(int));
        if (!ds_fail) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
        // this rank's ls_cap smallest, ascending (distance, global row), straight into its record; a bin selection that gives up
        // leaves a placeholder and says so in the record's header
        // (uneven shards: a rank with fewer rows than the list is long lists ALL of them, sorted, +inf behind -- its list is then
        // exhaustive and the rule of k_ls_check holds for it whatever the K-th is; ADVICE round 4: such a rank used to fail the
        // selection's K <= n check and took the communicator down with it)
        const size_t ls_n = ls_cap < n ? ls_cap : n;
        if (ls_n) ABC_TRY(launch_select_smallest(ctx, dist, n, ls_n, row0, (uint64_t*)R.idx(rec_mine, 0), (double*)R.dist(rec_mine, 0), true));
        const bool bins_deferred = ls_n && ctx->sel_bins_ran && ctx->sel_fail_dev && !ctx->sel_force_radix;
        ctx->sel_bins_ran = false;
        if (ls_n < ls_cap)
            hipLaunchKernelGGL(k_pad_tail, dim3((unsigned)((ls_cap - ls_n + 255) / 256)), dim3(256), 0, ctx->stream, (double*)R.dist(rec_mine, 0),
                               (unsigned long long*)R.idx(rec_mine, 0), ls_n, ls_cap);
        hipLaunchKernelGGL(k_ls_header, dim3(1), dim3(64), 0, ctx->stream, (unsigned long long*)R.hdr(rec_mine, 0), (unsigned long long)ls_n,
                           (unsigned long long)n, bins_deferred ? (const int*)ctx->sel_fail_dev : (const int*)nullptr);
        if (ls_n) ABC_TRY(launch_gather_rows(ctx, io->Y, n, n, P, (const uint64_t*)R.idx(rec_mine, 0), ls_n, row0, (double*)R.rows(rec_mine, 0), ls_cap));
        ABC_TRY(comm_all_gather(ctx, rec_mine, rec_all, R.rec_bytes));
        pfail_early = (int*)(ctx->status_pin + 40);
        *pfail_early = 0;
        ds_fail_dev = ds_fail;
        hipLaunchKernelGGL(k_ls_unpack, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, (const char*)rec_all, R, W, cand_dist,
                           (unsigned long long*)cand_idx, ds_fail, pfail_early);
        ABC_HIP(ctx, hipGetLastError());
        // W sorted runs -> one sequence; equal distances: lower rank (= lower global rows) first
        ABC_TRY(launch_merge_runs(ctx, cand_dist, cand_idx, W, ls_cap, mrg_dist, mrg_idx, mrg_src));
        hipLaunchKernelGGL(k_ls_check, dim3(1), dim3(64), 0, ctx->stream, (const char*)rec_all, R, W, (const double*)mrg_dist, K, ds_fail, pfail_early);
        if (moments_planned) {
            hipExtLaunchKernelGGL(k_ds_place, dim3((unsigned)((K * (P + 1) + 255) / 256)), dim3(256), 0, ctx->stream, nullptr, ctx->ev_theta, 0,
                                  (const char*)rec_all, R, (const double*)mrg_dist, (const unsigned long long*)mrg_src, K, (const int*)ds_fail,
                                  (unsigned long long*)io->idx, io->dist, theta);
            theta_ev_bound = true;
        } else
            hipLaunchKernelGGL(k_ds_place, dim3((unsigned)((K * (P + 1) + 255) / 256)), dim3(256), 0, ctx->stream, (const char*)rec_all, R,
                               (const double*)mrg_dist, (const unsigned long long*)mrg_src, K, (const int*)ds_fail,
                               (unsigned long long*)io->idx, io->dist, theta);
        ABC_HIP(ctx, hipGetLastError());
    } else {
        ABC_TRY(launch_select_begin(ctx, K, sel_state, sel_hist));
        for (int p = 0; p < 6; p++) {
            ABC_TRY(launch_select_hist(ctx, dist, n, sel_state, p, sel_hist));
            ABC_TRY(comm_all_reduce(ctx, sel_hist, 2048, ABC_DT_I32));
            ABC_TRY(launch_select_pick(ctx, sel_state, p, sel_hist, K));
        }
        ABC_TRY(launch_select_count(ctx, dist, n, sel_state, counts));
        ABC_TRY(comm_all_gather(ctx, counts, counts + 2, 2 * sizeof(long long)));
        long long* hc = (long long*)ctx->pin;
        ABC_HIP(ctx, hipMemcpyAsync(hc, counts + 2, (size_t)(2 * W) * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        // ties at the threshold go to the lowest global rows first = to the lowest ranks first (contiguous shards)
        long long remaining = (long long)K, maxw = 0, my_less = 0, my_take = 0;
        for (int q = 0; q < W; q++) remaining -= hc[2 * q];
        if (remaining < 0) ABC_FAIL(ctx, ABC_ERR_COMM, "distributed selection: %lld keys below the K-th", (long long)K - remaining);
        for (int q = 0; q < W; q++) {
            const long long tq = hc[2 * q + 1] < remaining ? hc[2 * q + 1] : remaining;
            remaining -= tq;
            const long long nwq = hc[2 * q] + tq;
            if (nwq > maxw) maxw = nwq;
            if (q == r) { my_less = hc[2 * q]; my_take = tq; }
        }
        if (remaining != 0) ABC_FAIL(ctx, ABC_ERR_COMM, "distributed selection: %lld winners missing", remaining);
        const size_t mw = (size_t)maxw, nw = (size_t)(my_less + my_take), tot = (size_t)W * mw;
        // exchange buffers: my run (idx, dist), all runs, merged runs + source map, packed rows of mine and of all
        const size_t xb = (2 * mw + 5 * tot) * 8 + (mw + tot) * P * 8 + 4096;
        ABC_TRY(xbuf_reserve(ctx, xb));
        char* xp = ctx->xbuf;
        auto take = [&](size_t bytes) { char* p0 = xp; xp += abc_align(bytes, 256); return p0; };
        uint64_t* loc_idx = (uint64_t*)take(mw * 8);
        double* loc_dist = (double*)take(mw * 8);
        uint64_t* cand_idx = (uint64_t*)take(tot * 8);
        double* cand_dist = (double*)take(tot * 8);
        uint64_t* mrg_idx = (uint64_t*)take(tot * 8);
        double* mrg_dist = (double*)take(tot * 8);
        uint64_t* mrg_src = (uint64_t*)take(tot * 8);
        double* rows_mine = (double*)take(mw * P * 8);
        double* rows_all = (double*)take(tot * P * 8);
        if ((size_t)(xp - ctx->xbuf) > ctx->xbuf_bytes) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: exchange buffer");
        ABC_TRY(launch_select_compact(ctx, dist, n, sel_state, (uint64_t)my_less, (uint64_t)my_take, row0, loc_idx, loc_dist));
        if (nw > 1) ABC_TRY(launch_sort_pairs(ctx, loc_dist, loc_idx, nw));       // stable: ties stay in row order
        if (nw < mw)
            hipLaunchKernelGGL(k_pad_tail, dim3((unsigned)((mw - nw + 255) / 256)), dim3(256), 0, ctx->stream, loc_dist,
                               (unsigned long long*)loc_idx, nw, mw);
        ABC_TRY(comm_all_gather(ctx, loc_idx, cand_idx, mw * 8));
        ABC_TRY(comm_all_gather(ctx, loc_dist, cand_dist, mw * 8));
        // G sorted runs -> one sequence; equal distances: lower rank (= lower global rows) first
        ABC_TRY(launch_merge_runs(ctx, cand_dist, cand_idx, W, mw, mrg_dist, mrg_idx, mrg_src));
        ABC_HIP(ctx, hipMemcpyAsync(io->idx, mrg_idx, K * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (io->dist) ABC_HIP(ctx, hipMemcpyAsync(io->dist, mrg_dist, K * 8, hipMemcpyDeviceToDevice, ctx->stream));
        // the winners' parameter rows: packed in run order, gathered from every rank, placed by the merge's source map
        if (nw) ABC_TRY(launch_gather_rows(ctx, io->Y, n, n, P, loc_idx, nw, row0, rows_mine, mw));
        ABC_TRY(comm_all_gather(ctx, rows_mine, rows_all, mw * P * 8));
        hipLaunchKernelGGL(k_place_rows, dim3((unsigned)((K * P + 255) / 256)), dim3(256), 0, ctx->stream, rows_all,
                           (const unsigned long long*)mrg_src, K, (int)P, mw, theta);
        ABC_HIP(ctx, hipGetLastError());
    }

    // the local-top selection failed its rule (every rank alike): once more, from the top, with the radix protocol
    auto repeat_with_radix = [&]() -> int {
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->side) ABC_HIP(ctx, hipStreamSynchronize(ctx->side));
        *rng = rng_entry;
        ctx->sel_force_radix = true;
        const int rc = sharded_core(ctx, cfg, io, rng, ncomp_host);
        ctx->sel_force_radix = false;
        return rc;
    };
    // ---- doubled variance, importance weights (pair sums: K / G rows per rank) ----------------------------------------------
    double* dv = io->dv ? io->dv : (double*)abc_ws_alloc(ctx, P * 8);
    double* theta_stats = nullptr;
    // As in the single-GPU driver: with proposals to draw behind a weight stage, the posterior's moments and everything that
    // follows from them (doubled variance, proposal factor, the perturbation's row-major copy and padded factor) run on the SIDE
    // stream beside the pair sums -- nothing waits for the host any more (the resampling table is built on the device).
    double* L_early = nullptr;
    abc_theta_fused side_out = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool moments_on_side = false, status_early = false;
    if (moments_planned) {
        if (cfg->multivariate) {
            L_early = io->L ? io->L : (double*)abc_ws_alloc(ctx, P * P * 8);
            if (!L_early) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
        }
        const int PPr = abc_perturb_pp(P);
        side_out.dv = dv; side_out.L = L_early; side_out.spd = spd_dev;
        // (as the fused driver: the generation's status words go straight into the pinned block from k_post_tail -- no copies behind
        // the proposals)
        side_out.model_hdr = model; side_out.hdr_pin = (double*)ctx->status_pin; side_out.spd_pin = L_early ? (int*)(ctx->status_pin + 32) : nullptr;
        ((double*)ctx->status_pin)[0] = 0.0; *(int*)(ctx->status_pin + 32) = 0;
        status_early = true;
        side_out.rows = (double*)abc_ws_alloc(ctx, K * (size_t)PPr * sizeof(double));
        if (L_early) side_out.Lpad = (double*)abc_ws_alloc(ctx, (size_t)PPr * PPr * sizeof(double));
        if (!side_out.rows || (L_early && !side_out.Lpad)) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
        if (!theta_ev_bound) ABC_HIP(ctx, hipEventRecord(ctx->ev_theta, ctx->stream));      // (the radix protocol's k_place_rows)
        ABC_HIP(ctx, hipStreamWaitEvent(ctx->side, ctx->ev_theta, 0));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->side;
        int rc = launch_theta_stats(ctx, theta, K, P, &theta_stats);
        if (rc == ABC_OK) rc = launch_post_tail(ctx, theta, K, P, theta_stats, &side_out);
        ctx->stream = main_stream;
        ABC_TRY(rc);
        ABC_HIP(ctx, hipEventRecord(ctx->ev_moments, ctx->side));
        moments_on_side = true;
    } else if (P <= 64 && K >= 2) {
        StageTimer tm(ctx, ST_GATHER_DV);
        ABC_TRY(launch_theta_stats(ctx, theta, K, P, &theta_stats));
        ABC_TRY(launch_dv_from_stats(ctx, theta_stats, P, dv));
    } else {
        ABC_TRY(launch_doubled_variance(ctx, theta, K, P, dv));
    }
    bool w_on_host = false;
    if (Kp == 0 || !io->theta_prev) {
        ABC_TRY(launch_fill(ctx, io->w, K, 1.0 / (double)K));                           // AbcUtil.cpp:543-544
    } else {
        if (wprev.ready) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_prev, 0));
        const double* sumsq = nullptr;
        if (W == 1) {
            ABC_TRY(launch_weights_raw(ctx, io->priors, theta, K, P, 0, K, io->theta_prev, Kp, io->w_prev, io->dv_prev, io->w, &wprev, &sumsq));
        } else {
            if (kn) ABC_TRY(launch_weights_raw(ctx, io->priors, theta, K, P, k0, kn, io->theta_prev, Kp, io->w_prev, io->dv_prev, w_mine, &wprev));
            ABC_TRY(comm_all_gather(ctx, w_mine, w_slices, kmax * 8));
            hipLaunchKernelGGL(k_unpad_slices, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, w_slices, K, W, kmax, io->w);
            ABC_HIP(ctx, hipGetLastError());
        }
        // (as in the single-GPU driver: the normalised weights go to the pinned scratch as they are written)
        double* mirror = nullptr;
        const bool alias_on_device = ctx->alias_mode == ABC_ALIAS_DEVICE && K >= ABC_ALIAS_DEV_MIN_K && K <= ABC_ALIAS_DEV_MAX_K;
        if (Nn && !alias_on_device) {          // (the device build reads the weights where they are)
            ABC_TRY(abc_pin_reserve(ctx, abc_alias_pin_bytes(K)));
            mirror = (double*)ctx->pin;
        }
        ABC_TRY(launch_normalize_l2(ctx, io->w, K, mirror, sumsq));                     // AbcUtil.cpp:583
        w_on_host = mirror != nullptr;
    }

    // ---- proposals for this rank's slice of the next set --------------------------------------------------------------------
    int spd = 0;
    bool have_spd = false;
    int alias_deferred = 0;
    uint64_t* parent_used = nullptr;
    double* L_used = nullptr;
    abc_perturb_prep prep_used = {nullptr, 0, nullptr};
    if (Nn) {
        uint64_t* parent = io->parent ? io->parent : (uint64_t*)abc_ws_alloc(ctx, Nn * 8);
        if (!parent) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sharded generation: workspace exhausted");
        double* L = nullptr;
        if (cfg->multivariate) {
            L = L_early ? L_early : (io->L ? io->L : (double*)abc_ws_alloc(ctx, P * P * 8));
            have_spd = true;
        }
        abc_perturb_prep prep = {moments_on_side ? side_out.rows : nullptr, io->seeds ? 1 : 0, moments_on_side ? side_out.Lpad : nullptr};
        if (moments_on_side) ABC_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_moments, 0));
        struct PrepArg {
            abc_ctx* ctx; const abc_rng* rng; const double* theta; const double* theta_stats; size_t K, P, Nn;
            uint64_t i0, seed_off; uint64_t* seeds; abc_perturb_prep* prep; double* L; int* spd_dev; const double* dv;
        };
        PrepArg pa = {ctx, rng, theta, theta_stats, K, P, Nn, cfg->next0, cfg->Nnext_total, io->seeds, &prep, moments_on_side ? nullptr : L, spd_dev, dv};
        auto hook = [](void* a) -> int {          // GPU work that does not need the alias table runs while the host builds it
            PrepArg* q = (PrepArg*)a;
            if (q->L) {
                if (q->theta_stats) {
                    StageTimer tm(q->ctx, ST_MVN);
                    ABC_TRY(launch_mvn_from_stats(q->ctx, q->theta_stats, q->P, q->L, q->spd_dev));
                } else {
                    ABC_TRY(launch_mvn_setup(q->ctx, q->theta, q->K, q->P, q->L, nullptr, q->spd_dev));
                }
            }
            return launch_perturb_prepare(q->ctx, q->rng, q->theta, q->K, q->P, q->i0, q->Nn, q->seeds, q->seed_off, q->prep,
                                          q->L ? 1 : 0, q->L ? q->L : q->dv);
        };
        {
            const int rc = launch_resample(ctx, rng, io->w, K, cfg->next0, Nn, parent, hook, &pa, uniform_w, raw_early, w_on_host,
                                           (ds_fail_dev && !uniform_w) ? pfail_early : nullptr, false, &alias_deferred);
            if (rc == ABC_INTERNAL_RETRY) return repeat_with_radix();
            ABC_TRY(rc);
        }
        ABC_TRY(launch_perturb(ctx, rng, theta, K, P, io->priors, parent, cfg->next0, Nn, cfg->multivariate,
                               cfg->multivariate ? L : dv, io->next, io->seeds, cfg->Nnext_total, &prep));
        parent_used = parent; L_used = L; prep_used = prep;
    } else if (cfg->multivariate && io->L) {
        have_spd = true;
        if (theta_stats) ABC_TRY(launch_mvn_from_stats(ctx, theta_stats, P, io->L, spd_dev));
        else ABC_TRY(launch_mvn_setup(ctx, theta, K, P, io->L, nullptr, spd_dev));
    }
    taus2_jump(rng, 2 * (uint64_t)cfg->Nnext_total);          // Nnext resampling draws + Nnext seeds of the whole set
    {
        double* hdr = (double*)ctx->status_pin;             // pinned, device-visible
        int* pspd = (int*)(ctx->status_pin + 32);
        if (!status_early) {                                // (first sets, generations without proposals: the two words by copies)
            hdr[0] = 0.0; *pspd = 0;
            ABC_HIP(ctx, hipMemcpyAsync(hdr, model, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            if (have_spd) ABC_HIP(ctx, hipMemcpyAsync(pspd, spd_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        }
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ncomp_host) *ncomp_host = (int32_t)hdr[0];
        spd = *pspd;
        if (ds_fail_dev && pfail_early && *pfail_early) return repeat_with_radix();      // (no host wait before this one)
        // the device build of the resampling table did not verify (every rank builds the same table from the same weights, so
        // every rank lands here alike): this rank's draws and proposals once more, with the table from the host
        if (alias_deferred && *(volatile int*)(ctx->status_pin + 44) && parent_used) {
            ctx->alias_dev_fallbacks++;
            const int mode = ctx->alias_mode;
            ctx->alias_mode = ABC_ALIAS_HOST;
            const int rc = launch_resample(ctx, &rng_entry, io->w, K, cfg->next0, Nn, parent_used);
            ctx->alias_mode = mode;
            ABC_TRY(rc);
            prep_used.seeds_done = 1;
            ABC_TRY(launch_perturb(ctx, &rng_entry, theta, K, P, io->priors, parent_used, cfg->next0, Nn, cfg->multivariate,
                                   cfg->multivariate ? L_used : dv, io->next, nullptr, cfg->Nnext_total, &prep_used));
            ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    // this rank's proposals the perturbation gave up on during THIS call (abc_generation_giveups; ADVICE round 4: the sharded driver
    // left the count of the context's last fused generation standing): the pinned word a proposal kernel raises when it gives up
    // tells whether the device counter has to be fetched at all
    {
        volatile unsigned* pgaveup = (volatile unsigned*)(ctx->status_pin + 56);
        if (Nn && *pgaveup && ctx->giveups_dev) {
            unsigned long long now = 0;
            ABC_HIP(ctx, hipMemcpyAsync(&now, ctx->giveups_dev, sizeof(now), hipMemcpyDeviceToHost, ctx->stream));
            ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->giveups_dev_known = now;
        }
        const unsigned long long gv = ctx->giveups_dev_known + ctx->giveups_host, before = ctx->giveups_seen;
        ctx->giveups_seen = gv;
        ctx->giveups_last_call = (Nn && gv > before) ? gv - before : 0ull;
    }
    if (spd) ABC_FAIL(ctx, ABC_ERR_NOT_SPD, "covariance of the selected particles is not positive definite");
    return ABC_OK;
}

// ---- host-pointer generation over several GPUs of one process -------------------------------------------------------------------
extern "C" int abc_generation_multi(abc_ctx* const* ctxs, int ndev, const abc_generation_cfg* cfg, const abc_generation_io* h,
                                    abc_rng* rng, int32_t* ncomp) {
    if (!ctxs || ndev < 1 || !ctxs[0]) return ABC_ERR_INVALID;
    abc_ctx* c0 = ctxs[0];
    if (!cfg || !h || !h->X || !h->Y || !h->obs || !h->priors || !h->idx || !h->w || !rng)
        ABC_FAIL(c0, ABC_ERR_INVALID, "multi-GPU generation: null argument");
    for (int d = 0; d < ndev; d++)
        if (!ctxs[d] || (ndev > 1 && (ctxs[d]->comm_kind != 1 || ctxs[d]->comm_world != ndev || ctxs[d]->comm_rank != d)))
            ABC_FAIL(c0, ABC_ERR_INVALID, "multi-GPU generation: context %d is not rank %d of an %d-rank communicator", d, d, ndev);
    const size_t N = cfg->N, M = cfg->M, P = cfg->P, K = cfg->K, Kp = cfg->Kp, Nn = cfg->Nnext;
    if (!N || !M || !P || !K || K > N) ABC_FAIL(c0, ABC_ERR_INVALID, "multi-GPU generation: bad sizes");
    if (Kp && (!h->theta_prev || !h->w_prev || !h->dv_prev)) ABC_FAIL(c0, ABC_ERR_INVALID, "multi-GPU generation: previous set missing");
    std::vector<int> rcs((size_t)ndev, ABC_OK);
    std::vector<int32_t> ncs((size_t)ndev, 0);
    std::vector<abc_rng> rngs((size_t)ndev, *rng);
    // one rendezvous of all workers: everyone deposits its set-up status, the last one to arrive releases the others
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0, any_failed = 0;
    auto agree = [&](int my_status) -> bool {
        std::unique_lock<std::mutex> lk(mu);
        if (my_status != ABC_OK) any_failed = 1;
        if (++arrived == ndev) cv.notify_all();
        else cv.wait(lk, [&] { return arrived == ndev; });
        return any_failed == 0;
    };
    auto worker = [&](int d) {
        abc_ctx* ctx = ctxs[d];
        auto fail = [&](int rc) { rcs[(size_t)d] = rc; };
        if (hipSetDevice(ctx->device) != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "hipSetDevice failed"); (void)agree(ABC_ERR_HIP); return fail(ABC_ERR_HIP); }
        // contiguous shards, as even as possible (low ranks first)
        auto lo = [&](size_t tot, int q) { return tot / (size_t)ndev * (size_t)q + ((size_t)q < tot % (size_t)ndev ? (size_t)q : tot % (size_t)ndev); };
        const size_t r0 = lo(N, d), n = lo(N, d + 1) - r0, i0 = lo(Nn, d), nn = lo(Nn, d + 1) - i0;
        // device buffers of this call (freed at the end; the host-pointer path is the convenience, not the fast one)
        std::vector<void*> bufs;
        auto dmalloc = [&](size_t bytes) -> void* {
            void* p = nullptr;
            if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return nullptr;
            bufs.push_back(p);
            return p;
        };
        auto cleanup = [&] { (void)hipStreamSynchronize(ctx->stream); for (void* p : bufs) (void)hipFree(p); };
        double* dX = (double*)dmalloc(n * M * 8);
        double* dY = (double*)dmalloc(n * P * 8);
        double* dobs = (double*)dmalloc(M * 8);
        abc_prior* dpri = (abc_prior*)dmalloc(P * sizeof(abc_prior));
        double *dtp = nullptr, *dwp = nullptr, *ddvp = nullptr;
        if (Kp) { dtp = (double*)dmalloc(Kp * P * 8); dwp = (double*)dmalloc(Kp * 8); ddvp = (double*)dmalloc(P * 8); }
        uint64_t* didx = (uint64_t*)dmalloc(K * 8);
        double* ddist = (double*)dmalloc(K * 8);
        double* dth = (double*)dmalloc(K * P * 8);
        double* dw = (double*)dmalloc(K * 8);
        double* ddv = (double*)dmalloc(P * 8);
        double* dL = (double*)dmalloc(P * P * 8);
        double* dnext = (double*)dmalloc(nn * P * 8);
        uint64_t* dpar = (uint64_t*)dmalloc(nn * 8);
        uint64_t* dseed = (uint64_t*)dmalloc(nn * 8);
        if (!dX || !dY || !dobs || !dpri || (Kp && (!dtp || !dwp || !ddvp)) || !didx || !ddist || !dth || !dw || !ddv || !dL || !dnext ||
            !dpar || !dseed) {
            snprintf(ctx->err, sizeof(ctx->err), "multi-GPU generation: device allocation failed");
            (void)agree(ABC_ERR_NOMEM);
            cleanup();
            return fail(ABC_ERR_NOMEM);
        }
        hipStream_t st = ctx->stream;
        bool ok = true;
        auto H = [&](hipError_t e) { if (e != hipSuccess && ok) { ok = false; snprintf(ctx->err, sizeof(ctx->err), "multi-GPU generation: %s", hipGetErrorString(e)); } };
        // a row shard of a column-major host matrix: one strided copy per matrix
        if (n) {
            H(hipMemcpy2DAsync(dX, n * 8, h->X + r0, N * 8, n * 8, M, hipMemcpyHostToDevice, st));
            H(hipMemcpy2DAsync(dY, n * 8, h->Y + r0, N * 8, n * 8, P, hipMemcpyHostToDevice, st));
        }
        H(hipMemcpyAsync(dobs, h->obs, M * 8, hipMemcpyHostToDevice, st));
        H(hipMemcpyAsync(dpri, h->priors, P * sizeof(abc_prior), hipMemcpyHostToDevice, st));
        if (Kp) {
            H(hipMemcpyAsync(dtp, h->theta_prev, Kp * P * 8, hipMemcpyHostToDevice, st));
            H(hipMemcpyAsync(dwp, h->w_prev, Kp * 8, hipMemcpyHostToDevice, st));
            H(hipMemcpyAsync(ddvp, h->dv_prev, P * 8, hipMemcpyHostToDevice, st));
        }
        // every worker reports whether its allocations and uploads went through BEFORE anyone enters the first collective:
        // if one failed, all skip the generation (a lone early return left the others inside RCCL for ever)
        if (!agree(ok ? ABC_OK : ABC_ERR_HIP)) {
            cleanup();
            if (ok) snprintf(ctx->err, sizeof(ctx->err), "multi-GPU generation: skipped, the set-up of another device failed");
            return fail(ok ? ABC_ERR_COMM : ABC_ERR_HIP);
        }
        abc_sharded_cfg sc;
        memset(&sc, 0, sizeof(sc));
        sc.n_local = n; sc.row0 = r0; sc.N_total = N; sc.M = M; sc.P = P; sc.K = K; sc.Kp = Kp;
        sc.nnext_local = nn; sc.next0 = i0; sc.Nnext_total = Nn; sc.train_frac = cfg->train_frac;
        sc.max_comp = cfg->max_comp; sc.rule = cfg->rule; sc.multivariate = cfg->multivariate;
        abc_generation_io io;
        memset(&io, 0, sizeof(io));
        io.X = dX; io.Y = dY; io.obs = dobs; io.priors = dpri; io.theta_prev = dtp; io.w_prev = dwp; io.dv_prev = ddvp;
        io.idx = didx; io.dist = ddist; io.theta = dth; io.w = dw; io.dv = ddv; io.L = dL; io.next = dnext; io.parent = dpar; io.seeds = dseed;
        const int rc = abc_generation_sharded_dev(ctx, &sc, &io, &rngs[(size_t)d], &ncs[(size_t)d]);
        if (rc != ABC_OK) { cleanup(); return fail(rc); }
        if (d == 0) {          // replicated outputs: rank 0's copy
            H(hipMemcpyAsync(h->idx, didx, K * 8, hipMemcpyDeviceToHost, st));
            if (h->dist) H(hipMemcpyAsync(h->dist, ddist, K * 8, hipMemcpyDeviceToHost, st));
            if (h->theta) H(hipMemcpyAsync(h->theta, dth, K * P * 8, hipMemcpyDeviceToHost, st));
            H(hipMemcpyAsync(h->w, dw, K * 8, hipMemcpyDeviceToHost, st));
            if (h->dv) H(hipMemcpyAsync(h->dv, ddv, P * 8, hipMemcpyDeviceToHost, st));
            if (h->L && cfg->multivariate) H(hipMemcpyAsync(h->L, dL, P * P * 8, hipMemcpyDeviceToHost, st));
        }
        if (nn) {              // this rank's rows of the next set
            if (h->next) H(hipMemcpy2DAsync(h->next + i0, Nn * 8, dnext, nn * 8, nn * 8, P, hipMemcpyDeviceToHost, st));
            if (h->parent) H(hipMemcpyAsync(h->parent + i0, dpar, nn * 8, hipMemcpyDeviceToHost, st));
            if (h->seeds) H(hipMemcpyAsync(h->seeds + i0, dseed, nn * 8, hipMemcpyDeviceToHost, st));
        }
        H(hipStreamSynchronize(st));
        cleanup();
        if (!ok) return fail(ABC_ERR_HIP);
    };
    if (ndev == 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int d = 0; d < ndev; d++) th.emplace_back(worker, d);
        for (auto& t : th) t.join();
    }
    for (int d = 0; d < ndev; d++)
        if (rcs[(size_t)d] != ABC_OK) {
            if (d != 0) snprintf(c0->err, sizeof(c0->err), "device %d: %s", ctxs[d]->device, ctxs[d]->err);
            return rcs[(size_t)d];
        }
    *rng = rngs[0];
    if (ncomp) *ncomp = ncs[0];
    return ABC_OK;
}
