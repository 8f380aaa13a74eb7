// Sufficient statistics of one SMC set in ONE pass over the particle x (metric|parameter) matrix:
// column sums and the Gram matrix of the shifted data [X - s | Y - s], separately for the PLS
// training rows and the validation rows.  Replaces the materialised z-score copies and the
// X'Y / X'X products inside PLS::Model (reference call sites AbcUtil.cpp:432-446; SURVEY 8a a2,a3).
//
// Layout: X (N x M) and Y (N x P) are column-major, i.e. each metric is a contiguous particle-major
// vector; a wave-instruction reads 1 KiB (128 particles) of one column, fully coalesced.
// A 128-row tile of all C16 = 16*C columns is staged in LDS ([column][row], row stride padded by
// 2 doubles so the 16-column x 4-row MFMA operand fetch is bank-conflict free), then every wave
// runs v_mfma_f64_16x16x4_f64 over its 32 rows of the tile for all C(C+1)/2 upper-triangular
// 16x16 blocks (fp64 in, fp64 accumulate).  HBM-bound: 8*(M+P) bytes per particle, read once.
#include "abc_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int TR = 128;      // rows per tile
constexpr int TRP = TR + 2;  // LDS column stride in doubles (== 2 mod 32 -> conflict-free b64 reads)

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int C>
struct GramDims {
    static constexpr int C16 = 16 * C;
    static constexpr int NBLK = C * (C + 1) / 2;
    static constexpr int NI = 4 * C;               // 16-byte vectors per thread per tile
    static constexpr int PSZ = NBLK * 256 + C16;   // doubles per work-group partial record
    static constexpr int LDS_D = (C16 * TRP > PSZ) ? C16 * TRP : PSZ;
};

// grid = (G, 2): blockIdx.y = partition (0: rows [0,split) training, 1: rows [split,n) validation)
template <int C>
__global__ __launch_bounds__(256) void k_gram(const double* __restrict__ X, const double* __restrict__ Y,
                                              size_t ldx, size_t ldy, int M, int P, long long n,
                                              long long split, const double* __restrict__ shift,
                                              double* __restrict__ partial, int vec_ok) {
    using D = GramDims<C>;
    extern __shared__ double lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: column pointers live in SGPRs
    const int part = blockIdx.y, G = gridDim.x, g = blockIdx.x;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;  // tiles start on an even row so 16-B loads stay aligned
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + TR - 1) / TR : 0;

    d4 acc[D::NBLK];
#pragma unroll
    for (int b = 0; b < D::NBLK; b++) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};
    double colsum[D::NI];
    double sh[D::NI];
    const double* cptr[D::NI];
#pragma unroll
    for (int i = 0; i < D::NI; i++) {
        const int c = wave + 4 * i;  // one column per wave-instruction
        colsum[i] = 0.0;
        if (c < M) { cptr[i] = X + (size_t)c * ldx; sh[i] = shift[c]; }
        else if (c < M + P) { cptr[i] = Y + (size_t)(c - M) * ldy; sh[i] = shift[c]; }
        else { cptr[i] = nullptr; sh[i] = 0.0; }
    }

    d2 v[D::NI];
    auto fetch = [&](long long tile) {
        const long long row0 = t0 + tile * TR;
        const long long r = row0 + 2 * lane;
        const bool full = (row0 >= r_begin) && (row0 + TR <= r_end);
        if (full && vec_ok) {
#pragma unroll
            for (int i = 0; i < D::NI; i++)
                v[i] = cptr[i] ? *reinterpret_cast<const d2*>(cptr[i] + r) : (d2){0.0, 0.0};
        } else {
#pragma unroll
            for (int i = 0; i < D::NI; i++) {
                d2 x = (d2){sh[i], sh[i]};  // masked rows contribute (x - shift) = 0
                if (cptr[i]) {
                    if (r >= r_begin && r < r_end) x.x = cptr[i][r];
                    if (r + 1 >= r_begin && r + 1 < r_end) x.y = cptr[i][r + 1];
                }
                v[i] = x;
            }
        }
    };

    long long tile = g;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += G) {
        __syncthreads();  // previous tile's operand reads are done
#pragma unroll
        for (int i = 0; i < D::NI; i++) {
            const int c = wave + 4 * i;
            d2 z = (d2){v[i].x - sh[i], v[i].y - sh[i]};
            colsum[i] += z.x + z.y;
            *reinterpret_cast<d2*>(&lds[c * TRP + 2 * lane]) = z;
        }
        __syncthreads();
        if (tile + G < ntiles) fetch(tile + G);  // next tile's HBM latency hides under the MFMAs
        const int cl = lane & 15, q = lane >> 4;
#pragma unroll 2
        for (int s = 0; s < TR / 16; s++) {
            const int rb = wave * (TR / 4) + 4 * s + q;
            double a[C];
#pragma unroll
            for (int b = 0; b < C; b++) a[b] = lds[(16 * b + cl) * TRP + rb];
            int blk = 0;
#pragma unroll
            for (int bi = 0; bi < C; bi++)
#pragma unroll
                for (int bj = bi; bj < C; bj++) {
                    acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[blk], 0, 0, 0);
                    blk++;
                }
        }
    }

    // cross-wave reduction in a fixed order (deterministic), through LDS
    __syncthreads();
    for (int w = 0; w < 4; w++) {
        if (wave == w) {
#pragma unroll
            for (int b = 0; b < D::NBLK; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int e = b * 256 + r * 64 + lane;
                    lds[e] = (w == 0 ? 0.0 : lds[e]) + acc[b][r];
                }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < D::NI; i++) {
        const double s = wave_sum(colsum[i]);
        if (lane == 0) lds[D::NBLK * 256 + wave + 4 * i] = s;
    }
    __syncthreads();
    double* out = partial + ((size_t)part * G + g) * D::PSZ;
    for (int e = t; e < D::PSZ; e += 256) out[e] = lds[e];
}

// Sum the per-work-group partial records in a fixed order and scatter into the stats record.
template <int C>
__global__ __launch_bounds__(256) void k_stats_reduce(const double* __restrict__ partial, int G,
                                                      double* __restrict__ stats, long long n_train,
                                                      long long n_test) {
    using D = GramDims<C>;
    const StatsLayout L = stats_layout(D::C16, 0);
    const int part = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e == 0 && part == 0) { stats[L.off_n] = (double)n_train; stats[L.off_n + 1] = (double)n_test; }
    if (e >= D::PSZ) return;
    const double* p = partial + (size_t)part * G * D::PSZ + e;
    double s = 0.0;
#pragma unroll 8
    for (int g = 0; g < G; g++) s += p[(size_t)g * D::PSZ];
    if (e >= D::NBLK * 256) {
        stats[L.off_sum[part] + (e - D::NBLK * 256)] = s;
        return;
    }
    int blk = e >> 8;
    const int r = (e >> 6) & 3, lane = e & 63;
    int bi = 0;
    while (blk >= C - bi) { blk -= C - bi; bi++; }
    const int bj = bi + blk;
    // v_mfma_f64_16x16x4_f64 C/D map: row = (lane>>4) + 4*reg, col = lane&15
    const int row = 16 * bi + (lane >> 4) + 4 * r, col = 16 * bj + (lane & 15);
    double* Gm = stats + L.off_G[part];
    Gm[row + (size_t)D::C16 * col] = s;
    if (bi != bj) Gm[col + (size_t)D::C16 * row] = s;
}

// shift[c] = mean of the first min(n,256) rows of column c (a pilot centre: keeps the one-pass
// Gram numerically equivalent to the reference's two-pass centred sums)
__global__ void k_pilot_shift(const double* __restrict__ X, const double* __restrict__ Y, size_t ldx,
                              size_t ldy, int M, int P, long long n, int C16, double* __restrict__ shift) {
    const int c = blockIdx.x;
    const int lane = threadIdx.x;
    const long long m = n < 256 ? n : 256;
    double s = 0.0;
    if (c < M + P) {
        const double* p = (c < M) ? X + (size_t)c * ldx : Y + (size_t)(c - M) * ldy;
        for (long long r = lane; r < m; r += 64) s += p[r];
    }
    s = wave_sum(s);
    if (lane == 0) shift[c] = (c < M + P && m > 0) ? s / (double)m : 0.0;
}

template <int C>
int run_gram(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
             size_t P, long long split, double* stats) {
    using D = GramDims<C>;
    const StatsLayout L = stats_layout(M, P);
    const long long ntr = split, nte = (long long)n - split;
    const long long tiles = ((ntr > nte ? ntr : nte) + TR - 1) / TR + 1;
    long long G = tiles / 4;
    if (G < 1) G = 1;
    if (G > 384) G = 384;
    const size_t pbytes = (size_t)2 * G * D::PSZ * sizeof(double);
    double* partial = (double*)abc_ws_alloc(ctx, pbytes);
    if (!partial) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted (%zu B)", pbytes);
    const int vec_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)Y & 15) == 0);
    const size_t lds_bytes = (size_t)D::LDS_D * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram<C>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
        attr_set = true;
    }
    {
        StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
        hipLaunchKernelGGL(k_gram<C>, dim3((unsigned)G, 2), dim3(256), lds_bytes, ctx->stream, X, Y, ldx, ldy, (int)M,
                           (int)P, (long long)n, split, stats + L.off_shift, partial, vec_ok);
    }
    ABC_HIP(ctx, hipGetLastError());
    StageTimer tm2(ctx, ctx->in_mvn ? -1 : ST_STATS_REDUCE);
    hipLaunchKernelGGL(k_stats_reduce<C>, dim3((D::PSZ + 255) / 256, 2), dim3(256), 0, ctx->stream, partial, (int)G,
                       stats, ntr, nte);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

}  // namespace

int launch_stats_shift(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                       size_t P, double* stats) {
    const StatsLayout L = stats_layout(M, P);
    hipLaunchKernelGGL(k_pilot_shift, dim3((unsigned)L.C16), dim3(64), 0, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P,
                       (long long)n, (int)L.C16, stats + L.off_shift);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_stats_accumulate(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy,
                            size_t M, size_t P, uint64_t row0, uint64_t n_train_global, double* stats) {
    long long split = 0;
    if (n_train_global > row0) split = (long long)((n_train_global - row0) < n ? (n_train_global - row0) : n);
    const size_t C = (M + P + 15) / 16;
    switch (C) {
        case 1: return run_gram<1>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        case 2: return run_gram<2>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        case 3: return run_gram<3>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        case 4: return run_gram<4>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        case 5: return run_gram<5>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        case 6: return run_gram<6>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        default:
            ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "gram: M+P = %zu exceeds 96 columns", M + P);
    }
}
