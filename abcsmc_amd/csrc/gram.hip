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
#include <stdlib.h>

#include <vector>

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

// C = 16-column blocks of [X|Y]; the trailing CY blocks hold only parameter (Y) / padding columns.
// Y'Y is needed on its diagonal only (PRESS, z-scores), so the CY(CY+1)/2 pure-Y blocks are skipped and
// the diagonal comes from a per-column sum of squares taken while the tile is staged.
// NW = waves per work-group: 8 for narrow sets (C <= 3).  fp64 MFMA only reaches its rate with several
// waves per SIMD issuing (scripts/ubench.hip: 33 TFLOP/s at 1 wave/SIMD, 44 at 2, 46-49 at 4-8), and the
// 50 KB LDS tile admits three work-groups per CU, so each work-group brings two waves per SIMD.
template <int C, int CY>
struct GramDims {
    static constexpr int C16 = 16 * C;
    static constexpr int NBLK = C * (C + 1) / 2 - CY * (CY + 1) / 2;
    static constexpr int NW = (C <= 3) ? 8 : 4;
    static constexpr int NT = 64 * NW;
    static constexpr int NI = 16 * C / NW;             // 16-byte vectors per thread per tile
    static constexpr int PSZ = NBLK * 256 + 2 * C16;   // doubles per work-group partial record
    static constexpr int LDS_D = (C16 * TRP > PSZ) ? C16 * TRP : PSZ;
    static constexpr int EPT = (NBLK * 256 + NT - 1) / NT;   // partial-record elements per thread
    static constexpr int IQ = 16 * (C - CY) / NW;            // staged columns i >= IQ may lie in a skipped Y'Y block
};

// grid = (G, 2): blockIdx.y = partition (0: rows [0,split) training, 1: rows [split,n) validation)
// TABLE = true: the columns come from a device table of column pointers (X reinterpreted as
// `const double* const*`, NULL = padding) instead of the two dense arrays; used to cover wide sets
// (M+P > 96) with several launches over column-group pairs.
template <int C, int CY, bool TABLE = false>
__global__ __launch_bounds__((GramDims<C, CY>::NT)) void k_gram(const double* __restrict__ X, const double* __restrict__ Y,
                                              size_t ldx, size_t ldy, int M, int P, long long n,
                                              long long split, const double* __restrict__ shift,
                                              double* __restrict__ partial, int vec_ok) {
    using D = GramDims<C, CY>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NW = D::NW, NT = D::NT;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // wave-uniform: column pointers live in SGPRs
    const int part = blockIdx.y, G = gridDim.x, g = blockIdx.x;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;  // tiles start on an even row so 16-B loads stay aligned
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + TR - 1) / TR : 0;

    d4 acc[D::NBLK];
#pragma unroll
    for (int b = 0; b < D::NBLK; b++) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};
    double colsum[D::NI], colsq[D::NI];
    double sh[D::NI], keep[D::NI];
    const double* cptr[D::NI];
#pragma unroll
    for (int i = 0; i < D::NI; i++) {
        const int c = wave + NW * i;  // one column per wave-instruction; padding columns re-read column 0, zeroed
        colsum[i] = 0.0;
        colsq[i] = 0.0;
        if constexpr (TABLE) {
            const double* const* tab = reinterpret_cast<const double* const*>(X);
            const double* q = tab[c];
            keep[i] = q ? 1.0 : 0.0;
            cptr[i] = q ? q : tab[0];
            sh[i] = q ? shift[c] : 0.0;
        } else {
            keep[i] = (c < M + P) ? 1.0 : 0.0;
            cptr[i] = (c < M) ? X + (size_t)c * ldx : (c < M + P) ? Y + (size_t)(c - M) * ldy : X;
            sh[i] = (c < M + P) ? shift[c] : 0.0;
        }
    }

    d2 v[D::NI];
    auto fetch = [&](long long tile) {
        const long long row0 = t0 + tile * TR;
        const long long r = row0 + 2 * lane;
        const bool full = (row0 >= r_begin) && (row0 + TR <= r_end);
        if (full && vec_ok) {
#pragma unroll
            for (int i = 0; i < D::NI; i++)      // non-temporal: read once, streamed past L2 / Infinity Cache (see dma16)
                v[i] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(cptr[i] + r));
        } else {   // edge tile / unaligned columns: clamp the address, mask the value (no divergence)
            const long long ra = r < r_begin ? r_begin : (r >= r_end ? r_end - 1 : r);
            const long long rb = r + 1 < r_begin ? r_begin : (r + 1 >= r_end ? r_end - 1 : r + 1);
            const bool oka = (r >= r_begin) && (r < r_end), okb = (r + 1 >= r_begin) && (r + 1 < r_end);
#pragma unroll
            for (int i = 0; i < D::NI; i++) {
                const double xa = cptr[i][ra], xb = cptr[i][rb];
                v[i] = (d2){oka ? xa : sh[i], okb ? xb : sh[i]};   // masked rows contribute (x - shift) = 0
            }
        }
    };

    long long tile = g;
    if (tile < ntiles) fetch(tile);
    for (; tile < ntiles; tile += G) {
        __syncthreads();  // previous tile's operand reads are done
#pragma unroll
        for (int i = 0; i < D::NI; i++) {
            const int c = wave + NW * i;
            d2 z = (d2){(v[i].x - sh[i]) * keep[i], (v[i].y - sh[i]) * keep[i]};
            colsum[i] += z.x + z.y;
            if (i >= D::IQ) {          // sums of squares are only needed where the Gram block is skipped
                colsq[i] = fma(z.x, z.x, colsq[i]);
                colsq[i] = fma(z.y, z.y, colsq[i]);
            }
            *reinterpret_cast<d2*>(&lds[c * TRP + 2 * lane]) = z;
        }
        __syncthreads();
        if (tile + G < ntiles) fetch(tile + G);  // next tile's HBM latency hides under the MFMAs
        const int cl = lane & 15, q = lane >> 4;
#pragma unroll
        for (int s = 0; s < TR / (4 * NW); s++) {
            const int rb = wave * (TR / NW) + 4 * s + q;
            double a[C];
#pragma unroll
            for (int b = 0; b < C; b++) a[b] = lds[(16 * b + cl) * TRP + rb];
            int blk = 0;
#pragma unroll
            for (int bi = 0; bi < C - CY; bi++)
#pragma unroll
                for (int bj = bi; bj < C; bj++) {
                    acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[blk], 0, 0, 0);
                    blk++;
                }
        }
    }

    // ---- epilogue: fixed-order (deterministic) reductions through the now idle tile buffer ------------
    double* out = partial + ((size_t)part * G + g) * D::PSZ;
    __syncthreads();
    if constexpr (4 * D::NBLK * 256 <= D::LDS_D) {
        // four waves at a time park their accumulators in slabs; every thread adds the slabs of "its" elements
        double tot[D::EPT];
#pragma unroll
        for (int k = 0; k < D::EPT; k++) tot[k] = 0.0;
        for (int round = 0; round < NW / 4; round++) {
            if ((wave >> 2) == round) {
#pragma unroll
                for (int b = 0; b < D::NBLK; b++)
#pragma unroll
                    for (int r = 0; r < 4; r++) lds[(wave & 3) * (D::NBLK * 256) + b * 256 + r * 64 + lane] = acc[b][r];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < D::EPT; k++) {
                const int e = t + k * NT;
                if (e < D::NBLK * 256)
                    tot[k] += ((lds[e] + lds[D::NBLK * 256 + e]) + lds[2 * D::NBLK * 256 + e]) + lds[3 * D::NBLK * 256 + e];
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < D::EPT; k++) {
            const int e = t + k * NT;
            if (e < D::NBLK * 256) out[e] = tot[k];
        }
    } else {
        for (int w = 0; w < NW; w++) {
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
        for (int e = t; e < D::NBLK * 256; e += NT) out[e] = lds[e];
    }
    // column sums / sums of squares: lane-major park, then one thread per column adds its 64 lanes in order
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < D::NI; i++) lds[i * NT + t] = half ? colsq[i] : colsum[i];
        __syncthreads();
        if (t < D::C16) {
            const int i = t / NW, w = t % NW;     // column t was staged by wave w as its i-th column
            double s = 0.0;
            for (int l = 0; l < 64; l++) s += lds[i * NT + w * 64 + l];
            out[D::NBLK * 256 + half * D::C16 + t] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant for narrow sets (C <= 3) with 16-B aligned columns: tiles go HBM -> LDS directly
// (global_load_lds_dwordx4, no VGPR round trip, no LDS write phase), a ring of three tile buffers (two DMAs in
// flight behind the tile being read, counted s_waitcnt vmcnt), ONE barrier per tile.
// The shift is subtracted when the MFMA operand is read (3 VALU ops per 5 MFMAs: free next to the matrix
// pipe), which is also where the column sums / sums of squares are taken and edge rows are masked.
template <int C, int CY, int NW, int R = 3>
struct GramDimsDma {
    static constexpr int C16 = 16 * C;
    static constexpr int NBLK = C * (C + 1) / 2 - CY * (CY + 1) / 2;
    static constexpr int NT = 64 * NW;
    static constexpr int NI = C16 / NW;                 // DMA instructions (one column x 128 rows) per wave per tile
    static constexpr int PSZ = NBLK * 256 + 2 * C16;    // same partial record as k_gram
    static constexpr int BUF = C16 * TRP;
    static constexpr int EPT = (NBLK * 256 + NT - 1) / NT;
    static constexpr int LDS_D = R * BUF;               // ring of R tiles: R - 1 DMAs in flight behind the one being read
};

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS at dst + 16 * lane (dst wave-uniform).
// Issued as inline assembly ON PURPOSE: with __builtin_amdgcn_global_load_lds the compiler, which cannot tell the ring
// buffer being refilled from the one being read, puts s_waitcnt vmcnt(0) in front of the next LDS operand read, i.e. it
// waits for the prefetch it has just issued and nothing overlaps (seen in the ISA; the kernel then ran at the SUM of
// its memory and MFMA times).  All waits for these loads are the explicit counted s_waitcnt in the loop.
template <bool NT = true>
__device__ __forceinline__ void dma16(const double* gsrc, double* dst) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) void*)dst);
    // nt: every byte of X and Y is read exactly once -- streamed past L2 / Infinity Cache instead of allocated there.  Without
    // it the kernel's time depended on what the previous kernels had left in the caches (1 GB flushed between launches:
    // 134 us; behind a generation's 128 MB of freshly written proposals: 95 us); with it 84-86 us in every state.
    if constexpr (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(m0v), "v"(gsrc) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory");
}

// PRIV = true (NW == 8): every wave stages and consumes its OWN 16 rows of each tile (8 columns x 16 rows per DMA
// instruction, its own ring of three 6 KB chunks), so the main loop has no work-group barrier at all: a wave only
// waits for its own DMAs and the eight waves drift apart instead of meeting once per tile.  Lane l of instruction i
// fetches rows 2*((l>>3) ^ (i&1)) .. +1 of column 8 i + (l&7); the row-pair swizzle of odd instructions puts the two
// 8-column halves of a 16-column MFMA operand on opposite halves of the LDS banks (conflict-free b64 reads, no padding).
// TABLE = true (wide sets, run_gram_grouped): X is a device table of C16 column pointers (NULL = padding column), Y unused
template <int C, int CY, int NW, bool PRIV, int R = 3, bool TABLE = false>
__global__ __launch_bounds__((GramDimsDma<C, CY, NW>::NT)) void k_gram_dma(
    const double* __restrict__ X, const double* __restrict__ Y, size_t ldx, size_t ldy, int M, int P, long long n,
    long long split, const double* __restrict__ shift, double* __restrict__ partial) {
    using D = GramDimsDma<C, CY, NW, R>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NT = D::NT;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int part = blockIdx.y, G = gridDim.x, g = blockIdx.x;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;
    // rows per tile: the shared-tile variants use TR; wave-private staging gives every wave 16 rows, so 8 waves cover TR
    // and the 4-wave variant for 4..6 column blocks (whose rings would not fit the LDS with 8 waves) covers 64
    constexpr int TRK = PRIV ? 16 * NW : TR;
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + TRK - 1) / TRK : 0;
    const long long rmax = (n - 2) & ~1LL;             // last in-bounds 16-B row pair (n is even on this path)

    constexpr int NIW = PRIV ? D::C16 / 8 : D::NI;     // DMA instructions per wave per tile
    constexpr int CHB = D::C16 * 16;                   // PRIV: doubles per wave chunk (16 rows x C16 columns)
    static_assert(!PRIV || ((NW == 8 && TR == 128) || NW == 4), "wave-private staging: 16 rows per wave, 8 or 4 waves");
    const double* cptr[NIW];
#pragma unroll
    for (int i = 0; i < NIW; i++) {
        const int c = PRIV ? 8 * i + (lane & 7) : wave + NW * i;   // padding columns re-read column 0, masked at operand read
        if constexpr (TABLE) {
            const double* const* tab = reinterpret_cast<const double* const*>(X);
            const double* pc = tab[c];
            cptr[i] = pc ? pc : tab[0];
        } else {
            cptr[i] = (c < M) ? X + (size_t)c * ldx : (c < M + P) ? Y + (size_t)(c - M) * ldy : X;
        }
    }
    double* const wring = lds + (PRIV ? wave * (R * CHB) : 0);
    auto stage = [&](long long tile, int slot) {
        if constexpr (PRIV) {
#pragma unroll
            for (int i = 0; i < NIW; i++) {
                long long r = t0 + tile * TRK + 16 * wave + 2 * ((lane >> 3) ^ (i & 1));
                r = r > rmax ? rmax : r;                // rows past the end are masked later; keep the address legal
                dma16(cptr[i] + r, wring + slot * CHB + i * 128);
            }
        } else {
            double* buf = lds + slot * D::BUF;
            long long r = t0 + tile * TRK + 2 * lane;
            r = r > rmax ? rmax : r;
#pragma unroll
            for (int i = 0; i < NIW; i++) {
                const int c = wave + NW * i;
                dma16(cptr[i] + r, buf + c * TRP);
            }
        }
    };

    const int cl = lane & 15, q = lane >> 4;
    double sh[C], keep[C];
#pragma unroll
    for (int b = 0; b < C; b++) {
        const int c = 16 * b + cl;
        bool real = c < M + P;
        if constexpr (TABLE) real = real && reinterpret_cast<const double* const*>(X)[c] != nullptr;
        keep[b] = real ? 1.0 : 0.0;
        sh[b] = real ? shift[c] : 0.0;
    }
    d4 acc[D::NBLK];
#pragma unroll
    for (int b = 0; b < D::NBLK; b++) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};
    double cs[C], cq[C];
#pragma unroll
    for (int b = 0; b < C; b++) { cs[b] = 0.0; cq[b] = 0.0; }

    // PRIV operand address inside a chunk: region (2 b + cl/8) * 128 + 2 (cl & 7) + (q & 1) + 16 * (rowpair ^ (cl >> 3))
    const int poff = PRIV ? (cl >> 3) * 128 + 2 * (cl & 7) + (q & 1) : 0;
    const int pswz = cl >> 3;
    int cur = 0;
    long long tile = g;
#pragma unroll
    for (int d = 0; d < R - 1; d++)
        if (tile + d * G < ntiles) stage(tile + d * G, d);
    for (; tile < ntiles; tile += G) {
        const double* buf = PRIV ? wring + cur * CHB : lds + cur * D::BUF;
        // The DMA of THIS tile must have landed; the next tile's (issued later by the same wave, NIW instructions)
        // may stay in flight.  hipcc does not reliably place these waits for LDS-DMA issued in an earlier loop
        // iteration (seen in the ISA), so they are explicit.
        // tiles still to come after this one, capped at R - 2 (those may stay in flight)
        {
            const long long ahead = (ntiles - 1 - tile) / G;
            if (R >= 6 && ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NIW) : "memory");
            else if (R >= 5 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NIW) : "memory");
            else if (R >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIW) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NIW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (!PRIV) __syncthreads();   // every wave's share of this tile is in LDS; the buffer refilled below is idle
        if (tile + (R - 1) * G < ntiles) stage(tile + (R - 1) * G, (cur + R - 1) % R);
        const long long row0 = t0 + tile * TRK;
        const bool full = (row0 >= r_begin) && (row0 + TRK <= r_end);
#pragma unroll
        for (int s = 0; s < TRK / (4 * NW); s++) {
            const int rb = wave * (TRK / NW) + 4 * s + q;
            const long long grow = row0 + rb;
            const bool ok = full || (grow >= r_begin && grow < r_end);
            double a[C];
#pragma unroll
            for (int b = 0; b < C; b++) {
                const double v = PRIV ? buf[b * 256 + poff + 16 * ((2 * s + (q >> 1)) ^ pswz)] : buf[(16 * b + cl) * TRP + rb];
                const double z = (v - sh[b]) * keep[b];
                a[b] = ok ? z : 0.0;
                cs[b] += a[b];
                if (b >= C - CY) cq[b] = fma(a[b], a[b], cq[b]);
            }
            int blk = 0;
#pragma unroll
            for (int bi = 0; bi < C - CY; bi++)
#pragma unroll
                for (int bj = bi; bj < C; bj++) {
                    acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[blk], 0, 0, 0);
                    blk++;
                }
        }
        cur = (cur + 1) % R;
    }

    // ---- epilogue (same partial record as k_gram) -------------------------------------------------------
    double* out = partial + ((size_t)part * G + g) * D::PSZ;
    __syncthreads();
    double tot[D::EPT];
#pragma unroll
    for (int k = 0; k < D::EPT; k++) tot[k] = 0.0;
    // accumulators of EW waves at a time go through LDS (four; two when four records of NBLK x 256 doubles exceed it)
    constexpr int EW = ((size_t)4 * D::NBLK * 256 * sizeof(double) <= 160 * 1024) ? 4 : 2;
    for (int round = 0; round < NW / EW; round++) {
        if ((wave / EW) == round) {
#pragma unroll
            for (int b = 0; b < D::NBLK; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) lds[(wave % EW) * (D::NBLK * 256) + b * 256 + r * 64 + lane] = acc[b][r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < D::EPT; k++) {
            const int e = t + k * NT;
            if (e < D::NBLK * 256) {
                if constexpr (EW == 4)
                    tot[k] += ((lds[e] + lds[D::NBLK * 256 + e]) + lds[2 * D::NBLK * 256 + e]) + lds[3 * D::NBLK * 256 + e];
                else
                    tot[k] += lds[e] + lds[D::NBLK * 256 + e];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < D::EPT; k++) {
        const int e = t + k * NT;
        if (e < D::NBLK * 256) out[e] = tot[k];
    }
    // column sums: lane (cl, q) of every wave holds the partial sum of column 16 b + cl over its rows
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < C; b++) lds[b * NT + t] = half ? cq[b] : cs[b];
        __syncthreads();
        if (t < D::C16) {
            const int b = t >> 4, c = t & 15;
            double sacc = 0.0;
            for (int w = 0; w < NW; w++)
                for (int qq = 0; qq < 4; qq++) sacc += lds[b * NT + w * 64 + qq * 16 + c];
            out[D::NBLK * 256 + half * D::C16 + t] = sacc;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// 49..96 columns (4..6 column blocks, BASELINE configs[3]: 64 metrics + 32 parameters), EIGHT waves: the shape is bound by the
// fp64 matrix pipe (18 blocks per four rows), and ONE wave per SIMD does not keep that pipe busy (33 TFLOP/s against 44-49 with
// two: scripts/ubench.hip; the four-wave variant above ran at exactly that rate, 0.35-0.37 ms on the configs[3] shard).  Two
// waves per SIMD need rings of at most 18 KB per wave: wave-private chunks of EIGHT rows (64-row tiles, three-chunk rings of
// 6 KB per wave at 96 columns).  DMA instruction i of a chunk brings the 16 columns of MFMA block i: lane l fetches rows
// 2 (l >> 4), +1 of column 16 i + (l & 15), 64-byte pieces of a column (the other half of the 128-byte line goes to the
// neighbouring wave of the same work-group at the same time: the loads are therefore NOT non-temporal, see run_gram_dma8); in LDS that is element (column c, row r) of block i at
// 128 i + 2 c + 32 (r >> 1) + (r & 1), so the operand read of k-step s (lane (cl, q) -> row 4 s + q) covers 64 consecutive
// doubles: conflict-free without a swizzle.  Everything else (shift at operand read, masks, column sums, epilogue) as above.
template <int C, int CY, bool NTL = true>
__global__ __launch_bounds__(512) void k_gram_dma8(const double* __restrict__ X, const double* __restrict__ Y, size_t ldx, size_t ldy,
                                                  int M, int P, long long n, long long split, const double* __restrict__ shift,
                                                  double* __restrict__ partial) {
    using D = GramDimsDma<C, CY, 8, 3>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int NW = 8, NT = 512, R = 3, TRK = 64, CH = D::C16 * 8;      // CH: doubles per wave chunk (8 rows x C16 columns)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int part = blockIdx.y, G = gridDim.x, g = blockIdx.x;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + TRK - 1) / TRK : 0;
    const long long rmax = (n - 2) & ~1LL;
    const double* cptr[C];
#pragma unroll
    for (int i = 0; i < C; i++) {
        const int c = 16 * i + (lane & 15);              // padding columns re-read column 0, masked at operand read
        cptr[i] = (c < M) ? X + (size_t)c * ldx : (c < M + P) ? Y + (size_t)(c - M) * ldy : X;
    }
    double* const wring = lds + wave * (R * CH);
    auto stage = [&](long long tile, int slot) {
        long long r = t0 + tile * TRK + 8 * wave + 2 * (lane >> 4);
        r = r > rmax ? rmax : r;                         // rows past the end are masked later; keep the address legal
#pragma unroll
        for (int i = 0; i < C; i++) dma16<NTL>(cptr[i] + r, wring + slot * CH + i * 128);
    };
    const int cl = lane & 15, q = lane >> 4;
    double sh[C], keep[C];
#pragma unroll
    for (int b = 0; b < C; b++) {
        const int c = 16 * b + cl;
        const bool real = c < M + P;
        keep[b] = real ? 1.0 : 0.0;
        sh[b] = real ? shift[c] : 0.0;
    }
    d4 acc[D::NBLK];
#pragma unroll
    for (int b = 0; b < D::NBLK; b++) acc[b] = (d4){0.0, 0.0, 0.0, 0.0};
    double cs[C], cq[C];
#pragma unroll
    for (int b = 0; b < C; b++) { cs[b] = 0.0; cq[b] = 0.0; }
    const int poff = 2 * cl + (q & 1) + 32 * (q >> 1);
    int cur = 0;
    long long tile = g;
#pragma unroll
    for (int d = 0; d < R - 1; d++)
        if (tile + d * G < ntiles) stage(tile + d * G, d);
    for (; tile < ntiles; tile += G) {
        const double* buf = wring + cur * CH;
        {   // this tile's DMA has landed; up to R - 2 later ones (C instructions each) may stay in flight
            const long long ahead = (ntiles - 1 - tile) / G;
            if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (tile + (R - 1) * G < ntiles) stage(tile + (R - 1) * G, (cur + R - 1) % R);
        const long long row0 = t0 + tile * TRK;
        const bool full = (row0 >= r_begin) && (row0 + TRK <= r_end);
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int rb = wave * 8 + 4 * s + q;
            const long long grow = row0 + rb;
            const bool ok = full || (grow >= r_begin && grow < r_end);
            double a[C];
#pragma unroll
            for (int b = 0; b < C; b++) {
                const double v = buf[b * 128 + poff + 64 * s];
                const double z = (v - sh[b]) * keep[b];
                a[b] = ok ? z : 0.0;
                cs[b] += a[b];
                if (b >= C - CY) cq[b] = fma(a[b], a[b], cq[b]);
            }
            int blk = 0;
#pragma unroll
            for (int bi = 0; bi < C - CY; bi++)
#pragma unroll
                for (int bj = bi; bj < C; bj++) {
                    acc[blk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[bi], a[bj], acc[blk], 0, 0, 0);
                    blk++;
                }
        }
        cur = (cur + 1) % R;
    }

    // ---- epilogue (same partial record as k_gram / k_gram_dma) ------------------------------------------
    double* out = partial + ((size_t)part * G + g) * D::PSZ;
    __syncthreads();
    constexpr int EPT = (D::NBLK * 256 + NT - 1) / NT;
    double tot[EPT];
#pragma unroll
    for (int k = 0; k < EPT; k++) tot[k] = 0.0;
    constexpr int EW = ((size_t)4 * D::NBLK * 256 * sizeof(double) <= 160 * 1024) ? 4 : 2;
    for (int round = 0; round < NW / EW; round++) {
        if ((wave / EW) == round) {
#pragma unroll
            for (int b = 0; b < D::NBLK; b++)
#pragma unroll
                for (int r = 0; r < 4; r++) lds[(wave % EW) * (D::NBLK * 256) + b * 256 + r * 64 + lane] = acc[b][r];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < EPT; k++) {
            const int e = t + k * NT;
            if (e < D::NBLK * 256) {
                if constexpr (EW == 4)
                    tot[k] += ((lds[e] + lds[D::NBLK * 256 + e]) + lds[2 * D::NBLK * 256 + e]) + lds[3 * D::NBLK * 256 + e];
                else
                    tot[k] += lds[e] + lds[D::NBLK * 256 + e];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < EPT; k++) {
        const int e = t + k * NT;
        if (e < D::NBLK * 256) out[e] = tot[k];
    }
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int b = 0; b < C; b++) lds[b * NT + t] = half ? cq[b] : cs[b];
        __syncthreads();
        if (t < D::C16) {
            const int b = t >> 4, c = t & 15;
            double sacc = 0.0;
            for (int w = 0; w < NW; w++)
                for (int qq = 0; qq < 4; qq++) sacc += lds[b * NT + w * 64 + qq * 16 + c];
            out[D::NBLK * 256 + half * D::C16 + t] = sacc;
        }
    }
}


// ---------------------------------------------------------------------------------------------------
// Wide sets in ONE launch (8..10 column blocks, e.g. BASELINE configs[4]: 128 metrics + 16 parameters = 9 blocks, 44 Gram
// blocks): the accumulators of all blocks do not fit one wave (44 x 8 VGPRs), so the BLOCKS are dealt out to the eight waves
// of a work-group (block b to wave b mod 8, five or six each) and every wave runs its blocks over all 64 rows of the tile --
// each block has one owner, so the epilogue needs no cross-wave reduction.  Staging as in k_gram (16-byte non-temporal loads a
// tile ahead, z-shifted and masked on the way into LDS, column sums / sums of squares there).  Every column is read once
// (the grouped path read each 48-column group twice, in three launches: 0.17 ms at N = 125 k against 0.07 here);
// 44 fp64 MFMA blocks per 4 rows make this shape matrix-pipe bound.
template <int C, int CY>
struct GramWide {
    static constexpr int C16 = 16 * C;
    static constexpr int NBLK = C * (C + 1) / 2 - CY * (CY + 1) / 2;
    static constexpr int NW = 8, NT = 512, TRW = 64, TRPW = TRW + 2;
    static constexpr int NI = C;                          // one column of every 16-column block per thread and tile
    static constexpr int PSZ = NBLK * 256 + 2 * C16;      // same partial record as k_gram
    static constexpr int NBW = (NBLK + NW - 1) / NW;      // blocks per wave
    static constexpr int LDS_D = C16 * TRPW;
};
template <int C, int CY>
__global__ __launch_bounds__(512) void k_gram_wide(const double* __restrict__ X, const double* __restrict__ Y, size_t ldx, size_t ldy,
                                                   int M, int P, long long n, long long split, const double* __restrict__ shift,
                                                   double* __restrict__ partial, int vec_ok) {
    using D = GramWide<C, CY>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int part = blockIdx.y, G = gridDim.x, g = blockIdx.x;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + D::TRW - 1) / D::TRW : 0;
    // this wave's blocks: the b-th block of the (bi, bj >= bi, bi < C - CY) enumeration with b = wave + 8 k
    int wbi[D::NBW], wbj[D::NBW];          // (indexed by unrolled constants only: they live in SGPRs)
#pragma unroll
    for (int k = 0; k < D::NBW; k++) {
        int rem = wave + 8 * k, bi = 0;
        while (bi < C - CY && rem >= C - bi) { rem -= C - bi; bi++; }
        wbi[k] = (bi < C - CY) ? bi : -1;
        wbj[k] = bi + rem;
    }
    d4 acc[D::NBW];
#pragma unroll
    for (int k = 0; k < D::NBW; k++) acc[k] = (d4){0.0, 0.0, 0.0, 0.0};
    // staging: thread -> (column 16 i + t / 32, row pair t % 32)
    const int cq = t >> 5, rp = t & 31;
    double colsum[D::NI], colsq[CY > 0 ? CY : 1], sh[D::NI], keep[D::NI];     // sums of squares: the trailing CY blocks only
    const double* cptr[D::NI];
#pragma unroll
    for (int q0 = 0; q0 < (CY > 0 ? CY : 1); q0++) colsq[q0] = 0.0;
#pragma unroll
    for (int i = 0; i < D::NI; i++) {
        const int c = 16 * i + cq;
        colsum[i] = 0.0;
        keep[i] = (c < M + P) ? 1.0 : 0.0;
        cptr[i] = (c < M) ? X + (size_t)c * ldx : (c < M + P) ? Y + (size_t)(c - M) * ldy : X;
        sh[i] = (c < M + P) ? shift[c] : 0.0;
    }
    d2 v[D::NI];
    auto fetch = [&](long long tile) {
        const long long row0 = t0 + tile * D::TRW;
        const long long r = row0 + 2 * rp;
        const bool full = (row0 >= r_begin) && (row0 + D::TRW <= r_end);
        if (full && vec_ok) {
#pragma unroll
            for (int i = 0; i < D::NI; i++) v[i] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(cptr[i] + r));
        } else {
            const long long ra = r < r_begin ? r_begin : (r >= r_end ? r_end - 1 : r);
            const long long rb = r + 1 < r_begin ? r_begin : (r + 1 >= r_end ? r_end - 1 : r + 1);
            const bool oka = (r >= r_begin) && (r < r_end), okb = (r + 1 >= r_begin) && (r + 1 < r_end);
#pragma unroll
            for (int i = 0; i < D::NI; i++) {
                const double xa = cptr[i][ra], xb = cptr[i][rb];
                v[i] = (d2){oka ? xa : sh[i], okb ? xb : sh[i]};
            }
        }
    };
    long long tile = g;
    if (tile < ntiles) fetch(tile);
    const int cl = lane & 15, q = lane >> 4;
    for (; tile < ntiles; tile += G) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < D::NI; i++) {
            const int c = 16 * i + cq;
            d2 z = (d2){(v[i].x - sh[i]) * keep[i], (v[i].y - sh[i]) * keep[i]};
            colsum[i] += z.x + z.y;
            if constexpr (CY > 0) {
                if (i >= C - CY) { colsq[i - (C - CY)] = fma(z.x, z.x, colsq[i - (C - CY)]); colsq[i - (C - CY)] = fma(z.y, z.y, colsq[i - (C - CY)]); }
            }
            *reinterpret_cast<d2*>(&lds[c * D::TRPW + 2 * rp]) = z;
        }
        __syncthreads();
        if (tile + G < ntiles) fetch(tile + G);
#pragma unroll 2
        for (int s = 0; s < D::TRW / 4; s++) {
            const int rb = 4 * s + q;
#pragma unroll
            for (int k = 0; k < D::NBW; k++) {
                if (wbi[k] >= 0) {          // wave-uniform
                    const double a = lds[(16 * wbi[k] + cl) * D::TRPW + rb], b = lds[(16 * wbj[k] + cl) * D::TRPW + rb];
                    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
                }
            }
        }
    }
    // epilogue: every block has exactly one owner
    double* out = partial + ((size_t)part * G + g) * D::PSZ;
#pragma unroll
    for (int k = 0; k < D::NBW; k++) {
        if (wbi[k] >= 0) {
            const int b = wave + 8 * k;
#pragma unroll
            for (int r = 0; r < 4; r++) out[b * 256 + r * 64 + lane] = acc[k][r];
        }
    }
    // column sums / sums of squares: 32 threads staged each column
    for (int half = 0; half < 2; half++) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < D::NI; i++) lds[(16 * i + cq) * 32 + rp] = half ? ((CY > 0 && i >= C - CY) ? colsq[i >= C - CY ? i - (C - CY) : 0] : 0.0) : colsum[i];
        __syncthreads();
        if (t < D::C16) {
            double s = 0.0;
            for (int l = 0; l < 32; l++) s += lds[t * 32 + l];
            out[D::NBLK * 256 + half * D::C16 + t] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Round 4: the Gram of wide sets OFF the fp64 matrix pipe.  k_gram_wide is bound by v_mfma_f64_16x16x4_f64 (44 blocks x 64 cycles
// per 4 rows: 0.61 ms at 1e6 rows x 144 columns, 0.235 of the HBM roofline).  Here every shifted value travels as a 32-bit
// FIXED-POINT number on a per-column grid -- q = rint(v 2^(31 - e_c)), |v| < 2^e_c -- cut into four balanced signed bytes
// (q = sum_b l_b 2^(8b), -128 <= l_b <= 127), and the cross products run as exact integer arithmetic on
// v_mfma_i32_32x32x32_i8 (the 16-bit pipe's rate at twice the depth): G_ab 2^(62 - e_a - e_b) = sum_rows q_a q_b =
// sum_{b,b'} 2^(8(b+b')) sum_rows l_b l'_b', of which the thirteen byte pairs with b + b' >= 2 are kept -- one i32 accumulator per
// order b + b' (exact: <= 2^21 per 32-row step, flushed to fp64 every 512 steps), 13 MFMAs per 32 x 32 tile and step, 14 tiles at
// 144 columns: 182 matrix-pipe cycles per row (14 x 13 MFMAs of 32 cycles per 32 rows) against 704 on the fp64 pipe.  The dropped pairs (b + b' <= 1) are products of the low bytes:
// zero-mean noise below 1e-11 of sigma_a sigma_b rows off the diagonal (with b + b' = 2 dropped as well it was 3e-9, and the 32nd
// loading of a 128-metric model moved by 4e-6) -- and a BIAS on the diagonal (l l' >= 0 there), so column sums and sums of squares are taken in fp64 on the
// vector pipe while the tile is staged (exact as before: means to 1e-12, deviations to 1e-11) and the diagonal of the result is
// theirs.  Conversion is three instructions per value: v + 1.5 2^(21 + e_c) leaves q in the low word of the double, and
// (q + 0x80808080) ^ 0x80808080 are its four balanced bytes.
// A tile is one 32-row step; the eight waves of a work-group work in two roles (GramI8 below).
// Range: e_c = ceil(log2(4 max |x - shift|)) over 4096 rows spread over the set (k_pilot_scale).  A row with a value outside its
// column's range ("far": a spike 4 times beyond anything in the sample) is left out of the byte products as a whole -- its bytes
// are zeroed in LDS -- and listed; k_gram_far adds the far rows' products in fp64.
// Partial records in k_gram's format (k_stats_reduce unchanged): a 32 x 32 tile is written as its 16 x 16 blocks.
template <int C, int CY>
struct GramI8 {
    static constexpr int C16 = 16 * C;
    static constexpr int CB = (C + 1) / 2;                       // 32-column super-blocks
    static constexpr int C32 = 32 * CB;
    static constexpr int NBLK = C * (C + 1) / 2 - CY * (CY + 1) / 2;
    static constexpr int PSZ = NBLK * 256 + 2 * C16;             // the partial record of k_gram
    static constexpr int NT = 512, NW = 8, TRW = 32;             // a tile = one 32-row step of the i8 MFMA
    static constexpr int NR = (C32 + 63) / 64;                   // conversion rounds per tile: a thread takes (column 64 r + t / 8, rows 4 (t % 8) .. + 3)
    static constexpr int DPW = (C32 / 4 + NW - 1) / NW;          // LDS-DMA instructions (1 KB each: 4 columns x 16 row pairs) per wave and tile, at most
    static constexpr int CS = 32;                                // bytes of a (byte plane, column) in LDS: 32 rows, the two 16-byte halves swapped
    //                                                              in columns 4..7 mod 8 (b128 operand reads and the conversion's b32 stores conflict-free)
    static constexpr int PLANE = C32 * CS;
    static constexpr bool skip_last = (2 * (CB - 1) >= C - CY);  // the last diagonal tile holds Y'Y / padding only
    static constexpr int NTILE = CB * (CB + 1) / 2 - (skip_last ? 1 : 0);
    static constexpr int TPW = (NTILE + NW - 1) / NW;            // tiles per wave
    // two or three raw tiles (LDS-DMA targets: the real columns only, [column][16 row pairs]), TWO sets of byte planes, a 16-byte
    // record per column (shift, high words of magic and limit), three sets of far flags (32 rows + any); the running column sums
    // and sums of squares live in registers (a thread's (column, four rows) slots are the same in every tile) and pass through
    // the raw tiles' space once, at the end
    static constexpr int FIXED_B = 8 * PLANE + C32 * 16 + 3 * 48 + 256;         // (+ 256 zero bytes: the "raw column" of the padding columns)
    static constexpr int LDS_MAX = 160 * 1024;
    static int raw_bytes(int cols) { return ((cols + 3) / 4) * 1024; }
    static int raw_tiles(int cols) { return FIXED_B + 3 * raw_bytes(cols) <= LDS_MAX ? 3 : 2; }          // (two: 157..160 columns)
    static int lds_bytes(int cols) { return FIXED_B + raw_tiles(cols) * raw_bytes(cols); }
    static_assert(2 * C32 * 8 * 8 <= 2 * ((C16 - 15 + 3) / 4) * 1024, "the final sums fit the raw tiles");
    static constexpr int FLUSH = 512;                            // tiles (32-row steps) between two flushes of the i32 accumulators
};
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// e[c]: the binade of the column's fixed-point range = 4 x a ROBUST size of |x - shift| among 4096 rows spread evenly over the n rows:
// the median of the 64 maxima of 64 samples each (for a Gaussian column 4 x 2.4 sigma, rounded up to a power of two: 10 .. 19 sigma; a spike in the sample moves one of the 64
// maxima, not their median -- with the plain maximum one sampled outlier of 1e4 sigma cost the column thirteen bits of resolution)
__global__ __launch_bounds__(1024) void k_pilot_scale(const double* __restrict__ X, const double* __restrict__ Y, size_t ldx, size_t ldy,
                                                      int M, int P, long long n, const double* __restrict__ shift, int* __restrict__ e) {
    const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ double gmax[64];
    const bool real = c < M + P;
    const double* p = (c < M) ? X + (size_t)c * ldx : Y + (size_t)((real ? c : M) - M) * ldy;
    const double sh = real ? shift[c] : 0.0;
    const long long S = n < 4096 ? n : 4096;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const long long q = (long long)t + 1024 * j;
        double a = 0.0;
        if (real && q < S) {
            const long long r = (long long)(((unsigned long long)q * (unsigned long long)n) / (unsigned long long)S);
            a = fabs(p[r] - sh);
            a = (a == a) ? a : 0.0;                                          // (a NaN never counts: such a row is far in the main pass)
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const double b = __shfl_xor(a, o, 64); a = b > a ? b : a; }
        if (lane == 0) gmax[wave * 4 + j] = a;
    }
    __syncthreads();
    if (wave == 0) {
        const double mine = gmax[lane];
        int below = 0;
        for (int i = 0; i < 64; i++) { const double o = gmax[i]; below += (o < mine || (o == mine && i < lane)) ? 1 : 0; }
        const unsigned long long pick = __ballot(below == 32);                // the 33rd smallest of the 64 (ties broken by position)
        const double med = __shfl(mine, __ffsll((long long)pick) - 1, 64);
        if (lane == 0) {
            int ex = -1000;
            if (med > 0.0 && med < 1.0e300) { (void)frexp(med, &ex); ex += 2; }      // med < 2^(ex - 2): range 2^ex >= 4 med
            e[c] = ex < -1000 ? -1000 : ex;
        }
    }
}

// Staging by LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without a register round trip), two raw tiles in flight behind the one
// being converted (HBM at 6 TB/s with ~2.5 us of latency wants ~60 KB per CU in flight).  What was measured on the way
// (1e6 rows x 144 columns): the next tile prefetched in registers, one tile ahead -- 336 us, the kernel ran at the latency of its
// loads (41 KB in flight per CU), and the ~250 registers a thread then needs spill, each scratch reload being a vmcnt wait that also
// waits for the prefetch; three converter waves feeding five multiplier waves (roles as separate code paths) -- 450 us with one tile
// ahead, 930 us with two (the compiler spilled the converters' second tile).  Round 4's conversion took a (column, row PAIR) per
// thread and step, five columns one after the other with the per-column constants and the running sums re-read from LDS each time
// and eight 2-byte plane stores per four values: twelve LDS instructions per step, 417 us.  Round 5, first: a thread converts FOUR
// rows of a column (the DMA lanes fetch the even row pairs into the first half of a column's 256 bytes and the odd ones into the
// second, so a thread's two 16-byte reads are 16-byte strided across the lanes: conflict-free), the 4 x 4 byte transpose is eight
// v_perm_b32, every plane store one ds_write_b32; the sums stay in registers: 360 us -- and the phases taken out one at a time
// (ABC_GRAM_ABL) said why: the refills alone 222 us (5.2 TB/s, what this access pattern gets), conversion + products WITHOUT
// refills 290 us, i.e. ~5000 cycles per tile of which the byte products own 1664 (four tile pairs x 13 MFMAs x 32 cycles on the
// busier SIMDs) and the conversion ~2500 (two waves of ~260 vector instructions per SIMD), one after the other between barriers.
// Round 5, second: the two run TOGETHER.  Two sets of byte planes; in the step of tile i the four waves 0..3 (one per SIMD) multiply
// tile i and then convert tile i + 1 into the other set, the waves 4..7 do the same in the opposite order -- a SIMD's two waves
// keep its matrix pipe and its vector pipe busy at the same time -- and ONE barrier per tile separates the steps.
// Per step: wait for the own DMAs of tile i + 1, barrier, refill tile i's raw slot with tile i + 3, tile i's far rows, the MFMAs
// of tile i / the conversion of tile i + 1 (raw -> byte planes, column sums and squares in fp64, far flags).
template <int C, int CY>
__global__ __launch_bounds__(512) void k_gram_i8(const double* __restrict__ X, const double* __restrict__ Y, size_t ldx, size_t ldy, int M,
                                                 int P, long long n, long long split, const double* __restrict__ shift,
                                                 const int* __restrict__ escale, double* __restrict__ partial,
                                                 unsigned long long* __restrict__ far_mask /* [2][tmax]: far rows of every 32-row tile */,
                                                 unsigned long long* __restrict__ far_sum /* [2][(tmax + 63) / 64]: tiles with any; then one word: any at all */,
                                                 long long tmax, int nraw /* raw tiles in LDS: 3, or 2 */,
                                                 int abl /* diagnostic (ABC_GRAM_ABL): 1 no conversion, 2 no products, 4 no refills */) {
    using D = GramI8<C, CY>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    const int ncols = M + P, ndma = (ncols + 3) / 4, rawb = ndma * 1024;
    unsigned char* planes0 = lds8;                                            // [2][4][C32][CS]
    double* colc = reinterpret_cast<double*>(lds8 + 8 * D::PLANE);            // [C32][2]: shift; high words of magic and limit
    unsigned char* rowfar0 = reinterpret_cast<unsigned char*>(colc + 2 * D::C32);           // [3][48]: 32 row flags + the any-flag word
    unsigned char* zero256 = rowfar0 + 3 * 48;                                // what the conversion reads for a padding column
    unsigned char* raw0 = zero256 + 256;                                      // [nraw][ncols up to 4][16 row pairs: 0, 2, .. 14, 1, 3, .. 15] doubles
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int part = blockIdx.y, G = gridDim.x, g = blockIdx.x;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + D::TRW - 1) / D::TRW : 0;
    const int nmine = (ntiles > g) ? (int)((ntiles - g + G - 1) / G) : 0;     // this work-group's tiles: g, g + G, ...  (tmax < 2^31)
    const long long rmax = (n - 2) & ~1LL;                                    // last in-bounds 16-B row pair (n is even on this path)
    double* out = partial + ((size_t)part * (G + 1) + g) * D::PSZ;           // (record G of a partition: k_gram_far's)
    auto blk_index = [&](int bi, int bj) -> int {                               // (bi, bj >= bi, bi < C - CY) enumeration of k_gram
        int b = 0;
        for (int i2 = 0; i2 < bi; i2++) b += C - i2;
        return b + (bj - bi);
    };
    // conversion: thread -> (column 64 r + t / 8, rows 4 (t % 8) .. + 3) in round r.  Per-column constants: shift, magic = 1.5 2^(21 + e)
    // (v + magic has q = rint(v 2^(31 - e)) in its low word), and the range limit (1 - 2^-6) 2^e (so that the balanced bytes of q
    // never carry out of the top one): the two from the binade e by integer arithmetic on the exponent field (their low words are 0)
    const int cq = t >> 3, qd = t & 7;
    double ssum[D::NR], ssq[D::NR];
#pragma unroll
    for (int r = 0; r < D::NR; r++) { ssum[r] = 0.0; ssq[r] = 0.0; }
    for (int c = t; c < D::C32; c += D::NT) {
        const bool real = c < ncols;
        const int e = real ? escale[c] : 0;
        colc[2 * c] = real ? shift[c] : 0.0;
        reinterpret_cast<int*>(colc + 2 * c + 1)[0] = (int)(0x3ff80000u + ((unsigned)(21 + e) << 20));      // 1.5 2^(21 + e)
        reinterpret_cast<int*>(colc + 2 * c + 1)[1] = (int)(0x3fef8000u + ((unsigned)e << 20));             // 0.984375 2^e
    }
    for (int e2 = t; e2 < 8 * D::PLANE / 16; e2 += D::NT) reinterpret_cast<v4i*>(planes0)[e2] = v4i{0, 0, 0, 0};   // (the padding columns' bytes stay 0)
    if (t < 36 + 64) reinterpret_cast<unsigned int*>(rowfar0)[t] = 0u;       // the three sets of far flags, the zero column
    // (the record is not zeroed: every element of it has one owner -- a lane of the wave that holds its tile pair, or the diagonal's
    // thread at the end -- whose FIRST flush stores and whose later ones add)
    // this wave's tiles: the b-th (I, J >= I) super-block pair in row-major order, b = wave + 8 k
    int wI[D::TPW], wJ[D::TPW];
#pragma unroll
    for (int k = 0; k < D::TPW; k++) {
        int rem = wave + 8 * k, I = 0;
        while (I < D::CB && rem >= D::CB - I) { rem -= D::CB - I; I++; }
        const bool ok = I < D::CB && !(D::skip_last && I == D::CB - 1);
        wI[k] = ok ? I : -1;
        wJ[k] = I + rem;
    }
    v16i acc[D::TPW][5];                                                      // orders b + b' = 2 .. 6
#pragma unroll
    for (int k = 0; k < D::TPW; k++)
#pragma unroll
        for (int o = 0; o < 5; o++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[k][o][r] = 0;
    // LDS-DMA: instruction d of a tile moves columns 4 d .. 4 d + 3; lane = 16 (column & 3) + p fetches row pair 2 p (p < 8) resp.
    // 2 (p - 8) + 1: rows 4 q .. 4 q + 3 of a column sit at its bytes 16 q and 128 + 16 q.  Wave w issues d = w, w + 8, ... < ndma:
    // my_dma instructions per tile (the counted waits below go by it)
    const int dma_rp = ((lane & 7) << 1) | ((lane >> 3) & 1);
    const int my_dma = (ndma > wave) ? (ndma - wave + D::NW - 1) / D::NW : 0;
    // (the slots and flag sets of the tiles are counted along by the loop: k % nraw with nraw a run-time number is a division
    // sequence, and the first version of this loop spent a third of a step in such scalar arithmetic)
    auto stage = [&](int k, int slot) {                                         // the k-th tile of this work-group -> raw slot k % nraw
        const long long r = t0 + ((long long)g + (long long)k * G) * D::TRW + 2 * dma_rp;
        const long long rr = r > rmax ? rmax : r;                               // (a legal address for rows past the end: masked later)
        unsigned char* dst = raw0 + slot * rawb;
#pragma unroll
        for (int j = 0; j < D::DPW; j++) {
            const int d = wave + D::NW * j, c = 4 * d + (lane >> 4);
            if (d < ndma) {
                const double* p = (c < M) ? X + (size_t)c * ldx : Y + (size_t)((c < ncols ? c : M) - M) * ldy;    // (a column past the last: fetched, never read)
                dma16(p + rr, reinterpret_cast<double*>(dst + (size_t)d * 1024));
            }
        }
    };
    // at most `tiles` of this wave's staged tiles still in flight (the immediates: my_dma is 3, 4 or 5 for 113 .. 160 columns)
    auto wait_in_flight = [&](int tiles) {
        const int nout = tiles * my_dma;
        if (nout == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (nout == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (nout == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (nout == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (nout == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (nout == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (nout == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto convert = [&](int k, int slot, int fset) {                             // raw slot k % nraw -> byte planes k & 1, far flags k % 3
        const unsigned char* raw = raw0 + slot * rawb;
        unsigned char* planes = planes0 + (k & 1) * 4 * D::PLANE;
        unsigned char* rowfar = rowfar0 + fset * 48;
        const long long tile = (long long)g + (long long)k * G, r = t0 + tile * D::TRW + 4 * qd;
        const bool edge = tile == 0 || tile == ntiles - 1;                      // (uniform) only a partition's first and last tile hold rows outside it
        bool far[4] = {false, false, false, false};
        // (straight-line code over the rounds: a padding column reads zeros against a zero shift and stores zero bytes like any other,
        // so the rounds' LDS reads can all be in flight before the first value is needed -- with a branch around each round a wave
        // paid the LDS latency three times per tile)
#pragma unroll
        for (int i = 0; i < D::NR; i++) {
            if (64 * i + 8 * wave >= D::C32) continue;                         // (scalar: the last round of a 160-column set has 32 columns)
            const int c = 64 * i + cq;
            {
                const unsigned char* src = (c < ncols) ? raw + c * 256 : zero256;
                const d2 va = *reinterpret_cast<const d2*>(src + 16 * qd);
                const d2 vb = *reinterpret_cast<const d2*>(src + 128 + 16 * qd);
                const v4i cc = *reinterpret_cast<const v4i*>(colc + 2 * c);
                const double sh = __hiloint2double(cc[1], cc[0]), magic = __hiloint2double(cc[2], 0), lim = __hiloint2double(cc[3], 0);
                double z[4] = {va.x - sh, va.y - sh, vb.x - sh, vb.y - sh};
                if (edge) {
#pragma unroll
                    for (int j = 0; j < 4; j++) z[j] = (r + j >= r_begin && r + j < r_end) ? z[j] : 0.0;      // (rows outside the partition: 0)
                }
                unsigned int bq[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    far[j] = far[j] || !(fabs(z[j]) <= lim);                   // (NaN: not in range)
                    const unsigned int q = (unsigned int)__double_as_longlong(z[j] + magic);
                    bq[j] = (q + 0x80808080u) ^ 0x80808080u;
                }
                ssum[i] += (z[0] + z[1]) + (z[2] + z[3]);
                ssq[i] = fma(z[3], z[3], fma(z[2], z[2], fma(z[1], z[1], fma(z[0], z[0], ssq[i]))));
                // 4 x 4 byte transpose: plane b gets byte b of the four rows' words (v_perm_b32: selector bytes 0-3 = the second operand)
                const unsigned int l01 = __builtin_amdgcn_perm(bq[1], bq[0], 0x05010400u), h01 = __builtin_amdgcn_perm(bq[1], bq[0], 0x07030602u);
                const unsigned int l23 = __builtin_amdgcn_perm(bq[3], bq[2], 0x05010400u), h23 = __builtin_amdgcn_perm(bq[3], bq[2], 0x07030602u);
                unsigned char* dst = planes + (size_t)c * D::CS + 16 * ((qd >> 2) ^ ((c >> 2) & 1)) + 4 * (qd & 3);
                *reinterpret_cast<unsigned int*>(dst) = __builtin_amdgcn_perm(l23, l01, 0x05040100u);
                *reinterpret_cast<unsigned int*>(dst + (size_t)D::PLANE) = __builtin_amdgcn_perm(l23, l01, 0x07060302u);
                *reinterpret_cast<unsigned int*>(dst + (size_t)2 * D::PLANE) = __builtin_amdgcn_perm(h23, h01, 0x05040100u);
                *reinterpret_cast<unsigned int*>(dst + (size_t)3 * D::PLANE) = __builtin_amdgcn_perm(h23, h01, 0x07060302u);
            }
        }
        if (far[0] || far[1] || far[2] || far[3]) {
#pragma unroll
            for (int j = 0; j < 4; j++) if (far[j]) rowfar[4 * qd + j] = 1;
            rowfar[32] = 1;
        }
    };
    // the byte products of a tile from plane set `buf`: lane -> column 32 I + lane % 32, rows 16 (lane / 32) .. + 15 of every plane
    auto products = [&](int buf) {
        const unsigned char* planes = planes0 + (size_t)buf * 4 * D::PLANE;
        const int hoff = 16 * ((lane >> 5) ^ ((lane >> 2) & 1));
#pragma unroll
        for (int k = 0; k < D::TPW; k++) {
            if (wI[k] < 0) continue;                                           // (wave-uniform)
            v4i fa[4];
            const unsigned char* pa = planes + (size_t)(32 * wI[k] + (lane & 31)) * D::CS + hoff;
            const unsigned char* pb = planes + (size_t)(32 * wJ[k] + (lane & 31)) * D::CS + hoff;
#pragma unroll
            for (int b = 0; b < 4; b++) fa[b] = *reinterpret_cast<const v4i*>(pa + (size_t)b * D::PLANE);
            // order o = b + b' - 2: (2,0)(1,1)(0,2) | (3,0)(2,1)(1,2)(0,3) | (3,1)(2,2)(1,3) | (3,2)(2,3) | (3,3)
#pragma unroll
            for (int b2 = 0; b2 < 4; b2++) {
                const v4i fb = *reinterpret_cast<const v4i*>(pb + (size_t)b2 * D::PLANE);
#pragma unroll
                for (int b = 0; b < 4; b++)
                    if (b + b2 >= 2) acc[k][b + b2 - 2] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[b], fb, acc[k][b + b2 - 2], 0, 0, 0);
            }
        }
    };
    // flush of the i32 accumulators into this work-group's partial record (fp64): tile (I, J), order o = b + b' - 3 carries weight
    // 2^(8 o) 2^(e_a + e_b - 62 + 24); a 32 x 32 tile is up to four 16 x 16 blocks of the record, stored in the f64 MFMA's C layout
    bool flushed = false;
    auto flush = [&]() {
#pragma unroll
        for (int k = 0; k < D::TPW; k++) {
            if (wI[k] < 0) continue;
            const int I = wI[k], J = wJ[k];
            const int ncol = 32 * J + (lane & 31);                            // this lane's column of the tile
            const int eb = (ncol < ncols) ? escale[ncol] : 0;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int mrow = 32 * I + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int ea = (mrow < ncols) ? escale[mrow] : 0;
                const double val = ldexp(((double)acc[k][4][r] * 4294967296.0 + (double)acc[k][3][r] * 16777216.0) +
                                         ((double)acc[k][2][r] * 65536.0 + ((double)acc[k][1][r] * 256.0 + (double)acc[k][0][r])), ea + eb - 62 + 16);
                const int bi = mrow >> 4, bj = ncol >> 4;
                if (bi > bj || bi >= C - CY || bj >= C || mrow == ncol) continue;   // lower triangle, pure Y'Y / padding, the diagonal
                // C/D layout of v_mfma_f64_16x16x4_f64: element (row, col) of a block sits at lane' = 16 (row & 3) + col, register row >> 2
                const int rr = mrow & 15, cc = ncol & 15;
                double* po = out + (size_t)blk_index(bi, bj) * 256 + (rr >> 2) * 64 + 16 * (rr & 3) + cc;     // (one owner per element)
                *po = flushed ? *po + val : val;
            }
#pragma unroll
            for (int o = 0; o < 5; o++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[k][o][r] = 0;
        }
        flushed = true;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // (the record's loads and stores out of the DMA count)
    };
    __syncthreads();
    const int npre = nmine < nraw ? nmine : nraw;
    for (int k = 0; k < npre; k++) stage(k, k);
    if (nmine > 0) {                                                           // tile 0 -> plane set 0
        wait_in_flight(npre - 1);
        __syncthreads();
        if (!(abl & 1)) convert(0, 0, 0);
    }
    int slot_i = 0, fset_i = 0;                                                // i % nraw, i % 3
    // (two loops: the accumulators are flushed between runs of FLUSH tiles -- the flush stays out of the inner loop's register budget)
    for (int i0 = 0; i0 < nmine; i0 += D::FLUSH) {
        const int i1 = (i0 + D::FLUSH < nmine) ? i0 + D::FLUSH : nmine;
#pragma unroll 1
        for (int i = i0; i < i1; i++) {
            const int slot_n = (slot_i + 1 == nraw) ? 0 : slot_i + 1, fset_n = (fset_i == 2) ? 0 : fset_i + 1;     // tile i + 1's
            // (1) the own DMAs of tile i + 1 have landed (the staged tiles behind it may stay in flight), then everybody's; the barrier
            //     also says: tile i's planes and far flags are complete, every wave is done with the MFMAs of tile i - 1 (the other
            //     plane set is free) and with tile i's raw slot
            if (nraw == 3 && i + 2 < nmine) {                                   // (the steady state, without the general case's chain of compares)
                if (my_dma == D::DPW) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::DPW) : "memory");
                else if (my_dma == D::DPW - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D::DPW - 1) : "memory");
                else wait_in_flight(1);
            } else {
                const int last = (i + nraw - 1 < nmine - 1) ? i + nraw - 1 : nmine - 1;     // last tile staged so far
                wait_in_flight(last > i + 1 ? last - (i + 1) : 0);
            }
            __syncthreads();
            const unsigned char any_far = rowfar0[fset_i * 48 + 32];           // (read here, needed behind the refill)
            // (2) tile i's far rows; refill tile i's raw slot with tile i + nraw.  (In this order: the mask's store counts in vmcnt like the
            //     DMAs, and issued BEHIND the refill it made wave 0's next counted wait -- "all but the newest tile's" -- wait for one
            //     DMA of the tile just requested, a full memory latency in every step with seven waves waiting at the barrier: rounds 4's
            //     order, and most of what kept that kernel from overlapping its refills with its arithmetic)
            unsigned char* rowfar = rowfar0 + fset_i * 48;
            const long long tile = (long long)g + (long long)i * G, row0 = t0 + tile * D::TRW;
            if (wave == D::NW - 1) {                                           // tile i's far rows as a mask (always written: no memset); the wave with the least else to do
                const unsigned long long m = __ballot(lane < 32 && rowfar[lane & 31] != 0 && row0 + lane >= r_begin && row0 + lane < r_end);
                if (lane == 0) {
                    far_mask[(size_t)part * tmax + tile] = m;
                    if (m) {
                        atomicOr(&far_sum[(size_t)part * ((tmax + 63) / 64) + (tile >> 6)], 1ull << (tile & 63));
                        far_sum[(size_t)2 * ((tmax + 63) / 64)] = 1ull;        // (k_gram_far's way out when nothing is far)
                    }
                }
                // the set of flags tile i + 2 will use: its last readers (tile i - 1) passed this step's barrier, its writers (the
                // conversion of tile i + 2, next step) have to pass the next one
                if (lane < 12) reinterpret_cast<unsigned int*>(rowfar0 + ((fset_n == 2) ? 0 : fset_n + 1) * 48)[lane] = 0u;
            }
            if (i + nraw < nmine && !(abl & 4)) stage(i + nraw, slot_i);
            if (any_far) {                                                     // (uniform) rare: zero the far rows' bytes
                unsigned char* planes = planes0 + (i & 1) * 4 * D::PLANE;
                for (int e2 = t; e2 < 4 * D::C32 * 32; e2 += D::NT) {
                    const int r = e2 & 31, c = (e2 >> 5) % D::C32, b = e2 / (32 * D::C32);
                    if (rowfar[r]) planes[(size_t)b * D::PLANE + (size_t)c * D::CS + 16 * ((r >> 4) ^ ((c >> 2) & 1)) + (r & 15)] = 0;
                }
                __syncthreads();
            }
            // (3) the byte products of tile i and the conversion of tile i + 1, in opposite orders on the two waves of a SIMD
            const bool conv = i + 1 < nmine && !(abl & 1);
            if (conv && wave >= 4) convert(i + 1, slot_n, fset_n);
            if (!(abl & 2)) products(i & 1);                                   // (one call site: the accumulators stay in their registers)
            if (conv && wave < 4) convert(i + 1, slot_n, fset_n);
            slot_i = slot_n; fset_i = fset_n;
        }
        if (i1 < nmine) flush();                                               // (rare: more than FLUSH tiles in this work-group)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    flush();
    // column sums / sums of squares (eight slots per column, through the raw tiles' space, added in a fixed order) and the exact diagonal
    double* lsum = reinterpret_cast<double*>(raw0);                                          // [C32][8], then the same of the squares
    double* lsq = lsum + D::C32 * 8;
#pragma unroll
    for (int r = 0; r < D::NR; r++) {
        const int c = 64 * r + cq;
        if (c < D::C32) { lsum[c * 8 + qd] = ssum[r]; lsq[c * 8 + qd] = ssq[r]; }
    }
    __syncthreads();
    if (t < D::C16) {
        for (int half = 0; half < 2; half++) {
            const double* src = half ? lsq : lsum;
            double sacc = 0.0;
            for (int l = 0; l < 8; l++) sacc += src[t * 8 + l];
            out[D::NBLK * 256 + half * D::C16 + t] = sacc;
            if (half == 1 && (t >> 4) < C - CY) {                              // the diagonal element of block (t / 16, t / 16)
                const int rr = t & 15;
                out[(size_t)blk_index(t >> 4, t >> 4) * 256 + (rr >> 2) * 64 + 16 * (rr & 3) + rr] = sacc;
            }
        }
    }
}

// the far rows of k_gram_i8 (a value outside its column's fixed-point range): their products in fp64, as one more partial record per
// partition (index G).  One work-group per 16 x 16 block of the record, a thread per element, the rows in ASCENDING order whatever
// order the tiles were processed in (the masks are walked, not a list: repeats are bit-identical); the sums of the record stay zero
// (k_gram_i8 counted the far rows in its column sums and squares), and so does the diagonal.
template <int C, int CY>
__global__ __launch_bounds__(256) void k_gram_far(const double* __restrict__ X, const double* __restrict__ Y, size_t ldx, size_t ldy, int M,
                                                  int P, long long n, long long split, const double* __restrict__ shift,
                                                  double* __restrict__ partial, int G, const unsigned long long* __restrict__ far_mask,
                                                  const unsigned long long* __restrict__ far_sum, long long tmax) {
    using D = GramI8<C, CY>;
    const int part = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    double* out = partial + ((size_t)part * (G + 1) + G) * D::PSZ;
    if (b == D::NBLK) {                                                        // the sums part of the record
        for (int e = t; e < 2 * D::C16; e += 256) out[D::NBLK * 256 + e] = 0.0;
        return;
    }
    int blk = b, bi = 0;
    while (blk >= C - bi) { blk -= C - bi; bi++; }
    const int bj = bi + blk;
    // element t of the block in the f64 MFMA's C layout: row = (lane >> 4) + 4 reg, col = lane & 15 with t = 64 reg + lane
    const int row = 16 * bi + ((t & 63) >> 4) + 4 * (t >> 6), col = 16 * bj + (t & 15);
    const bool live = row != col && row < M + P && col < M + P;
    const double* pr = (row < M) ? X + (size_t)row * ldx : Y + (size_t)((row < M + P ? row : M) - M) * ldy;
    const double* pc = (col < M) ? X + (size_t)col * ldx : Y + (size_t)((col < M + P ? col : M) - M) * ldy;
    const double sr = live ? shift[row] : 0.0, sc = live ? shift[col] : 0.0;
    const long long r_begin = part ? split : 0, r_end = part ? n : split;
    const long long t0 = r_begin & ~1LL;
    const long long ntiles = (r_end > r_begin) ? (r_end - t0 + D::TRW - 1) / D::TRW : 0;
    const unsigned long long* fs = far_sum + (size_t)part * ((tmax + 63) / 64);
    const unsigned long long* fm = far_mask + (size_t)part * tmax;
    double s = 0.0;
    const bool any = far_sum[(size_t)2 * ((tmax + 63) / 64)] != 0ull;
    for (long long w = 0; any && w < (ntiles + 63) / 64; w++) {
        unsigned long long tiles = fs[w];
        while (tiles) {
            const int tb = __ffsll((long long)tiles) - 1;
            tiles &= tiles - 1;
            unsigned long long rows = fm[w * 64 + tb];
            while (rows) {
                const int rb = __ffsll((long long)rows) - 1;
                rows &= rows - 1;
                const long long r = t0 + (w * 64 + tb) * D::TRW + rb;
                if (live) s = fma(pr[r] - sr, pc[r] - sc, s);
            }
        }
    }
    out[(size_t)b * 256 + t] = s;
}

// Sum the per-work-group partial records in a fixed order and scatter into the stats record.
constexpr int SR_SL = 16;          // slices of the G partial records per work-group (64 elements x SR_SL slices = 1024 threads)
template <int C, int CY>
__global__ __launch_bounds__(64 * SR_SL) void k_stats_reduce(const double* __restrict__ partial, int G,
                                                      double* __restrict__ stats, long long n_train,
                                                      long long n_test) {
    using D = GramDims<C, CY>;
    const StatsLayout L = stats_layout(D::C16, 0);
    const int part = blockIdx.y;
    // 64 consecutive record elements x SR_SL slices of the G partial records per block: a wave reads 512 contiguous bytes of one
    // record per instruction (16 elements x 16 slices read 128-byte pieces of four records: 30 us for the 23 MB of a 144-column
    // set); slices are combined in a fixed order, each slice as four interleaved running sums (four loads in flight per thread).
    // Sixteen slices on 1024 threads (four on 256 until round 3): the kernel is a chain of dependent loads on a nearly empty chip
    __shared__ double red[SR_SL][64];
    const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + el;
    if (blockIdx.x == 0 && threadIdx.x == 0 && part == 0) { stats[L.off_n] = (double)n_train; stats[L.off_n + 1] = (double)n_test; }
    if (CY > 0 && blockIdx.x == gridDim.x - 1) {
        // the skipped pure-Y blocks: zero off their diagonal (a memset of the whole record used to do this: one launch more)
        constexpr int y0 = 16 * (C - CY), ny = 16 * CY;
        double* Gz = stats + L.off_G[part];
        for (int i = threadIdx.x; i < ny * ny; i += 64 * SR_SL) {
            const int r = y0 + i % ny, c = y0 + i / ny;
            if (r != c) Gz[r + (size_t)D::C16 * c] = 0.0;
        }
    }
    double ps = 0.0;
    if (e < D::PSZ) {
        const double* p = partial + (size_t)part * G * D::PSZ + e;
        const int g0 = (int)((long long)G * sl / SR_SL), g1 = (int)((long long)G * (sl + 1) / SR_SL);
        double q0 = 0.0, q1 = 0.0, q2 = 0.0, q3 = 0.0;
        int g = g0;
        for (; g + 3 < g1; g += 4) {
            q0 += p[(size_t)g * D::PSZ];
            q1 += p[(size_t)(g + 1) * D::PSZ];
            q2 += p[(size_t)(g + 2) * D::PSZ];
            q3 += p[(size_t)(g + 3) * D::PSZ];
        }
        for (; g < g1; g++) q0 += p[(size_t)g * D::PSZ];
        ps = (q0 + q1) + (q2 + q3);
    }
    red[sl][el] = ps;
    __syncthreads();
    if (sl != 0 || e >= D::PSZ) return;
    double s = 0.0;
#pragma unroll
    for (int q4 = 0; q4 < SR_SL; q4 += 4) s += (red[q4][el] + red[q4 + 1][el]) + (red[q4 + 2][el] + red[q4 + 3][el]);
    if (e >= D::NBLK * 256) {
        const int c = e - D::NBLK * 256;
        if (c < D::C16) stats[L.off_sum[part] + c] = s;
        else if (c - D::C16 >= 16 * (C - CY)) {     // diagonal of the skipped pure-Y blocks (rest stays 0)
            const int cc = c - D::C16;
            stats[L.off_G[part] + cc + (size_t)D::C16 * cc] = s;
        }
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

template <int C, int CY>
int run_gram(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
             size_t P, long long split, double* stats) {
    using D = GramDims<C, CY>;
    const StatsLayout L = stats_layout(M, P);
    const long long ntr = split, nte = (long long)n - split;
    const long long tiles = ((ntr > nte ? ntr : nte) + TR - 1) / TR + 1;
    long long G = tiles / 2;           // >= 2 tiles per work-group amortise its prologue / epilogue
    if (G < 1) G = 1;
    const long long gmax = (D::NW == 8) ? 256 : 384;   // resident work-groups: 2 x 8 waves or 3 x 4 waves per CU
    if (G > gmax) G = gmax;
    const size_t pbytes = (size_t)2 * G * D::PSZ * sizeof(double);
    double* partial = (double*)abc_ws_alloc(ctx, pbytes);
    if (!partial) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted (%zu B)", pbytes);
    const int vec_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)Y & 15) == 0);
    const size_t lds_bytes = (size_t)D::LDS_D * sizeof(double);
    // per device and cheap: set on every launch (a function-level flag would be wrong for a second device)
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram<C, CY>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
    {
        StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
        hipLaunchKernelGGL((k_gram<C, CY>), dim3((unsigned)G, 2), dim3(D::NT), lds_bytes, ctx->stream, X, Y, ldx, ldy, (int)M,
                           (int)P, (long long)n, split, stats + L.off_shift, partial, vec_ok);
    }
    ABC_HIP(ctx, hipGetLastError());
    StageTimer tm2(ctx, ctx->in_mvn ? -1 : ST_STATS_REDUCE);
    hipLaunchKernelGGL((k_stats_reduce<C, CY>), dim3((D::PSZ + 63) / 64, 2), dim3(64 * SR_SL), 0, ctx->stream, partial, (int)G,
                       stats, ntr, nte);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

template <int C, int CY, int NW, bool PRIV, int R = 3, bool TABLE = false>
int run_gram_dma(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                 size_t P, long long split, double* stats) {
    using D = GramDimsDma<C, CY, NW, R>;
    const StatsLayout L = stats_layout(M, P);
    const long long ntr = split, nte = (long long)n - split;
    constexpr int TRK = PRIV ? 16 * NW : TR;          // rows per tile (k_gram_dma)
    const long long tiles = ((ntr > nte ? ntr : nte) + TRK - 1) / TRK + 1;
    long long G = tiles / 2;
    if (G < 1) G = 1;
    if (G > 128) G = 128;              // 150 KB of LDS: one work-group per CU, 2 partitions x 128
    const size_t pbytes = (size_t)2 * G * D::PSZ * sizeof(double);
    double* partial = (double*)abc_ws_alloc(ctx, pbytes);
    if (!partial) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted (%zu B)", pbytes);
    // wave-private staging: NW rings of R chunks of 16 rows x C16 columns; the epilogue stages four waves' accumulators
    constexpr size_t epi4 = (size_t)4 * D::NBLK * 256;          // the epilogue stages four waves' accumulators (two if that is too much)
    constexpr size_t lds_priv = (size_t)NW * R * D::C16 * 16, lds_epi = (epi4 * sizeof(double) <= 160 * 1024) ? epi4 : epi4 / 2,
                     lds_cs = (size_t)C * D::NT;
    constexpr size_t lds_pmax = lds_priv > lds_epi ? (lds_priv > lds_cs ? lds_priv : lds_cs) : (lds_epi > lds_cs ? lds_epi : lds_cs);
    const size_t lds_bytes = (PRIV ? lds_pmax : (size_t)D::LDS_D) * sizeof(double);
    static_assert(!PRIV || lds_pmax * sizeof(double) <= 160 * 1024, "k_gram_dma: ring + epilogue exceed the 160 KB of LDS");
    // per device and cheap: set on every launch (a function-level flag would be wrong for a second device)
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram_dma<C, CY, NW, PRIV, R, TABLE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)lds_bytes));
    {
        StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
        hipLaunchKernelGGL((k_gram_dma<C, CY, NW, PRIV, R, TABLE>), dim3((unsigned)G, 2), dim3(D::NT), lds_bytes, ctx->stream, X, Y, ldx,
                           ldy, (int)M, (int)P, (long long)n, split, stats + L.off_shift, partial);
    }
    ABC_HIP(ctx, hipGetLastError());
    StageTimer tm2(ctx, ctx->in_mvn ? -1 : ST_STATS_REDUCE);
    hipLaunchKernelGGL((k_stats_reduce<C, CY>), dim3((D::PSZ + 63) / 64, 2), dim3(64 * SR_SL), 0, ctx->stream, partial, (int)G,
                       stats, ntr, nte);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

template <int C, int CY>
int run_gram_dma8(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P,
                  long long split, double* stats) {
    using D = GramDimsDma<C, CY, 8, 3>;
    const StatsLayout L = stats_layout(M, P);
    const long long ntr = split, nte = (long long)n - split;
    const long long tiles = ((ntr > nte ? ntr : nte) + 63) / 64 + 1;
    long long G = tiles / 2;
    if (G < 1) G = 1;
    if (G > 128) G = 128;              // ~150 KB of LDS: one work-group (eight waves) per CU, 2 partitions x 128
    const size_t pbytes = (size_t)2 * G * D::PSZ * sizeof(double);
    double* partial = (double*)abc_ws_alloc(ctx, pbytes);
    if (!partial) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted (%zu B)", pbytes);
    constexpr size_t epi4 = (size_t)4 * D::NBLK * 256;
    constexpr size_t lds_ring = (size_t)8 * 3 * D::C16 * 8, lds_epi = (epi4 * sizeof(double) <= 160 * 1024) ? epi4 : epi4 / 2,
                     lds_cs = (size_t)C * 512;
    constexpr size_t lds_max = lds_ring > lds_epi ? (lds_ring > lds_cs ? lds_ring : lds_cs) : (lds_epi > lds_cs ? lds_epi : lds_cs);
    static_assert(lds_max * sizeof(double) <= 160 * 1024, "k_gram_dma8: ring / epilogue exceed the 160 KB of LDS");
    const size_t lds_bytes = lds_max * sizeof(double);
    // This kernel's loads go WITHOUT the non-temporal hint (round 4).  Its DMA pieces are 64 bytes of a column -- half a 128-byte
    // line, the other half going to the neighbouring wave at about the same time --, and a line fetched with the hint is not kept
    // for the neighbour: FETCH_SIZE showed 13.5 GB for the 7.68 GB of configs[3]'s set (1.76 x), 8.25 GB without the hint, and the
    // kernel 2.25 -> 2.09 ms (it is bound by the fp64 matrix pipe first, so the doubled traffic cost 7 %, not 76 %).  The kernels
    // whose pieces are whole lines (k_gram_dma: 16 rows, k_gram_i8: 32 rows of a column) keep the hint.  ABC_GRAM_DMA8_NT: A/B.
    static const bool nont = abc_diag_env("ABC_GRAM_DMA8_NT") == nullptr;
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram_dma8<C, CY, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram_dma8<C, CY, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    {
        StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
        if (nont)
            hipLaunchKernelGGL((k_gram_dma8<C, CY, false>), dim3((unsigned)G, 2), dim3(512), lds_bytes, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P,
                               (long long)n, split, stats + L.off_shift, partial);
        else
            hipLaunchKernelGGL((k_gram_dma8<C, CY, true>), dim3((unsigned)G, 2), dim3(512), lds_bytes, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P,
                               (long long)n, split, stats + L.off_shift, partial);
    }
    ABC_HIP(ctx, hipGetLastError());
    StageTimer tm2(ctx, ctx->in_mvn ? -1 : ST_STATS_REDUCE);
    hipLaunchKernelGGL((k_stats_reduce<C, CY>), dim3((D::PSZ + 63) / 64, 2), dim3(64 * SR_SL), 0, ctx->stream, partial, (int)G, stats, ntr,
                       nte);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

template <int C, int CY>
int run_gram_wide(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P,
                  long long split, double* stats) {
    using D = GramWide<C, CY>;
    const StatsLayout L = stats_layout(M, P);
    const long long ntr = split, nte = (long long)n - split;
    const long long tiles = ((ntr > nte ? ntr : nte) + D::TRW - 1) / D::TRW + 1;
    long long G = tiles / 2;
    if (G < 1) G = 1;
    if (G > 128) G = 128;              // 84 KB of LDS and 512 threads: one work-group per CU, 2 partitions x 128
    const size_t pbytes = (size_t)2 * G * D::PSZ * sizeof(double);
    double* partial = (double*)abc_ws_alloc(ctx, pbytes);
    if (!partial) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted (%zu B)", pbytes);
    const int vec_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)Y & 15) == 0);
    const size_t lds_bytes = (size_t)D::LDS_D * sizeof(double);
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram_wide<C, CY>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    {
        StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
        hipLaunchKernelGGL((k_gram_wide<C, CY>), dim3((unsigned)G, 2), dim3(D::NT), lds_bytes, ctx->stream, X, Y, ldx, ldy, (int)M,
                           (int)P, (long long)n, split, stats + L.off_shift, partial, vec_ok);
    }
    ABC_HIP(ctx, hipGetLastError());
    StageTimer tm2(ctx, ctx->in_mvn ? -1 : ST_STATS_REDUCE);
    hipLaunchKernelGGL((k_stats_reduce<C, CY>), dim3((D::PSZ + 63) / 64, 2), dim3(64 * SR_SL), 0, ctx->stream, partial, (int)G,
                       stats, ntr, nte);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

template <int C, int CY>
int run_gram_i8(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P,
                long long split, double* stats) {
    using D = GramI8<C, CY>;
    const StatsLayout L = stats_layout(M, P);
    const long long ntr = split, nte = (long long)n - split;
    const long long tmax = ((ntr > nte ? ntr : nte) + D::TRW - 1) / D::TRW + 2;      // tiles of a partition (t0 may start a row early)
    long long G = tmax / 2;
    if (G < 1) G = 1;
    if (G > 127) G = 127;              // 154 KB of LDS, 512 threads with ~250 registers: one work-group per CU; with k_gram_far's record
                                       // 128 records per partition (k_stats_reduce splits them evenly over its 16 slices)
    const size_t pbytes = (size_t)2 * (G + 1) * D::PSZ * sizeof(double);
    double* partial = (double*)abc_ws_alloc(ctx, pbytes);
    int* escale = (int*)abc_ws_alloc(ctx, (size_t)D::C32 * sizeof(int));
    const size_t sumw = (size_t)((tmax + 63) / 64);
    unsigned long long* far_mask = (unsigned long long*)abc_ws_alloc(ctx, (size_t)2 * tmax * 8);
    unsigned long long* far_sum = (unsigned long long*)abc_ws_alloc(ctx, (2 * sumw + 1) * 8);
    if (!partial || !escale || !far_mask || !far_sum) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted (%zu B)", pbytes);
    const int lds_b = D::lds_bytes((int)(M + P)), nraw = D::raw_tiles((int)(M + P));
    ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram_i8<C, CY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
    static const int abl = abc_diag_env("ABC_GRAM_ABL") ? atoi(abc_diag_env("ABC_GRAM_ABL")) : 0;      // (phases left out: timings only)
    {
        StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
        hipLaunchKernelGGL(k_pilot_scale, dim3((unsigned)D::C16), dim3(1024), 0, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P, (long long)n,
                           (const double*)(stats + L.off_shift), escale);
        ABC_HIP(ctx, hipMemsetAsync(far_sum, 0, (2 * sumw + 1) * 8, ctx->stream));
        hipLaunchKernelGGL((k_gram_i8<C, CY>), dim3((unsigned)G, 2), dim3(D::NT), (size_t)lds_b, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P,
                           (long long)n, split, (const double*)(stats + L.off_shift), (const int*)escale, partial, far_mask, far_sum, tmax, nraw, abl);
        hipLaunchKernelGGL((k_gram_far<C, CY>), dim3((unsigned)(D::NBLK + 1), 2), dim3(256), 0, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P,
                           (long long)n, split, (const double*)(stats + L.off_shift), partial, (int)G, (const unsigned long long*)far_mask,
                           (const unsigned long long*)far_sum, tmax);
    }
    ABC_HIP(ctx, hipGetLastError());
    if (abc_diag_env("ABC_GRAM_DEBUG")) {        // (diagnostic: the columns' fixed-point binades)
        std::vector<int> he(D::C32);
        ABC_HIP(ctx, hipMemcpy(he.data(), escale, D::C16 * sizeof(int), hipMemcpyDeviceToHost));
        for (int c = 0; c < (int)(M + P); c++) fprintf(stderr, "%d%s", he[c], (c + 1) % 32 ? " " : "\n");
        fprintf(stderr, "\n");
    }
    StageTimer tm2(ctx, ctx->in_mvn ? -1 : ST_STATS_REDUCE);
    hipLaunchKernelGGL((k_stats_reduce<C, CY>), dim3((D::PSZ + 63) / 64, 2), dim3(64 * SR_SL), 0, ctx->stream, partial, (int)(G + 1),
                       stats, ntr, nte);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// ---- wide sets (M+P > 96): column groups of <= 48, one k_gram<.,0,TABLE> launch per pair of groups ------------
__global__ void k_group_table(const double* X, const double* Y, size_t ldx, size_t ldy, int M, int P, int ga0, int gan,
                              int gb0, int gbn, const double* __restrict__ shift_big, const double** __restrict__ tab,
                              double* __restrict__ shift_loc, int* __restrict__ gmap, int C16loc) {
    const int a = threadIdx.x;
    if (a >= C16loc) return;
    int gc = -1;                                   // global column of local column a
    if (a < 48) { if (a < gan) gc = ga0 + a; }
    else if (a - 48 < gbn) gc = gb0 + (a - 48);
    gmap[a] = gc;
    tab[a] = (gc < 0) ? nullptr : (gc < M ? X + (size_t)gc * ldx : Y + (size_t)(gc - M) * ldy);
    shift_loc[a] = (gc < 0) ? 0.0 : shift_big[gc];
}

__global__ __launch_bounds__(256) void k_group_scatter(const double* __restrict__ loc, int C16loc, const int* __restrict__ gmap,
                                                       double* __restrict__ big, int C16big) {
    const StatsLayout LL = stats_layout(C16loc, 0), LB = stats_layout(C16big, 0);
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e == 0) { big[LB.off_n] = loc[LL.off_n]; big[LB.off_n + 1] = loc[LL.off_n + 1]; }
    if (e >= C16loc * C16loc) return;
    const int a = e % C16loc, b = e / C16loc;
    const int ga = gmap[a], gb = gmap[b];
    if (ga < 0 || gb < 0) return;
    for (int part = 0; part < 2; part++) {
        big[LB.off_G[part] + ga + (size_t)C16big * gb] = loc[LL.off_G[part] + a + (size_t)C16loc * b];
        if (b == 0) big[LB.off_sum[part] + ga] = loc[LL.off_sum[part] + a];
    }
}

int run_gram_grouped(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                     size_t P, long long split, double* stats) {
    const StatsLayout LB = stats_layout(M, P);
    const int ncol = (int)(M + P);
    const int ng = (ncol + 47) / 48;
    const StatsLayout LL = stats_layout(96, 0);
    double* loc = (double*)abc_ws_alloc(ctx, LL.len * sizeof(double));
    const double** tab = (const double**)abc_ws_alloc(ctx, 96 * sizeof(double*));
    int* gmap = (int*)abc_ws_alloc(ctx, 96 * sizeof(int));
    if (!loc || !tab || !gmap) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted");
    const size_t ws_mark = ctx->ws_off;
    for (int ga = 0; ga < ng; ga++)
        for (int gb = ga + 1; gb < ng; gb++) {
            ctx->ws_off = ws_mark;                       // the per-launch partial records reuse the same arena space
            const int ga0 = 48 * ga, gan = ncol - ga0 < 48 ? ncol - ga0 : 48;
            const int gb0 = 48 * gb, gbn = ncol - gb0 < 48 ? ncol - gb0 : 48;
            hipLaunchKernelGGL(k_group_table, dim3(1), dim3(128), 0, ctx->stream, X, Y, ldx, ldy, (int)M, (int)P, ga0, gan,
                               gb0, gbn, stats + LB.off_shift, tab, loc + LL.off_shift, gmap, 96);
            const bool dma_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (n % 2 == 0) && (((uintptr_t)X & 15) == 0) &&
                                (((uintptr_t)Y & 15) == 0) && n >= 2;
            if (dma_ok) {
                // the pair's 96 columns through the four-wave LDS-DMA kernel (column-pointer table instead of X / Y)
                ABC_TRY((run_gram_dma<6, 0, 4, true, 3, true>(ctx, (const double*)tab, nullptr, n, ldx, ldy, 96, 0, split, loc)));
            } else {
            using D = GramDims<6, 0>;
            const long long ntr = split, nte = (long long)n - split;
            const long long tiles = ((ntr > nte ? ntr : nte) + TR - 1) / TR + 1;
            long long G = tiles / 2;
            if (G < 1) G = 1;
            if (G > 128) G = 128;                        // 100 KB of LDS: one work-group per CU
            double* partial = (double*)abc_ws_alloc(ctx, (size_t)2 * G * D::PSZ * sizeof(double));
            if (!partial) ABC_FAIL(ctx, ABC_ERR_NOMEM, "gram: workspace exhausted");
            const int vec_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)Y & 15) == 0);
            const size_t lds_bytes = (size_t)D::LDS_D * sizeof(double);
            // per device and cheap: set on every launch (a function-level flag would be wrong for a second device)
            ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_gram<6, 0, true>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
            {
                StageTimer tm(ctx, ctx->in_mvn ? -1 : ST_GRAM);
                hipLaunchKernelGGL((k_gram<6, 0, true>), dim3((unsigned)G, 2), dim3(D::NT), lds_bytes, ctx->stream,
                                   (const double*)tab, (const double*)nullptr, ldx, ldy, 96, 0, (long long)n, split,
                                   loc + LL.off_shift, partial, vec_ok);
            }
            hipLaunchKernelGGL((k_stats_reduce<6, 0>), dim3((D::PSZ + 63) / 64, 2), dim3(64 * SR_SL), 0, ctx->stream, partial,
                               (int)G, loc, ntr, nte);
            }
            hipLaunchKernelGGL(k_group_scatter, dim3((96 * 96 + 255) / 256), dim3(256), 0, ctx->stream, loc, 96, gmap, stats,
                               (int)LB.C16);
            ABC_HIP(ctx, hipGetLastError());
        }
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

// Will launch_stats_accumulate take the byte-limb kernel for this set?  (The same conditions as below, for a caller that arranges its
// streams around that kernel: one 512-thread work-group per CU with 150-160 KB of LDS -- anything resident beside it costs it CUs.)
bool abc_gram_takes_i8(const abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P,
                       uint64_t n_train_global, size_t n_set) {
    const size_t C = (M + P + 15) / 16;
    size_t CY = C - (M + 15) / 16;
    if (CY > 2) CY = 2;
    const bool dma_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (n % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)Y & 15) == 0) && n >= 2;
    const size_t rows_set = n_set ? n_set : n;
    const size_t ntr_set = n_train_global < rows_set ? (size_t)n_train_global : rows_set, nte_set = rows_set - ntr_set;
    const size_t part_min = (ntr_set && nte_set) ? (ntr_set < nte_set ? ntr_set : nte_set) : (ntr_set ? ntr_set : nte_set);
    const bool i8_rows = ctx->gram_mode == ABC_GRAM_I8 ? rows_set >= 200000 : (ctx->gram_mode == ABC_GRAM_AUTO && part_min >= 400000);
    if (!i8_rows || !dma_ok || n < 4096) return false;
    if (C == 6) return rows_set >= 2000000 && CY >= 1 && !abc_diag_env("ABC_GRAM_DMA8_96");
    return C >= 7 && C <= 10;
}

int launch_stats_accumulate(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy,
                            size_t M, size_t P, uint64_t row0, uint64_t n_train_global, double* stats, size_t n_set) {
    long long split = 0;
    if (n_train_global > row0) split = (long long)((n_train_global - row0) < n ? (n_train_global - row0) : n);
    const size_t C = (M + P + 15) / 16;
    size_t CY = C - (M + 15) / 16;      // trailing blocks without any metric column
    if (CY > 2) CY = 2;
    // Two kernels.  Up to 48 columns (C <= 3) the VGPR-staged k_gram is the default since its loads carry the non-temporal
    // policy: 75 us against 84 us for the LDS-DMA kernel at N = 1e6, M = 32, P = 16 (0.68 against 0.60 of the HBM peak inside
    // a generation); 49..96 columns run the eight-wave LDS-DMA variant (0.29 against 0.68 ms on the configs[3] shape).
    // LDS-DMA staging needs 16-B aligned columns and an even row count (row pairs never straddle the array end).
    const bool dma_ok = (ldx % 2 == 0) && (ldy % 2 == 0) && (n % 2 == 0) && (((uintptr_t)X & 15) == 0) &&
                        (((uintptr_t)Y & 15) == 0) && n >= 2;
    // 4..6 column blocks (49..96 columns, e.g. BASELINE configs[3]: 64 metrics + 32 parameters): k_gram_dma8, EIGHT waves of
    // wave-private staging in 8-row chunks (two waves per SIMD keep the fp64 matrix pipe busy: 0.36 -> 0.29 ms on the configs[3]
    // shard against the four-wave, 16-row-chunk variant, which stays for the column-pointer-table mode of the grouped path).
    // (6, 0) needs 172 KB for its epilogue and stays on the VGPR-staged kernel.
    // 81..96 columns of a LARGE set (configs[3]: 64 metrics + 32 parameters, 1e7 rows) go through the byte-limb kernel of the wide
    // sets (k_gram_i8, below): 1.59 ms against k_gram_dma8's 1.99 at 1e7 rows x 96 columns (0.60 against 0.48 of HBM: the fp64
    // matrix pipe is 0.62 busy there), the same statistics to the byte products' error (section 4).  From 2e6 rows: where it was measured.
    static const bool dma8_96 = abc_diag_env("ABC_GRAM_DMA8_96") != nullptr;              // A/B switch: the fp64 kernel as before
    // WHERE THE BYTE-LIMB KERNEL IS THE DEFAULT (round 6).  Its noise -- every value rounded to a 32-bit grid -- is harmless at the
    // Gram's own scale (4e-11 of sqrt(G_aa G_bb)) but not always at the LOADINGS: a component that fits noise works on cross
    // products sqrt(rows) below that scale, with close eigenvalues, and tests/fuzz/wide_model_fuzz.py found a used loading column
    // 4.3e-6 off the oracle's (fp64 kernels: 4e-10) at 66 000 training rows x 29 responses x 30 components -- BASELINE.json allows
    // 1e-6.  The error falls faster than 1 / rows: 64 fuzzed sets whose partitions hold 450 000 rows and more stay within 2.6e-7
    // (profiles/r06_wide_model_fuzz_big.json).  So ABC_GRAM_AUTO takes the kernel only where EVERY non-empty partition of the
    // WHOLE set (training / validation rows: n_train_global, n_set) has at least 400 000 rows -- configs[4] (5e5 + 5e5), configs[3]
    // (5e6 + 5e6) -- and the fp64 kernels below that; ABC_GRAM_I8 is round 5's rule (200 000 rows in the whole set) for A/B runs and tests.
    const size_t rows_set = n_set ? n_set : n;
    const size_t ntr_set = n_train_global < rows_set ? (size_t)n_train_global : rows_set, nte_set = rows_set - ntr_set;
    const size_t part_min = (ntr_set && nte_set) ? (ntr_set < nte_set ? ntr_set : nte_set) : (ntr_set ? ntr_set : nte_set);
    const bool i8_rows = ctx->gram_mode == ABC_GRAM_I8 ? rows_set >= 200000 : (ctx->gram_mode == ABC_GRAM_AUTO && part_min >= 400000);
    if (!dma8_96 && C == 6 && dma_ok && i8_rows && rows_set >= 2000000 && n >= 4096) {
        if (CY == 2) return run_gram_i8<6, 2>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
        if (CY == 1) return run_gram_i8<6, 1>(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
    }
#define GRAM_DMA4_CASE(c, cy) if (C == c && CY == cy && dma_ok) return run_gram_dma8<c, cy>(ctx, X, Y, n, ldx, ldy, M, P, split, stats)
    GRAM_DMA4_CASE(4, 0); GRAM_DMA4_CASE(4, 1); GRAM_DMA4_CASE(4, 2); GRAM_DMA4_CASE(5, 0); GRAM_DMA4_CASE(5, 1); GRAM_DMA4_CASE(5, 2);
    GRAM_DMA4_CASE(6, 1); GRAM_DMA4_CASE(6, 2);
#undef GRAM_DMA4_CASE
#define GRAM_CASE(c, cy) if (C == c && CY == cy) return run_gram<c, cy>(ctx, X, Y, n, ldx, ldy, M, P, split, stats)
    GRAM_CASE(1, 0); GRAM_CASE(2, 0); GRAM_CASE(2, 1); GRAM_CASE(3, 0); GRAM_CASE(3, 1); GRAM_CASE(3, 2);
    GRAM_CASE(4, 0); GRAM_CASE(4, 1); GRAM_CASE(4, 2); GRAM_CASE(5, 0); GRAM_CASE(5, 1); GRAM_CASE(5, 2);
    GRAM_CASE(6, 0); GRAM_CASE(6, 1); GRAM_CASE(6, 2);
#undef GRAM_CASE
    // 8..10 column blocks: one launch, the Gram blocks dealt out to the waves of a work-group (7 blocks: the compiler spills
    // that instantiation; it stays on the grouped path)
    // ... large sets on the byte-limb kernel on the i8 matrix pipe (k_gram_i8: sums and the diagonal exact, off-diagonal products
    // to ~1e-10 of sqrt(G_aa G_bb)); small ones, and all with ABC_GRAM_FP64 (A/B runs, tests), on the fp64 matrix pipe
    const bool gram_fp64 = ctx->gram_mode == ABC_GRAM_FP64;      // (abc_ctx_set_gram_mode; ABC_DIAG=1 ABC_GRAM_FP64=1 presets it)
    // (LDS-DMA staging: 16-byte aligned columns, an even row count; from 200000 rows: the values are rounded to a 32-bit grid of
    // 10 .. 19 sigma, noise of 3e-9 sigma per value that averages out with the square root of the rows -- 1e-10 of sqrt(G_aa G_bb) at
    // 35000 rows per partition, which the 32nd loading of a 128-metric model amplifies to 2e-7; 4e-8 at 1e6 rows)
    // (a rank whose shard the kernel cannot take -- an odd row count, columns that are not 16-byte aligned, fewer than 4096 rows --
    // runs the fp64 kernel on ITS rows: the decision above is the same on every rank, this one is about what can run)
    const bool i8_ok = !gram_fp64 && i8_rows && n >= 4096 && dma_ok;
#define GRAM_WIDE_CASE(c, cy) if (C == c && CY == cy) return i8_ok ? run_gram_i8<c, cy>(ctx, X, Y, n, ldx, ldy, M, P, split, stats) \
                                                                    : run_gram_wide<c, cy>(ctx, X, Y, n, ldx, ldy, M, P, split, stats)
    // (7 blocks, 97..112 columns: the byte-limb kernel where it applies -- round 5; the fp64 one-launch kernel spills at that width, so
    // small sets of it stay on the grouped path below, which reads every column twice)
#define GRAM_I8_ONLY_CASE(c, cy) if (C == c && CY == cy && i8_ok) return run_gram_i8<c, cy>(ctx, X, Y, n, ldx, ldy, M, P, split, stats)
    GRAM_I8_ONLY_CASE(7, 0); GRAM_I8_ONLY_CASE(7, 1); GRAM_I8_ONLY_CASE(7, 2);
#undef GRAM_I8_ONLY_CASE
    GRAM_WIDE_CASE(8, 0); GRAM_WIDE_CASE(8, 1); GRAM_WIDE_CASE(8, 2);
    GRAM_WIDE_CASE(9, 0); GRAM_WIDE_CASE(9, 1); GRAM_WIDE_CASE(9, 2); GRAM_WIDE_CASE(10, 0); GRAM_WIDE_CASE(10, 1); GRAM_WIDE_CASE(10, 2);
#undef GRAM_WIDE_CASE
    // wider sets: column groups of 48, one launch per pair of groups (every column is read ceil(columns / 48) - 1 times)
    return run_gram_grouped(ctx, X, Y, n, ldx, ldy, M, P, split, stats);
}
