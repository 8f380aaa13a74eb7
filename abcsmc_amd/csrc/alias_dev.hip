// Walker alias table of the resampling step, built ON THE DEVICE, bit for bit the table GSL's gsl_ran_discrete_preproc builds
// (AbcUtil.cpp:111-120 -> gsl_ran_discrete_preproc; restated sequentially in alias_host.h, which stays the fallback).
//
// GSL's algorithm is two sequential chains of ROUNDED floating-point operations: the running total `s += w[k]`, and the
// serving loop in which the "big" entry on top of its stack gives every "small" entry what it lacks (`eb -= mean - E[s]`) and,
// once it falls below the mean itself, is served by the next big.  Every entry of the table depends on the rounding of all
// operations before it, so round 2 shipped the weights to the host, built the table there (0.18-0.37 ms at K = 1e5, the GPU
// idle) and shipped it back.  Round 3 runs both chains as PARALLEL PREFIX SCANS and stays exact:
//   * on a fixed-point grid (all values are multiples of g0 = ulp(mean) / 2, resp. of a grid 44 bits below the total's ulp),
//     one rounded operation  x -> RNE_k(x + c)  (round to nearest, ties to even, at binade level k) is a non-decreasing step
//     function with two steps of 2^k per period 2^(k+1),
//         f(y) = 2^k (floor((y - t0) / P) + floor((y - t1) / P)) + b,      P = 2^(k+1),  t0 <= t1 <= t0 + P
//     (the two thresholds differ by the tie rule), and this four-parameter family is CLOSED UNDER COMPOSITION (compose()): a chain
//     of such operations is an associative scan of maps, applied to the chain's start value at the end;
//   * the level k of every operation -- the binade of its RESULT -- and the interleaving of small-steps and hand-overs in the
//     serving loop are SPECULATED from exact, unrounded 128-bit prefix sums of the deficits mean - E[s] and the excesses
//     E[b] - mean (the rounded chain drifts from them by ~1e-12 relative: a wrong guess needs a value within that of a power of two
//     or of the mean);
//   * every step of the result is then VERIFIED in parallel with the real IEEE operation on the claimed operands, and every
//     comparison against the mean with the claimed values: if all steps hold, the claimed chain IS the sequential chain by
//     induction.  Any failed check (and any input outside the grid: non-finite or negative weights, a weight range beyond 2^44)
//     raises a flag; the caller then builds the table on the host as before.
// scripts/alias_scan_proto.py is the same algorithm in exact Python integers (900 random weight vectors of six kinds, tie-heavy
// and nearly uniform ones included: bit-exact, no flag); tests/test_gpu_parity.py::test_device_alias_table_* compare the device
// table with the oracle's entry by entry.
// Ten launches (the verdict is published by the caller's next kernel), no host involvement, no device-scope fences (reductions across work-groups go through kernel boundaries: every
// "apply" kernel re-derives its block's prefix from the per-block aggregates of the launch before).
#include <stdlib.h>

#include "abc_internal.h"

namespace {

typedef __int128 i128;
typedef unsigned __int128 u128;
typedef unsigned long long u64;

// Elements per thread: this file is compiled TWICE (Makefile: alias_dev.o with four, alias_dev_i2.o with -DABC_AL_I=2).  Every
// kernel of the build is latency-bound on a nearly empty chip, a thread's serial work (compositions of 128-bit step maps) grows
// with its elements and every work-group re-scans the aggregates of all work-groups in its prologue: two elements per thread win
// up to a few hundred thousand entries (0.099 -> 0.086 ms at 1e5), four beyond (0.225 against 0.278 ms at 1e6; one: 0.090 / 0.45).
#ifndef ABC_AL_I
#define ABC_AL_I 4
#endif
#define ABC_AL_CAT2(a, b) a##b
#define ABC_AL_CAT(a, b) ABC_AL_CAT2(a, b)
#define ABC_AL_FN(name) ABC_AL_CAT(name##_i, ABC_AL_I)
constexpr int AL_T = 256;                 // threads per work-group
constexpr int AL_I = ABC_AL_I;            // consecutive elements per thread
constexpr int AL_B = AL_T * AL_I;         // elements per work-group
constexpr int AL_MAXBLK = 8192;           // work-groups whose aggregates one work-group re-scans in its prologue (K <= 4.19e6 / 2.1e6)
constexpr int AL_SUM_BITS = 44;           // the total's grid lies this many bits below its ulp

struct RMap { i128 t0, t1, b; int k; int pad_[3]; };      // 64 bytes

__device__ __forceinline__ i128 shl(i128 v, int s) { return (i128)((u128)v << s); }
__device__ __forceinline__ int bitlen(i128 v) {           // v >= 0
    const u64 hi = (u64)((u128)v >> 64), lo = (u64)v;
    return hi ? 128 - __clzll((long long)hi) : (lo ? 64 - __clzll((long long)lo) : 0);
}
__device__ __forceinline__ int level_of(i128 v) { const int b = bitlen(v); return b > 53 ? b - 53 : 0; }

// y -> round-to-nearest-even at level k of (y + c); sticky: c stands for a real number slightly above the integer c
__device__ __forceinline__ RMap rne_map(int k, i128 c, bool sticky) {
    RMap m;
    m.k = k; m.b = 0; m.pad_[0] = m.pad_[1] = m.pad_[2] = 0;
    if (k == 0) { m.t0 = -c - 1; m.t1 = -c; return m; }
    const i128 P = shl(1, k + 1), h = shl(1, k - 1);
    const i128 t_even = -c - shl(1, k) + h;                // reaching an EVEN multiple of 2^k: a tie rounds up to it
    const i128 t_odd = -c + h + (sticky ? 0 : 1);          // reaching an ODD multiple: a tie stays below
    m.t0 = t_odd - P; m.t1 = t_even;
    return m;
}
__device__ __forceinline__ RMap rmap_identity() { RMap m; m.k = 0; m.t0 = -1; m.t1 = 0; m.b = 0; m.pad_[0] = m.pad_[1] = m.pad_[2] = 0; return m; }
__device__ __forceinline__ i128 rapply(const RMap& m, i128 y) {
    return shl(((y - m.t0) >> (m.k + 1)) + ((y - m.t1) >> (m.k + 1)), m.k) + m.b;
}
// min { y : m(y) >= v }
__device__ __forceinline__ i128 inv_min(const RMap& m, i128 v) {
    const i128 a = -((-(v - m.b)) >> m.k);                 // ceil((v - b) / 2^k): the step count that must be reached
    if (a & 1) return m.t0 + shl((a + 1) >> 1, m.k + 1);
    return m.t1 + shl(a >> 1, m.k + 1);
}
// first m1, then m2
__device__ __forceinline__ RMap compose(const RMap& m1, const RMap& m2) {
    RMap r;
    if (m2.k < m1.k) { r = m1; r.b = rapply(m2, m1.b); return r; }     // 2^k1 N is a multiple of m2's period
    r.k = m2.k; r.t0 = inv_min(m1, m2.t0); r.t1 = inv_min(m1, m2.t1); r.b = m2.b;
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
    return r;
}

// x (>= 0, finite) as an integer multiple of 2^e0; *inexact is set when bits fall below the grid (the result is then the floor)
__device__ __forceinline__ i128 to_grid(double x, int e0, bool* inexact) {
    const u64 bits = (u64)__double_as_longlong(x);
    const int be = (int)((bits >> 52) & 0x7ff);
    u64 mant = bits & 0xfffffffffffffull;
    int e;
    if (be == 0) { if (mant == 0) return 0; e = -1074; } else { mant |= 1ull << 52; e = be - 1075; }
    const int sh = e - e0;
    if (sh >= 0) return (sh < 70) ? shl((i128)mant, sh) : (i128)0;       // (sh >= 70 cannot happen for values the callers admit)
    if (sh <= -64) { *inexact = true; return 0; }
    if (mant & ((1ull << -sh) - 1)) *inexact = true;
    return (i128)(mant >> -sh);
}
// the double with value v 2^e0, if v has at most 53 significant bits (else *ok = false)
__device__ __forceinline__ double from_grid(i128 v, int e0, bool* ok) {
    if (v < 0) { *ok = false; return 0.0; }
    if (v == 0) return 0.0;
    const int bl = bitlen(v), sh = bl > 53 ? bl - 53 : 0;
    const u64 top = (u64)(v >> sh);
    if (shl((i128)top, sh) != v) { *ok = false; return 0.0; }
    return ldexp((double)top, sh + e0);
}

__device__ __forceinline__ i128 shfl_up_i128(i128 v, int d) {
    const u64 lo = (u64)__shfl_up((long long)(u64)v, d, 64), hi = (u64)__shfl_up((long long)(u64)((u128)v >> 64), d, 64);
    return (i128)(((u128)hi << 64) | lo);
}
__device__ __forceinline__ RMap shfl_up_map(const RMap& m, int d) {
    RMap r;
    r.t0 = shfl_up_i128(m.t0, d); r.t1 = shfl_up_i128(m.t1, d); r.b = shfl_up_i128(m.b, d); r.k = __shfl_up(m.k, d, 64);
    r.pad_[0] = r.pad_[1] = r.pad_[2] = 0;
    return r;
}
// inclusive scan of one RMap per thread over the work-group: inside a wave by shuffles (six steps), the four wave totals through
// LDS (first version: Hillis-Steele over all 256 threads through two LDS buffers, eight rounds with a barrier each); returns
// this thread's inclusive value; *excl gets the exclusive one (identity for thread 0)
__device__ __forceinline__ RMap block_scan_maps(RMap mine, RMap (*buf)[AL_T], RMap* excl) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    RMap inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const RMap up = shfl_up_map(inc, d);
        if (lane >= d) inc = compose(up, inc);
    }
    RMap exw = shfl_up_map(inc, 1);                       // exclusive inside the wave
    if (lane == 0) exw = rmap_identity();
    if (lane == 63) buf[0][wave] = inc;
    __syncthreads();
    RMap before = rmap_identity();                        // the waves in front of this one, in order
    for (int w = 0; w < wave; w++) before = compose(before, buf[0][w]);
    __syncthreads();
    *excl = compose(before, exw);
    return compose(before, inc);
}
// composition, in order, of the aggregates agg[0 .. nb): every thread takes a contiguous chunk, then the block scan
__device__ __forceinline__ RMap prefix_of_blocks(const RMap* __restrict__ agg, int nb, RMap (*buf)[AL_T]) {
    const int t = threadIdx.x;
    const int per = (nb + AL_T - 1) / AL_T;
    RMap m = rmap_identity();
    for (int j = 0; j < per; j++) { const int b = t * per + j; if (b < nb) m = compose(m, agg[b]); }
    RMap ex;
    (void)block_scan_maps(m, buf, &ex);
    __shared__ RMap s_tot;
    if (t == AL_T - 1) s_tot = compose(ex, m);
    __syncthreads();
    const RMap r = s_tot;
    __syncthreads();
    return r;
}

struct AlHead {               // device-side header of one build
    double approx_total;      // fp64 sum of the weights (any order: only its binade is used)
    double total;             // the sequential total, bit-exact (k_sum_apply)
    int e0_sum;               // grid of the total's chain
    int e0;                   // grid of the serving chain: ulp(mean) / 2
    unsigned ns, nb;          // smalls, bigs
    unsigned nsteps;          // steps of the serving chain
    int fail;
    i128 MEAN;                // mean on the serving grid
};

__device__ __forceinline__ void al_fail(AlHead* h) { h->fail = 1; }

// ---------------------------------------------------------------------------------------------------------------------------
// 1. the sequential total
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum_d(double v, double* sm) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
// fp64 block sums (approximate prefix sums: they only place every partial sum in its binade)
__global__ __launch_bounds__(AL_T) void k_al_bsum(const double* __restrict__ w, size_t K, double* __restrict__ bsum, AlHead* __restrict__ head,
                                                  int force_fail) {
    __shared__ double sm[4];
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    double s = 0.0;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < AL_I; j++) if (base + j < K) { const double x = w[base + j]; s += x; bad = bad || !(x >= 0.0) || !(x < 1.0e300); }
    // a negative / non-finite / huge weight poisons its block sum: the next kernel (the first that may raise the flag) sees it
    const double bs = block_sum_d(bad ? (double)NAN : s, sm);
    if (threadIdx.x == 0) bsum[blockIdx.x] = bs;
    if (blockIdx.x == 0 && threadIdx.x == 0) {            // this build's header (nothing else writes it in this launch)
        head->approx_total = 0.0; head->total = 0.0; head->e0_sum = 0; head->e0 = 0; head->ns = 0; head->nb = 0; head->nsteps = 0;
        head->fail = force_fail; head->MEAN = 0;
    }
}

// approximate sum in front of this block and of the whole array, from the block sums (same arithmetic in every block and launch)
__device__ __forceinline__ void approx_prefix(const double* __restrict__ bsum, int nblk, int b, double* sm, double* before, double* all) {
    double pb = 0.0, pa = 0.0;
    for (int j = threadIdx.x; j < nblk; j += AL_T) { const double v = bsum[j]; pa += v; if (j < b) pb += v; }
    *before = block_sum_d(pb, sm);
    *all = block_sum_d(pa, sm);
}

struct SumCtx { double before, all; int e0; };
// the map of element i given the approximate running sum INCLUDING it
__device__ __forceinline__ RMap sum_map(double x, double s_approx, int e0, AlHead* head) {
    if (x == 0.0) return rmap_identity();
    int ex;
    (void)frexp(s_approx, &ex);                     // s_approx < 2^ex: ex - e0 bits on the grid
    int k = ex - e0 - 53;
    if (k < 0) k = 0;
    bool sticky = false;
    const i128 c = to_grid(x, e0, &sticky);
    if (k == 0 && sticky) al_fail(head);            // a partial sum finer than the grid: not representable here
    return rne_map(k, c, sticky);
}
// thread-local exclusive fp64 prefix inside the block (fixed order)
__device__ __forceinline__ double block_excl_prefix_d(double mine, double* sd /* AL_T */) {
    const int t = threadIdx.x;
    sd[t] = mine;
    __syncthreads();
    for (int step = 1; step < AL_T; step <<= 1) {
        const double a = (t >= step) ? sd[t - step] : 0.0;
        __syncthreads();
        sd[t] += a;
        __syncthreads();
    }
    const double ex = t ? sd[t - 1] : 0.0;
    __syncthreads();
    return ex;
}

__global__ __launch_bounds__(AL_T) void k_al_sum_reduce(const double* __restrict__ w, size_t K, const double* __restrict__ bsum, int nblk,
                                                        RMap* __restrict__ agg, RMap* __restrict__ tpre, AlHead* __restrict__ head) {
    __shared__ RMap buf[2][AL_T];
    __shared__ double sd[AL_T];
    __shared__ double sm[4];
    double before, all;
    approx_prefix(bsum, nblk, blockIdx.x, sm, &before, &all);
    int etot;
    (void)frexp(all, &etot);
    const int e0 = etot - 53 - AL_SUM_BITS;
    if (!(all > 0.0) || !(all < 1.0e300)) { if (threadIdx.x == 0) al_fail(head); }
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    double x[AL_I], ts = 0.0;
#pragma unroll
    for (int j = 0; j < AL_I; j++) { x[j] = (base + j < K) ? w[base + j] : 0.0; ts += x[j]; }
    double run = before + block_excl_prefix_d(ts, sd);
    RMap m = rmap_identity();
#pragma unroll
    for (int j = 0; j < AL_I; j++) { run += x[j]; m = compose(m, sum_map(x[j], run, e0, head)); }
    RMap ex;
    const RMap inc = block_scan_maps(m, buf, &ex);
    tpre[(size_t)blockIdx.x * AL_T + threadIdx.x] = ex;            // the apply kernel starts from here instead of scanning again
    if (threadIdx.x == AL_T - 1) agg[blockIdx.x] = inc;
    if (blockIdx.x == 0 && threadIdx.x == 0) { head->approx_total = all; head->e0_sum = e0; }
}

__global__ __launch_bounds__(AL_T) void k_al_sum_apply(const double* __restrict__ w, size_t K, const double* __restrict__ bsum, int nblk,
                                                       const RMap* __restrict__ agg, const RMap* __restrict__ tpre, AlHead* __restrict__ head) {
    __shared__ RMap buf[2][AL_T];
    __shared__ double sd[AL_T];
    __shared__ double sm[4];
    double before, all;
    approx_prefix(bsum, nblk, blockIdx.x, sm, &before, &all);
    int etot;
    (void)frexp(all, &etot);
    const int e0 = etot - 53 - AL_SUM_BITS;
    const RMap pre = prefix_of_blocks(agg, blockIdx.x, buf);
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    double x[AL_I], ts = 0.0;
#pragma unroll
    for (int j = 0; j < AL_I; j++) { x[j] = (base + j < K) ? w[base + j] : 0.0; ts += x[j]; }
    double run = before + block_excl_prefix_d(ts, sd);
    RMap mj[AL_I];
#pragma unroll
    for (int j = 0; j < AL_I; j++) { run += x[j]; mj[j] = sum_map(x[j], run, e0, head); }
    const RMap ex = tpre[(size_t)blockIdx.x * AL_T + threadIdx.x];
    i128 v = rapply(ex, rapply(pre, 0));
    bool ok = true;
    double prev = from_grid(v, e0, &ok);
#pragma unroll
    for (int j = 0; j < AL_I; j++) {
        if (base + j < K) {
            v = rapply(mj[j], v);
            const double c = from_grid(v, e0, &ok);
            if (prev + x[j] != c) ok = false;              // the real addition on the claimed operands
            prev = c;
            if (base + j == K - 1) head->total = c;
        }
    }
    if (!ok) al_fail(head);
}

// ---------------------------------------------------------------------------------------------------------------------------
// 2. E = w / total, smalls and bigs in pop order (descending index), their deficits / values on the serving grid
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AL_T) void k_al_classify(const double* __restrict__ w, size_t K, double mean, AlHead* __restrict__ head,
                                                      double* __restrict__ E, unsigned* __restrict__ cnt, double* __restrict__ F,
                                                      uint32_t* __restrict__ A) {
    __shared__ unsigned sc[4];
    const double total = head->total;
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    unsigned c = 0;
#pragma unroll
    for (int j = 0; j < AL_I; j++) {
        const size_t k = base + j;
        if (k < K) {
            const double e = w[k] / total;
            E[k] = e;
            c += (e < mean) ? 1u : 0u;
            F[k] = 1.0;                                 // defaults: unserved smalls, leftover bigs (gsl: A = self, F = 1)
            A[k] = (uint32_t)k;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int em;
        (void)frexp(mean, &em);
        const int e0 = em - 53 - 1;                     // g0 = ulp(mean) / 2
        bool inexact = false;
        head->e0 = e0;
        head->MEAN = to_grid(mean, e0, &inexact);
        if (!(total > 0.0) || !(total < 1.0e300)) al_fail(head);
    }
}

__global__ __launch_bounds__(AL_T) void k_al_lists(const double* __restrict__ E, size_t K, double mean, const unsigned* __restrict__ cnt,
                                                   int nblk, AlHead* __restrict__ head, uint32_t* __restrict__ sidx, u64* __restrict__ dI,
                                                   uint32_t* __restrict__ bidx, i128* __restrict__ VI) {
    __shared__ unsigned su[AL_T];
    __shared__ unsigned s_before, s_all;
    const int t = threadIdx.x;
    // smalls in front of this block / in all: fixed-order sums of the block counts
    unsigned pb = 0, pa = 0;
    for (int j = t; j < nblk; j += AL_T) { const unsigned v = cnt[j]; pa += v; if (j < (int)blockIdx.x) pb += v; }
    su[t] = pb; __syncthreads();
    for (int s = AL_T / 2; s > 0; s >>= 1) { if (t < s) su[t] += su[t + s]; __syncthreads(); }
    if (t == 0) s_before = su[0];
    __syncthreads();
    su[t] = pa; __syncthreads();
    for (int s = AL_T / 2; s > 0; s >>= 1) { if (t < s) su[t] += su[t + s]; __syncthreads(); }
    if (t == 0) s_all = su[0];
    __syncthreads();
    const unsigned ns = s_all, nb = (unsigned)K - ns;
    const int e0 = head->e0;
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)t * AL_I;
    unsigned mine = 0;
    double e[AL_I];
#pragma unroll
    for (int j = 0; j < AL_I; j++) { e[j] = (base + j < K) ? E[base + j] : 0.0; if (base + j < K && e[j] < mean) mine++; }
    // exclusive scan of the per-thread small counts
    su[t] = mine; __syncthreads();
    for (int step = 1; step < AL_T; step <<= 1) {
        const unsigned a = (t >= step) ? su[t - step] : 0u;
        __syncthreads();
        su[t] += a;
        __syncthreads();
    }
    unsigned rs = s_before + su[t] - mine;                        // ascending rank among the smalls
    bool bad = false;
#pragma unroll
    for (int j = 0; j < AL_I; j++) {
        const size_t k = base + j;
        if (k >= K) break;
        if (e[j] < mean) {
            const unsigned pos = ns - 1 - rs;                      // pop order: highest index first
            bool inexact = false;
            const double d = mean - e[j];                          // the rounded deficit the algorithm subtracts
            sidx[pos] = (uint32_t)k;
            dI[pos] = (u64)to_grid(d, e0, &inexact);
            bad = bad || inexact || !(e[j] >= 0.0);
            rs++;
        } else {
            const unsigned rb = (unsigned)k - rs;                  // ascending rank among the bigs
            const unsigned pos = nb - 1 - rb;
            bool inexact = false;
            bidx[pos] = (uint32_t)k;
            VI[pos] = to_grid(e[j], e0, &inexact);
            bad = bad || inexact || !(e[j] <= 2.0);                // (NaN lands here: E < mean is false)
        }
    }
    if (bad) al_fail(head);
    if (blockIdx.x == 0 && t == 0) { head->ns = ns; head->nb = nb; head->nsteps = 0; }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 3. exact prefix sums D (deficits, pop order) and X (excesses E[b] - mean)
// ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ i128 block_sum_i(i128 v, i128* sh /* AL_T */) {
    const int t = threadIdx.x;
    sh[t] = v; __syncthreads();
    for (int s = AL_T / 2; s > 0; s >>= 1) { if (t < s) sh[t] += sh[t + s]; __syncthreads(); }
    const i128 r = sh[0];
    __syncthreads();
    return r;
}
// grid.y = 0: deficits, 1: excesses
__global__ __launch_bounds__(AL_T) void k_al_psum_reduce(const u64* __restrict__ dI, const i128* __restrict__ VI, const AlHead* __restrict__ head,
                                                         i128* __restrict__ bs /* [2][nblk] */, int nblk) {
    __shared__ i128 sh[AL_T];
    const bool bigs = blockIdx.y == 1;
    const size_t n = bigs ? head->nb : head->ns;
    const i128 MEAN = head->MEAN;
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    i128 s = 0;
#pragma unroll
    for (int j = 0; j < AL_I; j++) if (base + j < n) s += bigs ? (VI[base + j] - MEAN) : (i128)dI[base + j];
    const i128 tot = block_sum_i(s, sh);
    if (threadIdx.x == 0) bs[(size_t)blockIdx.y * nblk + blockIdx.x] = tot;
}
__global__ __launch_bounds__(AL_T) void k_al_psum_apply(const u64* __restrict__ dI, const i128* __restrict__ VI, const AlHead* __restrict__ head,
                                                        const i128* __restrict__ bs, int nblk, i128* __restrict__ D, i128* __restrict__ X,
                                                        i128* __restrict__ Dc, i128* __restrict__ Xc /* last sum of every block: coarse index */) {
    __shared__ i128 sh[AL_T];
    const bool bigs = blockIdx.y == 1;
    const size_t n = bigs ? head->nb : head->ns;
    if ((size_t)blockIdx.x * AL_B >= n) return;
    const i128 MEAN = head->MEAN;
    const int t = threadIdx.x;
    i128 pb = 0;
    for (int j = t; j < (int)blockIdx.x; j += AL_T) pb += bs[(size_t)blockIdx.y * nblk + j];
    const i128 before = block_sum_i(pb, sh);
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)t * AL_I;
    i128 v[AL_I], ts = 0;
#pragma unroll
    for (int j = 0; j < AL_I; j++) { v[j] = (base + j < n) ? (bigs ? (VI[base + j] - MEAN) : (i128)dI[base + j]) : (i128)0; ts += v[j]; }
    sh[t] = ts; __syncthreads();
    for (int step = 1; step < AL_T; step <<= 1) {
        const i128 a = (t >= step) ? sh[t - step] : (i128)0;
        __syncthreads();
        sh[t] += a;
        __syncthreads();
    }
    i128 run = before + sh[t] - ts;
    i128* out = bigs ? X : D;
#pragma unroll
    for (int j = 0; j < AL_I; j++) if (base + j < n) { run += v[j]; out[base + j] = run; }
    if (t == AL_T - 1) (bigs ? Xc : Dc)[blockIdx.x] = run;           // (a partly filled last block: the sum of what it holds)
}

// ---------------------------------------------------------------------------------------------------------------------------
// 4. the chain's structure from the exact sums: z[j] = first small (1-based) after which big j (1-based) has fallen below the
//    mean (ns + 1: never), jof[i] = the big serving small i, and the position of every small-step / hand-over in the chain
// ---------------------------------------------------------------------------------------------------------------------------
constexpr unsigned AL_HAND = 0x80000000u;
// one step of the serving chain, everything its map needs (the scans then build maps without a dependent load): a small-step
// subtracts val = the small's deficit at result level k1; a hand-over is -fl(mean - r) at level k1, then fl(val - .) with
// val = E[b'] at level k2.  idx: the small (0-based) resp. AL_HAND | the big handing over (0-based)
struct StepRec { i128 val; uint32_t idx; unsigned char k1, k2; unsigned char pad_[10]; };
// number of elements of the sorted array a[0 .. n) that are < v (strict) resp. <= v: first over the coarse index c (the last
// element of every AL_B block: a few hundred entries every thread probes -- cache hits), then inside one block (16 KB, shared
// with the neighbouring threads' searches).  A plain binary search over 1.6 MB arrays was 17 dependent L2 / fabric round trips
// per thread: 25 us of the build.
template <bool STRICT>
__device__ __forceinline__ unsigned rank_two_level(const i128* __restrict__ a, const i128* __restrict__ c, unsigned n, i128 v) {
    const unsigned nblk = (n + AL_B - 1) / AL_B;
    unsigned lo = 0, hi = nblk;                     // first block whose last element is not before v
    while (lo < hi) { const unsigned mid = (lo + hi) >> 1; const i128 x = c[mid]; if (STRICT ? (x < v) : (x <= v)) lo = mid + 1; else hi = mid; }
    if (lo >= nblk) return n;
    unsigned l2 = lo * AL_B, h2 = l2 + AL_B;
    if (h2 > n) h2 = n;
    while (l2 < h2) { const unsigned mid = (l2 + h2) >> 1; const i128 x = a[mid]; if (STRICT ? (x < v) : (x <= v)) l2 = mid + 1; else h2 = mid; }
    return l2;
}

__global__ __launch_bounds__(AL_T) void k_al_structure(const i128* __restrict__ D, const i128* __restrict__ X, const i128* __restrict__ Dc,
                                                       const i128* __restrict__ Xc, const u64* __restrict__ dI, const i128* __restrict__ VI,
                                                       AlHead* __restrict__ head, uint32_t* __restrict__ z, uint32_t* __restrict__ jof,
                                                       StepRec* __restrict__ step) {
    // weights the earlier launches flagged (non-finite, negative, off the grid): the exact sums are then not sums of what the
    // rank searches assume (a non-finite big has VI = 0: X decreases), slots of `step` would stay unwritten and the serving
    // kernels would follow garbage indices -- nothing below this point runs, the caller builds the table on the host.
    // (the flag was written by earlier LAUNCHES: visible here; nothing in this launch or the next sets it before they read it)
    if (head->fail) return;
    const unsigned ns = head->ns, nb = head->nb;
    const size_t t = (size_t)blockIdx.x * AL_T + threadIdx.x;
    unsigned pos = 0;                                          // 1-based position of this thread's step in the chain (0: none)
    if (ns != 0 && nb != 0 && t < (size_t)ns + nb) {
        if (t < ns) {
            const unsigned i = (unsigned)t + 1;                    // small i (1-based): served by big 1 + #{j : X_j < D_{i-1}}
            const i128 dprev = (i >= 2) ? D[i - 2] : (i128)0;
            const unsigned j = rank_two_level<true>(X, Xc, nb, dprev) + 1;      // lower bound of dprev in X
            jof[i - 1] = j;
            if (j <= nb) {
                pos = i + j - 1;
                const i128 star = head->MEAN + X[j - 1] - D[i - 1];               // exact value after the step: its binade is the level
                StepRec r;
                r.val = (i128)dI[i - 1]; r.idx = i - 1; r.k1 = (unsigned char)level_of(star < 0 ? (i128)0 : star); r.k2 = 0;
                step[pos - 1] = r;
            }
        } else {
            const unsigned j = (unsigned)(t - ns) + 1;             // big j (1-based): z = 1 + #{i : D_i <= X_j}
            const i128 xj = X[j - 1];
            const unsigned zz = rank_two_level<false>(D, Dc, ns, xj) + 1;       // upper bound of xj in D
            z[j - 1] = zz;
            if (zz <= ns && j < nb) {
                pos = zz + j;
                const i128 dz = D[zz - 1];
                const i128 dd = dz - xj;                                          // mean - r, exact
                const i128 star = head->MEAN + X[j] - dz;
                StepRec r;
                r.val = VI[j]; r.idx = AL_HAND | (j - 1);
                r.k1 = (unsigned char)level_of(dd < 0 ? (i128)0 : dd); r.k2 = (unsigned char)level_of(star < 0 ? (i128)0 : star);
                step[pos - 1] = r;
            }
        }
    }
    // the chain's length = the largest position: one atomic per work-group (one per thread on a single word was most of this
    // kernel's 25 us)
    __shared__ unsigned smax[4];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned v = __shfl_xor(pos, o, 64); pos = v > pos ? v : pos; }
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = pos;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = smax[0];
        for (int w = 1; w < 4; w++) m = smax[w] > m ? smax[w] : m;
        if (m) atomicMax(&head->nsteps, m);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// 5. the serving chain: maps, scan, verification, table entries
// ---------------------------------------------------------------------------------------------------------------------------
struct ServeArgs {
    const double* E; const uint32_t* sidx; const uint32_t* bidx; const u64* dI; const i128* VI; const i128* D; const i128* X;
    const uint32_t* z; const uint32_t* jof; const StepRec* step;
};
__device__ __forceinline__ RMap serve_map(const StepRec& r, i128 MEAN) {
    if (!(r.idx & AL_HAND)) return rne_map(r.k1, -r.val, false);
    const RMap m1 = rne_map(r.k1, -MEAN, false);              // r -> -fl(mean - r) (round-half-even is odd-symmetric)
    const RMap m2 = rne_map(r.k2, r.val, false);              // -> fl(E[b'] - fl(mean - r))
    return compose(m1, m2);
}

__global__ __launch_bounds__(AL_T) void k_al_serve_reduce(ServeArgs a, const AlHead* __restrict__ head, RMap* __restrict__ agg,
                                                          RMap* __restrict__ tpre) {
    __shared__ RMap buf[2][AL_T];
    const unsigned n = head->nsteps;
    if (head->fail) return;                                        // (see k_al_structure)
    if ((size_t)blockIdx.x * AL_B >= n) { if (threadIdx.x == 0) agg[blockIdx.x] = rmap_identity(); return; }
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    const i128 MEAN = head->MEAN;
    StepRec rec[AL_I];
#pragma unroll
    for (int j = 0; j < AL_I; j++) if (base + j < n) rec[j] = a.step[base + j];
    RMap m = rmap_identity();
#pragma unroll
    for (int j = 0; j < AL_I; j++) if (base + j < n) m = compose(m, serve_map(rec[j], MEAN));
    RMap ex;
    const RMap inc = block_scan_maps(m, buf, &ex);
    tpre[(size_t)blockIdx.x * AL_T + threadIdx.x] = ex;
    if (threadIdx.x == AL_T - 1) agg[blockIdx.x] = inc;
}

__global__ __launch_bounds__(AL_T) void k_al_serve_apply(ServeArgs a, AlHead* __restrict__ head, const RMap* __restrict__ agg,
                                                         const RMap* __restrict__ tpre, double mean,
                                                         double dK, double* __restrict__ F, uint32_t* __restrict__ A, int* __restrict__ fail_out,
                                                         int* __restrict__ fail_pin) {
    __shared__ RMap buf[2][AL_T];
    const unsigned n = head->nsteps, ns = head->ns, nb = head->nb;
    const bool last_block = ((size_t)(blockIdx.x + 1) * AL_B >= n);
    if ((size_t)blockIdx.x * AL_B >= n && blockIdx.x != 0) return;
    if (n == 0 || head->fail) {              // no small or no big: the defaults are the table; or a build flagged before this launch
        if (blockIdx.x == 0 && threadIdx.x == 0) { *fail_out = head->fail; if (fail_pin) *fail_pin = head->fail; }
        return;
    }
    const RMap pre = prefix_of_blocks(agg, blockIdx.x, buf);
    const size_t base = (size_t)blockIdx.x * AL_B + (size_t)threadIdx.x * AL_I;
    const i128 MEAN = head->MEAN;
    StepRec rec[AL_I];
    RMap mj[AL_I];
#pragma unroll
    for (int j = 0; j < AL_I; j++) {
        if (base + j < n) { rec[j] = a.step[base + j]; mj[j] = serve_map(rec[j], MEAN); }
        else { rec[j].idx = 0; mj[j] = rmap_identity(); }
    }
    const RMap ex = tpre[(size_t)blockIdx.x * AL_T + threadIdx.x];
    const int e0 = head->e0;
    i128 v = rapply(ex, rapply(pre, a.VI[0]));                     // the chain starts at E[b_1]
    bool ok = true;
    double prev = from_grid(v, e0, &ok);
#pragma unroll
    for (int j = 0; j < AL_I; j++) {
        const size_t t = base + j;
        if (t >= n) break;
        v = rapply(mj[j], v);
        const double c = from_grid(v, e0, &ok);
        const uint32_t s = rec[j].idx;
        unsigned jcur;
        bool below, at_last_small;
        if (!(s & AL_HAND)) {
            const unsigned ii = s;
            jcur = a.jof[ii];
            const uint32_t sm = a.sidx[ii];
            const double es = a.E[sm];
            if (prev - (mean - es) != c) ok = false;              // eb -= mean - E[s], the real operations
            below = (a.z[jcur - 1] == ii + 1);
            at_last_small = (ii + 1 == ns);
            A[sm] = a.bidx[jcur - 1];
            F[sm] = dK * es;
        } else {
            const unsigned j0 = (s & ~AL_HAND) + 1;                // big j0 hands over to big j0 + 1
            jcur = j0 + 1;
            const uint32_t bo = a.bidx[j0 - 1], bn = a.bidx[j0];
            const double dd = mean - prev;
            if (a.E[bn] - dd != c) ok = false;
            const unsigned zz = a.z[j0 - 1];
            below = (a.z[j0] == zz);
            at_last_small = (zz == ns);
            A[bo] = bn;
            F[bo] = dK * prev;
        }
        // the comparisons the loop makes (the last big after the last small ends as its own alias either way)
        if (!(jcur == nb && at_last_small)) {
            if (below) { if (!(c < mean)) ok = false; }
            else if (!(c > mean)) ok = false;                      // (== mean, GSL's "exactly full" branch: left to the host)
        }
        prev = c;
    }
    if (!ok) al_fail(head);
    (void)last_block;
}
__global__ void k_al_flag(const AlHead* __restrict__ head, int* __restrict__ fail_out, int* __restrict__ fail_pin) {
    if (threadIdx.x == 0) { *fail_out = head->fail; if (fail_pin) *fail_pin = head->fail; }
}

}  // namespace

#if ABC_AL_I == 4
int launch_alias_build_dev_i2(abc_ctx* ctx, const double* w, size_t K, double* F, uint32_t* A, int* fail_dev, int* fail_pin,
                              const int** verdict_src);
int launch_alias_build_dev_i4(abc_ctx* ctx, const double* w, size_t K, double* F, uint32_t* A, int* fail_dev, int* fail_pin,
                              const int** verdict_src);
static size_t alias_small_k() {             // tables up to this size: two elements per thread
    static const size_t k = abc_diag_env("ABC_ALIAS_SMALL_K") ? (size_t)atoll(abc_diag_env("ABC_ALIAS_SMALL_K")) : ABC_ALIAS_DEV_SMALL_K;
    return k;
}
size_t abc_alias_dev_need(size_t K) {       // (the larger of the two variants' needs: the smaller work-groups')
    constexpr size_t B = AL_T * 2;
    const size_t nblk = (K + B - 1) / B + 1, nblk2 = (2 * K + B - 1) / B + 1;
    return K * (8 + 4 + 4 + 8 + 16 + 16 + 16 + 4 + 4 + 64) + nblk * (8 + 4 + 64 + 64) + nblk2 * 64 * (AL_T + 1) + sizeof(AlHead) + 64 * 256;
}
int launch_alias_build_dev(abc_ctx* ctx, const double* w, size_t K, double* F, uint32_t* A, int* fail_dev, int* fail_pin,
                           const int** verdict_src) {
    if (K <= alias_small_k()) return launch_alias_build_dev_i2(ctx, w, K, F, A, fail_dev, fail_pin, verdict_src);
    return launch_alias_build_dev_i4(ctx, w, K, F, A, fail_dev, fail_pin, verdict_src);
}
#endif

// F (K doubles, cut-off fractions WITHOUT the KNUTH_CONVENTION map, as alias_preproc(..., knuth = false)) and A (K uint32) on the
// device; *fail_dev (and *fail_pin, optional, pinned) = 1 when the table must not be used (the caller builds it on the host)
int ABC_AL_FN(launch_alias_build_dev)(abc_ctx* ctx, const double* w, size_t K, double* F, uint32_t* A, int* fail_dev, int* fail_pin,
                                      const int** verdict_src) {
    if (K == 0 || K > (size_t)AL_MAXBLK * AL_B / 2) ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "device alias build: K = %zu", K);
    const int nblk = (int)((K + AL_B - 1) / AL_B), nblk2 = (int)((2 * K + AL_B - 1) / AL_B);
    AlHead* head = (AlHead*)abc_ws_alloc(ctx, sizeof(AlHead));
    double* bsum = (double*)abc_ws_alloc(ctx, (size_t)nblk * 8);
    RMap* agg = (RMap*)abc_ws_alloc(ctx, (size_t)(nblk2 > nblk ? nblk2 : nblk) * sizeof(RMap));
    RMap* tpre = (RMap*)abc_ws_alloc(ctx, (size_t)(nblk2 > nblk ? nblk2 : nblk) * AL_T * sizeof(RMap));
    double* E = (double*)abc_ws_alloc(ctx, K * 8);
    unsigned* cnt = (unsigned*)abc_ws_alloc(ctx, (size_t)nblk * 4);
    uint32_t* sidx = (uint32_t*)abc_ws_alloc(ctx, K * 4);
    uint32_t* bidx = (uint32_t*)abc_ws_alloc(ctx, K * 4);
    u64* dI = (u64*)abc_ws_alloc(ctx, K * 8);
    i128* VI = (i128*)abc_ws_alloc(ctx, K * 16);
    i128* D = (i128*)abc_ws_alloc(ctx, K * 16);
    i128* X = (i128*)abc_ws_alloc(ctx, K * 16);
    i128* bs = (i128*)abc_ws_alloc(ctx, (size_t)4 * nblk * 16);          // block sums of both arrays, then their coarse indices
    i128* Dc = bs ? bs + (size_t)2 * nblk : nullptr;
    i128* Xc = bs ? bs + (size_t)3 * nblk : nullptr;
    uint32_t* z = (uint32_t*)abc_ws_alloc(ctx, K * 4);
    uint32_t* jof = (uint32_t*)abc_ws_alloc(ctx, K * 4);
    StepRec* step = (StepRec*)abc_ws_alloc(ctx, 2 * K * sizeof(StepRec));
    if (!head || !bsum || !agg || !tpre || !E || !cnt || !sidx || !bidx || !dI || !VI || !D || !X || !bs || !z || !jof || !step)
        ABC_FAIL(ctx, ABC_ERR_NOMEM, "device alias build: workspace exhausted");
    const double mean = 1.0 / (double)K, dK = (double)K;
    hipStream_t st = ctx->stream;
    // ABC_ALIAS_FORCE_FAIL (tests): the build reports failure although it verified, so the callers' host fall-backs can be exercised
    const int force_fail = abc_diag_env("ABC_ALIAS_FORCE_FAIL") ? 1 : 0;
    hipLaunchKernelGGL(k_al_bsum, dim3(nblk), dim3(AL_T), 0, st, w, K, bsum, head, force_fail);
    hipLaunchKernelGGL(k_al_sum_reduce, dim3(nblk), dim3(AL_T), 0, st, w, K, (const double*)bsum, nblk, agg, tpre, head);
    hipLaunchKernelGGL(k_al_sum_apply, dim3(nblk), dim3(AL_T), 0, st, w, K, (const double*)bsum, nblk, (const RMap*)agg, (const RMap*)tpre, head);
    hipLaunchKernelGGL(k_al_classify, dim3(nblk), dim3(AL_T), 0, st, w, K, mean, head, E, cnt, F, A);
    hipLaunchKernelGGL(k_al_lists, dim3(nblk), dim3(AL_T), 0, st, (const double*)E, K, mean, (const unsigned*)cnt, nblk, head, sidx, dI, bidx, VI);
    hipLaunchKernelGGL(k_al_psum_reduce, dim3(nblk, 2), dim3(AL_T), 0, st, (const u64*)dI, (const i128*)VI, (const AlHead*)head, bs, nblk);
    hipLaunchKernelGGL(k_al_psum_apply, dim3(nblk, 2), dim3(AL_T), 0, st, (const u64*)dI, (const i128*)VI, (const AlHead*)head,
                       (const i128*)bs, nblk, D, X, Dc, Xc);
    hipLaunchKernelGGL(k_al_structure, dim3((unsigned)((K + AL_T - 1) / AL_T)), dim3(AL_T), 0, st, (const i128*)D, (const i128*)X, (const i128*)Dc, (const i128*)Xc, (const u64*)dI, (const i128*)VI, head, z, jof, step);
    ServeArgs a = {E, sidx, bidx, dI, VI, D, X, z, jof, step};
    hipLaunchKernelGGL(k_al_serve_reduce, dim3(nblk2), dim3(AL_T), 0, st, a, (const AlHead*)head, agg, tpre);
    hipLaunchKernelGGL(k_al_serve_apply, dim3(nblk2), dim3(AL_T), 0, st, a, head, (const RMap*)agg, (const RMap*)tpre, mean, dK, F, A, fail_dev, fail_pin);
    if (verdict_src) *verdict_src = &head->fail;          // the caller's next kernel publishes the verdict (k_alias_draw)
    else hipLaunchKernelGGL(k_al_flag, dim3(1), dim3(64), 0, st, (const AlHead*)head, fail_dev, fail_pin);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
