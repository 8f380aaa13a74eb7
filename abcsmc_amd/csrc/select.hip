// Selection of the K particles closest to the observation, in ascending (distance, index) order.
// Replaces [PLS] ordered() = ranker.h order() (AbcUtil.cpp:420,457; lib/ranker.h:46-53) followed by
// the caller's truncation to the predictive-prior size (AbcSmc.cpp:645-646): only the first K
// entries of the argsort are ever consumed, so this is an exact radix SELECT of the K-th smallest
// key + a stable compaction of the winners + a stable LSD radix SORT of those K (key, index) pairs.
// Integer/byte work, HBM-bound, bit-exact: keys are the IEEE-754 bit patterns of the distances
// mapped to an order-preserving uint64; ties are broken by particle index (declared).
#include "abc_internal.h"

namespace {

struct SelState {
    unsigned long long prefix;  // bits decided so far (high bits)
    unsigned long long mask;    // which bits are decided
    unsigned long long krem;    // 1-based rank still to find inside the current prefix bucket
    unsigned long long n_less;  // #keys strictly below the threshold (valid after the last pass)
    unsigned long long ties;    // #keys equal to the threshold to take (lowest indices first)
};

__device__ __forceinline__ unsigned long long key_of(double d) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dist_of(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

constexpr int SEL_BITS = 11;
constexpr int SEL_BINS = 1 << SEL_BITS;

__global__ void k_sel_init(SelState* st, unsigned long long K, unsigned int* hist) {
    for (int i = threadIdx.x; i < SEL_BINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) { st->prefix = 0; st->mask = 0; st->krem = K; st->n_less = 0; st->ties = 0; }
}

// histogram of the digit [shift, shift+nbits) over keys matching the decided prefix
__global__ __launch_bounds__(256) void k_sel_hist(const double* __restrict__ dist, size_t n,
                                                  const SelState* __restrict__ st, int shift, int nbits,
                                                  unsigned int* __restrict__ hist) {
    __shared__ unsigned int lh[SEL_BINS];
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, mask = st->mask;
    const unsigned int dm = (1u << nbits) - 1u;
    const size_t stride = (size_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const size_t nround = (n + stride - 1) / stride;      // uniform trip count: the ballots need every lane
    for (size_t it = 0; it < nround; it++) {
        const size_t i = it * stride + (size_t)blockIdx.x * 256 + threadIdx.x;
        const unsigned long long k = (i < n) ? key_of(dist[i]) : 0ull;
        const bool in = (i < n) && ((k & mask) == prefix);
        const unsigned int d = (unsigned int)(k >> shift) & dm;
        // distances cluster in a few digits: aggregate equal digits inside the wave, one LDS add per group
        unsigned long long peers = __ballot(in);
        for (int b = 0; b < nbits; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        if (in && (peers & lt_mask) == 0) atomicAdd(&lh[d], (unsigned int)__popcll(peers));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) {
        const unsigned int c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// find the digit whose cumulative count reaches krem and extend the decided prefix by it: the whole work-group cooperates
// (parallel scan of the bin counts); every thread returns the new state
__device__ __forceinline__ SelState sel_pick_block(const unsigned int* __restrict__ hist, SelState st, int shift, int nbits,
                                                   int last, unsigned long long K, unsigned long long* csum /* 256 */,
                                                   int* found_digit, unsigned long long* found_below) {
    const int t = threadIdx.x;
    const int bins = 1 << nbits;
    const int per = (bins + 255) / 256;
    unsigned long long loc = 0;
    for (int j = 0; j < per; j++) { const int b = t * per + j; if (b < bins) loc += hist[b]; }
    // exclusive scan of the 256 per-thread counts: wave scan by shuffles, then the four wave totals
    const int lane = t & 63, wave = t >> 6;
    unsigned long long inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) csum[wave] = inc;
    __syncthreads();
    unsigned long long run = inc - loc;
    for (int w = 0; w < wave; w++) run += csum[w];
    const unsigned long long krem = st.krem;
    for (int j = 0; j < per; j++) {
        const int b = t * per + j;
        if (b < bins) {
            const unsigned long long c = hist[b];
            if (run < krem && krem <= run + c) { *found_digit = b; *found_below = run; }
            run += c;
        }
    }
    __syncthreads();
    const unsigned long long dm = ((1ull << nbits) - 1ull) << shift;
    st.prefix |= ((unsigned long long)*found_digit) << shift;
    st.mask |= dm;
    st.krem = krem - *found_below;
    if (last) { st.ties = st.krem; st.n_less = K - st.krem; }
    __syncthreads();                       // csum / found_* may be reused by the caller
    return st;
}

// one work-group: the pick of the stage-wise (distributed) protocol; updates the state in place and clears the histogram
__global__ __launch_bounds__(256) void k_sel_pick(SelState* st, int shift, int nbits, unsigned int* hist,
                                                  int last, unsigned long long K) {
    __shared__ unsigned long long csum[256];
    __shared__ int found_digit;
    __shared__ unsigned long long found_below;
    const SelState nst = sel_pick_block(hist, *st, shift, nbits, last, K, csum, &found_digit, &found_below);
    if (threadIdx.x == 0) *st = nst;
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) hist[i] = 0;
}

// ---- single-GPU selection without the pick launches --------------------------------------------------------------
// The six histogram passes each get their OWN histogram (zeroed once), and every work-group of pass p first repeats
// pass p-1's pick on that finished histogram -- the same arithmetic in every block, hence the same state everywhere, no
// synchronisation -- before it counts its keys.  The last pick moves into the first compaction kernel.  Saves the six
// one-work-group k_sel_pick launches (and their gaps) of the stage-wise path, which the distributed driver keeps because
// it all-reduces the histograms between hist and pick.
__global__ void k_sel_init_fused(SelState* st_arr /* 7 */, unsigned long long K, unsigned int* hist_all /* 6 x SEL_BINS */) {
    for (int i = threadIdx.x; i < 6 * SEL_BINS; i += blockDim.x) hist_all[i] = 0;
    if (threadIdx.x == 0) { st_arr[0].prefix = 0; st_arr[0].mask = 0; st_arr[0].krem = K; st_arr[0].n_less = 0; st_arr[0].ties = 0; }
}

// pass p: st_arr[p] = state before this pass (= after the picks of passes 0..p-1); hist_all + p * SEL_BINS receives the counts
__global__ __launch_bounds__(256) void k_sel_hist_fused(const double* __restrict__ dist, size_t n, SelState* __restrict__ st_arr,
                                                        int p, int shift_prev, int nbits_prev, int shift, int nbits,
                                                        unsigned int* __restrict__ hist_all, unsigned long long K) {
    __shared__ unsigned int lh[SEL_BINS];
    __shared__ unsigned long long csum[256];
    __shared__ int found_digit;
    __shared__ unsigned long long found_below;
    SelState st = st_arr[p > 0 ? p - 1 : 0];
    if (p > 0) {
        st = sel_pick_block(hist_all + (size_t)(p - 1) * SEL_BINS, st, shift_prev, nbits_prev, 0, K, csum, &found_digit, &found_below);
        if (threadIdx.x == 0) st_arr[p] = st;          // every block writes the same value
    }
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    unsigned int* hist = hist_all + (size_t)p * SEL_BINS;
    const unsigned long long prefix = st.prefix, mask = st.mask;
    const unsigned int dm = (1u << nbits) - 1u;
    const size_t stride = (size_t)gridDim.x * 256;
    const int lane = threadIdx.x & 63;
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const size_t nround = (n + stride - 1) / stride;      // uniform trip count: the ballots need every lane
    for (size_t it = 0; it < nround; it++) {
        const size_t i = it * stride + (size_t)blockIdx.x * 256 + threadIdx.x;
        const unsigned long long k = (i < n) ? key_of(dist[i]) : 0ull;
        const bool in = (i < n) && ((k & mask) == prefix);
        const unsigned int d = (unsigned int)(k >> shift) & dm;
        unsigned long long peers = __ballot(in);
        for (int b = 0; b < nbits; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        if (in && (peers & lt_mask) == 0) atomicAdd(&lh[d], (unsigned int)__popcll(peers));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SEL_BINS; i += 256) {
        const unsigned int c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}


// ---- single-GPU selection by sampled range + linear bins (round 2) ------------------------------------------------------
// The radix select above reads the distances eight times (six digit passes, count, write) and the winners are then sorted
// by a chunk sort + rank merge: 105 + 97 us at N = 1e6, K = 1e5, all of it launch- and latency-bound.  Here:
//   k_bs_sample   one work-group: 4096 evenly spaced keys into LDS, their minimum (lo) and -- by an LDS radix select on the
//                 leading 24 bits -- an upper bound (hi) of the sample key of rank K/N * 4096 + 4 sigma + 8: the K-th
//                 smallest of the whole set lies below hi unless the sample is atypical (~3e-5) or the data are degenerate
//   k_bs_hist     NB = 4096 linear bins of the key range [lo, hi] (keys are order-preserving integers, so (key - lo) >> shift
//                 is monotone); per-block LDS histogram, flushed with global atomics.  Keys above hi are not counted.
//   k_bs_scan     exclusive scan of the bin counts; the bin b* holding the K-th key; checks (below): sets `fail` if a rule breaks
//   k_bs_scatter  every key of a bin <= b* goes to its bin's range of a (key, index) pair buffer (unordered inside the bin)
//   k_bs_sort     one work-group per bin: bitonic sort by (key, index) in LDS, written to its final position; of bin b* only
//                 the first K - #below are kept -- which is exactly "ties by lowest index" -- distances converted on the way.
// The result is the same array the radix select + stable sort produce (ascending (key, index) is a total order).
// `fail` (any bin above BS_CAP keys, or fewer than K keys at or below hi): k_bs_sort writes a harmless result (indices
// idx_base .. idx_base + K - 1) and the host, when it next synchronises, repeats the selection with the radix path.
constexpr int BS_NB = 4096;          // bins up to K = 2^18 ...
constexpr int BS_NB_BIG = 16384;     // ... and beyond, up to 2^20 (round 4: K = 1e6 of configs[3] took the radix select and eight LSD sort
                                     // passes, 2.7 ms; the same average bin filling -- 64 of the 1024 keys a bin may hold -- here)
constexpr int BS_S = 4096;           // sampled keys
constexpr int BS_CAP = 1024;         // keys one work-group sorts
constexpr int BS_CSTRIDE = 32;       // the scatter's per-bin cursors sit 128 bytes apart: neighbouring (equally busy) bins on the
                                     // same cache line serialised their atomics (63 us for 1e5 winners; 4096 x 4-byte cursors packed)
struct BinSel {
    unsigned long long lo, hi;
    int shift;
    int bstar;                       // bin of the K-th key
    unsigned long long need;         // keys kept of bin b*
    int fail;
    int pad_;
};

__global__ __launch_bounds__(1024) void k_bs_sample(const double* __restrict__ dist, size_t n, unsigned long long K,
                                                    BinSel* __restrict__ bs, unsigned int* __restrict__ hist /* nbins + 1 */,
                                                    int* __restrict__ fail_flag, int nbins) {
    __shared__ unsigned long long sk[BS_S];
    __shared__ unsigned int h[256];
    __shared__ unsigned long long red[16];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned int s_rank;
    const int t = threadIdx.x;
    for (int i = t; i < nbins + 1; i += 1024) hist[i] = 0;
    const size_t stride = n / BS_S;
    unsigned long long mn = ~0ull;
    {   // (the thread's samples are fetched TOGETHER: one at a time each was a trip to HBM / L2 of its own on an otherwise empty chip)
        static_assert(BS_S % 1024 == 0 && BS_S / 1024 <= 8, "samples per thread");
        double dv[BS_S / 1024];
#pragma unroll
        for (int j = 0; j < BS_S / 1024; j++) dv[j] = dist[(size_t)(t + 1024 * j) * stride + stride / 2];
#pragma unroll
        for (int j = 0; j < BS_S / 1024; j++) {
            const unsigned long long k = key_of(dv[j]);
            sk[t + 1024 * j] = k;
            mn = k < mn ? k : mn;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { const unsigned long long v = __shfl_xor(mn, o, 64); mn = v < mn ? v : mn; }
    if ((t & 63) == 0) red[t >> 6] = mn;
    // rank (0-based) of the upper key among the samples: expected position of the K-th + 4 standard deviations + 8
    const double f = (double)K / (double)n;
    double qd = f * BS_S + 4.0 * sqrt(BS_S * f * (1.0 - f)) + 8.0;
    unsigned int q = (qd >= (double)(BS_S - 1)) ? BS_S - 1 : (unsigned int)qd;
    if (t == 0) { s_prefix = 0; s_rank = q; }
    __syncthreads();
    for (int w = 0; w < 16; w++) mn = red[w] < mn ? red[w] : mn;
    // LDS radix select of the sample's q-th key, most significant bits first.  hi only has to lie at or above it: three passes
    // decide its top 24 bits (sign, exponent, 12 mantissa bits), the rest is filled with ones
    unsigned long long mask = 0;
    for (int pass = 0; pass < 3; pass++) {
        const int shift = 56 - 8 * pass;
        if (t < 256) h[t] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int i = t; i < BS_S; i += 1024)
            if ((sk[i] & mask) == prefix) atomicAdd(&h[(unsigned int)(sk[i] >> shift) & 255u], 1u);
        __syncthreads();
        if (t < 64) {          // one wave: four digits per lane, scan, pick
            unsigned int c[4], sum = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) { c[j] = h[4 * t + j]; sum += c[j]; }
            unsigned int inc = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o, 64); if (t >= o) inc += u; }
            unsigned int run = inc - sum;
            const unsigned int r = s_rank;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (run <= r && r < run + c[j]) { s_prefix = prefix | ((unsigned long long)(4 * t + j) << shift); s_rank = r - run; }
                run += c[j];
            }
        }
        mask |= 255ull << shift;
        __syncthreads();
    }
    if (t == 0) {
        const unsigned long long lo = mn, hi = s_prefix | ~mask;
        const unsigned long long span = hi - lo;
        int shift = 0;
        while (shift < 63 && (span >> shift) >= (unsigned long long)(nbins - 1)) shift++;
        bs->lo = lo; bs->hi = hi; bs->shift = shift; bs->bstar = 0; bs->need = 0; bs->fail = 0; bs->pad_ = 0;
        *fail_flag = 0;
    }
}

__device__ __forceinline__ int bs_bin(unsigned long long k, unsigned long long lo, int shift) {
    return (k <= lo) ? 0 : (int)((k - lo) >> shift);
}

__global__ __launch_bounds__(256) void k_bs_hist(const double* __restrict__ dist, size_t n, const BinSel* __restrict__ bs,
                                                 unsigned int* __restrict__ hist, unsigned int* __restrict__ cursor, int nbins) {
    extern __shared__ unsigned int lh[];              // nbins
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)nbins * BS_CSTRIDE; i += (size_t)gridDim.x * 256) cursor[i] = 0;
    for (int i = threadIdx.x; i < nbins; i += 256) lh[i] = 0;
    __syncthreads();
    const unsigned long long lo = bs->lo, hi = bs->hi;
    const int shift = bs->shift;
    // eight keys in flight per thread (a load directly in front of the LDS atomic that depends on it is one memory latency per key)
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n; i0 += 8 * stride) {
        double d[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const size_t i = i0 + u * stride; d[u] = dist[i < n ? i : i0]; }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned long long k = key_of(d[u]);
            if (i0 + u * stride < n && k <= hi) atomicAdd(&lh[bs_bin(k, lo, shift)], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nbins; i += 256) {
        const unsigned int c = lh[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// hist[b] -> exclusive offsets in place (hist[nbins] = total); b*, need, fail
template <int BPT>                           // bins per thread: nbins / 1024
__global__ __launch_bounds__(1024) void k_bs_scan(BinSel* __restrict__ bs, unsigned int* __restrict__ hist, unsigned long long K,
                                                  int* __restrict__ fail_flag) {
    __shared__ unsigned int wsum[16];
    __shared__ int s_fail;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_fail = 0;
    unsigned int c[BPT], sum = 0;
#pragma unroll
    for (int j = 0; j < BPT; j++) { c[j] = hist[BPT * t + j]; sum += c[j]; }
    unsigned int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int run = inc - sum;
    for (int w = 0; w < wave; w++) run += wsum[w];
    unsigned int tot = 0;
    for (int w = 0; w < 16; w++) tot += wsum[w];
    // keys of bins <= b* are sorted by one work-group each: none may exceed BS_CAP
#pragma unroll
    for (int j = 0; j < BPT; j++) {
        const unsigned int off = run;
        hist[BPT * t + j] = off;
        if ((unsigned long long)off < K && K <= (unsigned long long)off + c[j]) { bs->bstar = BPT * t + j; bs->need = K - off; }
        if ((unsigned long long)off < K && c[j] > (unsigned int)BS_CAP) s_fail = 1;
        run += c[j];
    }
    if (t == 0) hist[BPT * 1024] = tot;
    __syncthreads();
    if (t == 0) {
        const int fail = (s_fail || (unsigned long long)tot < K) ? 1 : 0;
        bs->fail = fail;
        *fail_flag = fail;
    }
}

__global__ __launch_bounds__(256) void k_bs_scatter(const double* __restrict__ dist, size_t n, const BinSel* __restrict__ bs,
                                                    const unsigned int* __restrict__ offs, unsigned int* __restrict__ cursor,
                                                    unsigned long long idx_base, unsigned long long* __restrict__ tkey,
                                                    unsigned long long* __restrict__ tidx) {
    if (bs->fail) return;
    const unsigned long long lo = bs->lo, hi = bs->hi;
    const int shift = bs->shift, bstar = bs->bstar;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = key_of(dist[i]);
    if (k > hi) return;
    const int b = bs_bin(k, lo, shift);
    if (b > bstar) return;
    const unsigned int p = offs[b] + atomicAdd(&cursor[(size_t)b * BS_CSTRIDE], 1u);
    tkey[p] = k;
    tidx[p] = idx_base + i;
}

__global__ __launch_bounds__(256) void k_bs_sort(const BinSel* __restrict__ bs, const unsigned int* __restrict__ offs,
                                                 const unsigned long long* __restrict__ tkey,
                                                 const unsigned long long* __restrict__ tidx, unsigned long long K,
                                                 unsigned long long idx_base, unsigned long long* __restrict__ oidx,
                                                 double* __restrict__ odist) {
    __shared__ unsigned long long sk[BS_CAP];
    __shared__ unsigned long long si[BS_CAP];
    const int b = blockIdx.x, t = threadIdx.x;
    if (bs->fail) {          // harmless, in-range result; the host repeats the selection with the radix path
        for (unsigned long long i = (unsigned long long)b * 256 + t; i < K; i += (unsigned long long)gridDim.x * 256) {
            oidx[i] = idx_base + i;
            if (odist) odist[i] = 0.0;
        }
        return;
    }
    if (b > bs->bstar) return;
    const unsigned int o0 = offs[b], cnt = offs[b + 1] - o0;
    if (cnt == 0) return;
    const unsigned int keep = (b == bs->bstar) ? (unsigned int)bs->need : cnt;
    unsigned int n2 = 2;
    while (n2 < cnt) n2 <<= 1;
    for (unsigned int e = t; e < n2; e += 256) {
        sk[e] = (e < cnt) ? tkey[o0 + e] : ~0ull;
        si[e] = (e < cnt) ? tidx[o0 + e] : ~0ull;
    }
    __syncthreads();
    for (unsigned int k = 2; k <= n2; k <<= 1)
        for (unsigned int j = k >> 1; j > 0; j >>= 1) {
            for (unsigned int p = t; p < n2 / 2; p += 256) {
                const unsigned int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const unsigned long long ka = sk[i], kb = sk[i + j], ia = si[i], ib = si[i + j];
                const bool gt = (ka > kb) || (ka == kb && ia > ib);
                if (gt == ((i & k) == 0)) { sk[i] = kb; sk[i + j] = ka; si[i] = ib; si[i + j] = ia; }
            }
            __syncthreads();
        }
    for (unsigned int e = t; e < keep; e += 256) {
        oidx[o0 + e] = si[e];
        if (odist) odist[o0 + e] = dist_of(sk[e]);
    }
}

constexpr int CP_ITEMS = 8;
constexpr int CP_CHUNK = 256 * CP_ITEMS;

// per-chunk counts of (key < T) and (key == T)
// fused path (last_hist != NULL): st points at the state before the last pick, which every block repeats here; the result
// goes to st[1] for k_cp_write
__global__ __launch_bounds__(256) void k_cp_count(const double* __restrict__ dist, size_t n,
                                                  SelState* __restrict__ st,
                                                  unsigned int* __restrict__ cnt /* [2][nb] */, int nb,
                                                  const unsigned int* __restrict__ last_hist = nullptr, int shift = 0,
                                                  int nbits = 0, unsigned long long K = 0) {
    __shared__ unsigned int sl[4], se[4];
    __shared__ unsigned long long csum[256];
    __shared__ int found_digit;
    __shared__ unsigned long long found_below;
    unsigned long long T;
    if (last_hist) {
        const SelState fin = sel_pick_block(last_hist, st[0], shift, nbits, 1, K, csum, &found_digit, &found_below);
        if (threadIdx.x == 0) st[1] = fin;             // every block writes the same value
        T = fin.prefix;
    } else {
        T = st->prefix;
    }
    const size_t base = (size_t)blockIdx.x * CP_CHUNK + (size_t)threadIdx.x * CP_ITEMS;
    unsigned int l = 0, e = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; j++) {
        const size_t i = base + j;
        if (i < n) { const unsigned long long k = key_of(dist[i]); l += (k < T); e += (k == T); }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { l += __shfl_xor(l, o, 64); e += __shfl_xor(e, o, 64); }
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = l; se[threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt[blockIdx.x] = sl[0] + sl[1] + sl[2] + sl[3];
        cnt[nb + blockIdx.x] = se[0] + se[1] + se[2] + se[3];
    }
}

// exclusive scan of m unsigned counters in place, one work-group (sequential over 256-wide slabs)
__global__ __launch_bounds__(256) void k_scan_u32(unsigned int* __restrict__ a, int m0, int m1,
                                                  unsigned long long* __restrict__ totals = nullptr) {
    __shared__ unsigned int s[256];
    __shared__ unsigned int carry;
    for (int seg = 0; seg < 2; seg++) {
        unsigned int* p = seg ? a + m0 : a;
        const int m = seg ? m1 : m0;
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (int b0 = 0; b0 < m; b0 += 256) {
            const int i = b0 + threadIdx.x;
            const unsigned int v = (i < m) ? p[i] : 0u;
            s[threadIdx.x] = v;
            __syncthreads();
            for (int o = 1; o < 256; o <<= 1) {
                const unsigned int add = (threadIdx.x >= o) ? s[threadIdx.x - o] : 0u;
                __syncthreads();
                s[threadIdx.x] += add;
                __syncthreads();
            }
            const unsigned int incl = s[threadIdx.x];
            const unsigned int c = carry;
            if (i < m) p[i] = c + incl - v;
            __syncthreads();
            if (threadIdx.x == 255) carry = c + incl;
            __syncthreads();
        }
        if (totals && threadIdx.x == 0) totals[seg] = carry;
        __syncthreads();
    }
}

// stable compaction: winners with key < T to [0, n_less) and the first `ties` keys == T to
// [n_less, K), both in particle-index order
__global__ __launch_bounds__(256) void k_cp_write(const double* __restrict__ dist, size_t n,
                                                  const SelState* __restrict__ st,
                                                  const unsigned int* __restrict__ off /* [2][nb] scanned */,
                                                  int nb, unsigned long long idx_base,
                                                  unsigned long long* __restrict__ okey,
                                                  unsigned long long* __restrict__ oidx,
                                                  const unsigned long long* __restrict__ lim = nullptr,
                                                  unsigned long long cap = ~0ull /* entries the "< T" part may occupy */) {
    __shared__ unsigned int wl[4], we[4];
    // lim (distributed select): {local count of keys below the threshold, ties this shard may take}
    const unsigned long long T = st->prefix, n_less = lim ? lim[0] : st->n_less, ties = lim ? lim[1] : st->ties;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t base = (size_t)blockIdx.x * CP_CHUNK + (size_t)t * CP_ITEMS;
    unsigned long long k[CP_ITEMS];
    unsigned int l = 0, e = 0;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; j++) {
        const size_t i = base + j;
        k[j] = (i < n) ? key_of(dist[i]) : ~0ull;
        if (i < n) { l += (k[j] < T); e += (k[j] == T); }
    }
    // exclusive scan over threads (wave scan + cross-wave)
    unsigned int il = l, ie = e;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int al = __shfl_up(il, o, 64), ae = __shfl_up(ie, o, 64);
        if (lane >= o) { il += al; ie += ae; }
    }
    if (lane == 63) { wl[wave] = il; we[wave] = ie; }
    __syncthreads();
    unsigned int pl = il - l, pe = ie - e;
    for (int w = 0; w < wave; w++) { pl += wl[w]; pe += we[w]; }
    unsigned long long posl = (unsigned long long)off[blockIdx.x] + pl;
    unsigned long long pose = (unsigned long long)off[nb + blockIdx.x] + pe;
#pragma unroll
    for (int j = 0; j < CP_ITEMS; j++) {
        const size_t i = base + j;
        if (i >= n) break;
        if (k[j] < T) { if (posl < cap) { okey[posl] = k[j]; oidx[posl] = idx_base + i; } posl++; }
        else if (k[j] == T) {
            if (pose < ties) { okey[n_less + pose] = k[j]; oidx[n_less + pose] = idx_base + i; }
            pose++;
        }
    }
}

__global__ __launch_bounds__(256) void k_init_pairs(const double* __restrict__ dist, size_t n,
                                                    unsigned long long idx_base,
                                                    unsigned long long* __restrict__ okey,
                                                    unsigned long long* __restrict__ oidx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { okey[i] = key_of(dist[i]); if (oidx) oidx[i] = idx_base + i; }
}

__global__ __launch_bounds__(256) void k_keys_to_dist(const unsigned long long* __restrict__ key, size_t n,
                                                      double* __restrict__ dist) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dist[i] = dist_of(key[i]);
}

// ---- stable LSD radix sort of (key, idx) pairs, 8 bits per pass ---------------------------------
constexpr int ST_ITEMS = 8;
constexpr int ST_CHUNK = 256 * ST_ITEMS;   // items per work-group; each wave owns ST_CHUNK/4 in order

__global__ __launch_bounds__(256) void k_sort_hist(const unsigned long long* __restrict__ key, size_t n, int shift,
                                                   unsigned int* __restrict__ bh /* [256][nb] */, int nb) {
    __shared__ unsigned int lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * ST_CHUNK;
#pragma unroll
    for (int j = 0; j < ST_ITEMS; j++) {
        const size_t i = base + (size_t)j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&lh[(unsigned int)(key[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    bh[(size_t)threadIdx.x * nb + blockIdx.x] = lh[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_sort_scatter(const unsigned long long* __restrict__ key,
                                                      const unsigned long long* __restrict__ idx, size_t n,
                                                      int shift, const unsigned int* __restrict__ bh, int nb,
                                                      const unsigned int* __restrict__ dtot,
                                                      unsigned long long* __restrict__ okey,
                                                      unsigned long long* __restrict__ oidx) {
    __shared__ unsigned int whist[4][256];
    __shared__ volatile unsigned int woff[4][256];
    __shared__ unsigned int dsum[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int w = 0; w < 4; w++) whist[w][t] = 0;
    // keys with smaller digits (all work-groups): exclusive scan of the 256 digit totals, one per thread
    const unsigned int mytot = dtot[t];
    unsigned int dinc = mytot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int a = __shfl_up(dinc, o, 64); if (lane >= o) dinc += a; }
    if (lane == 63) dsum[wave] = dinc;
    __syncthreads();
    unsigned int dbase = dinc - mytot;
#pragma unroll
    for (int w = 0; w < 4; w++) if (w < wave) dbase += dsum[w];
    const size_t seg = (size_t)blockIdx.x * ST_CHUNK + (size_t)wave * (ST_CHUNK / 4);
    unsigned long long k[ST_ITEMS];
#pragma unroll
    for (int j = 0; j < ST_ITEMS; j++) {
        const size_t i = seg + (size_t)j * 64 + lane;
        k[j] = (i < n) ? key[i] : 0ull;
        if (i < n) atomicAdd(&whist[wave][(unsigned int)(k[j] >> shift) & 255u], 1u);
    }
    __syncthreads();
    {
        unsigned int run = dbase + bh[(size_t)t * nb + blockIdx.x];   // keys of smaller digits + digit t's keys in earlier work-groups
#pragma unroll
        for (int w = 0; w < 4; w++) { woff[w][t] = run; run += whist[w][t]; }
    }
    __syncthreads();
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int j = 0; j < ST_ITEMS; j++) {
        const size_t i = seg + (size_t)j * 64 + lane;
        const bool valid = i < n;
        const unsigned int d = (unsigned int)(k[j] >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        if (valid) {
            const unsigned int rank = __popcll(peers & lt_mask);
            const unsigned int pos = woff[wave][d] + rank;
            okey[pos] = k[j];
            oidx[pos] = idx[i];
        }
        __builtin_amdgcn_wave_barrier();
        if (valid && (peers & lt_mask) == 0) woff[wave][d] += (unsigned int)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
    }
}

// exclusive scan of the digit-major [256][nb] block histogram in two levels: ONE WORK-GROUP PER DIGIT scans its row of nb block
// counts (its own prefix only) and leaves the digit's total in dtot[d]; k_sort_scatter adds the digits in front of its own -- an
// exclusive scan of 256 totals in its prologue.  (Rounds 1-3: one work-group of 1024 threads for the whole table, four threads
// per digit each walking a quarter of the row: 178 us per pass at 1e6 keys, eight passes per sort.)
__global__ __launch_bounds__(256) void k_sort_scan(unsigned int* __restrict__ bh, int nb, unsigned int* __restrict__ dtot) {
    __shared__ unsigned int wsum[4];
    const int d = blockIdx.x, t = threadIdx.x;
    unsigned int* row = bh + (size_t)d * nb;
    const int per = (nb + 255) / 256, i0 = t * per;
    unsigned int loc = 0;
    for (int c = 0; c < per; c++) if (i0 + c < nb) loc += row[i0 + c];
    unsigned int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int a = __shfl_up(inc, o, 64); if ((t & 63) >= o) inc += a; }
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    unsigned int run = inc - loc;
#pragma unroll
    for (int w = 0; w < 4; w++) if (w < (t >> 6)) run += wsum[w];
    for (int c = 0; c < per; c++)
        if (i0 + c < nb) { const unsigned int v = row[i0 + c]; row[i0 + c] = run; run += v; }
    if (t == 255) dtot[d] = run;
}

// ---- small sets: LDS chunk sort + rank merge (2 launches instead of 24) -----------------------------------------
// Up to SC_MAX_N pairs: every work-group bitonic-sorts SC_N pairs in LDS by (key, position in the chunk) -- the position
// rides in the top 16 bits of the index word (row indices stay below 2^48), which makes the network's result the STABLE
// order -- then k_merge_chunks ranks every element among the other chunks by binary search (equal keys: lower chunk
// first), exactly the order the stable radix passes produce.
constexpr int SC_N = 4096;
constexpr size_t SC_MAX_N = (size_t)1 << 18;

__global__ __launch_bounds__(1024) void k_sort_chunks(const unsigned long long* key, const unsigned long long* idx, size_t n,
                                                      unsigned long long* okey, unsigned long long* oidx) {
    __shared__ unsigned long long sk[SC_N];
    __shared__ unsigned long long si[SC_N];
    const size_t base = (size_t)blockIdx.x * SC_N;
    const unsigned cnt = (unsigned)((n - base < (size_t)SC_N) ? n - base : (size_t)SC_N);
    for (unsigned e = threadIdx.x; e < SC_N; e += 1024) {
        sk[e] = (e < cnt) ? key[base + e] : ~0ull;                         // padding sorts last (larger position on ties)
        si[e] = ((e < cnt) ? idx[base + e] : 0ull) | ((unsigned long long)e << 48);
    }
    // Compare-exchange pair p of phase (k, j) is elements i = 2j (p / j) + p % j and i + j.  A wave owns pairs
    // 64w..64w+63 (and the same 1024 further on): for j <= 64 those are exactly elements 128w..128w+127, so the run of
    // phases j = 64 .. 1 of every k stays inside the wave's own elements and needs no work-group barrier (a wave's LDS
    // operations are issued and performed in order); only the 15 phases with j >= 128 are fenced on both sides.
    __syncthreads();
    unsigned prev_j = 1;
    for (unsigned k = 2; k <= SC_N; k <<= 1) {
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            if (j >= 128 || prev_j >= 128) __syncthreads(); else __builtin_amdgcn_wave_barrier();
            prev_j = j;
            unsigned ii[2];
            unsigned long long ka[2], kb[2], ia[2], ib[2];
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const unsigned p = threadIdx.x + 1024u * m;
                ii[m] = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                ka[m] = sk[ii[m]]; kb[m] = sk[ii[m] + j]; ia[m] = si[ii[m]]; ib[m] = si[ii[m] + j];
            }
#pragma unroll
            for (int m = 0; m < 2; m++) {
                const bool gt = (ka[m] > kb[m]) || (ka[m] == kb[m] && (ia[m] >> 48) > (ib[m] >> 48));
                if (gt == ((ii[m] & k) == 0)) { sk[ii[m]] = kb[m]; sk[ii[m] + j] = ka[m]; si[ii[m]] = ib[m]; si[ii[m] + j] = ia[m]; }
            }
        }
    }
    __syncthreads();
    for (unsigned e = threadIdx.x; e < cnt; e += 1024) {
        okey[base + e] = sk[e];
        oidx[base + e] = si[e] & 0x0000ffffffffffffull;
    }
}

// Rank of x in chunk r = number of its elements that come before x: a fixed-trip branchless search (13 probes of a
// 4096-element chunk).  Eight chunks are searched at once, so a thread has eight independent L2 load chains in flight
// instead of one chain of 13 x (chunks - 1) dependent loads.
__global__ __launch_bounds__(256) void k_merge_chunks(const unsigned long long* __restrict__ key,
                                                      const unsigned long long* __restrict__ idx, size_t n,
                                                      unsigned long long* __restrict__ okey, unsigned long long* __restrict__ oidx) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned q = (unsigned)(t / SC_N), W = (unsigned)((n + SC_N - 1) / SC_N);
    const unsigned long long x = key[t];
    size_t pos = t - (size_t)q * SC_N;
    for (unsigned r0 = 0; r0 < W; r0 += 8) {
        unsigned lo[8], len[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const unsigned r = r0 + u;
            lo[u] = 0;
            len[u] = (r < W && r != q) ? (unsigned)((n - (size_t)r * SC_N < (size_t)SC_N) ? n - (size_t)r * SC_N : (size_t)SC_N) : 0u;
        }
#pragma unroll 1
        for (unsigned step = SC_N; step >= 1; step >>= 1) {
            unsigned long long y[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {                               // unconditional (clamped) loads: all eight in flight
                const unsigned c = lo[u] + step;                       // candidate count of elements before x
                const unsigned r = (r0 + u < W) ? r0 + u : q;
                y[u] = key[(size_t)r * SC_N + ((c <= len[u]) ? c - 1 : 0u)];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned c = lo[u] + step;
                const bool before = (r0 + u < q) ? (y[u] <= x) : (y[u] < x);     // equal keys: lower chunks first (stable)
                lo[u] = (c <= len[u] && before) ? c : lo[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) pos += lo[u];
    }
    okey[pos] = x;
    oidx[pos] = idx[t];
}

// sorts n pairs; result ends in (key0, idx0); (key1, idx1) is scratch of the same size
int sort_pairs_u64(abc_ctx* ctx, unsigned long long* key0, unsigned long long* idx0, unsigned long long* key1,
                   unsigned long long* idx1, size_t n, int byte_lo = 0, int byte_hi = 8) {
    if (n <= 1) return ABC_OK;
    StageTimer tm(ctx, ST_SORT);
    if (byte_lo == 0 && byte_hi == 8 && n <= SC_MAX_N) {
        const unsigned nc = (unsigned)((n + SC_N - 1) / SC_N);
        if (nc == 1) {
            hipLaunchKernelGGL(k_sort_chunks, dim3(1), dim3(1024), 0, ctx->stream, key0, idx0, n, key0, idx0);
        } else {        // sorted chunks into the scratch pair, merged back into (key0, idx0)
            hipLaunchKernelGGL(k_sort_chunks, dim3(nc), dim3(1024), 0, ctx->stream, key0, idx0, n, key1, idx1);
            hipLaunchKernelGGL(k_merge_chunks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, key1, idx1, n,
                               key0, idx0);
        }
        ABC_HIP(ctx, hipGetLastError());
        return ABC_OK;
    }
    const int nb = (int)((n + ST_CHUNK - 1) / ST_CHUNK);
    unsigned int* bh = (unsigned int*)abc_ws_alloc(ctx, ((size_t)256 * nb + 256) * sizeof(unsigned int));
    if (!bh) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sort: workspace exhausted");
    unsigned int* dtot = bh + (size_t)256 * nb;
    unsigned long long *ka = key0, *ia = idx0, *kb = key1, *ib = idx1;
    for (int pass = byte_lo; pass < byte_hi; pass++) {   // an even number of passes ends in (key0, idx0)
        const int shift = 8 * pass;
        hipLaunchKernelGGL(k_sort_hist, dim3(nb), dim3(256), 0, ctx->stream, ka, n, shift, bh, nb);
        hipLaunchKernelGGL(k_sort_scan, dim3(256), dim3(256), 0, ctx->stream, bh, nb, dtot);
        hipLaunchKernelGGL(k_sort_scatter, dim3(nb), dim3(256), 0, ctx->stream, ka, ia, n, shift, bh, nb, (const unsigned int*)dtot, kb, ib);
        unsigned long long* tk = ka; ka = kb; kb = tk;
        unsigned long long* ti = ia; ia = ib; ib = ti;
    }
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;   // 8 passes: even number of swaps -> result back in (key0, idx0)
}

}  // namespace

// stable sort of n (key, value) pairs on key bytes [byte_lo, byte_hi) (an even number of passes; values below 2^48:
// the full-key sort of small sets tags the value word, see k_sort_chunks);
// the result is left in (key0, val0); (key1, val1) are scratch.  Used by the Wilcoxon rank sums.
int abc_sort_u64_bytes(abc_ctx* ctx, unsigned long long* key0, unsigned long long* val0, unsigned long long* key1,
                       unsigned long long* val1, size_t n, int byte_lo, int byte_hi) {
    if ((byte_hi - byte_lo) % 2 != 0) ABC_FAIL(ctx, ABC_ERR_INVALID, "sort: odd number of radix passes");
    return sort_pairs_u64(ctx, key0, val0, key1, val1, n, byte_lo, byte_hi);
}

// ---- distributed radix select (sharded driver): the histogram of every pass is all-reduced between
// launch_select_hist and launch_select_pick, so all shards walk to the same global K-th key ----------------
static const int kSelShift[6] = {53, 42, 31, 20, 9, 0};
static const int kSelWidth[6] = {11, 11, 11, 11, 11, 9};

int launch_select_begin(abc_ctx* ctx, uint64_t K, long long* state, int* hist) {
    hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(256), 0, ctx->stream, (SelState*)state, (unsigned long long)K,
                       (unsigned int*)hist);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
int launch_select_hist(abc_ctx* ctx, const double* dist, size_t n, const long long* state, int pass, int* hist) {
    if (pass < 0 || pass > 5) ABC_FAIL(ctx, ABC_ERR_INVALID, "select: pass %d", pass);
    if (n == 0) return ABC_OK;
    StageTimer tm(ctx, ST_SELECT);
    size_t hb = (n + 255) / 256;
    if (hb > 512) hb = 512;
    hipLaunchKernelGGL(k_sel_hist, dim3((unsigned)hb), dim3(256), 0, ctx->stream, dist, n, (const SelState*)state,
                       kSelShift[pass], kSelWidth[pass], (unsigned int*)hist);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
int launch_select_pick(abc_ctx* ctx, long long* state, int pass, int* hist, uint64_t K) {
    if (pass < 0 || pass > 5) ABC_FAIL(ctx, ABC_ERR_INVALID, "select: pass %d", pass);
    StageTimer tm(ctx, ST_SELECT);
    hipLaunchKernelGGL(k_sel_pick, dim3(1), dim3(256), 0, ctx->stream, (SelState*)state, kSelShift[pass], kSelWidth[pass],
                       (unsigned int*)hist, (int)(pass == 5), (unsigned long long)K);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
// counts[0] = local #keys below the global threshold, counts[1] = local #keys equal to it
int launch_select_count(abc_ctx* ctx, const double* dist, size_t n, const long long* state, long long* counts) {
    StageTimer tm(ctx, ST_SELECT);
    const int nb = (int)((n + CP_CHUNK - 1) / CP_CHUNK);
    unsigned int* cnt = (unsigned int*)abc_ws_alloc(ctx, (size_t)2 * (nb + 1) * sizeof(unsigned int));
    if (!cnt) ABC_FAIL(ctx, ABC_ERR_NOMEM, "select: workspace exhausted");
    if (nb) hipLaunchKernelGGL(k_cp_count, dim3(nb), dim3(256), 0, ctx->stream, dist, n, (SelState*)state, cnt, nb);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(256), 0, ctx->stream, cnt, nb, nb, (unsigned long long*)counts);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
// local winners in particle-index order: the n_less keys below the threshold, then the first ties_take ties
int launch_select_compact(abc_ctx* ctx, const double* dist, size_t n, const long long* state, uint64_t n_less,
                          uint64_t ties_take, uint64_t idx_base, uint64_t* idx_out, double* dist_out) {
    StageTimer tm(ctx, ST_SELECT);
    const size_t nw = n_less + ties_take;
    if (nw == 0 || n == 0) return ABC_OK;
    const int nb = (int)((n + CP_CHUNK - 1) / CP_CHUNK);
    unsigned int* cnt = (unsigned int*)abc_ws_alloc(ctx, (size_t)2 * nb * sizeof(unsigned int));
    unsigned long long* lim = (unsigned long long*)abc_ws_alloc(ctx, 2 * sizeof(unsigned long long));
    unsigned long long* key = (unsigned long long*)abc_ws_alloc(ctx, nw * 8);
    if (!cnt || !lim || !key) ABC_FAIL(ctx, ABC_ERR_NOMEM, "select: workspace exhausted");
    const unsigned long long hl[2] = {n_less, ties_take};
    ABC_HIP(ctx, hipMemcpyAsync(lim, hl, sizeof(hl), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cp_count, dim3(nb), dim3(256), 0, ctx->stream, dist, n, (SelState*)state, cnt, nb);
    hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(256), 0, ctx->stream, cnt, nb, nb, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k_cp_write, dim3(nb), dim3(256), 0, ctx->stream, dist, n, (const SelState*)state, cnt, nb,
                       (unsigned long long)idx_base, key, (unsigned long long*)idx_out, lim);
    hipLaunchKernelGGL(k_keys_to_dist, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, ctx->stream, key, nw, dist_out);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// bin path: the caller (defer_check) or this function reads ctx->sel_fail_dev afterwards; see abc_select_check
static int select_by_bins(abc_ctx* ctx, const double* dist, size_t n, size_t K, uint64_t idx_base, uint64_t* idx, double* dist_out) {
    if (!ctx->sel_fail_dev) ABC_HIP(ctx, hipMalloc((void**)&ctx->sel_fail_dev, sizeof(int)));
    int* const fail_dev = ctx->sel_fail_dev;
    const int nbins = K <= ((size_t)1 << 18) ? BS_NB : BS_NB_BIG;
    BinSel* bs = (BinSel*)abc_ws_alloc(ctx, sizeof(BinSel));
    unsigned int* hist = (unsigned int*)abc_ws_alloc(ctx, ((size_t)nbins + 1) * sizeof(unsigned int));
    unsigned int* cursor = (unsigned int*)abc_ws_alloc(ctx, (size_t)nbins * BS_CSTRIDE * sizeof(unsigned int));
    unsigned long long* tkey = (unsigned long long*)abc_ws_alloc(ctx, (K + BS_CAP) * 8);
    unsigned long long* tidx = (unsigned long long*)abc_ws_alloc(ctx, (K + BS_CAP) * 8);
    if (!bs || !hist || !cursor || !tkey || !tidx) ABC_FAIL(ctx, ABC_ERR_NOMEM, "select: workspace exhausted");
    StageTimer tm(ctx, ST_SELECT);
    hipLaunchKernelGGL(k_bs_sample, dim3(1), dim3(1024), 0, ctx->stream, dist, n, (unsigned long long)K, bs, hist, fail_dev, nbins);
    size_t hb = (n + 255) / 256;
    if (hb > 512) hb = 512;
    const size_t hl = (size_t)nbins * sizeof(unsigned int);
    if (hl > (48u << 10)) ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_bs_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hl));
    hipLaunchKernelGGL(k_bs_hist, dim3((unsigned)hb), dim3(256), hl, ctx->stream, dist, n, (const BinSel*)bs, hist, cursor, nbins);
    if (nbins == BS_NB) hipLaunchKernelGGL(k_bs_scan<BS_NB / 1024>, dim3(1), dim3(1024), 0, ctx->stream, bs, hist, (unsigned long long)K, fail_dev);
    else hipLaunchKernelGGL(k_bs_scan<BS_NB_BIG / 1024>, dim3(1), dim3(1024), 0, ctx->stream, bs, hist, (unsigned long long)K, fail_dev);
    hipLaunchKernelGGL(k_bs_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dist, n, (const BinSel*)bs,
                       (const unsigned int*)hist, cursor, (unsigned long long)idx_base, tkey, tidx);
    hipLaunchKernelGGL(k_bs_sort, dim3((unsigned)nbins), dim3(256), 0, ctx->stream, (const BinSel*)bs, (const unsigned int*)hist,
                       (const unsigned long long*)tkey, (const unsigned long long*)tidx, (unsigned long long)K,
                       (unsigned long long)idx_base, (unsigned long long*)idx, dist_out);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

static int select_by_radix(abc_ctx* ctx, const double* dist, size_t n, size_t K, uint64_t idx_base, uint64_t* idx, double* dist_out) {
    unsigned long long* key0 = (unsigned long long*)abc_ws_alloc(ctx, K * 8);
    unsigned long long* key1 = (unsigned long long*)abc_ws_alloc(ctx, K * 8);
    unsigned long long* idx1 = (unsigned long long*)abc_ws_alloc(ctx, K * 8);
    if (!key0 || !key1 || !idx1) ABC_FAIL(ctx, ABC_ERR_NOMEM, "select: workspace exhausted");
    unsigned long long* idx0 = (unsigned long long*)idx;
    {
    StageTimer tm(ctx, ST_SELECT);
    if (K == n) {
        hipLaunchKernelGGL(k_init_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dist, n,
                           (unsigned long long)idx_base, key0, idx0);
    } else {
        SelState* st = (SelState*)abc_ws_alloc(ctx, 7 * sizeof(SelState));           // st[p]: state before pass p; st[6]: final
        unsigned int* hist = (unsigned int*)abc_ws_alloc(ctx, 6 * SEL_BINS * sizeof(unsigned int));
        const int nb = (int)((n + CP_CHUNK - 1) / CP_CHUNK);
        unsigned int* cnt = (unsigned int*)abc_ws_alloc(ctx, (size_t)2 * nb * sizeof(unsigned int));
        if (!st || !hist || !cnt) ABC_FAIL(ctx, ABC_ERR_NOMEM, "select: workspace exhausted");
        hipLaunchKernelGGL(k_sel_init_fused, dim3(1), dim3(256), 0, ctx->stream, st, (unsigned long long)K, hist);
        size_t hb = (n + 255) / 256;
        if (hb > 512) hb = 512;      // every block flushes up to 2048 bins with global atomics: keep the count low
        static const int shifts[6] = {53, 42, 31, 20, 9, 0};
        static const int widths[6] = {11, 11, 11, 11, 11, 9};
        for (int p = 0; p < 6; p++)
            hipLaunchKernelGGL(k_sel_hist_fused, dim3((unsigned)hb), dim3(256), 0, ctx->stream, dist, n, st, p,
                               p ? shifts[p - 1] : 0, p ? widths[p - 1] : 0, shifts[p], widths[p], hist, (unsigned long long)K);
        hipLaunchKernelGGL(k_cp_count, dim3(nb), dim3(256), 0, ctx->stream, dist, n, st + 5, cnt, nb,
                           (const unsigned int*)(hist + 5 * SEL_BINS), shifts[5], widths[5], (unsigned long long)K);
        hipLaunchKernelGGL(k_scan_u32, dim3(1), dim3(256), 0, ctx->stream, cnt, nb, nb, (unsigned long long*)nullptr);
        hipLaunchKernelGGL(k_cp_write, dim3(nb), dim3(256), 0, ctx->stream, dist, n, (const SelState*)(st + 6), cnt, nb,
                           (unsigned long long)idx_base, key0, idx0, (const unsigned long long*)nullptr);
    }
    }
    ABC_HIP(ctx, hipGetLastError());
    ABC_TRY(sort_pairs_u64(ctx, key0, idx0, key1, idx1, K));
    if (dist_out) {
        hipLaunchKernelGGL(k_keys_to_dist, dim3((unsigned)((K + 255) / 256)), dim3(256), 0, ctx->stream, key0, K,
                           dist_out);
        ABC_HIP(ctx, hipGetLastError());
    }
    return ABC_OK;
}

// Sampled-range bins when they apply (a large set, at most half of it kept, K within one work-group's reach per bin), the
// radix select + sort otherwise.  defer_check = true: the caller calls abc_select_check after its next synchronisation and,
// if that reports a failed bin selection, repeats its work with ctx->sel_force_radix set.
int launch_select_smallest(abc_ctx* ctx, const double* dist, size_t n, size_t K, uint64_t idx_base, uint64_t* idx,
                           double* dist_out, bool defer_check) {
    if (K == 0) return ABC_OK;
    if (K > n) ABC_FAIL(ctx, ABC_ERR_INVALID, "select: K = %zu > n = %zu", K, n);
    const bool bins = !ctx->sel_force_radix && n >= 4 * (size_t)BS_S && 2 * K <= n && K <= ((size_t)1 << 20);
    ctx->sel_bins_ran = bins;
    if (!bins) return select_by_radix(ctx, dist, n, K, idx_base, idx, dist_out);
    const size_t mark = ctx->ws_off;
    ABC_TRY(select_by_bins(ctx, dist, n, K, idx_base, idx, dist_out));
    if (defer_check) return ABC_OK;
    int failed = 0;
    ABC_TRY(abc_select_check(ctx, &failed));
    if (!failed) return ABC_OK;
    ctx->ws_off = mark;
    return select_by_radix(ctx, dist, n, K, idx_base, idx, dist_out);
}

int abc_select_check_queue(abc_ctx* ctx, int* slot) {
    *slot = 0;
    if (!ctx->sel_bins_ran || !ctx->sel_fail_dev) return ABC_OK;
    ABC_HIP(ctx, hipMemcpyAsync(slot, ctx->sel_fail_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return ABC_OK;
}
int abc_select_check_done(abc_ctx* ctx, const int* slot) {
    const int f = (ctx->sel_bins_ran && ctx->sel_fail_dev) ? *slot : 0;
    ctx->sel_bins_ran = false;
    return f;
}

// after a synchronisation point of the caller: did the last bin selection give up?  (synchronises the stream itself)
int abc_select_check(abc_ctx* ctx, int* failed) {
    *failed = 0;
    if (!ctx->sel_bins_ran || !ctx->sel_fail_dev) return ABC_OK;
    int f = 0;
    ABC_HIP(ctx, hipMemcpyAsync(&f, ctx->sel_fail_dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->sel_bins_ran = false;
    *failed = f;
    return ABC_OK;
}

// ---- merge of W sorted runs (sharded winners): rank every element by binary searches in the other runs ----------
// Run q occupies [q*len, (q+1)*len), sorted by (key, idx) with idx ascending in q: equal keys keep run order
// (elements of a lower run first), i.e. the result equals a stable sort of the concatenation.
__global__ __launch_bounds__(256) void k_merge_runs(const double* __restrict__ key, const unsigned long long* __restrict__ idx,
                                                    int W, size_t len, double* __restrict__ okey,
                                                    unsigned long long* __restrict__ oidx,
                                                    unsigned long long* __restrict__ osrc /* optional: input position */) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)W * len) return;
    const int q = (int)(t / len);
    const size_t i = t - (size_t)q * len;
    const double d = key[t];
    const unsigned long long x = key_of(d);
    size_t top = 1;                                    // smallest power of two >= len: first probe distance
    while (top < len) top <<= 1;
    size_t pos = i;
    // rank of x in run r = number of its elements that come before x: fixed-trip branch-free searches, eight runs at a
    // time (eight independent load chains instead of one), as in k_merge_chunks
    for (int r0 = 0; r0 < W; r0 += 8) {
        size_t lo[8];
#pragma unroll
        for (int u = 0; u < 8; u++) lo[u] = 0;
#pragma unroll 1
        for (size_t step = top; step >= 1; step >>= 1) {
            unsigned long long y[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int r = (r0 + u < W) ? r0 + u : q;
                const size_t c = lo[u] + step;
                y[u] = key_of(key[(size_t)r * len + ((c <= len) ? c - 1 : 0)]);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int r = r0 + u;
                const size_t c = lo[u] + step;
                const bool before = (r < q) ? (y[u] <= x) : (y[u] < x);       // equal keys: lower runs first (stable)
                lo[u] = (r < W && r != q && c <= len && before) ? c : lo[u];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) pos += lo[u];
    }
    okey[pos] = d;
    oidx[pos] = idx[t];
    if (osrc) osrc[pos] = t;
}

int launch_merge_runs(abc_ctx* ctx, const double* key, const uint64_t* idx, int W, size_t len, double* okey, uint64_t* oidx,
                      uint64_t* osrc) {
    if (W < 1) ABC_FAIL(ctx, ABC_ERR_INVALID, "merge: W = %d", W);
    const size_t n = (size_t)W * len;
    if (n == 0) return ABC_OK;
    StageTimer tm(ctx, ST_SORT);
    hipLaunchKernelGGL(k_merge_runs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, key,
                       (const unsigned long long*)idx, W, len, okey, (unsigned long long*)oidx, (unsigned long long*)osrc);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

int launch_sort_pairs(abc_ctx* ctx, double* key, uint64_t* idx, size_t n) {
    if (n <= 1) return ABC_OK;
    unsigned long long* key0 = (unsigned long long*)abc_ws_alloc(ctx, n * 8);
    unsigned long long* key1 = (unsigned long long*)abc_ws_alloc(ctx, n * 8);
    unsigned long long* idx1 = (unsigned long long*)abc_ws_alloc(ctx, n * 8);
    if (!key0 || !key1 || !idx1) ABC_FAIL(ctx, ABC_ERR_NOMEM, "sort: workspace exhausted");
    hipLaunchKernelGGL(k_init_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, key, n, 0ull, key0,
                       (unsigned long long*)nullptr);
    ABC_TRY(sort_pairs_u64(ctx, key0, (unsigned long long*)idx, key1, idx1, n));
    hipLaunchKernelGGL(k_keys_to_dist, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, key0, n, key);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
