// Wilcoxon signed-rank reduction of the number of PLS components ([PLS] optimal_num_components, called at
// AbcUtil.cpp:447-449): for every response j whose PRESS optimum is a*_j > 1, the
// smallest a' < a*_j whose absolute validation errors are not significantly different (two-sided, normal
// approximation, alpha = 0.1, "average" tie ranks as lib/ranker.h:66-76, zero differences dropped) replaces it.
//
// All (j, a') tests ("segments") are batched.  Rank sums are sums of half-integers (< 2^52), hence exact and order-independent,
// so integer / double atomics stay bit-reproducible; residuals use the same fixed fma order as the oracle.
//
// Round 4 -- the BINNED path (launch_wilcoxon_binned; 134 ms -> see DESIGN.md section 5 at 1e6 particles x 16 responses x 8
// components: this rule is the drop-in default of the C++ facade).  A rank is "elements below" + "position among the
// equal-or-close ones", so nothing has to be sorted globally:
//   k_wx_sample   per segment: |d| of evenly spaced validation rows, sorted in LDS (leading 32 key bits) -> NB - 1 splitters
//                 (equi-depth bins; a bin is a function of the key alone, so tied values share a bin);
//   k_wx_bin<.., false>   counting pass: a work-group takes a run of 256 R-row tiles of ONE response, keeps the residuals of all
//                 component counts of its rows in registers (the scores of a row are read once for all of the response's
//                 segments), bins every segment's keys through the splitters in LDS and leaves its per-bin counts;
//   k_wx_offsets  per segment: running offsets of the work-groups inside every bin, bin sizes, bin starts, the number of
//                 non-zero differences;
//   k_wx_bin<.., true>    the same sweep again, now placing the keys (sign in bit 63) bin by bin -- staged through LDS so that
//                 a work-group writes its share of a bin as one contiguous piece;
//   k_wx_ranks    one work-group per (segment, bin), the bin (<= 8192 keys) in LDS: 1024 linear sub-bins by counting, then
//                 every key counts the smaller and the equal keys of its sub-bin -> average rank = keys below the bin + below
//                 the sub-bin + smaller in it + (equal + 1) / 2; signed sum -> W[segment] (exact);
//   k_wx_decide   unchanged.
// Second half of round 4 -- MOST TESTS NEVER GET THAT FAR.  Before the sweeps above, one sweep (k_wx_bin<.., 2>) counts every test's
// keys, all and positive, in ~2.5 sqrt(n) fine bins (a sampled table key -> bin: one LDS read, one LDS atomic per key), and
// k_wx_bounds turns the counts into exact bounds on the signed rank sum: a test whose interval of |W| / sigma lies on one side of the
// decision threshold is settled; the sweeps above then only work on the undecided tests (their work-groups read the verdicts and
// leave).  Same component counts -- the bounds are rigorous --, a third of the time at 112 tests x 5e5 rows.
// 24 bytes of traffic per (row, test) instead of ~500 (ten 16-byte LSD radix passes), no host round trip in the middle.  A
// bin that outgrows LDS (massive ties, a degenerate sample) raises a flag: launch_wilcoxon reads it at the end and repeats the
// reduction on the SORTED path of rounds 1-3 (one stable LSD radix sort of all (key, segment) pairs), which also takes the
// shapes the binned path is not built for (more than 32 components, more than ~1.4e7 validation rows).
#include "abc_internal.h"
#include <vector>

// (k_wx_sample<32>'s sorting network is too long for the unroller's budget: it stays a loop there, which is correct, only slower)
#pragma clang diagnostic ignored "-Wpass-failed"

namespace {

constexpr size_t MAXSEG = 65535;      // (response, candidate) tests of one call: the segment id travels in two payload bytes of the
                                      // batched sort and in grid.y; P (A - 1) beyond that is ABC_ERR_UNSUPPORTED

struct WxPlan {            // built on the device from the model record; the arrays live in the workspace (P (A - 1), P entries)
    int nseg;
    int pad_;
    int* seg_j;            // response
    int* seg_a;            // candidate a' (1-based)
    int* astar;            // PRESS optimum per response
};

__global__ void k_wx_plan(const double* __restrict__ model, int M, int P, int A, WxPlan* __restrict__ plan,
                          int* __restrict__ seg_j, int* __restrict__ seg_a, int* __restrict__ astar, int nseg_max,
                          unsigned long long* __restrict__ nz, double* __restrict__ W, int* __restrict__ segbase,
                          int* __restrict__ fail, int* __restrict__ v3 = nullptr) {
    for (int s = threadIdx.x; s < nseg_max; s += blockDim.x) { nz[s] = 0; W[s] = 0.0; if (v3) v3[s] = 2; }
    if (threadIdx.x != 0) return;
    if (fail) { fail[0] = 0; fail[1] = 0; }           // [1]: tests the bounds leave undecided (k_wx_bounds)
    const ModelLayout ML = model_layout(M, P, A);
    int ns = 0;
    for (int j = 0; j < P; j++) {
        const int as = (int)model[ML.off_per + j];
        astar[j] = as;
        if (segbase) segbase[j] = ns;            // the tests of response j are segments segbase[j] .. segbase[j] + as - 2
        for (int a = 1; a < as && ns < nseg_max; a++) { seg_j[ns] = j; seg_a[ns] = a; ns++; }
    }
    plan->nseg = ns;
    plan->pad_ = 0;
    plan->seg_j = seg_j;
    plan->seg_a = seg_a;
    plan->astar = astar;
}

// scores of the validation rows: S[i + nt*k] = sum_m z(x_im) R[m,k]  (m ascending fma chain, as the oracle)
template <int KC>
__global__ __launch_bounds__(256) void k_wx_scores(const double* __restrict__ X, size_t ldx, size_t row_test, size_t nt,
                                                   int M, int P, int A, const double* __restrict__ model,
                                                   double* __restrict__ S) {
    const ModelLayout ML = model_layout(M, P, A);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    double s[KC];
#pragma unroll
    for (int k = 0; k < KC; k++) s[k] = 0.0;
    for (int m0 = 0; m0 < M; m0 += 8) {               // eight metrics' loads in flight (one per loop turn: M memory latencies in a row)
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = X[row_test + i + ldx * (size_t)(m0 + u < M ? m0 + u : M - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int m = m0 + u;
            if (m >= M) break;
            const double sd = model[ML.off_sd + m];
            const double z = (sd == 0.0) ? 0.0 : (x[u] - model[ML.off_mean + m]) / sd;
#pragma unroll
            for (int k = 0; k < KC; k++)
                if (k < A) s[k] = fma(z, model[ML.off_R + m + (size_t)M * k], s[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < KC; k++)
        if (k < A) S[i + nt * k] = s[k];
}

// more than 32 components: chunks of 32 (the row's metrics re-read per chunk; same fma chain per component)
__global__ __launch_bounds__(256) void k_wx_scores_wide(const double* __restrict__ X, size_t ldx, size_t row_test, size_t nt,
                                                        int M, int P, int A, const double* __restrict__ model,
                                                        double* __restrict__ S) {
    const ModelLayout ML = model_layout(M, P, A);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nt) return;
    for (int k0 = 0; k0 < A; k0 += 32) {
        double s[32];
#pragma unroll
        for (int k = 0; k < 32; k++) s[k] = 0.0;
        for (int m = 0; m < M; m++) {
            const double sd = model[ML.off_sd + m];
            const double z = (sd == 0.0) ? 0.0 : (X[row_test + i + ldx * m] - model[ML.off_mean + m]) / sd;
#pragma unroll
            for (int k = 0; k < 32; k++)
                if (k0 + k < A) s[k] = fma(z, model[ML.off_R + m + (size_t)M * (k0 + k)], s[k]);
        }
#pragma unroll
        for (int k = 0; k < 32; k++)
            if (k0 + k < A) S[i + nt * (k0 + k)] = s[k];
    }
}

// one thread per (validation row, segment): key / payload of the paired difference
__global__ __launch_bounds__(256) void k_wx_diffs(const double* __restrict__ Y, size_t ldy, size_t row_test, size_t nt,
                                                  int M, int P, int A, const double* __restrict__ model,
                                                  const double* __restrict__ S, const WxPlan* __restrict__ plan,
                                                  unsigned long long* __restrict__ key,
                                                  unsigned long long* __restrict__ val,
                                                  unsigned long long* __restrict__ nz) {
    const ModelLayout ML = model_layout(M, P, A);
    const int seg = blockIdx.y;
    if (seg >= plan->nseg) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int j = plan->seg_j[seg], a1 = plan->seg_a[seg], as = plan->astar[j];
    unsigned int nonzero = 0;
    if (i < nt) {
        const double sdy = model[ML.off_sd + M + j];
        const double zy = (sdy == 0.0) ? 0.0 : (Y[row_test + i + ldy * j] - model[ML.off_mean + M + j]) / sdy;
        double pred = 0.0, e_small = 0.0;
        for (int k = 0; k < as; k++) {
            pred = fma(S[i + nt * k], model[ML.off_Q + j + (size_t)P * k], pred);
            if (k + 1 == a1) e_small = zy - pred;
        }
        const double e_star = zy - pred;
        const double d = fabs(e_star) - fabs(e_small);
        const unsigned long long sign = d > 0.0 ? 1ull : 0ull;
        nonzero = d != 0.0;
        key[(size_t)seg * nt + i] = nonzero ? (unsigned long long)__double_as_longlong(fabs(d)) : ~0ull;
        val[(size_t)seg * nt + i] = ((unsigned long long)seg << 32) | (sign << 31);
    }
    const unsigned long long m = __ballot(nonzero);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&nz[seg], (unsigned long long)__popcll(m));
}

// after sorting by (segment, |d|): signed sum of average ranks of the non-zero differences of every segment
__global__ __launch_bounds__(256) void k_wx_ranksum(const unsigned long long* __restrict__ key,
                                                    const unsigned long long* __restrict__ val, size_t nt,
                                                    const WxPlan* __restrict__ plan,
                                                    const unsigned long long* __restrict__ nz, double* __restrict__ W) {
    const int seg = blockIdx.y;
    if (seg >= plan->nseg) return;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)nz[seg];
    double w = 0.0;
    if (i < n) {
        const unsigned long long* k = key + (size_t)seg * nt;
        const unsigned long long me = k[i];
        size_t lo = i, hi = i;                       // tie run [lo, hi] (ties of doubles are rare: short scans)
        while (lo > 0 && k[lo - 1] == me) lo--;
        while (hi + 1 < n && k[hi + 1] == me) hi++;
        const double rank = (double)(lo + hi) / 2.0 + 1.0;             // ranker.h:74-75 "average"
        const bool pos = (val[(size_t)seg * nt + i] >> 31) & 1ull;
        w = pos ? rank : -rank;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o, 64);       // exact: multiples of 1/2 below 2^52
    if ((threadIdx.x & 63) == 0 && w != 0.0) atomicAdd(&W[seg], w);
}


// ===========================================================================================================================
// the binned path
// ===========================================================================================================================
constexpr int WXT = 256;                         // threads of the sweep kernels
constexpr int WX_CAP = 16384;                    // keys one bin may hold (k_wx_ranks_big: the bin in 128 KB of LDS)
constexpr int WX_NBMAX = 4096;                   // bins per segment
constexpr int WX_TAB = 512;                      // cells of the bounds sweep's key -> fine bin table
constexpr unsigned long long WX_SIGN = 1ull << 63;
constexpr unsigned long long WX_MASK = ~WX_SIGN;

struct WxGeo {                 // geometry of one binned reduction (computed on the host, passed by value)
    unsigned long long nt;     // validation rows
    int NB;                    // bins per segment (a power of two, 1 .. WX_NBMAX)
    int SAMP;                  // sampled rows per segment (a power of two <= nt; unused when NB == 1)
    int ST;                    // work-groups per response in the sweeps, each taking `tps` consecutive tiles of 256 R rows
    int tps;
    int G;                     // segments of one response whose splitters / counters share LDS (the sweep loops over groups)
    int nseg_max;
    // the BOUNDS sweep (k_wx_bin<.., 2>): F linear sub-bins per bin (a power of two; 0: no bounds sweep), its own partition of the
    // tiles (a work-group's rows stay below 2^16: its per-bin counters are two 16-bit halves) and its own group size
    int F;
    int STb, tpsb, Gb;
};

// -DWX_STAMPS (diagnostic build, scripts/wx_stamps.sh): thread 0 of every work-group of the kernel WX_STAMPS names (1: k_wx_ranks, 2 / 3: the counting / placing
// sweep k_wx_bin) leaves s_memtime at its phase boundaries in a buffer of its own; the product build has no stamp
#ifdef WX_STAMPS
__device__ unsigned long long* wx_stamp_buf = nullptr;
#define WX_STAMP_K(kern, i) do { if (WX_STAMPS == (kern) && threadIdx.x == 0 && wx_stamp_buf) wx_stamp_buf[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define WX_STAMP(i) WX_STAMP_K(1, i)
#else
#define WX_STAMP_K(kern, i) do { } while (0)
#define WX_STAMP(i) do { } while (0)
#endif

// key of validation row i in test (j, a1): |d| = ||e_as| - |e_a1|| as its IEEE pattern, the sign of d in bit 63; d == 0: no key
__device__ __forceinline__ double wx_zy(const double* __restrict__ Y, size_t ldy, size_t row, int j, const double* __restrict__ model,
                                        const ModelLayout& ML, int M) {
    const double sdy = model[ML.off_sd + M + j];
    return (sdy == 0.0) ? 0.0 : (Y[row + ldy * j] - model[ML.off_mean + M + j]) / sdy;
}

// bitonic sort of T EPT keys, EPT per thread in registers (element t EPT + u): compare-exchanges inside a thread stay in
// registers, inside a wave they are shuffles, only the strides across waves go through LDS (10 of the 78 stages at 4096 keys; all
// 78 through LDS with a barrier each were 80 us of this kernel's 90).  lds: T EPT words.
template <int T, int EPT>
__device__ __forceinline__ void wx_sort_regs(unsigned int (&v)[EPT], unsigned int* lds) {
    constexpr int N = T * EPT;
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < EPT) {
#pragma unroll
                for (int u = 0; u < EPT; u++) {
                    const int x = u ^ j;
                    if (x > u) {
                        const bool asc = ((t * EPT + u) & k) == 0;
                        const unsigned int a = v[u], b = v[x];
                        if ((a > b) == asc) { v[u] = b; v[x] = a; }
                    }
                }
            } else {
                const int m = j / EPT;                                      // the partner thread is t ^ m
                const bool lower = (t & m) == 0;
                if (m >= 64) {
#pragma unroll
                    for (int u = 0; u < EPT; u++) lds[u * T + t] = v[u];
                    __syncthreads();
                }
#pragma unroll
                for (int u = 0; u < EPT; u++) {
                    const unsigned int o = (m >= 64) ? lds[u * T + (t ^ m)] : (unsigned int)__shfl_xor((int)v[u], m, 64);
                    const bool asc = ((t * EPT + u) & k) == 0;
                    const unsigned int lo = v[u] < o ? v[u] : o, hi = v[u] < o ? o : v[u];
                    v[u] = (lower == asc) ? lo : hi;
                }
                if (m >= 64) __syncthreads();
            }
        }
    }
}

// splitters of every segment from a sample of its keys: the leading 32 of the 63 key bits of SAMP = 1024 EPT evenly spaced rows,
// sorted; splitter b = the sample's ((b + 1) / NB)-quantile with its low 31 bits set, so bin(key) = #{splitters < key} puts equal
// keys (and keys equal in their leading 32 bits) in one bin
template <int EPT>
__global__ __launch_bounds__(1024) void k_wx_sample(const double* __restrict__ Y, size_t ldy, size_t row_test, WxGeo g, int M, int P,
                                                    int A, const double* __restrict__ model, const double* __restrict__ S,
                                                    const WxPlan* __restrict__ plan, unsigned long long* __restrict__ spl,
                                                    unsigned int* __restrict__ tab) {
    extern __shared__ unsigned int wx_sk[];
    __shared__ unsigned int s_m;
    const int seg = blockIdx.x;
    if (seg >= plan->nseg) return;
    const ModelLayout ML = model_layout(M, P, A);
    const int j = plan->seg_j[seg], a1 = plan->seg_a[seg], as = plan->astar[j];
    const int t = threadIdx.x;
    constexpr int SAMP = 1024 * EPT;
    if (t == 0) s_m = 0;
    __syncthreads();
    unsigned int v[EPT], mine = 0;
    {
        // the rows' chains side by side, component by component: EPT independent loads per step (row after row, every load of a
        // chain waited for the one before)
        size_t ri[EPT];
        double zy[EPT], pred[EPT], esm[EPT];
        const double* Qj = model + ML.off_Q + j;
        const size_t nt = (size_t)g.nt;
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const unsigned long long q = (unsigned long long)u * 1024 + t;
            ri[u] = (size_t)((q * g.nt) / (unsigned long long)SAMP);
            zy[u] = wx_zy(Y, ldy, row_test + ri[u], j, model, ML, M);
            pred[u] = 0.0;
            esm[u] = 0.0;
        }
        for (int k = 0; k < as; k++) {
            const double qk = Qj[(size_t)P * k];
#pragma unroll
            for (int u = 0; u < EPT; u++) {
                pred[u] = fma(S[ri[u] + nt * k], qk, pred[u]);
                if (k + 1 == a1) esm[u] = zy[u] - pred[u];
            }
        }
#pragma unroll
        for (int u = 0; u < EPT; u++) {
            const double d = fabs(zy[u] - pred[u]) - fabs(esm[u]);
            const bool nzr = d != 0.0;
            v[u] = nzr ? (unsigned int)((unsigned long long)__double_as_longlong(fabs(d)) >> 31) : 0xffffffffu;
            mine += nzr ? 1u : 0u;
        }
    }
    if (mine) atomicAdd(&s_m, mine);
    wx_sort_regs<1024, EPT>(v, wx_sk);
#pragma unroll
    for (int u = 0; u < EPT; u++) wx_sk[t * EPT + u] = v[u];
    __syncthreads();
    const unsigned int m = s_m;                     // non-zero differences of the sample: the first m entries
    for (int b = t; b < g.NB - 1; b += 1024) {
        unsigned long long sv = ~0ull >> 1;          // an empty sample: everything in bin 0
        if (m) {
            unsigned long long idx = ((unsigned long long)(b + 1) * m) / (unsigned long long)g.NB;
            if (idx >= m) idx = m - 1;
            sv = ((unsigned long long)wx_sk[idx] << 31) | 0x7fffffffull;
        }
        spl[(size_t)seg * g.NB + b] = sv;
    }
    // the bounds sweep's fine bins, without a search: WX_TAB cells, linear in the key prefix between the sample's extremes; a cell
    // starts at fine bin (sample keys below it) NBF / m and spreads its keys linearly over the fine bins up to the next cell's start
    // -- a non-decreasing function of the key (any such function makes bins the bounds hold for; the sample only makes them
    // evenly filled).  tab[seg]: WX_TAB x (start | span << 16), the first prefix, the cell width's shift
    if (tab && g.F) {
        unsigned int* tb = tab + (size_t)seg * (WX_TAB + 2);
        const unsigned int NBF = (unsigned int)(g.NB * g.F);
        if (m == 0) {
            for (int c = t; c < WX_TAB + 2; c += 1024) tb[c] = 0u;
        } else {
            const unsigned int kmin = wx_sk[0], kmax = wx_sk[m - 1];
            unsigned int sh = 0;
            while (((kmax - kmin) >> sh) >= (unsigned int)WX_TAB) sh++;
            auto below = [&](unsigned long long bound) -> unsigned int {          // sample keys < bound
                if (bound > 0xffffffffull) return m;
                unsigned int lo = 0, hi = m;
                while (lo < hi) { const unsigned int mid = (lo + hi) >> 1; if ((unsigned long long)wx_sk[mid] < bound) lo = mid + 1; else hi = mid; }
                return lo;
            };
            for (int c = t; c < WX_TAB; c += 1024) {
                const unsigned int c0 = below((unsigned long long)kmin + ((unsigned long long)c << sh));
                const unsigned int c1 = below((unsigned long long)kmin + ((unsigned long long)(c + 1) << sh));
                const unsigned int s0 = (unsigned int)(((unsigned long long)c0 * NBF) / m);
                const unsigned int s1 = (c == WX_TAB - 1) ? NBF : (unsigned int)(((unsigned long long)c1 * NBF) / m);
                tb[c] = s0 | ((s1 - s0) << 16);
            }
            if (t == 0) { tb[WX_TAB] = kmin; tb[WX_TAB + 1] = sh; }
        }
    }
}

// exclusive scan of n <= WX_NBMAX counters by the WXT threads of a work-group (in -> out; in and out may be the same array);
// returns the total.  wsum: WXT / 64 + 1 words of LDS.  Barriers inside: call it from uniform control flow.
__device__ __forceinline__ unsigned int wx_block_scan(const unsigned int* in, unsigned int* out, int n, unsigned int* wsum) {
    const int t = threadIdx.x, per = (n + WXT - 1) / WXT, i0 = t * per;
    unsigned int loc = 0;
    for (int c = 0; c < per; c++) if (i0 + c < n) loc += in[i0 + c];
    unsigned int inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned int up = __shfl_up(inc, d, 64); if ((t & 63) >= d) inc += up; }
    __syncthreads();                                  // (every read of `in` is done: `out` may alias it)
    if ((t & 63) == 63) wsum[t >> 6] = inc;
    __syncthreads();
    unsigned int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WXT / 64; w++) { const unsigned int v = wsum[w]; if (w < (t >> 6)) before += v; total += v; }
    unsigned int run = before + inc - loc;
    for (int c = 0; c < per; c++)
        if (i0 + c < n) { const unsigned int v = in[i0 + c]; out[i0 + c] = run; run += v; }
    __syncthreads();
    return total;
}

// The two sweeps over the validation rows.  A work-group = (run of tiles, response j): its threads keep the residuals of R rows
// for all component counts in registers (the scores of a row are read once for all segments of the response) and go through
// the response's segments in groups of G (splitters and per-bin counters of a group live in LDS).
//   SCATTER == false: blockhist[st][seg][bin] = this work-group's keys of that bin
//   SCATTER == true : blockhist holds the work-group's offset inside the bin (k_wx_offsets), binbase the bin's start: the keys
//                     of a (tile, segment) are staged in LDS in bin order and written out as contiguous pieces.  (Measured and
//                     NOT kept: no staging -- the LDS counter of a (segment, bin) as the work-group's write cursor, a returning
//                     LDS atomic hands every key its place, no barrier in the loop: 0.56 ms against 0.44 ms; the 64 lanes of
//                     a store then hit 64 different lines.)
//   MODE 2 (the bounds sweep, before the other two): counts per FINE bin = (bin, one of F linear sub-bins of the bin's key range),
//                     all keys in the low and the positive differences in the high half of one 32-bit counter -> blockfine
// MODE 0 / 1 skip the tests whose verdict the bounds have settled (v3, wx_need): mostly all of them.
template <int AM, int R, int MODE>
__global__ __launch_bounds__(WXT) void k_wx_bin(const double* __restrict__ Y, size_t ldy, size_t row_test, WxGeo g, int M, int P, int A,
                                                const double* __restrict__ model, const double* __restrict__ S,
                                                const WxPlan* __restrict__ plan, const int* __restrict__ segbase,
                                                const unsigned long long* __restrict__ spl, unsigned int* __restrict__ blockhist,
                                                const unsigned int* __restrict__ binbase, unsigned long long* __restrict__ keys,
                                                const int* __restrict__ v3, const unsigned int* __restrict__ tab) {
    constexpr int TR = WXT * R;
    constexpr bool SCATTER = MODE == 1, FINE = MODE == 2;
    extern __shared__ unsigned long long wx_smem[];
    const int ST = FINE ? g.STb : g.ST, tps = FINE ? g.tpsb : g.tps;
    // MODE 0 / 1: work-groups that share an XCD (blockIdx % 8) take neighbouring runs of tiles: the pieces they write into a bin
    // are neighbours in memory and meet in that XCD's L2.
    // MODE 2 (one-dimensional grid): the P work-groups of one run of tiles -- one per response, all reading the same scores, once
    // per group of tests -- follow each other on ONE XCD, so the scores come from HBM once per group and from that XCD's L2 for
    // the other responses (with the response in grid.y they were fetched P times: 41 GB at 1e7 rows x 32 responses)
    const int per_x = (ST + 7) / 8;
    const int j = FINE ? (int)((blockIdx.x / 8) % (unsigned)P) : (int)blockIdx.y;
    const int st = FINE ? (int)(blockIdx.x / (8u * (unsigned)P)) * 8 + (int)(blockIdx.x % 8) : (int)(blockIdx.x % 8) * per_x + (int)(blockIdx.x / 8);
    if (st >= ST) return;
    const int as = plan->astar[j];
    if (as <= 1) return;
    const ModelLayout ML = model_layout(M, P, A);
    const int NB = g.NB, G = FINE ? g.Gb : g.G, t = threadIdx.x;
    const int F = FINE ? g.F : 1, NBF = NB * F;
    const size_t nt = (size_t)g.nt;
    const int seg0 = segbase[j], nsj = as - 1;
    // tests of this response that still need their exact rank sum (bit a1 - 1): undecided by the bounds, no smaller a' has passed
    unsigned int needmask = 0xffffffffu;
    if (!FINE) {
        needmask = 0;
        bool passed = false;
        for (int a1 = 1; a1 <= nsj; a1++) {
            const int v = v3[seg0 + a1 - 1];
            if (v == 2 && !passed) needmask |= 1u << (a1 - 1);
            passed = passed || v == 1;
        }
        if (!needmask) return;
    }
    // a splitter is a 32-bit prefix followed by 31 ones (k_wx_sample): splitter < key <=> prefix < key >> 31 -- the search compares
    // 32-bit words (half the LDS traffic and half the compare instructions of the 64-bit one)
    unsigned int* spl_s = (unsigned int*)wx_smem;                          // [G][NB] splitter prefixes; FINE: [G][WX_TAB + 2], the tables
    const int SPW = FINE ? WX_TAB + 2 : NB;
    unsigned int* acc = spl_s + (size_t)G * SPW;                           // [G][NB F]: counts (0, 2) / running global offsets (1)
    unsigned int* lh = acc + (size_t)G * NBF;                              // [NB] keys of the current (tile, segment) per bin
    unsigned int* cst = lh + NB;                                           // [NB] their exclusive scan
    unsigned long long* skey = (unsigned long long*)(((size_t)(cst + NB) + 7) & ~(size_t)7);   // [TR] staged keys (SCATTER)
    unsigned int* sdst = (unsigned int*)(skey + TR);                       // [TR] their places in the bin
    __shared__ unsigned int wsum[WXT / 64 + 1];
    const size_t tile0 = (size_t)st * tps;
    const double* Qj = model + ML.off_Q + j;
    WX_STAMP_K(SCATTER ? 3 : 2, 0);
    for (int grp = 0; grp * G < nsj; grp++) {
        const int gn = (nsj - grp * G < G) ? nsj - grp * G : G;            // segments of this group: a1 = grp G + 1 .. grp G + gn
        if (!FINE && ((needmask >> (grp * G)) & ((gn >= 32) ? 0xffffffffu : ((1u << gn) - 1u))) == 0) continue;      // (uniform)
        if (FINE) {
            for (int e = t; e < gn * SPW; e += WXT) spl_s[e] = tab[(size_t)(seg0 + grp * G) * SPW + e];
            for (int e = t; e < gn * NBF; e += WXT) acc[e] = 0u;
        } else
            for (int e = t; e < gn * NB; e += WXT) {
                const int gs = e / NB, b = e - gs * NB, seg = seg0 + grp * G + gs;
                spl_s[e] = (b < NB - 1) ? (unsigned int)(spl[(size_t)seg * NB + b] >> 31) : 0xffffffffu;
                acc[e] = SCATTER ? binbase[(size_t)seg * NB + b] + blockhist[((size_t)st * g.nseg_max + seg) * NB + b] : 0u;
            }
        __syncthreads();
        if (grp == 0) WX_STAMP_K(SCATTER ? 3 : 2, 1);
        for (int tt = 0; tt < tps; tt++) {
            const size_t row_t = (tile0 + tt) * TR;
            if (row_t >= nt) break;
            // residuals of this thread's R rows at 1 .. as components (pred: the k-ascending fma chain of the oracle)
            // All loads first, unconditionally (clamped rows and columns): with the loads inside the `k < as` branches of the chain
            // every one waited for the one before -- R AM memory latencies in a row were 70-90 % of all three sweeps (in-kernel stamps).
            double e[R][AM], estar[R], yv[R];
            bool in[R];
            const double sdy = model[ML.off_sd + M + j], muy = model[ML.off_mean + M + j];
#pragma unroll
            for (int r = 0; r < R; r++) {
                const size_t i = row_t + (size_t)r * WXT + t;
                in[r] = i < nt;
                const size_t ic = in[r] ? i : nt - 1;
                yv[r] = Y[row_test + ic + ldy * (size_t)j];
#pragma unroll
                for (int k = 0; k < AM; k++) e[r][k] = S[ic + nt * (size_t)(k < A ? k : A - 1)];
            }
#pragma unroll
            for (int r = 0; r < R; r++) {
                estar[r] = 0.0;
                const double zy = (sdy == 0.0) ? 0.0 : (yv[r] - muy) / sdy;        // (wx_zy)
                double pred = 0.0;
#pragma unroll
                for (int k = 0; k < AM; k++)
                    if (k < as) {
                        pred = fma(e[r][k], Qj[(size_t)P * k], pred);
                        e[r][k] = zy - pred;
                        if (k == as - 1) estar[r] = fabs(zy - pred);
                    }
            }
            if (grp == 0 && tt == 0) WX_STAMP_K(SCATTER ? 3 : 2, 2);
#pragma unroll
            for (int a1 = 1; a1 < AM; a1++) {
                if (a1 >= as || (a1 - 1) / G != grp) continue;            // (uniform over the work-group)
                if (!FINE && !((needmask >> (a1 - 1)) & 1u)) continue;
                const int gs = (a1 - 1) - grp * G;
                const unsigned int* sp = spl_s + (size_t)gs * SPW;
                if (SCATTER) { for (int b = t; b < NB; b += WXT) lh[b] = 0; __syncthreads(); }
                unsigned long long key[R];
                unsigned int k32[R];
                int bin[R];
                unsigned int rank[R];
                bool nzr[R];
#pragma unroll
                for (int r = 0; r < R; r++) {
                    const double d = in[r] ? estar[r] - fabs(e[r][a1 - 1]) : 0.0;
                    nzr[r] = d != 0.0;
                    const unsigned long long k63 = (unsigned long long)__double_as_longlong(fabs(d));
                    k32[r] = (unsigned int)(k63 >> 31);
                    key[r] = k63 | (d > 0.0 ? WX_SIGN : 0ull);
                    bin[r] = 0;
                }
                // bin = #{splitters < key}: NB is a power of two, so the search is log2(NB) steps for every row -- no data-dependent
                // trip count, and the R chains of dependent LDS reads run interleaved (a `while (lo < hi)` per row ran them one
                // after the other: 8 LDS round trips per key were most of this kernel)
                if (FINE) {          // fine bin from the table: one LDS read and one LDS atomic per key, no dependent chain
                    const unsigned int kmin = sp[WX_TAB], sh = sp[WX_TAB + 1];
                    unsigned int* ac = acc + (size_t)gs * NBF;
#pragma unroll
                    for (int r = 0; r < R; r++) {
                        if (!nzr[r]) continue;
                        unsigned int fine = 0;
                        if (k32[r] >= kmin) {
                            const unsigned int off = k32[r] - kmin, cell = off >> sh;
                            if (cell >= (unsigned int)WX_TAB) fine = (unsigned int)NBF - 1u;
                            else {
                                const unsigned int e2 = sp[cell];
                                fine = (e2 & 0xffffu) + (unsigned int)(((unsigned long long)(off & ((1u << sh) - 1u)) * (e2 >> 16)) >> sh);
                                fine = fine < (unsigned int)NBF - 1u ? fine : (unsigned int)NBF - 1u;
                            }
                        }
                        atomicAdd(&ac[fine], 1u + ((key[r] & WX_SIGN) ? 65536u : 0u));
                    }
                    continue;
                }
                for (int step = NB >> 1; step >= 1; step >>= 1) {
#pragma unroll
                    for (int r = 0; r < R; r++) bin[r] += (sp[bin[r] + step - 1] < k32[r]) ? step : 0;
                }
#pragma unroll
                for (int r = 0; r < R; r++) {
                    rank[r] = 0;
                    if (!nzr[r]) { bin[r] = -1; continue; }
                    if (SCATTER) rank[r] = atomicAdd(&lh[bin[r]], 1u);
                    else atomicAdd(&acc[(size_t)gs * NB + bin[r]], 1u);
                }
                if (SCATTER) {
                    __syncthreads();
                    const unsigned int total = wx_block_scan(lh, cst, NB, wsum);
                    unsigned int* go = acc + (size_t)gs * NB;
#pragma unroll
                    for (int r = 0; r < R; r++)
                        if (bin[r] >= 0) {
                            const unsigned int slot = cst[bin[r]] + rank[r];
                            skey[slot] = key[r];
                            sdst[slot] = go[bin[r]] + rank[r];
                        }
                    __syncthreads();
                    unsigned long long* out = keys + (size_t)(seg0 + a1 - 1) * nt;
                    for (unsigned int q = t; q < total; q += WXT) out[sdst[q]] = skey[q];
                    for (int b = t; b < NB; b += WXT) go[b] += lh[b];
                    __syncthreads();
                }
            }
            if (grp == 0 && tt == 0) WX_STAMP_K(SCATTER ? 3 : 2, 3);
            if (grp == 0 && tt == 0) WX_STAMP_K(SCATTER ? 3 : 2, 4);
        }
        __syncthreads();
        if (grp == 0) WX_STAMP_K(SCATTER ? 3 : 2, 5);
        if (!SCATTER)
            for (int e2 = t; e2 < gn * NBF; e2 += WXT) {
                const int gs = e2 / NBF, b = e2 - gs * NBF, seg = seg0 + grp * G + gs;
                blockhist[((size_t)st * g.nseg_max + seg) * NBF + b] = acc[e2];
            }
        __syncthreads();
    }
    WX_STAMP_K(SCATTER ? 3 : 2, 6);
}

// does test `seg` (of a response whose tests start at seg_first) still need its exact rank sum?  v3: 0 rejected / 1 passed by the
// bounds, 2 undecided; a test behind a smaller a' that passed is never looked at (k_wx_decide stops there)
__device__ __forceinline__ bool wx_need(const int* __restrict__ v3, int seg_first, int seg) {
    if (v3[seg] != 2) return false;
    for (int s = seg_first; s < seg; s++)
        if (v3[s] == 1) return false;
    return true;
}
// the verdict of a test at |W| / sigma = x, exactly as k_wx_decide takes it
__device__ double normalcdf_poly(double z);
__device__ __forceinline__ bool wx_passes(double x) { return 2.0 * (1.0 - normalcdf_poly(x)) > 0.1; }

// BOUNDS on the signed rank sum from counts alone.  Fine bin b of a test holds c_b keys, p_b of them positive differences, B_b keys
// lie below it: whatever the order inside the bin, its ranks are B_b + 1 .. B_b + c_b (average ranks of ties: a doubly stochastic
// mix of those, which moves no subset sum beyond the extremes), so the positives' rank sum lies between the p_b lowest and the p_b
// highest of them and   2 W_b in [4 p B + 2 p (p + 1), 4 p B + 4 p c - 2 p (p - 1)] - (2 c B + c (c + 1)).
// Integers below 2^50, summed exactly.  The interval of |W| / sigma, widened by 1e-12 against the roundings of the division, is put
// through the decision function of k_wx_decide at both ends: equal answers = THE answer (the function is monotone but for the last
// bits next to its threshold); else the test stays undecided (2) and goes through the exact sweeps.  With ~3 sqrt(n) fine bins the
// interval is ~0.3 sigma wide: a test is undecided when its statistic lies within that of the threshold.
__global__ __launch_bounds__(1024) void k_wx_bounds(WxGeo g, const WxPlan* __restrict__ plan, const unsigned int* __restrict__ blockfine,
                                                    unsigned long long* __restrict__ nz, int* __restrict__ v3,
                                                    int* __restrict__ undecided) {
    extern __shared__ unsigned int wxb_cp[];              // [NBF] packed (all keys, positive keys) of the test's fine bins
    __shared__ long long red[3][16];
    __shared__ unsigned int wtot[16];
    const int seg = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (seg >= plan->nseg) return;
    const int NBF = g.NB * g.F;
    const size_t stride = (size_t)g.nseg_max * NBF;
    for (int b = t; b < NBF; b += 1024) {
        const unsigned int* src = blockfine + (size_t)seg * NBF + b;
        unsigned int c = 0, p = 0;
        int st = 0;
        for (; st + 8 <= g.STb; st += 8) {
            unsigned int v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = src[(size_t)(st + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; u++) { c += v[u] & 0xffffu; p += v[u] >> 16; }
        }
        for (; st < g.STb; st++) { const unsigned int v = src[(size_t)st * stride]; c += v & 0xffffu; p += v >> 16; }
        wxb_cp[2 * b] = c;
        wxb_cp[2 * b + 1] = p;
    }
    __syncthreads();
    // thread t owns the consecutive fine bins [t per, (t + 1) per): keys below them by a work-group scan of the threads' totals
    const int per = (NBF + 1023) / 1024, b0 = t * per;
    unsigned int loc = 0;
    for (int i = 0; i < per; i++) if (b0 + i < NBF) loc += wxb_cp[2 * (b0 + i)];
    unsigned int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    unsigned int below = inc - loc;
    for (int w = 0; w < wave; w++) below += wtot[w];
    long long lo2 = 0, hi2 = 0, m = loc;
    for (int i = 0; i < per; i++)
        if (b0 + i < NBF) {
            const long long c = wxb_cp[2 * (b0 + i)], p = wxb_cp[2 * (b0 + i) + 1], B = below;
            const long long all2 = 2 * c * B + c * (c + 1);
            lo2 += 4 * p * B + 2 * p * (p + 1) - all2;
            hi2 += 4 * p * B + 4 * p * c - 2 * p * (p - 1) - all2;
            below += (unsigned int)c;
        }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lo2 += __shfl_xor(lo2, o, 64); hi2 += __shfl_xor(hi2, o, 64); m += __shfl_xor(m, o, 64); }
    if (lane == 0) { red[0][wave] = lo2; red[1][wave] = hi2; red[2][wave] = m; }
    __syncthreads();
    if (t == 0) {
        lo2 = hi2 = m = 0;
        for (int w = 0; w < 16; w++) { lo2 += red[0][w]; hi2 += red[1][w]; m += red[2][w]; }
        nz[seg] = (unsigned long long)m;
        int v = 1;                                         // no non-zero difference: p = 1, the test passes (k_wx_decide)
        if (m > 0) {
            const double md = (double)m, sigma = sqrt(md * (md + 1.0) * (2.0 * md + 1.0) / 6.0);
            const long long alo = lo2 < 0 ? -lo2 : lo2, ahi = hi2 < 0 ? -hi2 : hi2;
            const long long mx2 = alo > ahi ? alo : ahi, mn2 = (lo2 <= 0 && hi2 >= 0) ? 0 : (alo < ahi ? alo : ahi);
            const double x_lo = 0.5 * (double)mn2 / sigma * (1.0 - 1e-12), x_hi = 0.5 * (double)mx2 / sigma * (1.0 + 1e-12);
            const bool p_lo = wx_passes(x_lo), p_hi = wx_passes(x_hi);
            v = (p_lo == p_hi) ? (p_lo ? 1 : 0) : 2;
        }
        v3[seg] = v;
        if (v == 2) atomicAdd(undecided, 1);
    }
}

// per segment: blockhist[st][seg][b] -> the work-group's offset inside bin b (exclusive over st), hist[seg][b] = the bin's size,
// binbase[seg][b] = keys in front of the bin, nz[seg] = non-zero differences of the test.  1024 threads = 256 bins x 4 quarters of
// the work-groups: partial sums per quarter (independent loads, eight in flight), then the offsets in a second sweep
__global__ __launch_bounds__(1024) void k_wx_offsets(WxGeo g, const WxPlan* __restrict__ plan, unsigned int* __restrict__ blockhist,
                                                     unsigned int* __restrict__ hist, unsigned int* __restrict__ binbase,
                                                     unsigned long long* __restrict__ nz, const int* __restrict__ v3,
                                                     const int* __restrict__ segbase) {
    __shared__ unsigned int tot[WX_NBMAX];
    __shared__ unsigned int part[4][256];
    const int seg = blockIdx.x, NB = g.NB;
    if (seg >= plan->nseg) return;
    if (!wx_need(v3, segbase[plan->seg_j[seg]], seg)) return;          // (uniform: settled by the bounds)
    const int bl = threadIdx.x & 255, q = threadIdx.x >> 8;
    const int per = (g.ST + 3) / 4, s0 = q * per, s1 = (s0 + per < g.ST) ? s0 + per : g.ST;
    const size_t stride = (size_t)g.nseg_max * NB;
    for (int b0 = 0; b0 < NB; b0 += 256) {
        const int b = b0 + bl;
        unsigned int sum = 0;
        if (b < NB) {
            const unsigned int* src = blockhist + (size_t)seg * NB + b;
            int st = s0;
            for (; st + 8 <= s1; st += 8) {
                unsigned int v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = src[(size_t)(st + u) * stride];
#pragma unroll
                for (int u = 0; u < 8; u++) sum += v[u];
            }
            for (; st < s1; st++) sum += src[(size_t)st * stride];
        }
        part[q][bl] = sum;
        __syncthreads();
        if (b < NB) {
            unsigned int run = 0;
            for (int qq = 0; qq < q; qq++) run += part[qq][bl];
            unsigned int* dst = blockhist + (size_t)seg * NB + b;
            int st = s0;
            for (; st + 8 <= s1; st += 8) {
                unsigned int v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = dst[(size_t)(st + u) * stride];
#pragma unroll
                for (int u = 0; u < 8; u++) { dst[(size_t)(st + u) * stride] = run; run += v[u]; }
            }
            for (; st < s1; st++) { const unsigned int v = dst[(size_t)st * stride]; dst[(size_t)st * stride] = run; run += v; }
            if (q == 3) { tot[b] = run; hist[(size_t)seg * NB + b] = run; }
        }
        __syncthreads();
    }
    unsigned int total = 0;
    if (threadIdx.x < 64) {         // NB <= 2048 counters: one wave, NB / 64 per lane
        const int perl = (NB + 63) / 64, i0 = threadIdx.x * perl;
        unsigned int loc = 0;
        for (int c = 0; c < perl; c++) if (i0 + c < NB) loc += tot[i0 + c];
        unsigned int inc = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned int up = __shfl_up(inc, d, 64); if ((int)threadIdx.x >= d) inc += up; }
        unsigned int run = inc - loc;
        for (int c = 0; c < perl; c++)
            if (i0 + c < NB) { const unsigned int v = tot[i0 + c]; binbase[(size_t)seg * NB + i0 + c] = run; run += v; }
        total = __shfl(inc, 63, 64);
        if (threadIdx.x == 0) nz[seg] = total;
    }
}

// bitonic sort of T SPT 64-bit keys, SPT per thread in registers (element t SPT + u): compare-exchanges inside a thread stay in
// registers, strides inside a wave are shuffles, only the strides across waves go through LDS.  lds: T SPT keys.
template <int T, int SPT>
__device__ __forceinline__ void wx_sort64(unsigned long long (&v)[SPT], unsigned long long* lds) {
    constexpr int N = T * SPT;
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j < SPT) {
#pragma unroll
                for (int u = 0; u < SPT; u++) {
                    const int x = u ^ j;
                    if (x > u) {
                        const bool asc = ((t * SPT + u) & k) == 0;
                        const unsigned long long a = v[u], b = v[x];
                        if ((a > b) == asc) { v[u] = b; v[x] = a; }
                    }
                }
            } else {
                const int m = j / SPT;                                      // the partner thread is t ^ m
                const bool lower = (t & m) == 0;
                if (m >= 64) {
#pragma unroll
                    for (int u = 0; u < SPT; u++) lds[u * T + t] = v[u];
                    __syncthreads();
                }
#pragma unroll
                for (int u = 0; u < SPT; u++) {
                    unsigned long long o;
                    if (m >= 64) o = lds[u * T + (t ^ m)];
                    else {
                        const unsigned int lo32 = (unsigned int)__shfl_xor((int)(unsigned int)v[u], m, 64);
                        const unsigned int hi32 = (unsigned int)__shfl_xor((int)(unsigned int)(v[u] >> 32), m, 64);
                        o = ((unsigned long long)hi32 << 32) | lo32;
                    }
                    const bool asc = ((t * SPT + u) & k) == 0;
                    const unsigned long long mn = v[u] < o ? v[u] : o, mx = v[u] < o ? o : v[u];
                    v[u] = (lower == asc) ? mn : mx;
                }
                if (m >= 64) __syncthreads();
            }
        }
    }
}

// The signed sum of the average ranks of one bin's keys: the bin is SORTED (magnitude in the upper 63 bits of the sort word, the
// sign below it) in registers -- SPT keys per thread, wx_sort64 -- and a key's rank is its position; a key with an equal
// neighbour looks up the ends of its tie run in the sorted copy in LDS.
// How this got here (in-kernel stamps, scripts/wx_stamps.sh, 1e6 particles, cycles per work-group of 2000 keys): sub-bins by
// counting -- first linear over the bin's key range (a single tiny |d| stretches the range over hundreds of binades and one sub-bin
// takes the whole bin), then by splitters sampled from the bin -- and a walk through the own sub-bin cost 112 000 (one LDS
// round trip per (key, position), as many rounds as the wave's longest sub-bin), 70 000 with four times the samples and
// branch-free rounds: sorting a quarter of the keys as a sample, searching it, counting, placing and walking is more work than
// sorting all of them once.
// T threads, SPT keys per thread.  ks: T SPT keys (exchange buffer of the sort, then the sorted bin); red: T / 64 doubles.
// Returns the sum in thread 0.
template <int T, int SPT>
__device__ __forceinline__ double wx_bin_ranksum(const unsigned long long* __restrict__ src, unsigned int n, unsigned int base,
                                                 unsigned long long* ks, double* red) {
    const int t = threadIdx.x;
    unsigned long long v[SPT];
#pragma unroll
    for (int u = 0; u < SPT; u++) {
        const unsigned int p = (unsigned int)t + (unsigned int)u * T;
        const unsigned long long k = p < n ? src[p] : ~0ull;               // (past the end: the largest word, sorted behind every key)
        v[u] = p < n ? ((k << 1) | (k >> 63)) : ~0ull;
    }
    wx_sort64<T, SPT>(v, ks);
#pragma unroll
    for (int u = 0; u < SPT; u++) ks[t * SPT + u] = v[u];
    __syncthreads();
    double w = 0.0;
#pragma unroll
    for (int u = 0; u < SPT; u++) {
        const unsigned int e = (unsigned int)(t * SPT + u);
        if (e >= n) continue;
        const unsigned long long m = v[u] >> 1;
        const bool tie = (e > 0 && (ks[e - 1] >> 1) == m) || (e + 1 < n && (ks[e + 1] >> 1) == m);
        double rank = (double)base + (double)e + 1.0;
        if (tie) {                                                           // ends of the run of equal magnitudes: two binary searches
            unsigned int lo = 0, hi = e;                                     // first position whose magnitude is not below m
            while (lo < hi) { const unsigned int mid = (lo + hi) >> 1; if ((ks[mid] >> 1) < m) lo = mid + 1; else hi = mid; }
            unsigned int lo2 = e, hi2 = n;                                   // first position whose magnitude is above m
            while (lo2 < hi2) { const unsigned int mid = (lo2 + hi2) >> 1; if ((ks[mid] >> 1) <= m) lo2 = mid + 1; else hi2 = mid; }
            rank = (double)base + ((double)lo + (double)(lo2 - 1)) * 0.5 + 1.0;     // ranker.h:74-75 "average"
        }
        w += (v[u] & 1ull) ? rank : -rank;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o, 64);            // exact: multiples of 1/2 below 2^52
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = w;
    __syncthreads();
    double tw = 0.0;
    if (t == 0)
        for (int ww = 0; ww < T / 64; ww++) tw += red[ww];
    __syncthreads();
    return tw;
}

// The common bin, without a sort: a quantile slice of the keys is close to uniform between its ends, so NS = 1024 LINEAR sub-bins
// (by counting) hold two keys each on average, and a key's rank is keys in front of the bin + keys in lower sub-bins + smaller keys
// of its own sub-bin + (equal ones, itself included, + 1) / 2 -- a walk of a handful of LDS reads.  Measured per work-group of 2000
// keys (in-kernel stamps, scripts/wx_stamps.sh): sorting the bin in registers (wx_bin_ranksum) 72 000 cycles, sub-bins from sampled
// splitters (sort a sample, search it, count, place, walk) 70 000 - 112 000, this 25 000.  What linear sub-bins cannot take is a
// bin whose keys are NOT spread over its range -- the first and the last bin of a test (a single tiny |d| stretches the range over
// hundreds of binades), heavy ties: a sub-bin above WX_WALK keys sends the bin to k_wx_ranks_big, which sorts it.
constexpr int WX_CAP_S = 4096;                  // keys of a bin this kernel takes (16 per thread)
constexpr int WX_NS = 1024;                     // linear sub-bins
constexpr int WX_WALK = 48;                     // longest sub-bin it walks
__global__ __launch_bounds__(256) void k_wx_ranks(WxGeo g, const WxPlan* __restrict__ plan, const unsigned long long* __restrict__ keys,
                                                  const unsigned int* __restrict__ hist, const unsigned int* __restrict__ binbase,
                                                  double* __restrict__ W, unsigned int* __restrict__ big /* [0] count, then (seg, bin) pairs */,
                                                  const int* __restrict__ v3, const int* __restrict__ segbase) {
    constexpr int T = 256, KPT = WX_CAP_S / T;
    __shared__ unsigned long long ks[WX_CAP_S + 1];
    __shared__ unsigned int cnt[WX_NS + 2], start[WX_NS + 2];
    __shared__ unsigned long long rmm[2 * T / 64];
    __shared__ double red[T / 64];
    __shared__ unsigned int s_max[T / 64];
    const int seg = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    if (seg >= plan->nseg) return;
    if (!wx_need(v3, segbase[plan->seg_j[seg]], seg)) return;
    const unsigned int n = hist[(size_t)seg * g.NB + b];
    if (n == 0) return;
    auto to_big = [&]() { if (t == 0) { const unsigned int e = atomicAdd(&big[0], 1u); big[1 + 2 * e] = (unsigned int)seg; big[2 + 2 * e] = (unsigned int)b; } };
    if (n > (unsigned int)WX_CAP_S) { to_big(); return; }
    const unsigned int base = binbase[(size_t)seg * g.NB + b];
    const unsigned long long* src = keys + (size_t)seg * g.nt + base;
    const int kpt = (int)((n + T - 1) / T);                                 // key slots in use (uniform over the work-group)
    WX_STAMP(0);
    unsigned long long k[KPT];
    bool valid[KPT];
    unsigned long long mn = ~0ull, mx = 0ull;
#pragma unroll
    for (int u = 0; u < KPT; u++) {
        const unsigned int p = (unsigned int)t + (unsigned int)u * T;
        valid[u] = u < kpt && p < n;
        k[u] = valid[u] ? src[p] : 0ull;
        const unsigned long long k63 = k[u] & WX_MASK;
        mn = (valid[u] && k63 < mn) ? k63 : mn;
        mx = (valid[u] && k63 > mx) ? k63 : mx;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const unsigned long long a = (unsigned long long)__shfl_xor((long long)mn, o, 64), c = (unsigned long long)__shfl_xor((long long)mx, o, 64);
        mn = a < mn ? a : mn;
        mx = c > mx ? c : mx;
    }
    if ((t & 63) == 0) { rmm[t >> 6] = mn; rmm[T / 64 + (t >> 6)] = mx; }
    for (int i = t; i < WX_NS + 2; i += T) cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < T / 64; w++) { mn = rmm[w] < mn ? rmm[w] : mn; mx = rmm[T / 64 + w] > mx ? rmm[T / 64 + w] : mx; }
    WX_STAMP(1);
    double w = 0.0;
    if (mn == mx) {                                                          // one value: every key has the bin's middle rank
        const double rank = (double)base + ((double)n + 1.0) * 0.5;
#pragma unroll
        for (int u = 0; u < KPT; u++)
            if (valid[u]) w += (k[u] & WX_SIGN) ? rank : -rank;
    } else {
        // sub-bin = floor(q scale) with q = (key - min) >> sh < 2^22 (exact in f32) and scale = NS / (qmax + 1): a product of a float
        // and a positive constant is monotone in q, so sub-bins are ordered like the keys and equal keys share one
        const unsigned long long range = mx - mn;
        const int bl = 64 - __clzll((long long)range), sh = bl > 22 ? bl - 22 : 0;
        const float scale = (float)WX_NS / ((float)(unsigned int)(range >> sh) + 1.0f);
        int sb[KPT];
        unsigned int within[KPT];
#pragma unroll
        for (int u = 0; u < KPT; u++) {
            const int v = (int)((float)(unsigned int)(((k[u] & WX_MASK) - mn) >> sh) * scale);
            sb[u] = valid[u] ? (v < WX_NS ? v : WX_NS - 1) : WX_NS + 1;      // (an unused slot counts in the spare counter)
            within[u] = 0;
            if (u < kpt) within[u] = atomicAdd(&cnt[sb[u]], 1u);             // the key's place inside its sub-bin: arrival order
        }
        __syncthreads();
        WX_STAMP(2);
        // exclusive scan of the NS counters (NS / T consecutive ones per thread) and their largest
        constexpr int PER = WX_NS / T;
        unsigned int c[PER], loc = 0, big_c = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) { c[i] = cnt[t * PER + i]; loc += c[i]; big_c = c[i] > big_c ? c[i] : big_c; }
        unsigned int inc = loc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const unsigned int up = __shfl_up(inc, d, 64); if ((t & 63) >= d) inc += up; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const unsigned int v = (unsigned int)__shfl_xor((int)big_c, o, 64); big_c = v > big_c ? v : big_c; }
        unsigned int* rws = (unsigned int*)rmm;
        if ((t & 63) == 63) rws[t >> 6] = inc;
        if ((t & 63) == 0) s_max[t >> 6] = big_c;
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < T / 64; ww++) big_c = s_max[ww] > big_c ? s_max[ww] : big_c;
        if (big_c > (unsigned int)WX_WALK) { to_big(); return; }            // (uniform: every thread sees the same maximum)
        unsigned int run = inc - loc;
#pragma unroll
        for (int ww = 0; ww < T / 64; ww++) if (ww < (t >> 6)) run += rws[ww];
#pragma unroll
        for (int i = 0; i < PER; i++) { start[t * PER + i] = run; run += c[i]; }
        if (t == T - 1) { start[WX_NS] = run; start[WX_NS + 1] = n; }       // (the spare sub-bin: the spare slot ks[n])
        __syncthreads();
#pragma unroll
        for (int u = 0; u < KPT; u++)
            if (u < kpt) ks[start[sb[u]] + (valid[u] ? within[u] : 0u)] = k[u];
        __syncthreads();
        WX_STAMP(3);
        // the walk through the own sub-bin: position q of every slot's sub-bin in turn, as many rounds as the longest sub-bin any lane
        // of the wave has; nothing inside branches on the lane (a slot past its end reads its first position again and does not count
        // it), so the slots' LDS reads of a round are in flight together
        unsigned int s0[KPT], len[KPT], acc[KPT], longest = 0;              // acc: smaller keys (low half) | equal keys (high half)
#pragma unroll
        for (int u = 0; u < KPT; u++) {
            s0[u] = 0; len[u] = 0; acc[u] = 0;
            if (valid[u]) {
                s0[u] = start[sb[u]];
                len[u] = start[sb[u] + 1] - s0[u];
                longest = len[u] > longest ? len[u] : longest;
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { const unsigned int v = (unsigned int)__shfl_xor((int)longest, o, 64); longest = v > longest ? v : longest; }
        for (unsigned int q = 0; q < longest; q++) {
#pragma unroll
            for (int u = 0; u < KPT; u++)
                if (u < kpt) {
                    const bool in = q < len[u];
                    const unsigned long long o = ks[s0[u] + (in ? q : 0u)] & WX_MASK, k63 = k[u] & WX_MASK;
                    acc[u] += (in && o < k63 ? 1u : 0u) + (in && o == k63 ? 0x10000u : 0u);
                }
        }
        WX_STAMP(4);
#pragma unroll
        for (int u = 0; u < KPT; u++)
            if (valid[u]) {
                const double rank = (double)base + (double)(s0[u] + (acc[u] & 0xffffu)) + ((double)(acc[u] >> 16) + 1.0) * 0.5;
                w += (k[u] & WX_SIGN) ? rank : -rank;
            }
    }
    WX_STAMP(5);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) w += __shfl_xor(w, o, 64);            // exact: multiples of 1/2 below 2^52
    if ((t & 63) == 0) red[t >> 6] = w;
    __syncthreads();
    if (t == 0) {
        double tw = 0.0;
        for (int ww = 0; ww < T / 64; ww++) tw += red[ww];
        if (tw != 0.0) atomicAdd(&W[seg], tw);
    }
    WX_STAMP(6);
}
// the bins k_wx_ranks passes on (above WX_CAP_S keys: a sparse sample quantile; keys not spread over the bin's range: the ends of a
// test, heavy ties): work-groups of 1024 threads with 128 KB of LDS walk the list and SORT each bin (wx_bin_ranksum); a bin above
// WX_CAP keys raises the flag that sends the reduction to the sorted path
__global__ __launch_bounds__(1024) void k_wx_ranks_big(WxGeo g, const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ hist,
                                                       const unsigned int* __restrict__ binbase, double* __restrict__ W,
                                                       const unsigned int* __restrict__ big, int* __restrict__ fail) {
    extern __shared__ unsigned long long wx_big_ks[];                       // WX_CAP keys
    __shared__ double red[32];
    const unsigned int nbig = big[0];
    for (unsigned int e = blockIdx.x; e < nbig; e += gridDim.x) {
        const unsigned int seg = big[1 + 2 * e], b = big[2 + 2 * e];
        const unsigned int n = hist[(size_t)seg * g.NB + b];
        if (n > (unsigned int)WX_CAP) { if (threadIdx.x == 0) *fail = 1; continue; }    // (the host repeats the reduction on the sorted path)
        const unsigned int base = binbase[(size_t)seg * g.NB + b];
        const unsigned long long* src = keys + (size_t)seg * g.nt + base;
        double tw;                                                           // (the network's length goes with the padded size)
        if (n <= 2048) tw = wx_bin_ranksum<1024, 2>(src, n, base, wx_big_ks, red);
        else if (n <= 4096) tw = wx_bin_ranksum<1024, 4>(src, n, base, wx_big_ks, red);
        else if (n <= 8192) tw = wx_bin_ranksum<1024, 8>(src, n, base, wx_big_ks, red);
        else tw = wx_bin_ranksum<1024, 16>(src, n, base, wx_big_ks, red);
        if (threadIdx.x == 0 && tw != 0.0) atomicAdd(&W[seg], tw);
    }
}

__device__ double normalcdf_poly(double z) {        // [PLS] normalcdf, Abramowitz & Stegun 26.2.18
    const double c1 = 0.196854, c2 = 0.115194, c3 = 0.000344, c4 = 0.019527;
    const double x = fabs(z);
    const double d = 1.0 + c1 * x + c2 * x * x + c3 * x * x * x + c4 * x * x * x * x;
    const double tail = 0.5 / (d * d * d * d);
    return (z >= 0.0) ? 1.0 - tail : tail;
}

// pass[s] (optional scratch of nseg_max bytes): the test's verdict, computed by one thread per test; thread 0 then walks the
// responses (single-threaded it was 50 us of square roots and divisions in a row at 112 tests)
__global__ void k_wx_decide(double* __restrict__ model, int M, int P, int A, const WxPlan* __restrict__ plan,
                            const unsigned long long* __restrict__ nz, const double* __restrict__ W, unsigned char* __restrict__ pass,
                            const int* __restrict__ v3) {
    const int nseg = plan->nseg;
    auto verdict = [&](int s) -> bool {
        if (v3 && v3[s] != 2) return v3[s] == 1;           // settled by the bounds (k_wx_bounds): the exact sum was never taken
        const double m = (double)nz[s];
        double p = 1.0;
        if (m > 0.0) {
            const double sigma = sqrt(m * (m + 1.0) * (2.0 * m + 1.0) / 6.0);
            p = 2.0 * (1.0 - normalcdf_poly(fabs(W[s] / sigma)));
        }
        return p > 0.1;
    };
    if (pass) {
        for (int s = threadIdx.x; s < nseg; s += blockDim.x) pass[s] = verdict(s) ? 1 : 0;
        __syncthreads();
    }
    const ModelLayout ML = model_layout(M, P, A);
    __shared__ int s_ncomp;
    if (threadIdx.x == 0) s_ncomp = 1;
    __syncthreads();
    for (int j = threadIdx.x; j < P; j += blockDim.x) {                     // a response per thread: its tests are consecutive segments
        int s = 0;
        for (int jj = 0; jj < j; jj++) { const int as = plan->astar[jj]; s += as > 1 ? as - 1 : 0; }
        const int as = plan->astar[j];
        int best = as;
        for (int a = 1; a < as; a++, s++) {
            if (s >= nseg) break;
            if (pass ? pass[s] != 0 : verdict(s)) { best = a; break; }
        }
        model[ML.off_per + j] = (double)best;
        atomicMax(&s_ncomp, best);
    }
    __syncthreads();
    if (threadIdx.x == 0) model[ML.off_hdr] = (double)s_ncomp;
}

// observed scores do not depend on ncomp (all A are stored), nothing else to refresh

}  // namespace

// The SORTED path (rounds 1-3): every (key, segment) pair of all tests in one stable LSD radix sort.  Any shape; the fallback of
// the binned path.  Allocates from the arena: the caller has reserved abc_wx_sorted_need().
static int launch_wilcoxon_sorted(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                    size_t P, size_t A, size_t row_test, double* model) {
    if (row_test >= n) return ABC_OK;                    // empty validation set: nothing to reduce
    if (P * (A - 1) > MAXSEG)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "wilcoxon: P (A - 1) = %zu tests, more than %zu", P * (A - 1), MAXSEG);
    const size_t nt = n - row_test;
    const size_t nseg_max = P * (A - 1);
    if (nseg_max == 0) return ABC_OK;
    WxPlan* plan = (WxPlan*)abc_ws_alloc(ctx, sizeof(WxPlan));
    int* seg_j = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    int* seg_a = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    int* astar = (int*)abc_ws_alloc(ctx, P * sizeof(int));
    unsigned long long* nz = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * 8);
    double* W = (double*)abc_ws_alloc(ctx, nseg_max * 8);
    double* S = (double*)abc_ws_alloc(ctx, nt * A * 8);
    unsigned long long* key0 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    unsigned long long* val0 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    unsigned long long* key1 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    unsigned long long* val1 = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    if (!plan || !seg_j || !seg_a || !astar || !nz || !W || !S || !key0 || !val0 || !key1 || !val1)
        ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%zu segments x %zu rows)", nseg_max, nt);
    hipLaunchKernelGGL(k_wx_plan, dim3(1), dim3(256), 0, ctx->stream, model, (int)M, (int)P, (int)A, plan, seg_j, seg_a, astar,
                       (int)nseg_max, nz, W, (int*)nullptr, (int*)nullptr);
    // the number of segments actually needed lives on the device; size the grid for the maximum, idle blocks exit
    int nseg_host = 0;
    ABC_HIP(ctx, hipMemcpyAsync(&nseg_host, &plan->nseg, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (nseg_host == 0) return ABC_OK;
    const unsigned rb = (unsigned)((nt + 255) / 256);
    int KC = 1;
    while (KC < (int)A) KC *= 2;
    if (A > 32) {
        hipLaunchKernelGGL(k_wx_scores_wide, dim3(rb), dim3(256), 0, ctx->stream, X, ldx, row_test, nt, (int)M, (int)P, (int)A, model, S);
        KC = 0;
    }
#define LAUNCH_SC(KCV) hipLaunchKernelGGL(k_wx_scores<KCV>, dim3(rb), dim3(256), 0, ctx->stream, X, ldx, row_test, nt, \
                                          (int)M, (int)P, (int)A, model, S)
    switch (KC) {
        case 0: break;
        case 1: LAUNCH_SC(1); break;
        case 2: LAUNCH_SC(2); break;
        case 4: LAUNCH_SC(4); break;
        case 8: LAUNCH_SC(8); break;
        case 16: LAUNCH_SC(16); break;
        default: LAUNCH_SC(32); break;
    }
#undef LAUNCH_SC
    hipLaunchKernelGGL(k_wx_diffs, dim3(rb, nseg_host), dim3(256), 0, ctx->stream, Y, ldy, row_test, nt, (int)M, (int)P,
                       (int)A, model, S, plan, key0, val0, nz);
    ABC_HIP(ctx, hipGetLastError());
    const size_t tot = (size_t)nseg_host * nt;
    // ascending |d| (8 byte passes), then stable by segment (payload bytes 4-5) -> segment-major, |d| ascending
    ABC_TRY(abc_sort_u64_bytes(ctx, key0, val0, key1, val1, tot, 0, 8));
    ABC_TRY(abc_sort_u64_bytes(ctx, val0, key0, val1, key1, tot, 4, 6));
    hipLaunchKernelGGL(k_wx_ranksum, dim3(rb, nseg_host), dim3(256), 0, ctx->stream, key0, val0, nt, plan, nz, W);
    hipLaunchKernelGGL(k_wx_decide, dim3(1), dim3(64), 0, ctx->stream, model, (int)M, (int)P, (int)A, plan, nz, W, (unsigned char*)nullptr, (const int*)nullptr);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

static size_t wx_sorted_need(size_t nt, size_t P, size_t A) {
    const size_t seg = P * (A > 0 ? A - 1 : 0);
    // (scores, four key / value buffers, the per-chunk digit histograms of the two radix sorts, the plan)
    return nt * A * 8 + 4 * seg * nt * 8 + 2 * 256 * ((seg * nt) / 1024 + 2) * 4 + seg * 32 + P * 8 + (2u << 20);
}

static bool wx_no_bounds() {            // A/B runs and tests (ABC_DIAG=1 ABC_WX_NOBOUNDS=1): every test through the exact sweeps
    static const bool on = abc_diag_env("ABC_WX_NOBOUNDS") != nullptr;
    return on;
}
// geometry of the binned path; false: the shape goes to the sorted path
static bool wx_geometry(size_t nt, size_t P, size_t A, WxGeo* g, int* R_out) {
    if (A > 32 || A < 2 || nt == 0 || P * (A - 1) > MAXSEG) return false;
    int NB = 1;
    if (nt > (size_t)WX_CAP / 2) while ((size_t)NB * 2048 < nt) NB *= 2;
    if (NB > WX_NBMAX || nt / (size_t)NB > 3500) return false;
    if (NB > 1 && nt < (size_t)(NB <= 256 ? 4096 : (NB <= 1024 ? 16384 : 32768))) return false;
    const int R = A <= 8 ? 4 : (A <= 16 ? 2 : 1);
    const size_t TR = (size_t)WXT * R, tiles = (nt + TR - 1) / TR;
    size_t tps = (tiles * P + 4095) / 4096;
    if (tps < 1) tps = 1;
    g->nt = nt;
    g->NB = NB;
    g->SAMP = NB <= 256 ? 4096 : (NB <= 1024 ? 16384 : 32768);      // >= 16 sampled rows per bin (k_wx_sample<SAMP / 1024>); NB > 1: nt > 4096
    g->tps = (int)tps;
    g->ST = (int)((tiles + tps - 1) / tps);
    int G = (int)((96u << 10) / ((size_t)NB * 8));         // splitter prefixes (4 bytes) + counters (4) of a group's bins in <= 96 KB of LDS
    if (G < 1) G = 1;
    if (G > (int)A - 1) G = (int)A - 1;
    g->G = G;
    g->nseg_max = (int)(P * (A - 1));
    // the bounds sweep: ~2.5 sqrt(nt) fine bins and up (an interval of <= 0.35 sigma), at most 8192; a work-group's rows below 2^16
    g->F = 0; g->STb = 0; g->tpsb = 0; g->Gb = 0;
    if (NB > 1 && !wx_no_bounds()) {
        size_t nbf = (size_t)NB;
        while (nbf < 8192 && (double)nbf < 2.5 * sqrt((double)nt)) nbf *= 2;
        g->F = (int)(nbf / (size_t)NB);
        size_t stb = (1024 + P - 1) / P;
        if (stb > tiles) stb = tiles;
        size_t tpsb = (tiles + stb - 1) / stb;
        const size_t tmax = 61440 / TR;
        if (tpsb > tmax) tpsb = tmax;
        g->tpsb = (int)tpsb;
        g->STb = (int)((tiles + tpsb - 1) / tpsb);
        int Gb = (int)((72u << 10) / ((size_t)(WX_TAB + 2) * 4 + nbf * 4));
        if (Gb < 1) Gb = 1;
        if (Gb > (int)A - 1) Gb = (int)A - 1;
        g->Gb = Gb;
    }
    *R_out = R;
    return true;
}
static size_t wx_binned_need(size_t nt, size_t P, size_t A) {
    WxGeo g;
    int R;
    if (!wx_geometry(nt, P, A, &g, &R)) return 0;
    const size_t seg = P * (A - 1);
    return nt * A * 8 + seg * nt * 8 + (size_t)g.ST * seg * g.NB * 4 + (size_t)g.STb * seg * g.NB * g.F * 4 + seg * g.NB * (8 + 4 + 4 + 8) +
           seg * (40 + (size_t)(WX_TAB + 2) * 4) + P * 16 + (1u << 20);
}
static bool wx_force_sorted() {          // A/B runs and tests (ABC_DIAG=1 ABC_WX_SORTED=1): the sorted path only
    static const bool on = abc_diag_env("ABC_WX_SORTED") != nullptr;
    return on;
}
size_t abc_wx_need(size_t nt, size_t P, size_t A) {
    const size_t b = wx_force_sorted() ? 0 : wx_binned_need(nt, P, A);
    return b ? b : wx_sorted_need(nt, P, A);       // (a binned reduction that has to be repeated on the sorted path allocates its own arena)
}

// mode 0: the counting sweep, 1: the placing sweep (blockhist: per-bin counts / offsets of the work-groups), 2: the bounds sweep
// (blockhist: the fine-bin counters)
template <int AM, int R>
static void wx_launch_bins(abc_ctx* ctx, int mode, const WxGeo& g, const double* Y, size_t ldy, size_t row_test, size_t M, size_t P,
                           size_t A, const double* model, const double* S, const WxPlan* plan, const int* segbase,
                           const unsigned long long* spl, unsigned int* blockhist, const unsigned int* binbase, unsigned long long* keys,
                           const int* v3, const unsigned int* tab) {
    const int ST = mode == 2 ? g.STb : g.ST;
    const dim3 grid = mode == 2 ? dim3((unsigned)(8 * ((ST + 7) / 8) * P), 1u) : dim3((unsigned)(8 * ((ST + 7) / 8)), (unsigned)P);
    size_t lds = (size_t)g.G * g.NB * 8 + (size_t)g.NB * 8 + 8;
    if (mode == 1) {
        lds += (size_t)WXT * R * 12 + 16;
        if (lds > (48u << 10)) (void)hipFuncSetAttribute((const void*)k_wx_bin<AM, R, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_wx_bin<AM, R, 1>), grid, dim3(WXT), lds, ctx->stream, Y, ldy, row_test, g, (int)M, (int)P, (int)A, model, S,
                           plan, segbase, spl, blockhist, binbase, keys, v3, tab);
    } else if (mode == 0) {
        if (lds > (48u << 10)) (void)hipFuncSetAttribute((const void*)k_wx_bin<AM, R, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_wx_bin<AM, R, 0>), grid, dim3(WXT), lds, ctx->stream, Y, ldy, row_test, g, (int)M, (int)P, (int)A, model, S,
                           plan, segbase, spl, blockhist, binbase, keys, v3, tab);
    } else {
        lds = (size_t)g.Gb * ((size_t)(WX_TAB + 2) * 4 + (size_t)g.NB * g.F * 4) + 64;
        if (lds > (48u << 10)) (void)hipFuncSetAttribute((const void*)k_wx_bin<AM, R, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_wx_bin<AM, R, 2>), grid, dim3(WXT), lds, ctx->stream, Y, ldy, row_test, g, (int)M, (int)P, (int)A, model, S,
                           plan, segbase, spl, blockhist, binbase, keys, v3, tab);
    }
}

// The binned path.  *fail_host = 1 when a bin outgrew LDS (the caller repeats the reduction on the sorted path).
static int launch_wilcoxon_binned(abc_ctx* ctx, const WxGeo& g, int R, const double* X, const double* Y, size_t ldx, size_t ldy, size_t M,
                                  size_t P, size_t A, size_t row_test, double* model, int* fail_host) {
    const size_t nt = (size_t)g.nt, nseg_max = (size_t)g.nseg_max, NB = (size_t)g.NB;
    WxPlan* plan = (WxPlan*)abc_ws_alloc(ctx, sizeof(WxPlan));
    int* seg_j = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    int* seg_a = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    int* astar = (int*)abc_ws_alloc(ctx, P * sizeof(int));
    int* segbase = (int*)abc_ws_alloc(ctx, P * sizeof(int));
    int* fail = (int*)abc_ws_alloc(ctx, 2 * sizeof(int));
    unsigned int* big = (unsigned int*)abc_ws_alloc(ctx, (1 + 2 * nseg_max * NB) * 4);       // bins above WX_CAP_S keys: count, (seg, bin) pairs
    unsigned long long* nz = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * 8);
    double* W = (double*)abc_ws_alloc(ctx, nseg_max * 8);
    double* S = (double*)abc_ws_alloc(ctx, nt * A * 8);
    unsigned long long* spl = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * NB * 8);
    unsigned int* hist = (unsigned int*)abc_ws_alloc(ctx, nseg_max * NB * 4);
    unsigned int* binbase = (unsigned int*)abc_ws_alloc(ctx, nseg_max * NB * 4);
    unsigned int* blockhist = (unsigned int*)abc_ws_alloc(ctx, (size_t)g.ST * nseg_max * NB * 4);
    unsigned int* blockfine = (unsigned int*)abc_ws_alloc(ctx, (size_t)g.STb * nseg_max * NB * g.F * 4 + 4);
    int* v3 = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
    unsigned int* tab = (unsigned int*)abc_ws_alloc(ctx, nseg_max * (size_t)(WX_TAB + 2) * 4);
    unsigned long long* keys = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * nt * 8);
    if (!blockfine || !v3 || !tab) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%zu segments x %zu rows)", nseg_max, nt);
    if (!plan || !seg_j || !seg_a || !astar || !segbase || !fail || !big || !nz || !W || !S || !spl || !hist || !binbase || !blockhist || !keys)
        ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%zu segments x %zu rows)", nseg_max, nt);
    hipStream_t st = ctx->stream;
#ifdef WX_STAMPS
    static unsigned long long* stamp_dev = nullptr;
    const size_t nslots = WX_STAMPS == 1 ? nseg_max * NB : (size_t)(8 * ((g.ST + 7) / 8)) * P;
    const size_t nstamp = nslots * 16;
    static size_t stamp_cap = 0;
    if (stamp_cap < nstamp) { if (stamp_dev) (void)hipFree(stamp_dev); ABC_HIP(ctx, hipMalloc((void**)&stamp_dev, nstamp * 8)); stamp_cap = nstamp; }
    ABC_HIP(ctx, hipMemsetAsync(stamp_dev, 0, nstamp * 8, st));
    ABC_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(wx_stamp_buf), &stamp_dev, sizeof(stamp_dev), 0, hipMemcpyHostToDevice, st));
#endif
    hipLaunchKernelGGL(k_wx_plan, dim3(1), dim3(256), 0, st, model, (int)M, (int)P, (int)A, plan, seg_j, seg_a, astar, (int)nseg_max, nz, W,
                       segbase, fail, v3);
    ABC_HIP(ctx, hipMemsetAsync(big, 0, 4, st));
    const unsigned rb = (unsigned)((nt + 255) / 256);
    int KC = 1;
    while (KC < (int)A) KC *= 2;
#define LAUNCH_SC(KCV) hipLaunchKernelGGL(k_wx_scores<KCV>, dim3(rb), dim3(256), 0, st, X, ldx, row_test, nt, (int)M, (int)P, (int)A, model, S)
    switch (KC) {
        case 1: LAUNCH_SC(1); break;
        case 2: LAUNCH_SC(2); break;
        case 4: LAUNCH_SC(4); break;
        case 8: LAUNCH_SC(8); break;
        case 16: LAUNCH_SC(16); break;
        default: LAUNCH_SC(32); break;
    }
#undef LAUNCH_SC
    // (the number of tests lives on the device: every grid is sized for the maximum, idle work-groups leave at once)
    if (g.NB > 1) {
#define WX_SAMPLE(EPTV)                                                                                                              \
    do {                                                                                                                             \
        if ((size_t)g.SAMP * 4 > (48u << 10))                                                                                        \
            ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_sample<EPTV>, hipFuncAttributeMaxDynamicSharedMemorySize, g.SAMP * 4)); \
        hipLaunchKernelGGL(k_wx_sample<EPTV>, dim3((unsigned)nseg_max), dim3(1024), (size_t)g.SAMP * 4, st, Y, ldy, row_test, g, (int)M, \
                           (int)P, (int)A, (const double*)model, (const double*)S, (const WxPlan*)plan, spl, tab);                 \
    } while (0)
        if (g.SAMP == 4096) WX_SAMPLE(4);
        else if (g.SAMP == 16384) WX_SAMPLE(16);
        else WX_SAMPLE(32);
#undef WX_SAMPLE
    }
#define WX_BINS(AMV, RV, MODE, BH) wx_launch_bins<AMV, RV>(ctx, MODE, g, Y, ldy, row_test, M, P, A, model, S, plan, segbase, spl, BH, binbase, keys, v3, tab)
    // the bounds sweep first: it settles every test whose statistic is not next to the threshold (v3); the counting, placing and
    // ranking launches behind it then only work on the undecided ones (their work-groups look at v3 and leave)
    bool exact = true;                      // are the exact sweeps needed?
    for (int pass = g.F ? -1 : 0; pass < 2; pass++) {
        const int mode = pass < 0 ? 2 : pass;
        unsigned int* bh = pass < 0 ? blockfine : blockhist;
        if (R == 4) WX_BINS(8, 4, mode, bh);
        else if (R == 2) WX_BINS(16, 2, mode, bh);
        else WX_BINS(32, 1, mode, bh);
        if (pass < 0) {
            if ((size_t)NB * g.F * 8 > (48u << 10))
                ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_bounds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(NB * g.F * 8)));
            hipLaunchKernelGGL(k_wx_bounds, dim3((unsigned)nseg_max), dim3(1024), (size_t)NB * g.F * 8, st, g, (const WxPlan*)plan,
                               (const unsigned int*)blockfine, nz, v3, fail + 1);
            // the reduction's host visit, here rather than at its end: with every test settled (the usual case) nothing else is
            // queued but the decision -- five launches of work-groups that would look at the verdicts and leave are 25 us
            int und = 0;
            ABC_HIP(ctx, hipMemcpyAsync(&und, fail + 1, sizeof(int), hipMemcpyDeviceToHost, st));
            ABC_HIP(ctx, hipStreamSynchronize(st));
            if (und == 0) { exact = false; break; }
        }
        if (pass == 0)
            hipLaunchKernelGGL(k_wx_offsets, dim3((unsigned)nseg_max), dim3(1024), 0, st, g, (const WxPlan*)plan, blockhist, hist, binbase, nz,
                               (const int*)v3, (const int*)segbase);
    }
#undef WX_BINS
    if (exact) {
        hipLaunchKernelGGL(k_wx_ranks, dim3((unsigned)NB, (unsigned)nseg_max), dim3(256), 0, st, g, (const WxPlan*)plan,
                           (const unsigned long long*)keys, (const unsigned int*)hist, (const unsigned int*)binbase, W, big, (const int*)v3,
                           (const int*)segbase);
        ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_ranks_big, hipFuncAttributeMaxDynamicSharedMemorySize, WX_CAP * 8));
        hipLaunchKernelGGL(k_wx_ranks_big, dim3(256), dim3(1024), (size_t)WX_CAP * 8, st, g, (const unsigned long long*)keys, (const unsigned int*)hist,
                           (const unsigned int*)binbase, W, (const unsigned int*)big, fail);
    }
    hipLaunchKernelGGL(k_wx_decide, dim3(1), dim3(1024), 0, st, model, (int)M, (int)P, (int)A, plan, nz, W, (unsigned char*)hist, (const int*)v3);
    ABC_HIP(ctx, hipGetLastError());
    *fail_host = 0;
    if (exact) {
        // did every bin of the exact sweeps fit?  (k_wx_decide has otherwise written a count from incomplete sums: the sorted path
        // overwrites it)
        ABC_HIP(ctx, hipMemcpyAsync(fail_host, fail, sizeof(int), hipMemcpyDeviceToHost, st));
        ABC_HIP(ctx, hipStreamSynchronize(st));
    }
    if (abc_diag_env("ABC_WX_DEBUG")) {          // (diagnostic: how the tests were settled)
        std::vector<int> hv(nseg_max);
        WxPlan hp;
        ABC_HIP(ctx, hipMemcpy(&hp, plan, sizeof(WxPlan), hipMemcpyDeviceToHost));
        ABC_HIP(ctx, hipMemcpy(hv.data(), v3, nseg_max * sizeof(int), hipMemcpyDeviceToHost));
        int cnt[3] = {0, 0, 0};
        for (int i = 0; i < hp.nseg; i++) cnt[hv[i] < 0 || hv[i] > 2 ? 2 : hv[i]]++;
        fprintf(stderr, "WX_DEBUG: %d tests over %zu rows, %d bins x %d fine: bounds rejected %d, passed %d, undecided %d%s\n", hp.nseg, nt, g.NB,
                g.F, cnt[0], cnt[1], cnt[2], *fail_host ? " (a bin outgrew LDS: repeat on the sorted path)" : "");
    }
#ifdef WX_STAMPS
    {
        std::vector<unsigned long long> h(nstamp);
        ABC_HIP(ctx, hipMemcpy(h.data(), stamp_dev, nstamp * 8, hipMemcpyDeviceToHost));
        double ph[6] = {0, 0, 0, 0, 0, 0}, worst = 0;
        size_t cntb = 0, worst_i = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (size_t i = 0; i < nslots; i++) {
            const unsigned long long* q = &h[i * 16];
            if (!q[0] || !q[6]) continue;
            cntb++;
            for (int p2 = 0; p2 < 6; p2++) ph[p2] += (double)(q[p2 + 1] - q[p2]);
            if ((double)(q[6] - q[0]) > worst) { worst = (double)(q[6] - q[0]); worst_i = i; }
            tmin = q[0] < tmin ? q[0] : tmin; tmax = q[6] > tmax ? q[6] : tmax;
        }
        fprintf(stderr, "WX_STAMPS kernel %d (1 k_wx_ranks: load, min/max, count, scan+place, walk, ranks; 2 / 3 k_wx_bin counting / placing: prologue, residuals of tile 0, its segments, -, later tiles, write-back): %zu work-groups, mean cycles %.0f %.0f %.0f %.0f %.0f %.0f; "
                "longest work-group %.0f cycles (n %llu, %llu); first start to last end %.0f cycles\n", (int)WX_STAMPS, cntb, ph[0] / cntb,
                ph[1] / cntb, ph[2] / cntb, ph[3] / cntb, ph[4] / cntb, ph[5] / cntb, worst, h[worst_i * 16 + 8], h[worst_i * 16 + 9],
                (double)(tmax - tmin));
    }
#endif
    return ABC_OK;
}

int launch_wilcoxon(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                    size_t P, size_t A, size_t row_test, double* model) {
    if (row_test >= n) return ABC_OK;                    // empty validation set: nothing to reduce
    if (A < 2 || P == 0) return ABC_OK;
    if (P * (A - 1) > MAXSEG)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "wilcoxon: P (A - 1) = %zu tests, more than %zu", P * (A - 1), MAXSEG);
    StageTimer tm(ctx, ST_PLS_MODEL);
    const size_t nt = n - row_test;
    WxGeo g;
    int R = 0;
    const bool binned = !wx_force_sorted() && wx_geometry(nt, P, A, &g, &R);
    if (binned) {
        // k_wx_decide needs the PRESS optima as the model fit left them: the binned pass rewrites them, so keep a copy for a repeat
        const ModelLayout ML = model_layout(M, P, A);
        int failed = 0;
        double* per_keep = (double*)abc_ws_alloc(ctx, (P + 1) * 8);
        if (!per_keep) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted");
        ABC_HIP(ctx, hipMemcpyAsync(per_keep, model + ML.off_per, P * 8, hipMemcpyDeviceToDevice, ctx->stream));
        ABC_HIP(ctx, hipMemcpyAsync(per_keep + P, model + ML.off_hdr, 8, hipMemcpyDeviceToDevice, ctx->stream));
        ABC_TRY(launch_wilcoxon_binned(ctx, g, R, X, Y, ldx, ldy, M, P, A, row_test, model, &failed));
        static const bool force_fail = abc_diag_env("ABC_WX_FORCE_FAIL") != nullptr;   // tests: exercise the repeat
        if (!failed && !force_fail) return ABC_OK;
        ABC_HIP(ctx, hipMemcpyAsync(model + ML.off_per, per_keep, P * 8, hipMemcpyDeviceToDevice, ctx->stream));
        ABC_HIP(ctx, hipMemcpyAsync(model + ML.off_hdr, per_keep + P, 8, hipMemcpyDeviceToDevice, ctx->stream));
        // the sorted path needs four (key, value) buffers over all tests: an arena of its own for the duration of the repeat
        char* const ws = ctx->ws;
        const size_t ws_bytes = ctx->ws_bytes, ws_off = ctx->ws_off;
        char* tmp = nullptr;
        const size_t need = wx_sorted_need(nt, P, A);
        ABC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (hipMalloc((void**)&tmp, need) != hipSuccess) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: %zu bytes for the sorted path", need);
        ctx->ws = tmp; ctx->ws_bytes = need; ctx->ws_off = 0;
        const int rc = launch_wilcoxon_sorted(ctx, X, Y, n, ldx, ldy, M, P, A, row_test, model);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->ws = ws; ctx->ws_bytes = ws_bytes; ctx->ws_off = ws_off;
        (void)hipFree(tmp);
        return rc;
    }
    return launch_wilcoxon_sorted(ctx, X, Y, n, ldx, ldy, M, P, A, row_test, model);
}
