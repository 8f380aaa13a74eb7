// Wilcoxon signed-rank reduction of the number of PLS components ([PLS] optimal_num_components, called at
// AbcUtil.cpp:447-449): for every response j whose PRESS optimum is a*_j > 1, the
// smallest a' < a*_j whose absolute validation errors are not significantly different (two-sided, normal
// approximation, alpha = 0.1, "average" tie ranks as lib/ranker.h:66-76, zero differences dropped) replaces it.
//
// All (j, a') tests ("segments") are batched.  Rank sums are sums of half-integers (< 2^52), hence exact and order-independent,
// so integer / double atomics stay bit-reproducible; residuals use the same fixed fma order as the oracle.
//
// Round 5 -- a CASCADE OF BOUNDS, additive over rows (so that row shards only exchange counts).  A test's statistic is
//   W = sum over the non-zero paired differences d_i of sign(d_i) rank(|d_i|),
// and ANY non-decreasing map key -> bin gives exact bounds on W from two counts per bin (all keys, positive keys): the ranks of
// a bin's keys are the integers between the keys below the bin and the keys up to its end, whatever their order (k_wx_bounds).
// Nothing is sampled and nothing is sorted:
//   level 0   every test at once, 192 cells that are LOGARITHMIC in |d| (sixteen to the binade, the leading bits of the IEEE
//             pattern: three integer operations per key), anchored at eight standard deviations of the test's paired
//             differences -- known before any row is read: |d| <= |e_a' - e_a*| and that increment is a combination of the
//             validation scores, whose second moments R' X'X R the model fit has anyway (ModelLayout::off_H).  A work-group of
//             1024 threads keeps the scores of its rows in registers and goes through ALL responses (the scores are read once
//             per group of ~190 tests, not once per response), one LDS atomic per key.  ~13 sigma of resolution: settles every
//             test whose statistic is far from the threshold -- at 1e6 particles x 16 responses x 8 components all 112 of them
//             but one;
//   level 1+  only the undecided tests, 2048 .. 16384 bins that are EQUI-DEPTH by construction: a cell of level 0 takes the
//             share of the bins its count asks for and spreads its keys over them linearly (a cell is a sixteenth of a
//             binade: the density hardly moves across it), one table read and one LDS atomic per key.  0.08 - 0.3 sigma;
//   exact     what is still undecided (statistic within that of the threshold): the keys themselves, placed into bins that are
//             unions of the last level's fine bins (their sizes and the keys below them are known from its counts), ranked
//             bin by bin in LDS (k_wx_ranks, k_wx_ranks_big: as round 4).
// Between the levels the host reads ONE word (how many tests are left; a spin on a pinned word the last work-group of the
// bounds kernel writes) and sizes the next launch.  Row-sharded sets (sharded.hip): every rank sweeps ITS validation rows, the
// counts of a level are all-reduced (T x 192 x 8 bytes, then (tests left) x bins x 8 bytes), bounds and verdicts are
// replicated; for the exact step the keys of the undecided tests are all-gathered (8 bytes x validation rows each).
// Round 4's path (a sorted sample per test for the splitters, one sweep per response, every test through ~2.5 sqrt(n) fine
// bins) is gone: 1.54 ms of its 3.1 at 1e6 x 128 metrics x 32 components were that sweep re-reading the scores 80 times from L2.
// The SORTED path of rounds 1-3 (one stable LSD radix sort of all (key, test) pairs) stays for small sets, for more than
// 32 components and as the fallback when a bin of the exact step outgrows LDS (massive ties).
#include "abc_internal.h"
#include <chrono>
#include <sched.h>
#include <time.h>
#include <vector>

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

constexpr int WX_T = 1024;                       // threads of the sweep kernels
constexpr int WX_NC0 = 192;                      // cells of level 0: 12 binades, 16 cells each
constexpr int WX_CSH = 17;                       // a cell = 2^17 key prefixes (prefix = the leading 32 of the 63 key bits: 2^21 to the binade)
constexpr int WX_LDS = 144 << 10;                // LDS of a sweep work-group's counters and tables
constexpr int WX_NBFMAX = 16384;                 // most fine bins of a level (start and span of a cell travel in 16 bits each)
constexpr int WX_FIRST_MAX = 8;                  // most responses whose tests go first (the largest count first: k_wx_plan)
constexpr int WX_NWORDS = 12;                    // ticket / list-length words of a run (levels of both halves)
constexpr unsigned long long WX_SIGN = 1ull << 63;
constexpr unsigned long long WX_MASK = ~WX_SIGN;
constexpr unsigned long long WX_NOKEY = ~0ull;   // a row without a key (zero difference, padding)

// plan of the tests and, for the cascade (kbase != nullptr), the anchor of every test's level-0 cells: cell(key) =
// min((max(prefix, kbase) - kbase) >> WX_CSH, WX_NC0 - 1) with kbase = prefix(8 sd) - (WX_NC0 - 1) cells, sd^2 = the mean
// square of e_a' - e_a* = sum_k q_jk S_ik over the components k between a' and a*, taken from the diagonal of H (the validation
// scores of different components are close to orthogonal; the anchor only decides how evenly the cells are filled)
__global__ __launch_bounds__(256) void k_wx_plan(const double* __restrict__ model, int M, int P, int A, WxPlan* __restrict__ plan,
                                                 int* __restrict__ seg_j, int* __restrict__ seg_a, int* __restrict__ astar, int nseg_max,
                                                 unsigned long long* __restrict__ nz, double* __restrict__ W, int* __restrict__ segbase,
                                                 int* __restrict__ fail, int* __restrict__ v3, unsigned int* __restrict__ kbase,
                                                 int* __restrict__ act, int* __restrict__ nact, unsigned int* __restrict__ tickets,
                                                 double nv_total, double* __restrict__ per_keep = nullptr, int first_r = 0,
                                                 int* __restrict__ act_rest = nullptr, unsigned char* __restrict__ in_first = nullptr) {
    __shared__ int s_ns;
    __shared__ double s_score[1024];
    __shared__ int s_pick[WX_FIRST_MAX];
    __shared__ int s_npick, s_fitmax;
    __shared__ int s_w[16];
    __shared__ int s_base;
    if (per_keep) {           // the PRESS optima and the header as the fit left them (k_wx_decide rewrites them; a repeat starts from these)
        const ModelLayout MLk = model_layout(M, P, A);
        for (int j = threadIdx.x; j < P; j += blockDim.x) per_keep[j] = model[MLk.off_per + j];
        if (threadIdx.x == 0) per_keep[P] = model[MLk.off_hdr];
    }
    for (int s = threadIdx.x; s < nseg_max; s += blockDim.x) { nz[s] = 0; W[s] = 0.0; if (v3) v3[s] = 2; if (act) act[s] = s; }
    if (in_first) for (int j = threadIdx.x; j < P; j += blockDim.x) in_first[j] = 0;
    const ModelLayout ML = model_layout(M, P, A);
    if (threadIdx.x == 0) {
        if (fail) { fail[0] = 0; fail[1] = 0; }
        if (tickets) for (int i = 0; i < WX_NWORDS; i++) tickets[i] = 0;
        s_base = 0;
        s_npick = 0;
        s_fitmax = 1;
    }
    __syncthreads();
    // a response per thread: its tests are the segments segbase[j] .. segbase[j] + a*_j - 2 (a scan of the counts over the work-group;
    // one thread writing all of them in turn was 14 of this kernel's 22 us at 486 tests)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    for (int j0 = 0; j0 < P; j0 += (int)blockDim.x) {
        const int j = j0 + (int)threadIdx.x;
        const int as = j < P ? (int)model[ML.off_per + j] : 1;
        const int cnt = as > 1 ? as - 1 : 0;
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int up = __shfl_up(inc, d, 64); if (lane >= d) inc += up; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        int base = s_base + inc - cnt;
        for (int w = 0; w < wave; w++) base += s_w[w];
        if (j < P) {
            astar[j] = as;
            if (segbase) segbase[j] = base < nseg_max ? base : nseg_max;
            for (int a = 1; a < as; a++)
                if (base + a - 1 < nseg_max) { seg_j[base + a - 1] = j; seg_a[base + a - 1] = a; }
        }
        __syncthreads();
        if (threadIdx.x == 0) { int tot = s_base; for (int w = 0; w < nw; w++) tot += s_w[w]; s_base = tot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int ns = s_base < nseg_max ? s_base : nseg_max;
        plan->nseg = ns;
        plan->pad_ = 0;
        plan->seg_j = seg_j;
        plan->seg_a = seg_a;
        plan->astar = astar;
        if (nact) nact[0] = ns;
        s_ns = ns;
    }
    __syncthreads();
    if (!kbase) return;
    const int ns = s_ns;
    // THE LARGEST COUNT FIRST (round 6; first_r > 0: a caller that only uses the largest per-response count, AbcUtil.cpp:449).  That
    // count stays the fit's as soon as ONE response that holds it keeps it, i.e. has all its tests rejected -- so the cascade starts
    // with the tests of the first_r responses most likely to: among the responses whose PRESS optimum is the largest, those whose
    // closest competitor lies furthest above the optimum in relative PRESS (a heuristic that only orders the work: whatever it
    // picks, the verdicts are the tests' own).  act = their tests (response order), act_rest = all the others (test order),
    // in_first[j] = 1 for the picked responses; nact[0] / nact[4] = the two lengths.  Nothing to pick from (one component, more
    // than 1024 responses, no more holders than picks would leave anything over): act = every test, as before.
    if (first_r > 0 && act_rest && in_first && P <= 1024 && ns > 0) {
        const int t = threadIdx.x;
        for (int j = t; j < P; j += blockDim.x) atomicMax(&s_fitmax, astar[j]);
        __syncthreads();
        const int fitmax = s_fitmax;
        for (int j = t; j < P; j += blockDim.x) {
            double sc = -1.0;
            const int as = astar[j];
            if (as == fitmax && as > 1) {
                const double best = model[ML.off_press + (as - 1) + (size_t)A * j];
                double dmin = 1e300;
                for (int a = 1; a < as; a++) { const double d = model[ML.off_press + (a - 1) + (size_t)A * j] - best; dmin = d < dmin ? d : dmin; }
                sc = best > 0.0 ? dmin / best : (dmin > 0.0 ? 1e300 : 0.0);
                if (!(sc >= 0.0)) sc = 0.0;                      // (NaN, a negative difference: still a holder, picked last)
            }
            s_score[j] = sc;
        }
        __syncthreads();
        if (t < 64) {                                            // wave 0: first_r rounds of (largest score, lowest response)
            const int want = first_r < WX_FIRST_MAX ? first_r : WX_FIRST_MAX;
            for (int r = 0; r < want; r++) {
                double bs = -1.0;
                int bj = 1 << 30;
                for (int j = t; j < P; j += 64) { const double sc = s_score[j]; if (sc > bs) { bs = sc; bj = j; } }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const double os = __shfl_xor(bs, o, 64);
                    const int oj = __shfl_xor(bj, o, 64);
                    if (os > bs || (os == bs && oj < bj)) { bs = os; bj = oj; }
                }
                if (bs < 0.0) break;                             // (uniform: no holder left)
                if (t == 0) { s_pick[s_npick++] = bj; s_score[bj] = -2.0; }
                __builtin_amdgcn_wave_barrier();                 // (one wave: its LDS accesses execute in program order)
            }
        }
        __syncthreads();
        const int np = s_npick;
        if (np > 0 && np < P) {
            for (int i = t; i < np; i += blockDim.x) in_first[s_pick[i]] = 1;
            __syncthreads();
            for (int j = t; j < P; j += blockDim.x) {
                const int as = astar[j], cnt = as > 1 ? as - 1 : 0, b0 = segbase[j];
                int pa = 0, mine = 0;                            // picked tests in front of response j
                for (int i = 0; i < np; i++) { const int pj = s_pick[i]; if (pj < j) pa += astar[pj] - 1; if (pj == j) mine = 1; }
                for (int a = 0; a < cnt; a++)
                    if (b0 + a < ns) { if (mine) act[pa + a] = b0 + a; else act_rest[b0 - pa + a] = b0 + a; }
            }
            if (t == 0) {
                int na = 0;
                for (int i = 0; i < np; i++) na += astar[s_pick[i]] - 1;
                nact[0] = na;
                nact[4] = ns - na;
            }
        } else if (t == 0)
            nact[4] = 0;
    } else if (nact && threadIdx.x == 0 && act_rest)
        nact[4] = 0;
    __syncthreads();
    for (int s = threadIdx.x; s < ns; s += blockDim.x) {
        const int j = seg_j[s], a1 = seg_a[s], as = astar[j];
        double var = 0.0;
        for (int k = a1; k < as; k++) {
            const double q = model[ML.off_Q + j + (size_t)P * k];
            var = fma(q * q, model[ML.off_H + k + (size_t)A * k], var);
        }
        var /= nv_total;
        if (!(var > 0.0) || !(var < 1e300)) var = model[ML.off_press + (a1 - 1) + (size_t)A * j] / nv_total;      // (degenerate scores)
        if (!(var > 0.0) || !(var < 1e300)) var = 1.0;
        const double top = 8.0 * sqrt(var);
        const unsigned int ktop = (unsigned int)((unsigned long long)__double_as_longlong(top) >> 31);
        const unsigned int span = (unsigned int)(WX_NC0 - 1) << WX_CSH;
        kbase[s] = ktop > span ? ktop - span : 0u;
    }
}

// scores of the validation rows: S[i + nt*k] = sum_m z(x_im) R[m,k]  (m ascending fma chain, as the oracle)
template <int KC>
__global__ __launch_bounds__(256) void k_wx_scores(const double* __restrict__ X, size_t ldx, size_t row_test, size_t nt,
                                                   int M, int P, int A, const double* __restrict__ model,
                                                   double* __restrict__ S, size_t sld /* rows of S (>= nt) */) {
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
        if (k < A) S[i + sld * k] = s[k];
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
// the cascade of bounds
// ===========================================================================================================================
// (the phase stamps of round 4's diagnostic build of k_wx_ranks: no-ops)
#define WX_STAMP(i) do { } while (0)

// cell of level 0: logarithmic in |d|, a non-decreasing function of the key prefix
__device__ __forceinline__ unsigned int wx_cell(unsigned int k32, unsigned int kb) {
    const unsigned int off = (k32 > kb ? k32 : kb) - kb, c = off >> WX_CSH;
    return c < (unsigned int)(WX_NC0 - 1) ? c : (unsigned int)(WX_NC0 - 1);
}
// fine bin of a level: cell c owns the bins [start_c, start_c + span_c) (tab[c] = start | span << 16) and spreads its prefixes
// linearly over them; non-decreasing in the prefix (starts are non-decreasing, a cell's last bin lies below the next start)
__device__ __forceinline__ unsigned int wx_fine(unsigned int k32, unsigned int kb, const unsigned int* __restrict__ tab) {
    const unsigned int off = (k32 > kb ? k32 : kb) - kb;
    unsigned int c = off >> WX_CSH;
    c = c < (unsigned int)(WX_NC0 - 1) ? c : (unsigned int)(WX_NC0 - 1);
    unsigned int r = off - (c << WX_CSH);
    r = r < (1u << WX_CSH) - 1u ? r : (1u << WX_CSH) - 1u;                // (the last cell takes everything above)
    const unsigned int e = tab[c];
    return (e & 0xffffu) + (__umul24(r, e >> 16) >> WX_CSH);              // r < 2^17, span <= 2^14: the product fits
}
// the table of a level with NBX fine bins from a test's level-0 counts, by ONE wave: cell c starts at bin floor(keys below c * NBX / m)
__device__ __forceinline__ void wx_wave_table(const unsigned int* __restrict__ c0row, int NBX, unsigned int* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    static_assert(WX_NC0 == 3 * 64, "three cells per lane");
    unsigned int c[3], loc = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) { c[i] = c0row[3 * lane + i]; loc += c[i]; }
    unsigned int inc = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned int up = __shfl_up(inc, d, 64); if (lane >= d) inc += up; }
    const unsigned int m = __shfl(inc, 63, 64);
    unsigned int cum = inc - loc;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int cell = 3 * lane + i;
        unsigned int s0 = 0, s1 = 0;
        if (m) {
            s0 = (unsigned int)(((unsigned long long)cum * (unsigned int)NBX) / m);
            s1 = (cell == WX_NC0 - 1) ? (unsigned int)NBX : (unsigned int)(((unsigned long long)(cum + c[i]) * (unsigned int)NBX) / m);
            if (s0 >= (unsigned int)NBX) { s0 = (unsigned int)NBX - 1u; s1 = s0; }
        }
        out[cell] = s0 | ((s1 - s0) << 16);
        cum += c[i];
    }
}

// The sweep over the validation rows.  A work-group = (a run of tiles of 1024 R rows, a group of <= G tests of the list `act`): a
// thread keeps the scores of its R rows in registers for the whole tile and goes through the group's tests response by response
// -- per (row, response) the prediction chain once for |e_a*| and once more for the |e_a'| of the group's tests of that response
// (pred: the k-ascending fma chain of the oracle), per (row, test) the key and
//   MODE 0: one LDS atomic on the test's level-0 cell        MODE 1: a table read and one LDS atomic on its fine bin
//   (all keys in the low, positive differences in the high half of a 32-bit counter: a work-group's rows stay below 2^16)
//   MODE 2: the key itself (sign of d in bit 63) to keys[slot][row] -- the exact step.
// The work-groups of one run of tiles (one per group of tests) follow each other on ONE XCD (blockIdx % 8), so the scores come
// from HBM once and from that XCD's L2 for the other groups.  TT threads: 1024 (four waves per SIMD, <= 128 registers), or 768 with two
// rows per thread at 17..32 components (three waves per SIMD, <= 170 registers: the 2 x 32 scores of a thread's rows alone are 128).
template <int AM, int R, int MODE, int TT>
__global__ __launch_bounds__(TT) void k_wx_sweep(const double* __restrict__ Y, size_t ldy, size_t row_test, size_t nt, int M, int P, int A,
                                                   const double* __restrict__ model, const double* __restrict__ S, size_t sld,
                                                   const int* __restrict__ seg_j, const int* __restrict__ seg_a, const int* __restrict__ astar,
                                                   const int* __restrict__ act, const int* __restrict__ nact_p, int act_lo, int act_n, int G,
                                                   int TG, int RR, int tpw, const unsigned int* __restrict__ kbase,
                                                   const unsigned int* __restrict__ c0, int NBX, unsigned int* __restrict__ blockcnt,
                                                   unsigned long long* __restrict__ keys, size_t kld) {
    constexpr int TR = TT * R;
    extern __shared__ unsigned int wx_lds[];
    const int q = (int)(blockIdx.x >> 3), rr = (q / TG) * 8 + (int)(blockIdx.x & 7), tg = q % TG;
    if (rr >= RR) return;
    int end = act_lo + act_n;
    { const int na = *nact_p; end = end < na ? end : na; }
    const int lo = act_lo + tg * G;
    if (lo >= end) return;
    const int ng = (end - lo < G) ? end - lo : G, t = threadIdx.x;
    unsigned int* cnt = wx_lds;                                           // [G][NBX] + spare counters (NBX / 64)  (MODE 0 / 1)
    unsigned int* tab_s = cnt + (MODE == 2 ? 0 : (size_t)G * NBX + (MODE == 0 ? NBX : 64));  // [G][WX_NC0]  (MODE 1)
    int* kb_s = (int*)(tab_s + (MODE == 1 ? (size_t)G * WX_NC0 : 0));     // [G] anchors
    int* sj = kb_s + G;                                                   // [G] response, [G] candidate of the group's tests
    int* sa = sj + G;
    // entries: one per response present in the group -- its index, the mask of its candidates a' (bit a' - 1), the slot of its first
    // test, its optimum a*, and (two words each) the mean and the deviation of the response
    int* ej = sa + G;
    int* em = ej + G;
    int* es = em + G;
    int* eas = es + G;
    double* emu = (double*)(((size_t)(eas + G) + 7) & ~(size_t)7);
    double* esd = emu + G;
    __shared__ int s_nent;
    __shared__ int s_wcnt[TT / 64];
    const ModelLayout ML = model_layout(M, P, A);
    if (MODE != 2) for (int e = t; e < ng * NBX; e += TT) cnt[e] = 0u;
    if (t < ng) { const int s = act[lo + t]; kb_s[t] = (int)kbase[s]; sj[t] = seg_j[s]; sa[t] = seg_a[s]; }
    if (MODE == 1)
        for (int slot = t >> 6; slot < ng; slot += TT / 64) wx_wave_table(c0 + (size_t)act[lo + slot] * WX_NC0, NBX, tab_s + (size_t)slot * WX_NC0);
    __syncthreads();
    {   // the entries, in parallel: the thread of a response's FIRST test in the group collects the response's mask (<= 31 steps);
        // its entry index = the starts in front of it (ballots; thread 0 walking the list alone was 7 us of a 60 us work-group)
        const bool start = t < ng && (t == 0 || sj[t - 1] != sj[t]);
        const unsigned long long bal = __ballot(start);
        if ((t & 63) == 0) s_wcnt[t >> 6] = __popcll(bal);
        __syncthreads();
        if (start) {
            int idx = __popcll(bal & ((1ull << (t & 63)) - 1ull));
            for (int w = 0; w < (t >> 6); w++) idx += s_wcnt[w];
            const int j = sj[t];
            unsigned int mask = 0;
            for (int e2 = t; e2 < ng && sj[e2] == j; e2++) mask |= 1u << (sa[e2] - 1);
            ej[idx] = j; em[idx] = (int)mask; es[idx] = t; eas[idx] = astar[j];
            emu[idx] = model[ML.off_mean + M + j];
            esd[idx] = model[ML.off_sd + M + j];
        }
        if (t == 0) { int ne = 0; for (int w = 0; w < TT / 64; w++) ne += s_wcnt[w]; s_nent = ne; }
    }
    __syncthreads();
    const int nent = s_nent;
    const int lane = t & 63;
    const size_t nrows = MODE == 2 ? kld : nt;                            // (MODE 2 also writes the padding rows nt .. kld - 1)
    for (int tt = 0; tt < tpw; tt++) {
        const size_t row_t = ((size_t)rr * tpw + tt) * TR;
        if (row_t >= nrows) break;
        double s[R][AM];
        size_t ic[R];
        bool in[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const size_t i = row_t + (size_t)r * TT + t;
            in[r] = i < nt;
            ic[r] = in[r] ? i : (nt ? nt - 1 : 0);
#pragma unroll
            for (int k = 0; k < AM; k++) s[r][k] = in[r] ? S[ic[r] + sld * (size_t)(k < A ? k : A - 1)] : 0.0;     // (rows past the end: zeros)
        }
        // Everything a response needs -- its y, its loadings (lane k holds q_jk), its parameters, the anchors of its tests (lane k: the
        // test a' = k + 1) -- is fetched while the arithmetic of the response BEFORE it runs; a step's operands then come out of the
        // lanes (v_readlane).  As scalar loads / LDS reads inside the steps every one of them was waited for on the spot.
        struct Ent { int j, mask, slot0, as; double mu, sd, q, y[R]; unsigned int kbv; };
        auto fetch = [&](int e) -> Ent {
            Ent n;
            n.j = ej[e]; n.mask = em[e]; n.slot0 = es[e]; n.as = eas[e]; n.mu = emu[e]; n.sd = esd[e];
            const int j = __builtin_amdgcn_readfirstlane(n.j);
#pragma unroll
            for (int r = 0; r < R; r++) n.y[r] = nt ? Y[row_test + ic[r] + ldy * (size_t)j] : 0.0;
            n.q = model[ML.off_Q + j + (size_t)P * (lane < A ? lane : A - 1)];
            const unsigned int mask = (unsigned int)__builtin_amdgcn_readfirstlane(n.mask);
            const unsigned int below = mask & ((1u << (lane & 31)) - 1u);
            n.kbv = (lane < 32 && ((mask >> lane) & 1u)) ? (unsigned int)kb_s[__builtin_amdgcn_readfirstlane(n.slot0) + __popc(below)] : 0u;
            return n;
        };
        Ent nx = fetch(0);
        for (int ent = 0; ent < nent; ent++) {
            const Ent cu = nx;
            if (ent + 1 < nent) nx = fetch(ent + 1);
            const unsigned int mask = (unsigned int)__builtin_amdgcn_readfirstlane(cu.mask);
            const int slot0 = __builtin_amdgcn_readfirstlane(cu.slot0);
            const int as = __builtin_amdgcn_readfirstlane(cu.as);
            const double muy = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(cu.mu)), __builtin_amdgcn_readfirstlane(__double2loint(cu.mu)));
            const double sdy = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(cu.sd)), __builtin_amdgcn_readfirstlane(__double2loint(cu.sd)));
            const unsigned int kbv = cu.kbv;
            double zy[R], estar[R], pred[R];
#pragma unroll
            for (int r = 0; r < R; r++) {
                zy[r] = (sdy == 0.0 || !in[r]) ? 0.0 : (cu.y[r] - muy) / sdy;
                pred[r] = 0.0;
            }
            const int qlo = __double2loint(cu.q), qhi = __double2hiint(cu.q);
#pragma unroll
            for (int k = 0; k < AM; k++)
                if (k < as) {
                    const double qk = __hiloint2double(__builtin_amdgcn_readlane(qhi, k), __builtin_amdgcn_readlane(qlo, k));
#pragma unroll
                    for (int r = 0; r < R; r++) pred[r] = fma(s[r][k], qk, pred[r]);
                }
#pragma unroll
            for (int r = 0; r < R; r++) { estar[r] = fabs(zy[r] - pred[r]); pred[r] = 0.0; }
            int slot = slot0;
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < AM - 1; k++)
                    if (k + 1 < as) {                                     // (uniform)
                        const double qk = __hiloint2double(__builtin_amdgcn_readlane(qhi, k), __builtin_amdgcn_readlane(qlo, k));
#pragma unroll
                        for (int r = 0; r < R; r++) pred[r] = fma(s[r][k], qk, pred[r]);
                        if ((mask >> k) & 1u) {                           // (uniform) the test (j, a' = k + 1) is in this group
#pragma unroll
                            for (int r = 0; r < R; r++) {
                                const double d = estar[r] - fabs(zy[r] - pred[r]);
                                const unsigned long long k63 = (unsigned long long)__double_as_longlong(fabs(d));
                                const size_t i = row_t + (size_t)r * TT + t;
                                if (i < kld) keys[(size_t)(lo - act_lo + slot) * kld + i] = (in[r] && d != 0.0) ? (k63 | (d > 0.0 ? WX_SIGN : 0ull)) : WX_NOKEY;
                            }
                            slot++;
                        }
                    }
            } else {
                // Four candidates at a time without a uniform branch between them: a step that is not a test of this group (beyond a*,
                // not in the list) still runs -- its keys go to a spare set of counters behind the group's.  Rows past the end carry
                // zeros (d = 0: no key).  A block without any test only extends the prediction chain.  (The atomic stays under
                // `d != 0`: adding 0 for the keyless lanes instead ran the kernel three times slower, 64 -> 193 us at 1e6 x 32 x 16 x 8
                // with the same instruction and LDS-conflict counts -- measured, not understood.)
                const unsigned int vmask = (as >= 2) ? (mask & (0xffffffffu >> (33 - as))) : 0u;        // tests a' = k + 1 < a*
#pragma unroll
                for (int k0 = 0; k0 < AM - 1; k0 += 4) {
                    if (k0 + 1 >= as) continue;                           // (uniform) nothing of this response from here on
                    if (((vmask >> k0) & 0xfu) != 0u) {                   // (uniform)
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int k = k0 + kk;
                            if (k >= AM - 1) continue;                    // (compile time)
                            const double qk = __hiloint2double(__builtin_amdgcn_readlane(qhi, k), __builtin_amdgcn_readlane(qlo, k));
                            const bool valid = ((vmask >> k) & 1u) != 0u; // (uniform)
                            const unsigned int kb = (unsigned int)__builtin_amdgcn_readlane((int)kbv, k);
                            const int sl = valid ? slot : G;              // (slot G: the spare counters -- NBX of them at level 0, 64 at the
                            unsigned int* cn = cnt + (size_t)sl * NBX;    //  fine levels, where the bin is cut to six bits; its table: any)
                            const unsigned int* tb = tab_s + (size_t)(valid ? slot : 0) * WX_NC0;
                            const unsigned int bmask = (MODE == 0 || valid) ? 0xffffffffu : 63u;
#pragma unroll
                            for (int r = 0; r < R; r++) {
                                pred[r] = fma(s[r][k], qk, pred[r]);
                                const double d = estar[r] - fabs(zy[r] - pred[r]);
                                const unsigned int k32 = (unsigned int)((unsigned long long)__double_as_longlong(fabs(d)) >> 31);
                                const unsigned int bin = MODE == 0 ? wx_cell(k32, kb) : (wx_fine(k32, kb, tb) & bmask);
                                if (d != 0.0) atomicAdd(&cn[bin], d > 0.0 ? 65537u : 1u);
                            }
                            slot += valid ? 1 : 0;
                        }
                    } else {
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const int k = k0 + kk;
                            if (k >= AM - 1) continue;
                            const double qk = __hiloint2double(__builtin_amdgcn_readlane(qhi, k), __builtin_amdgcn_readlane(qlo, k));
#pragma unroll
                            for (int r = 0; r < R; r++) pred[r] = fma(s[r][k], qk, pred[r]);
                        }
                    }
                }
            }
        }
    }
    if (MODE != 2) {
        __syncthreads();
        unsigned int* dst = blockcnt + ((size_t)rr * act_n + (size_t)(lo - act_lo)) * NBX;
        for (int e = t; e < ng * NBX; e += TT) dst[e] = cnt[e];
    }
}

// the counters of a level's sweep work-groups added up: totals[slot][bin] = all keys | positive keys << 32 (the halves add up
// separately -- also over the ranks of a row-sharded set, whose all-reduce takes these words -- while the set has fewer than 2^32
// validation rows).  A work-group = 32 counters x 8 slices of the runs of tiles, the slices' sums combined through LDS: every
// thread has RR / 8 independent loads (one work-group per test walking all RR runs was 440 us of latency at 112 tests x 488 runs)
__global__ __launch_bounds__(256) void k_wx_totals(int NBX, int RR, int act_n, const unsigned int* __restrict__ blockcnt,
                                                   unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long part[8][32];
    const size_t ne = (size_t)act_n * NBX, e = (size_t)blockIdx.x * 32 + (threadIdx.x & 31);
    const int sl = threadIdx.x >> 5;
    unsigned long long c = 0, p = 0;
    if (e < ne) {
        int rr = sl;
        for (; rr + 56 < RR; rr += 64) {
            unsigned int v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = blockcnt[(size_t)(rr + 8 * u) * ne + e];
#pragma unroll
            for (int u = 0; u < 8; u++) { c += v[u] & 0xffffu; p += v[u] >> 16; }
        }
        for (; rr < RR; rr += 8) { const unsigned int v = blockcnt[(size_t)rr * ne + e]; c += v & 0xffffu; p += v >> 16; }
    }
    part[sl][threadIdx.x & 31] = c | (p << 32);
    __syncthreads();
    if (sl == 0 && e < ne) {
        unsigned long long tsum = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) tsum += part[u][threadIdx.x];
        totals[e] = tsum;
    }
}

// does test `seg` (of a response whose tests start at seg_first) still need a closer look?  v3: 0 rejected / 1 passed by the
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

// BOUNDS on the signed rank sum from counts alone.  Bin b of a test holds c_b keys, p_b of them positive differences, B_b keys
// lie below it: whatever the order inside the bin, its ranks are B_b + 1 .. B_b + c_b (average ranks of ties: a doubly stochastic
// mix of those, which moves no subset sum beyond the extremes), so the positives' rank sum lies between the p_b lowest and the p_b
// highest of them and   2 W_b in [4 p B + 2 p (p + 1), 4 p B + 4 p c - 2 p (p - 1)] - (2 c B + c (c + 1)).
// Integers, summed exactly.  The interval of |W| / sigma, widened by 1e-12 against the roundings of the division, is put
// through the decision function of k_wx_decide at both ends: equal answers = THE answer (the function is monotone but for the last
// bits next to its threshold); else the test stays undecided (2).  With B bins of equal depth the interval is ~0.87 sqrt(m) / B
// sigma wide.
// One work-group per slot of the list; counts from totals (k_wx_totals; all-reduced over the ranks of a row-sharded set).  cl: the
// all-keys counts of the slot (the next level's table / the exact step's bins are made from them).  The LAST work-group of a level
// to finish (ticket) compacts the tests that are still needed into act_next, in test order, and tells the host how many.
__global__ __launch_bounds__(1024) void k_wx_bounds(int NBX, int act_n, const int* __restrict__ act, const int* __restrict__ nact_p,
                                                    int act_lo, const unsigned long long* __restrict__ totals, unsigned long long* __restrict__ nz,
                                                    int* __restrict__ v3, unsigned int* __restrict__ cl, size_t cl_ld, int cl_by_test,
                                                    int* __restrict__ slotmap, unsigned int* __restrict__ ticket, unsigned int nblocks_level,
                                                    const int* __restrict__ astar, int P, const int* __restrict__ segbase,
                                                    int* __restrict__ act_next, int* __restrict__ nact_next,
                                                    const double* __restrict__ nv_ranks, size_t nv_stride, int Wr, int* __restrict__ pin_words,
                                                    int force_undecided /* diagnostic: every test with keys stays undecided */,
                                                    double* __restrict__ model, int M, int A, double* __restrict__ dec,
                                                    const double* __restrict__ per_keep, int stop_at_max,
                                                    const unsigned char* __restrict__ in_first /* the first half: the picked responses only */,
                                                    const int* __restrict__ nrest_p /* ... and the number of tests outside it */) {
    extern __shared__ unsigned int wxb_cp[];              // [NBX] packed (all keys, positive keys) of the test's bins
    __shared__ long long red[3][16];
    __shared__ unsigned long long wtot[16];
    __shared__ int s_last;
    const int e = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int end = act_lo + act_n;
    { const int na = *nact_p; end = end < na ? end : na; }
    if (act_lo + e < end) {
        const int seg = act[act_lo + e];
        // (all, positive) of bin b as one 8-byte word at cp[b + b / per]: a thread owns `per` consecutive bins, and the pad word per
        // thread keeps the lanes of a wave on different banks (unpadded, lane t read word 2 per t: all on ONE bank at per = 16 --
        // 1.5 us per turn of the loops below, 25 of this kernel's 40 us at 16384 bins)
        unsigned long long* cp = (unsigned long long*)wxb_cp;
        const int per = (NBX + 1023) / 1024, b0t = t * per;
        for (int b0 = 0; b0 < NBX; b0 += 8 * 1024) {           // (eight loads in flight: one per turn was 16 memory latencies at 16384 bins)
            unsigned long long v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int b = b0 + u * 1024 + t; v[u] = b < NBX ? totals[(size_t)e * NBX + b] : 0ull; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int b = b0 + u * 1024 + t;
                if (b < NBX) {
                    cp[b + b / per] = v[u];
                    if (cl) cl[(size_t)(cl_by_test ? seg : act_lo + e) * cl_ld + b] = (unsigned int)v[u];
                }
            }
        }
        if (t == 0 && slotmap) slotmap[seg] = act_lo + e;
        __syncthreads();
        // thread t owns the consecutive bins [t per, (t + 1) per): keys below them by a work-group scan of the threads' totals
        unsigned long long loc = 0;
        for (int i = 0; i < per; i++) if (b0t + i < NBX) loc += (unsigned int)cp[b0t + i + t];
        unsigned long long inc = loc;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long u = (unsigned long long)__shfl_up((long long)inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        unsigned long long below = inc - loc;
        for (int w = 0; w < wave; w++) below += wtot[w];
        long long lo2 = 0, hi2 = 0, m = (long long)loc;
        for (int i = 0; i < per; i++)
            if (b0t + i < NBX) {
                const unsigned long long w = cp[b0t + i + t];
                const long long c = (long long)(unsigned int)w, p = (long long)(w >> 32), B = (long long)below;
                const long long all2 = 2 * c * B + c * (c + 1);
                lo2 += 4 * p * B + 2 * p * (p + 1) - all2;
                hi2 += 4 * p * B + 4 * p * c - 2 * p * (p - 1) - all2;
                below += (unsigned long long)c;
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
                v = (p_lo == p_hi && !force_undecided) ? (p_lo ? 1 : 0) : 2;
            }
            v3[seg] = v;
        }
    }
    // ---- the level's last work-group: the tests still needed, in test order ------------------------------------------------
    __syncthreads();
    if (t == 0) {
        __threadfence();
        s_last = (atomicAdd(ticket, 1u) == nblocks_level - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // a response per thread: its verdicts (<= 31 independent loads), the tests still needed = the undecided ones in front of its
    // first pass; their places by a scan over the responses (test order = response order).  (A thread per test that walked back to
    // the response's first test was up to 31 dependent loads in a row: 15-20 us of every level.)
    // Beside them, per response, the RANGE its final count can still take: the first passing candidate in front of any undecided one
    // is the count; an undecided candidate u in front of every pass leaves [u, first pass behind it (or a*)].  The caller of the
    // fused generations only uses the LARGEST count over the responses (AbcUtil.cpp:449: `.maxCoeff()`), so with stop_at_max the
    // cascade ends as soon as the ranges' lower and upper ends have the same maximum -- at 1e6 x 32 x 16 x 8 after level 0, with one
    // test of 112 still open: its response cannot exceed the 8 components fifteen others keep for certain.
    __shared__ int wsum[16];
    __shared__ int s_run, s_lo, s_hi;
    if (t == 0) { s_run = 0; s_lo = 1; s_hi = 1; }
    __syncthreads();
    const volatile int* v3v = v3;
    const ModelLayout MLd = model_layout(M, P, A);
    double* per_out = model ? (dec ? dec : model + MLd.off_per) : nullptr;
    for (int j0 = 0; j0 < P; j0 += 1024) {
        const int j = j0 + t;
        unsigned int need = 0;
        int b0 = 0, my_hi = 0;
        if (j < P) {
            b0 = segbase[j];
            const int as = astar[j], n = as - 1;
            unsigned int und = 0, pas = 0;
#pragma unroll 8
            for (int i = 0; i < n; i++) { const int v = v3v[b0 + i]; und |= (v == 2 ? 1u : 0u) << i; pas |= (v == 1 ? 1u : 0u) << i; }
            need = pas ? (und & ((pas & (0u - pas)) - 1u)) : und;
            const int fp = pas ? __ffs((int)pas) : n + 1, fu = need ? __ffs((int)need) : n + 1;      // 1-based candidates; n + 1: none (the count stays a*)
            const int hi = fp, lo = fu < fp ? fu : fp;
            // (the first half -- the largest count first, k_wx_plan: only a picked response that can still keep its optimum has tests
            // worth another level; the others' ranges enter the decision as they are, untested = [1, a*])
            if (in_first && (!in_first[j] || pas)) need = 0;
            atomicMax(&s_lo, lo);
            atomicMax(&s_hi, hi);
            if (per_out) per_out[j] = (double)hi;            // (exact once nothing of the response is open; an upper end until then)
            my_hi = hi;
        }
        // (round 6) a caller that only uses the LARGEST count does not need the open tests of a response whose count can no longer
        // exceed what another response keeps for certain (hi_j <= max_j lo_j): with noisy responses most responses have a passing
        // candidate low down, and their open tests in front of it were most of the next level's work.  (One response per thread:
        // up to 1024 responses, where the maximum is known here; beyond, every response keeps its tests.)
        // Their verdict becomes 3 = "left open, not needed": neither passed nor open to the levels after this one and to k_wx_decide,
        // so the response's count stays the upper end its first pass gives (as every count a stop_at_max run leaves behind).
        if (stop_at_max && P <= 1024) {
            __syncthreads();
            if (j < P && my_hi <= s_lo) {
                while (need) { const int i = __ffs((int)need) - 1; need &= need - 1u; v3[b0 + i] = 3; }
            }
        }
        const int mine = __popc(need);
        int inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int pos = s_run + inc - mine;
        for (int w = 0; w < wave; w++) pos += wsum[w];
        while (need) { const int i = __ffs((int)need) - 1; need &= need - 1u; act_next[pos++] = b0 + i; }
        __syncthreads();
        if (t == 0) { int tot = s_run; for (int w = 0; w < 16; w++) tot += wsum[w]; s_run = tot; }
        __syncthreads();
    }
    // Decided -- nothing left, or (stop_at_max) the largest count is certain: the counts are written here (k_wx_decide's rule: per
    // response the first candidate that passes, else the PRESS optimum; the largest over the responses) and the host, which is told
    // `0 left`, neither queues another level nor launches k_wx_decide (its launch behind the host's look was 20 us of the critical path).
    // dec != NULL -- the fused generation's speculative run, api.hip: the counts go to dec[0 .. P - 1], the largest to dec[P], and the
    // model record keeps what the fit wrote; pin_words[3] / [4]: the final count, and whether it differs from the fit's
    // (first half: nothing left to look at is NOT a decision -- the picked responses were all reduced or stay within the last level's
    // resolution of the threshold: the host goes on with the other responses' tests)
    const bool decided = in_first ? (s_lo == s_hi) : (s_run == 0 || (stop_at_max && s_lo == s_hi));
    const int total = decided ? 0 : s_run;
    if (decided && model && t == 0) {
        const int nc = s_hi;
        if (dec) dec[P] = (double)nc; else model[MLd.off_hdr] = (double)nc;
        __hip_atomic_store(&pin_words[3], nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pin_words[4], (per_keep && nc != (int)per_keep[P]) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (t == 0) {
        *nact_next = total;
        long long vmax = 0;
        if (nv_ranks) for (int r = 0; r < Wr; r++) { const long long v = (long long)nv_ranks[(size_t)r * nv_stride]; vmax = v > vmax ? v : vmax; }
        // (pinned, fine-grained host memory: the stores go straight through; a device-scope fence keeps their order.  A system-scope
        // fence here writes the whole L2 back first -- the sweep's counters, the scores: 10-25 us of every level)
        __hip_atomic_store(&pin_words[1], (int)(vmax & 0x7fffffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pin_words[2], (int)(vmax >> 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&pin_words[5], decided ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (nrest_p) {
            __hip_atomic_store(&pin_words[6], *nrest_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&pin_words[7], *nact_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);       // (the tests of this level's list)
        }
        __threadfence();
        __hip_atomic_store(&pin_words[0], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);      // (the host spins on this word)
    }
}

// ===========================================================================================================================
// the exact step: the tests whose statistic the last level leaves within its resolution of the threshold
// ===========================================================================================================================
// One work-group per undecided test (slot u of the batch): its bins are UNIONS OF THE LAST LEVEL'S FINE BINS -- fine bin f with
// `cum` keys below it goes to bin cum / target, so a bin holds at most target keys plus one fine bin's -- and with the fine counts
// all-reduced their sizes and the keys in front of them are known before a key is placed: binmap (fine bin -> bin), hist, binbase,
// the cursors of the placing kernel (0), the test's table for that level again (tabx).
__global__ __launch_bounds__(1024) void k_wx_xplan(int NBX, int nbcap, unsigned int target, const int* __restrict__ actx, const int* __restrict__ nx_p,
                                                   int x_lo, const int* __restrict__ slotmap, const unsigned int* __restrict__ cl,
                                                   const unsigned int* __restrict__ c0, unsigned int* __restrict__ tabx,
                                                   unsigned short* __restrict__ binmap, unsigned int* __restrict__ hist,
                                                   unsigned int* __restrict__ binbase, unsigned int* __restrict__ cursor) {
    extern __shared__ unsigned int wxp_lds[];             // hist_s[nbcap], base_s[nbcap]
    __shared__ unsigned int wtot[16];
    const int u = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (x_lo + u >= *nx_p) return;
    const int seg = actx[x_lo + u];
    unsigned int* hist_s = wxp_lds;
    unsigned int* base_s = hist_s + nbcap;
    for (int b = t; b < nbcap; b += 1024) { hist_s[b] = 0u; base_s[b] = 0xffffffffu; }
    if (wave == 0) wx_wave_table(c0 + (size_t)seg * WX_NC0, NBX, tabx + (size_t)u * WX_NC0);
    const unsigned int* row = cl + (size_t)slotmap[seg] * NBX;
    const int per = NBX / 1024, f0 = t * per;              // (NBX: a multiple of 1024, at most 16 of them)
    unsigned int cf[WX_NBFMAX / 1024], loc = 0;
#pragma unroll
    for (int i = 0; i < WX_NBFMAX / 1024; i++) { cf[i] = i < per ? row[f0 + i] : 0u; loc += cf[i]; }
    unsigned int inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned int up = __shfl_up(inc, o, 64); if (lane >= o) inc += up; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    unsigned int cum = inc - loc;
    for (int w = 0; w < wave; w++) cum += wtot[w];
    // (a thread's consecutive fine bins mostly share a bin: their counts are added up in registers and go to LDS once per bin -- one
    // atomic per fine bin had four to sixty-four lanes of a wave on the same counter)
    unsigned int run_b = 0xffffffffu, run_c = 0, run_base = 0;
#pragma unroll
    for (int i = 0; i < WX_NBFMAX / 1024; i++) {
        if (i >= per) break;
        const unsigned int c = cf[i];
        unsigned int b = cum / target;
        b = b < (unsigned int)nbcap - 1u ? b : (unsigned int)nbcap - 1u;
        binmap[(size_t)u * NBX + f0 + i] = (unsigned short)b;
        if (c) {
            if (b != run_b) {
                if (run_c) { atomicAdd(&hist_s[run_b], run_c); atomicMin(&base_s[run_b], run_base); }
                run_b = b; run_c = 0; run_base = cum;
            }
            run_c += c;
        }
        cum += c;
    }
    if (run_c) { atomicAdd(&hist_s[run_b], run_c); atomicMin(&base_s[run_b], run_base); }
    __syncthreads();
    for (int b = t; b < nbcap; b += 1024) {
        hist[(size_t)u * nbcap + b] = hist_s[b];
        binbase[(size_t)u * nbcap + b] = hist_s[b] ? base_s[b] : 0u;
        cursor[(size_t)u * nbcap + b] = 0u;
    }
}

// the keys of the batch's tests (as the sweep wrote them, gathered over the ranks: keys_all[rank][slot][kld]) into their bins: a
// work-group takes 16384 keys of one (rank, test), counts them per bin in LDS (a key's place among the work-group's keys of its bin
// = the returned count), reserves the work-group's share of every bin it met with ONE global atomic on the bin's cursor, and writes
// the keys there.  Which share a work-group gets depends on the order of arrival; the rank sums do not (k_wx_ranks counts smaller
// and equal keys: exact half-integers whatever the order inside the bin).
constexpr int WX_PK = 16;
__global__ __launch_bounds__(1024) void k_wx_place(int NBX, int nbcap, const unsigned long long* __restrict__ keys_all, size_t kld, int XBn,
                                                   const int* __restrict__ nx_p, int x_lo, const int* __restrict__ actx,
                                                   const unsigned int* __restrict__ kbase, const unsigned int* __restrict__ tabx,
                                                   const unsigned short* __restrict__ binmap, const unsigned int* __restrict__ binbase,
                                                   unsigned int* __restrict__ cursor, unsigned long long* __restrict__ keysx, size_t xld) {
    extern __shared__ unsigned int wxl_lds[];             // tab_s[WX_NC0], cntb[nbcap], baseb[nbcap], map_s[NBX] (16 bit)
    const int u = blockIdx.y, q = blockIdx.z, t = threadIdx.x;
    if (x_lo + u >= *nx_p) return;
    const int seg = actx[x_lo + u];
    unsigned int* tab_s = wxl_lds;
    unsigned int* cntb = tab_s + WX_NC0;
    unsigned int* baseb = cntb + nbcap;
    unsigned short* map_s = (unsigned short*)(baseb + nbcap);
    for (int c = t; c < WX_NC0; c += 1024) tab_s[c] = tabx[(size_t)u * WX_NC0 + c];
    for (int b = t; b < nbcap; b += 1024) cntb[b] = 0u;
    for (int f = t; f < NBX; f += 1024) map_s[f] = binmap[(size_t)u * NBX + f];
    __syncthreads();
    const unsigned int kb = kbase[seg];
    const unsigned long long* src = keys_all + ((size_t)q * XBn + u) * kld;
    const size_t i0 = (size_t)blockIdx.x * (1024 * WX_PK);
    unsigned long long key[WX_PK];
    int bin[WX_PK];
    unsigned int lr[WX_PK];
#pragma unroll
    for (int v = 0; v < WX_PK; v++) {
        const size_t i = i0 + (size_t)v * 1024 + t;
        key[v] = i < kld ? src[i] : WX_NOKEY;
    }
#pragma unroll
    for (int v = 0; v < WX_PK; v++) {
        bin[v] = -1; lr[v] = 0;
        if (key[v] != WX_NOKEY) {
            bin[v] = (int)map_s[wx_fine((unsigned int)((key[v] & WX_MASK) >> 31), kb, tab_s)];
            lr[v] = atomicAdd(&cntb[bin[v]], 1u);
        }
    }
    __syncthreads();
    for (int b = t; b < nbcap; b += 1024) {
        const unsigned int c = cntb[b];
        if (c) baseb[b] = binbase[(size_t)u * nbcap + b] + atomicAdd(&cursor[(size_t)u * nbcap + b], c);
    }
    __syncthreads();
    unsigned long long* out = keysx + (size_t)u * xld;
#pragma unroll
    for (int v = 0; v < WX_PK; v++)
        if (bin[v] >= 0) out[baseb[bin[v]] + lr[v]] = key[v];
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
constexpr int WX_CAP = 16384;                    // keys one bin may hold (k_wx_ranks_big: the bin in 128 KB of LDS)
constexpr int WX_CAP_S = 4096;                  // keys of a bin this kernel takes (16 per thread)
constexpr int WX_NS = 1024;                     // linear sub-bins
constexpr int WX_WALK = 48;                     // longest sub-bin it walks
// (round 5: a slot u of the exact step's batch = test actx[x_lo + u]; its placed keys at keys + u xld, its bins in hist / binbase [u][nbcap])
__global__ __launch_bounds__(256) void k_wx_ranks(int nbcap, const int* __restrict__ actx, const int* __restrict__ nx_p, int x_lo,
                                                  const unsigned long long* __restrict__ keys, size_t xld,
                                                  const unsigned int* __restrict__ hist, const unsigned int* __restrict__ binbase,
                                                  double* __restrict__ W, unsigned int* __restrict__ big /* [0] count, then (slot, bin) pairs */) {
    constexpr int T = 256, KPT = WX_CAP_S / T;
    __shared__ unsigned long long ks[WX_CAP_S + 1];
    __shared__ unsigned int cnt[WX_NS + 2], start[WX_NS + 2];
    __shared__ unsigned long long rmm[2 * T / 64];
    __shared__ double red[T / 64];
    __shared__ unsigned int s_max[T / 64];
    const int u = blockIdx.y, b = blockIdx.x, t = threadIdx.x;
    if (x_lo + u >= *nx_p) return;
    const int seg = actx[x_lo + u];
    const unsigned int n = hist[(size_t)u * nbcap + b];
    if (n == 0) return;
    auto to_big = [&]() { if (t == 0) { const unsigned int e = atomicAdd(&big[0], 1u); big[1 + 2 * e] = (unsigned int)u; big[2 + 2 * e] = (unsigned int)b; } };
    if (n > (unsigned int)WX_CAP_S) { to_big(); return; }
    const unsigned int base = binbase[(size_t)u * nbcap + b];
    const unsigned long long* src = keys + (size_t)u * xld + base;
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
__global__ __launch_bounds__(1024) void k_wx_ranks_big(int nbcap, const int* __restrict__ actx, int x_lo, const unsigned long long* __restrict__ keys,
                                                       size_t xld, const unsigned int* __restrict__ hist,
                                                       const unsigned int* __restrict__ binbase, double* __restrict__ W,
                                                       const unsigned int* __restrict__ big, int* __restrict__ fail) {
    extern __shared__ unsigned long long wx_big_ks[];                       // WX_CAP keys
    __shared__ double red[32];
    const unsigned int nbig = big[0];
    for (unsigned int e = blockIdx.x; e < nbig; e += gridDim.x) {
        const unsigned int u = big[1 + 2 * e], b = big[2 + 2 * e];
        const int seg = actx[x_lo + (int)u];
        const unsigned int n = hist[(size_t)u * nbcap + b];
        if (n > (unsigned int)WX_CAP) { if (threadIdx.x == 0) *fail = 1; continue; }    // (the host repeats the reduction on the sorted path)
        const unsigned int base = binbase[(size_t)u * nbcap + b];
        const unsigned long long* src = keys + (size_t)u * xld + base;
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
                            const int* __restrict__ v3, double* __restrict__ dec = nullptr, const double* __restrict__ per_keep = nullptr,
                            int* __restrict__ pin_words = nullptr) {
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
        if (dec) dec[j] = (double)best; else model[ML.off_per + j] = (double)best;
        atomicMax(&s_ncomp, best);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (dec) dec[P] = (double)s_ncomp; else model[ML.off_hdr] = (double)s_ncomp;
        if (pin_words) {
            __hip_atomic_store(&pin_words[3], s_ncomp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&pin_words[4], (per_keep && s_ncomp != (int)per_keep[P]) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// the speculative run's decision (dec: P counts, then the largest) into the model record
__global__ void k_wx_commit(double* __restrict__ model, int M, int P, int A, const double* __restrict__ dec, int with_hdr) {
    const ModelLayout ML = model_layout(M, P, A);
    for (int j = threadIdx.x; j < P; j += blockDim.x) model[ML.off_per + j] = dec[j];
    if (threadIdx.x == 0 && with_hdr) model[ML.off_hdr] = dec[P];
}


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
    hipLaunchKernelGGL(k_wx_plan, dim3(1), dim3(256), 0, ctx->stream, (const double*)model, (int)M, (int)P, (int)A, plan, seg_j, seg_a, astar,
                       (int)nseg_max, nz, W, (int*)nullptr, (int*)nullptr, (int*)nullptr, (unsigned int*)nullptr, (int*)nullptr, (int*)nullptr,
                       (unsigned int*)nullptr, 1.0);
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
                                          (int)M, (int)P, (int)A, model, S, nt)
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


static bool wx_force_sorted() {          // A/B runs and tests (ABC_DIAG=1 ABC_WX_SORTED=1): the sorted path only
    static const bool on = abc_diag_env("ABC_WX_SORTED") != nullptr;
    return on;
}
static bool wx_no_bounds() {             // A/B runs and tests (ABC_DIAG=1 ABC_WX_NOBOUNDS=1): every needed test through the exact step
    static const bool on = abc_diag_env("ABC_WX_NOBOUNDS") != nullptr;
    return on;
}
// the cascade takes sets of at least 16384 validation rows (over all ranks) and at most 32 components; the rest goes to the sorted path
bool abc_wx_cascade_applies(size_t nv_total, size_t P, size_t A) {
    return !wx_force_sorted() && A >= 2 && A <= 32 && nv_total >= 16384 && nv_total < ((size_t)1 << 31) && P * (A - 1) <= MAXSEG;
}

namespace {
struct WxLevel { int R, tiles, G, TG, RR, tpw, nslots, TT; };   // rows per thread, tiles of TT R rows, tests per work-group, groups of tests,
                                                            // runs of tiles, tiles per run, tests of this launch
// A launch over `want` tests with NBX bins each: whole groups of G tests (a group's counters fill the LDS of a work-group), as many
// tests as the counter buffer (bc_bytes) takes.  ONE work-group runs on a CU (LDS), so about 256 of them: R rows per thread as long
// as (tiles x groups) still fills three quarters of the chip (more rows per thread = more loads in flight per latency), then
// 256 / groups runs of tiles; a work-group's rows stay below 2^16 (its counters are 16-bit halves).
WxLevel wx_level_one(size_t nt, size_t A, int want, int NBX, size_t per_test_lds, size_t bc_bytes, int fixed_slots);
// ... and the tests of a batch from the validation rows of ALL ranks (nvt), so that every rank of a row-sharded set cuts a level into
// the same batches -- a batch is one all-reduce of nslots x NBX counts, and ranks with different row counts (the validation rows are the
// global tail: leading ranks have none) would otherwise issue different numbers of collectives of different sizes;
// the geometry over the rank's own rows then takes that many tests as given.
WxLevel wx_level(size_t nt, size_t nvt, bool sharded, size_t A, int want, int NBX, size_t per_test_lds, size_t bc_bytes) {
    if (!sharded) return wx_level_one(nt, A, want, NBX, per_test_lds, bc_bytes, 0);
    const WxLevel all = wx_level_one(nvt > nt ? nvt : nt, A, want, NBX, per_test_lds, bc_bytes, -1);
    return wx_level_one(nt, A, want, NBX, per_test_lds, bc_bytes, all.nslots);
}
// fixed_slots > 0: that many tests, whatever the buffer; -1: the tests of a batch for ANY rank's share of the rows (a rank with fewer
// rows may cut them into MORE runs of tiles than the whole set would be: the buffer is sized for the most runs a geometry can have)
WxLevel wx_level_one(size_t nt, size_t A, int want, int NBX, size_t per_test_lds, size_t bc_bytes, int fixed_slots) {
    WxLevel g;
    g.G = (int)(((size_t)WX_LDS - 1024) / per_test_lds);          // (1 KB: the spare counters of the sweep)
    if (g.G < 1) g.G = 1;
    if (g.G > want) g.G = want;
    static const bool t768 = abc_diag_env("ABC_WX_T768") != nullptr;          // A/B switch: 17..32 components on 768 threads x 2 rows
    const int rmax = A <= 8 ? 4 : (A <= 16 ? 2 : (t768 ? 2 : 1));
    int ns = fixed_slots > 0 ? fixed_slots : want;
    for (int it = 0; it < 8; it++) {
        g.nslots = ns;
        g.TG = (ns + g.G - 1) / g.G;
        g.R = rmax;
        auto threads = [&](int R) { return (A > 16 && R == 2) ? 768 : WX_T; };
        while (g.R > 1 && ((nt + (size_t)threads(g.R) * g.R - 1) / ((size_t)threads(g.R) * g.R)) * (size_t)g.TG < 192) g.R >>= 1;
        g.TT = threads(g.R);
        g.tiles = (int)((nt + (size_t)g.TT * g.R - 1) / ((size_t)g.TT * g.R));
        const int limit = 65535 / (g.TT * g.R);
        int rr_target = 256 / g.TG;
        if (rr_target < 1) rr_target = 1;
        int tpw = (g.tiles + rr_target - 1) / rr_target;
        if (tpw < 1) tpw = 1;
        if (tpw > limit) tpw = limit;
        g.tpw = tpw;
        g.RR = (g.tiles + tpw - 1) / tpw;
        if (g.RR < 1) g.RR = 1;
        const int rr_bytes = (fixed_slots < 0 && g.RR < rr_target) ? rr_target : g.RR;
        const size_t bytes = (size_t)rr_bytes * ns * NBX * 4;
        if (fixed_slots > 0) {
            // (cannot happen by the sizing above; if it did, fewer and longer runs of tiles -- never a different batch)
            while ((size_t)g.RR * ns * NBX * 4 > bc_bytes && g.RR > 1 && g.tpw < limit) { g.tpw++; g.RR = (g.tiles + g.tpw - 1) / g.tpw; }
            break;
        }
        if (bytes <= bc_bytes || ns <= g.G) break;
        int fit = (int)(bc_bytes / ((size_t)rr_bytes * NBX * 4)) / g.G * g.G;
        if (fit < g.G) fit = g.G;
        if (fit >= ns) break;
        ns = fit;
    }
    return g;
}
size_t wx_bc_bytes(size_t nv, size_t nseg_max) {           // the sweeps' counter buffer: [runs of tiles][tests of a launch][bins]
    const size_t tiles = (nv + WX_T - 1) / WX_T;
    size_t b = (tiles > 0 ? tiles : 1) * (nseg_max * 2048 * 4 > (size_t)8 * WX_NBFMAX * 4 ? nseg_max * 2048 * 4 : (size_t)8 * WX_NBFMAX * 4);
    size_t cap = (size_t)96 << 20;
    static const char* cap_env = abc_diag_env("ABC_WX_BC_CAP_KB");          // tests: a small buffer, so that a level takes several batches
    if (cap_env && atol(cap_env) > 0) cap = (size_t)atol(cap_env) << 10;
    return (b < cap ? b : cap) + (1u << 20);
}
// bins of a fine level over `nact` tests.  A sweep work-group keeps the counters of G(bins) tests in LDS and every group of tests is
// one more pass over the validation rows' scores (8 A bytes a row, from L2 / HBM: at 32 components the passes ARE the sweep's time --
// 85 us for ten tests in three groups at 8192 bins, 293 us for 60 tests in eight groups at 4096; rocprofv3, round 6).  So the bins
// are chosen by what a level is expected to cost: its passes, plus ~1.5 passes for every test it is expected to leave open (the exact
// step's key pass, placing and ranking; or the next level's share) -- the resolution of B equi-depth bins is ~0.87 sqrt(m) / B
// sigma, and a test that reaches a fine level has its statistic near the threshold, where |W| / sigma has density ~0.2: about
// 126 / B sqrt(m / 5e5) of them stay open.  Rounds 4-5 took 16384 / 8192 / 4096 / 2048 bins by the number of tests alone.
int wx_pick_bins(int nact, size_t nvt) {
    static const char* fixed = abc_diag_env("ABC_WX_BINS_BY_COUNT");          // A/B switch: the rule of rounds 4-5
    if (fixed) {
        int nb = nact <= 8 ? 16384 : (nact <= 32 ? 8192 : (nact <= 96 ? 4096 : 2048));
        while (nb > 1024 && (size_t)nb * 4 > nvt) nb >>= 1;                // (at least four keys to the bin)
        return nb;
    }
    const double open_per_bin = 126.0 * sqrt((double)nvt / 5.0e5);
    int best = 1024;
    double best_cost = 1e300;
    for (int nb = 16384; nb >= 1024; nb >>= 1) {
        if (nb > 1024 && (size_t)nb * 4 > nvt) continue;                     // (at least four keys to the bin)
        int G = (int)(((size_t)WX_LDS - 1024) / ((size_t)nb * 4 + WX_NC0 * 4 + 7 * 4 + 16));
        if (G < 1) G = 1;
        const double passes = (double)((nact + G - 1) / G);
        double open = open_per_bin / nb;
        if (open > 1.0) open = 1.0;
        const double cost = passes + 1.5 * open * nact;
        if (cost < best_cost) { best_cost = cost; best = nb; }
    }
    return best;
}
int wx_xb(size_t nvt) {                           // tests of one batch of the exact step
    size_t xb = ((size_t)1 << 25) / (nvt ? nvt : 1);
    return xb < 1 ? 1 : (xb > 8 ? 8 : (int)xb);
}
unsigned int wx_target(size_t nvt) { const size_t t = (nvt + 3499) / 3500; return (unsigned int)(t > 2048 ? t : 2048); }
}  // namespace

static size_t wx_cascade_need(size_t nv, size_t P, size_t A) {
    if (!abc_wx_cascade_applies(nv, P, A)) return 0;
    const size_t seg = P * (A - 1);
    const int xb = wx_xb(nv);
    const size_t nbcap = nv / wx_target(nv) + 2;
    return nv * A * 8 + wx_bc_bytes(nv, seg) + seg * (2048 * 4 + 2048 * 8) + (size_t)40 * WX_NBFMAX * 12 + seg * WX_NC0 * (4 + 8) + seg * 64 + P * 16 +
           (size_t)xb * (2 * nv * 8 + WX_NBFMAX * 2 + WX_NC0 * 4 + nbcap * 20) + (2u << 20) +
           // (the largest count first: the picked responses' own level 0 and fine levels, <= 124 tests -- 96 x 4096 and 32 x 16384 bins of
           // counts and totals at most --, the second list of tests, the picks)
           ((size_t)24 << 20) + seg * 4 + P + 64;
}
size_t abc_wx_need(size_t nt, size_t P, size_t A) {
    const size_t b = wx_cascade_need(nt, P, A);
    return b ? b : wx_sorted_need(nt, P, A);       // (a cascade that has to be repeated on the sorted path allocates its own arena)
}

// the host's look at one word the device writes when it is done (pinned, preset to -1): a spin, with a glance at the stream
// every few thousand turns so that a failed launch ends the wait
static int wx_wait_word(abc_ctx* ctx, volatile int* w, int* out) {
    // the first 2 ms a plain spin (the word is usually tens of microseconds away and the generation's critical path waits for it);
    // then the core is given up between looks (sched_yield, from 50 ms on 50 us naps: other ranks' enqueue threads may share it),
    // and a wait beyond ABC_WX_WAIT_S seconds (default 600) fails the call instead of spinning on a wedged stream for ever
    static const double limit_s = abc_diag_env("ABC_WX_WAIT_S") ? atof(abc_diag_env("ABC_WX_WAIT_S")) : 600.0;
    const auto t0 = std::chrono::steady_clock::now();
    double waited = 0.0;
    for (unsigned long spins = 0;; spins++) {
        const int v = *w;
        if (v >= 0) { *out = v; return ABC_OK; }
        if ((spins & 0x3ff) == 0x3ff) {
            waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > limit_s) ABC_FAIL(ctx, ABC_ERR_HIP, "wilcoxon: no word from the bounds kernel after %.0f s", waited);
        }
        if (waited > 2e-3) {
            if (waited > 50e-3) { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }
            else sched_yield();
        }
        if ((spins & 0x3fff) == 0x3fff || (waited > 2e-3 && (spins & 0xff) == 0xff)) {
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) {
                const int v2 = *w;
                if (v2 >= 0) { *out = v2; return ABC_OK; }
                ABC_FAIL(ctx, ABC_ERR_HIP, "wilcoxon: the stream went idle without the bounds kernel's word");
            }
            if (e != hipErrorNotReady) ABC_FAIL(ctx, ABC_ERR_HIP, "wilcoxon: %s", hipGetErrorString(e));
        }
    }
}

template <int AM, int R, int TT>
static void wx_launch_sweep(abc_ctx* ctx, int mode, const WxLevel& g, size_t lds, const double* Y, size_t ldy, size_t row_test, size_t nt, size_t M,
                            size_t P, size_t A, const double* model, const double* S, size_t sld, const int* seg_j, const int* seg_a, const int* astar,
                            const int* act, const int* nact_p, int act_lo, const unsigned int* kbase, const unsigned int* c0, int NBX,
                            unsigned int* blockcnt, unsigned long long* keys, size_t kld) {
    const dim3 grid((unsigned)(8 * ((g.RR + 7) / 8) * g.TG));
#define WX_SW(MODEV)                                                                                                                      \
    do {                                                                                                                                  \
        if (lds > (48u << 10)) (void)hipFuncSetAttribute((const void*)k_wx_sweep<AM, R, MODEV, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_wx_sweep<AM, R, MODEV, TT>), grid, dim3(TT), lds, ctx->stream, Y, ldy, row_test, nt, (int)M, (int)P, (int)A, model, S, sld, \
                           seg_j, seg_a, astar, act, nact_p, act_lo, g.nslots, g.G, g.TG, g.RR, g.tpw, kbase, c0, NBX, blockcnt, keys, kld);  \
    } while (0)
    if (mode == 0) WX_SW(0); else if (mode == 1) WX_SW(1); else WX_SW(2);
#undef WX_SW
}
static void wx_sweep(abc_ctx* ctx, size_t A, int mode, const WxLevel& g, size_t lds, const double* Y, size_t ldy, size_t row_test,
                     size_t nt, size_t M, size_t P, const double* model, const double* S, size_t sld, const int* seg_j, const int* seg_a, const int* astar,
                     const int* act, const int* nact_p, int act_lo, const unsigned int* kbase, const unsigned int* c0, int NBX,
                     unsigned int* blockcnt, unsigned long long* keys, size_t kld) {
#define WX_GO(AMV, RV, TV) wx_launch_sweep<AMV, RV, TV>(ctx, mode, g, lds, Y, ldy, row_test, nt, M, P, A, model, S, sld, seg_j, seg_a, astar, act, nact_p, act_lo, \
                                                    kbase, c0, NBX, blockcnt, keys, kld)
    if (A <= 8) { if (g.R == 4) WX_GO(8, 4, 1024); else if (g.R == 2) WX_GO(8, 2, 1024); else WX_GO(8, 1, 1024); }
    else if (A <= 16) { if (g.R == 2) WX_GO(16, 2, 1024); else WX_GO(16, 1, 1024); }
    else if (g.R == 2) WX_GO(32, 2, 768);
    else WX_GO(32, 1, 1024);
#undef WX_GO
}

// scores of the validation rows (S[i + nt k], all A components)
static void wx_scores(abc_ctx* ctx, const double* X, size_t ldx, size_t row_test, size_t nt, size_t M, size_t P, size_t A, const double* model, double* S) {
    // the projection's kernels where the shape is theirs (row pairs with 16-byte loads; 17..32 components on the fp64 matrix pipe: the
    // vector kernel below took 605 us at 5e5 rows x 128 metrics x 32 components, k_project_mfma does twice the rows in 270), the
    // vector kernel for what they leave (an odd last row, unaligned or narrow sets)
    static const bool plain = abc_diag_env("ABC_WX_PLAIN_SCORES") != nullptr;
    const size_t took = plain ? 0 : launch_project_scores(ctx, X + row_test, nt, ldx, M, P, A, model, S);
    if (took >= nt) return;
    const size_t nt_all = nt;
    X += took; S += took; nt -= took;
    const unsigned rb = (unsigned)((nt + 255) / 256);
    int KC = 1;
    while (KC < (int)A) KC *= 2;
#define LAUNCH_SC(KCV) hipLaunchKernelGGL(k_wx_scores<KCV>, dim3(rb), dim3(256), 0, ctx->stream, X, ldx, row_test, nt, (int)M, (int)P, (int)A, model, S, nt_all)
    switch (KC) {
        case 1: LAUNCH_SC(1); break;
        case 2: LAUNCH_SC(2); break;
        case 4: LAUNCH_SC(4); break;
        case 8: LAUNCH_SC(8); break;
        case 16: LAUNCH_SC(16); break;
        default: LAUNCH_SC(32); break;
    }
#undef LAUNCH_SC
}

// The cascade, in two halves: begin() queues everything up to level 0's bounds and returns; finish() waits for the host's first
// look and goes on from there.  (The fused generation queues its ranking between the two: api.hip.)
// *fail_host = 1 when a bin of the exact step outgrew LDS (the caller repeats the reduction on the sorted path).
struct abc_wx_run {
    abc_ctx* ctx; const double* X; const double* Y; size_t nt, ldx, ldy, M, P, A, row_test; double* model; abc_wx_shard shv; bool has_sh;
    double* per_keep; double* dec; int stop_at_max; const abc_wx_scores_hook* scores_hook; bool hold_level0, level0_queued;
    int Wr; bool sharded; size_t nvt, nseg_max, bc_bytes;
    WxPlan* plan; int *seg_j, *seg_a, *astar, *segbase, *fail; unsigned long long* nz; double* W; int* v3; unsigned int* kbase;
    int *actA, *actB, *nactv; unsigned int* tickets; int* slotmap; unsigned char* passb; double* S; size_t S_ld; unsigned int *c0, *blockcnt;
    volatile int* pin; const double* nv_ranks; size_t nv_stride;
    unsigned int* cl_fine; size_t cl_fine_ld;
    // the largest count first (k_wx_plan): the tests of first_r picked responses (actA, at most first_bound of them), then -- only
    // if none of those responses keeps its optimum -- the rest (act_rest)
    int first_r, first_bound; int* act_rest; unsigned char* in_first; bool looked0; int left0; bool decided0;

    // one level over the tests act[0 .. nact_host) (nact on the device at nact_p): sweeps in batches, the counts (all-reduced over
    // the ranks), bounds; the last bounds work-group leaves the tests still needed in act_out / nact_out and their number in the
    // pinned word, which level_wait reads
    int level_queue(int lvl, int mode, int NBX, const int* act, const int* nact_p, int nact_host, int* act_out, int* nact_out, unsigned int* cl,
                    size_t cl_ld, int cl_by_test, bool first_half) {
        hipStream_t st = ctx->stream;
        const size_t per_test = (size_t)NBX * 4 + (mode == 1 ? WX_NC0 * 4 : 0) + 7 * 4 + 16;
        pin[0] = -1;
        const size_t blds = ((size_t)NBX + 1024) * 8;
        if (blds > (48u << 10)) ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_bounds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds));
        for (int lo = 0; lo < nact_host;) {
            const WxLevel g = wx_level(nt, nvt, sharded, A, nact_host - lo, NBX, per_test, bc_bytes);
            if ((size_t)g.RR * g.nslots * NBX * 4 > bc_bytes && sharded)
                ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: %d runs x %d tests x %d bins do not fit the counter buffer", g.RR, g.nslots, NBX);
            const size_t lds = (size_t)g.G * per_test + (mode == 0 ? (size_t)NBX * 4 : 256) + 64, ne = (size_t)g.nslots * NBX;
            unsigned long long* totals = (unsigned long long*)abc_ws_alloc(ctx, ne * 8);
            if (!totals) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted");
            if (nt) {
                wx_sweep(ctx, A, mode, g, lds, Y, ldy, row_test, nt, M, P, model, S, S_ld, seg_j, seg_a, astar, act, nact_p, lo, kbase, c0, NBX, blockcnt, nullptr, 0);
                hipLaunchKernelGGL(k_wx_totals, dim3((unsigned)((ne + 31) / 32)), dim3(256), 0, st, NBX, g.RR, g.nslots, (const unsigned int*)blockcnt, totals);
            } else
                ABC_HIP(ctx, hipMemsetAsync(totals, 0, ne * 8, st));
            if (sharded) ABC_TRY(abc_comm_all_reduce(ctx, totals, ne, ABC_DT_I64));
            hipLaunchKernelGGL(k_wx_bounds, dim3((unsigned)g.nslots), dim3(1024), blds, st, NBX, g.nslots, act, nact_p, lo, (const unsigned long long*)totals,
                               nz, v3, cl, cl_ld, cl_by_test, slotmap, tickets + lvl, (unsigned int)nact_host, (const int*)astar, (int)P, (const int*)segbase,
                               act_out, nact_out, nv_ranks, nv_stride, Wr, (int*)pin, wx_no_bounds() ? 1 : 0, model, (int)M, (int)A, dec,
                               (const double*)per_keep, stop_at_max, first_half ? (const unsigned char*)in_first : (const unsigned char*)nullptr,
                               first_half ? (const int*)(nactv + 4) : (const int*)nullptr);
            ABC_HIP(ctx, hipGetLastError());
            lo += g.nslots;
        }
        return ABC_OK;
    }
    int level_wait(int* left, bool* decided) {
        ABC_TRY(wx_wait_word(ctx, pin, left));
        *decided = pin[5] != 0;                 // (written in front of the word the wait saw)
        return ABC_OK;
    }

    int begin() {
        Wr = (has_sh && ctx->comm_kind) ? ctx->comm_world : 1;
        sharded = Wr > 1;
        nvt = has_sh ? shv.nv_total : nt;
        nseg_max = P * (A - 1);
        bc_bytes = wx_bc_bytes(nvt, nseg_max);
        hipStream_t st = ctx->stream;
        plan = (WxPlan*)abc_ws_alloc(ctx, sizeof(WxPlan));
        seg_j = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        seg_a = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        astar = (int*)abc_ws_alloc(ctx, P * sizeof(int));
        segbase = (int*)abc_ws_alloc(ctx, P * sizeof(int));
        fail = (int*)abc_ws_alloc(ctx, 2 * sizeof(int));
        nz = (unsigned long long*)abc_ws_alloc(ctx, nseg_max * 8);
        W = (double*)abc_ws_alloc(ctx, nseg_max * 8);
        v3 = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        kbase = (unsigned int*)abc_ws_alloc(ctx, nseg_max * 4);
        actA = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        actB = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        nactv = (int*)abc_ws_alloc(ctx, WX_NWORDS * sizeof(int));
        tickets = (unsigned int*)abc_ws_alloc(ctx, WX_NWORDS * sizeof(int));
        // the largest count first: where the caller only uses the largest per-response count and there are more responses than picks
        static const char* first_env = abc_diag_env("ABC_WX_FIRST");            // A/B switch and tests: 0 = every test at level 0, as round 5
        first_r = 0;
        if (stop_at_max && P <= 1024 && A >= 2) {
            first_r = 32 / (int)(A - 1);
            first_r = first_r < 2 ? 2 : (first_r > 4 ? 4 : first_r);
            if (first_env) first_r = atoi(first_env) < WX_FIRST_MAX ? atoi(first_env) : WX_FIRST_MAX;
            if ((size_t)first_r >= P) first_r = 0;
        }
        first_bound = first_r ? first_r * (int)(A - 1) : (int)nseg_max;
        act_rest = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        in_first = (unsigned char*)abc_ws_alloc(ctx, P ? P : 1);
        looked0 = false; left0 = 0; decided0 = false;
        slotmap = (int*)abc_ws_alloc(ctx, nseg_max * sizeof(int));
        passb = (unsigned char*)abc_ws_alloc(ctx, nseg_max);
        S = nullptr;
        S_ld = nt;
        c0 = (unsigned int*)abc_ws_alloc(ctx, nseg_max * WX_NC0 * 4);
        blockcnt = (unsigned int*)abc_ws_alloc(ctx, bc_bytes);
        if (!plan || !seg_j || !seg_a || !astar || !segbase || !fail || !nz || !W || !v3 || !kbase || !actA || !actB || !nactv || !tickets || !slotmap ||
            !passb || !c0 || !blockcnt || !act_rest || !in_first)
            ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%zu tests x %zu rows)", nseg_max, nt);
        pin = (volatile int*)(ctx->status_pin + 64);
        nv_ranks = has_sh ? shv.nv_ranks : nullptr;
        nv_stride = has_sh ? shv.nv_stride : 0;
        cl_fine = nullptr;
        cl_fine_ld = 0;
        hipLaunchKernelGGL(k_wx_plan, dim3(1), dim3(256), 0, st, (const double*)model, (int)M, (int)P, (int)A, plan, seg_j, seg_a, astar, (int)nseg_max, nz, W,
                           segbase, fail, v3, kbase, actA, nactv, tickets, (double)nvt, per_keep, first_r, act_rest, in_first);
        if (nt) {
            int hooked = 1;                              // (1: the caller has no pass of its own for these rows)
            if (scores_hook) hooked = scores_hook->fn(scores_hook->arg, &S, &S_ld);      // (0: it has queued the pass and says where the scores go)
            if (hooked < 0) return hooked;
            if (hooked) {
                S = (double*)abc_ws_alloc(ctx, nt * A * 8);
                S_ld = nt;
                if (!S) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%zu x %zu scores)", nt, A);
                wx_scores(ctx, X, ldx, row_test, nt, M, P, A, model, S);
            }
        }
        ABC_HIP(ctx, hipGetLastError());
        level0_queued = false;
        return hold_level0 ? ABC_OK : level0();
    }
    // level 0 (sweep, totals, bounds): part of begin(), or queued by the caller once its own launches are out (hold_level0)
    int level0() {
        if (level0_queued) return ABC_OK;
        level0_queued = true;
        return level_queue(0, 0, WX_NC0, actA, nactv, first_bound, actB, nactv + 1, c0, WX_NC0, 1, first_r > 0);
    }
    // the host's first look (level 0 of the picked responses, or of everything): how many tests still want a closer look, and whether
    // the counts (the largest count) are settled already -- the fused generation calls it in front of its weight stage (api.hip)
    int look0(int* left, bool* decided) {
        if (!looked0) {
            ABC_TRY(level0());
            ABC_TRY(level_wait(&left0, &decided0));
            looked0 = true;
        }
        *left = left0; *decided = decided0;
        return ABC_OK;
    }

    int finish(int* fail_host, int* changed_host) {
        hipStream_t st = ctx->stream;
        int left = 0;
        bool decided = false;
        int* act_cur = actB;
        int* act_nxt = actA;
        const int* cur_n_p = nactv + 1;         // the device word that holds the length of act_cur
        int NBX_last = 0;
        ABC_TRY(look0(&left, &decided));
        static const bool dbg_levels = abc_diag_env("ABC_WX_DEBUG") != nullptr;
        if (dbg_levels) fprintf(stderr, "WX_LEVELS: level 0 (%s) leaves %d tests, decided %d\n", first_r > 0 ? "picked responses" : "all tests", left, (int)decided);
        // fine levels over what is left: at most two, the second only when few tests remain and finer bins are to be had.  word0: the
        // first of the half's ticket / length words (a level reads its list's length at word0 + 1 + f, leaves the next at + 2 + f)
        auto fine_levels = [&](int word0, bool first_half) -> int {
            for (int f = 0; f < 2 && !decided && left > 0; f++) {
                const int NBX = wx_pick_bins(left, nvt);
                if (f == 1 && (NBX <= NBX_last || left > 32)) break;
                cl_fine_ld = (size_t)NBX;
                cl_fine = (unsigned int*)abc_ws_alloc(ctx, (size_t)left * NBX * 4);
                if (!cl_fine) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (%d tests x %d bins)", left, NBX);
                const int nact_host = left;
                ABC_TRY(level_queue(word0 + 1 + f, 1, NBX, act_cur, nactv + word0 + 1 + f, nact_host, act_nxt, nactv + word0 + 2 + f, cl_fine, cl_fine_ld, 0,
                                    first_half));
                ABC_TRY(level_wait(&left, &decided));
                if (dbg_levels) fprintf(stderr, "WX_LEVELS: fine level (%s, %d tests, %d bins) leaves %d, decided %d\n", first_half ? "first half" : "all", nact_host, NBX, left, (int)decided);
                int* tmp = act_cur; act_cur = act_nxt; act_nxt = tmp;
                cur_n_p = nactv + word0 + 2 + f;
                NBX_last = NBX;
            }
            return ABC_OK;
        };
        if (first_r > 0) {
            // the first half: the picked responses' tests through the fine levels; then, unless one of them keeps the largest count,
            // level 0 of ALL THE OTHER tests -- from there on as if level 0 had taken every test at once (the picked responses' verdicts
            // stand; those of their tests that are still open join the others' in the fine levels and the exact step)
            // (tried, round 6: no fine level of their own when more than a third of the picked tests are still open after level 0 -- 15 of
            // 28, 16 of 26, 20 of 28 on sets whose count moves, 9 of 62 and none on the clean ones.  Slower, not faster: +0.42 against
            // +0.36 ms for a moved count at configs[2], same box -- the verdicts of that level let the level over the other responses drop
            // most of their tests, 4 of 83 left instead of 50)
            ABC_TRY(fine_levels(0, true));
            if (!decided) {
                // (the host sizes the launch for the tests there are -- pin[6], left by the first half's bounds kernels; sized for all
                // P (A - 1) that a set could have, two thirds of the work-groups of a 77-test level found nothing to do and the rest
                // had a third of the chip: 216 us at 1e6 x 128 x 16 x 32.  A level without tests still runs its decision.)
                int n_rest = pin[6];
                if (n_rest < 1) n_rest = 1;
                if (n_rest > (int)nseg_max) n_rest = (int)nseg_max;
                ABC_TRY(level_queue(4, 0, WX_NC0, act_rest, nactv + 4, n_rest, actB, nactv + 5, c0, WX_NC0, 1, false));
                ABC_TRY(level_wait(&left, &decided));
                if (dbg_levels) fprintf(stderr, "WX_LEVELS: level 0 of the other %d tests leaves %d, decided %d\n", n_rest, left, (int)decided);
                act_cur = actB; act_nxt = actA; cur_n_p = nactv + 5; NBX_last = 0;
                ABC_TRY(fine_levels(4, false));
            }
        } else
            ABC_TRY(fine_levels(0, false));
        // ---- the exact step -------------------------------------------------------------------------------------------------------
        bool exact = !decided && left > 0;
        if (exact) {
            const int nx = left, XB = wx_xb(nvt) < nx ? wx_xb(nvt) : nx, NBX = NBX_last;
            const unsigned int target = wx_target(nvt);
            const int nbcap = (int)(nvt / target) + 2;
            size_t vmax = nt;
            if (sharded) vmax = (size_t)pin[1] | ((size_t)pin[2] << 31);
            if (vmax < nt) ABC_FAIL(ctx, ABC_ERR_COMM, "wilcoxon: %zu validation rows on this rank, %zu at most on any", nt, vmax);
            unsigned long long* keys_loc = (unsigned long long*)abc_ws_alloc(ctx, (size_t)XB * (vmax ? vmax : 1) * 8);
            unsigned long long* keysx = (unsigned long long*)abc_ws_alloc(ctx, (size_t)XB * nvt * 8);
            unsigned int* tabx = (unsigned int*)abc_ws_alloc(ctx, (size_t)XB * WX_NC0 * 4);
            unsigned short* binmap = (unsigned short*)abc_ws_alloc(ctx, (size_t)XB * NBX * 2);
            unsigned int* hist = (unsigned int*)abc_ws_alloc(ctx, (size_t)XB * nbcap * 4);
            unsigned int* binbase = (unsigned int*)abc_ws_alloc(ctx, (size_t)XB * nbcap * 4);
            unsigned int* cursor = (unsigned int*)abc_ws_alloc(ctx, (size_t)XB * nbcap * 4);
            unsigned int* big = (unsigned int*)abc_ws_alloc(ctx, (1 + 2 * (size_t)XB * nbcap) * 4);
            const int* nxd = cur_n_p;
            if (!keys_loc || !keysx || !tabx || !binmap || !hist || !binbase || !cursor || !big)
                ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted (the exact step: %d tests x %zu rows)", XB, nvt);
            unsigned long long* keys_all = keys_loc;
            if (sharded) {
                ABC_TRY(abc_xbuf_reserve(ctx, (size_t)Wr * XB * vmax * 8 + 4096));
                keys_all = (unsigned long long*)ctx->xbuf;
            }
            const int rkeys = A <= 8 ? 4 : (A <= 16 ? 2 : 1);
            const size_t plds = (size_t)2 * nbcap * 4;
            const size_t llds = (size_t)WX_NC0 * 4 + (size_t)2 * nbcap * 4 + (size_t)NBX * 2 + 16;
            if (plds > (48u << 10)) ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_xplan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
            if (llds > (48u << 10)) ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_place, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
            ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_wx_ranks_big, hipFuncAttributeMaxDynamicSharedMemorySize, WX_CAP * 8));
            for (int x_lo = 0; x_lo < nx; x_lo += XB) {
                const int xb = nx - x_lo < XB ? nx - x_lo : XB;
                hipLaunchKernelGGL(k_wx_xplan, dim3((unsigned)xb), dim3(1024), plds, st, NBX, nbcap, target, (const int*)act_cur, nxd, x_lo,
                                   (const int*)slotmap, (const unsigned int*)cl_fine, (const unsigned int*)c0, tabx, binmap, hist, binbase, cursor);
                if (vmax) {
                    WxLevel g;
                    g.R = rkeys; g.TT = WX_T; g.tiles = (int)((vmax + (size_t)WX_T * rkeys - 1) / ((size_t)WX_T * rkeys));
                    g.G = xb; g.TG = 1; g.RR = g.tiles; g.tpw = 1; g.nslots = xb;
                    const size_t lds = (size_t)xb * (7 * 4 + 16) + 64;
                    wx_sweep(ctx, A, 2, g, lds, Y, ldy, row_test, nt, M, P, model, S, S_ld, seg_j, seg_a, astar, act_cur, nxd, x_lo, kbase, c0, NBX, nullptr,
                             keys_loc, vmax);
                    // (a batch shorter than XB leaves the tail of the block as it is: the placing kernel does not look at it)
                }
                if (sharded) ABC_TRY(abc_comm_all_gather(ctx, keys_loc, keys_all, (size_t)XB * vmax * 8));
                ABC_HIP(ctx, hipMemsetAsync(big, 0, 4, st));
                if (vmax)
                    hipLaunchKernelGGL(k_wx_place, dim3((unsigned)((vmax + 1024 * WX_PK - 1) / (1024 * WX_PK)), (unsigned)xb, (unsigned)Wr), dim3(1024), llds, st, NBX,
                                       nbcap, (const unsigned long long*)keys_all, vmax, XB, nxd, x_lo, (const int*)act_cur,
                                       (const unsigned int*)kbase, (const unsigned int*)tabx, (const unsigned short*)binmap, (const unsigned int*)binbase, cursor,
                                       keysx, nvt);
                hipLaunchKernelGGL(k_wx_ranks, dim3((unsigned)nbcap, (unsigned)xb), dim3(256), 0, st, nbcap, (const int*)act_cur, nxd, x_lo,
                                   (const unsigned long long*)keysx, nvt, (const unsigned int*)hist, (const unsigned int*)binbase, W, big);
                hipLaunchKernelGGL(k_wx_ranks_big, dim3(256), dim3(1024), (size_t)WX_CAP * 8, st, nbcap, (const int*)act_cur, x_lo,
                                   (const unsigned long long*)keysx, nvt, (const unsigned int*)hist, (const unsigned int*)binbase, W, (const unsigned int*)big, fail);
                ABC_HIP(ctx, hipGetLastError());
            }
        }
        if (exact)          // (else: the last bounds kernel has written the component counts itself)
            hipLaunchKernelGGL(k_wx_decide, dim3(1), dim3(1024), 0, st, model, (int)M, (int)P, (int)A, plan, nz, W, passb, (const int*)v3, dec,
                               (const double*)per_keep, (int*)pin);
        ABC_HIP(ctx, hipGetLastError());
        *fail_host = 0;
        if (exact) {
            // did every bin of the exact step fit?  (k_wx_decide has otherwise written a count from incomplete sums: the sorted path
            // overwrites it)
            ABC_HIP(ctx, hipMemcpyAsync(fail_host, fail, sizeof(int), hipMemcpyDeviceToHost, st));
            ABC_HIP(ctx, hipStreamSynchronize(st));
        }
        if (changed_host) *changed_host = pin[4];          // (visible: written in front of the word the last wait / the synchronisation saw)
        if (abc_diag_env("ABC_WX_DEBUG")) {          // (diagnostic: how the tests were settled)
            std::vector<int> hv(nseg_max);
            WxPlan hp;
            ABC_HIP(ctx, hipMemcpy(&hp, plan, sizeof(WxPlan), hipMemcpyDeviceToHost));
            ABC_HIP(ctx, hipMemcpy(hv.data(), v3, nseg_max * sizeof(int), hipMemcpyDeviceToHost));
            int cnt[3] = {0, 0, 0};
            for (int i = 0; i < hp.nseg; i++) cnt[hv[i] < 0 || hv[i] > 2 ? 2 : hv[i]]++;
            fprintf(stderr, "WX_DEBUG: %d tests over %zu rows (%zu here): rejected %d, passed %d, undecided %d by the bounds (exact step for %d tests, last level %d bins)%s\n",
                    hp.nseg, nvt, nt, cnt[0], cnt[1], cnt[2], left, NBX_last, *fail_host ? " (a bin outgrew LDS: repeat on the sorted path)" : "");
        }
        return ABC_OK;
    }
};

int launch_wilcoxon(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M,
                    size_t P, size_t A, size_t row_test, double* model, const abc_wx_shard* sh, double* dec, int* changed_host, int stop_at_max) {
    if (changed_host) *changed_host = 2;                 // (2: the model record itself was rewritten -- any path but the cascade's own end)
    const size_t nt = n > row_test ? n - row_test : 0;   // validation rows here
    const size_t nvt = sh ? sh->nv_total : nt;           // ... and over all ranks
    if (nvt == 0) return ABC_OK;                         // empty validation set: nothing to reduce
    if (A < 2 || P == 0) return ABC_OK;
    if (P * (A - 1) > MAXSEG)
        ABC_FAIL(ctx, ABC_ERR_UNSUPPORTED, "wilcoxon: P (A - 1) = %zu tests, more than %zu", P * (A - 1), MAXSEG);
    StageTimer tm(ctx, ST_PLS_MODEL);
    if (abc_wx_cascade_applies(nvt, P, A)) {
        // k_wx_decide needs the PRESS optima as the model fit left them: the cascade rewrites them, so keep a copy for a repeat
        const ModelLayout ML = model_layout(M, P, A);
        int failed = 0;
        double* per_keep = (double*)abc_ws_alloc(ctx, (P + 1) * 8);
        if (!per_keep) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: workspace exhausted");
        int changed = 2;
        abc_wx_run run;
        memset((void*)&run, 0, sizeof(run));
        run.ctx = ctx; run.X = X; run.Y = Y; run.nt = nt; run.ldx = ldx; run.ldy = ldy; run.M = M; run.P = P; run.A = A; run.row_test = row_test;
        run.model = model; run.has_sh = sh != nullptr; if (sh) run.shv = *sh;
        run.per_keep = per_keep; run.dec = dec; run.stop_at_max = stop_at_max;
        ABC_TRY(run.begin());                                   // (its plan kernel makes the copy of the PRESS optima)
        ABC_TRY(run.finish(&failed, &changed));
        static const bool force_fail = abc_diag_env("ABC_WX_FORCE_FAIL") != nullptr;   // tests: exercise the repeat
        if (!failed && !force_fail) { if (changed_host && dec) *changed_host = changed; return ABC_OK; }
        if (dec) return ABC_INTERNAL_RETRY;      // (a speculative run: the model record is as the fit left it; the caller runs the reduction again, by itself)
        ABC_HIP(ctx, hipMemcpyAsync(model + ML.off_per, per_keep, P * 8, hipMemcpyDeviceToDevice, ctx->stream));
        ABC_HIP(ctx, hipMemcpyAsync(model + ML.off_hdr, per_keep + P, 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (sh) return ABC_INTERNAL_RETRY;       // (row shards: the caller gathers the validation rows and calls again without `sh`)
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
    if (sh) return ABC_INTERNAL_RETRY;           // small sets: the caller gathers the rows
    return launch_wilcoxon_sorted(ctx, X, Y, n, ldx, ldy, M, P, A, row_test, model);
}

// the decision of a speculative run (launch_wilcoxon with dec) into the model record, on the context's stream
int launch_wilcoxon_commit(abc_ctx* ctx, double* model, size_t M, size_t P, size_t A, const double* dec, int with_hdr) {
    hipLaunchKernelGGL(k_wx_commit, dim3(1), dim3(256), 0, ctx->stream, model, (int)M, (int)P, (int)A, dec, with_hdr);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// The two halves for a caller that has work to queue between them (the fused generation's speculative ranking, api.hip): begin
// queues the cascade up to level 0's bounds on the context's stream and returns at once; finish waits for the host's looks, runs
// what is left, and says whether the largest count differs from the fit's (*changed_host: 0 / 1).  The decision goes to dec (P
// counts, then the largest); the model record stays as the fit wrote it.  ABC_INTERNAL_RETRY from finish: a bin of the exact step
// outgrew LDS -- the caller runs launch_wilcoxon in stream order instead.
int launch_wilcoxon_begin(abc_ctx* ctx, const double* X, const double* Y, size_t n, size_t ldx, size_t ldy, size_t M, size_t P, size_t A,
                          size_t row_test, double* model, double* dec, int stop_at_max, abc_wx_run** out, const abc_wx_scores_hook* scores,
                          int hold_level0, const abc_wx_shard* sh) {
    *out = nullptr;
    const size_t nt = n > row_test ? n - row_test : 0;
    if (!abc_wx_cascade_applies(sh ? sh->nv_total : nt, P, A) || !dec) ABC_FAIL(ctx, ABC_ERR_INVALID, "wilcoxon: not a set for the two-halves cascade");
    abc_wx_run* run = new (std::nothrow) abc_wx_run;
    if (!run) ABC_FAIL(ctx, ABC_ERR_NOMEM, "wilcoxon: host memory");
    memset((void*)run, 0, sizeof(*run));
    run->ctx = ctx; run->X = X; run->Y = Y; run->nt = nt; run->ldx = ldx; run->ldy = ldy; run->M = M; run->P = P; run->A = A; run->row_test = row_test;
    run->model = model; run->has_sh = sh != nullptr; if (sh) run->shv = *sh;
    run->dec = dec; run->stop_at_max = stop_at_max; run->scores_hook = scores; run->hold_level0 = hold_level0 != 0;
    run->per_keep = (double*)abc_ws_alloc(ctx, (P + 1) * 8);
    int rc = run->per_keep ? ABC_OK : ABC_ERR_NOMEM;
    if (rc == ABC_OK) rc = run->begin();
    run->scores_hook = nullptr;                          // (the caller's: not kept beyond this call)
    if (rc != ABC_OK) { delete run; if (rc == ABC_ERR_NOMEM && !ctx->err[0]) snprintf(ctx->err, sizeof(ctx->err), "wilcoxon: workspace exhausted"); return rc; }
    *out = run;
    return ABC_OK;
}
int launch_wilcoxon_level0(abc_ctx* ctx, abc_wx_run* run) { (void)ctx; return run->level0(); }
int launch_wilcoxon_finish(abc_ctx* ctx, abc_wx_run* run, int* changed_host) {
    int failed = 0, changed = 0;
    const int rc = run->finish(&failed, &changed);
    delete run;
    ABC_TRY(rc);
    static const bool force_fail = abc_diag_env("ABC_WX_FORCE_FAIL") != nullptr;   // tests: exercise the repeat
    if (failed || force_fail) return ABC_INTERNAL_RETRY;
    if (changed_host) *changed_host = changed;
    return ABC_OK;
}
// a begun cascade that will not be finished (an error between the halves): wait for what it queued, release the handle
void launch_wilcoxon_abandon(abc_ctx* ctx, abc_wx_run* run, hipStream_t its_stream) {
    if (!run) return;
    (void)hipStreamSynchronize(its_stream);
    delete run;
    (void)ctx;
}
