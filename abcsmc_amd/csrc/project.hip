// Per-particle projection onto the PLS components and Euclidean distance to the observed scores:
// dist_i = || z(x_i) R[:, :a] - obs_scores ||_2   (AbcUtil.cpp:434, 453-455; euclidean :320-324),
// or, for FILTER::SIMPLE, || z(x_i) - z(obs) ||_2 (AbcUtil.cpp:412-419).
//
// One particle per lane, metrics streamed column by column (each wave-instruction reads 64
// consecutive particles of one metric = 512 contiguous bytes); the z-score is applied on the fly
// (no materialised copy); R / mean / sd are wave-uniform and come through the scalar cache.
// The operation order is FIXED and shared with the oracle (orc_project_distance): for each
// metric m ascending, z = (x - mean)/sd (true division), s_k = fma(z, R[m,k], s_k); then
// d2 = fma(s_k - o_k, s_k - o_k, d2) for k ascending; dist = sqrt(d2).  HBM-bound: 8*M B/particle.
#include <stdlib.h>

#include <hip/hip_ext.h>
#include "abc_internal.h"

typedef double d2 __attribute__((ext_vector_type(2)));

namespace {

// two particles per lane (16-B loads: a wave-instruction reads 1 KiB of one metric column); needs 16-B aligned
// columns.  Per-particle arithmetic is identical to k_project_dist (same operation order -> same bits).
template <int KC>
__global__ __launch_bounds__(256) void k_project_dist2(const double* __restrict__ X, size_t npairs, size_t ldx, int M,
                                                       const double* __restrict__ mean,
                                                       const double* __restrict__ sd,
                                                       const double* __restrict__ Rpad,
                                                       const double* __restrict__ opad,
                                                       double* __restrict__ dist) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npairs; i += stride) {
        double s0[KC], s1[KC];
#pragma unroll
        for (int k = 0; k < KC; k++) { s0[k] = 0.0; s1[k] = 0.0; }
        const double* xp = X + 2 * i;
#pragma unroll 8
        for (int m = 0; m < M; m++) {
            const d2 x = *reinterpret_cast<const d2*>(xp + (size_t)m * ldx);
            const double sdm = sd[m], mu = mean[m];
            const double z0 = (sdm == 0.0) ? 0.0 : (x.x - mu) / sdm;
            const double z1 = (sdm == 0.0) ? 0.0 : (x.y - mu) / sdm;
#pragma unroll
            for (int k = 0; k < KC; k++) {
                const double r = Rpad[m * KC + k];
                s0[k] = fma(z0, r, s0[k]);
                s1[k] = fma(z1, r, s1[k]);
            }
        }
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const double t0 = s0[k] - opad[k], t1 = s1[k] - opad[k];
            d0 = fma(t0, t0, d0);
            d1 = fma(t1, t1, d1);
        }
        *reinterpret_cast<d2*>(dist + 2 * i) = (d2){sqrt(d0), sqrt(d1)};
    }
}

// 16 / 32 components: the scalar-register operands of k_project_dist2 no longer fit (32 components x 8 metrics in flight = 256
// SGPR pairs: hipcc spilled 488 of them to VGPR lanes and the kernel ran at a sixth of its memory time); here the padded loadings
// live in LDS (M x KC doubles, every work-group copies them once) and are read with wave-uniform 16-byte reads (broadcast), the
// metric columns of a row pair are fetched PF metrics ahead.  Same per-particle operation order as k_project_dist -> same bits.
template <int KC>
__global__ __launch_bounds__(256) void k_project_dist2_lds(const double* __restrict__ X, size_t npairs, size_t ldx, int M,
                                                           const double* __restrict__ mean, const double* __restrict__ sd,
                                                           const double* __restrict__ model, size_t off_R, size_t off_oscore,
                                                           double* __restrict__ dist, double* __restrict__ Sout = nullptr, size_t sld = 0,
                                                           int nc_force = 0, size_t row_split = 0) {
    // Sout (round 5, the Wilcoxon reduction's validation scores): the first nc_force scores of every row from row_split (an even
    // number) on go to Sout[row - row_split + sld k] -- the same fma chains, hence the bits the distances are made of.  Without
    // dist that is all (the rule's own pass); with dist the launch is the ranking's projection AND the rule's: all nc_force
    // components are scored, the distance takes the first model[0] of them (a term beyond those is fma(0, 0, d) = d whether its
    // score is a padding zero or switched off here: the same bits as the distance-only launch)
    extern __shared__ double Rl[];                                       // M*KC + KC
    double* const opad = Rl + (size_t)M * KC;
    const int ncomp_dist = dist ? (int)model[0] : 0;
    {   // the zero-padded loadings and observed scores straight from the model (what k_pad_model writes for the other kernels)
        const int ncomp = Sout ? nc_force : ncomp_dist;
        for (int e = threadIdx.x; e < M * KC; e += 256) {
            const int m = e / KC, k = e % KC;
            Rl[e] = (k < ncomp) ? model[off_R + m + (size_t)M * k] : 0.0;
        }
        if (threadIdx.x < KC) opad[threadIdx.x] = (threadIdx.x < ncomp_dist) ? model[off_oscore + threadIdx.x] : 0.0;
    }
    __syncthreads();
    constexpr int PF = 4, H = KC / 4;                                    // H: 16-byte reads per half row of loadings
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npairs; i += stride) {
        double s0[KC], s1[KC];
#pragma unroll
        for (int k = 0; k < KC; k++) { s0[k] = 0.0; s1[k] = 0.0; }
        const double* xp = X + 2 * i;
        // software pipeline, written out because hipcc would not build it: the metric columns two chunks of PF ahead in two
        // register sets (xa / xb), the loadings of a metric in two halves (ra / rb), each half fetched while the other is used
        d2 xa[PF], xb[PF], ra[H], rb[H];
        auto load_x = [&](d2 (&xq)[PF], int m0) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int m = (m0 + u < M) ? m0 + u : M - 1;             // in-range address; unused beyond M
                xq[u] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(xp + (size_t)m * ldx));
            }
        };
        auto load_r = [&](d2 (&rq)[H], int m, int half) {
            const d2* rr = reinterpret_cast<const d2*>(Rl + (size_t)((m < M) ? m : M - 1) * KC) + half * H;
#pragma unroll
            for (int h = 0; h < H; h++) rq[h] = rr[h];
        };
        auto chunk = [&](const d2 (&xq)[PF], int m0) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const int m = m0 + u;
                if (m < M) {                                             // (uniform)
                    const double sdm = sd[m], mu = mean[m];
                    const double z0 = (sdm == 0.0) ? 0.0 : (xq[u].x - mu) / sdm;
                    const double z1 = (sdm == 0.0) ? 0.0 : (xq[u].y - mu) / sdm;
                    load_r(rb, m, 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        s0[2 * h] = fma(z0, ra[h].x, s0[2 * h]);
                        s1[2 * h] = fma(z1, ra[h].x, s1[2 * h]);
                        s0[2 * h + 1] = fma(z0, ra[h].y, s0[2 * h + 1]);
                        s1[2 * h + 1] = fma(z1, ra[h].y, s1[2 * h + 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    load_r(ra, m + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < H; h++) {
                        s0[2 * H + 2 * h] = fma(z0, rb[h].x, s0[2 * H + 2 * h]);
                        s1[2 * H + 2 * h] = fma(z1, rb[h].x, s1[2 * H + 2 * h]);
                        s0[2 * H + 2 * h + 1] = fma(z0, rb[h].y, s0[2 * H + 2 * h + 1]);
                        s1[2 * H + 2 * h + 1] = fma(z1, rb[h].y, s1[2 * H + 2 * h + 1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        load_x(xa, 0);
        load_r(ra, 0, 0);
        for (int m0 = 0; m0 < M; m0 += 2 * PF) {
            load_x(xb, m0 + PF);
            __builtin_amdgcn_sched_barrier(0);
            chunk(xa, m0);
            load_x(xa, m0 + 2 * PF);
            __builtin_amdgcn_sched_barrier(0);
            chunk(xb, m0 + PF);
        }
        if (Sout) {
            if (2 * i >= row_split) {
#pragma unroll
                for (int k = 0; k < KC; k++)
                    if (k < nc_force) *reinterpret_cast<d2*>(Sout + (2 * i - row_split) + sld * (size_t)k) = (d2){s0[k], s1[k]};
            }
            if (!dist) continue;
        }
        double d0 = 0.0, d1 = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const bool on = !Sout || k < ncomp_dist;                             // (uniform; without Sout the padding zeros do it)
            const double t0 = on ? s0[k] - opad[k] : 0.0, t1 = on ? s1[k] - opad[k] : 0.0;
            d0 = fma(t0, t0, d0);
            d1 = fma(t1, t1, d1);
        }
        *reinterpret_cast<d2*>(dist + 2 * i) = (d2){sqrt(d0), sqrt(d1)};
    }
}

// 17..32 components (two 16-component tiles) on the fp64 MATRIX pipe.  v_mfma_f64_16x16x4_f64 chained through its accumulator IS the ascending fma chain
// s = fma(z_m, R[m,k], s), bit for bit (scripts/mfma_f64_probe.hip: 0 of 51200 results differ over K = 128, mixed magnitudes), so the
// scores -- and with them the distances and the ranking -- keep the bits of the vector kernels and of the oracle.  A wave takes 64
// particles as four groups of 16 and 4 metrics per step: lane (r, q) z-scores metric 4j + q of particle 16g + r (the A operand: one
// true division per lane and group, where the vector kernel divides twice per metric and lane), reads loading R[4j + q, 16t + r] from
// LDS (the B operand: one 8-byte read per lane and tile, where the vector kernel needs a broadcast 16-byte read per two fmas -- at 32
// components that LDS traffic, not the fp64 rate, bound it: 0.38 ms at N = 1e6 x 128 metrics), 4 x KT MFMAs per step.  Epilogue: the scores
// go through LDS (over the loadings, which nobody needs any more) so that every lane holds one particle's components in order
// for the distance's own fma chain.
typedef double pd4 __attribute__((ext_vector_type(4)));
// (n: an EVEN number of rows, 16-byte aligned columns: lane (r, q) loads the row pair 32 h + 2 r, + 1 of metric 4 j + q with one
// 16-byte load -- 256 contiguous bytes per 16 lanes -- and feeds group 2 h with the even row, group 2 h + 1 with the odd one;
// the loads run PF steps ahead of the step that consumes them)
template <int KT>
__global__ __launch_bounds__(256) void k_project_mfma(const double* __restrict__ X, size_t n, size_t ldx, int M,
                                                      const double* __restrict__ mean, const double* __restrict__ sd,
                                                      const double* __restrict__ model, size_t off_R, size_t off_oscore,
                                                      double* __restrict__ dist, int lds_main /* doubles in front of the observed scores */,
                                                      double* __restrict__ Sout = nullptr, size_t sld = 0, int nc_force = 0,
                                                      size_t row_split = 0) {
    constexpr int KC = 16 * KT, SROW = KC + 1, PF = 8;
    extern __shared__ double lds[];
    const int M4 = (M + 3) & ~3;
    double* const Rl = lds;                              // M4 x KC, rows beyond M zero
    double* const mu = Rl + (size_t)M4 * KC;             // M4
    double* const sg = mu + M4;                          // M4 (0: the metric takes no part -- zero variance or padding)
    double* const op = lds + lds_main;                   // KC observed scores, behind everything the epilogue overlays
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 15, q = lane >> 4;
    const int ncomp_dist = dist ? (int)model[0] : 0;
    const int ncomp = Sout ? nc_force : ncomp_dist;             // (Sout, row_split, Sout beside dist: as k_project_dist2_lds)
    for (int e = t; e < M4 * KC; e += 256) {
        const int m = e / KC, k = e % KC;
        Rl[e] = (m < M && k < ncomp) ? model[off_R + m + (size_t)M * k] : 0.0;
    }
    for (int m = t; m < M4; m += 256) { mu[m] = (m < M) ? mean[m] : 0.0; sg[m] = (m < M) ? sd[m] : 0.0; }
    if (t < KC) op[t] = (t < ncomp_dist) ? model[off_oscore + t] : 0.0;
    __syncthreads();
    const size_t base = ((size_t)blockIdx.x * 4 + wave) * 64;
    const double* xr[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        size_t row = base + 32 * h + 2 * r;
        if (row + 1 >= n) row = n - 2;                   // a legal, aligned address; the result is not stored
        xr[h] = X + row;
    }
    pd4 acc[4][KT];
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int tt = 0; tt < KT; tt++) acc[g][tt] = (pd4){0.0, 0.0, 0.0, 0.0};
    d2 xb[PF][2];
    auto loadx = [&](d2 (&x)[2], int j) {
        int m = 4 * j + q;
        if (m >= M) m = M - 1;                           // padding metric of the last step: in range, and its sd entry is 0
#pragma unroll
        for (int h = 0; h < 2; h++) x[h] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(xr[h] + (size_t)m * ldx));
    };
    const int steps = M4 / 4;
#pragma unroll
    for (int u = 0; u < PF; u++)
        if (u < steps) loadx(xb[u], u);
    for (int j0 = 0; j0 < steps; j0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; u++) {
            const int j = j0 + u;
            if (j < steps) {                             // (uniform)
                const int m = 4 * j + q;
                const double mm = mu[m], ss = sg[m];
                double b[KT];
#pragma unroll
                for (int tt = 0; tt < KT; tt++) b[tt] = Rl[(size_t)m * KC + 16 * tt + r];
                const double xv[4] = {xb[u][0].x, xb[u][0].y, xb[u][1].x, xb[u][1].y};
                if (j + PF < steps) loadx(xb[u], j + PF);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const double z = (ss == 0.0) ? 0.0 : (xv[g] - mm) / ss;
#pragma unroll
                    for (int tt = 0; tt < KT; tt++) acc[g][tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(z, b[tt], acc[g][tt], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                                     // every wave is done with the loadings: the scores go over them
    double* const stage = lds + (size_t)wave * 64 * SROW;
    // group g = 2 h + e holds rows 32 h + 2 row16 + e of the wave's 64
#pragma unroll
    for (int g = 0; g < 4; g++)
#pragma unroll
        for (int tt = 0; tt < KT; tt++)
#pragma unroll
            for (int i = 0; i < 4; i++)
                stage[(size_t)(32 * (g >> 1) + 2 * (q + 4 * i) + (g & 1)) * SROW + 16 * tt + r] = acc[g][tt][i];
    __syncthreads();
    if (Sout) {
        const size_t p = base + lane;
        if (p < n && p >= row_split)
            for (int k = 0; k < nc_force; k++) Sout[(p - row_split) + sld * (size_t)k] = stage[(size_t)lane * SROW + k];
        if (!dist) return;
    }
    double d2v = 0.0;
#pragma unroll 8
    for (int k = 0; k < KC; k++) {
        const double tt = (!Sout || k < ncomp_dist) ? stage[(size_t)lane * SROW + k] - op[k] : 0.0;
        d2v = fma(tt, tt, d2v);
    }
    const size_t p = base + lane;
    if (p < n) dist[p] = sqrt(d2v);
}

template <int KC>
__global__ __launch_bounds__(256) void k_project_dist(const double* __restrict__ X, size_t n, size_t ldx, int M,
                                                      const double* __restrict__ mean,
                                                      const double* __restrict__ sd,
                                                      const double* __restrict__ Rpad /* M x KC, row-major */,
                                                      const double* __restrict__ opad /* KC */,
                                                      double* __restrict__ dist) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double s[KC];
#pragma unroll
        for (int k = 0; k < KC; k++) s[k] = 0.0;
        const double* xp = X + i;
#pragma unroll 4
        for (int m = 0; m < M; m++) {
            const double x = xp[(size_t)m * ldx];
            const double sdm = sd[m];
            const double z = (sdm == 0.0) ? 0.0 : (x - mean[m]) / sdm;
#pragma unroll
            for (int k = 0; k < KC; k++) s[k] = fma(z, Rpad[m * KC + k], s[k]);
        }
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < KC; k++) {
            const double t = s[k] - opad[k];
            d2 = fma(t, t, d2);
        }
        dist[i] = sqrt(d2);
    }
}

// More than 32 components (the reference has no limit): the components in chunks of 32, the particle's metrics re-read and
// re-scored per chunk (same z, same fma order per component), the squared distance carried across the chunks in component order --
// the operation order of k_project_dist, so the same bits.  KCT = 32 ceil(A / 32) padded components.
__global__ __launch_bounds__(256) void k_project_dist_wide(const double* __restrict__ X, size_t n, size_t ldx, int M, int KCT,
                                                           const double* __restrict__ mean, const double* __restrict__ sd,
                                                           const double* __restrict__ Rpad /* M x KCT, row-major */,
                                                           const double* __restrict__ opad /* KCT */, double* __restrict__ dist) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double* xp = X + i;
        double d2 = 0.0;
        for (int c0 = 0; c0 < KCT; c0 += 32) {
            double s[32];
#pragma unroll
            for (int k = 0; k < 32; k++) s[k] = 0.0;
#pragma unroll 2
            for (int m = 0; m < M; m++) {
                const double x = xp[(size_t)m * ldx];
                const double sdm = sd[m];
                const double z = (sdm == 0.0) ? 0.0 : (x - mean[m]) / sdm;
#pragma unroll
                for (int k = 0; k < 32; k++) s[k] = fma(z, Rpad[(size_t)m * KCT + c0 + k], s[k]);
            }
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const double t = s[k] - opad[c0 + k];
                d2 = fma(t, t, d2);
            }
        }
        dist[i] = sqrt(d2);
    }
}

__global__ __launch_bounds__(256) void k_simple_dist(const double* __restrict__ X, size_t n, size_t ldx, int M,
                                                     const double* __restrict__ mean,
                                                     const double* __restrict__ sd,
                                                     const double* __restrict__ zobs,
                                                     double* __restrict__ dist) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double d2 = 0.0;
        const double* xp = X + i;
#pragma unroll 4
        for (int m = 0; m < M; m++) {
            const double sdm = sd[m];
            const double z = (sdm == 0.0) ? 0.0 : (xp[(size_t)m * ldx] - mean[m]) / sdm;
            const double t = z - zobs[m];
            d2 = fma(t, t, d2);
        }
        dist[i] = sqrt(d2);
    }
}

// Rpad[m*KC + k] = k < ncomp ? R[m + M*k] : 0 ; opad[k] = k < ncomp ? obs_scores[k] : 0.
// Zero padding is exact: fma(z, 0, s) == s and fma(0, 0, d2) == d2.
__global__ void k_pad_model(const double* __restrict__ model, int M, int P, int A, int KC,
                            double* __restrict__ Rpad, double* __restrict__ opad) {
    const ModelLayout ML = model_layout(M, P, A);
    const int ncomp = (int)model[ML.off_hdr];
    for (int e = threadIdx.x; e < M * KC; e += blockDim.x) {
        const int m = e / KC, k = e % KC;
        Rpad[e] = (k < ncomp) ? model[ML.off_R + m + (size_t)M * k] : 0.0;
    }
    for (int k = threadIdx.x; k < KC; k += blockDim.x) opad[k] = (k < ncomp) ? model[ML.off_oscore + k] : 0.0;
}

}  // namespace

int launch_project_distance(abc_ctx* ctx, const double* X, size_t n, size_t ldx, size_t M, size_t P, size_t A,
                            const double* model, int simple, double* dist) {
    if (n == 0) return ABC_OK;
    StageTimer tm(ctx, ST_PROJECT);
    size_t blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (simple) {
        const ModelLayout ML = model_layout(M, P, 0);
        hipLaunchKernelGGL(k_simple_dist, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X, n, ldx, (int)M,
                           model + ML.off_mean, model + ML.off_sd, model + ML.off_zobs, dist);
        ABC_HIP(ctx, hipGetLastError());
        return ABC_OK;
    }
    const ModelLayout ML = model_layout(M, P, A);
    // ncomp lives on the device; the kernel is compiled for the next power-of-two >= A and the
    // unused components are zero-padded (exact, see k_pad_model).
    int KC = 1;
    while (KC < (int)A) KC *= 2;
    if (KC > 32) KC = (int)((A + 31) / 32) * 32;
    double* Rpad = (double*)abc_ws_alloc(ctx, (M * KC + KC) * sizeof(double));
    if (!Rpad) ABC_FAIL(ctx, ABC_ERR_NOMEM, "project: workspace exhausted");
    double* opad = Rpad + M * KC;
    if (KC > 32) {
        hipLaunchKernelGGL(k_pad_model, dim3(1), dim3(256), 0, ctx->stream, model, (int)M, (int)P, (int)A, KC, Rpad, opad);
        ABC_HIP(ctx, hipGetLastError());
        hipLaunchKernelGGL(k_project_dist_wide, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X, n, ldx, (int)M, KC,
                           model + ML.off_mean, model + ML.off_sd, Rpad, opad, dist);
        ABC_HIP(ctx, hipGetLastError());
        return ABC_OK;
    }
    // fast path: row pairs with 16-B loads/stores; the odd last row (if any) goes through the scalar kernel
    const bool vec_ok = (ldx % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)dist & 15) == 0) && n >= 2;
    const size_t npairs = vec_ok ? n / 2 : 0;
    const size_t ntail = n - 2 * npairs;
    size_t pblocks = (npairs + 255) / 256;
    if (pblocks > 256 * 16) pblocks = 256 * 16;
    blocks = (ntail + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    // 17..32 components: the score contraction on the fp64 matrix pipe, while its LDS fits (aligned row pairs; an odd last row
    // goes through the scalar kernel below)
    bool mfma_kernel = false;
    size_t mfma_lb = 0, mfma_main = 0;
    static const bool valu_only = abc_diag_env("ABC_PROJECT_VALU") != nullptr;          // A/B switch for measurements
    if (KC == 32 && npairs && !valu_only) {       // (16 components: the vector kernel is 5-10 % faster, measured)
        const size_t M4 = (M + 3) & ~(size_t)3;
        mfma_main = (M4 * KC + 2 * M4 > (size_t)4 * 64 * (KC + 1)) ? M4 * KC + 2 * M4 : (size_t)4 * 64 * (KC + 1);
        mfma_lb = (mfma_main + KC) * sizeof(double);
        mfma_kernel = mfma_lb <= 150 * 1024;
    }
    if (mfma_kernel) {
        const unsigned gb = (unsigned)((2 * npairs + 255) / 256);
        ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_project_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mfma_lb));
        hipLaunchKernelGGL(k_project_mfma<2>, dim3(gb), dim3(256), mfma_lb, ctx->stream, X, 2 * npairs, ldx, (int)M, model + ML.off_mean,
                           model + ML.off_sd, model, ML.off_R, ML.off_oscore, dist, (int)mfma_main);
        ABC_HIP(ctx, hipGetLastError());
        if (ntail) {
            hipLaunchKernelGGL(k_pad_model, dim3(1), dim3(256), 0, ctx->stream, model, (int)M, (int)P, (int)A, KC, Rpad, opad);
            hipLaunchKernelGGL(k_project_dist<32>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, X + 2 * npairs, ntail, ldx, (int)M,
                                   model + ML.off_mean, model + ML.off_sd, Rpad, opad, dist + 2 * npairs);
            ABC_HIP(ctx, hipGetLastError());
        }
        return ABC_OK;
    }
    // the LDS kernel pads the loadings itself; the others read the padded copy k_pad_model leaves in the workspace
    const bool lds_kernel = (KC == 8 || KC == 16 || KC == 32) && (M * KC + KC) * sizeof(double) <= 64 * 1024 && npairs;
    if (!lds_kernel || ntail) {
        hipLaunchKernelGGL(k_pad_model, dim3(1), dim3(256), 0, ctx->stream, model, (int)M, (int)P, (int)A, KC, Rpad, opad);
        ABC_HIP(ctx, hipGetLastError());
    }
    // (16 / 32 components with the loadings in LDS; beyond 64 KB of them the scalar-operand kernel)
#define LAUNCH_PD_LDS(KCV)                                                                                             \
    do {                                                                                                               \
        const int lb = (int)((M * KCV + KCV) * sizeof(double));                                                        \
        if (npairs) {                                                                                                  \
            if (pblocks > 1024) pblocks = 1024;                                                                        \
            hipLaunchKernelGGL(k_project_dist2_lds<KCV>, dim3((unsigned)pblocks), dim3(256), lb, ctx->stream, X,       \
                               npairs, ldx, (int)M, model + ML.off_mean, model + ML.off_sd, model, ML.off_R,           \
                               ML.off_oscore, dist);                                                                   \
        }                                                                                                              \
        if (ntail)                                                                                                     \
            hipLaunchKernelGGL(k_project_dist<KCV>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,                 \
                               X + 2 * npairs, ntail, ldx, (int)M, model + ML.off_mean, model + ML.off_sd, Rpad, opad, \
                               dist + 2 * npairs);                                                                     \
    } while (0)
#define LAUNCH_PD(KCV)                                                                                                 \
    do {                                                                                                               \
        if (npairs)                                                                                                    \
            hipLaunchKernelGGL(k_project_dist2<KCV>, dim3((unsigned)pblocks), dim3(256), 0, ctx->stream, X, npairs,    \
                               ldx, (int)M, model + ML.off_mean, model + ML.off_sd, Rpad, opad, dist);                 \
        if (ntail)                                                                                                     \
            hipLaunchKernelGGL(k_project_dist<KCV>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,                 \
                               X + 2 * npairs, ntail, ldx, (int)M, model + ML.off_mean, model + ML.off_sd, Rpad, opad, \
                               dist + 2 * npairs);                                                                     \
    } while (0)
    switch (KC) {
        case 1: LAUNCH_PD(1); break;
        case 2: LAUNCH_PD(2); break;
        case 4: LAUNCH_PD(4); break;
        case 8: if (lds_kernel) LAUNCH_PD_LDS(8); else LAUNCH_PD(8); break;
        case 16: if (lds_kernel) LAUNCH_PD_LDS(16); else LAUNCH_PD(16); break;
        default: if (lds_kernel) LAUNCH_PD_LDS(32); else LAUNCH_PD(32); break;
    }
#undef LAUNCH_PD
#undef LAUNCH_PD_LDS
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}

// The validation scores of the Wilcoxon reduction (wilcoxon.hip): S[i + n k] = score k of row i of X (n rows from X on), all A components, by
// the projection kernels above (row pairs with 16-byte loads; 8 / 16 components with the loadings in LDS, 17..32 on the fp64 matrix
// pipe).  Returns the number of rows it took (an even number; 0: the shape or the alignment is not theirs) -- the caller scores the rest.
size_t launch_project_scores(abc_ctx* ctx, const double* X, size_t n, size_t ldx, size_t M, size_t P, size_t A, const double* model, double* S) {
    const ModelLayout ML = model_layout(M, P, A);
    int KC = 1;
    while (KC < (int)A) KC *= 2;
    if (KC < 8) KC = 8;
    const bool vec_ok = (ldx % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)S & 15) == 0) && n >= 2;
    if (!vec_ok || KC > 32) return 0;
    const size_t npairs = n / 2;
    if (KC == 32) {
        const size_t M4 = (M + 3) & ~(size_t)3;
        const size_t mfma_main = (M4 * KC + 2 * M4 > (size_t)4 * 64 * (KC + 1)) ? M4 * KC + 2 * M4 : (size_t)4 * 64 * (KC + 1);
        const size_t lb = (mfma_main + KC) * sizeof(double);
        if (lb > 150 * 1024) return 0;
        const unsigned gb = (unsigned)((2 * npairs + 255) / 256);
        if (hipFuncSetAttribute((const void*)k_project_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb) != hipSuccess) return 0;
        hipLaunchKernelGGL(k_project_mfma<2>, dim3(gb), dim3(256), lb, ctx->stream, X, 2 * npairs, ldx, (int)M, model + ML.off_mean, model + ML.off_sd,
                           model, ML.off_R, ML.off_oscore, (double*)nullptr, (int)mfma_main, S, n, (int)A);
        return 2 * npairs;
    }
    if ((M * KC + KC) * sizeof(double) > 64 * 1024 || (n & 1)) return 0;      // (16-byte stores into columns of n rows: an even n)
    size_t pblocks = (npairs + 255) / 256;
    if (pblocks > 1024) pblocks = 1024;
    const int lb = (int)((M * KC + KC) * sizeof(double));
    if (KC == 8)
        hipLaunchKernelGGL(k_project_dist2_lds<8>, dim3((unsigned)pblocks), dim3(256), lb, ctx->stream, X, npairs, ldx, (int)M, model + ML.off_mean,
                           model + ML.off_sd, model, ML.off_R, ML.off_oscore, (double*)nullptr, S, n, (int)A);
    else
        hipLaunchKernelGGL(k_project_dist2_lds<16>, dim3((unsigned)pblocks), dim3(256), lb, ctx->stream, X, npairs, ldx, (int)M, model + ML.off_mean,
                           model + ML.off_sd, model, ML.off_R, ML.off_oscore, (double*)nullptr, S, n, (int)A);
    return 2 * npairs;
}

// The ranking's projection AND the Wilcoxon reduction's validation scores in ONE pass over X (round 5: as two launches the
// validation half of X was read twice, and the second read competed with the selection for the chip): dist as
// launch_project_distance, S[i - row_test + sld k] = score k (all A components) of the rows from row_test on.
// 0: done; 1: not a shape for it (nothing queued: the caller launches the two separately).
// done: an event bound to the kernel's own completion signal (as launch_gather_rows' -- a hipEventRecord behind the launch is one more
// packet the selection's first kernel would wait for: ~7 us of the critical path)
int launch_project_distance_scores(abc_ctx* ctx, const double* X, size_t n, size_t ldx, size_t M, size_t P, size_t A, const double* model,
                                   double* dist, double* S, size_t sld, size_t row_test, hipEvent_t done) {
    static const bool off = abc_diag_env("ABC_PROJECT_SEPARATE") != nullptr;              // A/B switch for measurements
    const ModelLayout ML = model_layout(M, P, A);
    int KC = 1;
    while (KC < (int)A) KC *= 2;
    const bool vec_ok = (ldx % 2 == 0) && (((uintptr_t)X & 15) == 0) && (((uintptr_t)dist & 15) == 0) && (((uintptr_t)S & 15) == 0) &&
                        n >= 2 && !(n & 1) && !(row_test & 1) && !(sld & 1) && row_test < n;
    if (off || !vec_ok || (KC != 8 && KC != 16 && KC != 32)) return 1;
    StageTimer tm(ctx, ST_PROJECT);
    const size_t npairs = n / 2;
    if (KC == 32) {
        const size_t M4 = (M + 3) & ~(size_t)3;
        const size_t mfma_main = (M4 * KC + 2 * M4 > (size_t)4 * 64 * (KC + 1)) ? M4 * KC + 2 * M4 : (size_t)4 * 64 * (KC + 1);
        const size_t lb = (mfma_main + KC) * sizeof(double);
        static const bool valu_only = abc_diag_env("ABC_PROJECT_VALU") != nullptr;
        if (lb > 150 * 1024 || valu_only) return 1;
        const unsigned gb = (unsigned)((n + 255) / 256);
        ABC_HIP(ctx, hipFuncSetAttribute((const void*)k_project_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
        hipExtLaunchKernelGGL(k_project_mfma<2>, dim3(gb), dim3(256), lb, ctx->stream, nullptr, done, 0, X, n, ldx, (int)M, model + ML.off_mean,
                              model + ML.off_sd, model, ML.off_R, ML.off_oscore, dist, (int)mfma_main, S, sld, (int)A, row_test);
        ABC_HIP(ctx, hipGetLastError());
        return 0;
    }
    if ((M * KC + KC) * sizeof(double) > 64 * 1024) return 1;
    size_t pblocks = (npairs + 255) / 256;
    if (pblocks > 1024) pblocks = 1024;
    const int lb = (int)((M * KC + KC) * sizeof(double));
    if (KC == 8)
        hipExtLaunchKernelGGL(k_project_dist2_lds<8>, dim3((unsigned)pblocks), dim3(256), lb, ctx->stream, nullptr, done, 0, X, npairs, ldx, (int)M,
                              model + ML.off_mean, model + ML.off_sd, model, ML.off_R, ML.off_oscore, dist, S, sld, (int)A, row_test);
    else
        hipExtLaunchKernelGGL(k_project_dist2_lds<16>, dim3((unsigned)pblocks), dim3(256), lb, ctx->stream, nullptr, done, 0, X, npairs, ldx, (int)M,
                              model + ML.off_mean, model + ML.off_sd, model, ML.off_R, ML.off_oscore, dist, S, sld, (int)A, row_test);
    ABC_HIP(ctx, hipGetLastError());
    return 0;
}

// The distances once more from the scores the ranking's projection has left for EVERY row (round 6: a generation whose component
// count the Wilcoxon reduction lowered repeats its ranking -- the scores do not depend on the count, only how many of them the
// distance takes): dist[i] = sqrt(sum_{k < model[0]} (S[i + sld k] - obs_score[k])^2), the projection kernels' own fma chain in
// component order, hence their bits.  N x ncomp x 8 bytes instead of a second pass over X.
__global__ __launch_bounds__(256) void k_dist_from_scores(const double* __restrict__ S, size_t n, size_t sld, const double* __restrict__ model,
                                                          size_t off_oscore, double* __restrict__ dist) {
    const int nc = (int)model[0];
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; 2 * i < n; i += stride) {
        if (2 * i + 1 < n) {
            double d0 = 0.0, d1 = 0.0;
            for (int k = 0; k < nc; k++) {
                const d2 v = *reinterpret_cast<const d2*>(S + 2 * i + sld * (size_t)k);
                const double o = model[off_oscore + k], t0 = v.x - o, t1 = v.y - o;
                d0 = fma(t0, t0, d0);
                d1 = fma(t1, t1, d1);
            }
            *reinterpret_cast<d2*>(dist + 2 * i) = (d2){sqrt(d0), sqrt(d1)};
        } else {
            double d0 = 0.0;
            for (int k = 0; k < nc; k++) { const double t0 = S[2 * i + sld * (size_t)k] - model[off_oscore + k]; d0 = fma(t0, t0, d0); }
            dist[2 * i] = sqrt(d0);
        }
    }
}
// (S, dist 16-byte aligned, sld even: launch_project_distance_scores' own conditions, under which the scores exist)
int launch_distance_from_scores(abc_ctx* ctx, const double* S, size_t n, size_t sld, size_t M, size_t P, size_t A, const double* model, double* dist) {
    const ModelLayout ML = model_layout(M, P, A);
    StageTimer tm(ctx, ST_PROJECT);
    size_t blocks = (n / 2 + 256) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_dist_from_scores, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, S, n, sld, model, ML.off_oscore, dist);
    ABC_HIP(ctx, hipGetLastError());
    return ABC_OK;
}
