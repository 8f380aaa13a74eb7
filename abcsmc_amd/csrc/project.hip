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
                                                           double* __restrict__ dist) {
    extern __shared__ double Rl[];                                       // M*KC + KC
    double* const opad = Rl + (size_t)M * KC;
    {   // the zero-padded loadings and observed scores straight from the model (what k_pad_model writes for the other kernels)
        const int ncomp = (int)model[0];
        for (int e = threadIdx.x; e < M * KC; e += 256) {
            const int m = e / KC, k = e % KC;
            Rl[e] = (k < ncomp) ? model[off_R + m + (size_t)M * k] : 0.0;
        }
        if (threadIdx.x < KC) opad[threadIdx.x] = (threadIdx.x < ncomp) ? model[off_oscore + threadIdx.x] : 0.0;
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
