/*
 * abc_oracle.cpp -- CPU ORACLE (test infrastructure, see abc_oracle.h for scope/pinning).
 *
 * Every function cites the reference file:line it restates.  Paths are relative to
 * /root/reference.  "[PLS]" = tjhladish/PLS (absent submodule), "[GSL]" = GSL >= 2.2
 * (absent system library): restated from their published algorithms, see abc_oracle.h.
 *
 * Build: g++ -O2 -std=c++17 -ffp-contract=off -mfma -fPIC -shared   (oracle/Makefile)
 */
#include "abc_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <vector>

namespace {

inline double& at(double* A, size_t ld, size_t i, size_t j) { return A[i + ld * j]; }
inline double  at(const double* A, size_t ld, size_t i, size_t j) { return A[i + ld * j]; }

/* Cyclic Jacobi eigen-solve of a symmetric n x n matrix (column-major, destroyed).
 * Stands in for Eigen::EigenSolver on XY'XY in [PLS] plsr(); the matrix is symmetric PSD so
 * the real symmetric solve is exact.  Returns the unit eigenvector of the largest
 * eigenvalue; sign convention (declared, reference leaves it to Eigen): the component
 * of largest magnitude is positive. */
void dominant_eigenvector_sym(std::vector<double>& S, size_t n, double* q) {
    std::vector<double> V(n * n, 0.0);
    for (size_t i = 0; i < n; i++) V[i + n * i] = 1.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0.0, diag = 0.0;
        for (size_t j = 0; j < n; j++)
            for (size_t i = 0; i < n; i++) {
                if (i == j) diag += S[i + n * j] * S[i + n * j];
                else off += S[i + n * j] * S[i + n * j];
            }
        if (off <= 1e-32 * diag || off == 0.0) break;
        for (size_t p = 0; p + 1 < n; p++) {
            for (size_t r = p + 1; r < n; r++) {
                const double apq = S[p + n * r];
                if (apq == 0.0) continue;
                const double app = S[p + n * p], aqq = S[r + n * r];
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                for (size_t k = 0; k < n; k++) {            /* columns p, r */
                    const double akp = S[k + n * p], akq = S[k + n * r];
                    S[k + n * p] = c * akp - s * akq;
                    S[k + n * r] = s * akp + c * akq;
                }
                for (size_t k = 0; k < n; k++) {            /* rows p, r */
                    const double apk = S[p + n * k], aqk = S[r + n * k];
                    S[p + n * k] = c * apk - s * aqk;
                    S[r + n * k] = s * apk + c * aqk;
                }
                for (size_t k = 0; k < n; k++) {
                    const double vkp = V[k + n * p], vkq = V[k + n * r];
                    V[k + n * p] = c * vkp - s * vkq;
                    V[k + n * r] = s * vkp + c * vkq;
                }
            }
        }
    }
    size_t best = 0;
    for (size_t i = 1; i < n; i++) if (S[i + n * i] > S[best + n * best]) best = i;
    double nrm = 0.0; size_t big = 0;
    for (size_t i = 0; i < n; i++) {
        q[i] = V[i + n * best];
        nrm += q[i] * q[i];
        if (std::fabs(q[i]) > std::fabs(q[big])) big = i;
    }
    nrm = std::sqrt(nrm);
    const double sgn = (q[big] < 0.0) ? -1.0 : 1.0;
    for (size_t i = 0; i < n; i++) q[i] = sgn * q[i] / nrm;
}

} // namespace

extern "C" {

/* ============================ z-scores [PLS] ===================================== */
/* AbcUtil.cpp:412,432: X.colwise().mean() */
void orc_col_means(const double* X, size_t n, size_t c, double* mean) {
    for (size_t j = 0; j < c; j++) {
        double s = 0.0;
        for (size_t i = 0; i < n; i++) s += X[i + n * j];
        mean[j] = s / static_cast<double>(n);
    }
}

/* [PLS] colwise_stdev (call sites AbcUtil.cpp:413,433): centred, n-1 denominator,
 * N<2 -> 0.  n-1 pinned by tests/abcutil.cpp:11-21. */
void orc_colwise_stdev(const double* X, size_t n, size_t c, const double* mean, double* sd) {
    if (n < 2) { for (size_t j = 0; j < c; j++) sd[j] = 0.0; return; }
    const double ninv = 1.0 / static_cast<double>(n - 1);
    for (size_t j = 0; j < c; j++) {
        double s = 0.0;
        for (size_t i = 0; i < n; i++) { const double d = X[i + n * j] - mean[j]; s += d * d; }
        sd[j] = std::sqrt(s * ninv);
    }
}

/* [PLS] colwise_z_scores (AbcUtil.cpp:416,434-435). Declared deviation: sd==0 -> z=0
 * (the reference divides by zero). */
void orc_colwise_z_scores(const double* X, size_t n, size_t c, const double* mean,
                          const double* sd, double* Z) {
    for (size_t j = 0; j < c; j++)
        for (size_t i = 0; i < n; i++)
            Z[i + n * j] = (sd[j] == 0.0) ? 0.0 : (X[i + n * j] - mean[j]) / sd[j];
}

/* [PLS] z_scores (AbcUtil.cpp:414,436) */
void orc_z_scores(const double* row, size_t c, const double* mean, const double* sd, double* out) {
    for (size_t j = 0; j < c; j++) out[j] = (sd[j] == 0.0) ? 0.0 : (row[j] - mean[j]) / sd[j];
}

/* ============================ euclidean / ordered ================================= */
/* AbcUtil.cpp:320-324: (sims.rowwise() - ref).rowwise().norm(); pinned tests/abcutil.cpp:29-38.
 * Fixed order: k ascending fma chain. */
void orc_euclidean(const double* S, size_t n, size_t a, const double* ref, double* dist) {
    for (size_t i = 0; i < n; i++) {
        double d2 = 0.0;
        for (size_t k = 0; k < a; k++) { const double t = S[i + n * k] - ref[k]; d2 = std::fma(t, t, d2); }
        dist[i] = std::sqrt(d2);
    }
}

/* [PLS] ordered() (AbcUtil.cpp:420,457), same semantics as lib/ranker.h:46-53,143-145 order():
 * ascending argsort; pinned tests/pls.cpp:15-23.  Declared tie-break: (value, index). */
void orc_ordered(const double* v, size_t n, uint64_t* idx) {
    std::iota(idx, idx + n, uint64_t(0));
    std::sort(idx, idx + n, [v](uint64_t a, uint64_t b) {
        return (v[a] < v[b]) || (v[a] == v[b] && a < b);
    });
}

/* ============================ kernel PLS2 [PLS] ================================= */
/* PLS::Model ctor / plsr() (call site AbcUtil.cpp:443), SURVEY Appendix A.1.
 * X, Y are expected already z-scored by the caller (AbcUtil.cpp:432-435). */
int orc_pls_fit(const double* X, const double* Y, size_t n, size_t M, size_t P, size_t A,
                int method, double* W, double* Pm, double* Q, double* R) {
    if (A == 0 || A > M) return -1;
    std::vector<double> XY(M * P), XX;
    for (size_t j = 0; j < P; j++)
        for (size_t m = 0; m < M; m++) {
            double s = 0.0;
            for (size_t i = 0; i < n; i++) s = std::fma(X[i + n * m], Y[i + n * j], s);
            XY[m + M * j] = s;
        }
    if (method == 2) {
        XX.resize(M * M);
        for (size_t b = 0; b < M; b++)
            for (size_t a = 0; a <= b; a++) {
                double s = 0.0;
                for (size_t i = 0; i < n; i++) s = std::fma(X[i + n * a], X[i + n * b], s);
                XX[a + M * b] = XX[b + M * a] = s;
            }
    }
    std::vector<double> S(P * P), q(P), w(M), r(M), p(M), t(n), xxr(M);
    for (size_t i = 0; i < A; i++) {
        if (P == 1) {
            for (size_t m = 0; m < M; m++) w[m] = XY[m];
        } else {
            for (size_t b = 0; b < P; b++)
                for (size_t a = 0; a <= b; a++) {
                    double s = 0.0;
                    for (size_t m = 0; m < M; m++) s = std::fma(XY[m + M * a], XY[m + M * b], s);
                    S[a + P * b] = S[b + P * a] = s;
                }
            dominant_eigenvector_sym(S, P, q.data());
            for (size_t m = 0; m < M; m++) {
                double s = 0.0;
                for (size_t j = 0; j < P; j++) s = std::fma(XY[m + M * j], q[j], s);
                w[m] = s;
            }
        }
        double ww = 0.0;
        for (size_t m = 0; m < M; m++) ww = std::fma(w[m], w[m], ww);
        ww = std::sqrt(ww);
        for (size_t m = 0; m < M; m++) w[m] /= ww;
        for (size_t m = 0; m < M; m++) r[m] = w[m];
        for (size_t j = 0; j < i; j++) {
            double pw = 0.0;
            for (size_t m = 0; m < M; m++) pw = std::fma(Pm[m + M * j], w[m], pw);
            for (size_t m = 0; m < M; m++) r[m] -= pw * R[m + M * j];
        }
        double tt = 0.0;
        if (method == 2) {
            for (size_t a = 0; a < M; a++) {
                double s = 0.0;
                for (size_t b = 0; b < M; b++) s = std::fma(XX[a + M * b], r[b], s);
                xxr[a] = s;
            }
            for (size_t m = 0; m < M; m++) tt = std::fma(r[m], xxr[m], tt);
            for (size_t m = 0; m < M; m++) p[m] = xxr[m] / tt;
        } else {
            for (size_t k = 0; k < n; k++) t[k] = 0.0;
            for (size_t m = 0; m < M; m++) {
                const double rm = r[m];
                for (size_t k = 0; k < n; k++) t[k] = std::fma(X[k + n * m], rm, t[k]);
            }
            for (size_t k = 0; k < n; k++) tt = std::fma(t[k], t[k], tt);
            for (size_t m = 0; m < M; m++) {
                double s = 0.0;
                for (size_t k = 0; k < n; k++) s = std::fma(X[k + n * m], t[k], s);
                p[m] = s / tt;
            }
        }
        for (size_t j = 0; j < P; j++) {
            double s = 0.0;
            for (size_t m = 0; m < M; m++) s = std::fma(XY[m + M * j], r[m], s);
            Q[j + P * i] = s / tt;
        }
        for (size_t j = 0; j < P; j++)
            for (size_t m = 0; m < M; m++) XY[m + M * j] -= tt * (p[m] * Q[j + P * i]);
        for (size_t m = 0; m < M; m++) { W[m + M * i] = w[m]; Pm[m + M * i] = p[m]; R[m + M * i] = r[m]; }
    }
    return 0;
}

/* [PLS] Model::scores(Xnew, a) = Xnew * R[:, :a] (call sites AbcUtil.cpp:453-454) */
void orc_pls_scores(const double* Xnew, size_t n, size_t M, const double* R, size_t a, double* S) {
    for (size_t k = 0; k < a; k++)
        for (size_t i = 0; i < n; i++) {
            double s = 0.0;
            for (size_t m = 0; m < M; m++) s = std::fma(Xnew[i + n * m], R[m + M * k], s);
            S[i + n * k] = s;
        }
}

/* [PLS] cv_NEW_DATA (AbcUtil.cpp:446): residuals Y - X * R_a Q_a' for a = 1..A, reduced to
 * PRESS_j(a) = sum_i e_ij^2.  press is A x P column-major. */
void orc_pls_press(const double* Xt, const double* Yt, size_t nt, size_t M, size_t P, size_t A,
                   const double* R, const double* Q, double* press) {
    std::vector<double> S(nt * A);
    orc_pls_scores(Xt, nt, M, R, A, S.data());
    std::vector<double> pred(nt);
    for (size_t j = 0; j < P; j++) {
        std::fill(pred.begin(), pred.end(), 0.0);
        for (size_t a = 0; a < A; a++) {
            const double qja = Q[j + P * a];
            double s = 0.0;
            for (size_t i = 0; i < nt; i++) {
                pred[i] = std::fma(S[i + nt * a], qja, pred[i]);
                const double e = Yt[i + nt * j] - pred[i];
                s = std::fma(e, e, s);
            }
            press[a + A * j] = s;
        }
    }
}

/* [PLS] normalcdf: 4-term polynomial approximation (Abramowitz & Stegun 26.2.18) */
double orc_normalcdf(double z) {
    const double c1 = 0.196854, c2 = 0.115194, c3 = 0.000344, c4 = 0.019527;
    const double x = std::fabs(z);
    const double d = 1.0 + c1 * x + c2 * x * x + c3 * x * x * x + c4 * x * x * x * x;
    const double tail = 0.5 / (d * d * d * d);
    return (z >= 0.0) ? 1.0 - tail : tail;
}

/* [PLS] wilcoxon(): two-sided Wilcoxon signed-rank test of |e1| vs |e2|, normal
 * approximation, "average" tie ranks as lib/ranker.h:66-76.  Zero differences dropped. */
double orc_wilcoxon_p(const double* e1, const double* e2, size_t n) {
    std::vector<double> ad; std::vector<int> sg;
    ad.reserve(n); sg.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const double d = std::fabs(e1[i]) - std::fabs(e2[i]);
        if (d == 0.0) continue;
        ad.push_back(std::fabs(d)); sg.push_back(d > 0.0 ? 1 : -1);
    }
    const size_t m = ad.size();
    if (m == 0) return 1.0;
    std::vector<uint64_t> ord(m);
    orc_ordered(ad.data(), m, ord.data());
    double W = 0.0;
    for (size_t c = 0, reps; c < m; c += reps) {
        reps = 1;
        while (c + reps < m && ad[ord[c]] == ad[ord[c + reps]]) ++reps;
        const double rk = static_cast<double>(2 * c + reps - 1) / 2.0 + 1.0;   /* ranker.h:74-75 */
        for (size_t k = 0; k < reps; k++) W += sg[ord[c + k]] * rk;
    }
    const double dm = static_cast<double>(m);
    const double sigma = std::sqrt(dm * (dm + 1.0) * (2.0 * dm + 1.0) / 6.0);
    const double z = W / sigma;
    return 2.0 * (1.0 - orc_normalcdf(std::fabs(z)));
}

/* [PLS] optimal_num_components (AbcUtil.cpp:447-449): argmin PRESS per response (first
 * minimum), optionally reduced to the smallest a' whose |errors| are not Wilcoxon-different
 * (alpha = 0.1).  Returns max over responses (the caller's .maxCoeff()). */
int orc_pls_optimal_components(const double* Xt, const double* Yt, size_t nt, size_t M, size_t P,
                               size_t A, const double* R, const double* Q, int rule,
                               int32_t* per_response) {
    std::vector<double> press(A * P);
    orc_pls_press(Xt, Yt, nt, M, P, A, R, Q, press.data());
    std::vector<double> S;
    if (rule == ORC_RULE_WILCOXON) { S.resize(nt * A); orc_pls_scores(Xt, nt, M, R, A, S.data()); }
    auto errors = [&](size_t a_idx, size_t j, std::vector<double>& e) {   /* a_idx = a-1 */
        e.assign(nt, 0.0);
        std::vector<double> pred(nt, 0.0);
        for (size_t a = 0; a <= a_idx; a++) {
            const double qja = Q[j + P * a];
            for (size_t i = 0; i < nt; i++) pred[i] = std::fma(S[i + nt * a], qja, pred[i]);
        }
        for (size_t i = 0; i < nt; i++) e[i] = Yt[i + nt * j] - pred[i];
    };
    int best_max = 0;
    std::vector<double> e1, e2;
    for (size_t j = 0; j < P; j++) {
        size_t mi = 0;
        for (size_t a = 1; a < A; a++) if (press[a + A * j] < press[mi + A * j]) mi = a;
        int best = static_cast<int>(mi) + 1;
        if (rule == ORC_RULE_WILCOXON && mi > 0) {
            errors(mi, j, e1);
            for (size_t a = 0; a < mi; a++) {
                errors(a, j, e2);
                if (orc_wilcoxon_p(e1.data(), e2.data(), nt) > 0.1) { best = static_cast<int>(a) + 1; break; }
            }
        }
        if (per_response) per_response[j] = best;
        best_max = std::max(best_max, best);
    }
    return best_max;
}

/* Staged projection + distance: z-score on the fly, scores via an m-ascending fma chain per
 * component, distance via a k-ascending fma chain, sqrt.  Restates AbcUtil.cpp:434,453-455
 * with the summation order FIXED so the HIP kernel can match bit for bit. */
void orc_project_distance(const double* X, size_t n, size_t M, const double* mean, const double* sd,
                          const double* R, size_t a, const double* obs_scores, double* dist) {
    std::vector<double> s(a);
    for (size_t i = 0; i < n; i++) {
        for (size_t k = 0; k < a; k++) s[k] = 0.0;
        for (size_t m = 0; m < M; m++) {
            const double z = (sd[m] == 0.0) ? 0.0 : (X[i + n * m] - mean[m]) / sd[m];
            for (size_t k = 0; k < a; k++) s[k] = std::fma(z, R[m + M * k], s[k]);
        }
        double d2 = 0.0;
        for (size_t k = 0; k < a; k++) { const double t = s[k] - obs_scores[k]; d2 = std::fma(t, t, d2); }
        dist[i] = std::sqrt(d2);
    }
}

/* ============================ particle rankings ================================= */
/* AbcUtil.cpp:423-458 */
int orc_particle_ranking_pls(const double* X, const double* Y, const double* obs,
                             size_t N, size_t M, size_t P, double train_frac, int max_comp,
                             int rule, uint64_t* idx, double* dist, int32_t* ncomp_out,
                             double* R_out, double* Q_out, double* mean_out, double* sd_out,
                             double* press_out) {
    if (!(0.0 < train_frac && train_frac <= 1.0)) return -1;           /* :428 */
    const size_t A = (max_comp > 0) ? static_cast<size_t>(max_comp) : std::min(M, P);
    if (A > M) return -1;
    std::vector<double> mean(M), sd(M), ymean(P), ysd(P), zobs(M);
    orc_col_means(X, N, M, mean.data());                               /* :432 */
    orc_colwise_stdev(X, N, M, mean.data(), sd.data());                /* :433 */
    std::vector<double> zX(N * M), zY(N * P);
    orc_colwise_z_scores(X, N, M, mean.data(), sd.data(), zX.data()); /* :434 */
    orc_col_means(Y, N, P, ymean.data());
    orc_colwise_stdev(Y, N, P, ymean.data(), ysd.data());
    orc_colwise_z_scores(Y, N, P, ymean.data(), ysd.data(), zY.data()); /* :435 */
    orc_z_scores(obs, M, mean.data(), sd.data(), zobs.data());         /* :436 */

    const size_t ntrain = static_cast<size_t>(std::round(static_cast<double>(N) * train_frac)); /* :438 */
    const size_t ntest = N - ntrain;                                    /* :445 */
    /* topRows / bottomRows copies (:443, :446) */
    std::vector<double> Xtr(ntrain * M), Ytr(ntrain * P), Xte(ntest * M), Yte(ntest * P);
    for (size_t m = 0; m < M; m++) {
        std::copy(zX.begin() + N * m, zX.begin() + N * m + ntrain, Xtr.begin() + ntrain * m);
        std::copy(zX.begin() + N * m + ntrain, zX.begin() + N * (m + 1), Xte.begin() + ntest * m);
    }
    for (size_t j = 0; j < P; j++) {
        std::copy(zY.begin() + N * j, zY.begin() + N * j + ntrain, Ytr.begin() + ntrain * j);
        std::copy(zY.begin() + N * j + ntrain, zY.begin() + N * (j + 1), Yte.begin() + ntest * j);
    }
    std::vector<double> W(M * A), Pm(M * A), Q(P * A), R(M * A);
    if (orc_pls_fit(Xtr.data(), Ytr.data(), ntrain, M, P, A, 1, W.data(), Pm.data(), Q.data(), R.data())) return -2;
    if (press_out) orc_pls_press(Xte.data(), Yte.data(), ntest, M, P, A, R.data(), Q.data(), press_out);
    const int ncomp = orc_pls_optimal_components(Xte.data(), Yte.data(), ntest, M, P, A,
                                                 R.data(), Q.data(), rule, nullptr); /* :447-449 */
    /* obs_scores (:453), sim_scores + euclidean (:454-455) in the fixed order */
    std::vector<double> obs_scores(ncomp);
    for (int k = 0; k < ncomp; k++) {
        double s = 0.0;
        for (size_t m = 0; m < M; m++) s = std::fma(zobs[m], R[m + M * k], s);
        obs_scores[k] = s;
    }
    std::vector<double> dloc;
    double* d = dist;
    if (!d) { dloc.resize(N); d = dloc.data(); }
    orc_project_distance(X, N, M, mean.data(), sd.data(), R.data(), ncomp, obs_scores.data(), d);
    if (idx) orc_ordered(d, N, idx);                                    /* :457 */
    if (ncomp_out) *ncomp_out = ncomp;
    if (R_out) std::memcpy(R_out, R.data(), sizeof(double) * M * A);
    if (Q_out) std::memcpy(Q_out, Q.data(), sizeof(double) * P * A);
    if (mean_out) std::memcpy(mean_out, mean.data(), sizeof(double) * M);
    if (sd_out) std::memcpy(sd_out, sd.data(), sizeof(double) * M);
    return 0;
}

/* AbcUtil.cpp:408-421 */
int orc_particle_ranking_simple(const double* X, const double* obs, size_t N, size_t M,
                                uint64_t* idx, double* dist) {
    std::vector<double> mean(M), sd(M), zobs(M);
    orc_col_means(X, N, M, mean.data());
    orc_colwise_stdev(X, N, M, mean.data(), sd.data());
    orc_z_scores(obs, M, mean.data(), sd.data(), zobs.data());
    std::vector<double> dloc;
    double* d = dist;
    if (!d) { dloc.resize(N); d = dloc.data(); }
    for (size_t i = 0; i < N; i++) {
        double d2 = 0.0;
        for (size_t m = 0; m < M; m++) {
            const double z = (sd[m] == 0.0) ? 0.0 : (X[i + N * m] - mean[m]) / sd[m];
            const double t = z - zobs[m];
            d2 = std::fma(t, t, d2);
        }
        d[i] = std::sqrt(d2);
    }
    if (idx) orc_ordered(d, N, idx);
    return 0;
}

/* ============================ variances, priors, weights ======================== */
/* AbcUtil.cpp:528-537 with RunningStat.h:16-46 (Welford, n-1) */
void orc_doubled_variance(const double* theta, size_t K, size_t P, double* dv) {
    for (size_t p = 0; p < P; p++) {
        int n = 0; double oldM = 0, newM = 0, oldS = 0, newS = 0;
        for (size_t i = 0; i < K; i++) {
            const double x = theta[i + K * p];
            n++;
            if (n == 1) { oldM = newM = x; oldS = 0.0; }
            else {
                newM = oldM + (x - oldM) / n;
                newS = oldS + (x - oldM) * (x - newM);
                oldM = newM; oldS = newS;
            }
        }
        dv[p] = 2.0 * ((n > 1) ? newS / (n - 1) : 0.0);
    }
}

/* Priors.h:54-56 (Gaussian), :76-78 (DiscreteUniform), :102-104 (ContinuousUniform) */
double orc_prior_likelihood(const orc_prior_t* pr, double v) {
    switch (pr->kind) {
        case ORC_PRIOR_GAUSS: return orc_ran_gaussian_pdf(v - pr->a, pr->b);
        case ORC_PRIOR_UNIF_INT:
            return ((v == std::round(v)) && (pr->a <= v) && (v <= pr->b)) ? 1.0 / (pr->b - pr->a + 1.0) : 0.0;
        default: return ((pr->a <= v) && (v <= pr->b)) ? 1.0 / (pr->b - pr->a) : 0.0;
    }
}
/* Priors.h:58, :80, :106 */
double orc_prior_recast(const orc_prior_t* pr, double v) {
    return (pr->kind == ORC_PRIOR_UNIF_INT) ? std::round(v) : v;
}
/* Parameter.h:77 */
int orc_prior_valid(const orc_prior_t* pr, double v) { return orc_prior_likelihood(pr, v) != 0.0; }
/* Priors.h:35 with ctor args :49, :66, :90-92 */
double orc_prior_mean(const orc_prior_t* pr) {
    return (pr->kind == ORC_PRIOR_GAUSS) ? pr->a : (pr->b + pr->a) / 2.0;
}

/* AbcUtil.cpp:539-545 */
void orc_weights_uniform(size_t K, double* w) {
    const double u = 1.0 / static_cast<double>(K);
    for (size_t i = 0; i < K; i++) w[i] = u;
}

/* AbcUtil.cpp:547-586 (loop structure and per-factor pdf calls kept as in the reference) */
void orc_weights_importance(const orc_prior_t* priors, const double* theta, size_t K,
                            const double* theta_prev, size_t Kp, const double* w_prev,
                            const double* dv_prev, size_t P, int zero_dv_policy, double* w) {
    for (size_t i = 0; i < K; i++) {
        double numerator = 1.0, denominator = 0.0;
        for (size_t p = 0; p < P; p++) numerator *= orc_prior_likelihood(&priors[p], theta[i + K * p]);
        for (size_t j = 0; j < Kp; j++) {
            double running = w_prev[j];
            for (size_t p = 0; p < P; p++) {
                const double v = theta[i + K * p], ov = theta_prev[j + Kp * p], odv = dv_prev[p];
                if (odv != 0.0 || v != ov) {                                 /* :573 */
                    if (odv == 0.0 && zero_dv_policy == 0) running *= 0.0;   /* declared deviation */
                    else running *= orc_ran_gaussian_pdf(v - ov, std::sqrt(odv)); /* :574 */
                }
            }
            denominator += running;
        }
        w[i] = numerator / denominator;                                      /* :580 */
    }
    /* :583 weight.normalize(): divide by the L2 norm (Eigen: if squaredNorm > 0) */
    double sq = 0.0;
    for (size_t i = 0; i < K; i++) sq += w[i] * w[i];
    if (sq > 0.0) { const double nrm = std::sqrt(sq); for (size_t i = 0; i < K; i++) w[i] /= nrm; }
}

/* ============================ MVN setup ========================================= */
/* AbcUtil.cpp:462-488.  [GSL] gsl_ran_multivariate_gaussian_vcov (gsl_stats mean/covariance
 * recurrences with long double accumulators, n-1), diagonal doubled (:475-479), then
 * [GSL] gsl_linalg_cholesky_decomp1 (left-looking level-2 form, scale by 1/sqrt(ajj)). */
int orc_mvn_setup(const double* theta, size_t K, size_t P, double* L, double* cov_out) {
    std::vector<double> mean(P);
    for (size_t p = 0; p < P; p++) {
        long double m = 0;
        for (size_t i = 0; i < K; i++) m += (theta[i + K * p] - m) / (i + 1);
        mean[p] = static_cast<double>(m);
    }
    for (size_t a = 0; a < P; a++)
        for (size_t b = a; b < P; b++) {
            long double cov = 0;
            for (size_t i = 0; i < K; i++) {
                const long double d1 = theta[i + K * a] - mean[a];
                const long double d2 = theta[i + K * b] - mean[b];
                cov += (d1 * d2 - cov) / (i + 1);
            }
            const double c = static_cast<double>(cov) * (static_cast<double>(K) / static_cast<double>(K - 1));
            L[a + P * b] = L[b + P * a] = c;
        }
    for (size_t p = 0; p < P; p++) L[p + P * p] = 2.0 * L[p + P * p];
    if (cov_out) std::memcpy(cov_out, L, sizeof(double) * P * P);
    for (size_t j = 0; j < P; j++) {
        /* v = A(j:n, j) -= A(j:n, 0:j) * A(j, 0:j)' */
        for (size_t i = j; i < P; i++) {
            double temp = 0.0;
            for (size_t k = 0; k < j; k++) temp += L[j + P * k] * L[i + P * k];
            L[i + P * j] += -1.0 * temp;
        }
        double ajj = L[j + P * j];
        if (!(ajj > 0.0)) return -1;                /* GSL_EDOM: reference aborts here */
        ajj = std::sqrt(ajj);
        const double inv = 1.0 / ajj;
        for (size_t i = j; i < P; i++) L[i + P * j] *= inv;
    }
    return 0;
}

/* ============================ GSL RNG restatements [GSL] ========================== */
/* rng/taus.c: taus2_set / taus_get / taus_get_double (examples/include/examples.h:10) */
static inline uint32_t taus_step(uint32_t s, int a, int b, uint32_t c, int d) {
    return ((s & c) << d) ^ (((s << a) ^ s) >> b);
}
uint32_t orc_rng_get(orc_rng_t* r) {
    r->s1 = taus_step(r->s1, 13, 19, 4294967294u, 12);
    r->s2 = taus_step(r->s2, 2, 25, 4294967288u, 4);
    r->s3 = taus_step(r->s3, 3, 11, 4294967280u, 17);
    return r->s1 ^ r->s2 ^ r->s3;
}
void orc_rng_set(orc_rng_t* r, unsigned long seed) {
    uint32_t s = static_cast<uint32_t>(seed & 0xffffffffUL);
    if (s == 0) s = 1;
    r->s1 = 69069u * s;     if (r->s1 < 2)  r->s1 += 2;
    r->s2 = 69069u * r->s1; if (r->s2 < 8)  r->s2 += 8;
    r->s3 = 69069u * r->s2; if (r->s3 < 16) r->s3 += 16;
    for (int i = 0; i < 6; i++) orc_rng_get(r);
}
double orc_rng_uniform(orc_rng_t* r) { return orc_rng_get(r) / 4294967296.0; }
double orc_rng_uniform_pos(orc_rng_t* r) {
    double x;
    do { x = orc_rng_uniform(r); } while (x == 0.0);
    return x;
}
/* rng/rng.c gsl_rng_uniform_int: scale = range / n; reject k >= n (Priors.h:73, AbcSmc.cpp:859) */
unsigned long orc_rng_uniform_int(orc_rng_t* r, unsigned long n) {
    const unsigned long range = 0xffffffffUL;
    const unsigned long scale = range / n;
    unsigned long k;
    do { k = orc_rng_get(r) / scale; } while (k >= n);
    return k;
}
/* randist/gauss.c gsl_ran_gaussian: polar Box-Muller, second variate discarded (Priors.h:41) */
double orc_ran_gaussian(orc_rng_t* r, double sigma) {
    double x, y, r2;
    do {
        x = -1.0 + 2.0 * orc_rng_uniform_pos(r);
        y = -1.0 + 2.0 * orc_rng_uniform_pos(r);
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    return sigma * y * std::sqrt(-2.0 * std::log(r2) / r2);
}
/* randist/gauss.c gsl_ran_gaussian_pdf (AbcUtil.cpp:574, Priors.h:55) */
double orc_ran_gaussian_pdf(double x, double sigma) {
    const double u = x / std::fabs(sigma);
    return (1.0 / (std::sqrt(2.0 * M_PI) * std::fabs(sigma))) * std::exp(-u * u / 2.0);
}

/* randist/discrete.c gsl_ran_discrete_preproc (Walker alias, LIFO stacks, KNUTH_CONVENTION)
 * (AbcUtil.cpp:115) */
void orc_discrete_preproc(size_t K, const double* w, double* F, uint64_t* A) {
    double total = 0.0;
    for (size_t k = 0; k < K; k++) total += w[k];
    std::vector<double> E(K);
    for (size_t k = 0; k < K; k++) E[k] = w[k] / total;
    const double mean = 1.0 / static_cast<double>(K);
    std::vector<size_t> bigs, smalls;
    for (size_t k = 0; k < K; k++) { if (E[k] < mean) smalls.push_back(k); else bigs.push_back(k); }
    while (!smalls.empty()) {
        const size_t s = smalls.back(); smalls.pop_back();
        if (bigs.empty()) { A[s] = s; F[s] = 1.0; continue; }
        const size_t b = bigs.back(); bigs.pop_back();
        A[s] = b;
        F[s] = static_cast<double>(K) * E[s];
        const double d = mean - E[s];
        E[s] += d;
        E[b] -= d;
        if (E[b] < mean) smalls.push_back(b);
        else if (E[b] > mean) bigs.push_back(b);
        else { A[b] = b; F[b] = 1.0; }
    }
    while (!bigs.empty()) { const size_t b = bigs.back(); bigs.pop_back(); A[b] = b; F[b] = 1.0; }
    for (size_t k = 0; k < K; k++) { F[k] += static_cast<double>(k); F[k] /= static_cast<double>(K); }
}
/* randist/discrete.c gsl_ran_discrete (KNUTH_CONVENTION): exactly one uniform per draw
 * (AbcUtil.cpp:117) */
uint64_t orc_discrete_draw(orc_rng_t* r, size_t K, const double* F, const uint64_t* A) {
    const double u = orc_rng_uniform(r);
    const size_t c = static_cast<size_t>(u * static_cast<double>(K));
    const double f = F[c];
    if (f == 1.0) return c;
    return (u < f) ? c : A[c];
}

/* Optional Epanechnikov weight kernel (BASELINE.json north_star names one; the reference has none -- only a comment at
 * AbcUtil.cpp:476 -- so this is an EXTENSION, off by default, with no reference semantics to match): the product of
 * Gaussian factors of AbcUtil.cpp:572-576 is replaced by the radial Epanechnikov kernel of the same covariance,
 *   K(r2) = max(0, 1 - r2 / (P' + 4)),   r2 = sum_p (theta_ip - theta'_jp)^2 / dv'_p   over the P' parameters with dv'_p != 0
 * (a d-variate Epanechnikov kernel with support radius h has covariance h^2 I / (d + 4)); the constant in front cancels in
 * the L2 normalisation and is left out; a particle no previous particle supports gets weight 0. */
void orc_weights_epanechnikov(const orc_prior_t* priors, const double* theta, size_t K,
                              const double* theta_prev, size_t Kp, const double* w_prev,
                              const double* dv_prev, size_t P, double* w) {
    size_t pnz = 0;
    for (size_t p = 0; p < P; p++) pnz += dv_prev[p] != 0.0;
    const double h2 = static_cast<double>(pnz) + 4.0;
    for (size_t i = 0; i < K; i++) {
        double numerator = 1.0, denominator = 0.0;
        for (size_t p = 0; p < P; p++) numerator *= orc_prior_likelihood(&priors[p], theta[i + K * p]);
        for (size_t j = 0; j < Kp; j++) {
            double r2 = 0.0;
            for (size_t p = 0; p < P; p++) {
                if (dv_prev[p] == 0.0) continue;
                const double d = theta[i + K * p] - theta_prev[j + Kp * p];
                r2 += d * d / dv_prev[p];
            }
            const double k = 1.0 - r2 / h2;
            if (k > 0.0) denominator += w_prev[j] * k;
        }
        w[i] = denominator > 0.0 ? numerator / denominator : 0.0;
    }
    double sq = 0.0;
    for (size_t i = 0; i < K; i++) sq += w[i] * w[i];
    if (sq > 0.0) { const double nrm = std::sqrt(sq); for (size_t i = 0; i < K; i++) w[i] /= nrm; }
}

/* ============================ resample + perturb ================================== */
/* AbcUtil.cpp:111-120 */
void orc_resample(orc_rng_t* r, const double* w, size_t K, size_t n, uint64_t* idx) {
    std::vector<double> F(K); std::vector<uint64_t> A(K);
    orc_discrete_preproc(K, w, F.data(), A.data());
    for (size_t i = 0; i < n; i++) idx[i] = orc_discrete_draw(r, K, F.data(), A.data());
}

/* AbcUtil.cpp:377-389 + :145-158 + Priors.h:19-43 (INDEPENDENT noise, <=1000 tries, then prior mean) */
size_t orc_sample_predictive_priors(orc_rng_t* r, size_t n, const double* w, const double* theta,
                                    size_t K, size_t P, const orc_prior_t* priors,
                                    const double* dv, double* out, uint64_t* parent_idx) {
    std::vector<uint64_t> par(n);
    orc_resample(r, w, K, n, par.data());                    /* sample_posterior, :383 */
    std::vector<double> sigma(P);
    for (size_t p = 0; p < P; p++) sigma[p] = std::sqrt(dv[p]);   /* :150 */
    size_t fallbacks = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t p = 0; p < P; p++) {
            const double mu = theta[par[i] + K * p];
            size_t attempts = 1;
            double dev = orc_prior_recast(&priors[p], orc_ran_gaussian(r, sigma[p]) + mu);
            while (!orc_prior_valid(&priors[p], dev) && (attempts++ < 1000))
                dev = orc_prior_recast(&priors[p], orc_ran_gaussian(r, sigma[p]) + mu);
            if (!orc_prior_valid(&priors[p], dev)) { dev = orc_prior_mean(&priors[p]); fallbacks++; }
            out[i + n * p] = dev;
        }
    }
    if (parent_idx) std::memcpy(parent_idx, par.data(), n * sizeof(uint64_t));
    return fallbacks;
}

/* AbcUtil.cpp:391-404 + :122-143; [GSL] gsl_ran_multivariate_gaussian: z_i = ugaussian in
 * order, dtrmv(Lower,NoTrans,NonUnit) computed i = P-1..0, then + mu. */
size_t orc_sample_mvn_predictive_priors(orc_rng_t* r, size_t n, const double* w, const double* theta,
                                        size_t K, size_t P, const orc_prior_t* priors,
                                        const double* L, size_t max_tries, double* out,
                                        uint64_t* parent_idx) {
    std::vector<uint64_t> par(n);
    orc_resample(r, w, K, n, par.data());                    /* sample_posterior, :398 */
    std::vector<double> x(P), vals(P);
    size_t rejected = 0;
    for (size_t i = 0; i < n; i++) {
        bool success = false; size_t tries = 0;
        while (!success) {
            success = true;
            for (size_t p = 0; p < P; p++) x[p] = orc_ran_gaussian(r, 1.0);
            for (size_t a = P; a > 0 && a--;) {
                double temp = 0.0;
                for (size_t b = 0; b < a; b++) temp += x[b] * L[a + P * b];
                x[a] = temp + x[a] * L[a + P * a];
            }
            for (size_t p = 0; p < P; p++) x[p] += theta[par[i] + K * p];
            for (size_t p = 0; success && p < P; p++) {
                vals[p] = orc_prior_recast(&priors[p], x[p]);
                success = orc_prior_valid(&priors[p], vals[p]);
            }
            if (!success) { rejected++; if (max_tries && ++tries >= max_tries) break; }
        }
        for (size_t p = 0; p < P; p++) out[i + n * p] = vals[p];
    }
    if (parent_idx) std::memcpy(parent_idx, par.data(), n * sizeof(uint64_t));
    return rejected;
}

/* ============================ one generation ====================================== */
/* AbcSmc.cpp:634-664 (rank, truncate to K), :1041-1066 (dv, weights), :490-518 (proposals),
 * :535 (one gsl_rng_get seed per new particle, after all proposals). */
int orc_generation(const orc_generation_cfg_t* cfg, const double* X, const double* Y,
                   const double* obs, const orc_prior_t* priors, const double* theta_prev,
                   const double* w_prev, const double* dv_prev, orc_rng_t* rng,
                   uint64_t* idx, double* w, double* dv, double* L, double* next,
                   uint64_t* parent, uint64_t* seeds, int32_t* ncomp) {
    const size_t N = cfg->N, P = cfg->P, K = cfg->K;
    std::vector<uint64_t> full(N);
    int rc = orc_particle_ranking_pls(X, Y, obs, N, cfg->M, P, cfg->train_frac, cfg->max_comp, cfg->rule,
                                      full.data(), nullptr, ncomp, nullptr, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    std::memcpy(idx, full.data(), K * sizeof(uint64_t));       /* AbcSmc.cpp:645-646 */
    std::vector<double> theta(K * P);
    for (size_t p = 0; p < P; p++)
        for (size_t i = 0; i < K; i++) theta[i + K * p] = Y[idx[i] + N * p];
    orc_doubled_variance(theta.data(), K, P, dv);
    if (!theta_prev) orc_weights_uniform(K, w);
    else orc_weights_importance(priors, theta.data(), K, theta_prev, cfg->Kp, w_prev, dv_prev, P,
                                cfg->zero_dv_policy, w);
    if (cfg->Nnext == 0) return 0;
    if (cfg->multivariate) {
        std::vector<double> Lloc;
        double* Lp = L; if (!Lp) { Lloc.resize(P * P); Lp = Lloc.data(); }
        if (orc_mvn_setup(theta.data(), K, P, Lp, nullptr)) return -3;
        orc_sample_mvn_predictive_priors(rng, cfg->Nnext, w, theta.data(), K, P, priors, Lp, 0, next, parent);
    } else {
        orc_sample_predictive_priors(rng, cfg->Nnext, w, theta.data(), K, P, priors, dv, next, parent);
    }
    if (seeds) for (size_t i = 0; i < cfg->Nnext; i++) seeds[i] = orc_rng_get(rng);
    return 0;
}

} // extern "C"
