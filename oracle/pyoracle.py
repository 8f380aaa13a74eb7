"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under abcsmc_amd/ may import this module.

All matrices are numpy float64, column-major (Fortran order), as the reference's Eigen Mat2D.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PRIOR_GAUSS, PRIOR_UNIF_INT, PRIOR_UNIF_REAL = 0, 1, 2
RULE_MIN_PRESS, RULE_WILCOXON = 0, 1
RULE_DEFAULT = RULE_WILCOXON      # what the drop-in surface defaults to: the checker's default follows it


class Prior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


class Rng(C.Structure):
    _fields_ = [("s1", C.c_uint32), ("s2", C.c_uint32), ("s3", C.c_uint32)]


class GenCfg(C.Structure):
    _fields_ = [("N", C.c_size_t), ("M", C.c_size_t), ("P", C.c_size_t),
                ("K", C.c_size_t), ("Kp", C.c_size_t), ("Nnext", C.c_size_t),
                ("train_frac", C.c_double),
                ("max_comp", C.c_int), ("rule", C.c_int), ("multivariate", C.c_int),
                ("zero_dv_policy", C.c_int)]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = [os.path.join(_HERE, f) for f in ("abc_oracle.cpp", "abc_oracle.h")]
    stale = (not os.path.exists(so)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_rng_get.restype = C.c_uint32
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_uniform_pos.restype = C.c_double
        L.orc_rng_uniform_int.restype = C.c_ulong
        L.orc_rng_uniform_int.argtypes = [C.c_void_p, C.c_ulong]
        L.orc_rng_set.argtypes = [C.c_void_p, C.c_ulong]
        L.orc_ran_gaussian.restype = C.c_double
        L.orc_ran_gaussian.argtypes = [C.c_void_p, C.c_double]
        L.orc_ran_gaussian_pdf.restype = C.c_double
        L.orc_ran_gaussian_pdf.argtypes = [C.c_double, C.c_double]
        L.orc_normalcdf.restype = C.c_double
        L.orc_normalcdf.argtypes = [C.c_double]
        L.orc_wilcoxon_p.restype = C.c_double
        L.orc_prior_likelihood.restype = C.c_double
        L.orc_prior_likelihood.argtypes = [C.c_void_p, C.c_double]
        L.orc_prior_recast.restype = C.c_double
        L.orc_prior_recast.argtypes = [C.c_void_p, C.c_double]
        L.orc_prior_valid.argtypes = [C.c_void_p, C.c_double]
        L.orc_prior_mean.restype = C.c_double
        L.orc_discrete_draw.restype = C.c_uint64
        L.orc_sample_predictive_priors.restype = C.c_size_t
        L.orc_sample_mvn_predictive_priors.restype = C.c_size_t
    return _LIB


def _f(a):
    """float64 column-major contiguous view/copy."""
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _sz(x):
    return C.c_size_t(int(x))


def make_priors(spec):
    """spec: list of (kind, a, b) -> ctypes array of Prior."""
    arr = (Prior * len(spec))()
    for i, (k, a, b) in enumerate(spec):
        arr[i].kind, arr[i].a, arr[i].b = int(k), float(a), float(b)
    return arr


# ---- z-scores -------------------------------------------------------------------------
def col_means(X):
    X = _f(X); n, c = X.shape
    out = np.empty(c)
    lib().orc_col_means(_p(X), _sz(n), _sz(c), _p(out))
    return out


def colwise_stdev(X, mean):
    X = _f(X); n, c = X.shape
    mean = _f(mean); out = np.empty(c)
    lib().orc_colwise_stdev(_p(X), _sz(n), _sz(c), _p(mean), _p(out))
    return out


def colwise_z_scores(X, mean=None, sd=None):
    X = _f(X); n, c = X.shape
    if mean is None:
        mean = col_means(X)
    if sd is None:
        sd = colwise_stdev(X, mean)
    mean, sd = _f(mean), _f(sd)
    Z = np.empty((n, c), order="F")
    lib().orc_colwise_z_scores(_p(X), _sz(n), _sz(c), _p(mean), _p(sd), _p(Z))
    return Z


def euclidean(S, ref):
    S = _f(S); n, a = S.shape
    ref = _f(ref); out = np.empty(n)
    lib().orc_euclidean(_p(S), _sz(n), _sz(a), _p(ref), _p(out))
    return out


def ordered(v):
    v = _f(v); out = np.empty(v.size, dtype=np.uint64)
    lib().orc_ordered(_p(v), _sz(v.size), _p(out))
    return out


# ---- PLS ------------------------------------------------------------------------------
def pls_fit(X, Y, A, method=1):
    X, Y = _f(X), _f(Y)
    n, M = X.shape; P = Y.shape[1]
    W = np.empty((M, A), order="F"); Pm = np.empty((M, A), order="F")
    Q = np.empty((P, A), order="F"); R = np.empty((M, A), order="F")
    rc = lib().orc_pls_fit(_p(X), _p(Y), _sz(n), _sz(M), _sz(P), _sz(A), C.c_int(method),
                           _p(W), _p(Pm), _p(Q), _p(R))
    if rc:
        raise ValueError("orc_pls_fit rc=%d" % rc)
    return W, Pm, Q, R


def pls_press(Xt, Yt, R, Q):
    Xt, Yt, R, Q = _f(Xt), _f(Yt), _f(R), _f(Q)
    nt, M = Xt.shape; P = Yt.shape[1]; A = R.shape[1]
    press = np.empty((A, P), order="F")
    lib().orc_pls_press(_p(Xt), _p(Yt), _sz(nt), _sz(M), _sz(P), _sz(A), _p(R), _p(Q), _p(press))
    return press


def pls_optimal_components(Xt, Yt, R, Q, rule=RULE_MIN_PRESS):
    Xt, Yt, R, Q = _f(Xt), _f(Yt), _f(R), _f(Q)
    nt, M = Xt.shape; P = Yt.shape[1]; A = R.shape[1]
    per = np.empty(P, dtype=np.int32)
    best = lib().orc_pls_optimal_components(_p(Xt), _p(Yt), _sz(nt), _sz(M), _sz(P), _sz(A),
                                            _p(R), _p(Q), C.c_int(rule), _p(per))
    return best, per


def wilcoxon_p(e1, e2):
    e1, e2 = _f(e1), _f(e2)
    return lib().orc_wilcoxon_p(_p(e1), _p(e2), _sz(e1.size))


def project_distance(X, mean, sd, R, a, obs_scores):
    X, mean, sd, R, obs_scores = _f(X), _f(mean), _f(sd), _f(R), _f(obs_scores)
    n, M = X.shape
    assert R.shape[0] == M
    out = np.empty(n)
    lib().orc_project_distance(_p(X), _sz(n), _sz(M), _p(mean), _p(sd), _p(R), _sz(a),
                               _p(obs_scores), _p(out))
    return out


def particle_ranking_pls(X, Y, obs, train_frac, max_comp=0, rule=RULE_DEFAULT, want="idx"):
    """Returns dict(idx, dist, ncomp, R, Q, mean, sd, press)."""
    X, Y, obs = _f(X), _f(Y), _f(obs)
    N, M = X.shape; P = Y.shape[1]
    A = max_comp if max_comp > 0 else min(M, P)
    idx = np.empty(N, dtype=np.uint64); dist = np.empty(N)
    ncomp = C.c_int32(0)
    R = np.empty((M, A), order="F"); Q = np.empty((P, A), order="F")
    mean = np.empty(M); sd = np.empty(M); press = np.empty((A, P), order="F")
    rc = lib().orc_particle_ranking_pls(_p(X), _p(Y), _p(obs), _sz(N), _sz(M), _sz(P),
                                        C.c_double(train_frac), C.c_int(max_comp), C.c_int(rule),
                                        _p(idx), _p(dist), C.byref(ncomp), _p(R), _p(Q),
                                        _p(mean), _p(sd), _p(press))
    if rc:
        raise ValueError("orc_particle_ranking_pls rc=%d" % rc)
    return dict(idx=idx, dist=dist, ncomp=ncomp.value, R=R, Q=Q, mean=mean, sd=sd, press=press)


def particle_ranking_simple(X, obs):
    X, obs = _f(X), _f(obs)
    N, M = X.shape
    idx = np.empty(N, dtype=np.uint64); dist = np.empty(N)
    lib().orc_particle_ranking_simple(_p(X), _p(obs), _sz(N), _sz(M), _p(idx), _p(dist))
    return idx, dist


# ---- weights / variances --------------------------------------------------------------
def doubled_variance(theta):
    theta = _f(theta); K, P = theta.shape
    dv = np.empty(P)
    lib().orc_doubled_variance(_p(theta), _sz(K), _sz(P), _p(dv))
    return dv


def prior_likelihood(pr, v):
    return lib().orc_prior_likelihood(C.byref(pr), C.c_double(v))


def prior_recast(pr, v):
    return lib().orc_prior_recast(C.byref(pr), C.c_double(v))


def prior_valid(pr, v):
    return bool(lib().orc_prior_valid(C.byref(pr), C.c_double(v)))


def weights_uniform(K):
    w = np.empty(K)
    lib().orc_weights_uniform(_sz(K), _p(w))
    return w


def weights_epanechnikov(priors, theta, theta_prev, w_prev, dv_prev):
    theta, theta_prev, w_prev, dv_prev = _f(theta), _f(theta_prev), _f(w_prev), _f(dv_prev)
    K, P = theta.shape; Kp = theta_prev.shape[0]
    w = np.empty(K)
    lib().orc_weights_epanechnikov(priors, _p(theta), _sz(K), _p(theta_prev), _sz(Kp), _p(w_prev), _p(dv_prev), _sz(P), _p(w))
    return w


def weights_importance(priors, theta, theta_prev, w_prev, dv_prev, zero_dv_policy=0):
    theta, theta_prev, w_prev, dv_prev = _f(theta), _f(theta_prev), _f(w_prev), _f(dv_prev)
    K, P = theta.shape; Kp = theta_prev.shape[0]
    w = np.empty(K)
    lib().orc_weights_importance(priors, _p(theta), _sz(K), _p(theta_prev), _sz(Kp), _p(w_prev),
                                 _p(dv_prev), _sz(P), C.c_int(zero_dv_policy), _p(w))
    return w


def mvn_setup(theta):
    theta = _f(theta); K, P = theta.shape
    L = np.empty((P, P), order="F"); cov = np.empty((P, P), order="F")
    rc = lib().orc_mvn_setup(_p(theta), _sz(K), _sz(P), _p(L), _p(cov))
    return rc, L, cov


# ---- RNG ------------------------------------------------------------------------------
def rng(seed):
    r = Rng()
    lib().orc_rng_set(C.byref(r), C.c_ulong(seed))
    return r


def rng_get(r):
    return lib().orc_rng_get(C.byref(r))


def rng_uniform(r):
    return lib().orc_rng_uniform(C.byref(r))


def rng_uniform_int(r, n):
    return lib().orc_rng_uniform_int(C.byref(r), C.c_ulong(n))


def ran_gaussian(r, sigma):
    return lib().orc_ran_gaussian(C.byref(r), C.c_double(sigma))


def ran_gaussian_pdf(x, sigma):
    return lib().orc_ran_gaussian_pdf(C.c_double(x), C.c_double(sigma))


def discrete_preproc(w):
    w = _f(w); K = w.size
    F = np.empty(K); A = np.empty(K, dtype=np.uint64)
    lib().orc_discrete_preproc(_sz(K), _p(w), _p(F), _p(A))
    return F, A


def resample(r, w, n):
    w = _f(w)
    idx = np.empty(n, dtype=np.uint64)
    lib().orc_resample(C.byref(r), _p(w), _sz(w.size), _sz(n), _p(idx))
    return idx


def sample_predictive_priors(r, n, w, theta, priors, dv):
    w, theta, dv = _f(w), _f(theta), _f(dv)
    K, P = theta.shape
    out = np.empty((n, P), order="F"); par = np.empty(n, dtype=np.uint64)
    fb = lib().orc_sample_predictive_priors(C.byref(r), _sz(n), _p(w), _p(theta), _sz(K), _sz(P),
                                            priors, _p(dv), _p(out), _p(par))
    return out, par, fb


def sample_mvn_predictive_priors(r, n, w, theta, priors, L, max_tries=0):
    w, theta, L = _f(w), _f(theta), _f(L)
    K, P = theta.shape
    out = np.empty((n, P), order="F"); par = np.empty(n, dtype=np.uint64)
    rej = lib().orc_sample_mvn_predictive_priors(C.byref(r), _sz(n), _p(w), _p(theta), _sz(K),
                                                 _sz(P), priors, _p(L), _sz(max_tries), _p(out), _p(par))
    return out, par, rej


def generation(X, Y, obs, priors, K, Nnext, r, theta_prev=None, w_prev=None, dv_prev=None,
               train_frac=0.5, max_comp=0, rule=RULE_DEFAULT, multivariate=True,
               zero_dv_policy=0):
    X, Y, obs = _f(X), _f(Y), _f(obs)
    N, M = X.shape; P = Y.shape[1]
    cfg = GenCfg(N, M, P, K, 0 if theta_prev is None else theta_prev.shape[0], Nnext, train_frac,
                 max_comp, rule, int(multivariate), zero_dv_policy)
    if theta_prev is not None:
        theta_prev, w_prev, dv_prev = _f(theta_prev), _f(w_prev), _f(dv_prev)
    idx = np.empty(K, dtype=np.uint64); w = np.empty(K); dv = np.empty(P)
    L = np.empty((P, P), order="F"); nxt = np.empty((Nnext, P), order="F")
    parent = np.empty(Nnext, dtype=np.uint64); seeds = np.empty(Nnext, dtype=np.uint64)
    ncomp = C.c_int32(0)
    rc = lib().orc_generation(C.byref(cfg), _p(X), _p(Y), _p(obs), priors, _p(theta_prev),
                              _p(w_prev), _p(dv_prev), C.byref(r), _p(idx), _p(w), _p(dv), _p(L),
                              _p(nxt), _p(parent), _p(seeds), C.byref(ncomp))
    if rc:
        raise ValueError("orc_generation rc=%d" % rc)
    return dict(idx=idx, w=w, dv=dv, L=L, next=nxt, parent=parent, seeds=seeds, ncomp=ncomp.value)
