"""Host-side mirror of the reference's `namespace ABC` free functions (AbcUtil.h:78-172).

Same names, argument meaning and error behaviour as the reference (asserts / exits become
exceptions); every function is a thin call into the C ABI (include/abcsmc_hip.h), which runs the
HIP kernels.  Inputs are numpy arrays (host memory, as the reference's Eigen matrices); matrices
are converted to column-major float64, the reference's Mat2D layout.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Rng, lib, default_context


def _f(a):
    return np.asfortranarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _ctx(ctx):
    return ctx if ctx is not None else default_context(0)


def rng(seed):
    """gsl_rng_alloc(gsl_rng_taus2) + gsl_rng_set (examples/include/examples.h:10,64)."""
    r = Rng()
    lib().abc_rng_set(C.byref(r), C.c_ulong(seed))
    return r


def rng_get(r):
    return lib().abc_rng_get(C.byref(r))


def particle_ranking_PLS(X_orig, Y_orig, target_values, training_fraction, K=None, max_comp=0,
                         rule=_lib.RULE_DEFAULT, details=False, ctx=None):
    """ABC::particle_ranking_PLS (AbcUtil.cpp:423-458).  Returns the ascending-distance particle
    indices (first K; K=None -> all N as the reference)."""
    ctx = _ctx(ctx)
    X, Y, obs = _f(X_orig), _f(Y_orig), _f(target_values)
    N, M = X.shape
    P = Y.shape[1]
    if Y.shape[0] != N or obs.size != M:
        raise ValueError("shape mismatch")
    if not (0 < training_fraction <= 1):
        raise ValueError("training_fraction must be in (0,1]")        # assert at AbcUtil.cpp:428
    K = N if K is None else int(K)
    A = max_comp if max_comp > 0 else min(M, P)
    idx = np.empty(K, dtype=np.uint64)
    dist = np.empty(K)
    ncomp = C.c_int32(0)
    R = np.empty((M, A), order="F")
    mean = np.empty(M)
    sd = np.empty(M)
    ctx.check(lib().abc_particle_ranking_pls(ctx.handle, _p(X), _p(Y), _p(obs), N, M, P,
                                             float(training_fraction), int(max_comp), int(rule), K,
                                             _p(idx), _p(dist), C.addressof(ncomp), _p(R), _p(mean), _p(sd)))
    if details:
        return dict(idx=idx, dist=dist, ncomp=ncomp.value, R=R, mean=mean, sd=sd)
    return idx


def particle_ranking_simple(X_orig, Y_orig, target_values, K=None, details=False, ctx=None):
    """ABC::particle_ranking_simple (AbcUtil.cpp:408-421); Y_orig is unused, as in the reference."""
    ctx = _ctx(ctx)
    X, obs = _f(X_orig), _f(target_values)
    N, M = X.shape
    K = N if K is None else int(K)
    idx = np.empty(K, dtype=np.uint64)
    dist = np.empty(K)
    ctx.check(lib().abc_particle_ranking_simple(ctx.handle, _p(X), _p(obs), N, M, K, _p(idx), _p(dist)))
    if details:
        return dict(idx=idx, dist=dist)
    return idx


def calculate_doubled_variance(params, ctx=None):
    """ABC::calculate_doubled_variance (AbcUtil.cpp:528-537)."""
    ctx = _ctx(ctx)
    th = _f(params)
    K, P = th.shape
    dv = np.empty(P)
    ctx.check(lib().abc_calculate_doubled_variance(ctx.handle, _p(th), K, P, _p(dv)))
    return dv


def weight_predictive_prior(mpars, params, prev_params=None, prev_weights=None,
                            prev_doubled_variance=None, ctx=None):
    """ABC::weight_predictive_prior, both overloads (AbcUtil.cpp:539-586).  mpars: ctypes array of
    Prior (the POD form of the reference's vector<const Parameter*>)."""
    ctx = _ctx(ctx)
    th = _f(params)
    K, P = th.shape
    w = np.empty(K)
    if prev_params is None:
        ctx.check(lib().abc_weight_predictive_prior_uniform(ctx.handle, K, _p(w)))
        return w
    tp, wp, dvp = _f(prev_params), _f(prev_weights), _f(prev_doubled_variance)
    Kp = tp.shape[0]
    if tp.shape[1] != P or wp.size != Kp or dvp.size != P or len(mpars) != P:
        raise ValueError("shape mismatch")
    ctx.check(lib().abc_weight_predictive_prior(ctx.handle, C.addressof(mpars), _p(th), K, P, _p(tp), Kp,
                                                _p(wp), _p(dvp), _p(w)))
    return w


def setup_mvn_sampler(params, ctx=None):
    """ABC::setup_mvn_sampler (AbcUtil.cpp:462-488): P x P matrix, lower triangle = Cholesky factor."""
    ctx = _ctx(ctx)
    th = _f(params)
    K, P = th.shape
    L = np.empty((P, P), order="F")
    ctx.check(lib().abc_setup_mvn_sampler(ctx.handle, _p(th), K, P, _p(L)))
    return L


def gsl_rng_nonuniform_int(RNG, num_samples, weights, ctx=None):
    """ABC::gsl_rng_nonuniform_int (AbcUtil.cpp:111-120): advances RNG by num_samples outputs."""
    ctx = _ctx(ctx)
    w = _f(weights)
    idx = np.empty(int(num_samples), dtype=np.uint64)
    ctx.check(lib().abc_sample_posterior(ctx.handle, C.addressof(RNG), _p(w), w.size, int(num_samples), _p(idx)))
    return idx


def sample_posterior(RNG, num_samples, weights, posterior, ctx=None):
    """ABC::sample_posterior (AbcUtil.cpp:366-375)."""
    post = _f(posterior)
    return post[gsl_rng_nonuniform_int(RNG, num_samples, weights, ctx=ctx).astype(np.int64), :]


def _sample(fn, RNG, num_samples, weights, parameter_prior, pars, aux, want_seeds, ctx):
    ctx = _ctx(ctx)
    w, th, aux = _f(weights), _f(parameter_prior), _f(aux)
    K, P = th.shape
    n = int(num_samples)
    out = np.empty((n, P), order="F")
    parent = np.empty(n, dtype=np.uint64)
    seeds = np.empty(n, dtype=np.uint64) if want_seeds else None
    ctx.check(fn(ctx.handle, C.addressof(RNG), n, _p(w), _p(th), K, P, C.addressof(pars), _p(aux), _p(out),
                 _p(parent), _p(seeds)))
    return (out, parent, seeds) if want_seeds else (out, parent)


def sample_mvn_predictive_priors(RNG, num_samples, weights, parameter_prior, pars, L, seeds=False, ctx=None):
    """ABC::sample_mvn_predictive_priors (AbcUtil.cpp:391-404). Returns (noised_pars, parent_rows[, seeds])."""
    return _sample(lib().abc_sample_mvn_predictive_priors, RNG, num_samples, weights, parameter_prior, pars, L,
                   seeds, ctx)


def sample_predictive_priors(RNG, num_samples, weights, parameter_prior, pars, doubled_variance, seeds=False,
                             ctx=None):
    """ABC::sample_predictive_priors (AbcUtil.cpp:377-389). Returns (noised_pars, parent_rows[, seeds])."""
    return _sample(lib().abc_sample_predictive_priors, RNG, num_samples, weights, parameter_prior, pars,
                   doubled_variance, seeds, ctx)
