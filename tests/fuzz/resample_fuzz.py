"""Resampling and proposals against the CPU oracle at random sizes: the alias table of every kind of weight vector (uniform, random,
sixty binades, many exact zeros, a single survivor, nearly uniform) from 1 entry to 60 000 -- the host build below 20 000
entries, the device build (alias_dev.hip) from there --, parents bit for bit (abc_sample_posterior), the rng state behind them;
then abc_sample_mvn_predictive_priors / abc_sample_predictive_priors: parents and seeds bit for bit in the default noise mode,
every proposal inside its prior's support and on the integer grid where the prior is one, and -- reference-stream mode, small
sizes -- the proposals themselves, the seeds and the final rng state bit for bit.
    python tests/fuzz/resample_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from abcsmc_amd import _lib, abcutil
from oracle import pyoracle as oracle

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/resample_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = _lib.default_context(0)
g = np.random.default_rng(seed0)
rows, fails = [], []


def weights_of(kind, K):
    if kind == 0:
        return np.full(K, 1.0 / K)
    w = g.random(K) + 1e-12
    if kind == 2:
        w *= np.exp2(g.uniform(-60, 0, K))
    elif kind == 3:
        w[g.random(K) < 0.6] = 0.0
    elif kind == 4:
        w[:] = 0.0
        w[int(g.integers(0, K))] = 0.7
    elif kind == 5:
        w = 1.0 + 1e-9 * g.normal(size=K)
    if w.max() == 0.0:
        w[0] = 1.0
    return w


for case in range(cases):
    big = case % 4 == 0
    K = int(g.integers(20000, 60001)) if big else int(g.choice([1, 2, 3, 7, 64, 65, 1000])) if case % 4 == 1 else int(g.integers(1, 20000))
    kind = int(g.integers(0, 6))
    n = int(g.integers(1, 30000))
    sd = int(g.integers(1, 1 << 30))
    tag = dict(case=case, K=K, weights_kind=kind, draws=n, seed=sd)
    problems = []
    try:
        w = weights_of(kind, K)
        r, o = abcutil.rng(sd), oracle.rng(sd)
        par = abcutil.gsl_rng_nonuniform_int(r, n, w, ctx=ctx)
        opar = oracle.resample(o, w, n)
        if not np.array_equal(par, np.asarray(opar, dtype=np.uint64)):
            problems.append("parents differ (%d of %d)" % (int((par != np.asarray(opar, dtype=np.uint64)).sum()), n))
        if abcutil.rng_get(r) != oracle.rng_get(o):
            problems.append("rng state differs behind the draws")
        # proposals: a posterior of Kq particles, P parameters of mixed priors
        P = int(g.integers(1, 41))
        Kq = min(K, int(g.integers(P + 2, 3000)))
        if Kq >= P + 2:
            cols, spec = [], []
            for p in range(P):
                t = p % 3
                if t == 0:
                    cols.append(g.normal(5.0, 2.0, Kq)); spec.append((_lib.PRIOR_GAUSS, 5.0, float(g.choice([3.0, 50.0]))))
                elif t == 1:
                    cols.append(np.round(g.uniform(40, 60, Kq))); spec.append((_lib.PRIOR_UNIF_INT, 0, 100))
                else:
                    lo, hi = (0.0, 1.0) if g.integers(0, 2) else (0.25, 0.75)            # (the tight one: many rejections)
                    cols.append(g.uniform(0.3, 0.7, Kq)); spec.append((_lib.PRIOR_UNIF_REAL, lo, hi))
            th = np.asfortranarray(np.column_stack(cols))
            wq = weights_of(int(g.integers(1, 4)), Kq)
            nq = int(g.integers(1, 6000))
            mv = bool(g.integers(0, 2))
            tag.update(P=P, Kq=Kq, proposals=nq, multivariate=mv)
            pri, opri = _lib.make_priors(spec), oracle.make_priors(spec)
            L = abcutil.setup_mvn_sampler(th, ctx=ctx) if mv else None
            dv = None if mv else abcutil.calculate_doubled_variance(th, ctx=ctx)
            for mode in ("device", "reference") if nq <= 1500 else ("device",):
                ctx.set_noise_mode(_lib.NOISE_REFERENCE_STREAM if mode == "reference" else _lib.NOISE_DEVICE)
                r, o = abcutil.rng(sd + 1), oracle.rng(sd + 1)
                if mv:
                    outp, parent, seeds = abcutil.sample_mvn_predictive_priors(r, nq, wq, th, pri, L, seeds=True, ctx=ctx)
                    oout, opar2, _ = oracle.sample_mvn_predictive_priors(o, nq, wq, th, opri, L)
                else:
                    outp, parent, seeds = abcutil.sample_predictive_priors(r, nq, wq, th, pri, dv, seeds=True, ctx=ctx)
                    oout, opar2, _ = oracle.sample_predictive_priors(o, nq, wq, th, opri, dv)
                if not np.array_equal(parent, opar2):
                    problems.append("%s noise: parents differ" % mode)
                if not np.isfinite(outp).all():
                    problems.append("%s noise: proposals not finite" % mode)
                for p in range(P):
                    k, a, b = spec[p]
                    if k == _lib.PRIOR_UNIF_REAL and (outp[:, p].min() < a or outp[:, p].max() > b):
                        problems.append("%s noise: parameter %d outside its support" % (mode, p))
                    if k == _lib.PRIOR_UNIF_INT and (np.any(outp[:, p] != np.round(outp[:, p])) or outp[:, p].min() < a or outp[:, p].max() > b):
                        problems.append("%s noise: integer parameter %d off its grid / range" % (mode, p))
                if mode == "reference":
                    if not np.array_equal(outp, oout):
                        problems.append("reference stream: proposals differ from the oracle's (%d entries)" % int((outp != oout).sum()))
                    oseeds = np.array([oracle.rng_get(o) for _ in range(nq)], dtype=np.uint64)
                    if not np.array_equal(seeds, oseeds):
                        problems.append("reference stream: seeds differ")
                    if abcutil.rng_get(r) != oracle.rng_get(o):
                        problems.append("reference stream: final rng state differs")
                else:
                    o2 = oracle.rng(sd + 1)
                    for _ in range(nq):
                        oracle.rng_get(o2)
                    if not np.array_equal(seeds, np.array([oracle.rng_get(o2) for _ in range(nq)], dtype=np.uint64)):
                        problems.append("device noise: seeds differ")
            ctx.set_noise_mode(_lib.NOISE_DEVICE)
    except Exception as e:        # noqa: BLE001
        problems.append("exception: %r" % (e,))
        ctx.set_noise_mode(_lib.NOISE_DEVICE)
    tag["problems"] = problems
    rows.append(tag)
    if problems:
        fails.append(tag)
    print(("FAIL " if problems else "ok   ") + json.dumps(tag), flush=True)
b, f = ctx.alias_stats(reset=True)
json.dump({"cases": len(rows), "failed": len(fails), "device_alias_builds": b, "device_alias_fallbacks": f, "failures": fails, "rows": rows},
          open(out, "w"), indent=0)
print("%d cases, %d with problems; device alias builds %d, fall-backs %d" % (len(rows), len(fails), b, f))
