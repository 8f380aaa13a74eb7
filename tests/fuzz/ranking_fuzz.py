"""abc_particle_ranking_pls / _simple against the CPU oracle at random shapes and unfriendly data: any row count (odd, smaller than a
tile), 2..200 metrics, 1..70 responses, any component cap, training fractions in (0.2, 1], both component rules; duplicated
rows (exact ties), columns scaled by 1e-6..1e6, a column with a mean 1e7 standard deviations from zero (what the pilot shift
of the one-pass statistics is for), a constant column, nearly collinear columns.
What must hold (tests/test_gpu_parity.py::test_particle_ranking_pls): component count equal; means 1e-12, standard deviations
1e-10; the distances and the order are BIT-EXACT given the device's model (the oracle's projection fed that model); against the
oracle's own model distances to 1e-6 and the order up to near-ties.
    python tests/fuzz/ranking_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from abcsmc_amd import _lib, abcutil, synthetic
from oracle import pyoracle as oracle

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ranking_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = _lib.default_context(0)
g = np.random.default_rng(seed0)


def fma_dot(a, b):
    """ascending fma chain in float64 (what orc_* and the kernels compute), in exact arithmetic"""
    from fractions import Fraction
    s = 0.0
    for x, y in zip(a, b):
        s = float(Fraction(float(x)) * Fraction(float(y)) + Fraction(s))
    return s


def near_tie_ok(idx_a, idx_b, dist_full, tol=1e-9):
    bad = np.nonzero(idx_a != idx_b)[0]
    for k in bad:
        da, db = dist_full[int(idx_a[k])], dist_full[int(idx_b[k])]
        if abs(da - db) > tol * max(abs(da), abs(db), 1e-300):
            return False
    return True


rows, fails = [], []
for case in range(cases):
    M = int(g.integers(2, 201)) if case % 3 else int(g.choice([2, 16, 17, 32, 48, 49, 64, 96, 97, 112, 128, 144, 160, 161]))
    P = int(g.integers(1, 71)) if case % 4 == 0 else int(g.integers(1, 33))
    N = int(g.integers(60, 5000))
    tf = float(g.choice([0.5, 0.5, 1.0, g.uniform(0.2, 0.95)]))
    A = int(g.integers(0, min(M, 40) + 1))
    wil = bool(g.integers(0, 5) == 0) and P <= 40 and tf < 1.0
    mods = [m for m in ("dups", "scaled", "offset", "constant", "collinear") if g.integers(0, 4) == 0]
    sd = int(g.integers(1, 1 << 30))
    tag = dict(case=case, N=N, M=M, P=P, A=A, train_frac=tf, wilcoxon=wil, mods=mods, seed=sd)
    try:
        wl = synthetic.Workload(M, P, sd)
        X, Y = wl.rows(0, N)
        X, Y = X.copy(order="F"), Y.copy(order="F")
        obs = wl.observed().copy()
        if "scaled" in mods:
            sc = 10.0 ** g.integers(-6, 7, M)
            X *= sc; obs *= sc
        if "offset" in mods:
            c = int(g.integers(0, M)); off = 1e7 * X[:, c].std()
            X[:, c] += off; obs[c] += off
        if "constant" in mods and M > 2:
            X[:, int(g.integers(0, M))] = 3.25
            A = min(A if A > 0 else min(M, P), M - 1)   # (components beyond the rank of X are rounding noise, and so is the count that minimises PRESS among them; 0 = as many as fit)
            tag["A"] = A
        if "collinear" in mods and M > 3:
            a, b = g.choice(M, 2, replace=False)
            X[:, a] = 2.0 * X[:, b] + 1e-3 * X[:, b].std() * g.normal(size=N)
        if "dups" in mods:
            src, dst = g.integers(0, N, N // 10), g.integers(0, N, N // 10)
            X[dst] = X[src]
        rule = _lib.RULE_WILCOXON if wil else _lib.RULE_MIN_PRESS
        gd = abcutil.particle_ranking_PLS(X, Y, obs, tf, max_comp=A, rule=rule, details=True, ctx=ctx)
        od = oracle.particle_ranking_pls(X, Y, obs, tf, A, rule=(oracle.RULE_WILCOXON if wil else oracle.RULE_MIN_PRESS))
        problems = []
        if gd["ncomp"] != od["ncomp"]:
            problems.append("ncomp %d != %d" % (gd["ncomp"], od["ncomp"]))
        # (a mean is accurate to rounding RELATIVE TO THE COLUMN'S SPREAD: one that happens to be near zero has no relative accuracy)
        if not np.all(np.abs(gd["mean"] - od["mean"]) <= 1e-12 * np.abs(od["mean"]) + 1e-13 * od["sd"]):
            problems.append("mean %.1e" % np.max(np.abs(gd["mean"] - od["mean"]) / np.maximum(np.abs(od["mean"]), 1e-300)))
        sdz = od["sd"] == 0
        if not np.array_equal(gd["sd"] == 0, sdz):
            problems.append("zero-variance columns differ")
        elif not np.allclose(gd["sd"][~sdz], od["sd"][~sdz], rtol=1e-9):
            problems.append("sd %.1e" % np.max(np.abs(gd["sd"][~sdz] - od["sd"][~sdz]) / od["sd"][~sdz]))
        nc = gd["ncomp"]
        with np.errstate(invalid="ignore", divide="ignore"):
            zobs = np.where(gd["sd"] == 0, 0.0, (obs - gd["mean"]) / gd["sd"])
        so = np.array([fma_dot(zobs, gd["R"][:, k]) for k in range(nc)])
        d_staged = oracle.project_distance(X, gd["mean"], gd["sd"], gd["R"], nc, so)
        order_staged = oracle.ordered(d_staged)
        if not np.array_equal(gd["idx"], order_staged):
            problems.append("order not bit-exact given the device's model (%d positions)" % int((gd["idx"] != order_staged).sum()))
        elif not np.array_equal(gd["dist"], d_staged[order_staged.astype(int)]):
            problems.append("distances not bit-exact given the device's model")
        if gd["ncomp"] == od["ncomp"] and "collinear" not in mods and "constant" not in mods:
            if not np.allclose(gd["dist"], od["dist"][gd["idx"].astype(int)], rtol=1e-6):
                problems.append("distances vs the oracle's model %.1e" % np.max(np.abs(gd["dist"] - od["dist"][gd["idx"].astype(int)]) / od["dist"][gd["idx"].astype(int)]))
            elif not near_tie_ok(gd["idx"], od["idx"], od["dist"], 1e-7):
                problems.append("order differs from the oracle's beyond near-ties")
        # the simple ranking on the same matrix
        gs = abcutil.particle_ranking_simple(X, Y, obs, details=True, ctx=ctx)
        oi, odist = oracle.particle_ranking_simple(X, obs)
        if not np.allclose(gs["dist"], odist[gs["idx"].astype(int)], rtol=1e-9):
            problems.append("simple ranking distances")
        elif not near_tie_ok(gs["idx"], oi, odist):
            problems.append("simple ranking order")
        tag.update(ncomp=int(od["ncomp"]), problems=problems)
    except Exception as e:        # noqa: BLE001
        tag.update(problems=["exception: %r" % (e,)])
    rows.append(tag)
    if tag["problems"]:
        fails.append(tag)
    print(("FAIL " if tag["problems"] else "ok   ") + json.dumps(tag), flush=True)
json.dump({"cases": len(rows), "failed": len(fails), "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("%d cases, %d with problems" % (len(rows), len(fails)))
