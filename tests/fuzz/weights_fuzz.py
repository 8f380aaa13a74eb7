"""abc_weight_predictive_prior against the CPU oracle on adversarial sets: any number of far particles on either side, far in any
coordinate(s) (a row of the limb-tile kernel is spread over several lanes), previous weights of exactly 0 / over sixty binades /
beyond the kernel's exponent range, tiny sets (fewer rows than a tile), a parameter of zero variance, duplicated particles,
every parameter count from 1 to 70.  (Far means 8.5 to 18 proposal widths in one or two coordinates of a row: densities down to 2^-500.  Further out
the sums enter double precision's subnormal range -- 2^-1060 carries fourteen bits -- and device and oracle, both right to that
precision, differ by 1e-5: seen with rows 40 widths out, not a finding.)  Zero patterns must agree, every positive weight within the error budget of the kernel that
ran (fp64: 1e-9; split: 5e-7 / 5.5e-7 / 8e-7 by chunk count), repeated calls bit-identical.
    python tests/fuzz/weights_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from abcsmc_amd import _lib, abcutil, synthetic
from oracle import pyoracle as oracle

def sc_all(th, unit):
    return (th - np.median(th, axis=0)) / unit


out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/weights_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = _lib.default_context(0)
g = np.random.default_rng(seed0)
rows, fails = [], []
for case in range(cases):
    P = int(g.integers(1, 71))
    K = int(g.choice([1, 2, 31, 33, 64, 65, 200, 257, 700])) if case % 3 == 0 else int(g.integers(1, 900))
    Kp = int(g.choice([1, 2, 31, 33, 63, 64, 65, 129, 500])) if case % 3 == 1 else int(g.integers(1, 1200))
    sd = int(g.integers(1, 1 << 30))
    n_far_i = int(g.choice([0, 0, 1, 2, 5, K // 10 + 1, K // 3 + 1]))
    n_far_j = int(g.choice([0, 0, 1, 2, 5, 20, Kp // 4 + 1]))
    heavy = int(g.integers(0, 4))                 # 0: plain, 1: sixty binades, 2: some beyond 2^100 / below 2^-300, 3: many zeros
    zero_dv = bool(g.integers(0, 8) == 0) and P > 1
    dup = bool(g.integers(0, 5) == 0)
    tag = dict(case=case, P=P, K=K, Kp=Kp, seed=sd, far_new=n_far_i, far_prev=n_far_j, weights_kind=heavy, zero_dv=zero_dv, dup=dup)
    try:
        wl = synthetic.Workload(8, P, sd)
        _, th = wl.rows(0, K)
        th = np.asfortranarray(wl.mu_y + g.uniform(0.3, 1.0) * (th - wl.mu_y))
        tp, wp, dv = wl.previous_set(Kp)
        tp, dv = tp.copy(), dv.copy()
        wp = g.random(Kp) + 1e-3
        if heavy == 1:
            wp *= np.exp2(g.uniform(-60, 0, Kp))
        elif heavy == 2:
            wp *= np.exp2(g.uniform(-60, 0, Kp))
            wp[g.integers(0, Kp, max(1, Kp // 50))] *= 2.0 ** 120
            wp[g.integers(0, Kp, max(1, Kp // 50))] *= 2.0 ** -330
        elif heavy == 3:
            wp[g.random(Kp) < 0.4] = 0.0
        if wp.max() == 0.0:
            wp[0] = 1.0
        unit = np.sqrt(dv) / np.sqrt(np.log2(np.e))
        th = th.copy()
        for i in g.choice(K, min(n_far_i, K), replace=False):
            for p in g.choice(P, min(P, int(g.integers(1, 3))), replace=False):
                th[i, p] += g.choice([-1.0, 1.0]) * g.uniform(8.5, 18.0) * unit[p]
        for j in g.choice(Kp, min(n_far_j, Kp), replace=False):
            for p in g.choice(P, min(P, int(g.integers(1, 3))), replace=False):
                tp[j, p] += g.choice([-1.0, 1.0]) * g.uniform(8.5, 18.0) * unit[p]
        if dup and K > 3 and Kp > 3:
            th[1] = th[0]; tp[2] = tp[1]; tp[3] = th[0]
        if zero_dv:
            pz = int(g.integers(0, P))
            dv[pz] = 0.0
            tp[:, pz] = tp[0, pz]
            th[:, pz] = tp[0, pz]
            if K > 2:
                th[K // 2, pz] += 1.0              # off the point mass: a kernel of width 0 gives it nothing
        # natural units: parameters of scale 1e3 make the product of seventy prior densities (and of the kernel's constants)
        # underflow whatever the kernel does -- the oracle's weights then are as arbitrary as the device's
        scl = np.where(dv > 0, np.sqrt(dv / 2.0), 1.0)
        th = np.asfortranarray((th - wl.mu_y) / scl)
        tp = np.asfortranarray((tp - wl.mu_y) / scl)
        dv = dv / scl ** 2
        unit = np.sqrt(np.where(dv > 0, dv, 1.0)) / np.sqrt(np.log2(np.e))
        only = os.environ.get("FUZZ_ONLY")                # replay one case (the generator's draws above are consumed all the same)
        if only is not None and case != int(only):
            continue
        spec = [(_lib.PRIOR_GAUSS, 0.0, 30.0)] * P
        pri, opri = _lib.make_priors(spec), oracle.make_priors(spec)
        epan = bool(os.environ.get("FUZZ_EPAN"))         # the Epanechnikov extension instead of the reference's Gaussian kernel
        ctx.set_weight_kernel(_lib.WEIGHT_EPANECHNIKOV if epan else _lib.WEIGHT_GAUSSIAN)
        ref = (oracle.weights_epanechnikov if epan else oracle.weights_importance)(opri, th, tp, wp, dv)
        w = abcutil.weight_predictive_prior(pri, th, tp, wp, dv, ctx=ctx)
        ran = ctx.kde_last_kernel()
        w2 = abcutil.weight_predictive_prior(pri, th, tp, wp, dv, ctx=ctx)
        tol = 1e-9 if ran != _lib.KDE_RAN_SPLIT else (8e-7 if P > 32 else 5.5e-7 if P > 16 else 5e-7)
        problems = []
        if not np.array_equal(w, w2, equal_nan=True):
            problems.append("not bit-identical when repeated")
        fin = np.isfinite(ref)
        if not fin.all():
            # a particle whose density underflows: its raw weight is inf, the normalised set inf / inf and zeros -- in the oracle as on
            # the device, but WHICH rows underflow differs at the edge (P factors multiplied there, one exponential of the summed
            # exponent here: DESIGN.md, declared deviations); nothing to compare in such a set
            tag["degenerate"] = True
        elif not np.isfinite(w).all():
            problems.append("non-finite weights where the oracle's are finite")
        elif not np.array_equal((w == 0)[fin], (ref == 0)[fin]):
            problems.append("zero pattern differs (%d vs %d zeros)" % (int((w == 0).sum()), int((ref == 0).sum())))
        else:
            ok = fin & (ref > 0)
            err = float(np.max(np.abs(w - ref)[ok] / ref[ok])) if ok.any() else 0.0
            tag["err"] = err
            if only is not None:
                e = np.where(ok, np.abs(w - ref) / np.where(ok, ref, 1.0), 0.0)
                worst = np.argsort(-e)[:8]
                sc = (th - th.mean(axis=0)) / unit
                top = np.argsort(-ref)[:6]
                ctx.set_kde_mode(_lib.KDE_FP64)
                w64 = abcutil.weight_predictive_prior(pri, th, tp, wp, dv, ctx=ctx)
                ctx.set_kde_mode(_lib.KDE_AUTO)
                print("largest weights: rows", top, "ref", ref[top], "split/ref - 1", w[top] / ref[top] - 1, "fp64/ref - 1", w64[top] / ref[top] - 1,
                      "max |scaled coord|", np.abs(sc_all(th, unit)[top]).max(axis=1), "|row|^2", (sc_all(th, unit)[top] ** 2).sum(axis=1))
                print("|w| - 1: split %.3e fp64 %.3e oracle %.3e; split/fp64 - 1 over the rows: min %.3e max %.3e" % (
                    np.linalg.norm(w) - 1, np.linalg.norm(w64) - 1, np.linalg.norm(ref) - 1, (w[ok] / w64[ok] - 1).min(), (w[ok] / w64[ok] - 1).max()))
                print("worst rows", worst, "errs", e[worst], "max |scaled coord| of those rows", np.abs(sc[worst]).max(axis=1),
                      "far prev rows (|coord| > 8):", int((np.abs((tp - th.mean(axis=0)) / unit).max(axis=1) > 8).sum()),
                      "weights outside [2^-300, 2^100] x norm:", int(((wp > 0) & ((wp / np.linalg.norm(wp) > 2.0 ** 100) | (wp / np.linalg.norm(wp) < 2.0 ** -300))).sum()))
            if err > tol:
                problems.append("weights %.2e > %.1e" % (err, tol))
        tag.update(kernel="split" if ran == _lib.KDE_RAN_SPLIT else "fp64", problems=problems)
    except Exception as e:        # noqa: BLE001
        tag.update(problems=["exception: %r" % (e,)])
    ctx.set_weight_kernel(_lib.WEIGHT_GAUSSIAN)
    rows.append(tag)
    if tag["problems"]:
        fails.append(tag)
    print(("FAIL " if tag["problems"] else "ok   ") + json.dumps(tag), flush=True)
json.dump({"cases": len(rows), "failed": len(fails), "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("%d cases, %d with problems" % (len(rows), len(fails)))
