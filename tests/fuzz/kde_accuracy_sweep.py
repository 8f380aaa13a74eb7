"""Accuracy of the split-operand pair-sum kernel over every parameter count it takes (5..64), against the fp64 vector kernel
(itself held to <= 1e-9 of the oracle by tests/test_gpu_parity.py): random set sizes with ragged tiles, previous weights over a
few or over sixty binades, zero weights, one far row on each side; every variant of the kernel (plain / norm pieces folded into
spare K-slots / tiles in the order of the norm tops, the last one forced at these sizes).
    python tests/fuzz/kde_accuracy_sweep.py [out.json] [cases per parameter count]
Writes, per parameter count: the largest and the rms relative error of a weight, the number of weights compared."""
import json
import os
os.environ.setdefault("ABC_DIAG", "1")     # the library reads its diagnostic switches only beside this
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from abcsmc_amd import _lib, abcutil, synthetic

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kde_accuracy.json"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = _lib.default_context(0)
res = {}
worst = 0.0
for P in range(5, 65):
    errs = []
    for rep in range(reps):
        g = np.random.default_rng(1000 * P + rep)
        K, Kp = int(g.integers(300, 2600)), int(g.integers(300, 4200))
        wl = synthetic.Workload(8, P, 7000 + 13 * P + rep)
        _, th = wl.rows(0, K)
        th = np.asfortranarray(wl.mu_y + g.uniform(0.3, 1.0) * (th - wl.mu_y))
        tp, wp, dv = wl.previous_set(Kp)
        wp = g.random(Kp)
        if rep % 2:
            wp *= np.exp2(g.uniform(-60, 0, Kp))
        wp[:: 97] = 0.0
        unit = np.sqrt(dv) / np.sqrt(np.log2(np.e))
        th, tp = th.copy(), tp.copy()
        th[3, P // 2] += 12.0 * unit[P // 2]               # one far row on each side (takes the fp64 fix-up route)
        tp[7, P - 1] -= 11.0 * unit[P - 1]
        pri = _lib.make_priors(wl.prior_spec())
        ctx.set_kde_mode(_lib.KDE_FP64)
        ref = abcutil.weight_predictive_prior(pri, th, tp, wp, dv, ctx=ctx)
        ctx.set_kde_mode(_lib.KDE_AUTO)
        for topn in ("1e30", "0"):
            os.environ["ABC_KDE_TOPN_MIN_PAIRS"] = topn
            w = abcutil.weight_predictive_prior(pri, th, tp, wp, dv, ctx=ctx)
            assert ctx.kde_last_kernel() == _lib.KDE_RAN_SPLIT
            ok = ref > 0
            assert np.array_equal(w == 0, ref == 0)
            errs.append(np.abs(w - ref)[ok] / ref[ok])
        del os.environ["ABC_KDE_TOPN_MIN_PAIRS"]
    e = np.concatenate(errs)
    res[str(P)] = {"max_rel": float(e.max()), "rms_rel": float(np.sqrt((e ** 2).mean())), "weights": int(e.size)}
    worst = max(worst, float(e.max()))
    print("P = %2d  max %.2e  rms %.2e  (%d weights)" % (P, e.max(), np.sqrt((e ** 2).mean()), e.size), flush=True)
bound = {str(P): (8e-7 if P > 32 else 5.5e-7 if P > 16 else 5e-7) for P in range(5, 65)}
bad = [P for P in res if res[P]["max_rel"] >= bound[P]]
json.dump({"reference": "fp64 vector kernel (k_kde), same device", "asserted_bound": {"5..16": 5e-7, "17..32": 5.5e-7, "33..64": 8e-7},
           "north_star_bound": 1e-6, "worst_max_rel": worst, "over_bound": bad, "per_parameter_count": res}, open(out, "w"), indent=1)
print("worst %.2e; over the asserted bound: %s" % (worst, bad or "none"))
