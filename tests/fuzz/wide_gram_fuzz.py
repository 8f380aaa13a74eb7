"""The byte-limb statistics kernel of wide sets (gram.hip: k_pilot_scale, k_gram_i8, k_gram_far; 113..160 columns from 200 000 rows)
against numpy on random shapes and unfriendly data: row counts 200 000 .. 600 000 (even: the kernel's row pairs; odd counts stay on
the fp64 kernel and are drawn too), 113..160 columns split at random into metrics and parameters, training fractions 0.3..0.7, and
per case a random subset of: columns scaled over twelve decades, columns with a mean 1e6 standard deviations from zero, a constant
column, a column of tiny variance beside huge ones, single spikes up to 1e6 standard deviations (far rows: k_gram_far), a
Cauchy-tailed column (hundreds of far rows), a column that is constant on the pilot's sample rows only, duplicated rows.
What must hold (tests/test_gpu_parity.py::test_wide_gram_on_the_i8_matrix_pipe): column sums and the diagonal to fp64 rounding,
symmetry, bit-identical repeats, and every off-diagonal entry of X'X and X'Y within THE KERNEL'S ERROR MODEL, taken per case and per
entry (tests/_gram_model.py, shared with the fixed tests and quoted in include/abcsmc_hip.h):
    |G_ab - exact| <= 2^-32 x (4 x range_a x range_b x sqrt(rows of the partition) + range_a |S_b| + range_b |S_a|)
with range_c the column's fixed-point range as k_pilot_scale takes it (4 x the median of 64 group maxima of |x - shift| over 4096
evenly spread rows, rounded up to a power of two) and S_c the partition's sum of x - shift_c -- the rounding of every value to its
32-bit grid and the dropped low byte products as zero-mean noise per row, plus the coherent part a point-mass column adds.
Relative to sqrt(G_aa G_bb) the first term is 4 x 2^-32 (range/sigma)_a (range/sigma)_b / sqrt(rows): ~2e-10 for Gaussian-like
columns at 2e5 rows (range / sigma 10..19), more for a column whose mass sits in one point (range / sigma ~ 40).  `worst_vs_model`
in the output is the largest measured error in units of that bound.
    python tests/fuzz/wide_gram_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ABC_DIAG", "1")
import numpy as np

from abcsmc_amd import _lib, synthetic
import test_gpu_parity as T
from _gram_model import gram_error_bound

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wide_gram_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 17
ctx = _lib.default_context(0)
ctx.set_gram_mode(_lib.GRAM_I8)            # (round 6: ABC_GRAM_AUTO takes the kernel under test from 400 000 rows per partition only)
g = np.random.default_rng(seed0)

rows, fails = [], []
for case in range(cases):
    C = int(g.integers(97, 161))          # (97..112: the byte-limb kernel since round 5)
    P = int(g.integers(1, min(33, C - 80)))
    M = C - P
    N = int(g.integers(100_000, 300_001)) * 2 + (1 if g.integers(0, 8) == 0 else 0)
    tf = float(g.choice([0.5, g.uniform(0.3, 0.7)]))
    mods = [m for m in ("scaled", "offset", "constant", "tiny", "spikes", "cauchy", "pilot_constant", "dups") if g.integers(0, 3) == 0]
    sd = int(g.integers(1, 1 << 30))
    tag = dict(case=case, N=N, M=M, P=P, train_frac=tf, mods=mods, seed=sd)
    try:
        wl = synthetic.Workload(M, P, sd)
        X, Y = wl.rows(0, N)
        X, Y = X.copy(order="F"), Y.copy(order="F")
        r = np.random.default_rng(sd)
        if "scaled" in mods:
            X *= 10.0 ** r.integers(-6, 7, size=M)
        if "offset" in mods:
            c = int(r.integers(0, M))
            X[:, c] += 1e6 * X[:, c].std()
        if "constant" in mods:
            X[:, int(r.integers(0, M))] = -1.0
        if "tiny" in mods:
            c = int(r.integers(0, M))
            X[:, c] = 7.0 + 1e-9 * r.normal(size=N)
        if "spikes" in mods:
            for _ in range(int(r.integers(1, 8))):
                c = int(r.integers(0, M))
                X[int(r.integers(0, N)), c] = X[:, c].mean() + float(10.0 ** r.uniform(1.5, 6.0)) * X[:, c].std() * (1 if r.integers(0, 2) else -1)
            Y[int(r.integers(0, N)), 0] = Y[:, 0].mean() - 400.0 * Y[:, 0].std()
        if "cauchy" in mods:
            c = int(r.integers(0, M))
            X[:, c] = X[:, c].mean() + X[:, c].std() * r.standard_cauchy(size=N)
        if "pilot_constant" in mods:
            c = int(r.integers(0, M))
            X[:, c] = 3.25
            X[1::7, c] = 3.25 + r.normal(size=len(X[1::7, c])) * 1e-3
        if "dups" in mods:
            X[1::2], Y[1::2] = X[0:N - 1:2][:len(X[1::2])], Y[0:N - 1:2][:len(Y[1::2])]
        X, Y = np.asfortranarray(X), np.asfortranarray(Y)
        ntrain = int(round(N * tf))
        shift, sums, G = T._stats_record(ctx, X, Y, ntrain)
        shift2, sums2, G2 = T._stats_record(ctx, X, Y, ntrain)
        problems = []
        if not all(np.array_equal(a, b, equal_nan=True) for a, b in ((shift, shift2), (sums[0], sums2[0]), (sums[1], sums2[1]), (G[0], G2[0]), (G[1], G2[1]))):
            problems.append("repeat not bit-identical")
        Z = np.hstack([X, Y])
        worst = 0.0
        worst_vs_model = 0.0
        on_i8 = (N % 2 == 0)              # (odd row counts stay on the fp64 kernel: held to 1e-13 of sqrt(G_aa G_bb) instead)
        for part, (a, b) in enumerate(((0, ntrain), (ntrain, N))):
            V = Z[a:b] - shift[:C]
            ref = V.T @ V
            sref = V.sum(axis=0)
            dg = np.diag(ref)
            scale = np.sqrt(np.outer(dg, dg)) + 1e-300
            if not np.allclose(sums[part][:C], sref, rtol=1e-10, atol=1e-12 * np.abs(V).sum(axis=0).max()):
                problems.append("column sums, partition %d" % part)
            if not np.allclose(np.diag(G[part])[:C], dg, rtol=1e-11, atol=1e-300):
                problems.append("diagonal, partition %d: %.2e" % (part, float(np.max(np.abs(np.diag(G[part])[:C] - dg) / (dg + 1e-300)))))
            aerr = np.abs(G[part][:M, :C] - ref[:M, :C])
            err = aerr / scale[:M, :C]
            np.fill_diagonal(err[:, :M], 0.0)
            worst = max(worst, float(err.max()))
            bound = gram_error_bound(Z, shift[:C], a, b)[:M, :C] if on_i8 else 1e-13 * scale[:M, :C]
            ratio = aerr / (bound + 1e-300)
            np.fill_diagonal(ratio[:, :M], 0.0)
            worst_vs_model = max(worst_vs_model, float(ratio.max()))
            if not np.allclose(G[part][:M, :C], G[part][:C, :M].T):
                problems.append("not symmetric, partition %d" % part)
        if worst_vs_model > 1.0:
            problems.append("off-diagonal error %.2f x the error model's bound (%.2e of sqrt(G_aa G_bb))" % (worst_vs_model, worst))
        tag.update(worst_offdiag=worst, worst_vs_model=worst_vs_model, problems=problems)
    except Exception as e:        # noqa: BLE001
        tag.update(problems=["exception: %r" % (e,)])
    rows.append(tag)
    if tag["problems"]:
        fails.append(tag)
    print(("FAIL " if tag["problems"] else "ok   ") + json.dumps(tag), flush=True)
json.dump({"cases": len(rows), "failed": len(fails), "worst_offdiag": max((r.get("worst_offdiag", 0.0) for r in rows), default=0.0),
           "worst_vs_model": max((r.get("worst_vs_model", 0.0) for r in rows), default=0.0),
           "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("%d cases, %d with problems" % (len(rows), len(fails)))
