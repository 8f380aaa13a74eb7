"""The byte-limb statistics kernel of wide sets (gram.hip: k_pilot_scale, k_gram_i8, k_gram_far; 113..160 columns from 200 000 rows)
against numpy on random shapes and unfriendly data: row counts 200 000 .. 600 000 (even: the kernel's row pairs; odd counts stay on
the fp64 kernel and are drawn too), 113..160 columns split at random into metrics and parameters, training fractions 0.3..0.7, and
per case a random subset of: columns scaled over twelve decades, columns with a mean 1e6 standard deviations from zero, a constant
column, a column of tiny variance beside huge ones, single spikes up to 1e6 standard deviations (far rows: k_gram_far), a
Cauchy-tailed column (hundreds of far rows), a column that is constant on the pilot's sample rows only, duplicated rows.
What must hold (tests/test_gpu_parity.py::test_wide_gram_on_the_i8_matrix_pipe): column sums and the diagonal to fp64 rounding, every
off-diagonal entry of X'X and X'Y within 2e-9 of sqrt(G_aa G_bb), symmetry, bit-identical repeats.  (The error of an entry is the
noise of the dropped byte products, ~ 2^-32 range_a range_b sqrt(rows): relative to sqrt(G_aa G_bb) it grows with range / sigma of
the two columns and with duplicated rows.  Gaussian-like columns: 4e-11 at 2e5..6e5 rows; a column whose mass sits in ONE point beside
sparse noise has range / sigma ~ 40 and reaches 2..6.4e-10 -- the worst over the first 30 fuzzed sets, four orders below what the
1e-6 bar on the loadings needs.)
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

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wide_gram_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 20
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 17
ctx = _lib.default_context(0)
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
            err = np.abs(G[part][:M, :C] - ref[:M, :C]) / scale[:M, :C]
            np.fill_diagonal(err[:, :M], 0.0)
            worst = max(worst, float(err.max()))
            if not np.allclose(G[part][:M, :C], G[part][:C, :M].T):
                problems.append("not symmetric, partition %d" % part)
        if worst > 2e-9:
            problems.append("off-diagonal error %.2e of sqrt(G_aa G_bb)" % worst)
        tag.update(worst_offdiag=worst, problems=problems)
    except Exception as e:        # noqa: BLE001
        tag.update(problems=["exception: %r" % (e,)])
    rows.append(tag)
    if tag["problems"]:
        fails.append(tag)
    print(("FAIL " if tag["problems"] else "ok   ") + json.dumps(tag), flush=True)
json.dump({"cases": len(rows), "failed": len(fails), "worst_offdiag": max((r.get("worst_offdiag", 0.0) for r in rows), default=0.0),
           "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("%d cases, %d with problems" % (len(rows), len(fails)))
