"""The Wilcoxon component rule (wilcoxon.hip) against the CPU oracle on sets LARGE ENOUGH for its binned path and its bounds sweep
(16384 validation rows and more: the bounds cascade of round 5; tests/fuzz/ranking_fuzz.py stays below that and only sees the sorted path): random row counts
(20 000 .. 600 000), 4..48 metrics, 1..12 responses, 2..32 components, training fractions 0.3..0.7, and responses whose noise is
drawn so that the tests of a case are a MIX of decisive ones (settled by the bounds) and ones with statistics next to the threshold
(undecided: the exact sweeps); tie structures: duplicated validation rows, validation rows drawn from few distinct rows (tie groups
that fill bins and, with very few, outgrow them: the repeat on the sorted path), a response that is constant on the validation rows
(zero differences only), responses rounded to a grid (ties among the |d|).
What must hold: the per-response component counts the device leaves in the model record equal the oracle's reduction run on the
device's own model (tests/test_gpu_parity.py::_wilcoxon_per_response: same residual bits on both sides, so every rank sum and
every verdict must agree), and the PRESS optima it starts from equal the oracle's.
    python tests/fuzz/wilcoxon_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ABC_DIAG", "1")
os.environ["ABC_WX_DEBUG"] = "1"           # the library then says on stderr how the tests of a reduction were settled
import re
import tempfile

import numpy as np

from abcsmc_amd import _lib, synthetic
from oracle import pyoracle as oracle
import test_gpu_parity as T

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wilcoxon_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 9
ctx = _lib.default_context(0)
g = np.random.default_rng(seed0)

rows, fails = [], []
for case in range(cases):
    N = int(g.choice([36_000, 45_000, 70_000, 150_000, 300_000, 600_000])) + int(g.integers(0, 999))
    M = int(g.integers(4, 49))
    P = int(g.integers(1, 13))
    A = int(g.choice([2, 3, 5, 8, 8, 12, 16, 24, 32]))
    A = min(A, M)
    if N * P * A > 4e7:                    # (the oracle sorts every test: keep a case at a few seconds)
        P = max(1, int(4e7 / (N * A)))
    tf = float(g.choice([0.5, 0.5, g.uniform(0.3, 0.7)]))
    kind = str(g.choice(["plain", "plain", "pairs", "copies400", "copies40", "copies6", "constant", "grid"]))
    noise = float(g.choice([0.0, 0.3, 1.0, 1.5, 3.0, 10.0]))
    sd = int(g.integers(1, 1 << 30))
    tag = dict(case=case, N=N, M=M, P=P, A=A, train_frac=tf, kind=kind, noise=noise, seed=sd)
    try:
        wl = synthetic.Workload(M, P, sd)
        X, Y = wl.rows(0, N)
        X, Y = X.copy(order="F"), Y.copy(order="F")
        obs = wl.observed().copy()
        r = np.random.default_rng(sd)
        if noise:
            Y = np.asfortranarray(Y + r.normal(size=Y.shape) * Y.std(0) * noise)
        nt0 = int(round(N * tf))
        if kind == "pairs":
            X[nt0 + 1:N:2], Y[nt0 + 1:N:2] = X[nt0:N - 1:2], Y[nt0:N - 1:2]
        elif kind.startswith("copies"):
            c = int(kind[6:])
            src = nt0 + (np.arange(N - nt0) % c)
            X[nt0:], Y[nt0:] = X[src], Y[src]
        elif kind == "constant":
            Y[nt0:, 0] = Y[nt0, 0]
        elif kind == "grid":
            Y = np.asfortranarray(np.round(Y / (Y.std(0) * 0.05)) * (Y.std(0) * 0.05))
        X, Y = np.asfortranarray(X), np.asfortranarray(Y)
        sys.stderr.flush()
        with tempfile.TemporaryFile(mode="w+b") as cap:           # (the C library's stderr: file descriptor 2)
            keep = os.dup(2)
            os.dup2(cap.fileno(), 2)
            try:
                per_press, per_wx, o_press, o_wx, ncomp = T._wilcoxon_per_response(ctx, oracle, X, Y, obs, A, f=tf)
            finally:
                os.dup2(keep, 2)
                os.close(keep)
            cap.seek(0)
            said = cap.read().decode(errors="replace")
        m = re.search(r"rejected (\d+), passed (\d+), undecided (\d+) by the bounds \(exact step for (\d+) tests, last level (\d+) bins\)(.*)", said)
        if m:
            tag.update(bounds_rejected=int(m.group(1)), bounds_passed=int(m.group(2)), undecided=int(m.group(3)), exact_tests=int(m.group(4)),
                       last_level_bins=int(m.group(5)), sorted_repeat="repeat" in m.group(6))
        problems = []
        if not np.array_equal(per_press, o_press):
            problems.append("PRESS optima %s vs the oracle's %s" % (per_press.tolist(), o_press.tolist()))
        elif not np.array_equal(per_wx, o_wx):
            problems.append("reduced counts %s vs the oracle's %s (PRESS optima %s)" % (per_wx.tolist(), o_wx.tolist(), per_press.tolist()))
        elif ncomp != int(o_wx.max()):
            problems.append("ncomp %d vs %d" % (ncomp, int(o_wx.max())))
        tag.update(tests=int(np.maximum(per_press - 1, 0).sum()), reduced=int((per_wx < per_press).sum()), problems=problems)
    except Exception as e:        # noqa: BLE001
        tag.update(problems=["exception: %r" % (e,)])
    rows.append(tag)
    if tag["problems"]:
        fails.append(tag)
    print(("FAIL " if tag["problems"] else "ok   ") + json.dumps(tag), flush=True)
tot = {k: sum(r.get(k, 0) for r in rows) for k in ("tests", "bounds_rejected", "bounds_passed", "undecided", "exact_tests")}
tot["cases_repeated_on_the_sorted_path"] = sum(1 for r in rows if r.get("sorted_repeat"))
json.dump({"cases": len(rows), "failed": len(fails), "totals": tot, "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("totals", json.dumps(tot))
print("%d cases, %d with problems" % (len(rows), len(fails)))
