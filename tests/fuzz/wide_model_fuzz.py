"""A LOADINGS-LEVEL check of the byte-limb statistics kernel of wide sets (gram.hip: k_gram_i8; the default from 97 columns and
200 000 rows).  tests/fuzz/wide_gram_fuzz.py holds its Gram entries to a bound; BASELINE.json's bar is on what is made of them:
"float within 1e-6 rel for PLS loadings".  Here, per case: the statistics record through the staged entry points under
ABC_GRAM_AUTO and under ABC_GRAM_FP64, abc_pls_model_dev on each, and EVERY USED loading column (the first ncomp columns of R,
ncomp = the count the fit chose under argmin PRESS) against the oracle's own fit of the same rows (particle_ranking_PLS,
AbcUtil.cpp:423-458): ||R_k - R_k(oracle)|| / ||R_k(oracle)|| <= 1e-6, as tests/test_gpu_parity.py::_generation_size_properties
takes it.  Shapes: 97..160 columns split at random into metrics and 1..32 responses, up to 32 components, 200 000 .. 500 000 rows;
data: wide_gram_fuzz.py's kinds (columns scaled over twelve decades, a mean 1e6 sd from zero, a constant column, a tiny-variance
column, spikes, a Cauchy-tailed column, a column constant on the pilot's rows, duplicated rows) plus "lowrank" (the metrics carry
fewer factors than components are fitted: the late components fit noise and their loadings amplify any error in the Gram most).
A case counts as a problem when a used column under ABC_GRAM_AUTO is off by more than 1e-6 AND that is the byte-limb kernel's
doing -- more than twice the fp64 kernels' own distance from the oracle plus 1e-7: a column of variance 1e-18 about a mean of 7
("tiny") leaves its z-scores seven digits in ANY fp64 implementation, the device's fp64 path and the oracle then differ by 2e-6
.. 7e-6 in the last used loading and the byte-limb path by the same amount (`edge_of_fp64` in the output: reported, not a problem of
the kernel under test).  Reported per case and overall: the worst used column under each mode, BY COMPONENT INDEX.
The mode under test is ABC_GRAM_AUTO, the default; FUZZ_I8=1 puts ABC_GRAM_I8 (the byte-limb kernel wherever it can run) in its place,
FUZZ_BIG=1 draws sets of 0.9 .. 1.2e6 rows (450 000 and more in each partition: where the default takes the byte-limb kernel).
WHAT THE RUNS SAID (round 6): at 2e5 .. 5e5 rows (60 000 .. 350 000 a partition) the byte-limb kernel put loading columns 19 and 20 of a
30-component model 4.3e-6 off the oracle's (fp64 kernels: 4e-10) in one of 60 sets (profiles/r06_wide_model_fuzz_i8_small_sets.json:
round 5's default, FUZZ_I8=1 today) -- ABC_GRAM_AUTO therefore takes the kernel only from 400 000 rows in every partition, where
the worst of 64 fuzzed sets is 2.6e-7 (profiles/r06_wide_model_fuzz_big.json: 40 of them; a point-mass column beside 28 responses).
    python tests/fuzz/wide_model_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("ABC_DIAG", "1")
import numpy as np
import torch

from abcsmc_amd import _lib, device, sharded, synthetic
from oracle import pyoracle as O

out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wide_model_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 23
BOUND = 1e-6
ctx = _lib.default_context(0)
dev = "cuda:0"
be = sharded.HipBackend(dev, ctx)
g = np.random.default_rng(seed0)


def fit(dX, dY, dobs, M, P, A, ntrain, mode):
    ctx.set_gram_mode(mode)
    stats = be.zeros(be.stats_len(M, P))
    L = be.model_len(M, P, A)
    model = be.zeros(L + 8)
    be.stats_shift(dX, dY, stats)
    be.stats_accumulate(dX, dY, 0, ntrain, stats)
    which = None
    be.pls_model(stats, dobs, M, P, A, _lib.RULE_MIN_PRESS, model)
    torch.cuda.synchronize()
    m = model.cpu().numpy()
    off_R = 4 + 2 * (M + P) + M + A
    R = np.asfortranarray(m[off_R:off_R + M * A].reshape(A, M).T)
    return int(m[0]), R, which


rows, fails = [], []
by_comp = {"auto": np.zeros(32), "fp64": np.zeros(32)}
BIG = bool(os.environ.get("FUZZ_BIG"))
FORCE_I8 = bool(os.environ.get("FUZZ_I8"))     # the byte-limb kernel wherever it can run (ABC_GRAM_I8) in the "auto" slot: what the default avoids
for case in range(cases):
    mods_force = []
    C = int(g.integers(97, 161))
    P = int(g.integers(1, min(33, C - 80)))
    M = C - P
    A = int(g.integers(2, min(32, M, max(P, 2)) + 1)) if g.integers(0, 3) else min(32, M)
    N = int(g.integers(100_000, 250_001)) * 2
    tf = float(g.choice([0.5, g.uniform(0.3, 0.7)]))
    if BIG:                                   # FUZZ_BIG=1: both partitions of at least 400 000 rows (where ABC_GRAM_AUTO takes the byte-limb kernel)
        N = int(g.integers(450_000, 600_001)) * 2
        tf = float(g.uniform(0.45, 0.55))
        if g.integers(0, 2):
            mods_force = ["lowrank"]
    mods = [m for m in ("scaled", "offset", "constant", "tiny", "spikes", "cauchy", "pilot_constant", "dups", "lowrank") if g.integers(0, 3) == 0]
    mods = sorted(set(mods) | set(mods_force), key=("scaled", "offset", "constant", "tiny", "spikes", "cauchy", "pilot_constant", "dups", "lowrank").index)
    sd = int(g.integers(1, 1 << 30))
    tag = dict(case=case, N=N, M=M, P=P, A=A, train_frac=tf, mods=mods, seed=sd)
    try:
        wl = synthetic.Workload(M, P, sd)
        X, Y = wl.rows(0, N)
        X, Y = X.copy(order="F"), Y.copy(order="F")
        r = np.random.default_rng(sd)
        if "lowrank" in mods:                 # the metrics = 3 factors + noise: components beyond the third fit noise
            B = np.linalg.qr(r.normal(size=(M, 3)))[0]
            mu, s = X.mean(0), X.std(0)
            Z = ((X - mu) / s) @ B
            X = mu + s * (Z @ B.T + 0.3 * r.normal(size=X.shape))
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
        obs = wl.observed()
        ntrain = int(round(N * tf))
        dX, dY, dobs = device.colmajor(X, dev), device.colmajor(Y, dev), device.colmajor(obs, dev)
        t0 = time.time()
        o = O.particle_ranking_pls(X, Y, obs, tf, A, rule=O.RULE_MIN_PRESS)
        tag["oracle_s"] = round(time.time() - t0, 1)
        problems = []
        for name, mode in (("auto", _lib.GRAM_I8 if FORCE_I8 else _lib.GRAM_AUTO), ("fp64", _lib.GRAM_FP64)):
            nc, R, which = fit(dX, dY, dobs, M, P, A, ntrain, mode)
            errs = [float(np.linalg.norm(R[:, k] - o["R"][:, k]) / np.linalg.norm(o["R"][:, k])) for k in range(min(nc, o["ncomp"]))]
            tag[name] = dict(ncomp=nc, kernel=which, worst=max(errs) if errs else 0.0, worst_component=int(np.argmax(errs)) + 1 if errs else 0)
            if nc != o["ncomp"]:
                problems.append("%s: %d components, the oracle %d" % (name, nc, o["ncomp"]))
            tag[name]["errs"] = errs
        tag["oracle_ncomp"] = int(o["ncomp"])
        ea, ef = tag["auto"].pop("errs"), tag["fp64"].pop("errs")
        tag["edge_of_fp64"] = bool(ef and max(ef) > BOUND)
        for k in range(min(len(ea), len(ef))):
            if ea[k] > BOUND and ea[k] > 2.0 * ef[k] + 1e-7:
                problems.append("auto: loading column %d off by %.2e (fp64 kernels: %.2e)" % (k + 1, ea[k], ef[k]))
            if not tag["edge_of_fp64"]:
                by_comp["auto"][k] = max(by_comp["auto"][k], ea[k])
                by_comp["fp64"][k] = max(by_comp["fp64"][k], ef[k])
        tag["problems"] = problems
        del dX, dY, o
    except Exception as e:        # noqa: BLE001
        tag.update(problems=["exception: %r" % (e,)])
    ctx.set_gram_mode(_lib.GRAM_AUTO)
    rows.append(tag)
    if tag["problems"]:
        fails.append(tag)
    print(("FAIL " if tag["problems"] else "ok   ") + json.dumps(tag), flush=True)
clean = [r for r in rows if not r.get("edge_of_fp64")]
json.dump({"cases": len(rows), "failed": len(fails), "bound": BOUND, "edge_of_fp64_cases": len(rows) - len(clean),
           "mode_in_the_auto_slot": "ABC_GRAM_I8 (forced)" if FORCE_I8 else "ABC_GRAM_AUTO", "rows_drawn": "0.9e6 .. 1.2e6" if BIG else "2e5 .. 5e5",
           "worst_auto": max((r.get("auto", {}).get("worst", 0.0) for r in clean), default=0.0),
           "worst_fp64": max((r.get("fp64", {}).get("worst", 0.0) for r in clean), default=0.0),
           "worst_auto_over_all_cases": max((r.get("auto", {}).get("worst", 0.0) for r in rows), default=0.0),
           "worst_fp64_over_all_cases": max((r.get("fp64", {}).get("worst", 0.0) for r in rows), default=0.0),
           "worst_by_component_auto": [float(v) for v in by_comp["auto"]], "worst_by_component_fp64": [float(v) for v in by_comp["fp64"]],
           "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("%d cases, %d with problems" % (len(rows), len(fails)))
