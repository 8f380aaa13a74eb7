"""Measured mismatch / fallback rate of the DEVICE build of the resampling table (csrc/alias_dev.hip) on an MI355X: many random
weight vectors of several kinds and sizes through abc_alias_table, every table compared entry by entry (bit patterns of F, values
of A) with the CPU oracle's sequential gsl_ran_discrete_preproc.  Writes one JSON record.
    python tests/fuzz/alias_sweep.py [tables per (kind, size)] [out.json]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abcsmc_amd import _lib          # noqa: E402
from oracle import pyoracle as O     # noqa: E402  (the checker)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
out = sys.argv[2] if len(sys.argv) > 2 else None
ctx = _lib.default_context(0)
kinds = {
    "uniform": lambda g, K: g.random(K),
    "lognormal_sigma1.5": lambda g, K: np.exp(1.5 * g.normal(size=K)),
    "lognormal_sigma3": lambda g, K: np.exp(3.0 * g.normal(size=K)),
    "cubed_with_zeros": lambda g, K: np.where(g.random(K) < 0.02, 0.0, g.random(K) ** 3),
    "few_values": lambda g, K: np.round(g.random(K) * 8) / 8.0 + 0.125,
    "importance_like": lambda g, K: np.exp(-0.5 * (g.normal(size=(K, 4)) ** 2).sum(1)) / (1e-3 + g.random(K)),
    "near_uniform_1e-9": lambda g, K: 1.0 + 1e-9 * g.normal(size=K),
    "near_uniform_1e-4": lambda g, K: 1.0 + 1e-4 * g.normal(size=K),
}
sizes = [20000, 32768, 100000, 250001]
res = {"tables_per_cell": reps, "cells": {}, "total": {"tables": 0, "fallbacks": 0, "wrong_tables": 0}}
t0 = time.time()
for kind, gen in kinds.items():
    for K in sizes:
        g = np.random.default_rng(hash((kind, K)) % (1 << 31) if False else (len(kind) * 1000003 + K))
        nfb = nbad = 0
        for r in range(reps):
            w = gen(g, K)
            w = w / np.linalg.norm(w)
            F, A, on_device = ctx.alias_table(w)
            oF, oA = O.discrete_preproc(w)
            ok = np.array_equal(A, oA) and np.array_equal(F.view(np.uint64), oF.view(np.uint64))
            nfb += 0 if on_device else 1
            nbad += 0 if ok else 1
        res["cells"]["%s/K=%d" % (kind, K)] = {"tables": reps, "host_fallbacks": nfb, "wrong_tables": nbad}
        res["total"]["tables"] += reps
        res["total"]["fallbacks"] += nfb
        res["total"]["wrong_tables"] += nbad
        print("%-22s K=%-7d tables %d  fallbacks %d  wrong %d   (%.0f s)" % (kind, K, reps, nfb, nbad, time.time() - t0), flush=True)
res["seconds"] = round(time.time() - t0, 1)
print(json.dumps(res["total"]))
if out:
    json.dump(res, open(out, "w"), indent=1)
