"""sha256 of the multivariate proposals of a fixed set at the parameter counts given (A/B of two builds: ABCSMC_HIP_SO)
    python scripts/perturb_hash.py 33 40 48 64"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from abcsmc_amd import _lib, abcutil

ctx = _lib.default_context(0)
for P in (int(a) for a in sys.argv[1:]):
    g = np.random.default_rng(P)
    K, n = 700, 50000
    th = g.normal(size=(K, P)) * (1.0 + 0.1 * np.arange(P)) + g.normal(size=(K, 1))
    spec = [(_lib.PRIOR_UNIF_REAL, -3.0 - 0.3 * p, 3.0 + 0.3 * p) if p % 2 else (_lib.PRIOR_GAUSS, 0.0, 4.0) for p in range(P)]
    w = g.random(K)
    L = abcutil.setup_mvn_sampler(th, ctx=ctx)
    out, parent = abcutil.sample_mvn_predictive_priors(abcutil.rng(5), n, w, th, _lib.make_priors(spec), L, ctx=ctx)[:2]
    print("P=%d proposals %s parents %s" % (P, hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16],
                                          hashlib.sha256(np.ascontiguousarray(parent).tobytes()).hexdigest()[:16]))
