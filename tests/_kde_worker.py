"""Worker of test_weight_split_kernel_staging_does_not_change_the_sums: the importance weights of fixed sets at the parameter counts
given on the command line, one line per count: P, sha256 of the K weights' bytes, and the weights' sum to 17 digits.  The switches
that pick the kernel (ABC_KDE_LDS, ABC_KDE_CHUNKS3 under ABC_DIAG) are read once per process, hence a process per setting."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from abcsmc_amd import _lib, abcutil, synthetic

ctx = _lib.default_context(0)
for P in (int(a) for a in sys.argv[1:]):
    K, Kp = 1500 + P, 4100 - P                    # (ragged last tiles on both sides; an odd number of previous tiles per slice somewhere)
    wl = synthetic.Workload(8, P, 1000 + P)
    _, th = wl.rows(0, K)
    th = np.asfortranarray(wl.mu_y + 0.4 * (th - wl.mu_y))
    tp, wp, dv = wl.previous_set(Kp)
    wp = np.random.default_rng(P).random(Kp)
    wp[::89] = 0.0
    wp /= np.linalg.norm(wp)
    w = abcutil.weight_predictive_prior(_lib.make_priors(wl.prior_spec()), th, tp, wp, dv, ctx=ctx)
    assert ctx.kde_last_kernel() == _lib.KDE_RAN_SPLIT
    print("KDE %d %s %.17e" % (P, hashlib.sha256(np.ascontiguousarray(w).tobytes()).hexdigest(), float(w.sum())))
