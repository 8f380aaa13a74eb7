"""Cycle stamps of the latency-tuned model fit (k_pls_fit16), for a library built with -DPLS_STAMPS:
    make -C abcsmc_amd/csrc HIPFLAGS="... -DPLS_STAMPS" && python scripts/pls_stamps.py [M P A N]
Prints the cycles a work-group spent in each phase (summed over the components), as stamped by wave 0 / thread 0."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from abcsmc_amd import _lib, abcutil, synthetic

M, P, A, N = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (32, 16, 8, 200000)
ctx = _lib.default_context(0)
wl = synthetic.Workload(M, P, 12345)
X, Y = wl.rows(0, N)
for _ in range(3):
    abcutil.particle_ranking_PLS(X, Y, wl.observed(), 0.5, K=100, max_comp=A, ctx=ctx)
out = (C.c_double * 64)()
_lib.lib().abc_debug_pls_stamps(out)
names = {0: "?0", 1: "barrier after (1)", 2: "(2) eig + w (wave 0), others wait", 3: "(3) |w|, projections + barrier", 4: "(4) r + barrier", 5: "(5) X'X r, XY'r + barrier",
         6: "closing barrier", 7: "PRESS + tail", 8: "(1) tt", 9: "(1) 1/tt", 10: "(1) q, stores", 11: "(1) slab pass + MFMA", 12: "(2a) S from partials, trace", 13: "(2b) squarings", 14: "(2c) argmax, power step, q"}
tot = sum(out[i] for i in range(16))
for i in range(16):
    if out[i]:
        print("%-42s %9.0f cycles  %5.1f %%" % (names.get(i, str(i)), out[i], 100 * out[i] / tot))
print("total %.0f cycles (stamps of thread 0; 100 MHz-class counter or shader clock, see s_memtime)" % tot)
