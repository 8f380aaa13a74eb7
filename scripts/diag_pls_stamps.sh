#!/bin/bash
# Diagnostic (not the product): rebuild the library with -DPLS_STAMPS in a scratch dir and print where k_pls_fit
# spends its cycles (phase ids in abcsmc_amd/csrc/pls.hip).  Run on the GPU box from the repo root.
set -e
D=/tmp/abc_stamps; rm -rf $D; mkdir -p $D; cp -r abcsmc_amd include scripts $D/; cd $D/abcsmc_amd/csrc
make clean >/dev/null; make -j8 HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -DPLS_STAMPS" >/dev/null 2>&1
cd $D; python3 - <<'PY'
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
from abcsmc_amd import _lib, device, synthetic, sharded
L = _lib.lib(); dev = "cuda:0"; ctx = _lib.default_context(0); be = sharded.HipBackend(dev, ctx)
for (M, P, A) in [(32, 16, 8), (64, 32, 8), (128, 16, 32)]:
    wl = synthetic.Workload(M, P); X, Y = wl.rows(0, 20000)
    dX, dY, dobs = device.colmajor(X, dev), device.colmajor(Y, dev), device.colmajor(wl.observed(), dev)
    stats = be.zeros(be.stats_len(M, P)); model = be.empty(be.model_len(M, P, A))
    be.stats_shift(dX, dY, stats); be.stats_accumulate(dX, dY, 0, 10000, stats)
    for _ in range(3): be.pls_model(stats, dobs, M, P, A, 0, model)
    out = (C.c_double * 64)(); C.CDLL(_lib.SO_PATH).abc_debug_pls_stamps(out)
    names = {9: "prologue", 1: "S=XY'XY", 2: "eig squaring", 3: "eigvec post + w", 4: "normalise w, r-update", 5: "XX r, p",
             6: "q, deflate", 7: "(loop exit)", 8: "PRESS", 0: ""}
    if 2 <= P <= 32 and M > 16:
        names = {1: "close prev + deflate + partial S", 2: "wave 0: eig + w", 3: "(wait) + |w|, projections", 4: "r update", 5: "X'X r, XY'r partials",
                 6: "(loop exit)", 7: "PRESS + scores", 0: "", 8: "1a tt", 9: "1b 1/tt", 10: "1c q, stores", 11: "1d slabs", 12: "2a sum S, trace", 13: "2b squarings", 14: "2c column, power step, q"}
    tot = sum(out[:16])
    print("M=%d P=%d A=%d total %.0f kcycles" % (M, P, A, tot / 1e3), {names.get(i, i): round(out[i] / 1e3, 1) for i in range(16) if out[i]})
    print("   squaring groups per component:", [int(out[16 + k] + 1) for k in range(min(A, 48))])
PY
