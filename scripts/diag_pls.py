"""Diagnostic: time abc_pls_model_dev for varying component counts / response counts."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from abcsmc_amd import _lib, device, synthetic, sharded

dev = "cuda:0"
ctx = _lib.default_context(0)
be = sharded.HipBackend(dev, ctx)
for (M, P, A) in [(32, 16, 1), (32, 16, 2), (32, 16, 4), (32, 16, 8), (32, 1, 8), (32, 4, 4), (64, 32, 8), (32, 16, 16), (128, 16, 32)]:
    wl = synthetic.Workload(M, P)
    X, Y = wl.rows(0, 20000)
    dX, dY, dobs = device.colmajor(X, dev), device.colmajor(Y, dev), device.colmajor(wl.observed(), dev)
    stats = be.zeros(be.stats_len(M, P)); model = be.empty(be.model_len(M, P, A))
    be.stats_shift(dX, dY, stats); be.stats_accumulate(dX, dY, 0, 10000, stats)
    for _ in range(3):
        be.pls_model(stats, dobs, M, P, A, 0, model)
    torch.cuda.synchronize()
    ctx.timing_enable(True); ctx.timing_read(True)
    for _ in range(10):
        be.pls_model(stats, dobs, M, P, A, 0, model)
    t = ctx.timing_read(True)["pls_model"]
    ctx.timing_enable(False)
    print("M=%d P=%d A=%d  pls_model %.1f us  ncomp=%d" % (M, P, A, 1e3 * t[0] / t[2], be.model_ncomp(model, M, P, A)))
