"""Device time of the resampling table's build (HIP-event bracket of its launches) at a given size, log-normal weights.
    python scripts/alias_time.py K [K ...]        (ABC_ALIAS_SMALL_K overrides the size up to which two elements per thread are used)"""
import os
os.environ.setdefault("ABC_DIAG", "1")     # the library reads its diagnostic switches only beside this
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from abcsmc_amd import _lib, abcutil, sharded

dev = "cuda:0"
ctx = _lib.default_context(0)
be = sharded.HipBackend(dev, ctx)
r = abcutil.rng(3)
for K in [int(float(a)) for a in sys.argv[1:]]:
    g = np.random.default_rng(5)
    w = np.exp(1.5 * g.normal(size=K))
    dw = torch.from_numpy(w / np.linalg.norm(w)).to(dev)
    par = be.empty(1024, torch.int64)
    ctx.set_alias_mode(_lib.ALIAS_DEVICE)
    be.resample(r, dw, 0, 1024, par)
    torch.cuda.synchronize()
    ctx.timing_enable(1)
    ctx.timing_read(reset=True)
    for _ in range(10):
        be.resample(r, dw, 0, 1024, par)
    torch.cuda.synchronize()
    st = ctx.timing_read(reset=True)
    ctx.timing_enable(False)
    b, f = ctx.alias_stats(reset=True)
    print("K=%d device alias build %.4f ms (%d builds, %d fallbacks)" % (K, st["alias_host"][0] / max(st["alias_host"][2], 1), b, f))
