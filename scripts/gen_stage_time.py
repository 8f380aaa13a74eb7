"""Whole generations at a shape given on the command line, stage timers on: ms per stage and per generation.
    python scripts/gen_stage_time.py N M P A [K] [steps]          (K = K' = N / 10 by default, N+ = N, multivariate noise)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from abcsmc_amd import _lib, abcutil, device, synthetic

N, M, P, A = (int(a) for a in sys.argv[1:5])
K = int(sys.argv[5]) if len(sys.argv) > 5 else N // 10
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
prev = list(wl.previous_set_device(K, dev))
ctx = _lib.default_context(0)
gen = device.Generation(N, M, P, K, K, N, 0.5, A, multivariate=True, device=dev, ctx=ctx)
rng = abcutil.rng(67890)
for _ in range(3):
    gen.run(dX, dY, dobs, dpri, rng, *prev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    gen.run(dX, dY, dobs, dpri, rng, *prev)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / steps
ctx.timing_enable(True)
ctx.timing_read(reset=True)
for _ in range(steps):
    gen.run(dX, dY, dobs, dpri, rng, *prev)
torch.cuda.synchronize()
st = ctx.timing_read(reset=True)
print("N=%d M=%d P=%d A=%d K=K'=%d: %.3f ms per generation (timers off); stages (timers on, serialising): %s" % (
    N, M, P, A, K, ms, ", ".join("%s %.3f" % (k, v[0] / steps) for k, v in st.items() if v[0] > 0)))
