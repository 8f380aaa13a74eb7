"""Runs a few whole generations with the stage timers OFF, for `rocprofv3 --kernel-trace`: the timeline (kernel start /
end stamps) then shows GPU-busy time and the gaps between launches.
    python scripts/trace_step.py [config] [set0|full|moved|moved0] [steps]
moved / moved0: the set of bench.moved_count_data (the Wilcoxon rule lowers the largest component count), weighted / first set"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from abcsmc_amd import _lib, abcutil, device, synthetic

cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, M, P, A = cfg["N"], cfg["M"], cfg["P"], cfg["A"]
K = N // 10
Kp = 0 if mode in ("set0", "moved0") else K
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
prev = list(wl.previous_set_device(Kp, dev)) if Kp else []
ctx = _lib.default_context(0)
if mode.startswith("moved"):
    data = bench.moved_count_data(ctx, dX, dY, dobs, N, M, P, K, Kp, A, dev)
    assert data is not None, "the count does not move"
    _, dX, dY, dpri, prev = data
    prev = list(prev)
gen = device.Generation(N, M, P, K, Kp, N, 0.5, A, multivariate=True, device=dev, ctx=ctx)
rng = abcutil.rng(67890)
for _ in range(steps + 2):
    gen.run(dX, dY, dobs, dpri, rng, *prev)
torch.cuda.synchronize()
