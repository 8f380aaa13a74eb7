"""Times the generation on bench.moved_count_data's set (the Wilcoxon rule lowers the largest count) under the rule and under argmin
PRESS, weighted and first set -- for A/B runs of the diagnostic switches:   python scripts/moved_time.py [config] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from abcsmc_amd import _lib, abcutil, device, synthetic

cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, M, P, A = cfg["N"], cfg["M"], cfg["P"], cfg["A"]
K = Kp = N // 10
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
ctx = _lib.default_context(0)
data = bench.moved_count_data(ctx, dX, dY, dobs, N, M, P, K, Kp, A, dev)
assert data is not None
_, dX, dY, dpri, prev = data
rng = abcutil.rng(67890)
out = []
for kp, pv in ((Kp, prev), (0, ())):
    for rule in (_lib.RULE_WILCOXON, _lib.RULE_MIN_PRESS):
        g = device.Generation(N, M, P, K, kp, N, 0.5, A, rule=rule, multivariate=True, device=dev, ctx=ctx)
        for _ in range(3):
            g.run(dX, dY, dobs, dpri, rng, *pv)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            g.run(dX, dY, dobs, dpri, rng, *pv)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t) / reps)
print("moved config %s: weighted %.4f (press %.4f: +%.4f)   first set %.4f (press %.4f: +%.4f)" % (
    sys.argv[1] if len(sys.argv) > 1 else "3", out[0], out[1], out[0] - out[1], out[2], out[3], out[2] - out[3]))
