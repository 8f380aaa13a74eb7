"""Times the PLS ranking under the Wilcoxon component rule (device resident) and prints what the rule costs on top of argmin PRESS;
under rocprofv3 --kernel-trace --stats the k_wx_* rows are the reduction's kernels.
    python scripts/wx_time.py [N 1000000] [M 32] [P 16] [A 8] [reps 5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ABC_DIAG", "1")
if len(sys.argv) > 6:
    os.environ["ABC_WX_DEBUG"] = "1"
import torch

from abcsmc_amd import _lib, abcutil, device, synthetic

N, M, P, A, reps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 1_000_000), (2, 32), (3, 16), (4, 8), (5, 5)))
dev = "cuda:0"
ctx = _lib.default_context(0)
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
K = N // 10
out = {}
for name, rule in (("min_press", _lib.RULE_MIN_PRESS), ("wilcoxon", _lib.RULE_WILCOXON)):
    gen = device.Generation(N, M, P, K, 0, 0, 0.5, A, rule=rule, device=dev, ctx=ctx)
    r = abcutil.rng(1)
    for _ in range(2):
        gen.run(dX, dY, dobs, dpri, r)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        gen.run(dX, dY, dobs, dpri, r)
    torch.cuda.synchronize()
    out[name] = (1e3 * (time.perf_counter() - t) / reps, gen.ncomp.value)
print("N=%d M=%d P=%d A=%d: ranking min_press %.3f ms (ncomp %d), wilcoxon %.3f ms (ncomp %d): the rule costs %.3f ms"
      % (N, M, P, A, out["min_press"][0], out["min_press"][1], out["wilcoxon"][0], out["wilcoxon"][1], out["wilcoxon"][0] - out["min_press"][0]))
