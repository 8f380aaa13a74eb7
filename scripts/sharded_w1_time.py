"""Wall time per generation of the three ways to run configs[2] on one GPU: abc_generation_dev, the sharded driver without a
communicator, the sharded driver over a one-rank RCCL communicator (what bench.py's scaling_model takes as its base).
    python scripts/sharded_w1_time.py [steps] [config]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from abcsmc_amd import _lib, abcutil, device, sharded, synthetic

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg_id = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = bench.CONFIGS[cfg_id]
N, M, P, A = cfg["N"], cfg["M"], cfg["P"], cfg["A"]
K = N // 10
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
prev = list(wl.previous_set_device(K, dev))


def timed(gen, label):
    rng = abcutil.rng(67890)
    for _ in range(2):
        gen.run(dX, dY, dobs, dpri, rng, *prev)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        gen.run(dX, dY, dobs, dpri, rng, *prev)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / steps


ctx = _lib.default_context(0)
c0 = _lib.Context(0)
c1 = _lib.Context(0)
c1.comm_init_rccl(1, 0, _lib.comm_unique_id())
gens = [("abc_generation_dev", device.Generation(N, M, P, K, K, N, 0.5, A, multivariate=True, device=dev, ctx=ctx)),
        ("abc_generation_sharded_dev, no communicator", sharded.CabiShardedGeneration(c0, dev, N, M, P, K, K, N, 0.5, A, multivariate=True)),
        ("abc_generation_sharded_dev, RCCL world 1", sharded.CabiShardedGeneration(c1, dev, N, M, P, K, K, N, 0.5, A, multivariate=True))]
# interleaved rounds: the pair sums run at a power-limited clock that sags under sustained load, so back-to-back blocks of one
# route each would compare clocks, not drivers
res = {label: [] for label, _ in gens}
for rnd in range(6):
    order = gens[rnd % 3:] + gens[:rnd % 3]           # (every route takes every place in the round)
    for label, gen in order:
        res[label].append(timed(gen, label))
for label, _ in gens:
    v = res[label]
    print("%-48s %s  (best %.4f ms per generation)" % (label, " ".join("%.4f" % x for x in v), min(v)))
c1.comm_destroy()
c1.close()
c0.close()
