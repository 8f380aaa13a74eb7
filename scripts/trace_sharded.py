"""The C++ sharded driver (abc_generation_sharded_dev) at world size 1 over a one-rank RCCL communicator, for `rocprofv3 --kernel-trace`:
what every rank of a multi-GPU run executes besides its collectives' wire time (scripts/timeline.py prints the kernel timeline).
    python scripts/trace_sharded.py [config] [set0|full] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from abcsmc_amd import _lib, abcutil, device, sharded, synthetic

cfg = bench.CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 3]
mode = sys.argv[2] if len(sys.argv) > 2 else "full"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, M, P, A = cfg["N"], cfg["M"], cfg["P"], cfg["A"]
K = N // 10
Kp = 0 if mode == "set0" else K
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
prev = list(wl.previous_set_device(Kp, dev)) if Kp else [None, None, None]
ctx = _lib.Context(0)
ctx.comm_init_rccl(1, 0, _lib.comm_unique_id())
gen = sharded.CabiShardedGeneration(ctx, dev, N, M, P, K, Kp, N, 0.5, A, multivariate=True)
rng = abcutil.rng(67890)
for _ in range(steps + 2):
    gen.run(dX, dY, dobs, dpri, rng, *prev)
torch.cuda.synchronize()
ctx.comm_destroy()
ctx.close()
