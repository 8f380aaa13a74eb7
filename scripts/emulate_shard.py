"""Per-rank compute cost of the sharded generation at world size W on ONE GPU, without a second device: the collectives
are replaced by local stand-ins that produce data of the right SHAPE (integer all-reduce multiplies by W, all-gather
replicates this rank's block), so every kernel runs on the sizes rank 0 of W would see (K = 0.1 N_total winners to merge, sort,
gather and resample from; K/W weight rows).  Communication time is NOT included; results are not meaningful numerically.
Used to see which replicated stages grow with W.
    python scripts/emulate_shard.py [W] [config]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dist.is_initialized = lambda: True
dist.get_world_size = lambda group=None: W
dist.get_rank = lambda group=None: 0
dist.broadcast = lambda t, src=0, group=None: None


def _ar(t, op=None, group=None):
    if not t.is_floating_point():      # radix histograms: W identical shards; moments / theta stay this rank's own
        t.mul_(W)


def _ag(out, inp, group=None):
    out.view(W, -1).copy_(inp.view(1, -1).expand(W, -1))


dist.all_reduce = _ar
dist.all_gather_into_tensor = _ag

import bench
from abcsmc_amd import _lib, abcutil, device, sharded, synthetic

c = bench.CONFIGS[cfg]
n_loc, M, P, A = c["N"], c["M"], c["P"], c["A"]
N = n_loc * W
K, Kp = N // 10, n_loc // 10
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
X, Y = wl.rows(0, n_loc)
dX, dY, dobs = device.colmajor(X, dev), device.colmajor(Y, dev), device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
th_prev, w_prev, dv_prev = wl.previous_set(Kp)
dtp, dwp, ddvp = device.colmajor(th_prev, dev), device.colmajor(w_prev, dev), device.colmajor(dv_prev, dev)
rng = abcutil.rng(67890)
ctx = _lib.default_context(0)
# train_frac / W: the train/validation boundary falls inside this rank's rows, as it does on one GPU
gen = sharded.ShardedGeneration(sharded.HipBackend(dev, ctx), n_loc, M, P, K, Kp, n_loc, 0.5 / W, A, rule=_lib.RULE_MIN_PRESS, multivariate=True)
for _ in range(2):
    gen.run(dX, dY, dobs, dpri, rng, dtp, dwp, ddvp)
torch.cuda.synchronize()
ctx.timing_enable(True)
ctx.timing_read(reset=True)
steps = 5
t0 = time.perf_counter()
for _ in range(steps):
    gen.run(dX, dY, dobs, dpri, rng, dtp, dwp, ddvp)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / steps
st = ctx.timing_read(reset=True)
print("W=%d config %d: %.3f ms per step per rank (no communication)" % (W, cfg, ms))
print({k: round((v[0] + v[1]) / steps, 4) for k, v in st.items()})
