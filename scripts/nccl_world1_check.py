"""World-size-1 run of the `nccl` (= RCCL) backend on one GPU: every collective abcsmc_amd/sharded.py issues, on tensors of
the dtypes and shapes it uses, plus one ShardedGeneration step -- the RCCL path that gloo tests and one-GPU boxes never
execute (two ranks cannot share a device under RCCL).  It cannot show communication cost or multi-rank correctness (the
gloo world-size-2 tests cover the protocol); it shows that the backend accepts these calls.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 scripts/nccl_world1_check.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from abcsmc_amd import _lib, abcutil, device, sharded, synthetic

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = "cuda:0"
dist.init_process_group("nccl", device_id=torch.device(dev))
W = dist.get_world_size()
stats = torch.arange(4000, dtype=torch.float64, device=dev)
dist.broadcast(stats[2:50], src=0)
dist.all_reduce(stats, op=dist.ReduceOp.SUM)
hist = torch.ones(2048, dtype=torch.int32, device=dev)
dist.all_reduce(hist, op=dist.ReduceOp.SUM)
counts = torch.tensor([3, 4], dtype=torch.int64, device=dev)
allc = torch.zeros(2 * W, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(allc, counts)
idx = torch.arange(1000, dtype=torch.int64, device=dev)
cidx = torch.empty(1000 * W, dtype=torch.int64, device=dev)
dist.all_gather_into_tensor(cidx, idx[:1000])
d = torch.rand(1000, dtype=torch.float64, device=dev)
cd = torch.empty(1000 * W, dtype=torch.float64, device=dev)
dist.all_gather_into_tensor(cd, d)
theta = torch.rand((16, 1000), dtype=torch.float64, device=dev)
dist.all_reduce(theta, op=dist.ReduceOp.SUM)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert allc.tolist() == [3, 4] and torch.equal(cidx, idx) and torch.equal(cd, d) and hist.sum().item() == 2048

# one sharded generation under the nccl process group (world 1: same answer as the fused driver)
N, M, P, K, Kp, A = 20000, 32, 16, 2000, 2000, 8
wl = synthetic.Workload(M, P, seed=12345)
X, Y = wl.rows(0, N)
dX, dY, dobs = device.colmajor(X, dev), device.colmajor(Y, dev), device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
prev = [device.colmajor(a, dev) for a in wl.previous_set(Kp)]
ctx = _lib.default_context(0)
sg = sharded.ShardedGeneration(sharded.HipBackend(dev, ctx), N, M, P, K, Kp, N, 0.5, A, rule=_lib.RULE_MIN_PRESS, multivariate=True)
sg.run(dX, dY, dobs, dpri, abcutil.rng(67890), *prev)
fg = device.Generation(N, M, P, K, Kp, N, 0.5, A, multivariate=True, device=dev, ctx=ctx)
fg.run(dX, dY, dobs, dpri, abcutil.rng(67890), *prev)
torch.cuda.synchronize()
assert torch.equal(sg.idx, fg.idx) and torch.equal(sg.parent, fg.parent) and torch.equal(sg.w, fg.w)
print("nccl world-1 check ok: backend %s, collectives accepted, sharded step == fused step" % dist.get_backend())
dist.destroy_process_group()
