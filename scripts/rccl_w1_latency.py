"""Wall time of one small RCCL collective at world size 1 (what the scaling model of bench.py can measure on a one-GPU box): the host
cost of enqueueing it and its time on the stream, through torch.distributed's nccl (= RCCL) backend.
    python scripts/rccl_w1_latency.py"""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29733")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
t = torch.zeros(128, dtype=torch.float64, device=dev)
g = torch.zeros(128, dtype=torch.float64, device=dev)
for name, call in (("broadcast 1 KB", lambda: dist.broadcast(t, 0)), ("all_reduce 1 KB", lambda: dist.all_reduce(t)),
                   ("all_gather 1 KB", lambda: dist.all_gather_into_tensor(g, t))):
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-18s host enqueue %.1f us per call, with the stream drained %.1f us per call" % (name, 1e6 * (t1 - t0) / 500, 1e6 * (t2 - t0) / 500))
dist.destroy_process_group()
