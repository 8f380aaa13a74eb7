"""Times the one-pass statistics kernel alone (abc_stats_accumulate_dev) on random resident data; used for A/B runs
of k_gram variants (ABC_GRAM_DMA, ABC_GRAM_ABL) without running later stages on possibly invalid statistics.
    python scripts/gram_time.py [N] [M] [P]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from abcsmc_amd import _lib, sharded

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = "cuda:0"
ctx = _lib.default_context(0)
be = sharded.HipBackend(dev, ctx)
g = torch.Generator(device=dev).manual_seed(1)
X = torch.randn((M, N), dtype=torch.float64, device=dev, generator=g)
Y = torch.randn((P, N), dtype=torch.float64, device=dev, generator=g)
stats = be.zeros(be.stats_len(M, P))
be.stats_shift(X, Y, stats)
for _ in range(3):
    be.stats_accumulate(X, Y, 0, N // 2, stats)
torch.cuda.synchronize()
ctx.timing_enable(True)
ctx.timing_read(reset=True)
reps = 30
flush = torch.zeros(int(os.environ.get("GRAM_FLUSH_MB", "0")) * 131072, dtype=torch.float64, device=dev)
for _ in range(reps):
    if flush.numel():
        flush.add_(1.0)          # evict L2 / Infinity Cache / TLBs between launches (cold data, as inside a generation)
    be.stats_accumulate(X, Y, 0, N // 2, stats)
torch.cuda.synchronize()
st = ctx.timing_read(reset=True)
ms = st["k_gram"][0] / st["k_gram"][2]
ov = ctx.timing_overhead(50)
print("N=%d M=%d P=%d  k_gram bracket %.2f us  (overhead %.2f us)  %.0f GB/s bracket, %.0f GB/s corrected  checksum %.6e" % (
    N, M, P, 1e3 * ms, 1e3 * ov, 8.0 * N * (M + P) / (ms * 1e-3) / 1e9, 8.0 * N * (M + P) / ((ms - ov) * 1e-3) / 1e9,
    float(stats.sum().item())))
