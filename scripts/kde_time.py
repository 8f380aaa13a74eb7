"""Times the pair-sum kernel of the importance weights alone (abc_weights_raw_dev) on a synthetic posterior pair; used
for A/B runs of the split-operand kernel (ABC_KDE_G = exponentials per scheduling slot) against the fp64 kernel.
    python scripts/kde_time.py [K] [Kp] [P] [fp64]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from abcsmc_amd import _lib, device, sharded, synthetic

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
Kp = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
fp64 = len(sys.argv) > 4 and sys.argv[4] == "fp64"
dev = "cuda:0"
ctx = _lib.default_context(0)
ctx.set_kde_mode(_lib.KDE_FP64 if fp64 else _lib.KDE_AUTO)
be = sharded.HipBackend(dev, ctx)
wl = synthetic.Workload(8, P, seed=12345)
_, th = wl.rows(0, K)
th = np.asfortranarray(wl.mu_y + 0.5 * (th - wl.mu_y))        # a posterior: as tight as the previous one
tp, wp, dv = wl.previous_set(Kp)
dth, dtp, dwp, ddv = (device.colmajor(a, dev) for a in (th, tp, wp, dv))
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
out = be.empty(K)
for _ in range(2):
    be.weights_raw(dpri, dth, 0, K, dtp, dwp, ddv, out)
torch.cuda.synchronize()
ctx.timing_enable(True)
ctx.timing_read(reset=True)
reps = int(os.environ.get("KDE_REPS", "10"))      # ~60 reaches the sustained (power-limited) clock
for _ in range(reps):
    be.weights_raw(dpri, dth, 0, K, dtp, dwp, ddv, out)
torch.cuda.synchronize()
st = ctx.timing_read(reset=True)
ms = st["k_kde"][0] / reps
misc = st["weights_misc"][0] / reps - ms
print("K=%d K'=%d P=%d %s G=%s: k_kde %.3f ms (%.2f ps/pair), other weight kernels %.3f ms, checksum %.12e" % (
    K, Kp, P, "fp64" if fp64 else "split", os.environ.get("ABC_KDE_G", "1"), ms, 1e9 * ms / (K * Kp), misc,
    float(out.sum().item())))
