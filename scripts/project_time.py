"""Times the PLS ranking (device resident) at a given shape; the stage timers give the projection kernel's share.
    python scripts/project_time.py N M P A        (ABC_PROJECT_VALU=1: the vector-pipe projection for 16 / 32 components)"""
import os
os.environ.setdefault("ABC_DIAG", "1")     # the library reads its diagnostic switches only beside this
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from abcsmc_amd import _lib, abcutil, device, synthetic

N, M, P, A = (int(float(a)) for a in sys.argv[1:5])
dev = "cuda:0"
wl = synthetic.Workload(M, P, seed=12345)
dX, dY = wl.rows_device(0, N, dev)
dobs = device.colmajor(wl.observed(), dev)
dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
ctx = _lib.default_context(0)
gen = device.Generation(N, M, P, N // 10, 0, 0, 0.5, A, multivariate=True, device=dev, ctx=ctx)
rng = abcutil.rng(1)
for _ in range(3):
    gen.run(dX, dY, dobs, dpri, rng)
torch.cuda.synchronize()
ctx.timing_enable(1)
ctx.timing_read(reset=True)
for _ in range(10):
    gen.run(dX, dY, dobs, dpri, rng)
torch.cuda.synchronize()
st = ctx.timing_read(reset=True)
print("N=%d M=%d P=%d A=%d ncomp=%d: projection %.4f ms, Gram %.4f ms, fit %.4f ms" % (N, M, P, A, gen.ncomp.value, st["project_distance"][0] / 10, st["k_gram"][0] / 10, st["pls_model"][0] / 10))
