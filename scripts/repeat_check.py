"""The same generation many times over (inputs resident, outputs hashed each time): every output must be bit-identical from run
to run -- the two streams of a generation, the side stream's early work and the events between them leave room for races that a
single comparison against the oracle would not show.  The printed hash must also be the same with the streams' orchestration
switches flipped (ABC_FORK_ALWAYS, ABC_WAIT_LATE, ABC_SEEDS_FIRST, ABC_MOMENTS_MAIN, ABC_STATUS_KERNEL) and with the runtime serialising
every launch (AMD_SERIALIZE_KERNEL=3): a dependency the streams' events do not express would show as a different hash there.
    python scripts/repeat_check.py [repeats]"""
import hashlib
import os
os.environ.setdefault("ABC_DIAG", "1")     # the library reads its diagnostic switches only beside this
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from abcsmc_amd import _lib, abcutil, device, synthetic

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
bad = 0
for (N, M, P, K, Kp, Nn, A, mv) in [(100000, 32, 16, 10000, 10000, 100000, 8, True), (400000, 32, 16, 40000, 3000, 200000, 8, True),
                                    (100000, 32, 16, 10000, 0, 100000, 8, True), (60000, 48, 40, 3000, 2000, 50000, 6, False),
                                    (30000, 20, 12, 2500, 2500, 30000, 5, True)]:
    wl = synthetic.Workload(M, P, seed=4242)
    dX, dY = wl.rows_device(0, N, dev)
    dobs = device.colmajor(wl.observed(), dev)
    dpri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
    prev = list(wl.previous_set_device(Kp, dev)) if Kp else []
    gen = device.Generation(N, M, P, K, Kp, Nn, 0.5, A, multivariate=mv, device=dev)
    seen = {}
    for it in range(reps):
        rng = abcutil.rng(777)
        gen.run(dX, dY, dobs, dpri, rng, *prev)
        torch.cuda.synchronize()
        h = hashlib.sha1()
        for t in (gen.idx, gen.dist, gen.theta, gen.w, gen.dv, gen.L if mv else gen.dv, gen.next, gen.parent, gen.seeds):
            h.update(t.cpu().numpy().tobytes())
        seen[h.hexdigest()] = seen.get(h.hexdigest(), 0) + 1
    ok = len(seen) == 1
    bad += 0 if ok else 1
    print("%s N=%d P=%d K=%d K'=%d N+=%d %s: %d runs, %d distinct output sets %s sha1 %s" % ("ok  " if ok else "FAIL", N, P, K, Kp, Nn, "MVN" if mv else "independent", reps, len(seen), "" if ok else sorted(seen.values()), sorted(seen)[0][:16]), flush=True)
print("%d shapes, %d not reproducible" % (5, bad))
