"""Prototype (round 6): can a PERTURBATION PROBE tell when the byte-limb statistics (k_gram_i8) are not good enough for the loadings?
Per set: the record under ABC_GRAM_AUTO and ABC_GRAM_FP64, the fit of each (real error of every used loading column = AUTO against
FP64), and the fit of the AUTO record with every off-diagonal Gram entry moved by +-kappa x 2^-32 range_a range_b sqrt(rows)
(random symmetric signs): the probe's deviation from the unperturbed AUTO fit.
    python scripts/gram_probe_proto.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from abcsmc_amd import _lib, device, sharded, synthetic
from _gram_model import pilot_range

ctx = _lib.default_context(0)
dev = "cuda:0"
be = sharded.HipBackend(dev, ctx)


def fit_from(stats, dobs, M, P, A):
    L = be.model_len(M, P, A)
    model = be.zeros(L + 8)
    be.pls_model(stats, dobs, M, P, A, _lib.RULE_MIN_PRESS, model)
    torch.cuda.synchronize()
    m = model.cpu().numpy()
    off_R = 4 + 2 * (M + P) + M + A
    return int(m[0]), np.asfortranarray(m[off_R:off_R + M * A].reshape(A, M).T)


def coldiff(Ra, Rb, nc):
    return [min(np.linalg.norm(Ra[:, k] - Rb[:, k]), np.linalg.norm(Ra[:, k] + Rb[:, k])) / np.linalg.norm(Rb[:, k]) for k in range(nc)]


def run(tag, X, Y, obs, A, tf):
    N, M = X.shape
    P = Y.shape[1]
    C = M + P
    C16 = 16 * ((C + 15) // 16)
    ntrain = int(round(N * tf))
    dX, dY, dobs = device.colmajor(X, dev), device.colmajor(Y, dev), device.colmajor(obs, dev)
    recs = {}
    for name, mode in (("auto", _lib.GRAM_AUTO), ("fp64", _lib.GRAM_FP64)):
        ctx.set_gram_mode(mode)
        st = be.zeros(be.stats_len(M, P))
        be.stats_shift(dX, dY, st)
        be.stats_accumulate(dX, dY, 0, ntrain, st)
        torch.cuda.synchronize()
        recs[name] = st
    ctx.set_gram_mode(_lib.GRAM_AUTO)
    nc, Ra = fit_from(recs["auto"], dobs, M, P, A)
    ncf, Rf = fit_from(recs["fp64"], dobs, M, P, A)
    real = coldiff(Ra, Rf, min(nc, ncf))
    st = recs["auto"].cpu().numpy()
    shift = st[2:2 + C16][:C]
    rng_ = pilot_range(np.hstack([X, Y]), shift)
    g = np.random.default_rng(1)
    out = []
    for kappa in (0.25, 1.0):
        st2 = st.copy()
        for part, n in ((0, ntrain), (1, N - ntrain)):
            o = 2 + 3 * C16 + part * C16 * C16
            G = st2[o:o + C16 * C16].reshape(C16, C16)
            s = np.sign(g.normal(size=(C, C)))
            s = np.triu(s, 1)
            s = s + s.T
            G[:C, :C] += kappa * 2.0 ** -32 * np.outer(rng_, rng_) * np.sqrt(n) * s
        ncp, Rp = fit_from(torch.from_numpy(st2).to(dev), dobs, M, P, A)
        out.append((kappa, max(coldiff(Rp, Ra, min(nc, ncp)))))
    print("%-28s ncomp %2d/%2d  real error: worst %.2e (component %d)   probe: %s" % (
        tag, nc, ncf, max(real), int(np.argmax(real)) + 1, ", ".join("kappa %.2f -> %.2e" % o for o in out)), flush=True)


# the fuzzer's failing case (profiles/r06_wide_model_fuzz.json, case 30)
sd = 123565939
M, P, A, N, tf = 87, 29, 32, 220388, 0.30006042464615873
wl = synthetic.Workload(M, P, sd)
X, Y = wl.rows(0, N)
r = np.random.default_rng(sd)
B = np.linalg.qr(r.normal(size=(M, 3)))[0]
mu, s = X.mean(0), X.std(0)
X = mu + s * ((((X - mu) / s) @ B) @ B.T + 0.3 * r.normal(size=X.shape))
X *= 10.0 ** r.integers(-6, 7, size=M)
run("fuzz case 30 (lowrank)", np.asfortranarray(X), np.asfortranarray(Y), wl.observed(), A, tf)
for (M, P, A, N, seed, name) in ((128, 16, 32, 1_000_000, 12345, "configs[4]"), (64, 32, 8, 2_000_000, 12345, "configs[3] shape, 2e6 rows"),
                                 (128, 16, 32, 200_000, 21, "128x16x32, 2e5 rows"), (100, 30, 32, 300_000, 5, "100x30x32, 3e5 rows"),
                                 (140, 20, 16, 250_000, 6, "140x20x16")):
    wl = synthetic.Workload(M, P, seed)
    X, Y = wl.rows(0, N)
    run(name, X, Y, wl.observed(), A, 0.5)
