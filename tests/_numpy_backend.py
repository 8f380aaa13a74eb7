"""TEST-ONLY numpy stage backend for abcsmc_amd.sharded.ShardedGeneration (CPU tensors), so the
row-sharding logic, global offsets and collectives can be exercised at world_size 2 with gloo on a
machine without a GPU.  It mirrors the stage semantics of include/abcsmc_hip.h using numpy and the
CPU oracle; it is never imported by the product."""
import ctypes as C

import numpy as np
import torch

from abcsmc_amd._lib import lib
from oracle import pyoracle as O


def _np(t):
    return t.numpy()


class NumpyBackend:
    def empty(self, shape, dtype=torch.float64):
        return torch.zeros(shape, dtype=dtype)

    zeros = empty

    def stats_len(self, M, P):
        return lib().abc_stats_len(M, P)          # host-only helper of the product library

    def model_len(self, M, P, A):
        return lib().abc_model_len(M, P, A)

    @staticmethod
    def _lay(M, P):
        C16 = (M + P + 15) // 16 * 16
        o_shift = 2
        o_sum = [o_shift + C16, o_shift + 2 * C16]
        o_G = [o_shift + 3 * C16, o_shift + 3 * C16 + C16 * C16]
        return C16, o_shift, o_sum, o_G

    def stats_shift(self, X, Y, stats):
        M, n = X.shape
        P = Y.shape[0]
        C16, o_shift, _, _ = self._lay(M, P)
        Z = np.concatenate([_np(X), _np(Y)], 0)
        m = min(n, 256)
        s = _np(stats)
        s[o_shift:o_shift + C16] = 0
        s[o_shift:o_shift + M + P] = Z[:, :m].mean(1)

    def stats_accumulate(self, X, Y, row0, ntrain, stats):
        M, n = X.shape
        P = Y.shape[0]
        C16, o_shift, o_sum, o_G = self._lay(M, P)
        s = _np(stats)
        Z = np.concatenate([_np(X), _np(Y)], 0) - s[o_shift:o_shift + M + P, None]
        split = int(min(max(ntrain - row0, 0), n))
        for part, sl in enumerate((slice(0, split), slice(split, n))):
            z = Z[:, sl]
            s[part] = z.shape[1]
            s[o_sum[part]:o_sum[part] + C16] = 0
            s[o_sum[part]:o_sum[part] + M + P] = z.sum(1)
            G = np.zeros((C16, C16))
            G[:M + P, :M + P] = z @ z.T
            s[o_G[part]:o_G[part] + C16 * C16] = G.T.reshape(-1)

    @staticmethod
    def _mlay(M, P, A):
        o = {"hdr": 0, "mean": 4}
        o["sd"] = o["mean"] + M + P
        o["zobs"] = o["sd"] + M + P
        o["oscore"] = o["zobs"] + M
        o["R"] = o["oscore"] + A
        o["Q"] = o["R"] + M * A
        return o

    def pls_model(self, stats, obs, M, P, A, rule, model):
        C16, o_shift, o_sum, o_G = self._lay(M, P)
        s = _np(stats)
        n0, n1 = s[0], s[1]
        n = n0 + n1
        S = [s[o_sum[p]:o_sum[p] + M + P] for p in range(2)]
        G = [s[o_G[p]:o_G[p] + C16 * C16].reshape(C16, C16).T[:M + P, :M + P] for p in range(2)]
        d = (S[0] + S[1]) / n
        ss = np.diag(G[0] + G[1]) - n * d * d
        sd = np.sqrt(np.maximum(ss, 0) / (n - 1))
        mean = s[o_shift:o_shift + M + P] + d

        def zc(p, npart):
            cross = G[p] - np.outer(d, S[p]) - np.outer(S[p], d) + npart * np.outer(d, d)
            den = np.outer(sd, sd)
            return np.where(den > 0, cross / np.where(den > 0, den, 1), 0.0)
        Ztr, Zte = zc(0, n0), zc(1, n1)
        XY, XX = Ztr[:M, M:].copy(), Ztr[:M, :M]
        R, Q, Pm = np.zeros((M, A)), np.zeros((P, A)), np.zeros((M, A))
        for i in range(A):
            if P == 1:
                w = XY[:, 0].copy()
            else:
                ev, V = np.linalg.eigh(XY.T @ XY)
                w = XY @ V[:, -1]
            w /= np.linalg.norm(w)
            r = w.copy()
            for j in range(i):
                r -= (Pm[:, j] @ w) * R[:, j]
            xr = XX @ r
            tt = r @ xr
            p, q = xr / tt, (XY.T @ r) / tt
            XY -= tt * np.outer(p, q)
            R[:, i], Q[:, i], Pm[:, i] = r, q, p
        XYte, XXte, YYte = Zte[:M, M:], Zte[:M, :M], np.diag(Zte[M:, M:])
        per = np.zeros(P, dtype=int)
        for j in range(P):
            best = None
            for a in range(1, A + 1):
                b = R[:, :a] @ Q[j, :a]
                pr = YYte[j] - 2 * b @ XYte[:, j] + b @ XXte @ b
                if best is None or pr < best:
                    best, per[j] = pr, a
        ncomp = int(per.max())
        o = self._mlay(M, P, A)
        m = _np(model)
        m[0], m[1], m[2] = ncomp, A, n
        m[o["mean"]:o["mean"] + M + P] = mean
        m[o["sd"]:o["sd"] + M + P] = sd
        zobs = np.where(sd[:M] > 0, (_np(obs) - mean[:M]) / np.where(sd[:M] > 0, sd[:M], 1), 0.0)
        m[o["zobs"]:o["zobs"] + M] = zobs
        m[o["oscore"]:o["oscore"] + A] = zobs @ R
        m[o["R"]:o["R"] + M * A] = R.T.reshape(-1)
        m[o["Q"]:o["Q"] + P * A] = Q.T.reshape(-1)

    def model_ncomp(self, model, M, P, A):
        return int(_np(model)[0])

    def project_distance(self, X, P, A, model, out):
        M, n = X.shape
        o = self._mlay(M, P, A)
        m = _np(model)
        nc = int(m[0])
        R = m[o["R"]:o["R"] + M * A].reshape(A, M).T
        out.copy_(torch.from_numpy(O.project_distance(_np(X).T, m[o["mean"]:o["mean"] + M], m[o["sd"]:o["sd"] + M],
                                                      R, nc, m[o["oscore"]:o["oscore"] + nc])))

    def select_smallest(self, d, K, idx_base, idx_out, dist_out):
        dn = _np(d)
        o = O.ordered(dn)[:K].astype(np.int64)
        idx_out.copy_(torch.from_numpy(o + idx_base))
        dist_out.copy_(torch.from_numpy(dn[o]))

    # ---- distributed radix select, mirroring k_sel_* (state = [prefix, mask, krem, n_less, ties, ...]) ----
    _SHIFT = [53, 42, 31, 20, 9, 0]
    _WIDTH = [11, 11, 11, 11, 11, 9]

    @staticmethod
    def _keys(d):
        b = _np(d).view(np.uint64)
        neg = (b >> np.uint64(63)).astype(bool)
        return np.where(neg, ~b, b | np.uint64(1 << 63))

    def select_begin(self, K, state, hist):
        state.zero_()
        state[2] = K
        hist.zero_()

    def select_hist(self, d, state, p, hist):
        k = self._keys(d)
        st = _np(state).view(np.uint64)
        m = (k & st[1]) == st[0]
        dig = ((k[m] >> np.uint64(self._SHIFT[p])) & np.uint64((1 << self._WIDTH[p]) - 1)).astype(np.int64)
        hist += torch.from_numpy(np.bincount(dig, minlength=2048).astype(np.int32))

    def select_pick(self, state, p, hist, K):
        st = _np(state).view(np.uint64)
        h = _np(hist).astype(np.int64)
        cum = np.cumsum(h)
        krem = int(st[2])
        dsel = int(np.searchsorted(cum, krem, side="left"))
        below = int(cum[dsel - 1]) if dsel > 0 else 0
        st[0] |= np.uint64(dsel) << np.uint64(self._SHIFT[p])
        st[1] |= np.uint64(((1 << self._WIDTH[p]) - 1) << self._SHIFT[p])
        st[2] = np.uint64(krem - below)
        if p == 5:
            st[4] = st[2]
            st[3] = np.uint64(K - int(st[2]))
        hist.zero_()

    def select_count(self, d, state, counts):
        k = self._keys(d)
        T = _np(state).view(np.uint64)[0]
        counts[0] = int((k < T).sum())
        counts[1] = int((k == T).sum())

    def select_compact(self, d, state, n_less, ties_take, idx_base, idx_out, dist_out):
        k = self._keys(d)
        T = _np(state).view(np.uint64)[0]
        lo = np.nonzero(k < T)[0]
        eq = np.nonzero(k == T)[0][:ties_take]
        sel = np.concatenate([lo, eq]).astype(np.int64)
        assert lo.size == n_less
        idx_out[:sel.size] = torch.from_numpy(sel + idx_base)
        dist_out[:sel.size] = torch.from_numpy(_np(d)[sel])

    def sort_pairs(self, key, idx):
        o = np.argsort(_np(key), kind="stable")
        k, i = _np(key)[o].copy(), _np(idx)[o].copy()
        key.copy_(torch.from_numpy(k))
        idx.copy_(torch.from_numpy(i))

    def merge_runs(self, key, idx, n_runs, run_len, key_out, idx_out):
        k, i = _np(key)[:n_runs * run_len], _np(idx)[:n_runs * run_len]
        for q in range(n_runs):      # precondition of the device kernel: every run sorted by (key, idx)
            kk, ii = k[q * run_len:(q + 1) * run_len], i[q * run_len:(q + 1) * run_len]
            assert np.all((kk[1:] > kk[:-1]) | ((kk[1:] == kk[:-1]) & (ii[1:] >= ii[:-1]))), "run %d not sorted" % q
        o = np.argsort(k, kind="stable")
        key_out.copy_(torch.from_numpy(k[o].copy()))
        idx_out.copy_(torch.from_numpy(i[o].copy()))

    def gather_rows(self, Y, idx, idx_base, theta):
        P, n = Y.shape
        g = _np(idx) - idx_base
        own = (g >= 0) & (g < n)
        _np(theta)[:, own] = _np(Y)[:, g[own]]

    def doubled_variance(self, theta, dv):
        dv.copy_(torch.from_numpy(O.doubled_variance(_np(theta).T)))

    def weights_raw(self, priors, theta, k0, kn, theta_prev, w_prev, dv_prev, out):
        th, tp = _np(theta).T[k0:k0 + kn], _np(theta_prev).T
        dvp, wp = _np(dv_prev), _np(w_prev)
        pr = priors          # ctypes array of oracle priors (test passes them through)
        res = np.zeros(kn)
        sg = np.sqrt(dvp)
        C0 = np.prod(1.0 / (np.sqrt(2 * np.pi) * sg))
        for i in range(kn):
            num = np.prod([O.prior_likelihood(pr[p], th[i, p]) for p in range(th.shape[1])])
            e = (((th[i] - tp) / sg) ** 2).sum(1)
            res[i] = num / (C0 * (wp * np.exp(-0.5 * e)).sum())
        out[:kn].copy_(torch.from_numpy(res))

    def normalize_l2(self, w):
        w /= torch.linalg.norm(w)

    def setup_mvn(self, theta, L):
        rc, Lm, _ = O.mvn_setup(_np(theta).T)
        assert rc == 0
        L.copy_(torch.from_numpy(Lm.T.copy()))

    def resample(self, rng, w, i0, n, parent):
        r = O.Rng(rng.s1, rng.s2, rng.s3)
        lib().abc_rng_jump(C.addressof(r), i0)
        # draws i0..i0+n of the sequential stream (alias table from the full weights)
        parent[:n].copy_(torch.from_numpy(O.resample(r, _np(w), n).astype(np.int64)))

    def perturb(self, rng, theta, priors, parent, i0, n, multivariate, L_or_dv, out, seeds, seed_offset):
        out[:, :n].copy_(theta[:, parent[:n]])          # noise is not part of the CPU orchestration test
        r = O.Rng(rng.s1, rng.s2, rng.s3)
        lib().abc_rng_jump(C.addressof(r), seed_offset + i0)
        seeds[:n].copy_(torch.from_numpy(np.array([O.rng_get(r) for _ in range(n)], dtype=np.int64)))
