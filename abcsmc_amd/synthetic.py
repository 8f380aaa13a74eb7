"""Synthetic particle x (parameter | metric) workloads.

Counter-based: every value is a pure function of (seed, stream, global row, column), so any shard,
the CPU oracle and the GPU path see exactly the same numbers.  Latent-factor model
    L ~ N(0, I) (N x r, r = min(P, 8));  Y = L A_y + 0.3 E_y;  X = L A_x + 0.3 E_x
followed by a per-column affine rescale s_j = 10^U(-2,3), o_j = s_j U(-5,5) (exercises z-scoring).
Pure numpy (host); used by tests, bench.py and __graft_entry__.smoke().
"""
import numpy as np

from . import _lib

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _u01(seed, stream, rows, cols, sub):
    """uniform in (0,1), shape (len(rows), len(cols))"""
    with np.errstate(over="ignore"):
        key = _mix(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream * 2 + sub + 1))
        r = _mix(rows.astype(np.uint64)[:, None] * np.uint64(0xD1342543DE82EF95) + key)
        x = _mix(r ^ (cols.astype(np.uint64)[None, :] * np.uint64(0xA24BAED4963EE407) + np.uint64(0x9FB21C651E98DF25)))
    return ((x >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _normal(seed, stream, rows, cols):
    u1 = _u01(seed, stream, rows, cols, 0)
    u2 = _u01(seed, stream, rows, cols, 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


class Workload:
    """Column-major-friendly synthetic set.  rows(lo, hi) returns X (n x M), Y (n x P) for global rows."""

    def __init__(self, M, P, seed=12345):
        self.M, self.P, self.seed = int(M), int(P), int(seed)
        self.r = min(self.P, 8)
        k = np.arange(self.r)
        self.Ay = _normal(seed, 1, k, np.arange(self.P))
        self.Ax = _normal(seed, 2, k, np.arange(self.M))
        z = np.zeros(1, dtype=np.int64)
        self.sy = 10.0 ** (-2.0 + 5.0 * _u01(seed, 5, z, np.arange(self.P), 0)[0])
        self.oy = self.sy * (-5.0 + 10.0 * _u01(seed, 5, z, np.arange(self.P), 1)[0])
        self.sx = 10.0 ** (-2.0 + 5.0 * _u01(seed, 6, z, np.arange(self.M), 0)[0])
        self.ox = self.sx * (-5.0 + 10.0 * _u01(seed, 6, z, np.arange(self.M), 1)[0])
        # analytic moments of the parameter columns (shard independent)
        self.mu_y = self.oy
        self.sd_y = self.sy * np.sqrt((self.Ay ** 2).sum(0) + 0.09)

    def rows_by_index(self, rows):
        rows = np.asarray(rows, dtype=np.int64)
        L = _normal(self.seed, 0, rows, np.arange(self.r))
        Y = L @ self.Ay + 0.3 * _normal(self.seed, 3, rows, np.arange(self.P))
        X = L @ self.Ax + 0.3 * _normal(self.seed, 4, rows, np.arange(self.M))
        return np.asfortranarray(X * self.sx + self.ox), np.asfortranarray(Y * self.sy + self.oy)

    def rows(self, lo, hi, chunk=1 << 18):
        X = np.empty((hi - lo, self.M), order="F")
        Y = np.empty((hi - lo, self.P), order="F")
        for a in range(lo, hi, chunk):
            b = min(hi, a + chunk)
            x, y = self.rows_by_index(np.arange(a, b))
            X[a - lo:b - lo], Y[a - lo:b - lo] = x, y
        return X, Y

    def observed(self):
        """observed metrics = metrics of one extra generated row"""
        x, _ = self.rows_by_index(np.array([1 << 40]))
        return x[0].copy()

    def prior_spec(self):
        """even j: ContinuousUniform[mu - 6 sd, mu + 6 sd]; odd j: Gaussian(mu, 3 sd)"""
        spec = []
        for j in range(self.P):
            if j % 2 == 0:
                spec.append((_lib.PRIOR_UNIF_REAL, self.mu_y[j] - 6 * self.sd_y[j], self.mu_y[j] + 6 * self.sd_y[j]))
            else:
                spec.append((_lib.PRIOR_GAUSS, self.mu_y[j], 3 * self.sd_y[j]))
        return spec

    def previous_set(self, Kp):
        """previous generation: an independent draw of Kp parameter rows, uniform weights, dv = 2 Var"""
        _, th = self.rows_by_index((1 << 41) + np.arange(Kp))
        # a posterior is tighter than the prior draw: shrink towards the centre
        th = np.asfortranarray(self.mu_y + 0.5 * (th - self.mu_y))
        w = np.full(Kp, 1.0 / Kp)
        dv = 2.0 * th.var(axis=0, ddof=1)
        return th, w, dv
