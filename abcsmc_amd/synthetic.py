"""Synthetic particle x (parameter | metric) workloads.

Counter-based: every value is a pure function of (seed, stream, global row, column), so any shard,
the CPU oracle and the GPU path see exactly the same numbers.  Latent-factor model
    L ~ N(0, I) (N x r, r = min(P, 8));  Y = L A_y + 0.3 E_y;  X = L A_x + 0.3 E_x
followed by a per-column affine rescale s_j = 10^U(-2,3), o_j = s_j U(-5,5) (exercises z-scoring).
Host generator: pure numpy; used by tests, bench.py and __graft_entry__.smoke().  `Workload.rows_device` /
`previous_set_device` are the same counter-based construction evaluated with torch on the GPU (input plumbing for the sizes
numpy takes minutes on: 1e7 rows); integer hashing is bit-identical to the host's, the normal deviates differ from numpy's in
the last bits of log / cos, so a device-generated set is downloaded, not regenerated, when the CPU oracle needs it.
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


def _s64(c):
    """a 64-bit constant as the signed value torch's int64 holds (wrap-around arithmetic is the same bits)"""
    c &= 0xFFFFFFFFFFFFFFFF
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(x, k):
    return (x >> k) & ((1 << (64 - k)) - 1)          # logical shift of an int64 tensor


def _mix_t(x):
    x = (x ^ _lsr(x, 30)) * _s64(0xBF58476D1CE4E5B9)
    x = (x ^ _lsr(x, 27)) * _s64(0x94D049BB133111EB)
    return x ^ _lsr(x, 31)


def _u01_t(seed, stream, rows, ncols, sub, device):
    """_u01 on the device: rows int64 tensor (m,), columns 0..ncols-1 -> float64 (ncols, m)"""
    import torch
    with np.errstate(over="ignore"):
        key = int(_mix(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream * 2 + sub + 1)))
    r = _mix_t(rows * _s64(0xD1342543DE82EF95) + _s64(key))
    cols = torch.arange(ncols, dtype=torch.int64, device=device) * _s64(0xA24BAED4963EE407) + _s64(0x9FB21C651E98DF25)
    x = _mix_t(r[None, :] ^ cols[:, None])
    return (_lsr(x, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _normal_t(seed, stream, rows, ncols, device):
    import torch
    u1 = _u01_t(seed, stream, rows, ncols, 0, device)
    u2 = _u01_t(seed, stream, rows, ncols, 1, device)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * np.pi * u2)


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

    # ---- the same construction on the GPU (torch as plumbing: elementwise integer / float ops only) ---------------
    def rows_device(self, lo, hi, device, want_x=True, chunk=1 << 20):
        """(X, Y) as torch float64 column-major holders of shape (M, n) and (P, n) on `device`; X is None without want_x"""
        import torch
        n = hi - lo
        X = torch.empty((self.M, n), dtype=torch.float64, device=device) if want_x else None
        Y = torch.empty((self.P, n), dtype=torch.float64, device=device)
        tn = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        Ay, Ax, sy, oy, sx, ox = tn(self.Ay), tn(self.Ax), tn(self.sy), tn(self.oy), tn(self.sx), tn(self.ox)
        for a in range(lo, hi, chunk):
            b = min(hi, a + chunk)
            rows = torch.arange(a, b, dtype=torch.int64, device=device)
            L = _normal_t(self.seed, 0, rows, self.r, device)                   # (r, m)
            y = 0.3 * _normal_t(self.seed, 3, rows, self.P, device)
            for k in range(self.r):
                y.addcmul_(Ay[k][:, None], L[k][None, :])
            Y[:, a - lo:b - lo] = y * sy[:, None] + oy[:, None]
            if want_x:
                x = 0.3 * _normal_t(self.seed, 4, rows, self.M, device)
                for k in range(self.r):
                    x.addcmul_(Ax[k][:, None], L[k][None, :])
                X[:, a - lo:b - lo] = x * sx[:, None] + ox[:, None]
        return X, Y

    def previous_set_device(self, Kp, device):
        """previous_set on the GPU: (theta_prev (P, Kp), w_prev (Kp,), dv_prev (P,)) torch float64 tensors"""
        import torch
        _, th = self.rows_device(1 << 41, (1 << 41) + Kp, device, want_x=False)
        mu = torch.from_numpy(self.mu_y).to(device)[:, None]
        th = (mu + 0.5 * (th - mu)).contiguous()
        w = torch.full((Kp,), 1.0 / Kp, dtype=torch.float64, device=device)
        dv = 2.0 * th.var(dim=1, unbiased=True)
        return th, w, dv

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
