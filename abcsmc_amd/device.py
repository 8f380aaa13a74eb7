"""Device-resident driver: one SMC generation turn-over with all inputs and outputs in HBM.

torch is used ONLY as plumbing here -- device allocations (torch tensors) and the current HIP
stream.  All arithmetic is done by libabcsmc_hip.so through `abc_generation_dev` (and the stage-level
`*_dev` entry points used by sharded.py).  Matrices are column-major: an (n, c) matrix is held as a
contiguous torch tensor of shape (c, n), i.e. one particle-major vector per column.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import GenerationCfg, GenerationIO, Rng, lib


def colmajor(a, device):
    """numpy (n, c) -> torch (c, n) contiguous float64 on device (column-major storage)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return torch.from_numpy(np.ascontiguousarray(a.T)).to(device)


def to_numpy(t):
    """torch (c, n) column-major holder -> numpy (n, c)."""
    a = t.detach().cpu().numpy()
    return a.T if a.ndim == 2 else a


def priors_to_device(priors, device):
    raw = np.frombuffer(bytes(priors), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Generation:
    """Pre-allocated buffers + one call per generation (AbcSmc.cpp:634-664, 1041-1066, 490-518)."""

    def __init__(self, N, M, P, K, Kp, Nnext, train_frac=0.5, max_comp=0, rule=_lib.RULE_DEFAULT,
                 multivariate=True, device="cuda:0", ctx=None):
        self.device = torch.device(device)
        idx = self.device.index or 0
        self.ctx = ctx if ctx is not None else _lib.default_context(idx)
        self.cfg = GenerationCfg(N, M, P, K, Kp, Nnext, float(train_frac), int(max_comp), int(rule),
                                 int(bool(multivariate)), 0)
        f64, i64 = torch.float64, torch.int64
        d = self.device
        self.idx = torch.empty(K, dtype=i64, device=d)
        self.dist = torch.empty(K, dtype=f64, device=d)
        self.theta = torch.empty((P, K), dtype=f64, device=d)
        self.w = torch.empty(K, dtype=f64, device=d)
        self.dv = torch.empty(P, dtype=f64, device=d)
        self.L = torch.empty((P, P), dtype=f64, device=d)
        self.next = torch.empty((P, max(Nnext, 1)), dtype=f64, device=d)
        self.parent = torch.empty(max(Nnext, 1), dtype=i64, device=d)
        self.seeds = torch.empty(max(Nnext, 1), dtype=i64, device=d)
        self.ncomp = C.c_int32(0)
        self._io_key, self._io, self._args = None, None, None
        self._call = lib().abc_generation_dev

    def run(self, X, Y, obs, priors_dev, rng, theta_prev=None, w_prev=None, dv_prev=None):
        """X: (M, N), Y: (P, N), obs: (M,), all float64 on self.device; rng: _lib.Rng (advanced)."""
        cfg = self.cfg
        weighted = theta_prev is not None and cfg.Kp
        # (addresses AND shapes: torch's caching allocator hands the same address to a differently shaped tensor)
        key = (X.data_ptr(), Y.data_ptr(), obs.data_ptr(), priors_dev.data_ptr(),
               theta_prev.data_ptr() if weighted else 0, w_prev.data_ptr() if weighted else 0,
               dv_prev.data_ptr() if weighted else 0, X.shape, Y.shape, X.stride(), Y.stride(), obs.shape, priors_dev.shape,
               theta_prev.shape if weighted else None, theta_prev.stride() if weighted else None,
               w_prev.shape if weighted else None, dv_prev.shape if weighted else None)
        if key != self._io_key:           # (the argument block of the C call is rebuilt only when a buffer moved)
            assert X.shape == (cfg.M, cfg.N) and Y.shape == (cfg.P, cfg.N) and X.is_contiguous() and Y.is_contiguous()
            assert obs.numel() == cfg.M and obs.is_contiguous() and priors_dev.numel() >= 24 * cfg.P
            io = GenerationIO()
            io.X, io.Y, io.obs, io.priors = key[0], key[1], key[2], key[3]
            if weighted:
                assert theta_prev.shape == (cfg.P, cfg.Kp) and theta_prev.is_contiguous()
                assert w_prev.numel() == cfg.Kp and dv_prev.numel() == cfg.P and w_prev.is_contiguous() and dv_prev.is_contiguous()
                io.theta_prev, io.w_prev, io.dv_prev = key[4], key[5], key[6]
            io.idx, io.dist, io.theta = self.idx.data_ptr(), self.dist.data_ptr(), self.theta.data_ptr()
            io.w, io.dv, io.L = self.w.data_ptr(), self.dv.data_ptr(), self.L.data_ptr()
            io.next, io.parent, io.seeds = self.next.data_ptr(), self.parent.data_ptr(), self.seeds.data_ptr()
            self._io, self._io_key = io, key
            self._args = (self.ctx.handle, C.addressof(cfg), C.addressof(io), None, C.addressof(self.ncomp))
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)      # (a no-op unless the stream changed)
        a = self._args
        self.ctx.check(self._call(a[0], a[1], a[2], C.addressof(rng), a[4]))
        if cfg.Nnext:
            self.ctx.warn_generation_giveups()
        return self
