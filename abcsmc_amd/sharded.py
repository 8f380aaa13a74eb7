"""Row-sharded SMC generation: one process per GPU, particles split in contiguous row blocks,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests) for the
few real exchange steps of the path:

  1. broadcast of the pilot shift                       (16*ceil((M+P)/16) doubles)
  2. ONE packed all-reduce of the sufficient statistics  (counts, column sums, Gram blocks; <= 0.35 MB)
     -> the PLS deflation loop is then replicated on every rank (deterministic, identical model)
  3. exact distributed radix select: 6 all-reduces of a 2048-bin histogram (8 KB) find the global K-th key,
     then an all-gather of the K winners (dist, global row) padded to the largest shard + a replicated sort
  4. all-reduce of the K x P gathered posterior (each row is owned by exactly one rank)
  5. all-gather of the raw importance weights (KDE rows are sharded K/G per rank)
  6. resampling/perturbation need no exchange: every rank regenerates its own slice of the
     reference's sequential taus2 stream by jump-ahead and perturbs locally.

The numerical stages are behind a small backend interface: `HipBackend` (the product: C ABI ->
HIP kernels on device tensors).  tests/ provide a numpy backend so the orchestration, offsets and
collectives are covered at world_size 2 on CPU with gloo.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib
from ._lib import Rng, lib


class HipBackend:
    """Stage calls on device-resident torch tensors through include/abcsmc_hip.h (*_dev entry points)."""

    def __init__(self, device, ctx=None):
        self.device = torch.device(device)
        self.ctx = ctx if ctx is not None else _lib.default_context(self.device.index or 0)

    def _s(self):
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        return self.ctx.handle

    def empty(self, shape, dtype=torch.float64):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=torch.float64):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def stats_len(self, M, P):
        return lib().abc_stats_len(M, P)

    def model_len(self, M, P, A):
        return lib().abc_model_len(M, P, A)

    def stats_shift(self, X, Y, stats):
        M, n = X.shape
        P = Y.shape[0]
        self.ctx.check(lib().abc_stats_shift_dev(self._s(), X.data_ptr(), Y.data_ptr(), n, n, n, M, P, stats.data_ptr()))

    def stats_accumulate(self, X, Y, row0, n_train_global, stats):
        M, n = X.shape
        P = Y.shape[0]
        self.ctx.check(lib().abc_stats_accumulate_dev(self._s(), X.data_ptr(), Y.data_ptr(), n, n, n, M, P, row0,
                                                      n_train_global, stats.data_ptr()))

    def pls_model(self, stats, obs, M, P, A, rule, model):
        self.ctx.check(lib().abc_pls_model_dev(self._s(), stats.data_ptr(), obs.data_ptr(), M, P, A, rule, model.data_ptr()))

    def model_ncomp(self, model, M, P, A):
        nc = C.c_int32(0)
        self.ctx.check(lib().abc_model_ncomp(self._s(), model.data_ptr(), M, P, A, C.addressof(nc)))
        return nc.value

    def project_distance(self, X, P, A, model, out):
        M, n = X.shape
        self.ctx.check(lib().abc_project_distance_dev(self._s(), X.data_ptr(), n, n, M, P, A, model.data_ptr(), 0,
                                                      out.data_ptr()))

    def select_smallest(self, d, K, idx_base, idx_out, dist_out):
        self.ctx.check(lib().abc_select_smallest_dev(self._s(), d.data_ptr(), d.numel(), K, idx_base, idx_out.data_ptr(),
                                                     dist_out.data_ptr()))

    def sort_pairs(self, key, idx):
        self.ctx.check(lib().abc_sort_pairs_dev(self._s(), key.data_ptr(), idx.data_ptr(), key.numel()))

    def merge_runs(self, key, idx, n_runs, run_len, key_out, idx_out):
        self.ctx.check(lib().abc_merge_sorted_runs_dev(self._s(), key.data_ptr(), idx.data_ptr(), n_runs, run_len,
                                                       key_out.data_ptr(), idx_out.data_ptr()))

    # distributed radix select (histograms are all-reduced by the driver between hist and pick)
    def select_begin(self, K, state, hist):
        self.ctx.check(lib().abc_select_begin_dev(self._s(), K, state.data_ptr(), hist.data_ptr()))

    def select_hist(self, d, state, p, hist):
        self.ctx.check(lib().abc_select_hist_dev(self._s(), d.data_ptr(), d.numel(), state.data_ptr(), p, hist.data_ptr()))

    def select_pick(self, state, p, hist, K):
        self.ctx.check(lib().abc_select_pick_dev(self._s(), state.data_ptr(), p, hist.data_ptr(), K))

    def select_count(self, d, state, counts):
        self.ctx.check(lib().abc_select_count_dev(self._s(), d.data_ptr(), d.numel(), state.data_ptr(), counts.data_ptr()))

    def select_compact(self, d, state, n_less, ties_take, idx_base, idx_out, dist_out):
        self.ctx.check(lib().abc_select_compact_dev(self._s(), d.data_ptr(), d.numel(), state.data_ptr(), n_less, ties_take,
                                                    idx_base, idx_out.data_ptr(), dist_out.data_ptr()))

    def gather_rows(self, Y, idx, idx_base, theta):
        P, n = Y.shape
        K = idx.numel()
        self.ctx.check(lib().abc_gather_rows_dev(self._s(), Y.data_ptr(), n, n, P, idx.data_ptr(), K, idx_base,
                                                 theta.data_ptr(), K))

    def doubled_variance(self, theta, dv):
        P, K = theta.shape
        self.ctx.check(lib().abc_doubled_variance_dev(self._s(), theta.data_ptr(), K, P, dv.data_ptr()))

    def weights_raw(self, priors, theta, k0, kn, theta_prev, w_prev, dv_prev, out):
        P, K = theta.shape
        Kp = theta_prev.shape[1]
        self.ctx.check(lib().abc_weights_raw_dev(self._s(), priors.data_ptr(), theta.data_ptr(), K, P, k0, kn,
                                                 theta_prev.data_ptr(), Kp, w_prev.data_ptr(), dv_prev.data_ptr(),
                                                 out.data_ptr()))

    def normalize_l2(self, w):
        self.ctx.check(lib().abc_normalize_l2_dev(self._s(), w.data_ptr(), w.numel()))

    def setup_mvn(self, theta, L):
        P, K = theta.shape
        self.ctx.check(lib().abc_setup_mvn_sampler_dev(self._s(), theta.data_ptr(), K, P, L.data_ptr()))

    def resample(self, rng, w, i0, n, parent):
        self.ctx.check(lib().abc_resample_dev(self._s(), C.addressof(rng), w.data_ptr(), w.numel(), i0, n, parent.data_ptr()))

    def perturb(self, rng, theta, priors, parent, i0, n, multivariate, L_or_dv, out, seeds, seed_offset):
        P, K = theta.shape
        self.ctx.check(lib().abc_perturb_dev(self._s(), C.addressof(rng), theta.data_ptr(), K, P, priors.data_ptr(),
                                             parent.data_ptr(), i0, n, int(multivariate), L_or_dv.data_ptr(),
                                             out.data_ptr(), seeds.data_ptr() if seeds is not None else None, seed_offset))


class ShardedGeneration:
    """One generation turn-over over `world` ranks; every rank holds n_local rows of the set.

    The stage-level driver (torch.distributed collectives between C ABI stage calls): argmin-PRESS component rule only.  The
    Wilcoxon rule -- the drop-in default -- needs its bounds cascade's counts exchanged between kernels of ONE stage; that lives
    in the C++ driver (CabiShardedGeneration -> abc_generation_sharded_dev), which is also the fast one."""

    def __init__(self, backend, n_local, M, P, K, Kp, nnext_local, train_frac=0.5, max_comp=0,
                 rule=None, multivariate=True, group=None):
        self.be = backend
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_local, self.M, self.P, self.K, self.Kp = n_local, M, P, K, Kp
        self.N = n_local * self.world
        self.nnext_local = nnext_local
        self.Nnext = nnext_local * self.world
        self.A = max_comp if max_comp > 0 else min(M, P)
        # (no default: everything else without a rule argument applies the Wilcoxon rule -- a silent argmin PRESS here would give
        # other component counts and selections than device.Generation / CabiShardedGeneration on the same set)
        if rule != _lib.RULE_MIN_PRESS:
            raise ValueError("ShardedGeneration (stage-level driver) applies argmin PRESS only: pass rule=RULE_MIN_PRESS to say so, "
                             "or use CabiShardedGeneration for the Wilcoxon rule (the default everywhere else)")
        self.train_frac, self.rule, self.multivariate = train_frac, rule, multivariate
        be = backend
        self.k_local = min(K, n_local)
        self.stats = be.zeros(be.stats_len(M, P))
        self.model = be.empty(be.model_len(M, P, self.A))
        self.dist_local = be.empty(n_local)
        self.cand_idx = be.empty(self.k_local * self.world, torch.int64)
        self.cand_dist = be.empty(self.k_local * self.world)
        self.merged_idx = be.empty(self.k_local * self.world, torch.int64)
        self.merged_dist = be.empty(self.k_local * self.world)
        self.loc_idx = be.empty(self.k_local, torch.int64)
        self.loc_dist = be.empty(self.k_local)
        self.sel_state = be.zeros(8, torch.int64)
        self.sel_hist = be.zeros(2048, torch.int32)
        self.sel_counts = be.zeros(2, torch.int64)
        self.sel_all_counts = be.zeros(2 * self.world, torch.int64)
        self.idx = be.empty(K, torch.int64)
        self.dist = be.empty(K)
        self.theta = be.zeros((P, K))
        self.dv = be.empty(P)
        self.w = be.empty(K)
        self.L = be.empty((P, P))
        # KDE row ranges per rank (contiguous, as even as possible)
        base, rem = divmod(K, self.world)
        self.k_counts = [base + (1 if r < rem else 0) for r in range(self.world)]
        self.k_offsets = [sum(self.k_counts[:r]) for r in range(self.world)]
        self.kmax = max(self.k_counts)
        self.w_slices = be.zeros(self.kmax * self.world)
        self.w_mine = be.zeros(self.kmax)
        self.next = be.empty((P, max(nnext_local, 1)))
        self.parent = be.empty(max(nnext_local, 1), torch.int64)
        self.seeds = be.empty(max(nnext_local, 1), torch.int64)
        self.ncomp = 0
        self.shift_len = 16 * ((M + P + 15) // 16)

    def _ar(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def run(self, X, Y, obs, priors, rng, theta_prev=None, w_prev=None, dv_prev=None):
        be, W, r = self.be, self.world, self.rank
        M, P, K, A = self.M, self.P, self.K, self.A
        row0 = r * self.n_local
        ntrain = int(math.floor(self.N * self.train_frac + 0.5))   # AbcUtil.cpp:438 (std::round, half away from zero)
        # 1-2: sufficient statistics
        be.stats_shift(X, Y, self.stats)
        shift = self.stats[2:2 + self.shift_len]
        if W > 1:
            dist.broadcast(shift, src=0, group=self.group)
        be.stats_accumulate(X, Y, row0, ntrain, self.stats)
        if W > 1:
            if r != 0:
                shift.zero_()
            self._ar(self.stats)
        # replicated model fit
        be.pls_model(self.stats, obs, M, P, A, self.rule, self.model)
        # 3: distances, local winners, global merge
        be.project_distance(X, P, A, self.model, self.dist_local)
        if W > 1:
            # exact distributed selection: 6 all-reduced radix histograms -> global K-th key on every rank
            be.select_begin(K, self.sel_state, self.sel_hist)
            for p in range(6):
                be.select_hist(self.dist_local, self.sel_state, p, self.sel_hist)
                self._ar(self.sel_hist)
                be.select_pick(self.sel_state, p, self.sel_hist, K)
            be.select_count(self.dist_local, self.sel_state, self.sel_counts)
            dist.all_gather_into_tensor(self.sel_all_counts, self.sel_counts, group=self.group)
            ac = self.sel_all_counts.cpu().tolist()
            less = [ac[2 * q] for q in range(W)]
            eq = [ac[2 * q + 1] for q in range(W)]
            remaining = K - sum(less)                 # ties at the threshold go to the lowest global rows first
            take = []
            for q in range(W):
                tq = min(eq[q], max(remaining, 0))
                take.append(tq)
                remaining -= tq
            nw = [less[q] + take[q] for q in range(W)]
            maxw = max(nw)
            be.select_compact(self.dist_local, self.sel_state, less[r], take[r], row0, self.loc_idx, self.loc_dist)
            if nw[r] > 1:                             # every rank sorts its own winners (stable: ties stay in row order)
                be.sort_pairs(self.loc_dist[:nw[r]], self.loc_idx[:nw[r]])
            if nw[r] < maxw:                          # pad to the common length with sentinels that sort last
                self.loc_idx[nw[r]:maxw].fill_(1 << 62)
                self.loc_dist[nw[r]:maxw].fill_(float("inf"))
            cidx, cdist = self.cand_idx[:W * maxw], self.cand_dist[:W * maxw]
            dist.all_gather_into_tensor(cidx, self.loc_idx[:maxw], group=self.group)
            dist.all_gather_into_tensor(cdist, self.loc_dist[:maxw], group=self.group)
            # W sorted runs -> one sequence; equal distances: lower rank (= lower global rows) first
            midx, mdist = self.merged_idx[:W * maxw], self.merged_dist[:W * maxw]
            be.merge_runs(cdist, cidx, W, maxw, mdist, midx)
            self.idx.copy_(midx[:K])
            self.dist.copy_(mdist[:K])
        else:
            be.select_smallest(self.dist_local, self.k_local, row0, self.loc_idx, self.loc_dist)
            self.idx.copy_(self.loc_idx[:K])
            self.dist.copy_(self.loc_dist[:K])
        # 4: posterior rows
        self.theta.zero_()
        be.gather_rows(Y, self.idx, row0, self.theta)
        self._ar(self.theta)
        be.doubled_variance(self.theta, self.dv)
        # 5: weights
        if theta_prev is None or self.Kp == 0:
            self.w.fill_(1.0 / K)                                                      # AbcUtil.cpp:543-544
        else:
            k0, kn = self.k_offsets[r], self.k_counts[r]
            be.weights_raw(priors, self.theta, k0, kn, theta_prev, w_prev, dv_prev, self.w_mine)
            if W > 1:
                dist.all_gather_into_tensor(self.w_slices, self.w_mine, group=self.group)
                for q in range(W):
                    self.w[self.k_offsets[q]:self.k_offsets[q] + self.k_counts[q]].copy_(
                        self.w_slices[q * self.kmax:q * self.kmax + self.k_counts[q]])
            else:
                self.w.copy_(self.w_mine[:K])
            be.normalize_l2(self.w)                                                    # AbcUtil.cpp:583
        # 6: proposals for this rank's slice of the next set
        if self.nnext_local:
            i0 = r * self.nnext_local
            if self.multivariate:
                be.setup_mvn(self.theta, self.L)
            be.resample(rng, self.w, i0, self.nnext_local, self.parent)
            be.perturb(rng, self.theta, priors, self.parent, i0, self.nnext_local, self.multivariate,
                       self.L if self.multivariate else self.dv, self.next, self.seeds, self.Nnext)
            lib().abc_rng_jump(C.addressof(rng), 2 * self.Nnext)   # Nnext resampling draws + Nnext seeds (host-only call)
        self.ncomp = be.model_ncomp(self.model, M, P, A)
        return self


# ---------------------------------------------------------------------------------------------------------------------
# The same protocol inside the C ABI: abc_generation_sharded_dev (abcsmc_amd/csrc/sharded.hip) with an RCCL
# communicator, or with the collectives of torch.distributed handed in as callbacks (any backend: the tests run two
# ranks over gloo on one GPU).  This is the product path of bench.py --gpus N; the Python driver above is the test
# harness that documents the protocol stage by stage.
# ---------------------------------------------------------------------------------------------------------------------
class _DevView:
    """zero-copy torch view of a raw device buffer (CUDA array interface)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": (int(nbytes),), "typestr": "|u1", "version": 2}


def _view(ptr, nbytes, dtype, device):
    return torch.as_tensor(_DevView(ptr, nbytes), device=device).view(dtype)


def attach_torch_distributed(ctx, device, group=None):
    """abc_comm_init_callbacks over torch.distributed: every collective of the C++ driver is forwarded to the process
    group (gloo or nccl).  The context must run on torch's current stream (Context.set_stream)."""
    dev = torch.device(device)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dts = {_lib.DT_F64: (torch.float64, 8), _lib.DT_I32: (torch.int32, 4), _lib.DT_I64: (torch.int64, 8)}

    ctx.comm_calls = {"all_reduce": 0, "all_gather": 0, "broadcast": 0}      # collectives the driver has issued (tests, bench)

    def all_reduce_sum(buf, count, dtype, stream):
        ctx.comm_calls["all_reduce"] += 1
        t, sz = dts[dtype]
        dist.all_reduce(_view(buf, count * sz, t, dev), op=dist.ReduceOp.SUM, group=group)
        return 0

    def all_gather(send, recv, nbytes, stream):
        ctx.comm_calls["all_gather"] += 1
        dist.all_gather_into_tensor(_view(recv, nbytes * world, torch.uint8, dev), _view(send, nbytes, torch.uint8, dev).clone(),
                                    group=group)
        return 0

    def broadcast(buf, nbytes, root, stream):
        ctx.comm_calls["broadcast"] += 1
        dist.broadcast(_view(buf, nbytes, torch.uint8, dev), src=dist.get_global_rank(group, root) if group is not None else root,
                       group=group)
        return 0

    ctx.comm_init_callbacks(world, rank, all_reduce_sum, all_gather, broadcast)


def attach_rccl(ctx, device, group=None):
    """abc_comm_init_rank: the 128-byte RCCL id is created on rank 0 and broadcast through torch.distributed"""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ident = [_lib.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=0, group=group)
    ctx.comm_init_rccl(world, rank, ident[0])


class CabiShardedGeneration:
    """abc_generation_sharded_dev with pre-allocated torch buffers; same attributes as ShardedGeneration.
    The context's communicator decides the world (Context.comm_info); without one it is a single-GPU generation."""

    def __init__(self, ctx, device, n_local, M, P, K, Kp, nnext_local, train_frac=0.5, max_comp=0,
                 rule=_lib.RULE_DEFAULT, multivariate=True, row0=None, N_total=None, next0=None, Nnext_total=None):
        self.ctx, self.device = ctx, torch.device(device)
        _, world, rank = ctx.comm_info()
        self.world, self.rank = world, rank
        N_total = n_local * world if N_total is None else N_total
        Nnext_total = nnext_local * world if Nnext_total is None else Nnext_total
        row0 = rank * n_local if row0 is None else row0
        next0 = rank * nnext_local if next0 is None else next0
        self.cfg = _lib.ShardedCfg(n_local, row0, N_total, M, P, K, Kp, nnext_local, next0, Nnext_total, float(train_frac),
                                   int(max_comp), int(rule), int(bool(multivariate)), 0)
        f64, i64, d = torch.float64, torch.int64, self.device
        self.idx = torch.empty(K, dtype=i64, device=d)
        self.dist = torch.empty(K, dtype=f64, device=d)
        self.theta = torch.empty((P, K), dtype=f64, device=d)
        self.w = torch.empty(K, dtype=f64, device=d)
        self.dv = torch.empty(P, dtype=f64, device=d)
        self.L = torch.empty((P, P), dtype=f64, device=d)
        self.next = torch.empty((P, max(nnext_local, 1)), dtype=f64, device=d)
        self.parent = torch.empty(max(nnext_local, 1), dtype=i64, device=d)
        self.seeds = torch.empty(max(nnext_local, 1), dtype=i64, device=d)
        self.ncomp = 0
        self._nc = C.c_int32(0)
        self._io_key, self._io, self._args = None, None, None
        self._call = lib().abc_generation_sharded_dev

    def run(self, X, Y, obs, priors, rng, theta_prev=None, w_prev=None, dv_prev=None):
        cfg = self.cfg
        weighted = theta_prev is not None and cfg.Kp
        # (as device.Generation.run: the argument block of the C call is rebuilt, and the shapes checked, only when a buffer moved
        # or changed shape -- at a millisecond per generation on eight GPUs a rebuilt ctypes structure per call is several percent)
        key = (X.data_ptr(), Y.data_ptr(), obs.data_ptr(), priors.data_ptr(),
               theta_prev.data_ptr() if weighted else 0, w_prev.data_ptr() if weighted else 0,
               dv_prev.data_ptr() if weighted else 0, X.shape, Y.shape, X.stride(), Y.stride(), obs.shape, priors.shape,
               theta_prev.shape if weighted else None, theta_prev.stride() if weighted else None,
               w_prev.shape if weighted else None, dv_prev.shape if weighted else None)
        if key != self._io_key:
            assert X.shape == (cfg.M, cfg.n_local) and Y.shape == (cfg.P, cfg.n_local) and X.is_contiguous() and Y.is_contiguous()
            assert obs.numel() == cfg.M and obs.is_contiguous() and priors.numel() >= 24 * cfg.P
            io = _lib.GenerationIO()
            io.X, io.Y, io.obs, io.priors = key[0], key[1], key[2], key[3]
            if weighted:
                assert theta_prev.shape == (cfg.P, cfg.Kp) and theta_prev.is_contiguous()
                assert w_prev.numel() == cfg.Kp and dv_prev.numel() == cfg.P and w_prev.is_contiguous() and dv_prev.is_contiguous()
                io.theta_prev, io.w_prev, io.dv_prev = key[4], key[5], key[6]
            io.idx, io.dist, io.theta = self.idx.data_ptr(), self.dist.data_ptr(), self.theta.data_ptr()
            io.w, io.dv, io.L = self.w.data_ptr(), self.dv.data_ptr(), self.L.data_ptr()
            io.next, io.parent, io.seeds = self.next.data_ptr(), self.parent.data_ptr(), self.seeds.data_ptr()
            self._io, self._io_key = io, key
            self._args = (self.ctx.handle, C.addressof(cfg), C.addressof(io), None, C.addressof(self._nc))
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        a = self._args
        self.ctx.check(self._call(a[0], a[1], a[2], C.addressof(rng), a[4]))
        self.ncomp = self._nc.value
        return self
