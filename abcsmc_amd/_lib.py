"""ctypes binding of libabcsmc_hip.so -- exactly the symbols include/abcsmc_hip.h declares."""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# ABCSMC_HIP_SO: developer override for A/B runs of diagnostic builds (scripts/); default = the in-tree build
SO_PATH = os.environ.get("ABCSMC_HIP_SO") or os.path.join(_HERE, "libabcsmc_hip.so")

PRIOR_GAUSS, PRIOR_UNIF_INT, PRIOR_UNIF_REAL = 0, 1, 2
RULE_MIN_PRESS, RULE_WILCOXON = 0, 1
# the drop-in's default everywhere (C++ facade ABC::component_rule, the shell's "pls_component_rule", these mirrors, bench.py): SURVEY A.2's
# best knowledge of upstream's optimal_num_components -- argmin PRESS reduced by the Wilcoxon signed-rank test
RULE_DEFAULT = RULE_WILCOXON
KDE_AUTO, KDE_FP64 = 0, 1
GRAM_AUTO, GRAM_FP64, GRAM_I8 = 0, 1, 2
KDE_RAN_NONE, KDE_RAN_FP64, KDE_RAN_SPLIT = 0, 1, 2
DT_F64, DT_I32, DT_I64 = 0, 1, 2
COMM_ID_BYTES = 128
COMM_NONE, COMM_RCCL, COMM_CALLBACKS = 0, 1, 2
NOISE_DEVICE, NOISE_REFERENCE_STREAM = 0, 1
WEIGHT_GAUSSIAN, WEIGHT_EPANECHNIKOV = 0, 1
ALIAS_DEVICE, ALIAS_HOST = 0, 1


class LibraryMissing(ImportError):
    pass


class AbcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("abcsmc_hip error %d: %s" % (code, msg))
        self.code = code


class AbcWarning(UserWarning):
    pass




class Prior(C.Structure):
    _fields_ = [("kind", C.c_int32), ("pad_", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


class Rng(C.Structure):
    _fields_ = [("s1", C.c_uint32), ("s2", C.c_uint32), ("s3", C.c_uint32)]


class GenerationCfg(C.Structure):
    _fields_ = [("N", C.c_size_t), ("M", C.c_size_t), ("P", C.c_size_t),
                ("K", C.c_size_t), ("Kp", C.c_size_t), ("Nnext", C.c_size_t),
                ("train_frac", C.c_double),
                ("max_comp", C.c_int32), ("rule", C.c_int32), ("multivariate", C.c_int32),
                ("reserved", C.c_int32)]


class GenerationIO(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "X", "Y", "obs", "priors", "theta_prev", "w_prev", "dv_prev", "idx", "dist", "theta", "w",
        "dv", "L", "next", "parent", "seeds")]


class ShardedCfg(C.Structure):
    _fields_ = [("n_local", C.c_size_t), ("row0", C.c_size_t), ("N_total", C.c_size_t),
                ("M", C.c_size_t), ("P", C.c_size_t), ("K", C.c_size_t), ("Kp", C.c_size_t),
                ("nnext_local", C.c_size_t), ("next0", C.c_size_t), ("Nnext_total", C.c_size_t),
                ("train_frac", C.c_double),
                ("max_comp", C.c_int32), ("rule", C.c_int32), ("multivariate", C.c_int32), ("reserved", C.c_int32)]


ALL_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
BROADCAST_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class CommCallbacks(C.Structure):
    _fields_ = [("all_reduce_sum", ALL_REDUCE_FN), ("all_gather", ALL_GATHER_FN), ("broadcast", BROADCAST_FN),
                ("user", C.c_void_p)]


def make_priors(spec):
    arr = (Prior * len(spec))()
    for i, (k, a, b) in enumerate(spec):
        arr[i].kind, arr[i].a, arr[i].b = int(k), float(a), float(b)
    return arr


# symbol -> (restype, argtypes); MUST list every function declared in include/abcsmc_hip.h
_vp, _sz, _u64, _i, _d = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int, C.c_double
SIGNATURES = {
    "abc_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "abc_ctx_destroy": (None, [_vp]),
    "abc_last_error": (C.c_char_p, [_vp]),
    "abc_ctx_set_stream": (_i, [_vp, _vp]),
    "abc_ctx_use_own_stream": (_i, [_vp]),
    "abc_ctx_set_kde_mode": (_i, [_vp, _i]),
    "abc_ctx_set_gram_mode": (_i, [_vp, _i]),
    "abc_kde_last_kernel": (_i, [_vp, _vp]),
    "abc_ctx_set_noise_mode": (_i, [_vp, _i]),
    "abc_ctx_set_weight_kernel": (_i, [_vp, _i]),
    "abc_perturb_giveups": (_i, [_vp, _vp, _i]),
    "abc_generation_giveups": (_i, [_vp, _vp]),
    "abc_generation_repeats": (_i, [_vp, _vp, _vp, _i]),
    "abc_ctx_set_alias_mode": (_i, [_vp, _i]),
    "abc_alias_stats": (_i, [_vp, _vp, _vp, _i]),
    "abc_alias_table": (_i, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "abc_ctx_synchronize": (_i, [_vp]),
    "abc_version": (_i, []),
    "abc_timing_enable": (_i, [_vp, _i]),
    "abc_timing_read": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i]),
    "abc_timing_overhead": (_i, [_vp, _i, _vp]),
    "abc_rng_set": (None, [_vp, C.c_ulong]),
    "abc_rng_get": (C.c_uint32, [_vp]),
    "abc_rng_jump": (None, [_vp, _u64]),
    "abc_particle_ranking_pls": (_i, [_vp, _vp, _vp, _vp, _sz, _sz, _sz, _d, _i, _i, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "abc_particle_ranking_simple": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _vp, _vp]),
    "abc_calculate_doubled_variance": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "abc_weight_predictive_prior_uniform": (_i, [_vp, _sz, _vp]),
    "abc_weight_predictive_prior": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _sz, _vp, _vp, _vp]),
    "abc_setup_mvn_sampler": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "abc_sample_posterior": (_i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "abc_sample_mvn_predictive_priors": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "abc_sample_predictive_priors": (_i, [_vp, _vp, _sz, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp]),
    "abc_generation_dev": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "abc_stats_len": (_sz, [_sz, _sz]),
    "abc_stats_shift_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp]),
    "abc_stats_accumulate_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _u64, _u64, _vp]),
    "abc_model_len": (_sz, [_sz, _sz, _sz]),
    "abc_pls_model_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _i, _vp]),
    "abc_pls_wilcoxon_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz, _sz, _vp]),
    "abc_model_ncomp": (_i, [_vp, _vp, _sz, _sz, _sz, _vp]),
    "abc_simple_model_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "abc_project_distance_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _vp, _i, _vp]),
    "abc_select_smallest_dev": (_i, [_vp, _vp, _sz, _sz, _u64, _vp, _vp]),
    "abc_select_begin_dev": (_i, [_vp, _u64, _vp, _vp]),
    "abc_select_hist_dev": (_i, [_vp, _vp, _sz, _vp, _i, _vp]),
    "abc_select_pick_dev": (_i, [_vp, _vp, _i, _vp, _u64]),
    "abc_select_count_dev": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "abc_select_compact_dev": (_i, [_vp, _vp, _sz, _vp, _u64, _u64, _u64, _vp, _vp]),
    "abc_sort_pairs_dev": (_i, [_vp, _vp, _vp, _sz]),
    "abc_merge_sorted_runs_dev": (_i, [_vp, _vp, _vp, _i, _sz, _vp, _vp]),
    "abc_gather_rows_dev": (_i, [_vp, _vp, _sz, _sz, _sz, _vp, _sz, _u64, _vp, _sz]),
    "abc_doubled_variance_dev": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "abc_weights_raw_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _sz, _sz, _vp, _sz, _vp, _vp, _vp]),
    "abc_normalize_l2_dev": (_i, [_vp, _vp, _sz]),
    "abc_setup_mvn_sampler_dev": (_i, [_vp, _vp, _sz, _sz, _vp]),
    "abc_resample_dev": (_i, [_vp, _vp, _vp, _sz, _u64, _sz, _vp]),
    "abc_perturb_dev": (_i, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _u64, _sz, _i, _vp, _vp, _vp, _u64]),
    "abc_comm_unique_id": (_i, [_vp]),
    "abc_comm_init_rank": (_i, [_vp, _i, _i, _vp]),
    "abc_comm_init_callbacks": (_i, [_vp, _i, _i, _vp]),
    "abc_comm_destroy": (_i, [_vp]),
    "abc_comm_info": (_i, [_vp, _vp, _vp, _vp]),
    "abc_ctx_create_multi": (_i, [_vp, _i, _vp]),
    "abc_generation_sharded_dev": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "abc_generation_multi": (_i, [_vp, _i, _vp, _vp, _vp, _vp]),
}

_LIB = None


def lib():
    """Load the HIP library; fail loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise LibraryMissing(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C abcsmc_amd/csrc` (there is no CPU fallback)" % SO_PATH)
        # One HIP runtime per process: PyTorch ships its own libamdhip64.so (same SONAME as /opt/rocm's).  Loaded after
        # PyTorch's, this library binds to that copy; loaded BEFORE it, it pulls in the system copy and PyTorch later adds its
        # own -- two runtimes, and the second to touch the device fails (abc_ctx_create then reports no usable GPU).  The
        # device / multi-GPU drivers of this package use torch for memory and streams anyway, so it goes first when present.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


class Context:
    """One abc_ctx per GPU (include/abcsmc_hip.h: abc_ctx_create)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().abc_ctx_create(int(device), C.byref(self._h))
        if rc:
            raise AbcError(rc, "abc_ctx_create(device=%d) failed: no usable GPU (HIP path is mandatory)" % device)
        self.device = device

    @classmethod
    def from_handle(cls, handle, device):
        """a view of a context somebody else owns (MultiContext.context): not destroyed with this object"""
        self = cls.__new__(cls)
        self._h = C.c_void_p(handle)
        self.device = int(device)
        self._borrowed = True
        return self

    @property
    def handle(self):
        return self._h

    def check(self, rc):
        """non-zero status: AbcError (the ABI has no positive status)"""
        if rc:
            raise AbcError(rc, lib().abc_last_error(self._h).decode())

    def generation_repeats(self, reset=False):
        """(ranking repeats, generation repeats) since the context was created / the last reset: what the speculation on the
        component count has cost (abc_generation_repeats)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.check(lib().abc_generation_repeats(self._h, C.byref(a), C.byref(b), int(reset)))
        return a.value, b.value

    def generation_giveups(self):
        """proposals the most recent generation call gave up on (abc_generation_giveups); no synchronisation"""
        n = C.c_uint64(0)
        lib().abc_generation_giveups(self._h, C.byref(n))
        return n.value

    def warn_generation_giveups(self):
        """after abc_generation_dev: a Python warning when the perturbation gave up on proposals during that call (they are
        their valid parents / prior means; the reference would still be retrying, AbcUtil.cpp:132).  No synchronisation."""
        n = C.c_uint64(0)
        lib().abc_generation_giveups(self._h, C.byref(n))
        if n.value:
            import warnings
            self.last_warning = ("generation complete, but the perturbation gave up on %d proposal(s): they are their parents "
                                 "(MULTIVARIATE, after 16384 rejected attempts) or prior means (INDEPENDENT, after 1000); "
                                 "abc_perturb_giveups has the total" % n.value)
            warnings.warn(AbcWarning(self.last_warning), stacklevel=3)
        return n.value

    def set_stream(self, stream_ptr):
        if getattr(self, "_stream", -1) != stream_ptr:      # abc_ctx_set_stream synchronises: only on change
            self.check(lib().abc_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))
            self._stream = stream_ptr

    def timing_enable(self, on=True):
        """False / 0: off; True / 1: every stage; 2: only the k_gram / k_kde kernel brackets"""
        self.check(lib().abc_timing_enable(self._h, int(on)))

    def timing_read(self, reset=True):
        """-> {stage: (device_ms, host_ms, launches)} accumulated since the last reset"""
        n = 32
        names = (C.c_char_p * n)()
        ms, hms, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_longlong * n)()
        k = lib().abc_timing_read(self._h, names, ms, hms, cnt, n, int(reset))
        if k < 0:
            self.check(k)
        return {names[i].decode(): (ms[i], hms[i], cnt[i]) for i in range(k)}

    def timing_overhead(self, reps=50):
        """ms an event pair around one kernel reports beyond the kernel's execution (abc_timing_overhead)"""
        ov = C.c_double(0.0)
        self.check(lib().abc_timing_overhead(self._h, int(reps), C.byref(ov)))
        return ov.value

    def synchronize(self):
        self.check(lib().abc_ctx_synchronize(self._h))

    def kde_last_kernel(self):
        """KDE_RAN_FP64 / KDE_RAN_SPLIT: which kernel summed the pairs of the last weight call (abc_kde_last_kernel)"""
        w = C.c_int(0)
        self.check(lib().abc_kde_last_kernel(self._h, C.byref(w)))
        return w.value

    def set_weight_kernel(self, kernel):
        """WEIGHT_GAUSSIAN (the reference's, default) or WEIGHT_EPANECHNIKOV (extension): abc_ctx_set_weight_kernel"""
        self.check(lib().abc_ctx_set_weight_kernel(self._h, int(kernel)))

    def set_noise_mode(self, mode):
        """NOISE_DEVICE (Philox stream on the device, default) or NOISE_REFERENCE_STREAM (the reference's sequential taus2
        consumption, host loop): abc_ctx_set_noise_mode"""
        self.check(lib().abc_ctx_set_noise_mode(self._h, int(mode)))

    def perturb_giveups(self, reset=False):
        """proposals the perturbation gave up on (abc_perturb_giveups)"""
        n = C.c_uint64(0)
        self.check(lib().abc_perturb_giveups(self._h, C.byref(n), int(reset)))
        return n.value

    def set_alias_mode(self, mode):
        """ALIAS_DEVICE (default: the resampling table built on the GPU by verified prefix scans) or ALIAS_HOST: abc_ctx_set_alias_mode"""
        self.check(lib().abc_ctx_set_alias_mode(self._h, int(mode)))

    def alias_stats(self, reset=False):
        """(device builds, host fallbacks) of the resampling table (abc_alias_stats)"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.check(lib().abc_alias_stats(self._h, C.byref(a), C.byref(b), int(reset)))
        return a.value, b.value

    def alias_table(self, w):
        """(F, A, on_device): the Walker alias table of the weights w as the context builds it (abc_alias_table)"""
        import numpy as np
        w = np.ascontiguousarray(w, dtype=np.float64)
        F, A, on = np.empty(w.size), np.empty(w.size, dtype=np.uint64), C.c_int(0)
        self.check(lib().abc_alias_table(self._h, w.ctypes.data, w.size, F.ctypes.data, A.ctypes.data, C.byref(on)))
        return F, A, on.value

    def set_gram_mode(self, mode):
        """GRAM_AUTO (byte-limb statistics kernel for wide sets whose partitions hold >= 400 000 rows), GRAM_FP64, or GRAM_I8 (the
        byte-limb kernel from 200 000 rows: A/B runs, its own tests) -- abc_ctx_set_gram_mode"""
        self.check(lib().abc_ctx_set_gram_mode(self._h, int(mode)))

    def set_kde_mode(self, mode):
        """KDE_AUTO (split-operand matrix-pipe kernel where it applies) or KDE_FP64 (abc_ctx_set_kde_mode)"""
        self.check(lib().abc_ctx_set_kde_mode(self._h, int(mode)))

    # ---- communicator of the row-sharded generation (include/abcsmc_hip.h, "Multi-GPU") ----------------------
    def comm_init_rccl(self, world, rank, unique_id):
        """RCCL communicator from the 128-byte id rank 0 obtained with comm_unique_id() and handed to every rank"""
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        self.check(lib().abc_comm_init_rank(self._h, int(world), int(rank), buf))

    def comm_init_callbacks(self, world, rank, all_reduce_sum, all_gather, broadcast):
        """collectives supplied by the caller: python callables (buf_ptr, count, dtype, stream) etc. -> 0 on success"""
        # A Python exception inside a ctypes callback is printed and turned into a return value of 0 -- which the C++ driver
        # reads as success and carries on with un-reduced statistics.  Every callback body is therefore guarded: an exception
        # (transport timeout, shape error) is logged, remembered (Context.comm_callback_error) and reported as a non-zero
        # status, so the collective -- and the generation -- fail with ABC_ERR_COMM.
        self.comm_callback_error = None

        def guarded(fn, what):
            def run(*a):
                try:
                    rc = fn(*a)
                    return int(rc) if rc is not None else 0
                except BaseException as e:       # noqa: BLE001 -- nothing may propagate into the C++ caller
                    import traceback
                    self.comm_callback_error = e
                    sys.stderr.write("abcsmc_amd: %s callback raised %s: %s\n" % (what, type(e).__name__, e))
                    traceback.print_exc()
                    return 1
            return run

        ar, ag, bc = guarded(all_reduce_sum, "all_reduce_sum"), guarded(all_gather, "all_gather"), guarded(broadcast, "broadcast")
        self._cb = CommCallbacks(ALL_REDUCE_FN(lambda u, b, n, dt, st: ar(b, n, dt, st)),
                                 ALL_GATHER_FN(lambda u, s, r, nb, st: ag(s, r, nb, st)),
                                 BROADCAST_FN(lambda u, b, nb, root, st: bc(b, nb, root, st)), None)
        self.check(lib().abc_comm_init_callbacks(self._h, int(world), int(rank), C.byref(self._cb)))

    def comm_info(self):
        k, w, r = C.c_int(0), C.c_int(1), C.c_int(0)
        self.check(lib().abc_comm_info(self._h, C.byref(k), C.byref(w), C.byref(r)))
        return k.value, w.value, r.value

    def comm_destroy(self):
        self.check(lib().abc_comm_destroy(self._h))

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            lib().abc_ctx_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    """128-byte RCCL id (abc_comm_unique_id); created on ONE rank and sent to the others"""
    buf = (C.c_char * COMM_ID_BYTES)()
    rc = lib().abc_comm_unique_id(buf)
    if rc:
        raise AbcError(rc, "abc_comm_unique_id failed (librccl.so.1 not loadable?)")
    return bytes(buf)


class MultiContext:
    """ndev contexts of ONE process joined by RCCL communicators (abc_ctx_create_multi)"""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        n = len(self.devices)
        self._arr = (C.c_void_p * n)()
        rc = lib().abc_ctx_create_multi((C.c_int * n)(*self.devices), n, self._arr)
        if rc:
            raise AbcError(rc, "abc_ctx_create_multi(%r) failed" % (self.devices,))

    def context(self, i):
        """the context of device i as a Context (a view: the MultiContext keeps the ownership); each is driven from its own
        host thread (abc_generation_sharded_dev is collective over the ndev contexts)"""
        return Context.from_handle(self._arr[i], self.devices[i])

    def generation(self, cfg, io, rng):
        """host-pointer generation over all devices (abc_generation_multi); -> ncomp"""
        nc = C.c_int32(0)
        rc = lib().abc_generation_multi(self._arr, len(self.devices), C.byref(cfg), C.byref(io), C.byref(rng), C.byref(nc))
        if rc:
            raise AbcError(rc, lib().abc_last_error(self._arr[0]).decode())
        return nc.value

    def close(self):
        for i in range(len(self.devices)):
            if self._arr[i]:
                lib().abc_ctx_destroy(self._arr[i])
                self._arr[i] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX = {}


def default_context(device=0):
    if device not in _DEFAULT_CTX:
        _DEFAULT_CTX[device] = Context(device)
    return _DEFAULT_CTX[device]
