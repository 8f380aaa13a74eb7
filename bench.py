#!/usr/bin/env python3
"""bench.py -- particles/s of one whole SMC generation turn-over (PLS rank + weights + resample/perturb)
on synthetic particle x (parameter | metric) matrices, with the inputs resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one generation: abc_generation_dev on one GPU, or the row-sharded driver (RCCL collectives) on N GPUs.
Default partitioning is STRONG scaling of the stated configuration: the set has the BASELINE.json size whatever N is
(configs[2]: 1e6 particles, K = K' = 1e5), its rows and the rows of the next set are split evenly over the ranks.
--scaling weak keeps the particles per GPU fixed instead (K = K' = 0.1 x all particles then grow with N, and the pair sums
of the weight stage, K K' / N per GPU, with them).
Prints ONE JSON line on rank 0 carrying the driver contract plus
  roofline            the kernel that dominates the step (the pair sums of the importance weights, k_kde_split):
                      achieved = ALGORITHMIC flops, K K' (3 P + 1) per launch, / HIP-event time measured live,
                      against the dense f16 MFMA peak (the pipe the dot products run on); the matrix work actually ISSUED
                      (limb products of the fp64 operands) is reported beside it as mfma_issue_frac
  roofline_hbm        the dominant HBM kernel (k_gram): algorithmic bytes / HIP-event time
  roofline_streaming  SURVEY 8(d): algorithmic bytes of a generation / (step time - pair-sum kernel), and the same for set 0
  sustained           >= 2 s of back-to-back steps after the timed region (the dominant kernel power-limits the chip)
  extra               legs outside the timed region: fp64 pair-sum kernel, INDEPENDENT noise (the reference's default),
                      particle_ranking_simple, the host-pointer drop-in call, the host alias build on log-normal weights
  cpu_baseline        the single-threaded CPU oracle on a bounded sample of the same workload (rank 0, N = 1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# A context drives a generation on TWO streams (ranking / weights beside the rng streams, the previous set's tiles, the posterior's
# moments).  The runtime maps a process's streams onto four hardware queues by default; in a process that holds more streams than
# that (several contexts; a rank under torch.distributed has PyTorch's and RCCL's as well) a generation was measured 0.1 .. 0.2 ms
# slower on one of the contexts (scripts/sharded_w1_time.py), and with eight queues it is not; the one-context
# timed region of an N = 1 run is the same with four and with eight.  (Set before anything initialises the GPU; a value already in
# the environment wins.)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

# BASELINE.json configs: TOTAL sizes as stated (K = K' = 0.1 N, N_next = N, train fraction 0.5, MULTIVARIATE); `gpus` = the
# GPU count the configuration is stated for (weak scaling: particles per GPU = N / gpus)
CONFIGS = {
    2: dict(name="configs[1]: synthetic 100k particles x 16 params x 32 metrics, PLS 8 components, full generation",
            N=100_000, M=32, P=16, A=8, gpus=1),
    3: dict(name="configs[2]: synthetic 1M particles x 16 params x 32 metrics, PLS 8 components, full generation",
            N=1_000_000, M=32, P=16, A=8, gpus=1),
    4: dict(name="configs[3]: synthetic 10M particles x 32 params x 64 metrics, PLS 8 components (stated for 8 GPUs)",
            N=10_000_000, M=64, P=32, A=8, gpus=8),
    5: dict(name="configs[4]: synthetic 1M particles x 16 params x 128 metrics, PLS 32 components (stated for 8 GPUs)",
            N=1_000_000, M=128, P=16, A=32, gpus=8),
}
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_PEAK_TF = 2500.0     # same guide: dense f16 / bf16 MFMA peak (v_mfma_f32_32x32x16_f16: 32 cycles per SIMD at 2.4 GHz)
MFMA_SUSTAINED_TF = 1247.0    # same guide, 'DVFS give-back' (1): a bare bf16 MFMA loop on RANDOM operands (the chip holds 1.90-1.95 GHz)
FP64_VALU_PEAK_TF = 78.6      # same guide: fp64 vector peak
ASSUMED_XGMI_COLLECTIVE_MS = 0.02   # scaling_model: a small RCCL collective over xGMI, ASSUMED (not measurable on one GPU)
ASSUMED_EXCHANGE_KERNELS_MS = 0.03  # scaling_model: the kernels around the exchanges that a one-GPU run does not launch (list header, unpack, merge, check, placement, slice unpadding: six launches of ~5 us, the floor of a dependent launch in every timeline under profiles/) -- not measurable at world 1
ASSUMED_WILCOXON_REPLICATED_MS = 0.045  # scaling_model: per level of the Wilcoxon cascade, what does not shrink with the rows: k_wx_totals 7-10 us + k_wx_bounds 17-24 us + the host's look and the next launch 11-13 us (profiles/r06_timeline_config3_weighted.txt, r06_timeline_config5_weighted.txt)


def pmc_traffic(kernel_prefix, config, world):
    """HBM bytes per launch of a kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE with the gfx950 correction,
    written by scripts/summarize_profiles.py); only for the exact single-GPU configuration profiled, else None"""
    if world != 1:
        return None
    for name in ("r06_pmc_hbm_traffic.json", "history/r05_pmc_hbm_traffic.json", "history/r04_pmc_hbm_traffic.json", "history/r03_pmc_hbm_traffic.json"):
        prof = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(prof):
            continue
        try:
            ent = json.load(open(prof))["configs"].get(str(config), {})
            for k, v in ent.items():
                if k.startswith(kernel_prefix):
                    return v["hbm_bytes_per_launch"]
        except Exception:
            pass
    return None


def split(total, world, rank):
    """contiguous shares, as even as possible (low ranks first): (offset, count)"""
    base, rem = divmod(total, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json config: 3 = configs[2] (1M particles, the full weight+resample generation the metric "
                         "is quoted on; default), 2 = configs[1] (100k), 4 / 5 = configs[3] / configs[4] at their stated totals")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong (default): the stated total whatever --gpus is; weak: particles per GPU fixed at the "
                         "configuration's per-GPU share (its total / the GPU count it is stated for)")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N > 1 inside THIS process: abc_ctx_create_multi (ncclCommInitAll) and one host thread per GPU, each "
                         "driving abc_generation_sharded_dev on its device-resident shard; the fallback of a bare `--gpus N` when "
                         "torch.distributed.run is not available")
    ap.add_argument("--rule", choices=["press", "wilcoxon"], default="wilcoxon",
                    help="PLS component rule of the timed generation (AbcUtil.cpp:447-449): wilcoxon = argmin PRESS reduced by the "
                         "Wilcoxon signed-rank test (SURVEY A.2; the default of the drop-in: C++ facade, shell, Python mirrors), "
                         "press = plain argmin PRESS; the other rule is timed in `extra`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs outside the timed region (sustained run, fp64 "
                    "kernel, INDEPENDENT noise, Wilcoxon rule, simple ranking, host-pointer call, log-normal alias build, scaling model)")
    ap.add_argument("--kde-mode", choices=["auto", "fp64"], default="auto",
                    help="weight kernel: auto = split-operand kernel where it applies (default), fp64 = the fp64 vector kernel (A/B runs)")
    ap.add_argument("--cpu-budget-s", type=float, default=25.0)
    ap.add_argument("--sustained-s", type=float, default=2.0)
    ap.add_argument("--prev-size", type=int, default=0,
                    help="size K' of the previous predictive prior (default: K = 0.1 x all particles, the stated configuration)")
    return ap.parse_args(argv)


# ---- how a rank talks to the others ---------------------------------------------------------------------------------------
class SoloEnv:
    """--gpus 1"""
    world, rank, local_rank, launcher = 1, 0, 0, "single GPU"

    def __init__(self):
        self.dev = "cuda:0"

    def context(self):
        from abcsmc_amd import _lib
        return _lib.default_context(0)

    def attach(self, ctx):
        return None

    def barrier(self):
        import torch
        torch.cuda.synchronize()

    def max_over_ranks(self, x):
        return float(x)

    def finish(self):
        pass


class TorchrunEnv:
    """one process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment)"""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.world = int(os.environ["WORLD_SIZE"])
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # ABC_BENCH_BACKEND=gloo: a dry run of the N > 1 path on a box with ONE GPU (all ranks on cuda:0, the C++ driver's collectives
        # forwarded to gloo as callbacks -- RCCL does not let two ranks share a device): exercises the sharded code path, not a measurement
        self.backend = os.environ.get("ABC_BENCH_BACKEND", "nccl")
        if self.backend == "gloo":
            self.local_rank = 0
        self.dev = "cuda:%d" % self.local_rank
        self.launcher = ("torch.distributed.run (started by bench.py itself)" if os.environ.get("ABC_BENCH_SELF_LAUNCHED")
                         else "torch.distributed.run")
        torch.cuda.set_device(self.local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self.backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(self.dev))
        self._dist, self._torch = dist, torch

    def context(self):
        from abcsmc_amd import _lib
        return _lib.default_context(self.local_rank)

    def attach(self, ctx):
        """the row-sharded driver inside the C ABI (abc_generation_sharded_dev): RCCL communicator created from an id that rank 0
        broadcasts; if RCCL cannot be initialised from the library, the same C++ driver runs with torch.distributed's (RCCL)
        collectives handed in as callbacks -- said loudly and recorded in the JSON line"""
        from abcsmc_amd import sharded
        dist, torch = self._dist, self._torch
        ctx.set_stream(torch.cuda.current_stream(torch.device(self.dev)).cuda_stream)
        ok = torch.ones(1, dtype=torch.int32, device="cpu" if self.backend == "gloo" else self.dev)
        try:
            if self.backend == "gloo":
                raise RuntimeError("dry run over gloo requested (ABC_BENCH_BACKEND)")
            sharded.attach_rccl(ctx, self.dev)
        except Exception as e:           # noqa: BLE001 -- any failure of the in-library communicator
            print("bench.py rank %d: in-library RCCL communicator failed (%s)" % (self.rank, e), file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)       # the choice is collective: every rank runs the same transport
        if int(ok.item()) == 1:
            return "rccl (C ABI, abc_comm_init_rank)"
        if self.rank == 0:
            print("bench.py: using torch.distributed collectives as callbacks of the C++ driver", file=sys.stderr)
        sharded.attach_torch_distributed(ctx, self.dev)
        return "torch.distributed callbacks (%s)" % self.backend

    def barrier(self):
        self._dist.barrier()
        self._torch.cuda.synchronize()

    def max_over_ranks(self, x):
        t = self._torch.tensor([x], dtype=self._torch.float64, device="cpu" if self.backend == "gloo" else self.dev)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def finish(self):
        self._dist.destroy_process_group()


class ThreadEnv:
    """--single-process: rank r is host thread r of this process, its context one of abc_ctx_create_multi's (joined by
    ncclCommInitAll); barriers and the max over ranks go through the threads' shared memory"""
    launcher = "one process, one host thread per GPU (abc_ctx_create_multi)"

    def __init__(self, shared, rank):
        import torch
        self.shared, self.rank, self.world, self.local_rank = shared, rank, shared["world"], rank
        self.dev = "cuda:%d" % rank
        torch.cuda.set_device(rank)
        self._torch = torch

    def context(self):
        return self.shared["multi"].context(self.rank)

    def attach(self, ctx):
        ctx.set_stream(self._torch.cuda.current_stream(self._torch.device(self.dev)).cuda_stream)
        return "rccl (C ABI, abc_ctx_create_multi / ncclCommInitAll)"

    def barrier(self):
        self._torch.cuda.synchronize()
        self.shared["barrier"].wait()

    def max_over_ranks(self, x):
        self.shared["vals"][self.rank] = float(x)
        self.shared["barrier"].wait()
        m = max(self.shared["vals"])
        self.shared["barrier"].wait()
        return m

    def finish(self):
        pass


def launcher_command(args_list, gpus, port):
    """what a bare `python bench.py --gpus N` starts (the driver's own N > 1 command line, with our arguments)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(args_list)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a launcher around it (no WORLD_SIZE): start the N ranks from here, BEFORE this process
    makes any GPU call (a child process, never an exec), relay the one JSON line and the exit status.  Returns the exit status."""
    import importlib.util
    import socket
    import subprocess
    if importlib.util.find_spec("torch.distributed.run") is None:
        print("bench.py: torch.distributed.run is not available -- running the %d ranks as host threads of this process "
              "(--single-process)" % args.gpus, file=sys.stderr)
        return run_threads(args)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    env["ABC_BENCH_SELF_LAUNCHED"] = "1"
    cmd = launcher_command(argv, args.gpus, port)
    try:
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)          # (stderr goes straight through)
    except OSError as e:
        print("bench.py: could not start %s: %s" % (" ".join(cmd[:4]), e), file=sys.stderr)
        return 2
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    other = [ln for ln in p.stdout.splitlines() if not ln.startswith('{"metric"')]
    if other:
        print("\n".join(other), file=sys.stderr)
    if p.returncode == 0 and len(lines) == 1:
        print(lines[0])
        return 0
    print("bench.py: the %d-rank run failed (exit status %d, %d JSON line(s)); command: %s"
          % (args.gpus, p.returncode, len(lines), " ".join(cmd)), file=sys.stderr)
    return p.returncode if p.returncode else 1


def run_threads(args):
    """--single-process: N host threads, one per GPU, each running the body of a rank (run_rank) on a context of
    abc_ctx_create_multi.  ctypes releases the GIL inside the library, so the ranks' generations overlap like those of N processes."""
    import threading
    import traceback
    import torch
    from abcsmc_amd import _lib
    n = args.gpus
    if torch.cuda.device_count() < n:
        print("bench.py: --single-process --gpus %d, but this process sees %d GPU(s)" % (n, torch.cuda.device_count()), file=sys.stderr)
        return 2
    try:
        multi = _lib.MultiContext(list(range(n)))
    except Exception as e:           # noqa: BLE001
        print("bench.py: abc_ctx_create_multi failed: %s" % e, file=sys.stderr)
        return 3
    shared = {"world": n, "multi": multi, "barrier": threading.Barrier(n), "vals": [0.0] * n}
    failed = []

    def body(r):
        try:
            run_rank(args, ThreadEnv(shared, r))
        except BaseException:        # noqa: BLE001 -- a rank that dies would leave the others inside a collective for ever
            traceback.print_exc()
            failed.append(r)
            sys.stderr.flush()
            sys.stdout.flush()
            os._exit(4)
    th = [threading.Thread(target=body, args=(r,), name="rank%d" % r) for r in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    multi.close()
    return 4 if failed else 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.single_process:
        return run_threads(args)
    if world_env is None and args.gpus > 1:
        # a bare `python bench.py --gpus N`: nothing has touched the GPU yet
        return self_launch(args, argv)
    if world_env is not None and int(world_env) != args.gpus:
        if int(os.environ.get("RANK", "0")) == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%s (the launcher's rank count and --gpus must agree)" % (args.gpus, world_env),
                  file=sys.stderr)
        return 2
    env = TorchrunEnv() if (world_env is not None and args.gpus > 1) else SoloEnv()
    run_rank(args, env)
    return 0


def run_rank(args, env):
    import numpy as np
    import torch
    from abcsmc_amd import _lib, abcutil, device, sharded, synthetic

    world, rank, dev = env.world, env.rank, env.dev
    RULE = _lib.RULE_WILCOXON if args.rule == "wilcoxon" else _lib.RULE_MIN_PRESS

    cfg = CONFIGS[args.config]
    M, P, A = cfg["M"], cfg["P"], cfg["A"]
    if args.scaling == "strong":
        N = cfg["N"]                                    # the stated total, whatever the number of GPUs
    else:
        N = (cfg["N"] // cfg["gpus"]) * world           # fixed particles per GPU
    row0, n_loc = split(N, world, rank)
    next0, nn_loc = split(N, world, rank)               # N_next = N
    K = N // 10                   # predictive-prior fraction 0.1 of the whole set
    # previous predictive prior: K' = K (the previous set has the size of this one).
    # --prev-size bounds it (sets may grow between generations, reference.json num_samples); the workload string then says so.
    Kp = args.prev_size if args.prev_size > 0 else K

    # ---- synthetic inputs, generated ON the device (counter-based: any shard reproduces any row), resident in HBM -----------
    wl = synthetic.Workload(M, P, seed=12345)
    dX, dY = wl.rows_device(row0, row0 + n_loc, dev)
    obs = wl.observed()
    spec = wl.prior_spec()
    dobs = device.colmajor(obs, dev)
    dpri = device.priors_to_device(_lib.make_priors(spec), dev)
    dtp, dwp, ddvp = wl.previous_set_device(Kp, dev)
    rng = abcutil.rng(67890)

    ctx = env.context()
    ctx.alias_stats(reset=True)
    ctx.set_kde_mode(_lib.KDE_FP64 if args.kde_mode == "fp64" else _lib.KDE_AUTO)
    comm_kind = None
    # (--single-process --gpus 1 still goes through the sharded driver and a one-rank RCCL communicator: the thread plumbing and
    # abc_ctx_create_multi can then be exercised on a one-GPU box)
    use_sharded = world > 1 or isinstance(env, ThreadEnv)
    if not use_sharded:
        gen = device.Generation(N, M, P, K, Kp, nn_loc, 0.5, A, rule=RULE, multivariate=True, device=dev, ctx=ctx)
    else:
        comm_kind = env.attach(ctx)
        gen = sharded.CabiShardedGeneration(ctx, dev, n_loc, M, P, K, Kp, nn_loc, 0.5, A, rule=RULE, multivariate=True,
                                            row0=row0, N_total=N, next0=next0, Nnext_total=N)

    def step():
        gen.run(dX, dY, dobs, dpri, rng, dtp, dwp, ddvp)

    barrier = env.barrier

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.timing_enable(2)             # HIP events on the stream the kernels are launched on: only the k_gram / k_kde brackets
    ctx.timing_read(reset=True)      # (a pair per STAGE costs ~10 us of dispatch gap each, ~0.15 ms per generation)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    stages = ctx.timing_read(reset=True)
    # per-stage breakdown (informational): a separate short pass with every stage timer on, outside the timed region
    nb = min(args.steps, 5)
    ctx.timing_enable(1)
    for _ in range(nb):
        step()
    barrier()
    stages_all = ctx.timing_read(reset=True)
    ctx.timing_enable(False)
    elapsed = env.max_over_ranks(elapsed)

    ms_per_step = 1e3 * elapsed / args.steps
    value = N / (elapsed / args.steps)

    def per_launch_ms(stage, st=None, steps=None):
        """average HIP-event bracket of one launch of a stage (ms) and launches per step, from the recorded counts"""
        ms, _, cnt = (st or stages)[stage]
        steps = steps or args.steps
        assert cnt % steps == 0 and cnt > 0, "stage %s: %d samples over %d steps (timer ring lost samples?)" % (stage, cnt, steps)
        return ms / cnt, cnt // steps

    event_overhead_ms = ctx.timing_overhead(50)      # what an event pair reports beyond the kernel itself (empty-kernel calibration)
    stage_ms = {k: round((v[0] + v[1]) / nb, 5) for k, v in stages_all.items()}
    stage_launches = {k: v[2] // nb for k, v in stages_all.items()}

    # ---- roofline of the DOMINANT kernel: the pair sums of the importance weights (k_kde_split / k_kde) ------------------
    # ALGORITHMIC work: K K' pairs x (3 P flops of the scaled squared distance + 1 exponential) -- `achieved`.
    # How it is done: k_kde_split takes the pair dot products as exact f16 limb products on the matrix
    # pipe -- 6 ceil(P/16) + 3 v_mfma_f32_32x32x16_{f16,bf16} per 32 x 32 pairs = 32 ISSUED flop per pair and MFMA
    # (`mfma_issue_frac`) -- and 2 vector instructions per pair (v_exp_f32 [8 issue cycles], f32 add) + 18 per batch of 16
    # pairs = 4.3 issue slots per pair.  k_kde (fp64 fallback): 1 add + PP FMAs + 13 for 2^x per pair, no matrix work.
    kde_bracket_ms, kde_launches = per_launch_ms("k_kde") if Kp else (0.0, 0)
    kde_ms = max(kde_bracket_ms - event_overhead_ms, 0.0)
    pairs = float(split(K, world, rank)[1]) * Kp
    flops_alg = pairs * (3.0 * P + 1.0)
    PPad = 2
    while PPad < P:
        PPad *= 2
    which = ctx.kde_last_kernel() if Kp else _lib.KDE_RAN_NONE
    issue_peak = 256 * 4 * 2.4e9 / 4.0                # wave-instructions per second: 1024 SIMDs, 4 cycles each, 2.4 GHz
    alg_tf = flops_alg / (kde_ms * 1e-3) / 1e12 if kde_ms > 0 else 0.0
    if which == _lib.KDE_RAN_SPLIT:
        nch_ = 1 if P <= 16 else (2 if P <= 32 else 4)          # 16-parameter chunks of the split kernel
        # (norm steps folded into spare K-slots where the last chunk has three; else, from 4e9 pairs, tiles in the order of the norm
        # tops and one step for top and batch reference)
        mfma_per_block = 6 * nch_ + (1 if P + 3 <= 16 * nch_ else (2 if pairs >= 4.0e9 else 3))
        flops_issued = pairs * mfma_per_block * 32.0   # 32 x 32 x 16 x 2 flop per MFMA over 1024 pairs
        # vector issue slots of 4 cycles per pair besides the MFMAs, counted in the kernel's ISA at 16 parameters: 52.5 vector
        # instructions per 1024 pairs (16 v_exp_f32 at two slots each, 15 f32 adds, 8 v_max3_f32, 13.5 others) = 68.5 slots / 16
        slots_per_pair = 4.28
        issued_tf = flops_issued / (kde_ms * 1e-3) / 1e12 if kde_ms > 0 else 0.0
        roofline = {"kernel": "k_kde_split", "bound": "mfma", "achieved": round(alg_tf, 1), "peak": MFMA_F16_PEAK_TF,
                    "unit": "TFLOP/s", "frac": round(alg_tf / MFMA_F16_PEAK_TF, 4),
                    "traffic": pmc_traffic("k_kde_split", args.config, world),
                    "flops_algorithmic": flops_alg, "frac_algorithmic": round(alg_tf / MFMA_F16_PEAK_TF, 4),
                    "flops_issued_mfma": flops_issued, "achieved_issued_mfma": round(issued_tf, 1),
                    "mfma_issue_frac": round(issued_tf / MFMA_F16_PEAK_TF, 4),
                    "issued_per_algorithmic_flop": round(flops_issued / flops_alg, 2),
                    # what bare bf16 MFMA loops sustain on random operands on this chip (the clock gives way under matrix load:
                    # /opt/skills/guides/MI355X_MICROARCH.md, 'DVFS give-back' item 1): the practical ceiling of issued matrix work
                    "mfma_sustained_random_data_TFLOPs": MFMA_SUSTAINED_TF,
                    "mfma_issue_vs_sustained": round(issued_tf / MFMA_SUSTAINED_TF, 4),
                    "algorithmic_vs_fp64_vector_peak": round(alg_tf / FP64_VALU_PEAK_TF, 3),
                    "mfma_32x32x16_per_1024_pairs": mfma_per_block,
                    "pairs_per_launch": pairs, "pairs_per_s": pairs / (kde_ms * 1e-3) if kde_ms > 0 else 0.0,
                    "valu_issue_slots_per_pair": slots_per_pair,
                    "valu_issue_frac": round((pairs / 64.0) * (slots_per_pair + mfma_per_block * 2.0 / 16.0) / (kde_ms * 1e-3) / issue_peak, 4)
                    if kde_ms > 0 else 0.0,
                    "note": "achieved / frac = ALGORITHMIC flops K K' (3P + 1) (SURVEY 8d) against the dense f16 MFMA peak, the pipe "
                            "the dot products run on; the fp64 operands travel as three f16 limbs, so the matrix pipe ISSUES "
                            "issued_per_algorithmic_flop times that (mfma_issue_frac); the exponentials run as v_exp_f32 on the "
                            "vector pipe (valu_issue_frac); weights within 5e-7 of the fp64 oracle by the kernel's error budget, largest seen 3.1e-7 (profiles/history/r03_kde_accuracy.json, r03_generation_fuzz.json).  The chip clocks down "
                            "under this kernel (profiles/: clock from GRBM_GUI_ACTIVE)"}
    else:
        instr_pair = 1 + PPad + 13
        roofline = {"kernel": "k_kde", "bound": "fp64_valu", "achieved": round(alg_tf, 2), "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
                    "frac": round(alg_tf / FP64_VALU_PEAK_TF, 4), "traffic": pmc_traffic("k_kde<", args.config, world),
                    "flops_algorithmic": flops_alg, "frac_algorithmic": round(alg_tf / FP64_VALU_PEAK_TF, 4),
                    "pairs_per_launch": pairs, "valu_instr_per_pair": instr_pair,
                    "valu_issue_frac": round((pairs / 64.0) * instr_pair / (kde_ms * 1e-3) / issue_peak, 4) if kde_ms > 0 else 0.0}
    roofline.update({"kernel_ms": round(kde_ms, 5), "launches_per_step": kde_launches, "event_bracket_ms": round(kde_bracket_ms, 5),
                     "event_overhead_ms": round(event_overhead_ms, 5), "share_of_step": round(kde_ms * kde_launches / ms_per_step, 4)})

    # ---- the dominant HBM kernel (k_gram): one launch per step reads X and Y (local rows) exactly once ---------------------
    gram_bracket_ms, gram_launches = per_launch_ms("k_gram")
    gram_ms = max(gram_bracket_ms - event_overhead_ms, 0.0)
    # (beyond 160 columns the set goes through column-group pairs, several launches: the algorithmic bytes of the set are
    # spread over them, i.e. `achieved` is then bytes of the set / total time of those launches)
    gram_bytes = 8.0 * n_loc * (M + P) / gram_launches
    gram_gbs = gram_bytes / (gram_ms * 1e-3) / 1e9 if gram_ms > 0 else 0.0
    roofline_hbm = {"kernel": "k_gram", "bound": "hbm", "achieved": round(gram_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gram_gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("k_gram", args.config, world),
                    "algorithmic_bytes_per_launch": gram_bytes, "kernel_ms": round(gram_ms, 5), "launches_per_step": gram_launches,
                    "event_bracket_ms": round(gram_bracket_ms, 5),
                    "achieved_event_bracket": round(gram_bytes / (gram_bracket_ms * 1e-3) / 1e9, 1) if gram_bracket_ms > 0 else 0.0}

    # ---- all streaming stages together = the step without the pair-sum kernel ------------------------------
    # Under the Wilcoxon rule the reference's path reads the validation rows once more ("an extra pass over the test rows":
    # per-observation residuals of every candidate count): + 8 N_v (M + P) bytes, N_v = the rows behind the training fraction.  The
    # scores this implementation writes and re-reads between its kernels are NOT algorithmic bytes.
    nv_loc = (n_loc - int(round(n_loc * 0.5))) if RULE == _lib.RULE_WILCOXON else 0

    def alg_bytes(kp, rule_pass=True):
        nn, k = nn_loc, K
        return (8.0 * n_loc * (M + P) + 8.0 * n_loc * M + 8.0 * n_loc + 16.0 * n_loc + 16.0 * k * P + nn * (16.0 * P + 8.0)
                + 8.0 * kp * P + 8.0 * (k + kp) + (8.0 * nv_loc * (M + P) if rule_pass else 0.0))
    stream_ms = ms_per_step - kde_ms * kde_launches
    # `achieved` / `frac`: SURVEY 8(d)'s formula AS WRITTEN (the implementation does not read the validation half of
    # X a second time -- the ranking's projection leaves the scores in the same pass -- so the rule's pass is not in the headline
    # bytes; the variant with it stays beside it under its own key, as rounds 4 and 5 reported it)
    stream_gbs = alg_bytes(Kp, False) / (stream_ms * 1e-3) / 1e9
    roofline_streaming = {"bound": "hbm", "algorithmic_bytes_per_step": alg_bytes(Kp, False), "bytes_per_particle": round(alg_bytes(Kp, False) / n_loc, 1),
                          "ms": round(stream_ms, 5), "achieved": round(stream_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(stream_gbs / HBM_PEAK_GBS, 4),
                          "frac_without_the_rules_pass": round(stream_gbs / HBM_PEAK_GBS, 4),
                          "algorithmic_bytes_with_the_rules_pass": alg_bytes(Kp),
                          "frac_with_the_rules_pass": round(alg_bytes(Kp) / (stream_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "note": "SURVEY 8(d) as written: B_alg / (wall time of a step minus the pair-sum kernel), per GPU; host work "
                                  "and launch gaps included.  frac_with_the_rules_pass adds 8 N_v (M + P) bytes under the Wilcoxon "
                                  "component rule -- the pass over the validation rows SURVEY A.2 names for the reference's path, which "
                                  "this implementation does not make (rounds 4-5 quoted that variant as `frac`)"}

    # set 0 (uniform weights, AbcUtil.cpp:539-545) has no O(K K') stage: reported separately, outside the timed region
    set0 = None
    sustained = None
    extra = None
    scaling_model = None
    if world == 1:
        rng0 = abcutil.rng(67890)
        gen0 = device.Generation(N, M, P, K, 0, nn_loc, 0.5, A, rule=RULE, multivariate=True, device=dev, ctx=ctx)
        for _ in range(2):
            gen0.run(dX, dY, dobs, dpri, rng0)
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            gen0.run(dX, dY, dobs, dpri, rng0)
        barrier()
        dt0 = (time.perf_counter() - t1) / 5
        g0 = alg_bytes(0, False) / dt0 / 1e9
        set0 = {"value": N / dt0, "unit": "particles/s", "ms_per_step": 1e3 * dt0,
                "roofline_streaming": {"algorithmic_bytes_per_step": alg_bytes(0, False), "achieved": round(g0, 1), "peak": HBM_PEAK_GBS,
                                       "unit": "GB/s", "frac": round(g0 / HBM_PEAK_GBS, 4)},
                "note": "first SMC set: rank + uniform weights + resample/perturb (no importance-weight stage)"}

    if world == 1 and not args.no_extra:
        # ---- sustained: back-to-back steps for >= --sustained-s seconds (the pair-sum kernel power-limits the chip: the timed
        # region above is 20 steps on a chip that was idle a moment ago) ---------------------------------------------------------
        ctx.timing_enable(2)
        ctx.timing_read(reset=True)
        ns = 0
        barrier()
        t1 = time.perf_counter()
        while True:
            for _ in range(25):
                step()
            ns += 25
            torch.cuda.synchronize()
            if time.perf_counter() - t1 >= args.sustained_s:
                break
        dts = time.perf_counter() - t1
        st_s = ctx.timing_read(reset=True)
        ctx.timing_enable(False)
        k_ms = (st_s["k_kde"][0] / st_s["k_kde"][2] - event_overhead_ms) if (Kp and st_s["k_kde"][2]) else 0.0
        g_ms = st_s["k_gram"][0] / max(st_s["k_gram"][2], 1) - event_overhead_ms
        sustained = {"seconds": round(dts, 3), "steps": ns, "ms_per_step": round(1e3 * dts / ns, 5), "value": N / (dts / ns),
                     "pair_sum_kernel_ms": round(k_ms, 5), "k_gram_ms": round(g_ms, 5),
                     "pair_sum_ms_vs_timed_region": round(k_ms / kde_ms, 4) if kde_ms > 0 else None,
                     "note": "same step, run back to back for >= %.1f s right after the timed region; the ratio of the pair-sum "
                             "brackets is the clock the chip holds under sustained load relative to the timed region" % args.sustained_s}
        extra = extra_legs(args, ctx, wl, dX, dY, dobs, dpri, dtp, dwp, ddvp, N, M, P, K, Kp, A, dev, event_overhead_ms)
        scaling_model = scaling_model_leg(args, dX, dY, dobs, dpri, dtp, dwp, ddvp, N, M, P, K, Kp, A, dev, RULE,
                                          ms_per_step, kde_ms * kde_launches, stage_ms, stage_launches, event_overhead_ms, step,
                                          min_press_step_ms=(extra or {}).get("min_press_rule_step_ms"))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        X, Y = dX.cpu().numpy().T, dY.cpu().numpy().T
        th_prev, w_prev, dv_prev = dtp.cpu().numpy().T, dwp.cpu().numpy(), ddvp.cpu().numpy()
        cpu = cpu_baseline(cfg, wl, X, Y, obs, spec, th_prev, w_prev, dv_prev, K, A, args.cpu_budget_s)

    if rank == 0:
        out = {
            "metric": "particles/sec per SMC generation (PLS+weight+resample), 1/2/4/8 GPU",
            "value": value, "unit": "particles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("f64 (inputs, model, distances, proposals, results; pair sums of the weight stage: fp64 operands as three "
                      "f16 limbs on the matrix pipe + f32 v_exp_f32, weights <= 5e-7 rel of the fp64 oracle, largest seen 3.1e-7)"
                      if which == _lib.KDE_RAN_SPLIT else "f64"),
            "data": "synthetic",
            "config": {"workload": cfg["name"] + ("" if Kp == K else " [previous predictive prior bounded to K' = %d]" % Kp)
                                   + ("" if args.scaling == "strong" or world == cfg["gpus"] else " [weak scaling: %d particles per GPU x %d GPUs]" % (N // world, world)),
                       "particles_per_gpu": n_loc, "particles_total": N, "metrics": M,
                       "params": P, "pls_components": A, "pred_prior_size": K, "prev_pred_prior_size": Kp,
                       "next_set_size": N, "noise": "MULTIVARIATE", "train_fraction": 0.5,
                       "ncomp_chosen": int(gen.ncomp if use_sharded else gen.ncomp.value),
                       "pls_component_rule": "wilcoxon" if args.rule == "wilcoxon" else "min_press",
                       "parallelism": "row-sharded x%d" % world, "launcher": env.launcher, "collectives": comm_kind,
                       "collectives_per_step": stage_launches.get("collectives", 0) if use_sharded else 0,
                       "collectives_ms_per_step": stage_ms.get("collectives", 0.0) if use_sharded else 0.0},
            "roofline": roofline,
            "roofline_hbm": roofline_hbm,
            "roofline_streaming": roofline_streaming,
            "set0": set0,
            "sustained": sustained,
            "extra": extra,
            "scaling_model": scaling_model,
            "cpu_baseline": cpu,
            "stage_ms_per_step": stage_ms,
        }
        print(json.dumps(out))
    env.finish()


def extra_legs(args, ctx, wl, dX, dY, dobs, dpri, dtp, dwp, ddvp, N, M, P, K, Kp, A, dev, event_overhead_ms):
    """Measurements outside the timed region (single GPU): what the reference does by default or at its own boundary, and the
    A/B legs the headline's labelling refers to."""
    import numpy as np
    import torch
    from abcsmc_amd import _lib, abcutil, device, sharded
    lib = _lib.lib()
    out = {}

    def timed(fn, reps=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t) / reps

    this = _lib.RULE_WILCOXON if args.rule == "wilcoxon" else _lib.RULE_MIN_PRESS       # the rule of the timed region
    # (1) the fp64 vector kernel on the same pairs (--kde-mode fp64): what the split-operand kernel is an alternative to
    if Kp:
        rng = abcutil.rng(67890)
        gen = device.Generation(N, M, P, K, Kp, N, 0.5, A, rule=this, multivariate=True, device=dev, ctx=ctx)
        ctx.set_kde_mode(_lib.KDE_FP64)
        reps = 3 if K * Kp <= 2e10 else 1
        gen.run(dX, dY, dobs, dpri, rng, dtp, dwp, ddvp)
        torch.cuda.synchronize()
        ctx.timing_enable(2)
        ctx.timing_read(reset=True)
        t = time.perf_counter()
        for _ in range(reps):
            gen.run(dX, dY, dobs, dpri, rng, dtp, dwp, ddvp)
        torch.cuda.synchronize()
        step_ms = 1e3 * (time.perf_counter() - t) / reps
        st = ctx.timing_read(reset=True)
        ctx.timing_enable(False)
        ctx.set_kde_mode(_lib.KDE_FP64 if args.kde_mode == "fp64" else _lib.KDE_AUTO)
        out["kde_fp64_ms"] = round(st["k_kde"][0] / max(st["k_kde"][2], 1) - event_overhead_ms, 5)
        out["kde_fp64_step_ms"] = round(step_ms, 5)
        # (2) noise = INDEPENDENT, the reference's default (AbcSmc.cpp:419, AbcSmc.h:159)
        geni = device.Generation(N, M, P, K, Kp, N, 0.5, A, rule=this, multivariate=False, device=dev, ctx=ctx)
        out["independent_noise_step_ms"] = round(timed(lambda: geni.run(dX, dY, dobs, dpri, rng, dtp, dwp, ddvp)), 5)
    # (2b) the OTHER component rule (AbcUtil.cpp:447-449; SURVEY A.2: argmin PRESS reduced by a Wilcoxon signed-rank test): the whole
    # generation and the PLS ranking alone under it -- `wilcoxon_*` when the timed region ran argmin PRESS, `min_press_*` otherwise
    other = _lib.RULE_MIN_PRESS if args.rule == "wilcoxon" else _lib.RULE_WILCOXON
    oname = "min_press" if args.rule == "wilcoxon" else "wilcoxon"
    rngo = abcutil.rng(67890)
    geno = device.Generation(N, M, P, K, Kp, N, 0.5, A, rule=other, multivariate=True, device=dev, ctx=ctx)
    out[oname + "_rule_step_ms"] = round(timed(lambda: geno.run(dX, dY, dobs, dpri, rngo, dtp, dwp, ddvp)), 5)
    out[oname + "_rule_ncomp"] = int(geno.ncomp.value)
    genor = device.Generation(N, M, P, K, 0, 0, 0.5, A, rule=other, multivariate=True, device=dev, ctx=ctx)
    out["ranking_pls_" + oname + "_ms"] = round(timed(lambda: genor.run(dX, dY, dobs, dpri, rngo)), 5)
    del geno, genor
    # (the Wilcoxon rule's work: one signed-rank test per (response j, candidate a' < a*_j) over the n - n_train validation rows;
    # the PRESS optima a*_j are read from a model record fitted under argmin PRESS through the staged entry points)
    bew = sharded.HipBackend(dev, ctx)
    wstats, wmodel = bew.zeros(bew.stats_len(M, P)), bew.zeros(bew.model_len(M, P, A) + 8)
    bew.stats_shift(dX, dY, wstats)
    bew.stats_accumulate(dX, dY, 0, int(round(N * 0.5)), wstats)
    bew.pls_model(wstats, dobs, M, P, A, _lib.RULE_MIN_PRESS, wmodel)
    torch.cuda.synchronize()
    Lm = bew.model_len(M, P, A)
    per = wmodel[Lm - P:Lm].cpu().numpy().astype(int)
    out["wilcoxon_tests"] = int(np.maximum(per - 1, 0).sum())
    out["wilcoxon_validation_rows"] = N - int(round(N * 0.5))
    del bew, wstats, wmodel
    # (3) particle_ranking_simple (AbcUtil.cpp:408-421), device resident, through the staged entry points: moments of the
    # metrics, z-scored distance to the observation, the K smallest
    be = sharded.HipBackend(dev, ctx)
    stats = be.zeros(be.stats_len(M, 0))
    model = be.zeros(be.model_len(M, 0, 0) + 8)
    dist_all = be.empty(N)
    sidx, sdist = be.empty(K, torch.int64), be.empty(K)

    def simple():
        h = be._s()
        ctx.check(lib.abc_stats_shift_dev(h, dX.data_ptr(), dX.data_ptr(), N, N, N, M, 0, stats.data_ptr()))
        ctx.check(lib.abc_stats_accumulate_dev(h, dX.data_ptr(), dX.data_ptr(), N, N, N, M, 0, 0, N, stats.data_ptr()))
        ctx.check(lib.abc_simple_model_dev(h, stats.data_ptr(), dobs.data_ptr(), M, 0, model.data_ptr()))
        ctx.check(lib.abc_project_distance_dev(h, dX.data_ptr(), N, N, M, 0, 0, model.data_ptr(), 1, dist_all.data_ptr()))
        be.select_smallest(dist_all, K, 0, sidx, sdist)
    out["ranking_simple_ms"] = round(timed(simple), 5)
    out["ranking_simple_particles_per_s"] = N / (out["ranking_simple_ms"] * 1e-3)
    # ... and the PLS ranking alone (no weights, no proposals), device resident
    genr = device.Generation(N, M, P, K, 0, 0, 0.5, A, rule=this, multivariate=True, device=dev, ctx=ctx)
    rngr = abcutil.rng(1)
    out["ranking_pls_ms"] = round(timed(lambda: genr.run(dX, dY, dobs, dpri, rngr)), 5)
    # (4) the drop-in call as the reference makes it (AbcUtil.h:149-153): host matrices in, host index vector out
    if N * (M + P) * 8 <= 2 << 30:
        X, Y, obs = dX.cpu().numpy().T, dY.cpu().numpy().T, dobs.cpu().numpy()
        ms = timed(lambda: abcutil.particle_ranking_PLS(X, Y, obs, 0.5, K=K, max_comp=A, rule=this, ctx=ctx), reps=3, warm=1)
        hbytes = 8.0 * N * (M + P)
        # PCIe time of the same bytes from pinned memory, measured here (the floor of any host-pointer call)
        pin = torch.empty(int(hbytes // 8), dtype=torch.float64).pin_memory()
        dst = torch.empty_like(pin, device=dev)
        pin.zero_()                                     # (every page touched before the first transfer)

        def one_copy():
            torch.cuda.synchronize()
            t = time.perf_counter()
            dst.copy_(pin, non_blocking=True)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t)
        one_copy()
        pcie = min(one_copy() for _ in range(5))        # the FLOOR: the fastest of five (one box's mean came out at 26 ms for 384 MB)
        out["host_entry_ms"] = round(ms, 4)
        out["host_entry_bytes"] = hbytes
        out["host_entry_pcie_floor_ms"] = round(pcie, 4)
        out["host_entry_over_pcie_floor"] = round(ms / pcie, 3)
        del pin, dst
    # (4b) WHAT A WRONG GUESS OF THE COMPONENT COUNT COSTS.  The timed region's synthetic responses never make
    # the Wilcoxon reduction lower the LARGEST per-response count, so its speculation (ranking on the fit's count beside the reduction)
    # is always right there.  Here the same workload with noise added to the responses until the rule's count falls below argmin
    # PRESS's: the step under the rule on that data, the argmin-PRESS step on the SAME data (the difference is the rule plus the
    # repair), how often the ranking / the whole generation was repeated per step (abc_generation_repeats), and the same for a first set.
    if args.rule == "wilcoxon":
        out["moved_count"] = moved_count_legs(ctx, dX, dY, dobs, dpri, dtp, dwp, ddvp, N, M, P, K, Kp, A, dev, timed)
    # (5) the resampling table (gsl_ran_discrete_preproc): device build (default; HIP-event bracket of its ten launches) and the
    # host's sequential build it replaces (host ms; the GPU idles behind it, plus two PCIe hops), on log-normal weights
    g = np.random.default_rng(5)
    r = abcutil.rng(3)
    for name, w in (("lognormal_sigma1.5", np.exp(1.5 * g.normal(size=K))), ("lognormal_sigma3", np.exp(3.0 * g.normal(size=K)))):
        dw = torch.from_numpy(w / np.linalg.norm(w)).to(dev)
        par = be.empty(1024, torch.int64)
        for mode, key, col in ((_lib.ALIAS_DEVICE, "alias_device_ms_", 0), (_lib.ALIAS_HOST, "alias_host_ms_", 1)):
            ctx.set_alias_mode(mode)
            be.resample(r, dw, 0, 1024, par)
            torch.cuda.synchronize()
            ctx.timing_enable(1)
            ctx.timing_read(reset=True)
            for _ in range(5):
                be.resample(r, dw, 0, 1024, par)
            torch.cuda.synchronize()
            st = ctx.timing_read(reset=True)
            ctx.timing_enable(False)
            out[key + name] = round(st["alias_host"][col] / max(st["alias_host"][2], 1), 5)
    ctx.set_alias_mode(_lib.ALIAS_DEVICE)
    builds, fallbacks = ctx.alias_stats(reset=True)           # over the WHOLE run: warm-up, timed region, sustained leg, these legs
    out["alias_device_builds"], out["alias_device_fallbacks"] = builds, fallbacks
    return out


def moved_count_data(ctx, dX, dY, dobs, N, M, P, K, Kp, A, dev):
    """A set that needs FEWER components than argmin PRESS keeps (for extra["moved_count"] and scripts/trace_step.py).  The centred
    metrics are projected onto a random r-dimensional subspace (r = 3, 2 in turn) and independent noise of half each column's factor
    share is put back on it (after z-scoring the same in every column: the best predictor of the factors then lies in their own r
    directions); the responses are random combinations of those r factors + N(0, 1) x their standard deviation.  r components carry
    everything the metrics know about the responses, the PRESS values beyond them are flat, argmin PRESS lands somewhere on the
    plateau -- the largest over the responses near its end -- and the Wilcoxon rule takes the surplus back.  (Lower-rank RESPONSES
    over the workload's own metrics do not move the count at 5e5 validation rows: with eight latent factors in the metrics every
    one of eight components is significant.)  Priors, previous set and its variances are taken from THAT data the way
    synthetic.Workload takes them from its own (uniform [mu - 6 sd, mu + 6 sd] / Gaussian(mu, 3 sd); K' rows shrunk halfway to the
    mean), so the proposals' acceptance is as in the timed region.
    Returns None when the rule's largest count stays argmin PRESS's at every r, else (r, dXn, dYn, dprin, prev)."""
    import torch
    from abcsmc_amd import _lib, abcutil, device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    rng = abcutil.rng(67890)
    gw = device.Generation(N, M, P, K, 0, 0, 0.5, A, rule=_lib.RULE_WILCOXON, multivariate=True, device=dev, ctx=ctx)
    gp = device.Generation(N, M, P, K, 0, 0, 0.5, A, rule=_lib.RULE_MIN_PRESS, multivariate=True, device=dev, ctx=ctx)
    mux, sdx = dX.mean(dim=1, keepdim=True), dX.std(dim=1, keepdim=True)
    muy, sdy = dY.mean(dim=1, keepdim=True), dY.std(dim=1, keepdim=True)
    f64 = dY.dtype
    dpri0 = device.priors_to_device(_lib.make_priors([(_lib.PRIOR_GAUSS, 0.0, 1.0)] * P), dev)
    for r in (3, 2):
        Bx = torch.linalg.qr(torch.randn((M, r), generator=g, device=dev, dtype=f64))[0]
        Z = Bx.T @ ((dX - mux) / sdx)                            # r x N factors of the z-scored metrics
        Z = Z / Z.std(dim=1, keepdim=True)
        rown = Bx.norm(dim=1, keepdim=True)
        dXn = Bx @ Z
        for j in range(M):                                       # (column by column: no second M x N temporary)
            dXn[j] += 0.5 * rown[j, 0] * torch.randn((N,), generator=g, device=dev, dtype=f64)
        dXn = (mux + sdx / (rown * 1.25 ** 0.5) * dXn).contiguous()
        Wy = torch.randn((P, r), generator=g, device=dev, dtype=f64) / r ** 0.5
        dYn = (muy + sdy * (Wy @ Z + torch.randn((P, N), generator=g, device=dev, dtype=f64))).contiguous()
        del Z
        gw.run(dXn, dYn, dobs, dpri0, rng)                       # (ranking only: the priors are not looked at)
        gp.run(dXn, dYn, dobs, dpri0, rng)
        torch.cuda.synchronize()
        if gw.ncomp.value < gp.ncomp.value:
            mun, sdn = dYn.mean(dim=1).cpu().numpy(), dYn.std(dim=1).cpu().numpy()
            spec = [(_lib.PRIOR_UNIF_REAL, mun[j] - 6 * sdn[j], mun[j] + 6 * sdn[j]) if j % 2 == 0 else (_lib.PRIOR_GAUSS, mun[j], 3 * sdn[j])
                    for j in range(P)]
            dprin = device.priors_to_device(_lib.make_priors(spec), dev)
            prev = ()
            if Kp:
                rows = torch.randint(0, N, (Kp,), generator=g, device=dev)
                mn = dYn.mean(dim=1, keepdim=True)
                th = (mn + 0.5 * (dYn[:, rows] - mn)).contiguous()
                prev = (th, torch.full((Kp,), 1.0 / Kp, dtype=f64, device=dev), 2.0 * th.var(dim=1, unbiased=True))
            return r, dXn, dYn, dprin, prev
        dXn = dYn = None
    return None


def moved_count_legs(ctx, dX, dY, dobs, dpri, dtp, dwp, ddvp, N, M, P, K, Kp, A, dev, timed):
    """extra["moved_count"]: see the caller and moved_count_data."""
    from abcsmc_amd import _lib, abcutil, device
    data = moved_count_data(ctx, dX, dY, dobs, N, M, P, K, Kp, A, dev)
    if data is None:
        return {"found": False}
    r, dXn, dYn, dprin, prev = data
    res = {"found": True, "factor_rank": r}
    rng = abcutil.rng(67890)
    gw = device.Generation(N, M, P, K, Kp, N, 0.5, A, rule=_lib.RULE_WILCOXON, multivariate=True, device=dev, ctx=ctx)
    gp = device.Generation(N, M, P, K, Kp, N, 0.5, A, rule=_lib.RULE_MIN_PRESS, multivariate=True, device=dev, ctx=ctx)
    gw0 = device.Generation(N, M, P, K, 0, N, 0.5, A, rule=_lib.RULE_WILCOXON, multivariate=True, device=dev, ctx=ctx)
    gp0 = device.Generation(N, M, P, K, 0, N, 0.5, A, rule=_lib.RULE_MIN_PRESS, multivariate=True, device=dev, ctx=ctx)

    def leg(gen_w, gen_p, args_prev):
        ctx.generation_repeats(reset=True)
        reps = 5 if K * max(Kp, 1) <= 2e10 else 2
        ms_w = timed(lambda: gen_w.run(dXn, dYn, dobs, dprin, rng, *args_prev), reps=reps, warm=2)
        rr, gr = ctx.generation_repeats(reset=True)
        ms_p = timed(lambda: gen_p.run(dXn, dYn, dobs, dprin, rng, *args_prev), reps=reps, warm=2)
        return {"moved_count_step_ms": round(ms_w, 5), "moved_count_ncomp": int(gen_w.ncomp.value), "fit_ncomp": int(gen_p.ncomp.value),
                "min_press_step_ms_same_data": round(ms_p, 5), "rule_and_repair_ms": round(ms_w - ms_p, 5),
                "ranking_repeats_per_step": rr / (reps + 2), "generation_repeats_per_step": gr / (reps + 2),
                "proposal_giveups_last_step": ctx.generation_giveups()}
    res.update(leg(gw, gp, prev))
    res["first_set"] = leg(gw0, gp0, ())
    res["note"] = ("metrics = factor_rank factors of the workload's + noise (half the factors' share per column), responses = combinations "
                   "of those factors + N(0,1) x their standard deviation: argmin PRESS keeps components "
                   "that only fit noise and the Wilcoxon rule takes them back, i.e. the ranking that ran beside the reduction on the "
                   "fit's count is repeated (ranking_repeats_per_step 1) -- the case the timed region's clean responses never reach; "
                   "rule_and_repair_ms = this step minus the argmin-PRESS step on the same data")
    return res


def predict_scaling(ms_per_step, kde_ms, sharded_ms, collectives, collective_ms, gpus=(2, 4, 8)):
    """STRONG scaling of one generation over G GPUs from single-GPU measurements (pure arithmetic, unit-tested on the CPU):
         t(G) = kde_ms / G                     the pair sums of the weight stage: K / G rows per rank
              + sharded_ms / G                 the row-proportional streaming kernels (Gram, projection, draws + proposals)
              + (ms_per_step - kde_ms - sharded_ms)      everything else is REPLICATED on every rank (model fit, selection,
                                                         gather, weight prologue / epilogue, alias table, host gaps)
              + collectives * collective_ms    the exchange steps, each priced at the measured world-1 latency of an RCCL call
                                               (a floor: real xGMI hops add to it)"""
    repl = ms_per_step - kde_ms - sharded_ms
    out = {}
    for g in gpus:
        t = kde_ms / g + sharded_ms / g + repl + collectives * collective_ms
        out[str(g)] = {"ms_per_step": round(t, 5), "speedup": round(ms_per_step / t, 3), "efficiency": round(ms_per_step / t / g, 4)}
    return out


def rccl_world1_latency(dev, messages, reps=20):
    """{message name: ms} of a one-rank RCCL collective of each message's size (torch.distributed, backend nccl = RCCL, a file store:
    no network) -- all-reduce for the names ending in all_reduce, all-gather otherwise; {} when the group cannot be set up"""
    import tempfile
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return {}
    out = {}
    store = tempfile.NamedTemporaryFile(prefix="abc_rccl_w1_", delete=False)
    store.close()
    os.unlink(store.name)
    try:
        torch.cuda.set_device(torch.device(dev))
        import datetime
        dist.init_process_group("nccl", init_method="file://" + store.name, world_size=1, rank=0, timeout=datetime.timedelta(seconds=60))
        for name, nbytes in messages.items():
            n = max(int(nbytes) // 8, 1)
            a = torch.zeros(n, dtype=torch.float64, device=dev)
            b = torch.empty_like(a)
            fn = (lambda: dist.all_reduce(a)) if name.endswith("all_reduce") else (lambda: dist.all_gather_into_tensor(b, a))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out[name] = round(1e3 * (time.perf_counter() - t) / reps, 5)
    except Exception:          # noqa: BLE001 -- no RCCL / no store: the model keeps its assumed price
        out = {}
    finally:
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:      # noqa: BLE001
            pass
        try:
            os.unlink(store.name)
        except OSError:
            pass
    return out


def scaling_model_leg(args, dX, dY, dobs, dpri, dtp, dwp, ddvp, N, M, P, K, Kp, A, dev, rule, ms_per_step, kde_ms, stage_ms,
                      stage_launches, event_overhead_ms, fused_step=None, min_press_step_ms=None):
    """The N = 1 line's PREDICTION of the strong-scaling curve, so that the first real multi-GPU run can be held against a
    stated number: stage times from this run + the latency of the sharded driver's collectives measured on a ONE-rank RCCL
    communicator (abc_comm_init_rank at world 1: every collective of the protocol is a real RCCL call on this GPU)."""
    import torch
    from abcsmc_amd import _lib, abcutil, sharded
    # row-proportional kernels, from the per-stage pass (each stage bracket carries one event pair of overhead)
    def st(name):
        return max(stage_ms.get(name, 0.0) - event_overhead_ms * stage_launches.get(name, 0), 0.0)
    sharded_ms = st("k_gram") + st("project_distance") + st("perturb")
    # collectives of one generation at world > 1 (the all-gathers of the ranks' statistics records, of their sorted
    # lists with their rows, and for weighted sets of the weight slices)
    coll_ms, ncoll, note, step1, measured = None, 3 if Kp else 2, None, None, 0
    # ... and under the Wilcoxon rule the all-reduces of its bounds cascade (wilcoxon.hip): level 0 always, ONE fine level counted
    # here (a second one and the all-gather of the undecided tests' keys only when statistics sit next to the threshold).  The rule's
    # time = this step minus the argmin-PRESS step measured beside it; its sweeps and scores are row-proportional, what is not
    # (plan, totals, bounds, the host's looks at the level counts) is ASSUMED_WILCOXON_REPLICATED_MS per level
    wilcoxon_ms, wilcoxon_sharded_ms = None, 0.0
    if rule == _lib.RULE_WILCOXON:
        ncoll += 2
        if min_press_step_ms is not None:
            wilcoxon_ms = max(ms_per_step - min_press_step_ms, 0.0)
            wilcoxon_sharded_ms = max(wilcoxon_ms - 2 * ASSUMED_WILCOXON_REPLICATED_MS, 0.0)
            sharded_ms += wilcoxon_sharded_ms
    ratios = []
    c1 = None
    try:
        c1 = _lib.Context(int(torch.device(dev).index or 0))
        c1.comm_init_rccl(1, 0, _lib.comm_unique_id())
        g1 = sharded.CabiShardedGeneration(c1, dev, N, M, P, K, Kp, N, 0.5, A, rule=rule, multivariate=True)
        r1 = abcutil.rng(67890)

        def block(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / reps
        sharded_step = lambda: g1.run(dX, dY, dobs, dpri, r1, dtp, dwp, ddvp)
        block(sharded_step, 2)
        # The sharded driver against the fused one in INTERLEAVED blocks (reported, not used: see below)
        s_blocks = []
        for _ in range(3):
            sb = block(sharded_step)
            s_blocks.append(sb)
            if fused_step is not None:
                ratios.append(sb / block(fused_step))
        ratios.sort()
        step1 = min(s_blocks)
        c1.timing_enable(1)
        c1.timing_read(reset=True)
        for _ in range(5):
            g1.run(dX, dY, dobs, dpri, r1, dtp, dwp, ddvp)
        torch.cuda.synchronize()
        stc = c1.timing_read(reset=True)
        c1.timing_enable(False)
        ms, _, cnt = stc["collectives"]
        if cnt:          # (at world 1 the driver skips the exchanges that have nobody to talk to: what runs is the pilot broadcast)
            measured = cnt // 5
            coll_ms = max(ms / cnt - event_overhead_ms, 0.0)
        c1.comm_destroy()
    except Exception as e:           # noqa: BLE001 -- RCCL not loadable on this box: the model is then quoted without the collectives
        note = "world-1 RCCL communicator unavailable (%s): collectives priced at 0" % e
    finally:
        if c1 is not None:
            c1.close()
    # the code that runs at G > 1 is the sharded driver: its own world-1 step (fewer overlaps than the fused single-GPU driver) is
    # the base of the prediction when it could be measured; a collective is priced at the larger of the measured world-1 RCCL call
    # and ASSUMED_XGMI_COLLECTIVE_MS (a one-GPU box cannot measure a hop over xGMI: world-1 RCCL calls return in ~0 us)
    # The base of the prediction is the timed region's own step plus the exchange steps' kernels that only exist at G > 1 (the list
    # header, unpack, merge, check and placement, the weight slices' unpadding: six launches at ~5 us, ASSUMED_EXCHANGE_KERNELS_MS).
    # The sharded driver's world-1 step is reported beside it but NOT used: on one GPU it runs the fused driver's kernels (same
    # timeline, scripts/trace_sharded.py), and which of two contexts of one process is faster depends on the hardware queues their
    # streams land on (four by default: a second context's side stream can share its main stream's queue and lose the overlap,
    # +0.09 .. 0.18 ms per generation; with GPU_MAX_HW_QUEUES=8 the routes agree: scripts/sharded_w1_time.py) -- not on the driver.
    # ... and what a one-rank RCCL communicator takes for the collectives of one generation AT THEIR REAL MESSAGE SIZES (round 6:
    # the sharded driver skips its exchanges at world 1, so the figure above stayed null): torch.distributed's nccl backend (= RCCL)
    # on a one-rank group, every message of the protocol timed on its own.  A FLOOR (launch + one local copy; a hop over xGMI adds
    # to it) -- the model prices a collective at the larger of this and ASSUMED_XGMI_COLLECTIVE_MS.
    C16 = 16 * ((M + P + 15) // 16)
    nv = N - int(round(N * 0.5))
    exch = {"statistics_records_all_gather": 8 * (2 + 3 * C16 + 2 * C16 * C16)}
    exch["sorted_lists_with_rows_all_gather"] = int((K / 8 + 8 * (K / 8) ** 0.5 + 64) * (16 + 8 * P))        # (per rank, at 8 ranks)
    if Kp:
        exch["weight_slices_all_gather"] = 8 * ((K + 7) // 8)
    if rule == _lib.RULE_WILCOXON:
        first = (2 if A > 17 else (4 if A <= 9 else 2)) * (A - 1)
        exch["wilcoxon_level0_counts_all_reduce"] = 8 * 192 * first
        exch["wilcoxon_fine_level_counts_all_reduce"] = 8 * 2048 * 8
    rccl_sizes = rccl_world1_latency(dev, exch)
    if rccl_sizes:
        coll_ms = sum(v for v in rccl_sizes.values()) / len(rccl_sizes)
        measured = len(rccl_sizes)
    base = ms_per_step + ASSUMED_EXCHANGE_KERNELS_MS
    price = max(coll_ms or 0.0, ASSUMED_XGMI_COLLECTIVE_MS)
    pred = predict_scaling(base, kde_ms, sharded_ms, ncoll, price)
    for g, v in pred.items():              # speed-up and efficiency against the fused single-GPU step (what --gpus 1 measures)
        v["speedup"] = round(ms_per_step / v["ms_per_step"], 3)
        v["efficiency"] = round(ms_per_step / v["ms_per_step"] / int(g), 4)
    return {"scaling": "strong", "from": {"ms_per_step": round(ms_per_step, 5), "pair_sums_ms": round(kde_ms, 5),
                                          "row_sharded_streaming_ms": round(sharded_ms, 5),
                                          "replicated_ms": round(base - kde_ms - sharded_ms, 5),
                                          "collective_price_ms": round(price, 5),
                                          "collectives_per_step": ncoll,
                                          "pls_component_rule": "wilcoxon" if rule == _lib.RULE_WILCOXON else "min_press",
                                          "wilcoxon_rule_ms": None if wilcoxon_ms is None else round(wilcoxon_ms, 5),
                                          "wilcoxon_row_sharded_ms": round(wilcoxon_sharded_ms, 5),
                                          "rccl_world1_collective_ms": None if coll_ms is None else round(coll_ms, 5),
                                          "rccl_world1_ms_by_message": rccl_sizes or None,
                                          "exchanged_bytes_per_rank_at_8_ranks": exch,
                                          "rccl_world1_collectives_measured_per_step": measured,
                                          "sharded_driver_world1_step_ms": None if step1 is None else round(step1, 5),
                                          "sharded_over_fused_step_ratio": (round(ratios[len(ratios) // 2], 4) if ratios else None)},
            "predicted": pred,
            "formula": "t(G) = pair_sums/G + row_sharded_streaming/G + replicated + collectives_per_step x collective_price_ms; "
                       "replicated = ms_per_step + %.2f ms of exchange kernels (assumed) - pair_sums - row_sharded_streaming" % ASSUMED_EXCHANGE_KERNELS_MS,
            "note": note or "a prediction to hold the first measured curve against, not a measurement; the replicated chain (model fit, "
                            "selection, alias table, weight prologue / epilogue) is the Amdahl term"}


def cpu_baseline(cfg, wl, X, Y, obs, spec, th_prev, w_prev, dv_prev, K, A, budget_s):
    """Single-threaded CPU oracle (oracle/, a restatement of the reference path: the reference itself
    cannot be built here) timed on a bounded sample of the same workload."""
    import numpy as np
    from oracle import pyoracle as O
    N = X.shape[0]
    pri = O.make_priors(spec)
    # the O(K K' P) weight stage dominates on the CPU: cap K, K' so the whole sample takes ~budget
    # (~1.3e8 pdf evaluations/s/core measured on this class of host)
    cap = int(min(K, max(1000, (budget_s * 0.6 * 1.0e8 / max(1, X.shape[1] // 2)) ** 0.5)))
    Ks, Kps = min(K, cap), min(th_prev.shape[0], cap)
    # the ranking and the proposals are timed on at most 1e6 rows (linear in the rows) and scaled
    Ns = min(N, 1_000_000)
    t0 = time.perf_counter()
    r = O.particle_ranking_pls(X[:Ns], Y[:Ns], obs, 0.5, A)
    t_rank = (time.perf_counter() - t0) * (N / Ns)
    idx = r["idx"][:Ks].astype(np.int64)
    theta = np.asfortranarray(Y[idx])
    t0 = time.perf_counter()
    O.doubled_variance(theta)
    w = O.weights_importance(pri, theta, th_prev[:Kps], w_prev[:Kps], dv_prev)
    t_w = time.perf_counter() - t0
    t0 = time.perf_counter()
    rc, L, _ = O.mvn_setup(theta)
    rng = O.rng(67890)
    O.sample_mvn_predictive_priors(rng, Ns, w, theta, pri, L)
    t_s = (time.perf_counter() - t0) * (N / Ns)
    # scale the weight stage to the full K x K' pair count (labelled extrapolation when capped)
    scale = (K / Ks) * (th_prev.shape[0] / Kps)
    total = t_rank + t_w * scale + t_s
    return {"value": N / total, "unit": "particles/s", "cores": 1, "kind": "port",
            "sample": ("ranking and resample+perturb timed on %d of %d rows (linear, scaled x%.1f); weight stage timed at "
                       "K=%d x K'=%d of %d x %d pairs and scaled by %.1fx (extrapolated)"
                       % (Ns, N, N / Ns, Ks, Kps, K, th_prev.shape[0], scale)),
            "seconds": {"rank_pls": round(t_rank, 3), "weights_sampled": round(t_w, 3),
                        "weights_scaled": round(t_w * scale, 3), "resample_perturb": round(t_s, 3)},
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    sys.exit(main())
