"""World-size-2 coverage of the row-sharded generation (abcsmc_amd/sharded.py).
CPU: gloo + the test-only numpy stage backend (orchestration, offsets, collectives) vs the
single-process oracle.  GPU: gloo + the HIP backend, both ranks sharing cuda:0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# BASELINE configs[3] / configs[4] column shapes (64 metrics x 32 parameters, 8 components; 128 metrics, 32 components)
SHAPES = {"small": "1500,12,5,4,500,300,1000", "wx": "2100,10,3,6,400,300,900", "config4": "1500,64,32,8,400,300,1200", "config5": "1400,128,16,32,300,250,1000",
          # large enough for the local-top selection of the C++ driver (N / W >= 4096): 2 x 40000 rows, 8000 kept
          "big": "40000,32,16,8,8000,3000,20000", "big0": "40000,32,16,8,8000,0,20000",
          # ... and for the resampling table to be built on the device (from 20000 entries) on every rank
          "huge": "120000,32,16,8,24000,1000,20000",
          # 40 parameters: each rank's row slice of the pair sums on the four-chunk split-operand kernel (33..64 parameters)
          "p40": "1600,48,40,6,450,380,1100",
          # large enough for the Wilcoxon rule's bounds cascade over the shards (>= 16384 validation rows in all): 2 x 4e4 rows at
          # configs[2]'s column shape, and the configs[3] / configs[4] column shapes (up to 224 / 496 tests)
          "wxbig": "40000,32,16,8,2000,500,3000", "wxc4": "30000,64,32,8,1500,400,2000", "wxc5": "30000,128,16,32,1500,400,2000",
          # three ranks, uneven shards (4 : 2 : 1): 15000 rows, the smallest shard shorter than the local-top list
          "w3": "5000,64,32,8,6000,400,2000", "w3wx": "14000,64,32,8,3000,400,2000"}


def _launch(backend, tmp_path, port, shape="small", rule="press", data="plain", world=2, split="even", env_extra=None):
    out = str(tmp_path / ("sharded_%s_%s_%s_%s_%d_%s.json" % (backend, shape, rule, data, world, split)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(env_extra or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_sharded_worker.py"), backend, out, SHAPES[shape], rule, data, split]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.load(open(out))


def _launch_many(backend, tmp_path, port, cases, world=2, env_extra=None):
    """several cases -- (shape, rule, data, split) -- on ONE launch of the ranks (round 6: starting them is 3-4 s of every test);
    returns their results in order"""
    import json as _json
    out = str(tmp_path / ("sharded_many_%s_%d_%d.json" % (backend, world, port)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update(env_extra or {})
    first, more = cases[0], [[SHAPES[c[0]], c[1], c[2], c[3]] for c in cases[1:]]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_sharded_worker.py"), backend, out, SHAPES[first[0]], first[1], first[2], first[3]] + ([_json.dumps(more)] if more else [])
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.load(open(out))
    return res["cases"] if more else [res]


def _check(res):
    assert res["ncomp"][0] == res["ncomp"][1]
    assert res["idx_equal"] and res["theta_equal"]
    assert res["w_maxrel"] < 1e-6 and res["dv_maxrel"] < 1e-9
    assert res["parent_equal"] and res["seeds_equal"] and res["rng_equal"] and res["next_finite"]


@pytest.mark.parametrize("shape,port", [("small", 29611), ("config4", 29613)])
def test_sharded_world2_gloo_cpu(tmp_path, shape, port):
    _check(_launch("numpy", tmp_path, port, shape))


@pytest.mark.gpu
def test_sharded_world2_gloo_hip(tmp_path):
    """(the three shapes small / config4 / config5 on one launch of the two ranks)"""
    shapes = ("small", "config4", "config5")
    for shape, res in zip(shapes, _launch_many("hip", tmp_path, 29612, [(sh, "press", "plain", "even") for sh in shapes])):
        try:
            _check(res)
        except AssertionError as e:
            raise AssertionError("shape %s: %s" % (shape, res)) from e


@pytest.mark.gpu
def test_sharded_world2_cabi_driver(tmp_path):
    """abc_generation_sharded_dev (the C++ driver behind the C ABI) on two ranks sharing cuda:0, its collectives forwarded to
    gloo through abc_comm_init_callbacks: the same checks against the single-process oracle as the Python driver -- the shapes
    small / config4 / config5 / p40 on one launch of the two ranks"""
    shapes = ("small", "config4", "config5", "p40")
    for shape, res in zip(shapes, _launch_many("cabi", tmp_path, 29616, [(sh, "press", "plain", "even") for sh in shapes])):
        try:
            _check(res)
        except AssertionError as e:
            raise AssertionError("shape %s: %s" % (shape, res)) from e


@pytest.mark.gpu
@pytest.mark.parametrize("data,port", [("plain", 29631), ("ties", 29633)])
def test_sharded_world2_cabi_driver_local_top_selection(tmp_path, data, port):
    """sets large enough for the C++ driver's local-top selection (every rank's K / W + 8 sigma smallest distances, sorted, with
    their rows in ONE all-gather; the merge of the W runs on every rank; the rule that no rank may hold an unlisted key at or below
    the K-th): weighted and first-set generations against the single-process oracle; with massively tied distances the rule
    fails, every rank takes the same decision and the generation repeats itself with the radix protocol (weighted: at the host's
    wait for the weights; set 0: at its end) -- same results"""
    shapes = ("big", "big0", "huge")                      # (one launch of the two ranks for the three shapes: round 6)
    for shape, res in zip(shapes, _launch_many("cabi", tmp_path, port, [(sh, "press", data, "even") for sh in shapes])):
        try:
            _check(res)
        except AssertionError as e:
            raise AssertionError("shape %s: %s" % (shape, res)) from e
        calls = res["comm_calls"]
        total = sum(calls.values())
        if data == "plain":
            # the all-gather of the ranks' statistics records (each about its own pilot shift), the all-gather of the ranks' sorted
            # lists with their rows, and (weighted generations) the all-gather of the raw weight slices: 3 collectives, 2 in the first set
            assert {k: v for k, v in calls.items() if v} == {"all_gather": 2 if shape == "big0" else 3}, (shape, calls)
        else:
            assert total > 10 and calls["all_reduce"] >= 6, (shape, calls)           # ... + the whole radix protocol of the repeat


@pytest.mark.gpu
@pytest.mark.parametrize("shape,port", [("small", 29641), ("big", 29642), ("config4", 29643), ("config5", 29644), ("huge", 29645)])
def test_sharded_world2_rccl_two_gpus(tmp_path, shape, port):
    """abc_comm_init_rank at world size 2, one GPU per rank: every collective of the sharded generation through RCCL itself
    (the callbacks tests share one GPU, which RCCL does not allow) -- the radix protocol (small), the local-top selection
    (big), the BASELINE configs[3] / configs[4] column shapes (64 metrics x 32 parameters; 128 metrics, 32 components) and a set
    whose resampling table is built on the device on every rank (huge).  Skipped on a one-GPU box; runs on any box with two."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (found %d)" % torch.cuda.device_count())
    _check(_launch("rccl", tmp_path, port, shape))


@pytest.mark.gpu
def test_sharded_world2_cabi_driver_wilcoxon_rule(tmp_path):
    """the Wilcoxon component rule (AbcUtil.cpp:447-449) on two ranks: the validation rows of both shards are gathered and
    ranked together on every rank; component count, selection and everything downstream equal the single-process oracle's
    (shapes small / wx on one launch of the ranks)"""
    for shape, res in zip(("small", "wx"), _launch_many("cabi", tmp_path, 29621, [("small", "wilcoxon", "plain", "even"), ("wx", "wilcoxon", "plain", "even")])):
        try:
            _check(res)
        except AssertionError as e:
            raise AssertionError("shape %s: %s" % (shape, res)) from e


@pytest.mark.gpu
@pytest.mark.parametrize("world,cases,port", [(2, (("wxbig", "even"), ("wxc4", "even"), ("wxc5", "even"), ("wxbig", "uneven")), 29651),
                                              (3, (("w3wx", "uneven"),), 29655)])
def test_sharded_cabi_driver_wilcoxon_cascade_over_the_shards(tmp_path, world, cases, port):
    """the Wilcoxon rule on sets large enough for its bounds cascade (wilcoxon.hip, round 5): every rank sweeps ITS validation rows,
    the counts of a level are all-reduced, verdicts are replicated, only the keys of the tests the bounds leave undecided are
    all-gathered -- component count, selection and everything downstream equal the single-process oracle's; no rank gathers rows
    (the exchange: the statistics records, the cascade's all-reduces and at most one key gather per batch of undecided tests, the
    sorted lists, the weight slices); even and uneven shards, two and three ranks"""
    results = _launch_many("cabi", tmp_path, port, [(sh, "wilcoxon", "plain", sp) for sh, sp in cases], world)     # (one launch of the ranks)
    for (shape, split), res in zip(cases, results):
        try:
            _check(res)
        except AssertionError as e:
            raise AssertionError("shape %s, %s shards: %s" % (shape, split, res)) from e
        calls = res["comm_calls"]
        # all-reduces: the cascade's levels only; all-gathers: statistics, lists + rows, weight slices (+ the undecided tests' keys,
        # in batches of eight) -- the row gather of rounds 1-4 took two more all-gathers and a count exchange
        if split == "even":   # (shares of 2 : 1 and beyond: the largest shard holds more winners than its local-top list is long, the
            #                   generation repeats with the radix protocol and its six all-reduces -- same results, checked above)
            # (round 6: the largest count first -- one all-reduce when a picked response keeps its optimum at level 0; at most two
            # fine levels for them, level 0 of the others, two fine levels)
            assert 1 <= calls["all_reduce"] <= 6, (shape, calls)
            assert 3 <= calls["all_gather"] <= 3 + 4, (shape, calls)


@pytest.mark.gpu
@pytest.mark.parametrize("shape,world,data,port", [("wxc5", 2, "plain", 29661), ("w3wx", 3, "noisy", 29662)])
def test_sharded_wilcoxon_cascade_in_batches_on_uneven_shards(tmp_path, shape, world, data, port):
    """a level of the cascade that does not fit the sweeps' counter buffer is cut into batches, one all-reduce
    each -- and the cut was made from the RANK'S OWN validation rows, which differ between ranks (the validation rows are the global
    tail: with shares of 2 : 1 the leading rank holds a third of them, with 4 : 2 : 1 none), so ranks could issue different numbers of
    collectives of different sizes.  The batches now follow from the validation rows of all ranks.  Here: the buffer capped at 64 KB
    (ABC_WX_BC_CAP_KB) and the bounds switched off (ABC_WX_NOBOUNDS: every test stays open, so every level runs over every test it
    is given -- the picked responses' through two fine levels, then all the others', up to 496 tests in batches of a few work-groups'),
    uneven shards on two and three ranks, clean and noisy responses -- every output the single-process oracle's, and many more
    all-reduces than levels."""
    res = _launch("cabi", tmp_path, port, shape, "wilcoxon", data, world, "uneven", {"ABC_DIAG": "1", "ABC_WX_BC_CAP_KB": "64", "ABC_WX_NOBOUNDS": "1"})
    _check(res)
    assert res["comm_calls"]["all_reduce"] >= 8, res["comm_calls"]


@pytest.mark.gpu
def test_sharded_statistics_merge_of_a_column_constant_within_every_shard(tmp_path):
    """a metric that takes one value per shard (sorted or blocked data): every rank's centred sums and products of that column are
    exactly zero, the merged record's entries for it consist of the re-centring terms alone -- the merge takes which entries exist
    from the block structure, not from the values"""
    _check(_launch("cabi", tmp_path, 29659, "small", "press", "blocked"))


@pytest.mark.gpu
def test_sharded_wilcoxon_cascade_equals_the_row_gather(tmp_path):
    """... and the same generation with the validation rows gathered on every rank instead (ABC_WX_GATHER: the path of rounds 1-4,
    still the fallback for small sets): the same outputs against the oracle, more collectives"""
    res = _launch("cabi", tmp_path, 29656, "wxbig", "wilcoxon", "plain", 2, "even", {"ABC_DIAG": "1", "ABC_WX_GATHER": "1"})
    _check(res)
    assert res["comm_calls"]["all_reduce"] == 0 and res["comm_calls"]["all_gather"] >= 5, res["comm_calls"]


@pytest.mark.gpu
@pytest.mark.parametrize("shape,rule,port", [("w3", "press", 29657), ("big", "press", 29658)])
def test_sharded_cabi_driver_uneven_shards(tmp_path, shape, rule, port):
    """uneven shards: three ranks holding 4 : 2 : 1 of the rows at configs[3]'s column shape -- the smallest shard is
    shorter than the local-top list, lists ALL its rows and is exhaustive; the largest holds more winners than its list is long,
    the rule says so on every rank alike and the generation repeats with the radix protocol -- and two ranks at 2 : 1 where the
    lists suffice; results = the single-process oracle's either way"""
    world = 3 if shape == "w3" else 2
    _check(_launch("cabi", tmp_path, port, shape, rule, "plain", world, "uneven"))


@pytest.mark.gpu
def test_cabi_rccl_world1_and_multi_context(gpu_ctx):
    """RCCL inside the C ABI on the one GPU of the box: a one-rank communicator from abc_comm_unique_id /
    abc_comm_init_rank runs every collective of the sharded generation through RCCL, and abc_ctx_create_multi +
    abc_generation_multi (host pointers, ncclCommInitAll) -- both must reproduce abc_generation_dev bit for bit"""
    import ctypes as C
    import numpy as np
    import torch
    from abcsmc_amd import _lib, abcutil, device, sharded, synthetic
    N, M, P, K, Kp, Nn, A = 6000, 32, 16, 700, 500, 5000, 8
    wl = synthetic.Workload(M, P)
    X, Y = wl.rows(0, N)
    obs, spec = wl.observed(), wl.prior_spec()
    thp, wp, dvp = wl.previous_set(Kp)
    dev = "cuda:0"
    args = [device.colmajor(a, dev) for a in (X, Y, obs)]
    pri = device.priors_to_device(_lib.make_priors(spec), dev)
    prev = [device.colmajor(a, dev) for a in (thp, wp, dvp)]
    g1 = device.Generation(N, M, P, K, Kp, Nn, 0.5, A, device=dev, ctx=gpu_ctx)
    r1 = abcutil.rng(5)
    g1.run(*args, pri, r1, *prev)
    torch.cuda.synchronize()
    # (a) one-rank RCCL communicator
    ctx = _lib.Context(0)
    ctx.comm_init_rccl(1, 0, _lib.comm_unique_id())
    assert ctx.comm_info() == (_lib.COMM_RCCL, 1, 0)
    g2 = sharded.CabiShardedGeneration(ctx, dev, N, M, P, K, Kp, Nn, 0.5, A)
    r2 = abcutil.rng(5)
    g2.run(*args, pri, r2, *prev)
    torch.cuda.synchronize()
    for a, b in ((g1.idx, g2.idx), (g1.w, g2.w), (g1.dv, g2.dv), (g1.parent, g2.parent), (g1.seeds, g2.seeds),
                 (g1.next, g2.next), (g1.theta, g2.theta), (g1.L, g2.L)):
        assert torch.equal(a, b)
    assert (r1.s1, r1.s2, r1.s3) == (r2.s1, r2.s2, r2.s3) and g2.ncomp == g1.ncomp.value
    ctx.comm_destroy()
    assert ctx.comm_info() == (_lib.COMM_NONE, 1, 0)
    ctx.close()
    # (b) one process, "several" GPUs (one here), host pointers
    mc = _lib.MultiContext([0])
    cfg = _lib.GenerationCfg(N, M, P, K, Kp, Nn, 0.5, A, _lib.RULE_MIN_PRESS, 1, 0)
    host = dict(idx=np.zeros(K, np.uint64), dist=np.zeros(K), theta=np.zeros((K, P), order="F"), w=np.zeros(K), dv=np.zeros(P),
                L=np.zeros((P, P), order="F"), next=np.zeros((Nn, P), order="F"), parent=np.zeros(Nn, np.uint64),
                seeds=np.zeros(Nn, np.uint64))
    io = _lib.GenerationIO()
    keep = [np.asfortranarray(X), np.asfortranarray(Y), np.ascontiguousarray(obs), _lib.make_priors(spec),
            np.asfortranarray(thp), np.ascontiguousarray(wp), np.ascontiguousarray(dvp)]
    io.X, io.Y, io.obs = (a.ctypes.data for a in keep[:3])
    io.priors = C.addressof(keep[3])
    io.theta_prev, io.w_prev, io.dv_prev = (a.ctypes.data for a in keep[4:])
    for k, v in host.items():
        setattr(io, k, v.ctypes.data)
    r3 = abcutil.rng(5)
    assert mc.generation(cfg, io, r3) == g1.ncomp.value
    assert np.array_equal(host["idx"], g1.idx.cpu().numpy().astype(np.uint64))
    assert np.array_equal(host["w"], g1.w.cpu().numpy()) and np.array_equal(host["theta"], device.to_numpy(g1.theta))
    assert np.array_equal(host["parent"], g1.parent.cpu().numpy().astype(np.uint64))
    assert np.array_equal(host["seeds"], g1.seeds.cpu().numpy().astype(np.uint64))
    assert np.array_equal(host["next"], device.to_numpy(g1.next)) and np.array_equal(host["L"], device.to_numpy(g1.L))
    assert (r3.s1, r3.s2, r3.s3) == (r1.s1, r1.s2, r1.s3)
    mc.close()


@pytest.mark.gpu
def test_sharded_world1_equals_fused(gpu_ctx):
    import numpy as np
    import torch
    from abcsmc_amd import _lib, abcutil, device, sharded, synthetic
    N, M, P, K, Kp, Nn, A = 6000, 32, 16, 700, 500, 6000, 8
    wl = synthetic.Workload(M, P)
    X, Y = wl.rows(0, N)
    dev = "cuda:0"
    args = [device.colmajor(a, dev) for a in (X, Y, wl.observed())]
    pri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
    prev = [device.colmajor(a, dev) for a in wl.previous_set(Kp)]
    g1 = device.Generation(N, M, P, K, Kp, Nn, 0.5, A, rule=_lib.RULE_MIN_PRESS, device=dev, ctx=gpu_ctx)     # (the stage-level driver's rule)
    r1 = abcutil.rng(5)
    g1.run(*args, pri, r1, *prev)
    g2 = sharded.ShardedGeneration(sharded.HipBackend(dev, gpu_ctx), N, M, P, K, Kp, Nn, 0.5, A, rule=_lib.RULE_MIN_PRESS)
    r2 = abcutil.rng(5)
    g2.run(*args, pri, r2, *prev)
    torch.cuda.synchronize()
    for a, b in ((g1.idx, g2.idx), (g1.w, g2.w), (g1.dv, g2.dv), (g1.parent, g2.parent), (g1.seeds, g2.seeds),
                 (g1.next, g2.next), (g1.theta, g2.theta)):
        assert torch.equal(a, b)
    assert (r1.s1, r1.s2, r1.s3) == (r2.s1, r2.s2, r2.s3)


@pytest.mark.gpu
def test_a_raising_collective_callback_fails_the_generation():
    """abc_comm_init_callbacks with Python collectives: an exception inside one (transport timeout, shape error) must not be
    swallowed by ctypes (which would report success and let the driver carry on with un-reduced statistics): the generation
    returns ABC_ERR_COMM and the exception is kept on the context"""
    import torch
    from abcsmc_amd import _lib, abcutil, device, sharded, synthetic
    N, M, P, K, Kp, Nn, A = 3000, 12, 5, 300, 200, 1000, 4
    wl = synthetic.Workload(M, P)
    X, Y = wl.rows(0, N)
    dev = "cuda:0"
    args = [device.colmajor(a, dev) for a in (X, Y, wl.observed())]
    pri = device.priors_to_device(_lib.make_priors(wl.prior_spec()), dev)
    prev = [device.colmajor(a, dev) for a in wl.previous_set(Kp)]
    ctx = _lib.Context(0)
    calls = []

    def ok_reduce(buf, count, dtype, stream):
        calls.append("reduce")
        return 0

    def bad_gather(send, recv, nbytes, stream):
        calls.append("gather")
        raise TimeoutError("peer did not answer")

    def ok_broadcast(buf, nbytes, root, stream):
        calls.append("broadcast")
        return 0

    # rank 0 of a world of two: the first exchange of a generation is the all-gather of the statistics records
    ctx.comm_init_callbacks(2, 0, ok_reduce, bad_gather, ok_broadcast)
    g = sharded.CabiShardedGeneration(ctx, dev, N, M, P, K, Kp, Nn, 0.5, A)
    with pytest.raises(_lib.AbcError) as e:
        g.run(*args, pri, abcutil.rng(5), *prev)
    assert e.value.code == -6 and "all_gather" in str(e.value)                  # ABC_ERR_COMM
    assert isinstance(ctx.comm_callback_error, TimeoutError) and calls == ["gather"]
    torch.cuda.synchronize()
    ctx.close()
