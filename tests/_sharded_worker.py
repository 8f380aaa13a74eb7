"""Worker for the world_size-2 tests of the row-sharded generation (launched by torch.distributed.run).
argv: backend ("numpy" -> CPU tensors + gloo; "hip" -> cuda:0 tensors + gloo, both ranks on one GPU, stage by stage from
Python; "cabi" -> the same two ranks through abc_generation_sharded_dev with the gloo collectives handed in as callbacks;
"rccl" -> abc_generation_sharded_dev over an in-library RCCL communicator, ONE GPU PER RANK (needs as many GPUs as ranks)),
out_json, [shape = "n_local,M,P,A,K,Kp,nnext_local"], [rule = "press" | "wilcoxon" (cabi only)], [data = "plain" | "ties":
the metric rows repeat 4 distinct ones, so the distances are massively tied | "blocked": metric 1 takes one value per shard |
"noisy": 1.5 sd of noise on every response], [split = "even" | "uneven" (cabi / rccl only): the
n_local * world rows dealt to the ranks in shares 2^(world-1-rank) -- two thirds / one third on two ranks, 4 : 2 : 1 on three]."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    """argv as in the module's docstring; an optional 7th argument holds FURTHER cases for the same ranks, a JSON list of
    [shape, rule, data, split] (round 6: one launch of the ranks for several cases -- starting them is 3-4 s of every test): the
    output file then holds {"cases": [result, ...]} in the order given, else the one result"""
    backend, out_path = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    first = [sys.argv[3] if len(sys.argv) > 3 else "1500,12,5,4,500,300,1000", sys.argv[4] if len(sys.argv) > 4 else "press",
             sys.argv[5] if len(sys.argv) > 5 else "plain", sys.argv[6] if len(sys.argv) > 6 else "even"]
    more = json.loads(sys.argv[7]) if len(sys.argv) > 7 else []
    results = [run_case(backend, *c) for c in [first] + more]
    if dist.get_rank() == 0:
        with open(out_path, "w") as f:
            json.dump(results[0] if not more else {"cases": results}, f)
    dist.barrier()
    dist.destroy_process_group()


def run_case(backend, shape_s, rule_s, data_s, split_s):
    rank, world = dist.get_rank(), dist.get_world_size()
    from abcsmc_amd import _lib, sharded, synthetic
    from oracle import pyoracle as O
    n_loc, M, P, A, K, Kp, nn_loc = (int(v) for v in shape_s.split(","))
    N = n_loc * world
    rule = _lib.RULE_WILCOXON if rule_s == "wilcoxon" else _lib.RULE_MIN_PRESS
    wl = synthetic.Workload(M, P, 777)
    ties = data_s == "ties"

    blocked = data_s == "blocked"
    noisy = data_s == "noisy"
    noise_all = np.random.default_rng(7).normal(size=(N, P)) if noisy else None
    ctx = None

    def rows(lo, hi):
        X, Y = wl.rows(lo, hi)
        if ties:
            X4, _ = wl.rows(0, 4)
            X = np.asfortranarray(X4[np.arange(lo, hi) % 4])
        if blocked:        # a metric that is constant WITHIN every (even) shard and differs between them: its Gram entries and sums are
            X[:, 1] = 3.0 + 2.0 * (np.arange(lo, hi) // n_loc)           # zero on every rank, its re-centring terms are all there is
        if noisy:          # responses the metrics hardly predict: the Wilcoxon rule takes components back that argmin PRESS keeps
            Y = np.asfortranarray(Y + 1.5 * wl.sd_y * noise_all[lo:hi])
        return X, Y

    uneven = split_s == "uneven"
    if uneven:
        assert backend in ("cabi", "rccl")
        shares = np.array([2.0 ** (world - 1 - r) for r in range(world)])
        cuts = np.concatenate([[0], np.floor(N * np.cumsum(shares) / shares.sum() + 0.5).astype(int)])
        cuts[-1] = N
        row_lo, row_hi = int(cuts[rank]), int(cuts[rank + 1])
    else:
        row_lo, row_hi = rank * n_loc, (rank + 1) * n_loc
    X, Y = rows(row_lo, row_hi)
    obs, spec = wl.observed(), wl.prior_spec()
    thp, wp, dvp = wl.previous_set(Kp) if Kp else (None, None, None)
    if backend == "numpy":
        from _numpy_backend import NumpyBackend
        be, dev = NumpyBackend(), "cpu"
        priors = O.make_priors(spec)
    else:
        dev = ("cuda:%d" % rank) if backend == "rccl" else "cuda:0"
        torch.cuda.set_device(torch.device(dev))
        be = sharded.HipBackend(dev) if backend == "hip" else None
        from abcsmc_amd import device
        priors = device.priors_to_device(_lib.make_priors(spec), dev)

    def cm(a):
        if a is None:
            return None
        a = np.asarray(a, dtype=np.float64)
        return torch.from_numpy(np.ascontiguousarray(a.T if a.ndim == 2 else a)).to(dev)

    if backend in ("cabi", "rccl"):
        ctx = _lib.Context(torch.device(dev).index)  # a context of its own: the communicator is attached to it
        ctx.set_stream(torch.cuda.current_stream(torch.device(dev)).cuda_stream)
        if backend == "rccl":
            sharded.attach_rccl(ctx, dev)
            assert ctx.comm_info() == (_lib.COMM_RCCL, world, rank)
        else:
            sharded.attach_torch_distributed(ctx, dev)
            assert ctx.comm_info() == (_lib.COMM_CALLBACKS, world, rank)
        gen = sharded.CabiShardedGeneration(ctx, dev, row_hi - row_lo, M, P, K, Kp, nn_loc, 0.5, A, rule=rule, multivariate=True,
                                            row0=row_lo, N_total=N, next0=rank * nn_loc, Nnext_total=nn_loc * world)
    else:
        gen = sharded.ShardedGeneration(be, n_loc, M, P, K, Kp, nn_loc, 0.5, A, rule=_lib.RULE_MIN_PRESS, multivariate=True)
    rng = _lib.Rng()
    _lib.lib().abc_rng_set(__import__("ctypes").byref(rng), 4242)
    gen.run(cm(X), cm(Y), cm(obs), priors, rng, cm(thp), cm(wp), cm(dvp))
    if dev != "cpu":
        torch.cuda.synchronize()
    # gather the sharded outputs on rank 0
    parents = [torch.zeros(nn_loc, dtype=torch.int64) for _ in range(world)]
    seeds = [torch.zeros(nn_loc, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(parents, gen.parent[:nn_loc].cpu())
    dist.all_gather(seeds, gen.seeds[:nn_loc].cpu())
    res = {"ok": True}
    if rank == 0:
        Xa, Ya = rows(0, N)
        o = O.rng(4242)
        ref = O.generation(Xa, Ya, obs, O.make_priors(spec), K, nn_loc * world, o, thp, wp, dvp, train_frac=0.5,
                           max_comp=A, rule=(O.RULE_WILCOXON if rule == _lib.RULE_WILCOXON else O.RULE_MIN_PRESS), multivariate=True)
        # declared deviation: seeds are the taus2 outputs right after the Nnext resampling draws
        # (the reference draws them after its data-dependent noise consumption)
        o2 = O.rng(4242)
        for _ in range(nn_loc * world):
            O.rng_get(o2)
        exp_seeds = np.array([O.rng_get(o2) for _ in range(nn_loc * world)], dtype=np.uint64)
        idx = gen.idx.cpu().numpy().astype(np.uint64)
        w = gen.w.cpu().numpy()
        par = torch.cat(parents).numpy().astype(np.uint64)
        sd = torch.cat(seeds).numpy().astype(np.uint64)
        res = {
            "ncomp": [int(gen.ncomp), int(ref["ncomp"])],
            "idx_equal": bool(np.array_equal(idx, ref["idx"])),
            # (a row outside a uniform prior's support has weight 0 in both: there the difference itself has to be 0)
            "w_maxrel": float(np.max(np.abs(w - ref["w"]) / np.where(ref["w"] > 0, ref["w"], 1.0))),
            "dv_maxrel": float(np.max(np.abs(gen.dv.cpu().numpy() - ref["dv"]) / ref["dv"])),
            "theta_equal": bool(np.array_equal(gen.theta.cpu().numpy().T, Ya[ref["idx"].astype(int)])),
            "parent_equal": bool(np.array_equal(par, ref["parent"])),
            "seeds_equal": bool(np.array_equal(sd, exp_seeds)),
            "rng_equal": [rng.s1, rng.s2, rng.s3] == [o2.s1, o2.s2, o2.s3],
            "next_finite": bool(torch.isfinite(gen.next).all().item()),
            "comm_calls": getattr(ctx, "comm_calls", None) if backend == "cabi" else None,
        }
    dist.barrier()
    del gen
    if ctx is not None:          # (this case's context: its arena and communicator go before the next case starts)
        try:
            ctx.close()
        except Exception:        # noqa: BLE001
            pass
    if dev != "cpu":
        torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
