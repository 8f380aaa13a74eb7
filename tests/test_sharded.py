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
SHAPES = {"small": "1500,12,5,4,500,300,1000", "config4": "1500,64,32,8,400,300,1200", "config5": "1400,128,16,32,300,250,1000"}


def _launch(backend, tmp_path, port, shape="small"):
    out = str(tmp_path / ("sharded_%s_%s.json" % (backend, shape)))
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_sharded_worker.py"), backend, out, SHAPES[shape]]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return json.load(open(out))


def _check(res):
    assert res["ncomp"][0] == res["ncomp"][1]
    assert res["idx_equal"] and res["theta_equal"]
    assert res["w_maxrel"] < 1e-6 and res["dv_maxrel"] < 1e-9
    assert res["parent_equal"] and res["seeds_equal"] and res["rng_equal"] and res["next_finite"]


@pytest.mark.parametrize("shape,port", [("small", 29611), ("config4", 29613)])
def test_sharded_world2_gloo_cpu(tmp_path, shape, port):
    _check(_launch("numpy", tmp_path, port, shape))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,port", [("small", 29612), ("config4", 29614), ("config5", 29615)])
def test_sharded_world2_gloo_hip(tmp_path, shape, port):
    _check(_launch("hip", tmp_path, port, shape))


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
    g1 = device.Generation(N, M, P, K, Kp, Nn, 0.5, A, device=dev, ctx=gpu_ctx)
    r1 = abcutil.rng(5)
    g1.run(*args, pri, r1, *prev)
    g2 = sharded.ShardedGeneration(sharded.HipBackend(dev, gpu_ctx), N, M, P, K, Kp, Nn, 0.5, A)
    r2 = abcutil.rng(5)
    g2.run(*args, pri, r2, *prev)
    torch.cuda.synchronize()
    for a, b in ((g1.idx, g2.idx), (g1.w, g2.w), (g1.dv, g2.dv), (g1.parent, g2.parent), (g1.seeds, g2.seeds),
                 (g1.next, g2.next), (g1.theta, g2.theta)):
        assert torch.equal(a, b)
    assert (r1.s1, r1.s2, r1.s3) == (r2.s1, r2.s2, r2.s3)
