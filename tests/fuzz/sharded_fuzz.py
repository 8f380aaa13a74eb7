"""abc_generation_sharded_dev (the C++ driver behind the C ABI, collectives through gloo callbacks) on two or three ranks sharing
cuda:0, at random shapes, against the single-process oracle: tests/_sharded_worker.py once per case, its checks
(tests/test_sharded.py::_check) applied here.  Shapes cover parameter counts on every weight kernel (fp64, one / two / four
chunks), first sets, the Wilcoxon rule, massively tied distances, local row counts below and above the gathered-sample
selection's threshold.
    python tests/fuzz/sharded_fuzz.py [out.json] [cases] [seed]"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/sharded_fuzz.json"
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 11
g = np.random.default_rng(seed0)
rows, fails = [], []
tmp = tempfile.mkdtemp()
for case in range(cases):
    world = 3 if case % 3 == 2 else 2
    P = int(g.choice([1, 3, 5, 12, 16, 17, 24, 32, 33, 40, 48])) if case % 2 else int(g.integers(1, 41))
    M = int(g.integers(max(3, P // 2), 66))
    big = case % 4 == 3
    n_loc = int(g.integers(4200, 9000)) if big else int(g.integers(500, 2500))
    cascade = case % 4 == 1        # (round 5) sets whose validation rows take the Wilcoxon rule's bounds cascade over the shards: >= 16384 of them
    if cascade:
        n_loc = int(g.integers(18000, 40000))
    N = n_loc * world
    K = int(g.integers(max(40, 2 * P + 8), N // 5))
    Kp = 0 if case % 5 == 4 else int(g.integers(max(40, 2 * P + 8), 900))
    nn_loc = int(g.integers(200, 3000))
    A = int(g.integers(1, min(M, 10) + 1))
    rule = "wilcoxon" if ((case % 6 == 1 or cascade) and P <= 40) else "press"
    split = "uneven" if case % 5 == 3 else "even"      # (round 5) shards of 2 : 1 (4 : 2 : 1 on three ranks)
    data = "ties" if case % 7 == 5 else "plain"
    if data == "ties":
        A = min(A, 3)        # (four distinct metric rows: beyond their rank the PRESS values are rounding noise and so is the count that minimises them)
    shape = "%d,%d,%d,%d,%d,%d,%d" % (n_loc, M, P, A, K, Kp, nn_loc)
    tag = dict(case=case, world=world, shape=shape, rule=rule, data=data, split=split)
    res_path = os.path.join(tmp, "c%d.json" % case)
    env = dict(os.environ)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29700 + case % 200), os.path.join(ROOT, "tests", "_sharded_worker.py"), "cabi", res_path, shape, rule, data, split]
    problems = []
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        if p.returncode != 0:
            problems.append("worker failed: " + (p.stdout[-600:] + p.stderr[-1200:]))
        else:
            r = json.load(open(res_path))
            tag.update({k: r[k] for k in ("ncomp", "w_maxrel", "dv_maxrel", "comm_calls") if k in r})
            if len(set(r["ncomp"])) != 1:
                problems.append("ncomp differs over the ranks")
            for k in ("idx_equal", "theta_equal", "parent_equal", "seeds_equal", "rng_equal", "next_finite"):
                if not r[k]:
                    problems.append(k + " is false")
            if not r["w_maxrel"] < 1e-6:
                problems.append("weights %.2e" % r["w_maxrel"])
            if not r["dv_maxrel"] < 1e-9:
                problems.append("dv %.2e" % r["dv_maxrel"])
    except subprocess.TimeoutExpired:
        problems.append("timeout")
    tag["problems"] = problems
    rows.append(tag)
    if problems:
        fails.append(tag)
    print(("FAIL " if problems else "ok   ") + json.dumps(tag)[:1500], flush=True)
json.dump({"cases": len(rows), "failed": len(fails), "failures": fails, "rows": rows}, open(out, "w"), indent=0)
print("%d cases, %d with problems" % (len(rows), len(fails)))
