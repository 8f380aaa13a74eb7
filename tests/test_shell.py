"""AbcSmc shell: JSON configuration, SQLite storage, set-0 sampling, the simulate step and
the job status machine of abcsmc_amd/cxx/AbcSmcHip.hpp, driven through examples/abc_dice.cpp.

CPU tests cover everything up to the first --process of a finished set (that step ranks on the GPU and must fail
loudly here); the GPU test runs a whole four-set fit and re-derives every stored posterior rank with the oracle.
Expected values are restated from the reference's sources: schema AbcSmc.cpp:819-834, set-0 order AbcUtil.cpp:515-521
and AbcSmc.cpp:843-871, odometer ParRNG.h:54-66, status machine AbcSmc.cpp:948-1037, exit codes as cited inline."""
import json
import math
import os
import sqlite3
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AbcLogBar = "=" * 89
LIBDIR = os.path.join(ROOT, "abcsmc_amd")


def _compile(src, exe):
    # ABC_TEST_CXXFLAGS: extra flags for the host programs, e.g. "-fsanitize=address,undefined -g" (with ASAN_OPTIONS=detect_leaks=0:
    # the HIP runtime the library links keeps its allocations) -- how the shell was run under the sanitizers on the CPU
    extra = os.environ.get("ABC_TEST_CXXFLAGS", "").split()
    subprocess.check_call(["g++", "-std=c++17", "-O1"] + extra + [src, "-o", exe, "-L" + LIBDIR, "-labcsmc_hip", "-ldl",
                           "-Wl,-rpath," + LIBDIR])
    return exe


@pytest.fixture(scope="module")
def dice(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "libabcsmc_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    d = tmp_path_factory.mktemp("shellbin")
    return _compile(os.path.join(ROOT, "examples", "abc_dice.cpp"), str(d / "abc_dice"))


@pytest.fixture(scope="module")
def probe(tmp_path_factory, dice):
    d = tmp_path_factory.mktemp("probebin")
    return _compile(os.path.join(ROOT, "tests", "cxx", "shell_probe.cpp"), str(d / "shell_probe"))


DICE = {
    "smc_iterations": 4,
    "num_samples": [300, 400],
    "predictive_prior_fraction": 0.25,
    "pls_training_fraction": 0.5,
    "noise": "MULTIVARIATE",
    "parameters": [
        {"name": "number of dice", "short_name": "ndice", "dist_type": "UNIFORM", "num_type": "INT", "par1": 1, "par2": 1000},
        {"name": "number of sides", "short_name": "sides", "dist_type": "UNIFORM", "num_type": "INT", "par1": 1, "par2": 1000},
    ],
    "metrics": [
        {"name": "sum", "num_type": "INT", "value": 44},
        {"name": "sd", "num_type": "FLOAT", "value": 2.39925},
    ],
}


def write_cfg(tmp_path, cfg, name="config.json", **over):
    c = json.loads(json.dumps(cfg))
    c.update(over)
    c.setdefault("database_filename", str(tmp_path / "abc.sqlite"))
    p = tmp_path / name
    # the reference's configuration files carry comments (jsoncpp accepts them)
    p.write_text("// fit configuration\n" + json.dumps(c, indent=2) + "\n/* end */\n")
    return str(p), c["database_filename"]


def run(exe, *args, check=True):
    r = subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stderr[-2000:]
    return r


def six(v):
    """the shell stores doubles through a default-formatted ostream: 6 significant digits"""
    return float("%.6g" % v)


def dice_metrics(ndice, sides, seed):
    r = orc.rng(seed)
    rolls = np.array([orc.rng_uniform_int(r, sides) + 1 for _ in range(ndice)], dtype=np.float64)
    sd = float(np.sqrt(((rolls - rolls.mean()) ** 2).sum() / (ndice - 1))) if ndice > 1 else 0.0
    return float(rolls.sum()), sd


def test_first_set_schema_and_stream(dice, tmp_path):
    cfg, db = write_cfg(tmp_path, DICE)
    run(dice, cfg, "--process", "--seed", "7")
    c = sqlite3.connect(db)
    sql = {r[0]: " ".join(r[1].split()) for r in c.execute("select name, sql from sqlite_master where sql is not null")}
    assert sql["job"] == ("CREATE TABLE job ( serial int primary key asc, smcSet int, particleIdx int, startTime int, "
                          "duration real, status text, posterior int, attempts int )")
    assert sql["idx1"] == "CREATE INDEX idx1 on job (status, attempts)"
    assert sql["par"] == "CREATE TABLE par ( serial int primary key, seed blob, ndice real, sides real )"
    assert sql["met"] == "CREATE TABLE met ( serial int primary key, sum real, sd real )"
    assert "upar" not in sql
    # set 0: for each particle every parameter in order, THEN one seed per particle
    r = orc.rng(7)
    want = [(orc.rng_uniform_int(r, 1000) + 1, orc.rng_uniform_int(r, 1000) + 1) for _ in range(300)]
    seeds = [orc.rng_get(r) for _ in range(300)]
    rows = c.execute("select serial, seed, ndice, sides from par order by serial").fetchall()
    assert [(int(a), int(b)) for _, _, a, b in rows] == want
    assert [int(s) for _, s, _, _ in rows] == seeds
    assert [r[0] for r in rows] == list(range(300))
    jobs = c.execute("select serial, smcSet, particleIdx, duration, status, posterior, attempts from job order by serial").fetchall()
    assert jobs == [(i, 0, i, None, "Q", -1, 0) for i in range(300)]
    assert c.execute("select count(*) from met where sum is null and sd is null").fetchone()[0] == 300
    # a second --process on an unfinished set must refuse (AbcSmc.cpp:579-582) and leave the database alone
    out = run(dice, cfg, "--process", "--seed", "8")
    assert "not all particles are complete in set 0" in out.stderr
    assert c.execute("select count(*) from job").fetchone()[0] == 300


def test_simulate_status_machine(dice, tmp_path):
    cfg, db = write_cfg(tmp_path, DICE)
    run(dice, cfg, "--process", "--seed", "3")
    c = sqlite3.connect(db)
    run(dice, cfg, "--simulate", "-n", "120")
    assert dict(c.execute("select status, count(*) from job group by status")) == {"D": 120, "Q": 180}
    assert c.execute("select max(serial) from job where status = 'D'").fetchone()[0] == 119      # queue order
    # a paused job is skipped by the queue but still takes results when asked for by serial (AbcSmc.cpp:966-972, 1006-1008)
    c.execute("update job set status = 'P' where serial = 200")
    c.commit()
    run(dice, cfg, "--simulate", "-n", "1000")
    assert dict(c.execute("select status, count(*) from job group by status")) == {"D": 299, "P": 1}
    for serial, seed, nd, sd_, s, m in c.execute("select P.serial, P.seed, ndice, sides, sum, sd from par P, met M where "
                                                  "P.serial = M.serial and P.serial != 200"):
        es, esd = dice_metrics(int(nd), int(sd_), int(seed))
        assert s == six(es) and m == pytest.approx(six(esd), rel=1e-12), serial
    assert c.execute("select min(attempts), max(attempts) from job where serial != 200").fetchone() == (1, 1)
    assert c.execute("select count(*) from job where duration is null").fetchone()[0] == 1
    # nothing left: another call is a no-op
    run(dice, cfg, "--simulate", "-n", "10")
    assert c.execute("select max(attempts) from job").fetchone()[0] == 1


def test_concurrent_workers_share_one_database(dice, tmp_path):
    """the reference's farm model: many `--simulate` workers on one file; BEGIN EXCLUSIVE + busy-retry
    (AbcSmc.cpp:886, sqdb.cpp:271-290) hand every queued particle to exactly one of them"""
    cfg, db = write_cfg(tmp_path, DICE)
    run(dice, cfg, "--process", "--seed", "3")
    procs = [subprocess.Popen([dice, cfg, "--simulate", "-n", "60"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
             for _ in range(4)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    c = sqlite3.connect(db)
    assert dict(c.execute("select status, count(*) from job group by status")) == {"D": 240, "Q": 60}
    assert c.execute("select min(attempts), max(attempts) from job where status = 'D'").fetchone() == (1, 1)
    assert c.execute("select count(*) from met where sum is not null").fetchone()[0] == 240


def test_process_needs_the_gpu(dice, tmp_path):
    """ranking a finished set is the HIP path: on a machine without the device the shell reports and exits non-zero,
    it never computes on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg, db = write_cfg(tmp_path, DICE)
    run(dice, cfg, "--process", "--seed", "3")
    run(dice, cfg, "--simulate", "-n", "300")
    r = run(dice, cfg, "--process", "--seed", "4", check=False)
    assert r.returncode == 3 and "HIP path failed" in r.stderr
    c = sqlite3.connect(db)
    assert c.execute("select count(*) from job where posterior > -1").fetchone()[0] == 0


def test_reference_shipped_configuration(dice, probe, tmp_path):
    """the CONTENT of the reference's own configuration -- examples/reference.json merged with examples/shared/partial.json, what
    `make run_shared` runs (examples/Makefile:38-39), committed as the data fixture tests/golden/reference_shared_config.json by
    tests/golden/make_reference_fixtures.py -- through the shell's parser: thirty sets growing 300 / 500 / 500 / 750 / 1000 / 1000 ...,
    half of each kept, two integer uniform priors on [1, 1000], the observed sum 44 and sd 2.39925 of 13 eight-sided dice
    (examples/README.md:29-34), MULTIVARIATE noise, the simulator a shared object with the `simulator` symbol (AbcSim.h:57-76).
    Then the first steps of the fit on it: set 0 sampled into `shared.sqlite`, particles simulated through libdice.so."""
    fixture = os.path.join(ROOT, "tests", "golden", "reference_shared_config.json")
    ref = json.load(open(fixture))
    assert ref["shared"] == "libdice.so" and ref["database_filename"] == "shared.sqlite" and ref["smc_iterations"] == 30
    # a libdice.so with the reference's plugin signature: sum and sample sd of `ndice` dice with `sides` sides (examples/include/dice.h:14-45
    # does the same with the example's own static RNG; here the particle's seed drives a taus2 stream so the test can re-derive it)
    src = tmp_path / "dice_plugin.cpp"
    src.write_text('#include <cmath>\n#include <vector>\n#include "%s"\n'
                   'extern "C" std::vector<double> simulator(std::vector<double> p, const unsigned long seed, const unsigned long) {\n'
                   '    ABC::RNG r(seed); const int n = (int)p[0], m = (int)p[1]; double s = 0; std::vector<double> x;\n'
                   '    for (int i = 0; i < n; i++) { x.push_back(1.0 + ABC::rng_uniform_int(&r, (unsigned long)m)); s += x.back(); }\n'
                   '    double q = 0; for (double v : x) q += (v - s / n) * (v - s / n);\n'
                   '    return {s, n > 1 ? std::sqrt(q / (n - 1)) : 0.0};\n}\n' % os.path.join(ROOT, "abcsmc_amd", "cxx", "AbcUtilHip.hpp"))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), str(src), "-o",
                           str(tmp_path / "libdice.so"), "-L" + LIBDIR, "-labcsmc_hip", "-Wl,-rpath," + LIBDIR])
    cfg = tmp_path / "shared.json"
    cfg.write_text(open(fixture).read())                           # (the file as committed: relative names resolve in the run directory)
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([probe, "--describe", str(cfg)], capture_output=True, text=True, timeout=60, cwd=str(tmp_path), env=env)
    assert out.returncode == 0, out.stderr[-1500:]
    got = {}
    for ln in out.stdout.splitlines():
        k, _, v = ln.partition(" ")
        got.setdefault(k, []).append(v)
    sizes = [300, 500, 500, 750] + [1000] * 26
    assert got["iterations"] == ["30"] and got["noise"] == ["MULTIVARIATE"] and got["filtering"] == ["PLS"]
    assert got["component_rule"] == ["wilcoxon"]                    # the facade's default
    assert [int(v) for v in got["set_sizes"][0].split()] == sizes
    assert [int(v) for v in got["pred_prior_sizes"][0].split()] == [round(0.5 * n) for n in sizes]
    assert got["parameter"] == ["ndice kind 1 a 1 b 1000 name number of dice", "sides kind 1 a 1 b 1000 name number of sides"]
    assert [m.split()[0] for m in got["metric"]] == ["sum", "sd"]
    assert [float(m.split()[2]) for m in got["metric"]] == [44.0, 2.39925]
    # set 0 into shared.sqlite, twenty particles through the configured shared object
    for args in (["--process", "--seed", "5", "--configured-simulator"], ["--simulate", "-n", "20", "--configured-simulator"]):
        r = subprocess.run([dice, str(cfg)] + args, capture_output=True, text=True, timeout=120, cwd=str(tmp_path), env=env)
        assert r.returncode == 0, r.stderr[-1500:]
    c = sqlite3.connect(str(tmp_path / "shared.sqlite"))
    assert c.execute("select count(*) from job where smcSet = 0").fetchone()[0] == 300
    rows = c.execute("select P.seed, ndice, sides, sum, sd from par P, met M where P.serial = M.serial and sum is not null").fetchall()
    assert len(rows) == 20
    for seed, nd, sd_, s_, d in rows:
        assert 1 <= nd <= 1000 and 1 <= sd_ <= 1000 and nd == int(nd) and sd_ == int(sd_)
        es, esd = dice_metrics(int(nd), int(sd_), int(seed))
        assert s_ == six(es) and d == pytest.approx(six(esd), rel=1e-12)


def test_iteration_rules(probe, tmp_path):
    def sizes(**over):
        base = {k: v for k, v in DICE.items() if k not in ("predictive_prior_fraction", "smc_iterations", "num_samples")}
        cfg, _ = write_cfg(tmp_path, base, **over)
        r = run(probe, cfg, "x", check=False)
        out = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines()) if r.returncode == 0 else {}
        return r, out

    # fractions: both lists extended by their last entry, sizes rounded (AbcSmc.cpp:100-111); iterations default to
    # the longer list (:128)
    r, o = sizes(num_samples=[300, 500, 750], predictive_prior_fraction=[0.5, 0.1])
    assert o["iterations"] == "3" and o["set_sizes"] == "300 500 750" and o["pred_prior_sizes"] == "150 50 75"
    r, o = sizes(num_samples=101, predictive_prior_fraction=0.5, smc_iterations=3)
    assert o["set_sizes"] == "101 101 101" and o["pred_prior_sizes"] == "51 51 51"     # round half away from zero
    r, o = sizes(num_samples=[200, 100], predictive_prior_size=[50, 40, 30], smc_iterations=5)
    assert o["pred_prior_sizes"] == "50 40 30 30 30" and o["set_sizes"] == "200 100 100 100 100"
    # errors: both / neither size option, predictive prior larger than its set, fraction outside (0,1], training
    # fraction outside (0,1): exit(1) (AbcSmc.cpp:84-98, 112-126)
    for over in (dict(num_samples=100), dict(num_samples=100, predictive_prior_fraction=0.5, predictive_prior_size=10),
                 dict(num_samples=[100, 20], predictive_prior_size=[10, 30]),
                 dict(num_samples=100, predictive_prior_fraction=1.5),
                 dict(num_samples=100, predictive_prior_fraction=0.5, pls_training_fraction=1.0)):
        r, _ = sizes(**over)
        assert r.returncode == 1, over
    r, _ = sizes(num_samples=100, predictive_prior_fraction=0.5, noise="CORRELATED")
    assert r.returncode == (-210) % 256                                               # AbcSmc.cpp:424-427
    bad = json.loads(json.dumps(DICE))
    bad["parameters"][0]["dist_type"] = "CAUCHY"
    cfg, _ = write_cfg(tmp_path, bad, name="bad.json")
    assert run(probe, cfg, check=False).returncode == (-205) % 256                    # :270-272
    bad = json.loads(json.dumps(DICE))
    bad["metrics"][0]["num_type"] = "COMPLEX"
    cfg, _ = write_cfg(tmp_path, bad, name="bad2.json")
    assert run(probe, cfg, check=False).returncode == (-209) % 256                    # :147-150
    (tmp_path / "broken.json").write_text('{"parameters": [ {"name": ] }')
    assert run(probe, str(tmp_path / "broken.json"), check=False).returncode == 1     # :284-288
    assert run(probe, str(tmp_path / "missing.json"), check=False).returncode == 1    # :277-280


def test_pseudo_grid_and_transforms(dice, probe, tmp_path):
    """projection mode: PSEUDO parameters enumerate their grid with the FIRST parameter fastest (ParRNG.h:54-66,
    tests/ParRNG.test.cpp), one set, no predictive prior; `untransform` creates the upar table the simulator reads"""
    cfg = {
        "parameters": [
            {"name": "number of dice", "short_name": "ndice", "dist_type": "PSEUDO", "num_type": "INT", "par1": 1, "par2": 5},
            {"name": "number of sides", "short_name": "sides", "dist_type": "PSEUDO", "num_type": "INT", "vals": [2, 4, 6, 8, 10]},
            {"name": "unused", "dist_type": "PSEUDO", "num_type": "FLOAT", "par1": 0.5, "par2": 1.0, "step": 0.25},
        ],
        "metrics": DICE["metrics"],
    }
    path, db = write_cfg(tmp_path, cfg)
    out = run(probe, path).stdout
    assert "iterations 1" in out and "set_sizes 75" in out
    run(dice, path, "--process", "--seed", "5")
    c = sqlite3.connect(db)
    rows = c.execute("select ndice, sides, unused from par order by serial").fetchall()
    want = [(float(a), float(b), u) for u in (0.5, 0.75, 1.0) for b in (2, 4, 6, 8, 10) for a in (1, 2, 3, 4, 5)]
    assert rows == want
    r = run(probe, write_cfg(tmp_path, cfg, name="c2.json", num_samples=74)[0], check=False)
    assert r.returncode == (-201) % 256                                               # AbcSmc.cpp:69-73
    r = run(probe, write_cfg(tmp_path, cfg, name="c3.json", smc_iterations=2)[0], check=False)
    assert r.returncode == (-202) % 256                                               # :63-66

    # transforms (AbcSmc.cpp:154-210, ParXform.h:25-37): POW_10 as a string; a LOGISTIC object with bounds and an
    # untransformed factor
    tcfg = json.loads(json.dumps(DICE))
    tcfg["parameters"] = [
        {"name": "log dice", "short_name": "ldice", "dist_type": "UNIFORM", "num_type": "FLOAT", "par1": 0, "par2": 2, "untransform": "POW_10"},
        {"name": "sides", "dist_type": "UNIFORM", "num_type": "INT", "par1": 2, "par2": 12},
        {"name": "frac", "dist_type": "NORMAL", "num_type": "FLOAT", "par1": 0, "par2": 1,
         "untransform": {"type": "LOGISTIC", "min": 1, "max": 3, "untransformed_factor": ["sides"]}},
    ]
    path, db = write_cfg(tmp_path, tcfg, name="t.json", database_filename=str(tmp_path / "t.sqlite"))
    run(dice, path, "--process", "--seed", "9")
    c = sqlite3.connect(db)
    assert c.execute("select count(*) from upar").fetchone()[0] == 300
    r = orc.rng(9)
    for (s1, ld, sides, fr), (s2, u0, u1, u2) in zip(c.execute("select serial, ldice, sides, frac from par order by serial"),
                                                   c.execute("select serial, ldice, sides, frac from upar order by serial")):
        e0 = orc.rng_uniform(r) * 2.0
        e1 = orc.rng_uniform_int(r, 11) + 2
        e2 = orc.ran_gaussian(r, 1.0) + 0.0
        assert (ld, sides, fr) == (six(e0), float(e1), six(e2)) and s1 == s2
        assert u0 == six(10.0 ** e0) and u1 == float(e1)
        assert u2 == six((3 - 1) * ((1.0 / (1.0 + math.exp(-e2))) * e1) + 1)


def test_configured_simulators(dice, tmp_path):
    """`shared` (dlopen + the `simulator` symbol, AbcSim.h:62-76, 106-117) and `executable` (parameters as arguments,
    metrics on stdout, AbcSim.h:122-157); an unset simulator exits 100 (AbcSim.h:45-52), a missing object 101, a
    missing symbol 102, a wrong metric count -211 (AbcSmc.cpp:998)"""
    src = tmp_path / "sim.cpp"
    src.write_text('#include <vector>\nextern "C" std::vector<double> simulator(std::vector<double> p, const unsigned long seed, '
                   'const unsigned long serial) { return {p[0] + p[1], (double)(seed % 1000) + 0.5}; }\n'
                   'extern "C" int unrelated() { return 0; }\n')
    so = str(tmp_path / "libsim.so")
    subprocess.check_call(["g++", "-std=c++17", "-shared", "-fPIC", str(src), "-o", so])
    cfg, db = write_cfg(tmp_path, DICE, name="so.json", shared=so, database_filename=str(tmp_path / "so.sqlite"))
    run(dice, cfg, "--process", "--seed", "2", "--configured-simulator")
    run(dice, cfg, "--simulate", "-n", "50", "--configured-simulator")
    c = sqlite3.connect(db)
    rows = c.execute("select seed, ndice, sides, sum, sd from par P, met M where P.serial = M.serial and sum is not null").fetchall()
    assert len(rows) == 50
    for seed, a, b, s_, d in rows:
        assert s_ == a + b and d == int(seed) % 1000 + 0.5

    script = tmp_path / "sim.sh"
    script.write_text("#!/bin/sh\necho $(( $1 * 2 )) $2.25\n")
    script.chmod(0o755)
    cfg, db = write_cfg(tmp_path, DICE, name="ex.json", executable=str(script), database_filename=str(tmp_path / "ex.sqlite"))
    run(dice, cfg, "--process", "--seed", "2", "--configured-simulator")
    run(dice, cfg, "--simulate", "-n", "20", "--configured-simulator")
    c = sqlite3.connect(db)
    rows = c.execute("select ndice, sides, sum, sd from par P, met M where P.serial = M.serial and sum is not null").fetchall()
    assert len(rows) == 20
    for a, b, s_, d in rows:
        assert s_ == 2 * a and d == b + 0.25

    cfg, db = write_cfg(tmp_path, DICE, name="unset.json", database_filename=str(tmp_path / "unset.sqlite"))
    run(dice, cfg, "--process", "--seed", "2", "--configured-simulator")
    assert run(dice, cfg, "--simulate", "--configured-simulator", check=False).returncode == 100
    cfg, _ = write_cfg(tmp_path, DICE, name="noso.json", shared=str(tmp_path / "nope.so"))
    assert run(dice, cfg, "--process", "--configured-simulator", check=False).returncode == 101
    src.write_text('extern "C" int unrelated() { return 0; }\n')
    subprocess.check_call(["g++", "-shared", "-fPIC", str(src), "-o", str(tmp_path / "libnosym.so")])
    cfg, _ = write_cfg(tmp_path, DICE, name="nosym.json", shared=str(tmp_path / "libnosym.so"))
    assert run(dice, cfg, "--process", "--configured-simulator", check=False).returncode == 102
    script.write_text("#!/bin/sh\necho 1 2 3\n")
    cfg, db = write_cfg(tmp_path, DICE, name="ex3.json", executable=str(script), database_filename=str(tmp_path / "ex3.sqlite"))
    run(dice, cfg, "--process", "--seed", "2", "--configured-simulator")
    assert run(dice, cfg, "--simulate", "--configured-simulator", check=False).returncode == (-211) % 256


def test_posterior_projection(dice, tmp_path):
    """POSTERIOR parameters replay the posterior rows of an earlier fit (AbcSmc.cpp:293-335, AbcUtil.cpp:515-523):
    the PSEUDO odometer turns fastest, the posterior cursor advances when it wraps; retain_posterior_rank copies
    the row index into job.posterior (AbcSmc.cpp:848-853)"""
    src = str(tmp_path / "earlier.sqlite")
    c = sqlite3.connect(src)
    c.execute("create table job ( serial int primary key asc, smcSet int, particleIdx int, startTime int, duration real, status text, posterior int, attempts int )")
    c.execute("create table par ( serial int primary key, seed blob, ndice real, sides real )")
    post = {}
    for serial in range(12):
        rank = {2: 1, 5: 0, 7: 3, 11: 2}.get(serial, -1)
        c.execute("insert into job values (?, 0, ?, 0, NULL, 'D', ?, 1)", (serial, serial, rank))
        c.execute("insert into par values (?, '1', ?, ?)", (serial, 10.0 + serial, 100.0 + serial))
        if rank > -1:
            post[serial] = (10.0 + serial, 100.0 + serial)
    c.commit()
    cfg = {
        "posterior_database_filename": src,
        "retain_posterior_rank": True,
        "parameters": [
            {"name": "ndice", "dist_type": "POSTERIOR", "num_type": "INT", "par1": 0, "par2": 3},
            {"name": "scale", "dist_type": "PSEUDO", "num_type": "FLOAT", "vals": [0.5, 2.0]},
            {"name": "sides", "dist_type": "POSTERIOR", "num_type": "INT", "par1": 0, "par2": 3},
        ],
        "metrics": DICE["metrics"],
    }
    path, db = write_cfg(tmp_path, cfg)
    run(dice, path, "--process", "--seed", "1")
    d = sqlite3.connect(db)
    rows = d.execute("select J.posterior, ndice, scale, sides from job J, par P where J.serial = P.serial order by J.serial").fetchall()
    order = [post[s] for s in sorted(post)]                 # rows come in storage order, not rank order
    want = [(i, order[i][0], sc, order[i][1]) for i in range(4) for sc in (0.5, 2.0)]
    assert rows == want
    cfg.pop("posterior_database_filename")
    path, _ = write_cfg(tmp_path, cfg, name="nopost.json")
    assert run(dice, path, "--process", check=False).returncode == (-204) % 256      # AbcSmc.cpp:382-385


def _taus2_state_with_zero_output():
    """a valid taus2 state whose NEXT output is 0: the three component steps are GF(2)-linear, so solve
    step1(s1) = step2(s2) ^ step3(s3) for s1 by elimination"""
    def step(s, a, b, c, d):
        return (((s & c) << d) & 0xffffffff) ^ ((((s << a) & 0xffffffff) ^ s) >> b)
    s1f = lambda s: step(s, 13, 19, 4294967294, 12)
    s2f = lambda s: step(s, 2, 25, 4294967288, 4)
    s3f = lambda s: step(s, 3, 11, 4294967280, 17)
    basis = [(s1f(1 << i), 1 << i) for i in range(1, 32)]           # bit 0 of s1 does not reach the output
    for s3 in range(1000, 1200):
        s2 = 0x9e3779b9
        target, comb = s2f(s2) ^ s3f(s3), 0
        rows = list(basis)
        for bit in range(31, -1, -1):                                # Gaussian elimination, one pivot per output bit
            piv = next((r for r in rows if (r[0] >> bit) & 1), None)
            if piv is None:
                continue
            rows = [(r[0] ^ piv[0], r[1] ^ piv[1]) if ((r[0] >> bit) & 1 and r is not piv) else r for r in rows if r is not piv]
            if (target >> bit) & 1:
                target ^= piv[0]
                comb ^= piv[1]
        if target == 0 and comb >= 2:
            assert s1f(comb) ^ s2f(s2) ^ s3f(s3) == 0
            return comb, s2, s3
    raise AssertionError("no state found")


def test_host_gaussian_redraws_a_zero_uniform(probe):
    """gsl_ran_gaussian draws with gsl_rng_uniform_pos (gauss.c): a taus2 output of exactly 0 is drawn again.  The facade's
    host ran_gaussian (set-0 sampling of Gaussian priors) must consume the stream the same way: forced here with a state
    whose next output is 0, against the oracle's restatement"""
    s1, s2, s3 = _taus2_state_with_zero_output()
    r = orc.rng(1)
    r.s1, r.s2, r.s3 = s1, s2, s3
    probe_first = orc.rng(1)
    probe_first.s1, probe_first.s2, probe_first.s3 = s1, s2, s3
    assert orc.rng_get(probe_first) == 0
    want = [orc.ran_gaussian(r, 2.5) for _ in range(5)]
    out = run(probe, "--gauss", str(s1), str(s2), str(s3), "2.5", "5").stdout.split("\n")
    got = [float.fromhex(x) for x in out[:5]]
    assert got == want
    assert out[5] == "state %d %d %d" % (r.s1, r.s2, r.s3)


def _nrmse(mets, obs):
    """AbcUtil.cpp:326-345"""
    sim = mets.mean(axis=0)
    exp = (np.abs(obs) + np.abs(sim)) / 2.0
    exp[sim == obs] = 1.0
    return float(np.sqrt((((sim - obs) / exp) ** 2).mean()))


def _median(col):
    """AbcUtil.cpp:46-62"""
    v = np.sort(col)
    n = len(v)
    return float((v[n // 2 - 1] + v[n // 2]) / 2 if n % 2 == 0 else v[n // 2])


def _table_rows(text, title, nrows, ncols_left):
    """the rows printed under `title` + header line of a filtering report -> list of (pars, mets) float lists"""
    lines = text.split("\n")
    i = lines.index(title)
    out = []
    for ln in lines[i + 2:i + 2 + nrows]:
        left, right = ln.split(" | ")
        out.append(([float(x) for x in left.split()], [float(x) for x in right.split()]))
        assert len(out[-1][0]) == ncols_left
    return out


def _check_filter_report(text, t, ppars, pmets, obs, prec):
    """every number of AbcLog::filtering_report (AbcLog.cpp:79-123) against numpy on the same posterior rows"""
    tol = dict(rel=2 * 10.0 ** (1 - prec), abs=1e-12)
    lines = text.split("\n")
    assert "Set %d" % t in lines and "Observed:" in lines
    obs_row = lines[lines.index("Observed:") + 2].split(" | ")
    assert obs_row[0].split() == ["---"] * ppars.shape[1]
    assert [float(x) for x in obs_row[1].split()] == pytest.approx(list(obs), **tol)
    nr = [l for l in lines if l.startswith("Normalized RMSE for metric means (lower is better):")]
    assert len(nr) >= 1 and float(nr[0].split(":")[1]) == pytest.approx(_nrmse(pmets, np.asarray(obs, dtype=float)), **tol)
    (mp, mm), = _table_rows(text, "Posterior means:", 1, ppars.shape[1])
    assert mp == pytest.approx(list(ppars.mean(axis=0)), **tol) and mm == pytest.approx(list(pmets.mean(axis=0)), **tol)
    (dp, dm), = _table_rows(text, "Posterior medians:", 1, ppars.shape[1])
    assert dp == pytest.approx([_median(c) for c in ppars.T], **tol) and dm == pytest.approx([_median(c) for c in pmets.T], **tol)
    best = _table_rows(text, "Best five:", 5, ppars.shape[1])
    worst = _table_rows(text, "Worst five:", 5, ppars.shape[1])
    for q in range(5):
        assert best[q][0] == pytest.approx(list(ppars[q]), **tol) and best[q][1] == pytest.approx(list(pmets[q]), **tol)
        k = len(ppars) - 5 + q
        assert worst[q][0] == pytest.approx(list(ppars[k]), **tol) and worst[q][1] == pytest.approx(list(pmets[k]), **tol)


def test_filtering_report_numbers(probe, tmp_path):
    """AbcLog::filtering_report on hand-made posterior rows (no GPU): observed row, NRMSE, means, medians, best and worst five"""
    cfg, _ = write_cfg(tmp_path, DICE)
    K = 12
    text = run(probe, "--filter-report", cfg, str(K)).stdout
    i = np.arange(K)[:, None]
    ppars = np.array([[(ii * 37 + 11 * j) % 101 + 0.25 * j for j in range(2)] for ii in range(K)], dtype=float)
    pmets = np.array([[40.0 + ((ii * 13 + 7 * j) % 17) * (0.125 if j else 1.0) - 30.0 * j for j in range(2)] for ii in range(K)], dtype=float)
    header = text.split("\n")[text.split("\n").index("Observed:") + 1]
    assert header.split() == ["ndice", "sides", "|", "sum", "sd"]
    _check_filter_report(text, 3, ppars, pmets, [44.0, 2.39925], prec=6)


# ---- whole fit on the GPU -----------------------------------------------------------------------------------
@pytest.mark.gpu
def test_dice_fit_end_to_end(dice, tmp_path):
    cfg, db = write_cfg(tmp_path, DICE)
    r = run(dice, cfg, "--process", "--simulate", "--all", "--seed", "11")
    assert "Database already contains 4 complete sets." in r.stderr      # the closing --process of the loop
    c = sqlite3.connect(db)
    sizes = [300, 400, 400, 400]
    K = [75, 100, 100, 100]
    assert c.execute("select smcSet, count(*) from job group by smcSet order by smcSet").fetchall() == list(enumerate(sizes))
    assert c.execute("select count(*) from job where status != 'D'").fetchone()[0] == 0
    err, posts = [], []
    for t, n in enumerate(sizes):
        rows = c.execute("select J.particleIdx, J.posterior, P.seed, ndice, sides, sum, sd from job J, par P, met M where J.serial = P.serial "
                         "and J.serial = M.serial and smcSet = ? order by particleIdx", (t,)).fetchall()
        pars = np.array([[r[3], r[4]] for r in rows])
        mets = np.array([[r[5], r[6]] for r in rows])
        # proposals are valid under the integer priors (recast + rejection on the device)
        assert np.all(pars == np.round(pars)) and pars.min() >= 1 and pars.max() <= 1000
        # the stored ranks are the oracle's ranking of exactly what is in the database (bit-exact indices)
        # (the shell's default component rule is the facade's: upstream's Wilcoxon reduction)
        want = orc.particle_ranking_pls(mets, pars, np.array([44.0, 2.39925]), 0.5, rule=orc.RULE_WILCOXON)["idx"][:K[t]]
        got = sorted((r[1], r[0]) for r in rows if r[1] > -1)
        assert [g[0] for g in got] == list(range(K[t]))
        assert [g[1] for g in got] == [int(i) for i in want], "set %d" % t
        if t > 0:
            # simulator seeds: taus2(seed + t) outputs n .. 2n-1, after the n resampling draws
            g = orc.rng(11 + t)
            for _ in range(n):
                orc.rng_get(g)
            assert [int(r[2]) for r in rows] == [orc.rng_get(g) for _ in range(n)]
        post = mets[[g[1] for g in got]]
        err.append(float(np.abs(post[:, 0] - 44).mean()))
        posts.append((pars[[g[1] for g in got]], post))
    assert err[-1] < 0.2 * err[0], err            # the posterior closes in on the observed sum
    # ---- the stderr reports of the run (AbcLog.cpp:24-123), every number re-derived from the database contents ----------
    # one filtering report per set, printed by the --process that ranked it (reports print with setprecision(5), AbcSmc.cpp:465)
    chunks = r.stderr.split(AbcLogBar + "\nSet ")
    assert len(chunks) == 1 + len(sizes)
    for t in range(len(sizes)):
        text = "Set " + chunks[t + 1]
        _check_filter_report(text, t, posts[t][0], posts[t][1], [44.0, 2.39925], prec=5)
    # convergence data: the last block of the closing --process covers the final set against the one before it
    conv = r.stderr.split("Convergence data for predictive priors:\n")[-1]
    tol = dict(rel=2e-4, abs=1e-12)
    prior_mean, prior_sd = (1000 + 1) / 2.0, (1000 - 1) / math.sqrt(12.0)            # Priors.h:64-67
    for j, name in enumerate(["number of dice", "number of sides"]):
        block = conv.split('Par %d: "%s"\n' % (j, name))[1].split("  Par ")[0]
        nums = [[float(x.strip().rstrip("%")) for x in ln.split("):")[1].replace("(", ",").replace(")", "").split(",")]
                for ln in block.split("\n") if "( delta, % )" in ln]
        cur, last = posts[-1][0][:, j], posts[-2][0][:, j]
        cm, lm, cs, ls = cur.mean(), last.mean(), cur.std(ddof=1), last.std(ddof=1)
        assert nums[0] == pytest.approx([prior_mean, cm, cm - prior_mean, 100 * (cm - prior_mean) / prior_mean], **tol)   # Prior, current means
        assert nums[1] == pytest.approx([lm, cm, cm - lm, 100 * (cm - lm) / lm], rel=2e-4, abs=2e-3)                       # Last, current means
        assert nums[2] == pytest.approx([prior_sd, cs, cs - prior_sd, 100 * (cs - prior_sd) / prior_sd], **tol)            # standard deviations
        assert nums[3] == pytest.approx([ls, cs, cs - ls, 100 * (cs - ls) / ls], rel=2e-4, abs=2e-3)
    # a finished fit: --process reports and changes nothing
    r = run(dice, cfg, "--process", "--seed", "1")
    assert "Database already contains 4 complete sets." in r.stderr
    assert c.execute("select count(*) from job").fetchone()[0] == sum(sizes)


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["flag", "json"])
def test_dice_fit_component_rule_switch(dice, tmp_path, how):
    """ABC::set_component_rule through the shell: `--component-rule press` / "pls_component_rule": "min_press" rank every set by
    the plain PRESS argmin (the facade's default is the Wilcoxon reduction); an unknown rule is a configuration error"""
    over = {"pls_component_rule": "min_press"} if how == "json" else {}
    cfg, db = write_cfg(tmp_path, DICE, **over)
    extra = ["--component-rule", "press"] if how == "flag" else []
    run(dice, cfg, "--process", "--simulate", "--all", "--seed", "17", *extra)
    c = sqlite3.connect(db)
    sizes, K = [300, 400, 400, 400], [75, 100, 100, 100]
    for t, n in enumerate(sizes):
        rows = c.execute("select J.particleIdx, J.posterior, ndice, sides, sum, sd from job J, par P, met M where J.serial = P.serial "
                         "and J.serial = M.serial and smcSet = ? order by particleIdx", (t,)).fetchall()
        pars = np.array([[r[2], r[3]] for r in rows])
        mets = np.array([[r[4], r[5]] for r in rows])
        want = orc.particle_ranking_pls(mets, pars, np.array([44.0, 2.39925]), 0.5, rule=orc.RULE_MIN_PRESS)["idx"][:K[t]]
        got = sorted((r[1], r[0]) for r in rows if r[1] > -1)
        assert [g[1] for g in got] == [int(i) for i in want], "set %d" % t
    if how == "json":
        bad, _ = write_cfg(tmp_path, DICE, name="bad.json", pls_component_rule="median")
        r = run(dice, bad, "--process", check=False)
        assert r.returncode != 0 and "pls_component_rule" in r.stderr


@pytest.mark.gpu
def test_dice_fit_through_the_multi_device_path(dice, tmp_path):
    """--devices 0: every set is ranked and weighted by abc_generation_multi over the contexts of abc_ctx_create_multi (RCCL
    communicator, one rank on this box); the database must equal the one the single-context path writes for the same seeds"""
    dbs = []
    for tag, extra in (("single", []), ("multi", ["--devices", "0"])):
        d = tmp_path / tag
        d.mkdir()
        cfg, db = write_cfg(d, DICE)
        run(dice, cfg, "--process", "--simulate", "--all", "--seed", "21", *extra)
        c = sqlite3.connect(db)
        dbs.append(c.execute("select J.serial, smcSet, particleIdx, posterior, P.seed, ndice, sides, sum, sd from job J, par P, met M "
                             "where J.serial = P.serial and J.serial = M.serial order by J.serial").fetchall())
    assert len(dbs[0]) == 1500 and dbs[0] == dbs[1]


@pytest.mark.gpu
def test_dice_fit_reference_stream_reproduces_an_oracle_driven_fit(dice, tmp_path):
    """--reference-stream: proposals and simulator seeds of every set come from the shared taus2 stream consumed as the
    reference consumes it.  The whole database -- parameters and seeds of all four sets, posterior ranks -- is re-derived
    here with the oracle alone (set-0 sampling, ranking, weights, covariance factor, resampling, sequential Gaussian noise
    with rejection, seeds) from the same per-step seeds, and must be identical."""
    cfg, db = write_cfg(tmp_path, DICE)
    run(dice, cfg, "--process", "--simulate", "--all", "--seed", "33", "--reference-stream")
    c = sqlite3.connect(db)
    sizes, K = [300, 400, 400, 400], [75, 100, 100, 100]
    pri = orc.make_priors([(1, 1, 1000), (1, 1, 1000)])
    obs = np.array([44.0, 2.39925])
    # set 0 from the oracle's stream: all parameter draws, then all seeds (AbcUtil.cpp:515-521, AbcSmc.cpp:843-871)
    r = orc.rng(33)
    pars = np.array([[orc.rng_uniform_int(r, 1000) + 1 for _ in range(2)] for _ in range(sizes[0])], dtype=float)
    seeds = np.array([orc.rng_get(r) for _ in range(sizes[0])], dtype=np.uint64)
    prev = None
    for t, n in enumerate(sizes):
        rows = c.execute("select J.particleIdx, J.posterior, P.seed, ndice, sides, sum, sd from job J, par P, met M where J.serial = P.serial "
                         "and J.serial = M.serial and smcSet = ? order by particleIdx", (t,)).fetchall()
        assert np.array_equal(np.array([[q[3], q[4]] for q in rows]), pars), "parameters of set %d" % t
        assert [int(q[2]) for q in rows] == [int(x) for x in seeds], "seeds of set %d" % t
        mets = np.array([[six(v) for v in dice_metrics(int(a), int(b), int(sd))] for (a, b), sd in zip(pars, seeds)])
        assert np.allclose(mets, np.array([[q[5], q[6]] for q in rows]), rtol=1e-5)
        mets = np.array([[q[5], q[6]] for q in rows])                       # what the next --process reads back
        idx = orc.particle_ranking_pls(mets, pars, obs, 0.5, rule=orc.RULE_WILCOXON)["idx"][:K[t]].astype(int)
        assert [q[0] for q in sorted((q for q in rows if q[1] > -1), key=lambda q: q[1])] == [int(i) for i in idx]
        theta = np.asfortranarray(pars[idx])
        dv = orc.doubled_variance(theta)
        w = orc.weights_uniform(len(idx)) if prev is None else orc.weights_importance(pri, theta, prev[0], prev[1], prev[2])
        prev = (theta, w, dv)
        if t + 1 < len(sizes):
            rc, L, _ = orc.mvn_setup(theta)
            assert rc == 0
            r = orc.rng(33 + t + 1)                                         # the step-th --process of AbcSmc::run reseeds
            pars, _, _ = orc.sample_mvn_predictive_priors(r, sizes[t + 1], w, theta, pri, L)
            seeds = np.array([orc.rng_get(r) for _ in range(sizes[t + 1])], dtype=np.uint64)
