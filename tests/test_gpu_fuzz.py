"""Randomised differential tests: the fuzzers under tests/fuzz/ (each a stand-alone program that replays random shapes and unfriendly
data against the CPU oracle and prints one verdict line) at a small number of cases with fixed seeds.  Their long runs are
summarised under profiles/history/r03_*_fuzz.json; the far-row bug of round 3 (weights.hip, k_wrows: a far coordinate beyond the first
eight parameters) was found by the first of them."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("script,cases,seed", [
    ("kde_accuracy_sweep.py", 1, None),          # every parameter count 5..64, one set each, both orders of the previous tiles
    # (round 6: half the cases of rounds 3-5 -- the GPU suite had grown to 505 s of its 900; the long runs are the ones under profiles/)
    ("weights_fuzz.py", 75, 21),
    ("generation_fuzz.py", 30, 22),
    ("ranking_fuzz.py", 30, 23),
    ("resample_fuzz.py", 40, 24),
    ("sharded_fuzz.py", 3, 25),
    ("wilcoxon_fuzz.py", 6, 26),
    ("wide_gram_fuzz.py", 2, 27),
    ("wide_model_fuzz.py", 1, 28),               # round 6: the byte-limb statistics kernel held at the LOADINGS (1e-6 of every used column)
])
def test_fuzzer_finds_nothing(tmp_path, script, cases, seed):
    out = str(tmp_path / (script + ".json"))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "fuzz", script), out, str(cases)] + ([str(seed)] if seed is not None else [])
    env = dict(os.environ, FUZZ_BIG="1") if script == "wide_model_fuzz.py" else None     # (sets the default sends to the byte-limb kernel)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    tail = (p.stdout[-3000:] + p.stderr[-2000:])
    assert p.returncode == 0, tail
    last = p.stdout.strip().splitlines()[-1]
    if script == "kde_accuracy_sweep.py":
        assert last.endswith("over the asserted bound: none"), tail
    else:
        assert " 0 with problems" in last, "\n".join(l for l in p.stdout.splitlines() if l.startswith("FAIL"))[:4000] + "\n" + last


@pytest.mark.gpu
def test_results_do_not_depend_on_what_the_workspace_held():
    """a slice of the parity suite with every entry point starting from a NaN-filled workspace (ABC_WS_POISON=ff): a kernel that
    reads a workspace word nobody wrote in the same call would inherit the poison instead of the previous call's leftovers"""
    env = dict(os.environ)
    env["ABC_WS_POISON"] = "ff"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k",
                        "generation_matches or generation_with_40 or weight_split_kernel or weight_far or particle_ranking_pls_wilcoxon or resample_bit_exact "
                        "or perturb or speculates_on_the_component_count"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1500:]
    assert " passed" in p.stdout and "failed" not in p.stdout.splitlines()[-1], p.stdout[-500:]


def test_fuzzers_compile():
    """(CPU) the stand-alone fuzzers are valid Python: a syntax error in one would otherwise surface only on the GPU box"""
    import glob
    import py_compile
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "fuzz", "*.py")))
    assert len(files) >= 7
    for f in files:
        py_compile.compile(f, doraise=True)
