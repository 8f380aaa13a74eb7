"""CPU: the C-ABI shared library loads and exports exactly what include/abcsmc_hip.h declares;
host-only entry points (RNG) behave; no compute call is made."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "abcsmc_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(abc_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def L():
    import __graft_entry__ as g
    from abcsmc_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        g.build()
    return _lib.lib()


def test_every_declared_symbol_is_exported_and_bound(L):
    from abcsmc_amd import _lib
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), "libabcsmc_hip.so does not export %s" % n
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)


def test_nothing_but_the_abi_is_exported(L):
    """-Wl,--version-script (csrc/exports.map): the dynamic symbol table of the library is the header's list, nothing else --
    no mangled C++ helpers, no kernels' host stubs"""
    import subprocess
    from abcsmc_amd import _lib
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.SO_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared(), set(exported) ^ set(_declared())


def test_version_and_layout_helpers(L):
    assert L.abc_version() >= 100
    assert L.abc_stats_len(32, 16) == 2 + 3 * 48 + 2 * 48 * 48
    assert L.abc_model_len(32, 16, 8) > 32 * 8


def test_rng_matches_gsl_known_answer_and_oracle(L, oracle):
    from abcsmc_amd import _lib
    r = _lib.Rng()
    L.abc_rng_set(C.byref(r), 123)
    assert L.abc_rng_get(C.byref(r)) == 2720986350            # GSL manual known answer (taus, seed 123)
    o = oracle.rng(987654321)
    L.abc_rng_set(C.byref(r), 987654321)
    assert [L.abc_rng_get(C.byref(r)) for _ in range(100)] == [oracle.rng_get(o) for _ in range(100)]


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 123457, 2 ** 33 + 12345])
def test_rng_jump_equals_sequential(L, n):
    from abcsmc_amd import _lib
    a, b = _lib.Rng(), _lib.Rng()
    L.abc_rng_set(C.byref(a), 42)
    L.abc_rng_set(C.byref(b), 42)
    L.abc_rng_jump(C.byref(a), n)
    if n <= 200000:
        for _ in range(n):
            L.abc_rng_get(C.byref(b))
    else:                       # compose jumps: n = n1 + n2
        L.abc_rng_jump(C.byref(b), n - 7777)
        for _ in range(7777):
            L.abc_rng_get(C.byref(b))
    assert (a.s1, a.s2, a.s3) == (b.s1, b.s2, b.s3)


def test_missing_library_fails_loudly(monkeypatch):
    from abcsmc_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libabcsmc_hip.so")
    with pytest.raises(_lib.LibraryMissing):
        _lib.lib()


def test_no_gpu_fails_loudly(L):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from abcsmc_amd import _lib
    with pytest.raises(_lib.AbcError):
        _lib.Context(0)


def test_product_never_imports_oracle():
    """the oracle is test infrastructure: nothing in the package, the examples, the public header or the measurement / diagnostic
    scripts loads it (the fuzzers that do live under tests/fuzz/); bench.py may, in its cpu_baseline leg only"""
    for top in ("abcsmc_amd", "examples", "include", "scripts"):
        for dp, _, fns in os.walk(os.path.join(ROOT, top)):
            for fn in fns:
                if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, fn)).read()
                    assert "pyoracle" not in txt and "abc_oracle" not in txt and "liboracle" not in txt, os.path.join(top, fn)
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert bench.count("pyoracle") >= 1 and "def cpu_baseline" in bench
    before = bench.split("def cpu_baseline")[0]
    assert "import pyoracle" not in before and "from oracle" not in before, "bench.py loads the oracle outside its cpu_baseline leg"


def test_header_is_plain_c_and_links(tmp_path, L):
    """include/abcsmc_hip.h must be consumable from C (no C++ types): compile and link a C99 translation unit
    against the shared library, call only host-side entry points."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text(r'''
#include "abcsmc_hip.h"
#include <stdio.h>
int main(void) {
    abc_rng r; abc_rng_set(&r, 123);
    unsigned v = abc_rng_get(&r);
    abc_prior p = {ABC_PRIOR_GAUSS, 0, 0.0, 1.0};
    abc_generation_cfg cfg = {0};
    (void)p; (void)cfg;
    printf("%u %d %zu\n", v, abc_version(), abc_stats_len(32, 16));
    return v == 2720986350u ? 0 : 1;
}
''')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-L" + os.path.join(ROOT, "abcsmc_amd"), "-labcsmc_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "abcsmc_amd"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
