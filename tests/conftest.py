import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library honours its diagnostic switches (ABC_WS_POISON, ABC_ALIAS_FORCE_FAIL, ABC_KDE_TOPN_MIN_PAIRS, ...) only beside
# ABC_DIAG=1, read once at its first call: the tests that flip them need it set before the library loads
os.environ.setdefault("ABC_DIAG", "1")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from abcsmc_amd import _lib
    return _lib.default_context(0)
