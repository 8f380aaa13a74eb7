"""abcsmc_amd -- MI355X (gfx950) implementation of AbcSmc's per-generation numerical hot path.

Everything numeric runs in hand-written HIP kernels behind the C ABI in include/abcsmc_hip.h
(abcsmc_amd/libabcsmc_hip.so).  This package is only the host-side mirror of the reference's
`namespace ABC` free functions (abcutil.py), the device-resident generation driver (device.py,
sharded.py) and the synthetic workload generator (synthetic.py).  There is NO CPU fallback: if the
HIP library is missing or no GPU is present, calls raise.
"""
from ._lib import lib, LibraryMissing, Prior, Rng, make_priors, PRIOR_GAUSS, PRIOR_UNIF_INT, PRIOR_UNIF_REAL  # noqa: F401
from ._lib import RULE_MIN_PRESS, RULE_WILCOXON, AbcError  # noqa: F401

__all__ = ["lib", "LibraryMissing", "AbcError", "Prior", "Rng", "make_priors"]
