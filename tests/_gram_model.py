"""The error model of the byte-limb statistics kernel (gram.hip: k_gram_i8), one statement of it for the header
(include/abcsmc_hip.h, abc_ctx_set_gram_mode), the fixed tests (tests/test_gpu_parity.py::test_wide_gram_on_the_i8_matrix_pipe)
and the fuzzer (tests/fuzz/wide_gram_fuzz.py).  For a != b (the diagonal, the column sums and the row counts are exact):

    |G_ab - exact G_ab|  <=  2^-32 x ( KAPPA x range_a x range_b x sqrt(rows)  +  range_a |S_b|  +  range_b |S_a| )

range_c = the fixed-point range k_pilot_scale gives column c: S = min(rows of the set, 4096) rows r_q = floor(q n / S), 64 groups
of 64 consecutive q, the MEDIAN (33rd smallest) of the groups' maxima of |x - shift_c|, times 4, rounded up to a power of two;
S_c = the partition's sum of x - shift_c (the record holds it, exactly).
Every value is rounded to a grid of 2^-31 range_c (an error of at most 2^-32 range_c) and the byte products below 2^-32 of the top
one are dropped.  First term: those errors as zero-mean noise per row, adding up as sqrt(rows); KAPPA = 4 covers the largest of the
~1e4 entries of a record (Gaussian-like columns: the worst entry lies at 0.06 .. 0.1 of the bound, profiles/r06_wide_gram_fuzz.json
`worst_vs_model`).  The other two: a column whose mass sits in ONE value carries the SAME rounding error in most rows, which then
adds up coherently against the other column's sum -- small because the shift is the pilot's mean, but growing with the rows where
the noise term grows with their square root.  Relative to sqrt(G_aa G_bb) the noise term is KAPPA 2^-32 (range/sigma)_a
(range/sigma)_b / sqrt(rows): 2e-10 for Gaussian-like columns at 2e5 rows (range / sigma 10 .. 19), 1.5e-9 for a point-mass column
(range / sigma ~ 40) beside one."""
import numpy as np

KAPPA = 4.0


def pilot_range(Z, shift):
    """range_c of every column of Z (rows x C, the WHOLE set) about shift (C,), as k_pilot_scale computes it"""
    n = Z.shape[0]
    S = min(n, 4096)
    q = np.arange(S, dtype=np.uint64)
    rows = ((q * np.uint64(n)) // np.uint64(S)).astype(np.int64)
    a = np.abs(Z[rows] - shift)
    a = np.where(np.isnan(a), 0.0, a)
    if S < 4096:
        a = np.vstack([a, np.zeros((4096 - S, a.shape[1]))])
    gm = a.reshape(4, 16, 64, -1).max(axis=2).reshape(64, -1)          # q = 1024 j + 64 wave + lane
    med = np.sort(gm, axis=0)[32]
    with np.errstate(divide="ignore"):
        ex = np.where(med > 0, np.frexp(med)[1] + 2, -1000)
    return np.ldexp(1.0, ex)


def gram_error_bound(Z, shift, lo, hi):
    """C x C matrix of absolute bounds on the off-diagonal entries of the Gram of the partition rows lo .. hi - 1"""
    r = pilot_range(Z, shift)
    S = np.abs((Z[lo:hi] - shift).sum(axis=0))
    return 2.0 ** -32 * (KAPPA * np.outer(r, r) * np.sqrt(float(hi - lo)) + np.outer(r, S) + np.outer(S, r))
