"""(CPU) the error model of the byte-limb statistics kernel as the tests state it (tests/_gram_model.py): the restatement of
k_pilot_scale's range and the bound's shape -- so that a typo in the shared model shows without a GPU."""
import numpy as np

from _gram_model import KAPPA, gram_error_bound, pilot_range


def test_pilot_range_is_a_power_of_two_around_the_sampled_spread():
    g = np.random.default_rng(1)
    n = 50_000
    Z = np.column_stack([g.normal(size=n), 1e3 * g.normal(size=n) + 7.0, np.full(n, 2.5)])
    shift = np.array([0.0, 7.0, 2.5])
    r = pilot_range(Z, shift)
    assert r.shape == (3,)
    for c, sd in ((0, 1.0), (1, 1e3)):
        assert np.log2(r[c]) == np.round(np.log2(r[c]))                  # a power of two
        assert 4 * 2.0 * sd < r[c] < 8 * 3.2 * sd                         # 4 x (the median of 64 maxima of 64 normals ~ 2.4 sd), rounded up
    assert r[2] == 2.0 ** -1000                                           # a constant column: no spread in the sample
    # one spike in the sample moves one of the 64 group maxima, not their median
    Z2 = Z.copy()
    Z2[(7 * n) // 4096, 0] = 1e6
    assert pilot_range(Z2, shift)[0] == r[0]


def test_bound_is_symmetric_and_grows_with_the_square_root_of_the_rows():
    g = np.random.default_rng(2)
    Z = g.normal(size=(40_000, 5)) * np.array([1.0, 10.0, 0.1, 3.0, 1.0])
    shift = Z[:4096].mean(axis=0)
    b1 = gram_error_bound(Z, shift, 0, 10_000)
    b4 = gram_error_bound(Z, shift, 0, 40_000)
    assert np.allclose(b1, b1.T) and np.all(b1 > 0)
    r = pilot_range(Z, shift)
    noise1 = KAPPA * 2.0 ** -32 * np.outer(r, r) * np.sqrt(10_000.0)
    noise4 = KAPPA * 2.0 ** -32 * np.outer(r, r) * np.sqrt(40_000.0)
    assert np.all(b1 >= noise1) and np.all(b4 >= noise4) and np.allclose(noise4, 2 * noise1)
    # the coherent part: range_a |S_b| + range_b |S_a|
    S = np.abs((Z[:10_000] - shift).sum(axis=0))
    assert np.allclose(b1 - noise1, 2.0 ** -32 * (np.outer(r, S) + np.outer(S, r)))
