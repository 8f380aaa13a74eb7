"""diagnostic: where the i8 Gram differs from numpy on the 'spikes' set of tests/test_gpu_parity.py::test_wide_gram_on_the_i8_matrix_pipe"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import test_gpu_parity as T
from abcsmc_amd import _lib
N, M, P = 90000, 128, 16
wl, X, Y, obs = T._wl(M, P, N, 21)
spikes = ((5, 3, 900.0), (N // 2 + 1, 77, -2000.0), (N - 2, M - 1, 1e6), (12345, 0, 50.0), (N // 2, 100, 1e4))
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for k, (r, c, f) in enumerate(spikes):
    if which == "all" or str(k) in which:
        X[r, c] = X[:, c].mean() + f * X[:, c].std()
if which == "all" or "y" in which:
    Y[777, 2] = Y[:, 2].mean() - 300.0 * Y[:, 2].std()
X, Y = np.asfortranarray(X), np.asfortranarray(Y)
shift, sums, G = T._stats_record(_lib.default_context(0), X, Y, N // 2)
Z = np.hstack([X, Y]); C = M + P
for part, (a, b) in enumerate(((0, N // 2), (N // 2, N))):
    V = Z[a:b] - shift[:C]
    ref = V.T @ V
    scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-300
    err = (G[part][:C, :C] - ref) / scale
    err[M:, M:] = 0
    np.fill_diagonal(err, 0)
    bad = np.argwhere(np.abs(err) > 2e-8)
    print("partition", part, "entries off by more than 2e-8:", len(bad), "worst", np.abs(err).max())
    if len(bad):
        cols = np.bincount(bad.ravel(), minlength=C)
        print("  columns involved (count):", [(int(c), int(n)) for c, n in enumerate(cols) if n][:40])
        for (i, j) in bad[:6]:
            print("   (%d,%d): G %.10g ref %.10g diff %.6g" % (i, j, G[part][i, j], ref[i, j], G[part][i, j] - ref[i, j]))
