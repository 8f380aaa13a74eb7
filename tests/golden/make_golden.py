"""Generates tests/golden/*.npz / reference_vectors.json.  Run in the build container
(`python tests/golden/make_golden.py`); needs numpy, scipy and scikit-learn, none of which the
product uses.  Fixtures are DATA: seeded inputs + outputs of INDEPENDENT implementations
(scikit-learn NIPALS PLS, numpy covariance/Cholesky, scipy normal pdf) and the known answers held by
the reference's own tests (cited per entry).  Nothing here reads or copies reference source."""
import json
import os

import numpy as np
from scipy.stats import norm
from sklearn.cross_decomposition import PLSRegression

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    # --- known answers owned by the reference's tests / docs / GSL manual -------------------------
    ref = {
        "ordered": [  # /root/reference/tests/pls.cpp:15-23
            {"in": [1, 2, 3], "out": [0, 1, 2]},
            {"in": [2, 1, 3], "out": [1, 0, 2]},
        ],
        "colwise_z_scores": {  # /root/reference/tests/abcutil.cpp:11-21 (tolerance: sum sq < 1e-6)
            "in": [[1, 1, 1], [2, 3, 4], [3, 5, 7]],
            "out": [[-1, -1, -1], [0, 0, 0], [1, 1, 1]],
        },
        "euclidean": {  # /root/reference/tests/abcutil.cpp:29-38 (tolerance 1e-6)
            "sims": [[1, 1], [3, 3]], "ref": [1, 1], "out": [0, 2.828427],
        },
        "dice": {  # /root/reference/examples/README.md:29-34, examples/reference.json:27-36
            "n": 13, "sides": 8, "sum_mean": 44, "sum_sd": 2.39925,
        },
        "gsl_taus_seed123_first": 2720986350,  # GSL manual, "Random number environment variables" example
    }
    with open(os.path.join(HERE, "reference_vectors.json"), "w") as f:
        json.dump(ref, f, indent=1)

    # --- PLS2 vs scikit-learn (independent NIPALS implementation) ---------------------------------
    rng = np.random.default_rng(20240517)
    out = {}
    for tag, (N, M, P, A) in {"a": (300, 10, 4, 4), "b": (500, 32, 16, 8), "c": (200, 6, 1, 3)}.items():
        Lf = rng.normal(size=(N, 5))
        X = Lf @ rng.normal(size=(5, M)) + 0.3 * rng.normal(size=(N, M))
        Y = Lf @ rng.normal(size=(5, P)) + 0.3 * rng.normal(size=(N, P))
        X = (X - X.mean(0)) / X.std(0, ddof=1)
        Y = (Y - Y.mean(0)) / Y.std(0, ddof=1)
        sk = PLSRegression(n_components=A, scale=False, tol=1e-14, max_iter=10000).fit(X, Y)
        out[tag + "_X"], out[tag + "_Y"] = X, Y
        out[tag + "_rot"] = sk.x_rotations_
        coefs = []
        for a in range(1, A + 1):
            ska = PLSRegression(n_components=a, scale=False, tol=1e-14, max_iter=10000).fit(X, Y)
            coefs.append(ska.coef_.T.reshape(M, P))
        out[tag + "_coef"] = np.stack(coefs)          # [a-1] = M x P coefficients with a components
    np.savez_compressed(os.path.join(HERE, "pls_sklearn.npz"), **out)

    # --- covariance / Cholesky / pdf / variance --------------------------------------------------
    th = rng.normal(size=(400, 6)) @ rng.normal(size=(6, 6)) + rng.normal(size=6) * 10
    cov = np.cov(th, rowvar=False)
    cov2 = cov.copy()
    cov2[np.diag_indices(6)] *= 2
    xs = rng.normal(size=64) * 3
    sg = np.abs(rng.normal(size=64)) + 0.1
    np.savez_compressed(os.path.join(HERE, "numerics.npz"), theta=th, cov_doubled_diag=cov2,
                        chol=np.linalg.cholesky(cov2), dv=2 * th.var(0, ddof=1),
                        pdf_x=xs, pdf_sigma=sg, pdf=norm.pdf(xs, scale=sg))


if __name__ == "__main__":
    main()
