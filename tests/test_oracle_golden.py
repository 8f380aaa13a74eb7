"""CPU: pins the oracle against (a) the known answers the reference's own tests hold, (b) published
GSL known answers, (c) independent implementations (scikit-learn / numpy / scipy fixtures)."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ref():
    with open(os.path.join(G, "reference_vectors.json")) as f:
        return json.load(f)


def test_ordered_reference_vectors(oracle, ref):          # tests/pls.cpp:15-23
    for case in ref["ordered"]:
        assert list(oracle.ordered(case["in"])) == case["out"]


def test_ordered_ties_and_empty(oracle):
    assert list(oracle.ordered([3.0, 1.0, 3.0, 1.0])) == [1, 3, 0, 2]    # declared (value, index) tie-break
    assert list(oracle.ordered([5.0])) == [0]


def test_colwise_z_scores_reference_vector(oracle, ref):  # tests/abcutil.cpp:11-21
    z = oracle.colwise_z_scores(np.array(ref["colwise_z_scores"]["in"], dtype=float))
    assert ((z - np.array(ref["colwise_z_scores"]["out"])) ** 2).sum() < 1e-6
    # zero-variance column: declared deviation z = 0
    z = oracle.colwise_z_scores(np.array([[1.0, 2.0], [1.0, 4.0], [1.0, 6.0]]))
    assert np.all(z[:, 0] == 0.0)


def test_euclidean_reference_vector(oracle, ref):         # tests/abcutil.cpp:29-38
    e = ref["euclidean"]
    d = oracle.euclidean(np.array(e["sims"], dtype=float), np.array(e["ref"], dtype=float))
    assert np.linalg.norm(d - np.array(e["out"])) < 1e-6


def test_dice_identities(oracle, ref):                    # examples/README.md:29-34, examples/include/dice.h:24-42
    """dice simulator restated on the oracle's taus2 + gsl_rng_uniform_int: mean of the sum is
    n(m+1)/2 and the per-roll variance of a face is (m^2-1)/12 (the identities the reference documents)."""
    d = ref["dice"]
    n, m = d["n"], d["sides"]
    r = oracle.rng(2024)
    sums, var = [], []
    for _ in range(4000):
        faces = np.array([oracle.rng_uniform_int(r, m) + 1 for _ in range(n)], dtype=float)
        sums.append(faces.sum())
        var.append(faces.var(ddof=1))
    assert abs(np.mean(sums) - n * (m + 1) / 2) < 0.5
    assert abs(np.mean(var) - (m * m - 1) / 12.0) < 0.15
    # the observed metrics in examples/reference.json (sum 44, sd 2.39925) are one plausible roll
    assert abs(d["sum_mean"] - n * (m + 1) / 2) < 4 * np.sqrt(n * (m * m - 1) / 12.0)
    assert abs(d["sum_sd"] ** 2 - (m * m - 1) / 12.0) < 1.0


def test_taus2_gsl_manual_known_answer(oracle, ref):
    r = oracle.rng(123)
    assert oracle.rng_get(r) == ref["gsl_taus_seed123_first"]


def test_taus2_seed_zero_is_seed_one(oracle):
    a, b = oracle.rng(0), oracle.rng(1)
    assert [oracle.rng_get(a) for _ in range(5)] == [oracle.rng_get(b) for _ in range(5)]


def test_uniform_int_range(oracle):
    r = oracle.rng(7)
    v = [oracle.rng_uniform_int(r, 8) for _ in range(2000)]
    assert min(v) == 0 and max(v) == 7


@pytest.mark.parametrize("tag,A", [("a", 4), ("b", 8), ("c", 3)])
def test_pls_matches_sklearn(oracle, tag, A):
    z = np.load(os.path.join(G, "pls_sklearn.npz"))
    X, Y = z[tag + "_X"], z[tag + "_Y"]
    for method in (1, 2):
        W, Pm, Q, R = oracle.pls_fit(X, Y, A, method)
        Rs = z[tag + "_rot"]
        sgn = np.sign((Rs * R).sum(0))
        Rn = R / np.linalg.norm(R, axis=0)
        Rsn = sgn * Rs / np.linalg.norm(Rs, axis=0)
        assert np.abs(Rn - Rsn).max() < 1e-6           # sklearn's NIPALS tolerance bounds this
        for a in range(1, A + 1):
            coef = R[:, :a] @ Q[:, :a].T
            assert np.abs(coef - z[tag + "_coef"][a - 1]).max() < 1e-7


def test_pls_type1_equals_type2(oracle):
    z = np.load(os.path.join(G, "pls_sklearn.npz"))
    X, Y = z["b_X"], z["b_Y"]
    R1 = oracle.pls_fit(X, Y, 8, 1)[3]
    R2 = oracle.pls_fit(X, Y, 8, 2)[3]
    assert np.abs(R1 - R2).max() < 1e-11


def test_press_matches_direct_residuals(oracle):
    z = np.load(os.path.join(G, "pls_sklearn.npz"))
    X, Y = z["b_X"], z["b_Y"]
    Xtr, Ytr, Xte, Yte = X[:250], Y[:250], X[250:], Y[250:]
    W, Pm, Q, R = oracle.pls_fit(Xtr, Ytr, 8, 1)
    press = oracle.pls_press(Xte, Yte, R, Q)
    for a in range(1, 9):
        E = Yte - Xte @ (R[:, :a] @ Q[:, :a].T)
        assert np.allclose(press[a - 1], (E ** 2).sum(0), rtol=1e-10)
    best, per = oracle.pls_optimal_components(Xte, Yte, R, Q, 0)
    assert best == (np.argmin(press, axis=0) + 1).max()
    bw, perw = oracle.pls_optimal_components(Xte, Yte, R, Q, 1)
    assert np.all(perw <= per) and bw <= best


def test_wilcoxon_against_scipy(oracle):
    from scipy.stats import wilcoxon
    rng = np.random.default_rng(1)
    e1, e2 = rng.normal(size=400), rng.normal(size=400) * 1.2
    p = oracle.wilcoxon_p(e1, e2)
    ps = wilcoxon(np.abs(e1), np.abs(e2), correction=False, mode="approx").pvalue
    assert abs(p - ps) < 5e-3      # 4-term polynomial normal cdf; scipy applies a tie correction


def test_numerics_fixtures(oracle):
    z = np.load(os.path.join(G, "numerics.npz"))
    rc, L, cov = oracle.mvn_setup(z["theta"])
    assert rc == 0
    assert np.allclose(cov, z["cov_doubled_diag"], rtol=1e-11, atol=1e-12)
    assert np.allclose(np.tril(L), z["chol"], rtol=1e-10, atol=1e-12)
    iu = np.triu_indices(6, 1)
    assert np.allclose(L[iu], z["cov_doubled_diag"][iu], rtol=1e-11)      # decomp1 keeps the upper triangle
    assert np.allclose(oracle.doubled_variance(z["theta"]), z["dv"], rtol=1e-12)
    pdf = [oracle.ran_gaussian_pdf(x, s) for x, s in zip(z["pdf_x"], z["pdf_sigma"])]
    assert np.allclose(pdf, z["pdf"], rtol=1e-13)


def test_mvn_setup_rejects_degenerate(oracle):
    th = np.ones((10, 3))
    rc, _, _ = oracle.mvn_setup(th)
    assert rc != 0


def test_priors(oracle):
    pr = oracle.make_priors([(0, 1.0, 2.0), (1, 1, 1000), (2, -1.0, 3.0)])
    assert oracle.prior_likelihood(pr[0], 1.0) == pytest.approx(1 / (np.sqrt(2 * np.pi) * 2))
    assert oracle.prior_likelihood(pr[1], 5.0) == pytest.approx(1 / 1000)       # Priors.h:76-78
    assert oracle.prior_likelihood(pr[1], 5.5) == 0.0
    assert oracle.prior_likelihood(pr[1], 1001.0) == 0.0
    assert oracle.prior_likelihood(pr[2], 3.0) == pytest.approx(0.25)            # Priors.h:102-104
    assert oracle.prior_recast(pr[1], 2.5) == 3.0 and oracle.prior_recast(pr[1], -2.5) == -3.0   # std::round
    assert oracle.prior_valid(pr[2], 3.0001) is False


def test_alias_table_and_resample(oracle):
    rng = np.random.default_rng(3)
    w = rng.random(37)
    w[5] = 0.0
    F, A = oracle.discrete_preproc(w)
    K = w.size
    # exact marginal implied by the table: P(k) = sum over cells
    prob = np.zeros(K)
    for c in range(K):
        f = F[c] * K - c                      # undo the KNUTH_CONVENTION shift
        prob[c] += f / K
        prob[int(A[c])] += (1 - f) / K
    assert np.allclose(prob, w / w.sum(), atol=1e-12)
    r = oracle.rng(11)
    idx = oracle.resample(r, w, 50000)
    assert not np.any(idx == 5)
    # exactly one RNG output per draw
    r2 = oracle.rng(11)
    for _ in range(50000):
        oracle.rng_get(r2)
    assert (r.s1, r.s2, r.s3) == (r2.s1, r2.s2, r2.s3)


def test_weights_uniform_and_importance(oracle):
    rng = np.random.default_rng(5)
    K, Kp, P = 40, 50, 3
    th, tp = rng.normal(size=(K, P)), rng.normal(size=(Kp, P))
    wp = np.full(Kp, 1 / Kp)
    dv = 2 * tp.var(0, ddof=1)
    pri = oracle.make_priors([(0, 0.0, 3.0), (2, -10, 10), (0, 0.0, 2.0)])
    assert np.allclose(oracle.weights_uniform(K), 1 / K)
    w = oracle.weights_importance(pri, th, tp, wp, dv)
    from scipy.stats import norm
    num = norm.pdf(th[:, 0], 0, 3) * (1 / 20) * norm.pdf(th[:, 2], 0, 2)
    den = np.array([(wp * np.prod(norm.pdf(th[i] - tp, scale=np.sqrt(dv)), axis=1)).sum() for i in range(K)])
    ref = num / den
    ref /= np.linalg.norm(ref)                         # L2 normalisation, AbcUtil.cpp:583
    assert np.allclose(w, ref, rtol=1e-10)
    assert np.linalg.norm(w) == pytest.approx(1.0)
    # converged parameter: dv == 0 and equal values -> factor skipped (AbcUtil.cpp:573)
    th2, tp2 = th.copy(), tp.copy()
    th2[:, 1] = 4.0
    tp2[:, 1] = 4.0
    dv2 = dv.copy()
    dv2[1] = 0.0
    w2 = oracle.weights_importance(pri, th2, tp2, wp, dv2)
    den2 = np.array([(wp * np.prod(norm.pdf(th2[i][[0, 2]] - tp2[:, [0, 2]], scale=np.sqrt(dv2[[0, 2]])), axis=1)).sum()
                     for i in range(K)])
    num2 = norm.pdf(th2[:, 0], 0, 3) * (1 / 20) * norm.pdf(th2[:, 2], 0, 2)
    ref2 = num2 / den2
    assert np.allclose(w2, ref2 / np.linalg.norm(ref2), rtol=1e-10)


def test_ranking_pls_pipeline_consistency(oracle):
    from abcsmc_amd import synthetic
    wl = synthetic.Workload(12, 5)
    X, Y = wl.rows(0, 600)
    obs = wl.observed()
    r = oracle.particle_ranking_pls(X, Y, obs, 0.5, 0)
    assert sorted(r["idx"].tolist()) == list(range(600))
    d = r["dist"][r["idx"].astype(int)]
    assert np.all(np.diff(d) >= 0)
    # numpy restatement of AbcUtil.cpp:432-457
    mu, sd = X.mean(0), X.std(0, ddof=1)
    zx = (X - mu) / sd
    s = zx @ r["R"][:, :r["ncomp"]]
    so = ((obs - mu) / sd) @ r["R"][:, :r["ncomp"]]
    dn = np.linalg.norm(s - so, axis=1)
    assert np.allclose(dn, r["dist"], rtol=1e-10)
    i2, d2 = oracle.particle_ranking_simple(X, obs)
    assert np.allclose(d2, np.linalg.norm(zx - (obs - mu) / sd, axis=1), rtol=1e-11)


def test_samplers_respect_priors(oracle):
    rng = np.random.default_rng(9)
    K, P, n = 200, 3, 3000
    th = np.column_stack([rng.normal(5, 1, K), np.round(rng.uniform(1, 20, K)), rng.uniform(0, 1, K)])
    pri = oracle.make_priors([(0, 5.0, 3.0), (1, 1, 20), (2, 0.0, 1.0)])
    w = np.full(K, 1 / K)
    rc, L, _ = oracle.mvn_setup(th)
    assert rc == 0
    out, par, rej = oracle.sample_mvn_predictive_priors(oracle.rng(1), n, w, th, pri, L)
    assert np.all(out[:, 1] == np.round(out[:, 1])) and out[:, 1].min() >= 1 and out[:, 1].max() <= 20
    assert out[:, 2].min() >= 0 and out[:, 2].max() <= 1 and rej > 0
    dv = oracle.doubled_variance(th)
    out2, par2, fb = oracle.sample_predictive_priors(oracle.rng(1), n, w, th, pri, dv)
    assert np.array_equal(par, par2)           # both modes draw all parents first, from the same stream
    assert out2[:, 2].min() >= 0 and out2[:, 2].max() <= 1
    d = out2[:, 0] - th[par2.astype(int), 0]
    assert abs(d.var() / dv[0] - 1) < 0.15


def test_alias_host_build_is_bit_exact(tmp_path):
    """abcsmc_amd/csrc/alias_host.h (host side of the resampling table, part of the product): its blocked exact sum and
    the whole Walker table against the naive sequential loops / GSL's algorithm, bit for bit, on random and adversarial
    weight vectors (exact ties, zeros, negatives, denormals, overflow, NaN) -- tests/cxx/alias_probe.cpp"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "alias_probe")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall",
                           os.path.join(root, "tests", "cxx", "alias_probe.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr


@pytest.mark.parametrize("P,bound_rms,bound_max", [(16, 2e-8, 1e-7), (32, 3e-8, 1.5e-7)])
def test_split_operand_arithmetic_emulation(P, bound_rms, bound_max):
    """the operand split of the weight kernel k_kde_split, emulated in numpy (scripts/split_precision.py): the leading
    accumulator X is exact in f32 and the f16 operands that have to be exact are (asserted inside the script for every case)
    and the exponent error of the shipped split (3 limbs, 6 products per 16 parameters) stays inside the bound the header states"""
    import os
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "split_precision.py"), str(P), "250"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    line = [l for l in out.stdout.splitlines() if l.startswith("shipped")][0]
    rms, mx = (float(x) for x in re.findall(r"with the f32 accumulator: rms (\S+) max (\S+)", line)[0])
    assert rms < bound_rms and mx < bound_max, line
    # the f32 evaluation of a batch of 16 terms (ks_slots): what a weight inherits at worst (a row one batch dominates); budget 1e-6, the GPU tests hold 2.5e-7
    tline = [l for l in out.stdout.splitlines() if "f32 term" in l][0]
    trms, tmax = (float(x) for x in re.findall(r"relative error rms (\S+) max (\S+)", tline)[0])
    assert trms < 6e-8 and tmax < 3e-7, tline
