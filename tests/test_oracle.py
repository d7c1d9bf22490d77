"""Pins the CPU oracle (oracle/lmm_oracle.py) with the reference's own relational tests
(test/ilmm.jl:10-14,23-26; test/oilmm.jl:10-14,23-26; test/independent_mogp.jl:40-60) and the
six literal numbers of the reference notebook (tests/golden/notebook_literals.json)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import lmm_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def toy(seed, m, p=3, orth=False, n_train=3, n_test=2):
    """Shape of test/test_utils.jl:1-35 (5 points on [0,10], 3 outputs, 3/2 split)."""
    rng = np.random.default_rng(seed)
    x = np.linspace(0.0, 10.0, n_train + n_test)
    perm = rng.permutation(n_train + n_test)
    xtr, xte = x[perm[:n_train]], x[perm[n_train:]]
    kinds = ["se", "matern32", "matern32"][:m]
    gps = [{"kind": k, "variance": 1.0, "lengthscale": 1.0, "mean": 0.0} for k in kinds]
    A = rng.uniform(size=(p, m))
    if orth:
        U, S, _ = np.linalg.svd(A, full_matrices=False)
    else:
        U = S = None
    ytr, yte = rng.standard_normal(n_train * p), rng.standard_normal(n_test * p)
    return gps, A, U, S, xtr, xte, ytr, yte


def test_notebook_literals():
    g = json.load(open(os.path.join(HERE, "golden", "notebook_literals.json")))
    x = np.linspace(0.0, 20.0, 576)
    assert x[1] - x[0] == pytest.approx(g["spacing"], rel=1e-15)
    gp = {"kind": "matern52", "variance": 1.0, "lengthscale": 1.0}
    k = lambda r: float(O.kernel_eval("matern52", 1.0, 1.0, np.array([r]))[0])
    assert k(g["spacing"]) == pytest.approx(g["K_21"], rel=1e-14)
    assert k(20.0) == pytest.approx(g["K_n1_at_20"], rel=1e-13)
    assert k(20.0 - g["spacing"]) == pytest.approx(g["K_nm1_1_at_20_minus_spacing"], rel=1e-12)
    # First 2x2 of the Cholesky with the projected noise s = U11^2 - 1 on the diagonal.
    s = g["U_11"] ** 2 - 1.0
    K = O.kernelmatrix(gp, x[:2]) + s * np.eye(2)
    Uf = np.linalg.cholesky(K).T
    assert Uf[0, 0] == pytest.approx(g["U_11"], rel=1e-15)
    assert Uf[0, 1] == pytest.approx(g["U_12"], rel=1e-14)
    assert Uf[1, 1] == pytest.approx(g["U_22"], rel=1e-9)     # cancellation-limited


@pytest.mark.parametrize("m", [3, 2, 1])
def test_ilmm_equals_naive(m):
    """test/ilmm.jl:10-14,23-26 with sigma^2 = 1e-6."""
    gps, H, _, _, xtr, xte, ytr, yte = toy(10 + m, m)
    s2 = 1e-6
    assert O.ilmm_logpdf(gps, H, xtr, s2, ytr) == pytest.approx(O.naive_logpdf(gps, H, xtr, s2, ytr), rel=1e-6)
    M, C = O.ilmm_mean_cov(gps, H, xtr, s2)
    np.testing.assert_allclose(M, O.naive_mean(gps, H, xtr), atol=1e-12)
    np.testing.assert_allclose(C, O.naive_cov(gps, H, xtr) + s2 * np.eye(len(M)), rtol=1e-8, atol=1e-12)
    post = O.ilmm_posterior(gps, H, xtr, s2, ytr)
    Mp, Cp = O.ilmm_mean_cov(post, H, xte, s2)
    Mn, Cn = O.naive_posterior_mean_cov(gps, H, xtr, s2, ytr, xte)
    np.testing.assert_allclose(Mp, Mn, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(Cp, Cn + s2 * np.eye(len(Mn)), rtol=1e-5, atol=1e-6)
    lp = O.ilmm_logpdf(post, H, xte, s2, yte)
    ln = O.gaussian_logpdf(Mn, Cn + s2 * np.eye(len(Mn)), yte)
    assert lp == pytest.approx(ln, rel=1e-6)


@pytest.mark.parametrize("m", [3, 2, 1])
def test_oilmm_equals_ilmm_and_naive(m):
    """test/oilmm.jl:10-14,23-26 with sigma^2 = 0.1."""
    gps, _, U, S, xtr, xte, ytr, yte = toy(20 + m, m, orth=True)
    H = O.orthogonal_dense(U, S)
    s2 = 0.1
    lo = O.oilmm_logpdf(gps, U, S, xtr, s2, ytr)
    assert lo == pytest.approx(O.ilmm_logpdf(gps, H, xtr, s2, ytr), rel=1e-8)
    assert lo == pytest.approx(O.naive_logpdf(gps, H, xtr, s2, ytr), rel=1e-12)
    Mo, Vo = O.oilmm_mean_var(gps, U, S, xtr, s2)
    Mi, Vi = O.ilmm_mean_var(gps, H, xtr, s2)
    np.testing.assert_allclose(Mo, Mi, atol=1e-12)
    np.testing.assert_allclose(Vo, Vi, rtol=1e-10)
    po = O.oilmm_posterior(gps, U, S, xtr, s2, ytr)
    pi = O.ilmm_posterior(gps, H, xtr, s2, ytr)
    Mo, Vo = O.oilmm_mean_var(po, U, S, xte, s2)
    Mi, Vi = O.ilmm_mean_var(pi, H, xte, s2)
    np.testing.assert_allclose(Mo, Mi, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(Vo, Vi, rtol=1e-6, atol=1e-8)
    Mn, Cn = O.naive_posterior_mean_cov(gps, H, xtr, s2, ytr, xte)
    np.testing.assert_allclose(Mo, Mn, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(Vo, np.diag(Cn) + s2, rtol=1e-9, atol=1e-10)
    assert O.oilmm_logpdf(po, U, S, xte, s2, yte) == pytest.approx(O.ilmm_logpdf(pi, H, xte, s2, yte), rel=1e-6)
    assert O.oilmm_logpdf(po, U, S, xte, s2, yte) == pytest.approx(
        O.gaussian_logpdf(Mn, Cn + s2 * np.eye(len(Mn)), yte), rel=1e-9)


def test_mogp_equals_singles():
    """test/independent_mogp.jl:40-43,53-60."""
    rng = np.random.default_rng(5)
    x, xs = np.linspace(1, 2, 5)[:3], np.linspace(1, 2, 5)[3:]
    f1 = {"kind": "matern32", "variance": 1.0, "lengthscale": 1.0, "mean": 30.0}
    f2 = {"kind": "se", "variance": 1.0, "lengthscale": 1.0, "mean": 10.0}
    y = rng.standard_normal(6) + np.repeat([30.0, 10.0], 3)
    assert O.mogp_logpdf([f1, f2], x, 0.1, y) == pytest.approx(
        O.gp_logpdf(f1, x, 0.1, y[:3]) + O.gp_logpdf(f2, x, 0.1, y[3:]), rel=1e-14)
    # H = I, U = I, S = 1 makes the OILMM an IndependentMOGP (test/independent_mogp.jl:126-128 idea)
    assert O.mogp_logpdf([f1, f2], x, 0.1, y) == pytest.approx(
        O.oilmm_logpdf([f1, f2], np.eye(2), np.ones(2), x, 0.1, y), rel=1e-12)
    post = O.mogp_posterior([f1, f2], x, 0.1, y)
    m, v = O.mogp_mean_var(post, xs)
    m1, v1 = O.gp_mean_var(O.gp_posterior(f1, x, 0.1, y[:3]), xs)
    np.testing.assert_allclose(m[:2], m1)
    np.testing.assert_allclose(v[:2], v1)


def test_rand_matches_cov_structure():
    """rand with caller-supplied normals: the linear map z -> sample has covariance = model cov."""
    gps, _, U, S, xtr, *_ = toy(3, 2, orth=True)
    n, p, m = 3, 3, 2
    H = O.orthogonal_dense(U, S)
    cols = []
    for k in range(m * n + n * p):
        e = np.zeros(m * n + n * p); e[k] = 1.0
        cols.append(O.oilmm_rand(gps, U, S, xtr, 0.1, e[:m * n], e[m * n:]))
    A = np.stack(cols, axis=1)
    np.testing.assert_allclose(A @ A.T, O.naive_cov(gps, H, xtr) + 0.1 * np.eye(n * p), rtol=1e-9, atol=1e-12)
    A2 = np.stack([O.ilmm_rand(gps, H, xtr, 0.1, np.eye(m * n + n * p)[k][:m * n], np.eye(m * n + n * p)[k][m * n:])
                   for k in range(m * n + n * p)], axis=1)
    np.testing.assert_allclose(A2 @ A2.T, O.naive_cov(gps, H, xtr) + 0.1 * np.eye(n * p), rtol=1e-9, atol=1e-9)


def test_helpers():
    """test/ilmm.jl:55-71, test/orthogonal_matrix.jl:1-14."""
    assert O.reshape_y(np.arange(16.0), 8).shape == (2, 8)
    assert O.reshape_y(np.arange(16.0), 2).shape == (8, 2)
    with pytest.raises(ValueError):
        O.orthogonal_validate(np.random.default_rng(0).uniform(size=(5, 3)))
    U, S, _ = np.linalg.svd(np.random.default_rng(0).uniform(size=(5, 3)), full_matrices=False)
    O.orthogonal_validate(U)
    np.testing.assert_allclose(O.orthogonal_dense(U, S), U @ np.diag(np.sqrt(S)))
    with pytest.raises(RuntimeError, match="out dim"):
        O.oilmm_logpdf([{"kind": "se"}], U[:, :1], S[:1], np.arange(3.0), 0.1, np.zeros(3 * 4))


def test_oilmm_logpdf_grad_matches_finite_differences():
    """Analytic gradient of the oracle's OILMM logpdf (what Zygote.gradient(logpdf, fx, y) would return;
    reference test/oilmm.jl:31-32) against central finite differences."""
    rng = np.random.default_rng(42)
    n, p, m = 9, 4, 2
    x = np.sort(rng.uniform(0, 5, n))
    gps = [{"kind": "matern52", "variance": 1.3, "lengthscale": 0.8, "mean": 0.2},
           {"kind": "se", "variance": 0.7, "lengthscale": 1.4, "mean": -0.5}]
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    y = rng.standard_normal(n * p)
    G = O.oilmm_logpdf_grad(gps, U, S, x, 0.3, y)
    assert G["value"] == pytest.approx(O.oilmm_logpdf(gps, U, S, x, 0.3, y), rel=1e-13)
    h = 1e-6

    def fd(fun):
        return (fun(h) - fun(-h)) / (2 * h)

    for k in (0, 7, 20):
        e = np.zeros(n * p); e[k] = 1.0
        assert G["y"][k] == pytest.approx(fd(lambda t: O.oilmm_logpdf(gps, U, S, x, 0.3, y + t * e)), rel=1e-6, abs=1e-7)
    assert G["sigma2"] == pytest.approx(fd(lambda t: O.oilmm_logpdf(gps, U, S, x, 0.3 + t, y)), rel=1e-6)
    for l in range(m):
        e = np.zeros(m); e[l] = 1.0
        assert G["S"][l] == pytest.approx(fd(lambda t: O.oilmm_logpdf(gps, U, S + t * e, x, 0.3, y)), rel=1e-6)
        for key in ("variance", "lengthscale", "mean"):
            def f(t, l=l, key=key):
                g2 = [dict(g) for g in gps]; g2[l][key] += t
                return O.oilmm_logpdf(g2, U, S, x, 0.3, y)
            assert G["gps"][l][key] == pytest.approx(fd(f), rel=1e-5, abs=1e-7)
    for (o, l) in ((0, 0), (3, 1), (2, 0)):
        E = np.zeros((p, m)); E[o, l] = 1.0
        assert G["U"][o, l] == pytest.approx(fd(lambda t: O.oilmm_logpdf(gps, U + t * E, S, x, 0.3, y)), rel=1e-5, abs=1e-6)


def test_mogp_diagonal_noise_oracle_reduces_to_scalar_noise():
    """oracle.mogp_logpdf_diag (dense generic path with a general Diagonal) == the per-latent scalar-noise path for a
    constant diagonal (reference src/independent_mogp.jl:74-80 vs the generic fallback behind :222-229)."""
    rng = np.random.default_rng(5)
    x = np.sort(rng.uniform(0, 4, 25))
    gps = [{"kind": "se", "variance": 1.3, "lengthscale": 0.7, "mean": 0.2}, {"kind": "matern52", "variance": 0.6, "lengthscale": 1.1, "mean": -0.1}]
    y = rng.standard_normal(50)
    assert O.mogp_logpdf_diag(gps, x, np.full(50, 0.3), y) == pytest.approx(O.mogp_logpdf(gps, x, 0.3, y), rel=1e-12)


def test_dense_ilmm_sequential_conditioning_matches_naive_two_batch_gp():
    """oracle.ilmm_posterior_condition (reference src/ilmm.jl:184-198 applied to a posterior; TestUtils on `pi`,
    test/ilmm.jl:34-37) == the naive dense GP conditioned on both batches, each with its own observation noise."""
    rng = np.random.default_rng(1)
    m, p, n1, n2, ns = 2, 3, 12, 7, 5
    x1, x2, xs = np.sort(rng.uniform(0, 6, n1)), np.sort(rng.uniform(0, 6, n2)), np.sort(rng.uniform(0, 6, ns))
    gps = [{"kind": "se", "variance": 1.2, "lengthscale": 0.8, "mean": 0.1}, {"kind": "matern32", "variance": 0.7, "lengthscale": 1.3, "mean": -0.2}]
    H = rng.uniform(0.2, 1, (p, m))
    y1, y2 = rng.standard_normal(n1 * p), rng.standard_normal(n2 * p)
    po2 = O.ilmm_posterior_condition(O.ilmm_posterior(gps, H, x1, 0.1, y1), H, x2, 0.3, y2)
    mo, vo = O.ilmm_mean_var(po2, H, xs, 0.05)
    K11, K12, K22 = O.naive_cov(gps, H, x1), O.naive_cov(gps, H, x1, x2), O.naive_cov(gps, H, x2)
    Kall = np.block([[K11 + 0.1 * np.eye(n1 * p), K12], [K12.T, K22 + 0.3 * np.eye(n2 * p)]])
    Ks = np.vstack([O.naive_cov(gps, H, x1, xs), O.naive_cov(gps, H, x2, xs)])
    yall = np.concatenate([y1 - O.naive_mean(gps, H, x1), y2 - O.naive_mean(gps, H, x2)])
    mn = O.naive_mean(gps, H, xs) + Ks.T @ np.linalg.solve(Kall, yall)
    vn = np.diag(O.naive_cov(gps, H, xs) - Ks.T @ np.linalg.solve(Kall, Ks)) + 0.05
    np.testing.assert_allclose(mo, mn, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(vo, vn, rtol=1e-7)


def _fixture_cases():
    return json.load(open(os.path.join(HERE, "golden", "oracle_fixtures.json")))["cases"]


@pytest.mark.parametrize("case", _fixture_cases(), ids=lambda c: c["name"])
def test_oracle_reproduces_committed_fixtures(case):
    """tests/golden/oracle_fixtures.json (written by tests/golden/make_oracle_fixtures.py) freezes the oracle's values on
    small seeded problems; the same file is the fixed target of tests/test_gpu_parity.py::test_golden_fixtures."""
    x, xs, y, ys, s2, gps = (np.array(case["x"]), np.array(case["xs"]), np.array(case["y"]), np.array(case["ys"]),
                             case["sigma2"], case["gps"])
    if case["orthogonal"]:
        U, S = np.array(case["U"]), np.array(case["S"])
        assert O.oilmm_logpdf(gps, U, S, x, s2, y) == pytest.approx(case["logpdf"], rel=1e-12)
        post = O.oilmm_posterior(gps, U, S, x, s2, y)
        mu, var = O.oilmm_mean_var(post, U, S, xs, s2)
        assert O.oilmm_logpdf(post, U, S, xs, s2, ys) == pytest.approx(case["post_logpdf"], rel=1e-10)
    else:
        H = np.array(case["H"])
        assert O.ilmm_logpdf(gps, H, x, s2, y) == pytest.approx(case["logpdf"], rel=1e-12)
        post = O.ilmm_posterior(gps, H, x, s2, y)
        mu, var = O.ilmm_mean_var(post, H, xs, s2)
        assert O.ilmm_logpdf(post, H, xs, s2, ys) == pytest.approx(case["post_logpdf"], rel=1e-10)
    np.testing.assert_allclose(mu, case["post_mean"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(var, case["post_var"], rtol=1e-10)


def test_workload_generator_matches_oracle_copy():
    """linearmixingmodels.jl_amd/workloads.py (what bench.py feeds the HIP path) == the oracle's own generator."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lmm_workloads", os.path.join(HERE, "..", "linearmixingmodels.jl_amd", "workloads.py"))
    W = importlib.util.module_from_spec(spec); spec.loader.exec_module(W)
    for orth in (True, False):
        a, b = W.synthetic_problem(3, 5, 40, "matern52", orth, s2=0.2, seed=4), O.synthetic_problem(3, 5, 40, "matern52", orth, s2=0.2, seed=4)
        assert a.keys() == b.keys()
        for k in a:
            if isinstance(a[k], np.ndarray):
                np.testing.assert_array_equal(a[k], b[k])
            else:
                assert a[k] == b[k]


def test_mogp_cross_cov_equals_naive_lmm_kernel_with_identity_mixing():
    """reference test/independent_mogp.jl:140-141: cov(f, x, x') == cov(GP(LinearMixingModelKernel(kernels, I)), x, x') for by-features
    x and by-outputs x' (and every other combination: src/independent_mogp.jl:66-71, 184-215)."""
    rng = np.random.default_rng(8)
    m, n, n2 = 2, 3, 4
    gps = [{"kind": "se", "variance": 1.0, "lengthscale": 1.0, "mean": 0.0}, {"kind": "se", "variance": 0.5, "lengthscale": 1.0, "mean": 0.0}]
    x, y = rng.uniform(0, 3, n), np.linspace(0.0, 3.0, n2)
    naive = O.naive_cov(gps, np.eye(m), x, y)
    for xf in (False, True):
        for yf in (False, True):
            ri = O.reorder_indices_outputs_to_features(n, m) if xf else np.arange(m * n)
            ci = O.reorder_indices_outputs_to_features(n2, m) if yf else np.arange(m * n2)
            np.testing.assert_allclose(O.mogp_cross_cov(gps, x, y, xf, yf), naive[np.ix_(ri, ci)], rtol=1e-14, atol=1e-15)
    # known answer of the index vector (test/independent_mogp.jl:86-98: [1,1,1,2,2,2] <-> [1,2,1,2,1,2], 0-based here)
    assert list(np.array([1, 1, 1, 2, 2, 2])[O.reorder_indices_outputs_to_features(3, 2)]) == [1, 2, 1, 2, 1, 2]
    # posterior latents: cov(f, x, x) of the cross form is the posterior covariance
    po = O.mogp_posterior(gps, x, 0.1, rng.standard_normal(m * n))
    np.testing.assert_allclose(O.mogp_cross_cov(po, y, y), O.mogp_cov(po, y), rtol=1e-12, atol=1e-14)
