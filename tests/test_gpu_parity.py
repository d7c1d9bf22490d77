"""-m gpu parity tests: the HIP path (through the C ABI / the host mirror) against the CPU oracle on the same
seeded inputs.  Bar: rtol 1e-6 in Float64 (BASELINE.json north_star); most checks are far tighter."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import lmm_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-6     # the north-star tolerance (Float64)


@pytest.fixture(scope="module")
def lmm():
    import lmm_amd
    lmm_amd.init(0)
    return lmm_amd


def _gps(kinds, rng=None):
    out = []
    for k in kinds:
        g = {"kind": k, "variance": 1.0, "lengthscale": 1.0, "mean": 0.0}
        if rng is not None:
            g.update(variance=float(rng.uniform(0.5, 2.0)), lengthscale=float(rng.uniform(0.5, 2.0)),
                     mean=float(rng.normal()))
        out.append(g)
    return out


def _to_model(lmm, gps):
    K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
    return lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])


def _orth(rng, p, m):
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    return np.ascontiguousarray(U), S


# ---------------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------------
def test_mfma_gemm_nt_sub(lmm):
    """f64 MFMA tile kernel vs torch (asymmetric operands: catches a swapped C/D map)."""
    import torch
    lib = lmm.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    for (M, N, K, lower) in [(128, 128, 64, 0), (192, 64, 16, 0), (320, 192, 208, 0), (256, 256, 128, 1), (448, 320, 64, 1)]:
        ldc, lda, ldb = M + 16, M + 2, N + 4
        Ct = torch.randn(N, ldc, generator=g, device="cuda", dtype=torch.float64)     # column-major: [col][row]
        At = torch.randn(K, lda, generator=g, device="cuda", dtype=torch.float64)
        Bt = torch.randn(K, ldb, generator=g, device="cuda", dtype=torch.float64)
        C0 = Ct.clone()
        torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
        rc = lib.lmm_dev_gemm_nt_sub(C.c_void_p(Ct.data_ptr()), ldc, C.c_void_p(At.data_ptr()), lda,
                                     C.c_void_p(Bt.data_ptr()), ldb, M, N, K, lower)
        assert rc == 0, lib.lmm_last_error_string()
        A = At[:, :M].T; B = Bt[:, :N].T                      # M x K, N x K
        ref = C0[:, :M].T - A @ B.T                           # M x N
        got = Ct[:, :M].T
        if lower:
            mask = torch.tril(torch.ones(M, N, device="cuda", dtype=torch.bool))
            # 16x16 tiles straddling the diagonal are computed in full; only the lower triangle is contractual
            assert torch.allclose(got[mask], ref[mask], rtol=1e-12, atol=1e-11)
        else:
            assert torch.allclose(got, ref, rtol=1e-12, atol=1e-11)
        assert torch.equal(Ct[:, M:], C0[:, M:])              # padding rows untouched


@pytest.mark.parametrize("n,riders", [(64, 0), (100, 1), (200, 1), (700, 3), (1500, 1)])
def test_potrf_vs_lapack(lmm, n, riders):
    import torch
    lib = lmm.load()
    rng = np.random.default_rng(n)
    NC = (n + 63) // 64 * 64
    NR = (NC + riders + 63) // 64 * 64
    ld = NR + 2
    G = rng.standard_normal((n, n + 8))
    Kmat = G @ G.T / n + 0.5 * np.eye(n)
    full = np.zeros((NR, NC))
    full[:n, :n] = np.tril(Kmat)
    full[np.arange(n, NC), np.arange(n, NC)] = 1.0
    rhs = rng.standard_normal((riders, n))
    full[NC:NC + riders, :n] = rhs
    A = torch.zeros(NC, ld, dtype=torch.float64, device="cuda")
    A[:, :NR] = torch.from_numpy(full.T.copy()).cuda()
    W = torch.zeros(NC // 64 * 4096, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
    rc = lib.lmm_dev_potrf(C.c_void_p(A.data_ptr()), NR, NC, ld, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr()))
    assert rc == 0, lib.lmm_last_error_string()
    assert int(info.item()) == 0
    got = A[:, :NR].T.cpu().numpy()
    Lref = np.linalg.cholesky(Kmat)
    np.testing.assert_allclose(np.tril(got[:n, :n]), Lref, rtol=1e-9, atol=1e-11)
    if riders:
        import scipy.linalg as sla
        zref = sla.solve_triangular(Lref, rhs.T, lower=True).T
        np.testing.assert_allclose(got[NC:NC + riders, :n], zref, rtol=1e-8, atol=1e-10)
    # inverse diagonal blocks
    Wh = W.cpu().numpy().reshape(NC // 64, 64, 64)
    b = 0
    Lb = np.tril(got[:64, :64]) if n >= 64 else None
    if Lb is not None:
        np.testing.assert_allclose(Wh[b].T @ Lb, np.eye(64), atol=1e-9)


def test_potrf_reports_not_pd(lmm):
    import torch
    lib = lmm.load()
    n = 128
    M = np.eye(n); M[70, 70] = -1.0
    A = torch.from_numpy(M.copy()).cuda()
    W = torch.zeros(2 * 4096, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
    assert lib.lmm_dev_potrf(C.c_void_p(A.data_ptr()), n, n, n, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr())) == 0
    assert int(info.item()) == 71          # LAPACK-style 1-based failing pivot


@pytest.mark.parametrize("kind", ["se", "matern32", "matern52"])
@pytest.mark.parametrize("d", [1, 3])
def test_gram_vs_oracle(lmm, kind, d):
    import torch
    lib = lmm.load()
    rng = np.random.default_rng(7)
    n = 150
    x = rng.uniform(0, 5, size=(n,) if d == 1 else (d, n))
    gp = {"kind": kind, "variance": 1.7, "lengthscale": 0.8, "mean": 0.0}
    NC = NR = 192
    ld = NR
    A = torch.full((NC, ld), float("nan"), dtype=torch.float64, device="cuda")
    xd = torch.from_numpy(np.ascontiguousarray(x.T if d > 1 else x)).cuda()
    from lmm_amd import _lib as L
    garr = L.gps_array([gp])
    torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
    assert lib.lmm_dev_gram(C.c_void_p(A.data_ptr()), ld, NR, NC, C.c_void_p(xd.data_ptr()), d, n, garr, C.c_double(0.25)) == 0
    got = A.T.cpu().numpy()
    ref = O.kernelmatrix(gp, x) + 0.25 * np.eye(n)
    il = np.tril_indices(n)
    np.testing.assert_allclose(got[:n, :n][il], ref[il], rtol=2e-13, atol=1e-300)
    np.testing.assert_array_equal(np.tril(got[n:, n:]), np.eye(NC - n))
    assert np.all(got[n:, :n] == 0.0)


# ---------------------------------------------------------------------------------------------------
# the reference's relational tests at its own shapes, HIP path vs oracle (test/oilmm.jl, test/ilmm.jl)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [3, 2, 1])
def test_oilmm_toy_shapes(lmm, m):
    """test/oilmm.jl:44-63 shapes: p=3, n_train=3, n_test=2, sigma2=0.1, kernels SE/Matern32."""
    rng = np.random.default_rng(100 + m)
    x = np.linspace(0, 10, 5); perm = rng.permutation(5)
    xtr, xte = x[perm[:3]], x[perm[3:]]
    gps = _gps(["se", "matern32", "matern32"][:m])
    U, S = _orth(rng, 3, m)
    ytr, yte = rng.standard_normal(9), rng.standard_normal(6)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    fx = f(lmm.MOInputIsotopicByOutputs(xtr, 3), 0.1)
    ref = O.oilmm_logpdf(gps, U, S, xtr, 0.1, ytr)
    assert lmm.logpdf(fx, ytr) == pytest.approx(ref, rel=1e-11)
    assert lmm.logpdf(fx, ytr) == pytest.approx(O.naive_logpdf(gps, O.orthogonal_dense(U, S), xtr, 0.1, ytr), rel=1e-11)
    # prior marginals
    mu, v = lmm.mean_and_var(fx)
    mo, vo = O.oilmm_mean_var(gps, U, S, xtr, 0.1)
    np.testing.assert_allclose(mu, mo, atol=1e-13); np.testing.assert_allclose(v, vo, rtol=1e-13)
    # posterior
    post = lmm.posterior(fx, ytr)
    pox = post(lmm.MOInputIsotopicByOutputs(xte, 3), 0.1)
    po = O.oilmm_posterior(gps, U, S, xtr, 0.1, ytr)
    mo, vo = O.oilmm_mean_var(po, U, S, xte, 0.1)
    mu, v = lmm.mean_and_var(pox)
    np.testing.assert_allclose(mu, mo, rtol=1e-9, atol=1e-11); np.testing.assert_allclose(v, vo, rtol=1e-9)
    assert lmm.logpdf(pox, yte) == pytest.approx(O.oilmm_logpdf(po, U, S, xte, 0.1, yte), rel=1e-9)
    mg = lmm.marginals(pox)
    np.testing.assert_allclose(mg.sigma, np.sqrt(vo), rtol=1e-9)
    assert len(lmm.rand(np.random.default_rng(0), pox)) == 3 * 2


@pytest.mark.parametrize("m", [3, 2, 1])
def test_ilmm_dense_toy_shapes(lmm, m):
    """test/ilmm.jl:44-78 shapes: dense H 3 x m, sigma2 = 1e-6."""
    rng = np.random.default_rng(200 + m)
    x = np.linspace(0, 10, 5); perm = rng.permutation(5)
    xtr = x[perm[:3]]
    gps = _gps(["se", "matern32", "matern32"][:m])
    H = rng.uniform(size=(3, m))
    ytr = rng.standard_normal(9)
    fx = lmm.ILMM(_to_model(lmm, gps), H)(lmm.MOInputIsotopicByOutputs(xtr, 3), 1e-6)
    got = lmm.logpdf(fx, ytr)
    assert got == pytest.approx(O.ilmm_logpdf(gps, H, xtr, 1e-6, ytr), rel=RTOL)
    assert got == pytest.approx(O.naive_logpdf(gps, H, xtr, 1e-6, ytr), rel=RTOL)
    # prior marginals (test/ilmm.jl:10-14) and posterior marginals at the test points (test/ilmm.jl:23-26)
    mu, v = lmm.mean_and_var(fx)
    mo, vo = O.ilmm_mean_var(gps, H, xtr, 1e-6)
    np.testing.assert_allclose(mu, mo, atol=1e-12); np.testing.assert_allclose(v, vo, rtol=1e-12)
    xte = x[perm[3:]]
    post = lmm.posterior(fx, ytr)
    mu, v = lmm.mean_and_var(post(lmm.MOInputIsotopicByOutputs(xte, 3), 1e-6))
    mn, Cn = O.naive_posterior_mean_cov(gps, H, xtr, 1e-6, ytr, xte)
    mo, vo = O.ilmm_mean_var(O.ilmm_posterior(gps, H, xtr, 1e-6, ytr), H, xte, 1e-6)
    np.testing.assert_allclose(mu, mo, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(v, vo, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(mu, mn, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(v, np.diag(Cn) + 1e-6, rtol=1e-5, atol=1e-7)
    # logpdf(pi, y_test) (test/ilmm.jl:25) and rand(rng, pi) (test/ilmm.jl:27) on the dense-H posterior
    yte = rng.standard_normal(6)
    pix = post(lmm.MOInputIsotopicByOutputs(xte, 3), 1e-6)
    assert lmm.logpdf(pix, yte) == pytest.approx(O.ilmm_logpdf(O.ilmm_posterior(gps, H, xtr, 1e-6, ytr), H, xte, 1e-6, yte), rel=1e-5)
    assert len(lmm.rand(np.random.default_rng(0), pix, jitters=(1e-9, 1e-9, 1e-9))) == 3 * 2


@pytest.mark.parametrize("m", [3, 2])
def test_cov_and_mean_and_cov(lmm, m):
    """cov / mean_and_cov (test/ilmm.jl:12, test/oilmm.jl:12; reference src/ilmm.jl:132-147): prior OILMM, prior dense-H
    ILMM, posterior OILMM at test points, and the block-diagonal IndependentMOGP covariance."""
    rng = np.random.default_rng(300 + m)
    x = np.linspace(0, 10, 7); xtr, xte = x[:4], x[4:]
    gps = _gps(["se", "matern32", "matern52"][:m], rng)
    U, S = _orth(rng, 3, m)
    H = O.orthogonal_dense(U, S)
    ytr = rng.standard_normal(12)
    xin = lmm.MOInputIsotopicByOutputs(xtr, 3)
    fo = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    fd = lmm.ILMM(_to_model(lmm, gps), H)
    Mref, Cref = O.ilmm_mean_cov(gps, H, xtr, 0.1)
    for f in (fo, fd):
        M, Cm = lmm.mean_and_cov(f(xin, 0.1))
        np.testing.assert_allclose(M, Mref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(Cm, Cref, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(lmm.cov(f(xin, 0.1)), O.naive_cov(gps, H, xtr) + 0.1 * np.eye(12), rtol=1e-12, atol=1e-13)
    post = lmm.posterior(fo(xin, 0.1), ytr)
    M, Cm = lmm.mean_and_cov(post(lmm.MOInputIsotopicByOutputs(xte, 3), 0.1))
    Mn, Cn = O.naive_posterior_mean_cov(gps, H, xtr, 0.1, ytr, xte)
    np.testing.assert_allclose(M, Mn, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(Cm, Cn + 0.1 * np.eye(9), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(np.diag(Cm), lmm.var(post(lmm.MOInputIsotopicByOutputs(xte, 3), 0.1)), rtol=1e-10)
    # IndependentMOGP: dense block-diagonal covariance (reference src/independent_mogp.jl:60-63) + Sigma_y
    fm = _to_model(lmm, gps)
    Cb = lmm.cov(fm(lmm.MOInputIsotopicByOutputs(xtr, m), 0.1))
    np.testing.assert_allclose(Cb, O.mogp_cov(gps, xtr) + 0.1 * np.eye(4 * m), rtol=1e-12, atol=1e-13)


def test_mogp_toy(lmm):
    """test/independent_mogp.jl:33-60."""
    rng = np.random.default_rng(5)
    x = np.linspace(1, 2, 5); xtr, xte = x[:3], x[3:]
    gps = [{"kind": "matern32", "variance": 1.0, "lengthscale": 1.0, "mean": 30.0},
           {"kind": "se", "variance": 1.0, "lengthscale": 1.0, "mean": 10.0}]
    y = rng.standard_normal(6) + np.repeat([30.0, 10.0], 3)
    ys = rng.standard_normal(4) + np.repeat([30.0, 10.0], 2)
    f = _to_model(lmm, gps)
    fx = f(lmm.MOInputIsotopicByOutputs(xtr, 2), 0.1)
    assert lmm.logpdf(fx, y) == pytest.approx(O.mogp_logpdf(gps, xtr, 0.1, y), rel=1e-11)
    post = lmm.posterior(fx, y)
    pfx = post(lmm.MOInputIsotopicByOutputs(xte, 2), 0.1)
    po = O.mogp_posterior(gps, xtr, 0.1, y)
    mo, vo = O.mogp_mean_var(po, xte)
    mu, v = lmm.mean_and_var(pfx)
    np.testing.assert_allclose(mu, mo, rtol=1e-10); np.testing.assert_allclose(v, vo + 0.1, rtol=1e-10)
    assert lmm.logpdf(pfx, ys) == pytest.approx(O.mogp_logpdf(po, xte, 0.1, ys), rel=1e-9)


# ---------------------------------------------------------------------------------------------------
# BASELINE configs and mid sizes
# ---------------------------------------------------------------------------------------------------
def test_c0_oilmm_logpdf_posterior(lmm):
    """BASELINE configs[0]: OILMM, 3 SE latents, p=5, n=200, Float64."""
    P = O.synthetic_problem(3, 5, 200, "se", True, s2=0.1, seed=0)
    f = lmm.ILMM(_to_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    fx = f(lmm.MOInputIsotopicByOutputs(P["x"], 5), 0.1)
    ref = O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    assert lmm.logpdf(fx, P["y"]) == pytest.approx(ref, rel=1e-10)
    xs = P["x"][:50] + 0.011
    po = O.oilmm_posterior(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    mo, vo = O.oilmm_mean_var(po, P["U"], P["S"], xs, 0.1)
    mu, v = lmm.mean_and_var(lmm.posterior(fx, P["y"])(lmm.MOInputIsotopicByOutputs(xs, 5), 0.1))
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9); np.testing.assert_allclose(v, vo, rtol=1e-7)


@pytest.mark.parametrize("kind,n,m,p,d", [("matern52", 1000, 4, 6, 1), ("matern32", 777, 5, 5, 2), ("se", 513, 2, 3, 1)])
def test_oilmm_mid_sizes(lmm, kind, n, m, p, d):
    rng = np.random.default_rng(n)
    x = np.arange(n) * (20.0 / 575.0) if d == 1 else rng.uniform(0, 8, size=(d, n))
    gps = _gps([kind] * m, rng)
    U, S = _orth(rng, p, m)
    S = np.linspace(2.0, 1.0, m)
    y = rng.standard_normal(n * p)
    fx = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), 0.1)
    assert lmm.logpdf(fx, y) == pytest.approx(O.oilmm_logpdf(gps, U, S, x, 0.1, y), rel=1e-9)


def test_notebook_shape_ill_conditioned(lmm):
    """The reference notebook's shape (examples/oilmm_and_ilmm.ipynb:124-129): p=600, m=20, n=552, Matern52, sigma2=1e-6 with
    S = singular values of rand(600,20) (projected noise ~2e-8, cond(K + noise) ~1e9-1e10).  The parity bar
    (rtol 1e-6) must hold even here."""
    rng = np.random.default_rng(2)
    p, m = 600, 20
    x = np.linspace(0.0, 20.0, 576)[np.sort(np.random.default_rng(1).permutation(576)[:552])]
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    gps = _gps(["matern52"] * m)
    y = np.random.default_rng(3).standard_normal(552 * p)
    fx = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), 1e-6)
    assert lmm.logpdf(fx, y) == pytest.approx(O.oilmm_logpdf(gps, U, S, x, 1e-6, y), rel=RTOL)


def test_oilmm_device_resident_inputs(lmm):
    """x and y handed over as device (torch) tensors: same answer as host arrays."""
    import torch
    P = O.synthetic_problem(4, 8, 640, "matern52", True, seed=3)
    f = lmm.ILMM(_to_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    a = lmm.logpdf(f(lmm.MOInputIsotopicByOutputs(P["x"], 8), 0.1), P["y"])
    b = lmm.logpdf(f(lmm.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 8), 0.1), torch.from_numpy(P["y"]).cuda())
    assert a == pytest.approx(b, rel=1e-13)      # split-K tail uses f64 atomics: last-bit run-to-run differences
    assert a == pytest.approx(O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"]), rel=1e-9)


def test_sharded_partials_sum_to_whole(lmm):
    """Latent shards (as on 2 and 3 GPUs) add up to the unsharded logpdf exactly as the all-reduce would."""
    P = O.synthetic_problem(5, 7, 300, "matern52", True, seed=4)
    H = lmm.Orthogonal(P["U"], P["S"])
    x = lmm.MOInputIsotopicByOutputs(P["x"], 7)
    whole = lmm.logpdf(lmm.ILMM(_to_model(lmm, P["gps"]), H)(x, 0.1), P["y"])
    for world in (2, 3):
        parts = [lmm.logpdf(lmm.ILMM(_to_model(lmm, P["gps"]), H, shard=lmm.latent_shard(5, r, world))(x, 0.1), P["y"], r == 0)
                 for r in range(world)]
        assert sum(parts) == pytest.approx(whole, rel=1e-13)


def test_ilmm_dense_mid(lmm):
    """Dense-H ILMM: one (mn) x (mn) factorisation (reference src/ilmm.jl:150-163), m=3, n=150 -> 450."""
    P = O.synthetic_problem(3, 5, 150, "se", False, seed=5)
    P["gps"][1]["kind"] = "matern32"; P["gps"][2]["lengthscale"] = 0.7
    fx = lmm.ILMM(_to_model(lmm, P["gps"]), P["H"])(lmm.MOInputIsotopicByOutputs(P["x"], 5), 0.1)
    assert lmm.logpdf(fx, P["y"]) == pytest.approx(O.ilmm_logpdf(P["gps"], P["H"], P["x"], 0.1, P["y"]), rel=1e-8)
    xs = P["x"][:40] + 0.013
    mu, v = lmm.mean_and_var(lmm.posterior(fx, P["y"])(lmm.MOInputIsotopicByOutputs(xs, 5), 0.1))
    mo, vo = O.ilmm_mean_var(O.ilmm_posterior(P["gps"], P["H"], P["x"], 0.1, P["y"]), P["H"], xs, 0.1)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9); np.testing.assert_allclose(v, vo, rtol=1e-7)
    # logpdf and a sample of the dense-H posterior at the new points (coupled (m ns) x (m ns) latent covariance)
    ys = np.random.default_rng(8).standard_normal(40 * 5)
    po = O.ilmm_posterior(P["gps"], P["H"], P["x"], 0.1, P["y"])
    pix = lmm.posterior(fx, P["y"])(lmm.MOInputIsotopicByOutputs(xs, 5), 0.1)
    assert lmm.logpdf(pix, ys) == pytest.approx(O.ilmm_logpdf(po, P["H"], xs, 0.1, ys), rel=1e-8)
    s = lmm.rand(np.random.default_rng(21), pix, jitters=(1e-9, 1e-8, 1e-8))
    g2 = np.random.default_rng(21); z = g2.standard_normal(3 * 40); eps = g2.standard_normal(40 * 5)
    mlat, Clat = O._ilmm_latent_joint(po, xs)
    lat = mlat + np.linalg.cholesky(Clat + 1e-8 * np.eye(120)) @ z
    ref = (P["H"] @ lat.reshape(3, 40)).reshape(-1) + math.sqrt(0.1) * eps
    np.testing.assert_allclose(s, ref, rtol=1e-6, atol=1e-7)


def test_ilmm_dense_posterior_cov_and_sequential_conditioning(lmm):
    """AbstractGPs.TestUtils secondary interface on the dense-H posterior `pi` (reference test/ilmm.jl:34-37): cov /
    mean_and_cov(pi) (src/ilmm.jl:132-147 on the PosteriorGP latent) and posterior(pi, y2) (src/ilmm.jl:184-198 again).
    Checked against the oracle restatement and against the naive dense GP conditioned on both batches, each with its own
    observation noise."""
    rng = np.random.default_rng(4242)
    m, p, n1, n2, ns = 3, 4, 40, 25, 9
    x1, x2, xs = np.sort(rng.uniform(0, 6, n1)), np.sort(rng.uniform(0, 6, n2)), np.sort(rng.uniform(0, 6, ns))
    gps = _gps(["se", "matern32", "matern52"], rng)
    H = rng.uniform(0.2, 1.0, size=(p, m))
    y1, y2 = rng.standard_normal(n1 * p), rng.standard_normal(n2 * p)
    f = lmm.ILMM(_to_model(lmm, gps), H)
    post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x1, p), 0.1), y1)
    po = O.ilmm_posterior(gps, H, x1, 0.1, y1)
    xsin = lmm.MOInputIsotopicByOutputs(xs, p)
    mu, Cg = lmm.mean_and_cov(post(xsin, 0.05))
    mo, Co = O.ilmm_mean_cov(po, H, xs, 0.05)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Cg, Co, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Cg, Cg.T, rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.diag(Cg), lmm.mean_and_var(post(xsin, 0.05))[1], rtol=1e-9)
    np.testing.assert_allclose(lmm.cov(post(xsin, 0.05)), Cg, rtol=0, atol=0)
    # sequential conditioning with a different noise on the second batch
    post2 = lmm.posterior(post(lmm.MOInputIsotopicByOutputs(x2, p), 0.3), y2)
    po2 = O.ilmm_posterior_condition(po, H, x2, 0.3, y2)
    mu2, v2 = lmm.mean_and_var(post2(xsin, 0.05))
    mo2, vo2 = O.ilmm_mean_var(po2, H, xs, 0.05)
    np.testing.assert_allclose(mu2, mo2, rtol=1e-8, atol=1e-10); np.testing.assert_allclose(v2, vo2, rtol=1e-8)
    # naive dense GP over by-outputs stacked data: batch 1 with noise 0.1, batch 2 with noise 0.3
    K11, K12, K22 = O.naive_cov(gps, H, x1), O.naive_cov(gps, H, x1, x2), O.naive_cov(gps, H, x2)
    Kall = np.block([[K11 + 0.1 * np.eye(n1 * p), K12], [K12.T, K22 + 0.3 * np.eye(n2 * p)]])
    Ks = np.vstack([O.naive_cov(gps, H, x1, xs), O.naive_cov(gps, H, x2, xs)])
    yall = np.concatenate([y1 - O.naive_mean(gps, H, x1), y2 - O.naive_mean(gps, H, x2)])
    mn = O.naive_mean(gps, H, xs) + Ks.T @ np.linalg.solve(Kall, yall)
    vn = np.diag(O.naive_cov(gps, H, xs) - Ks.T @ np.linalg.solve(Kall, Ks)) + 0.05
    np.testing.assert_allclose(mu2, mn, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(v2, vn, rtol=1e-5)
    # the first handle is still valid and unchanged
    np.testing.assert_allclose(lmm.mean_and_cov(post(xsin, 0.05))[1], Cg, rtol=0, atol=0)


def _golden_cases():
    import json, os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_fixtures.json")))["cases"]


@pytest.mark.parametrize("case", _golden_cases(), ids=lambda c: c["name"])
def test_golden_fixtures(lmm, case):
    """The HIP path against the committed fixture vectors (tests/golden/oracle_fixtures.json: frozen oracle values with
    their full inputs): logpdf, posterior marginals at x*, logpdf of the posterior at (x*, y*).  rtol 1e-6 is the bar."""
    x, xs, y, ys, s2, p = np.array(case["x"]), np.array(case["xs"]), np.array(case["y"]), np.array(case["ys"]), case["sigma2"], case["p"]
    H = lmm.Orthogonal(np.array(case["U"]), np.array(case["S"])) if case["orthogonal"] else np.array(case["H"])
    f = lmm.ILMM(_to_model(lmm, case["gps"]), H)
    fx = f(lmm.MOInputIsotopicByOutputs(x, p), s2)
    assert lmm.logpdf(fx, y) == pytest.approx(case["logpdf"], rel=1e-9)
    post = lmm.posterior(fx, y)
    pix = post(lmm.MOInputIsotopicByOutputs(xs, p), s2)
    mu, var = lmm.mean_and_var(pix)
    np.testing.assert_allclose(mu, case["post_mean"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(var, case["post_var"], rtol=1e-7)
    assert lmm.logpdf(pix, ys) == pytest.approx(case["post_logpdf"], rel=1e-7)


def test_ilmm_identical_kernels_decoupled_equals_dense(lmm):
    """Dense-H ILMM whose latents share one kernel (BASELINE configs[1] shape): the decoupled shortcut (m independent
    n x n factorisations under the eigen-rotation of SigmaT) equals the reference's single (mn) x (mn) factorisation."""
    from lmm_amd import model as M
    P = O.synthetic_problem(4, 6, 200, "se", False, seed=7)
    for g, mu in zip(P["gps"], [0.3, -1.0, 0.0, 2.0]):
        g["mean"] = mu                      # means may differ; kernels are identical
    fx = lmm.ILMM(_to_model(lmm, P["gps"]), P["H"])(lmm.MOInputIsotopicByOutputs(P["x"], 6), 0.1)
    ref = O.ilmm_logpdf(P["gps"], P["H"], P["x"], 0.1, P["y"])
    try:
        M.ILMM_ALLOW_DECOUPLED = True
        dec = lmm.logpdf(fx, P["y"]); assert M.ILMM_LAST_PATH == "decoupled"
        M.ILMM_ALLOW_DECOUPLED = False
        den = lmm.logpdf(fx, P["y"]); assert M.ILMM_LAST_PATH == "dense"
    finally:
        M.ILMM_ALLOW_DECOUPLED = True
    assert dec == pytest.approx(ref, rel=1e-9) and den == pytest.approx(ref, rel=1e-9)
    assert dec == pytest.approx(den, rel=1e-10)
    # distinct kernels never take the shortcut
    P["gps"][1]["lengthscale"] = 0.5
    fx2 = lmm.ILMM(_to_model(lmm, P["gps"]), P["H"])(lmm.MOInputIsotopicByOutputs(P["x"], 6), 0.1)
    assert lmm.logpdf(fx2, P["y"]) == pytest.approx(O.ilmm_logpdf(P["gps"], P["H"], P["x"], 0.1, P["y"]), rel=1e-9)
    assert M.ILMM_LAST_PATH == "dense"


def test_mogp_by_features(lmm):
    """MOInputIsotopicByFeatures on an IndependentMOGP (reference src/independent_mogp.jl:128-229; test/independent_mogp.jl:
    106-128 shape: x = range(0,2;length=3), kernels SE and 0.5*SE, scalar noise): equals the by-outputs answer after
    reordering, and the naive GP with H = I."""
    rng = np.random.default_rng(123456)
    xv = np.linspace(0.0, 2.0, 3)
    gps = [{"kind": "se", "variance": 1.0, "lengthscale": 1.0, "mean": 0.0}, {"kind": "se", "variance": 0.5, "lengthscale": 1.0, "mean": 0.0}]
    f = _to_model(lmm, gps)
    xf = lmm.MOInputIsotopicByFeatures(xv, 2)
    y_bf = rng.standard_normal(6)
    f2o = lmm.indices_which_reorder_features_to_outputs(xf) - 1
    y_bo = y_bf[f2o]
    assert lmm.logpdf(f(xf, 0.1), y_bf) == pytest.approx(O.mogp_logpdf(gps, xv, 0.1, y_bo), rel=1e-12)
    assert lmm.logpdf(f(xf, 0.1), y_bf) == pytest.approx(O.naive_logpdf(gps, np.eye(2), xv, 0.1, y_bo), rel=1e-12)
    mu, v = lmm.mean_and_var(f(xf, 0.1))
    mo, vo = O.mogp_mean_var(gps, xv)
    o2f = lmm.indices_which_reorder_outputs_to_features(xf) - 1
    np.testing.assert_allclose(mu, mo[o2f], atol=1e-14); np.testing.assert_allclose(v, (vo + 0.1)[o2f], rtol=1e-13)
    post = lmm.posterior(f(xf, 0.1), y_bf)
    mu, v = lmm.mean_and_var(post(xf, 0.1))
    mo, vo = O.mogp_mean_var(O.mogp_posterior(gps, xv, 0.1, y_bo), xv)
    np.testing.assert_allclose(mu, mo[o2f], rtol=1e-10); np.testing.assert_allclose(v, (vo + 0.1)[o2f], rtol=1e-10)
    assert len(lmm.rand(np.random.default_rng(0), f(xf, 0.1))) == 6


def test_mogp_heteroscedastic_diagonal_noise(lmm):
    """General `Diagonal` observation noise on an IndependentMOGP (reference src/independent_mogp.jl:149-159 reorder of
    Sigma_y, :222-229 by-features logpdf): per-point noise rides the Gram diagonal; equals the dense Gaussian the reference's
    generic fallback evaluates, in both input orders; a constant vector reproduces the scalar-noise answer; ILMM rejects it."""
    rng = np.random.default_rng(77)
    n, m = 70, 3
    xv = np.sort(rng.uniform(0, 5, n))
    gps = _gps(["se", "matern32", "matern52"], rng)
    f = _to_model(lmm, gps)
    noise_bo = rng.uniform(0.05, 0.5, n * m)
    y_bo = rng.standard_normal(n * m)
    ref = O.mogp_logpdf_diag(gps, xv, noise_bo, y_bo)
    xo, xf = lmm.MOInputIsotopicByOutputs(xv, m), lmm.MOInputIsotopicByFeatures(xv, m)
    assert lmm.logpdf(f(xo, noise_bo), y_bo) == pytest.approx(ref, rel=1e-11)
    o2f = lmm.indices_which_reorder_outputs_to_features(xf) - 1
    assert lmm.logpdf(f(xf, noise_bo[o2f]), y_bo[o2f]) == pytest.approx(ref, rel=1e-11)
    assert lmm.logpdf(f(xo, np.full(n * m, 0.2)), y_bo) == pytest.approx(lmm.logpdf(f(xo, 0.2), y_bo), rel=1e-13)
    import torch
    assert lmm.logpdf(f(xo, torch.from_numpy(noise_bo).cuda()), torch.from_numpy(y_bo).cuda()) == pytest.approx(ref, rel=1e-11)
    U, S = _orth(rng, 4, m)
    with pytest.raises(TypeError):
        lmm.logpdf(lmm.FiniteGP(lmm.ILMM(f, lmm.Orthogonal(U, S)), lmm.MOInputIsotopicByOutputs(xv, 4), rng.uniform(0.1, 0.2, 4 * n)),
                   rng.standard_normal(4 * n))


def test_mean_only_path(lmm):
    """mean(fx) (reference src/ilmm.jl:142) through the means-only entry (var_out = NULL: mu + K(x*,x) alpha, no solve) equals
    mean_and_var(fx)[0] and the oracle, prior and posterior, d = 1 and 2."""
    rng = np.random.default_rng(31)
    for d in (1, 2):
        n, ns, p, m = 120, 37, 4, 3
        x = np.sort(rng.uniform(0, 5, n)) if d == 1 else rng.uniform(0, 3, (d, n))
        xs = np.sort(rng.uniform(0, 5, ns)) if d == 1 else rng.uniform(0, 3, (d, ns))
        gps = _gps(["se", "matern32", "matern52"], rng)
        U, S = _orth(rng, p, m)
        f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
        y = rng.standard_normal(n * p)
        xin, xsin = lmm.MOInputIsotopicByOutputs(x, p), lmm.MOInputIsotopicByOutputs(xs, p)
        np.testing.assert_allclose(lmm.mean(f(xsin, 0.1)), O.oilmm_mean_var(gps, U, S, xs, 0.1)[0], rtol=1e-12, atol=1e-13)
        post = lmm.posterior(f(xin, 0.1), y)
        mo = O.oilmm_mean_var(O.oilmm_posterior(gps, U, S, x, 0.1, y), U, S, xs, 0.1)[0]
        got = lmm.mean(post(xsin, 0.1))
        np.testing.assert_allclose(got, mo, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(got, lmm.mean_and_var(post(xsin, 0.1))[0], rtol=1e-9, atol=1e-10)


def test_device_normals_and_device_rand(lmm):
    """lmm_normals (Philox4x32-10 + Box-Muller on the device; SURVEY.md 8a K7 'optional Philox'): reproducible per (seed, stream),
    standard-normal moments, and rand() fed from it equals rand() fed the same numbers from the host."""
    import torch
    g1, g2 = lmm.DeviceNormals(1234), lmm.DeviceNormals(1234)
    a, b = g1.standard_normal(200001), g2.standard_normal(200001)
    assert torch.equal(a, b)
    c = g1.standard_normal(200001)                       # next stream: different numbers
    assert not torch.equal(a, c)
    x = torch.cat([a, c]).cpu().numpy()
    assert abs(x.mean()) < 0.01 and abs(x.var() - 1.0) < 0.01 and abs((x ** 3).mean()) < 0.03 and abs((x ** 4).mean() - 3.0) < 0.08
    assert np.isfinite(x).all() and np.abs(x).max() < 7.0
    host = np.empty(1001)                                 # host output pointer, odd count
    from lmm_amd import _lib as L
    import ctypes as C
    L.check(lmm.load().lmm_normals(C.c_ulonglong(1234), C.c_ulonglong(0), C.c_size_t(1001), L.Arr(host, True).ptr))
    np.testing.assert_array_equal(host, a[:1001].cpu().numpy())
    # a sample drawn with device normals == the same normals handed over from the host
    rng = np.random.default_rng(5)
    n, p, m = 90, 4, 3
    xx = np.sort(rng.uniform(0, 5, n))
    gps = _gps(["se", "matern32", "matern52"], rng)
    U, S = _orth(rng, p, m)
    fx = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(xx, p), 0.1)
    s_dev = lmm.rand(lmm.DeviceNormals(77), fx, jitters=(1e-9, 1e-8, 1e-8))
    gd = lmm.DeviceNormals(77)
    z, eps = gd.standard_normal(m * n).cpu().numpy(), gd.standard_normal(n * p).cpu().numpy()

    class Replay:
        def __init__(self, bufs): self.bufs = list(bufs)
        def standard_normal(self, k): v = self.bufs.pop(0); assert len(v) == k; return v
    s_host = lmm.rand(Replay([z, eps]), fx, jitters=(1e-9, 1e-8, 1e-8))
    np.testing.assert_allclose(s_dev.cpu().numpy(), s_host, rtol=1e-12, atol=1e-13)
    S3 = lmm.rand(lmm.DeviceNormals(9), fx, 3, jitters=(1e-9, 1e-8, 1e-8))
    assert tuple(S3.shape) == (n * p, 3) and bool(torch.isfinite(S3).all())


def test_matrix_y_logpdf_and_rand_n(lmm):
    """logpdf(fx, Y::Matrix) and rand(rng, fx, N) (AbstractGPs.TestUtils primary interface): one
    factorisation per latent serves every column / sample; values equal the per-column reference answers."""
    rng = np.random.default_rng(17)
    n, p, m, N = 150, 4, 3, 5
    x = np.sort(rng.uniform(0, 10, n))
    gps = _gps(["matern32", "matern52", "se"], rng)
    U, S = _orth(rng, p, m)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    fx = f(lmm.MOInputIsotopicByOutputs(x, p), 0.1)
    Y = rng.standard_normal((n * p, N))
    got = lmm.logpdf(fx, Y)
    ref = np.array([O.oilmm_logpdf(gps, U, S, x, 0.1, Y[:, c]) for c in range(N)])
    np.testing.assert_allclose(got, ref, rtol=1e-10)
    np.testing.assert_allclose(got, [lmm.logpdf(fx, np.ascontiguousarray(Y[:, c])) for c in range(N)], rtol=1e-12)
    # IndependentMOGP matrix-Y
    fm = _to_model(lmm, gps)
    Ym = rng.standard_normal((n * m, 3))
    np.testing.assert_allclose(lmm.logpdf(fm(lmm.MOInputIsotopicByOutputs(x, m), 0.1), Ym),
                               [O.mogp_logpdf(gps, x, 0.1, Ym[:, c]) for c in range(3)], rtol=1e-10)
    # rand(rng, fx, N): same normals => column q equals the single-sample call made with the same stream position
    jit = (1e-9, 1e-6, 1e-6)
    Smp = lmm.rand(np.random.default_rng(4), fx, N, jitters=jit)
    assert Smp.shape == (n * p, N)
    g2 = np.random.default_rng(4)
    for q in range(N):
        z = g2.standard_normal(m * n); eps = g2.standard_normal(n * p)
        X = np.stack([O.gp_rand(g, x, 1e-6, z[l * n:(l + 1) * n]) for l, g in enumerate(gps)])
        ref_q = (O.orthogonal_dense(U, S) @ X).reshape(-1) + math.sqrt(0.1) * eps
        np.testing.assert_allclose(Smp[:, q], ref_q, rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("n,d", [(9, 1), (150, 1), (130, 2)])
def test_logpdf_gradient_vs_oracle(lmm, n, d):
    """Gradient of the OILMM logpdf (reference test/oilmm.jl:31-32 `gradient(logpdf, oilmmx, y_train)`)
    against the oracle's analytic gradient (itself checked against finite differences in tests/test_oracle.py)."""
    rng = np.random.default_rng(1000 + n)
    p, m = 4, 3
    x = np.sort(rng.uniform(0, 6, n)) if d == 1 else rng.uniform(0, 4, size=(d, n))
    gps = _gps(["matern52", "se", "matern32"], rng)
    U, S = _orth(rng, p, m)
    y = rng.standard_normal(n * p)
    fx = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), 0.3)
    G = lmm.logpdf_and_gradient(fx, y)
    R = O.oilmm_logpdf_grad(gps, U, S, x, 0.3, y)
    assert G["value"] == pytest.approx(R["value"], rel=1e-10)
    assert G["value"] == pytest.approx(lmm.logpdf(fx, y), rel=1e-12)
    np.testing.assert_allclose(G["y"], R["y"], rtol=1e-7, atol=1e-9)
    assert G["sigma2"] == pytest.approx(R["sigma2"], rel=1e-7)
    np.testing.assert_allclose(G["S"], R["S"], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(G["U"], R["U"], rtol=1e-7, atol=1e-8)
    for l in range(m):
        for key in ("variance", "lengthscale", "mean"):
            assert G["gps"][l][key] == pytest.approx(R["gps"][l][key], rel=1e-6, abs=1e-8), (l, key)
    # shards add up (as the all-reduce would)
    parts = [lmm.logpdf_and_gradient(lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S), shard=sh)(
        lmm.MOInputIsotopicByOutputs(x, p), 0.3), y, r == 0) for r, sh in enumerate([(0, 2), (2, 3)])]
    np.testing.assert_allclose(parts[0]["y"] + parts[1]["y"], G["y"], rtol=1e-9, atol=1e-11)
    assert parts[0]["sigma2"] + parts[1]["sigma2"] == pytest.approx(G["sigma2"], rel=1e-10)
    np.testing.assert_allclose(parts[0]["U"] + parts[1]["U"], G["U"], rtol=1e-9, atol=1e-11)


def test_wide_mixing_and_many_columns(lmm):
    """Shapes that exercise the chunked projection kernels and multi-block riders: p = 40 > 32 outputs, m = 18 > 16 latents,
    70 > 64 right-hand-side columns (rider rows spill into a second 64-row block), d = 5 inputs."""
    rng = np.random.default_rng(77)
    n, p, m, d = 90, 40, 18, 5
    x = rng.uniform(0, 3, size=(d, n))
    gps = _gps((["matern52", "se", "matern32"] * 6)[:m], rng)
    U, S = _orth(rng, p, m)
    S = np.linspace(2.0, 0.5, m)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    fx = f(lmm.MOInputIsotopicByOutputs(x, p), 0.2)
    Y = rng.standard_normal((n * p, 70))
    got = lmm.logpdf(fx, Y)
    ref = np.array([O.oilmm_logpdf(gps, U, S, x, 0.2, Y[:, c]) for c in range(70)])
    np.testing.assert_allclose(got, ref, rtol=1e-10)
    y = np.ascontiguousarray(Y[:, 0])
    post = lmm.posterior(fx, y)
    xs = rng.uniform(0, 3, size=(d, 33))
    mu, v = lmm.mean_and_var(post(lmm.MOInputIsotopicByOutputs(xs, p), 0.2))
    mo, vo = O.oilmm_mean_var(O.oilmm_posterior(gps, U, S, x, 0.2, y), U, S, xs, 0.2)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-10); np.testing.assert_allclose(v, vo, rtol=1e-9)


def test_deterministic_mode_is_bitwise_reproducible():
    """LMM_DETERMINISTIC=1 disables the split-K f64 atomics: two runs give identical bits (fresh processes)."""
    import subprocess, sys, os
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, lmm_amd; from oracle import lmm_oracle as O; lmm_amd.init(0);"
            "P = O.synthetic_problem(3, 5, 2500, 'matern52', True, seed=9);"
            "f = lmm_amd.ILMM(lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(3)]), lmm_amd.Orthogonal(P['U'], P['S']));"
            "print(repr(lmm_amd.logpdf(f(lmm_amd.MOInputIsotopicByOutputs(P['x'], 5), 0.1), P['y'])))") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = [subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LMM_DETERMINISTIC="1"), capture_output=True, text=True).stdout.strip().splitlines()[-1]
            for _ in range(2)]
    assert outs[0] == outs[1] and "-" in outs[0]
    P = O.synthetic_problem(3, 5, 2500, "matern52", True, seed=9)
    assert float(outs[0]) == pytest.approx(O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"]), rel=1e-9)


def test_sequential_conditioning(lmm):
    """posterior(po(x2, s2), y2) (AbstractGPs.TestUtils exercises `posterior` on `po`: reference test/oilmm.jl:34-37): the
    posterior of the posterior equals conditioning the prior on both data sets with their own noise levels."""
    rng = np.random.default_rng(31)
    n1, n2, p, m = 70, 45, 4, 3
    x1, x2 = np.sort(rng.uniform(0, 8, n1)), np.sort(rng.uniform(0, 8, n2))
    gps = _gps(["matern52", "se", "matern32"], rng)
    U, S = _orth(rng, p, m)
    y1, y2 = rng.standard_normal(n1 * p), rng.standard_normal(n2 * p)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    po = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x1, p), 0.1), y1)
    po2 = lmm.posterior(po(lmm.MOInputIsotopicByOutputs(x2, p), 0.3), y2)        # different noise for the second batch
    ro = O.oilmm_posterior(O.oilmm_posterior(gps, U, S, x1, 0.1, y1), U, S, x2, 0.3, y2)
    xs = np.linspace(0, 8, 37)
    mu, v = lmm.mean_and_var(po2(lmm.MOInputIsotopicByOutputs(xs, p), 0.2))
    mo, vo = O.oilmm_mean_var(ro, U, S, xs, 0.2)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-10); np.testing.assert_allclose(v, vo, rtol=1e-8)
    ys = rng.standard_normal(37 * p)
    assert lmm.logpdf(po2(lmm.MOInputIsotopicByOutputs(xs, p), 0.2), ys) == pytest.approx(O.oilmm_logpdf(ro, U, S, xs, 0.2, ys), rel=1e-8)
    # the first posterior is untouched
    mu1, _ = lmm.mean_and_var(po(lmm.MOInputIsotopicByOutputs(xs, p), 0.2))
    np.testing.assert_allclose(mu1, O.oilmm_mean_var(O.oilmm_posterior(gps, U, S, x1, 0.1, y1), U, S, xs, 0.2)[0], rtol=1e-8, atol=1e-10)
    # IndependentMOGP
    fm = _to_model(lmm, gps)
    ym1, ym2 = rng.standard_normal(n1 * m), rng.standard_normal(n2 * m)
    pm2 = lmm.posterior(lmm.posterior(fm(lmm.MOInputIsotopicByOutputs(x1, m), 0.1), ym1)(lmm.MOInputIsotopicByOutputs(x2, m), 0.1), ym2)
    mu, v = lmm.mean_and_var(pm2(lmm.MOInputIsotopicByOutputs(xs, m), 0.1))
    rm = O.mogp_posterior(O.mogp_posterior(gps, x1, 0.1, ym1), x2, 0.1, ym2)
    mo, vo = O.mogp_mean_var(rm, xs)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-10); np.testing.assert_allclose(v, vo + 0.1, rtol=1e-8)


def test_mogp_gradient(lmm):
    """gradient(logpdf, fx, y_train) on an IndependentMOGP (reference test/independent_mogp.jl:65-66) vs central finite
    differences of the oracle."""
    rng = np.random.default_rng(55)
    n, m = 60, 2
    x = np.sort(rng.uniform(0, 5, n))
    gps = [{"kind": "matern32", "variance": 1.1, "lengthscale": 0.9, "mean": 2.0}, {"kind": "se", "variance": 0.8, "lengthscale": 1.3, "mean": -1.0}]
    y = rng.standard_normal(n * m) + np.repeat([2.0, -1.0], n)
    G = lmm.logpdf_and_gradient(_to_model(lmm, gps)(lmm.MOInputIsotopicByOutputs(x, m), 0.2), y)
    assert G["value"] == pytest.approx(O.mogp_logpdf(gps, x, 0.2, y), rel=1e-11)
    h = 1e-6
    fd = lambda fun: (fun(h) - fun(-h)) / (2 * h)
    assert G["sigma2"] == pytest.approx(fd(lambda t: O.mogp_logpdf(gps, x, 0.2 + t, y)), rel=1e-6)
    for k in (0, 59, 100):
        e = np.zeros(n * m); e[k] = 1.0
        assert G["y"][k] == pytest.approx(fd(lambda t: O.mogp_logpdf(gps, x, 0.2, y + t * e)), rel=1e-6, abs=1e-7)
    for l in range(m):
        for key in ("variance", "lengthscale", "mean"):
            def f(t, l=l, key=key):
                g2 = [dict(g) for g in gps]; g2[l][key] += t
                return O.mogp_logpdf(g2, x, 0.2, y)
            assert G["gps"][l][key] == pytest.approx(fd(f), rel=1e-5, abs=1e-7)


def test_rand_matches_oracle_given_normals(lmm):
    """Same standard normals in the reference's draw order => same sample (reference src/oilmm.jl:40-54)."""
    rng = np.random.default_rng(11)
    n, p, m = 40, 4, 2
    x = np.sort(rng.uniform(0, 10, n))
    gps = _gps(["matern32", "matern52"], rng)
    U, S = _orth(rng, p, m)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    fx = f(lmm.MOInputIsotopicByOutputs(x, p), 0.1)
    jit = (1e-9, 1e-6, 1e-6)            # widened latent jitter: 1e-18 is not PD in Float64 for smooth kernels
    got = lmm.rand(np.random.default_rng(99), fx, jitters=jit)
    g2 = np.random.default_rng(99); z = g2.standard_normal(m * n); eps = g2.standard_normal(n * p)
    X = np.stack([O.gp_rand(g, x, 1e-6, z[l * n:(l + 1) * n]) for l, g in enumerate(gps)])
    ref = (O.orthogonal_dense(U, S) @ X).reshape(-1) + math.sqrt(0.1) * eps
    np.testing.assert_allclose(got, ref, rtol=1e-7, atol=1e-8)
    # posterior sample, dense-H mixing
    y = rng.standard_normal(n * p)
    post = lmm.posterior(fx, y)
    xs = x[:16] + 0.05
    s = lmm.rand(np.random.default_rng(5), post(lmm.MOInputIsotopicByOutputs(xs, p), 0.1), jitters=jit)
    g2 = np.random.default_rng(5); z = g2.standard_normal(m * 16); eps = g2.standard_normal(16 * p)
    po = O.oilmm_posterior(gps, U, S, x, 0.1, y)
    X = np.stack([O.gp_rand(g, xs, 1e-6, z[l * 16:(l + 1) * 16]) for l, g in enumerate(po)])
    ref = (O.orthogonal_dense(U, S) @ X).reshape(-1) + math.sqrt(0.1) * eps
    np.testing.assert_allclose(s, ref, rtol=1e-6, atol=1e-8)


def test_sampling_consistency(lmm):
    """test/test_utils.jl:41-48: sample from the prior at tiny noise, condition, recover the sample."""
    rng = np.random.default_rng(3)
    x = np.linspace(0, 10, 5)[:3]
    gps = _gps(["matern32", "matern52"])
    U, S = _orth(rng, 3, 2)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(U, S))
    xin = lmm.MOInputIsotopicByOutputs(x, 3)
    jit = (1e-9, 1e-12, 1e-12)
    # sample noise-free-ish from the model so y lies (almost) in the column space of H
    y = lmm.rand(np.random.default_rng(1), f(xin, 1e-6), jitters=jit)
    post = lmm.posterior(f(xin, 1e-6), y)
    mu, v = lmm.mean_and_var(post(xin, 1e-18))
    # posterior mean reproduces the projection of y onto span(H); variance collapses
    Y = y.reshape(3, 3)
    proj = (U @ (U.T @ Y)).reshape(-1)
    np.testing.assert_allclose(mu, proj, rtol=1e-2, atol=1e-2)
    assert np.all(np.abs(v) < 1e-2)


def test_errors(lmm):
    rng = np.random.default_rng(0)
    U, S = _orth(rng, 3, 2)
    f = lmm.ILMM(_to_model(lmm, _gps(["se", "se"])), lmm.Orthogonal(U, S))
    with pytest.raises(RuntimeError, match="out dim of x != out dim of f"):        # src/ilmm.jl:52
        lmm.logpdf(f(lmm.MOInputIsotopicByOutputs(np.arange(4.0), 4), 0.1), np.zeros(16))
    with pytest.raises(ValueError, match="not an orthogonal matrix"):             # src/orthogonal_matrix.jl:22
        lmm.Orthogonal(rng.uniform(size=(5, 3)), np.ones(3))
    # duplicate inputs + (numerically) zero noise: Cholesky must report the failing pivot, not return NaN
    x = np.array([0.0, 1.0, 1.0, 2.0])
    fbad = lmm.ILMM(_to_model(lmm, _gps(["se"])), lmm.Orthogonal(np.array([[1.0], [0.0]]), np.array([1.0])))
    with pytest.raises(lmm.PosDefException) as ei:
        lmm.logpdf(fbad(lmm.MOInputIsotopicByOutputs(x, 2), 1e-300), np.zeros(8))
    assert ei.value.latent == 0 and ei.value.info == 3


def test_full_size_properties(lmm):
    """Size-independent properties at a size the oracle cannot reach quickly (n = 4096):
    (i) shards add up; (ii) logpdf is quadratic in y: l(a*y) - l(0) = a^2 (l(y) - l(0)); (iii) mogp == oilmm with U = I."""
    import torch
    n, m, p = 4096, 2, 2
    x = np.arange(n) * (20.0 / 575.0)
    gps = _gps(["matern52"] * m)
    rng = np.random.default_rng(8)
    y = rng.standard_normal(n * p)
    f = lmm.ILMM(_to_model(lmm, gps), lmm.Orthogonal(np.eye(2), np.ones(2)))
    xin = lmm.MOInputIsotopicByOutputs(x, p)
    l1 = lmm.logpdf(f(xin, 0.1), y); l0 = lmm.logpdf(f(xin, 0.1), np.zeros(n * p)); l2 = lmm.logpdf(f(xin, 0.1), 2.0 * y)
    assert (l2 - l0) == pytest.approx(4.0 * (l1 - l0), rel=1e-10)
    assert lmm.logpdf(_to_model(lmm, gps)(xin, 0.1), y) == pytest.approx(l1, rel=1e-12)
    # against LAPACK on one latent (scipy on the host: 4096^3/3 flops, seconds)
    ref = O.gp_logpdf(gps[0], x, 0.1, y[:n])
    one = lmm.logpdf(lmm.independent_mogp([lmm.GP(lmm.Matern52Kernel())])(lmm.MOInputIsotopicByOutputs(x, 1), 0.1), y[:n])
    assert one == pytest.approx(ref, rel=1e-9)
