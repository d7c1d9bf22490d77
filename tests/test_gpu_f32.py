"""-m gpu: the fp32 compute mode (lmm_set_compute_dtype(LMM_F32); BASELINE configs[4] names an fp32 rand / marginals workload).
Matrices (Grams, factors, cross-solve blocks) are Float32 and the updates run on v_mfma_f32_32x32x2_f32; the oracle stays
Float64.  Stated tolerance: rtol 2e-4 on logpdf / marginals / samples at sigma2 = 0.1 and unit-scale kernels (cond(K + s I) ~ 1e3;
Float32 eps 6e-8 x condition x a sqrt(n) accumulation factor), against rtol 1e-6 in the Float64 parity mode."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import lmm_oracle as O

pytestmark = pytest.mark.gpu
RTOL32 = 2e-4


@pytest.fixture()
def lmm32():
    import lmm_amd
    lmm_amd.init(0)
    lmm_amd.set_compute_dtype("f32")
    yield lmm_amd
    lmm_amd.set_compute_dtype("f64")


def _model(lmm, gps):
    K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
    return lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])


def test_f32_gemm_and_potrf_building_blocks(lmm32):
    """The f32 MFMA tile kernel against torch (asymmetric operands: catches a swapped C/D map) and the blocked f32 Cholesky
    (diag64 in f64 registers, TRSM and updates on v_mfma_f32) against LAPACK."""
    import torch
    lib = lmm32.load()
    g = torch.Generator(device="cuda").manual_seed(1)
    for (M, N, K, lower) in [(128, 128, 64, 0), (192, 64, 16, 0), (320, 192, 208, 0), (256, 256, 128, 1), (448, 320, 64, 1), (1024, 512, 1024, 1)]:
        ldc, lda, ldb = M + 16, M + 4, N + 8
        Ct = torch.randn(N, ldc, generator=g, device="cuda", dtype=torch.float32)     # column-major: [col][row]
        At = torch.randn(K, lda, generator=g, device="cuda", dtype=torch.float32)
        Bt = torch.randn(K, ldb, generator=g, device="cuda", dtype=torch.float32)
        C0 = Ct.clone()
        torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
        rc = lib.lmm_dev_gemm_nt_sub(C.c_void_p(Ct.data_ptr()), ldc, C.c_void_p(At.data_ptr()), lda, C.c_void_p(Bt.data_ptr()), ldb, M, N, K, lower)
        assert rc == 0, lib.lmm_last_error_string()
        ref = C0[:, :M].double() - (Bt[:, :N].double().T @ At[:, :M].double())          # [col][row]
        got = Ct[:, :M].double()
        if lower:
            # tiles on/below the diagonal of the 128-blocked lower trapezoid are updated; compare where row >= col
            r = torch.arange(M, device="cuda")[None, :]; c = torch.arange(N, device="cuda")[:, None]
            mask = r >= c
            assert torch.allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4 * math.sqrt(K))
        else:
            assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4 * math.sqrt(K))
        assert torch.equal(Ct[:, M:], C0[:, M:])                                       # padding rows untouched
    rng = np.random.default_rng(3)
    for n, riders in [(64, 64), (192, 64), (448, 128), (1024, 64)]:
        X = rng.standard_normal((n, n))
        A = X @ X.T / n + 2.0 * np.eye(n)
        R = rng.standard_normal((riders, n))
        ld = n + riders
        buf = np.zeros((n, ld), dtype=np.float32)                                      # [col][row]
        buf[:, :n] = np.tril(A).T
        buf[:, n:] = R.T
        Ad = torch.from_numpy(buf).cuda()
        W = torch.zeros((n // 64) * 4096, dtype=torch.float32, device="cuda")
        info = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
        assert lib.lmm_dev_potrf(C.c_void_p(Ad.data_ptr()), ld, n, ld, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr())) == 0
        assert int(info.item()) == 0
        out = Ad.cpu().numpy().astype(np.float64)
        L = np.linalg.cholesky(A)
        np.testing.assert_allclose(np.tril(out[:, :n].T), L, rtol=2e-5, atol=2e-5)
        import scipy.linalg as sla
        np.testing.assert_allclose(out[:, n:].T, sla.solve_triangular(L, R.T, lower=True).T, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,N,K,lower", [(4096, 4096, 512, 0), (6208, 6144, 320, 1), (8256, 4096, 288, 1)])
def test_f32_wide_update_256_tiles(lmm32, M, N, K, lower):
    """The 256 x 256-tile fp32 update (gemm32w_kernel, its own translation unit: AccVGPR accumulators) takes over where its tiles fill
    the device: full tiles; a ragged last row tile (M = 6208 = 24.25 tiles) with a split-K tail (324 tiles on 256 CUs); the lower
    trapezoid with a split-K tail -- against torch in Float64."""
    import torch
    lib = lmm32.load()
    g = torch.Generator(device="cuda").manual_seed(M + K)
    ldc, lda, ldb = M + 16, M + 4, N + 8
    Ct = torch.randn(N, ldc, generator=g, device="cuda", dtype=torch.float32)
    At = torch.randn(K, lda, generator=g, device="cuda", dtype=torch.float32)
    Bt = torch.randn(K, ldb, generator=g, device="cuda", dtype=torch.float32)
    C0 = Ct.clone()
    torch.cuda.synchronize()
    rc = lib.lmm_dev_gemm_nt_sub(C.c_void_p(Ct.data_ptr()), ldc, C.c_void_p(At.data_ptr()), lda, C.c_void_p(Bt.data_ptr()), ldb, M, N, K, lower)
    assert rc == 0, lib.lmm_last_error_string()
    ref = C0[:, :M].double() - (Bt[:, :N].double().T @ At[:, :M].double())
    got = Ct[:, :M].double()
    if lower:
        r = torch.arange(M, device="cuda")[None, :]; c = torch.arange(N, device="cuda")[:, None]
        mask = r >= c
        assert torch.allclose(got[mask], ref[mask], rtol=1e-4, atol=1e-4 * math.sqrt(K))
    else:
        assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4 * math.sqrt(K))
    assert torch.equal(Ct[:, M:], C0[:, M:])


@pytest.mark.parametrize("kind,n,m,p,d", [("matern52", 700, 3, 5, 1), ("se", 1100, 2, 4, 2), ("matern32", 2100, 4, 6, 1)])
def test_f32_oilmm_logpdf_posterior_marginals_rand_vs_f64_oracle(lmm32, kind, n, m, p, d):
    lmm = lmm32
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0, n * 20.0 / 575.0, n)) if d == 1 else rng.uniform(0, 12.0, size=(d, n))
    gps = [{"kind": kind, "variance": float(rng.uniform(0.7, 1.5)), "lengthscale": float(rng.uniform(0.8, 1.5)), "mean": float(rng.normal())}
           for _ in range(m)]
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    S = np.linspace(2.0, 1.0, m)
    y = rng.standard_normal(n * p)
    f = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
    xin = lmm.MOInputIsotopicByOutputs(x, p)
    assert lmm.get_compute_dtype() == "f32"
    got = lmm.logpdf(f(xin, 0.1), y)
    assert got == pytest.approx(O.oilmm_logpdf(gps, U, S, x, 0.1, y), rel=RTOL32)
    post = lmm.posterior(f(xin, 0.1), y)
    xs = (x[:96] + 0.013) if d == 1 else (x[:, :96] + 0.013)
    xsin = lmm.MOInputIsotopicByOutputs(xs, p)
    mu, v = lmm.mean_and_var(post(xsin, 0.1))
    po = O.oilmm_posterior(gps, U, S, x, 0.1, y)
    mo, vo = O.oilmm_mean_var(po, U, S, xs, 0.1)
    np.testing.assert_allclose(mu, mo, rtol=RTOL32, atol=RTOL32)
    np.testing.assert_allclose(v, vo, rtol=RTOL32)
    np.testing.assert_allclose(lmm.mean(post(xsin, 0.1)), mo, rtol=RTOL32, atol=RTOL32)
    assert lmm.logpdf(post(xsin, 0.1), y[:96 * p]) == pytest.approx(O.oilmm_logpdf(po, U, S, xs, 0.1, y[:96 * p]), rel=5 * RTOL32)
    # posterior sample on the same normals (latent jitter 1e-3: Float32 needs a real jitter)
    jit = (1e-9, 1e-3, 1e-3)
    s = lmm.rand(np.random.default_rng(5), post(xsin, 0.1), jitters=jit)
    g2 = np.random.default_rng(5); z = g2.standard_normal(m * 96); eps = g2.standard_normal(96 * p)
    X = np.stack([O.gp_rand(g, xs, 1e-3, z[l * 96:(l + 1) * 96]) for l, g in enumerate(po)])
    np.testing.assert_allclose(s, (O.orthogonal_dense(U, S) @ X).reshape(-1) + math.sqrt(0.1) * eps, rtol=20 * RTOL32, atol=20 * RTOL32)
    # prior sample
    s0 = lmm.rand(np.random.default_rng(6), f(xsin, 0.1), jitters=jit)
    g2 = np.random.default_rng(6); z = g2.standard_normal(m * 96); eps = g2.standard_normal(96 * p)
    X = np.stack([O.gp_rand(g, xs, 1e-3, z[l * 96:(l + 1) * 96]) for l, g in enumerate(gps)])
    np.testing.assert_allclose(s0, (O.orthogonal_dense(U, S) @ X).reshape(-1) + math.sqrt(0.1) * eps, rtol=20 * RTOL32, atol=20 * RTOL32)


def test_f32_dense_ilmm_logpdf(lmm32):
    """logpdf of a dense-H ILMM with DIFFERENT latent kernels (the reference's single (mn) x (mn) factorisation, src/ilmm.jl:150-163) and
    its matrix-Y form in the fp32 compute mode (Float32 (mn) x (mn) matrix, Float64 projections and reductions) against the Float64
    oracle.  Tolerance: RTOL32 on the value (the quadratic form and the log-determinant are sums over mn Float32 pivots)."""
    lmm = lmm32
    rng = np.random.default_rng(8)
    n, p, m = 180, 5, 3
    x = np.sort(rng.uniform(0, 8, n))
    gps = [{"kind": k, "variance": 0.8 + 0.2 * l, "lengthscale": 0.9 + 0.3 * l, "mean": 0.1 * l} for l, k in enumerate(["se", "matern32", "matern52"])]
    H = rng.standard_normal((p, m))
    y = rng.standard_normal(n * p)
    fx = lmm.ILMM(_model(lmm, gps), H)(lmm.MOInputIsotopicByOutputs(x, p), 0.3)
    ref = O.ilmm_logpdf(gps, H, x, 0.3, y)
    assert lmm.logpdf(fx, y) == pytest.approx(ref, rel=RTOL32)
    Y = rng.standard_normal((n * p, 3))
    got = lmm.logpdf(fx, Y)
    np.testing.assert_allclose(got, [O.ilmm_logpdf(gps, H, x, 0.3, Y[:, c]) for c in range(3)], rtol=RTOL32)


@pytest.mark.parametrize("ns", [9, 70])
def test_f32_dense_posterior_vs_f64_oracle(lmm32, ns):
    """Round 4: the dense-H ILMM posterior (reference src/ilmm.jl:184-198, generic over T <: Real) in the fp32 compute mode --
    posterior, mean_and_var, logpdf of pi(x*), rand, mean_and_cov and the latent view -- against the Float64 oracle at the mode's
    stated tolerance (RTOL32 at sigma2 = 0.1, unit-scale kernels).  The means take the rider form mu + R (L^-1 delta): the shortcut
    mu + K(x*, x) alpha cancels over weights whose Float32-factor error is amplified by the condition number."""
    lmm = lmm32
    rng = np.random.default_rng(4400 + ns)
    n, m, p, s2 = 150, 3, 4, 0.1
    x, xs = np.sort(rng.uniform(0, 6, n)), np.sort(rng.uniform(0, 6, ns))
    gps = [{"kind": k, "variance": float(rng.uniform(0.7, 1.3)), "lengthscale": float(rng.uniform(0.8, 1.5)), "mean": float(rng.normal())}
           for k in ["se", "matern32", "matern52"]]
    H = rng.uniform(0.2, 1.0, size=(p, m))
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    fx = lmm.ILMM(_model(lmm, gps), H)(lmm.MOInputIsotopicByOutputs(x, p), s2)
    post = lmm.posterior(fx, y)
    pix = post(lmm.MOInputIsotopicByOutputs(xs, p), s2)
    po = O.ilmm_posterior(gps, H, x, s2, y)
    mo, vo = O.ilmm_mean_var(po, H, xs, s2)
    mu, v = lmm.mean_and_var(pix)
    scale = np.abs(mo).max()
    np.testing.assert_allclose(mu, mo, rtol=RTOL32, atol=RTOL32 * scale)
    np.testing.assert_allclose(v, vo, rtol=5 * RTOL32)
    assert lmm.logpdf(pix, ys) == pytest.approx(O.ilmm_logpdf(po, H, xs, s2, ys), rel=5 * RTOL32)
    jit = (1e-9, 1e-4, 1e-4)
    smp = lmm.rand(np.random.default_rng(21), pix, jitters=jit)
    g2 = np.random.default_rng(21); z = g2.standard_normal(m * ns); eps = g2.standard_normal(ns * p)
    mlat, Clat = O._ilmm_latent_joint(po, xs)
    lat = mlat + np.linalg.cholesky(Clat + 1e-4 * np.eye(m * ns)) @ z
    ref = (H @ lat.reshape(m, ns)).reshape(-1) + math.sqrt(s2) * eps
    np.testing.assert_allclose(smp, ref, rtol=20 * RTOL32, atol=20 * RTOL32 * np.abs(ref).max())
    Mg, Cg = lmm.mean_and_cov(pix)
    Mo, Co = O.ilmm_mean_cov(po, H, xs, s2)
    np.testing.assert_allclose(Mg, Mo, rtol=RTOL32, atol=RTOL32 * scale)
    np.testing.assert_allclose(Cg, Co, rtol=5 * RTOL32, atol=5 * RTOL32 * np.abs(Co).max())
    lat_fx = lmm.get_latent_gp(post)(lmm.MOInputIsotopicByOutputs(xs, m), 0.07)
    ml, vl = lmm.mean_and_var(lat_fx)
    np.testing.assert_allclose(ml, mlat, rtol=RTOL32, atol=RTOL32 * np.abs(mlat).max())
    np.testing.assert_allclose(vl, np.diag(Clat) + 0.07, rtol=5 * RTOL32)
    # sequential conditioning on the dense-H posterior (posterior(pi(x2, s2), y2): src/ilmm.jl:184-198 again)
    x2 = np.sort(rng.uniform(0, 6, 40)); y2 = rng.standard_normal(40 * p)
    post2 = lmm.posterior(post(lmm.MOInputIsotopicByOutputs(x2, p), 0.2), y2)
    mo2, vo2 = O.ilmm_mean_var(O.ilmm_posterior_condition(po, H, x2, 0.2, y2), H, xs, s2)
    mu2, v2 = lmm.mean_and_var(post2(lmm.MOInputIsotopicByOutputs(xs, p), s2))
    np.testing.assert_allclose(mu2, mo2, rtol=RTOL32, atol=RTOL32 * np.abs(mo2).max())
    np.testing.assert_allclose(v2, vo2, rtol=5 * RTOL32)


def test_f32_mode_boundaries(lmm32):
    """Handles remember their dtype; unsupported paths say so instead of computing in the wrong precision; switching back to
    Float64 restores the parity mode bit for bit."""
    lmm = lmm32
    from lmm_amd import _lib as L
    P = O.synthetic_problem(3, 5, 200, "se", True, s2=0.1, seed=0)
    f = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    xin = lmm.MOInputIsotopicByOutputs(P["x"], 5)
    post32 = lmm.posterior(f(xin, 0.1), P["y"])
    v32 = lmm.logpdf(f(xin, 0.1), P["y"])
    g32 = lmm.logpdf_and_gradient(f(xin, 0.1), P["y"])            # round 3: served in fp32 (values: test_f32_oilmm_logpdf_gradient)
    assert g32["value"] == pytest.approx(v32, rel=1e-6)
    # (round 4: the full covariance and the dense-H gradients are served in fp32 too: test_f32_full_covariance,
    # test_f32_dense_gradients; nothing on the path returns LMM_ERR_UNSUPPORTED for the compute dtype any more)
    pd32 = lmm.posterior(lmm.ILMM(_model(lmm, P["gps"]), np.abs(P["U"]) + 0.1)(xin, 0.1), P["y"])
    lmm.set_compute_dtype("f64")
    v64 = lmm.logpdf(f(xin, 0.1), P["y"])
    ref = O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    assert v64 == pytest.approx(ref, rel=1e-12)
    assert v32 == pytest.approx(ref, rel=RTOL32) and v32 != v64
    with pytest.raises(ValueError, match="other compute dtype"):
        lmm.mean_and_var(post32(lmm.MOInputIsotopicByOutputs(P["x"][:8], 5), 0.1))
    with pytest.raises(ValueError, match="other compute dtype"):   # ... and so does a dense-H handle
        lmm.mean_and_var(pd32(lmm.MOInputIsotopicByOutputs(P["x"][:8], 5), 0.1))
    lmm.set_compute_dtype("f32")
    mu, _ = lmm.mean_and_var(post32(lmm.MOInputIsotopicByOutputs(P["x"][:8], 5), 0.1))
    assert np.all(np.isfinite(mu))
    mu, _ = lmm.mean_and_var(pd32(lmm.MOInputIsotopicByOutputs(P["x"][:8], 5), 0.1))
    assert np.all(np.isfinite(mu))


def test_f32_c4_share_full_size(lmm32):
    """configs[4] AS NAMED: OILMM rand / marginals, p = 256 outputs, 128 latents, n = 32768, fp32 -- one lock-step batch (8 of one
    GPU's 16 latents).  Checked against the Float64 HIP path on latent 0 (itself checked against host LAPACK in
    test_gpu_full_size.py::test_c4_shape_rand_marginals) at the stated fp32 tolerance, plus the size-independent properties."""
    import torch
    lmm = lmm32
    from lmm_amd import _lib as L
    n, p, m, nl, ns = 32768, 256, 128, 8, 1024
    P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
    fs, H = _model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"])
    xd = torch.from_numpy(P["x"]).cuda(); yd = torch.from_numpy(P["y"]).cuda()
    xin = lmm.MOInputIsotopicByOutputs(xd, p)
    xs = P["x"][:ns] + 0.01
    xsin = lmm.MOInputIsotopicByOutputs(xs, p)
    Hs = O.orthogonal_dense(P["U"], P["S"])[:, :nl]
    post = lmm.posterior(lmm.ILMM(fs, H, shard=(0, nl))(xin, 0.1), yd)
    ml, vl = np.empty(nl * ns), np.empty(nl * ns)
    L.check(lmm.load().lmm_latent_marginals(post.f._post.ptr, None, nl, L.Arr(xs).ptr, 1, ns, L.Arr(ml, True).ptr, L.Arr(vl, True).ptr))
    mup, vp = lmm.mean_and_var(post(xsin, 0.1))
    # two calls = two Float32 cross-solves whose split-K tails add with f32 atomics: equal to rounding, not bitwise
    np.testing.assert_allclose(mup, (Hs @ ml.reshape(nl, ns)).reshape(-1), rtol=RTOL32, atol=RTOL32)
    assert np.all(vl > 0) and np.all(vl < 1.0 + 1e-6)
    jit = (1e-9, 1e-3, 1e-3)
    s = lmm.rand(np.random.default_rng(0), post(xsin, 0.1), jitters=jit, add_noise=False)
    R = (s - mup).reshape(p, ns)
    resid = R - Hs @ np.linalg.lstsq(Hs, R, rcond=None)[0]
    assert np.abs(resid).max() < 1e-9 * max(1.0, np.abs(R).max())
    lp32 = lmm.logpdf(lmm.ILMM(fs, H, shard=(0, 1))(xin, 0.1), yd, False)
    del post
    # the same latent 0 in Float64
    lmm.set_compute_dtype("f64")
    post64 = lmm.posterior(lmm.ILMM(fs, H, shard=(0, 1))(xin, 0.1), yd)
    m64, v64 = np.empty(ns), np.empty(ns)
    L.check(lmm.load().lmm_latent_marginals(post64.f._post.ptr, None, 1, L.Arr(xs).ptr, 1, ns, L.Arr(m64, True).ptr, L.Arr(v64, True).ptr))
    lp64 = lmm.logpdf(lmm.ILMM(fs, H, shard=(0, 1))(xin, 0.1), yd, False)
    lmm.set_compute_dtype("f32")
    np.testing.assert_allclose(ml[:ns], m64, rtol=RTOL32, atol=RTOL32)
    np.testing.assert_allclose(vl[:ns], v64, rtol=10 * RTOL32)
    assert lp32 == pytest.approx(lp64, rel=RTOL32)


def test_f32_oilmm_logpdf_gradient(lmm32):
    """Round 3: value and gradient of logpdf(fx::FiniteGP{<:OILMM}, y) in the fp32 compute mode (Float32 factor, triangular inverse
    and K^-1 = L^-T L^-1 on v_mfma_f32; every reduction in Float64) against the Float64 oracle's analytic gradient -- the OILMM
    training loop is where fp32 pays.  STATED TOLERANCE (sigma2 = 0.1, unit-scale kernels, n ~ 10^3; include/lmm_hip.h): value rtol
    2e-4; d/dy and d/dU within 1e-4 of their largest component; d/dsigma2 rtol 1e-4; d/dS and the kernel parameters rtol 2e-3 + 1e-2
    absolute (differences tr(K^-1 dK) - a' dK a of O(n) terms).  Observed on the box: 1e-5 / 1e-6 / 4e-7 / 1e-5 ... 7e-4.  Also the
    predictive logpdf's gradient (two noise blocks through the same core) against the Float64 HIP path."""
    lmm = lmm32
    rng = np.random.default_rng(17)
    n, p, m = 900, 5, 3
    x = np.sort(rng.uniform(0, 30, n))
    gps = [{"kind": k, "variance": float(rng.uniform(0.7, 1.5)), "lengthscale": float(rng.uniform(0.7, 1.5)), "mean": float(rng.normal())}
           for k in ["se", "matern32", "matern52"]]
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    S = np.linspace(2, 1, m)
    y = rng.standard_normal(n * p)
    fx = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), 0.1)
    G = lmm.logpdf_and_gradient(fx, y)
    R = O.oilmm_logpdf_grad(gps, U, S, x, 0.1, y)
    assert G["value"] == pytest.approx(R["value"], rel=RTOL32)
    np.testing.assert_allclose(G["y"], R["y"], rtol=0, atol=1e-4 * np.abs(R["y"]).max())
    assert G["sigma2"] == pytest.approx(R["sigma2"], rel=1e-4)
    np.testing.assert_allclose(G["S"], R["S"], rtol=2e-3, atol=1e-2)
    np.testing.assert_allclose(G["U"], R["U"], rtol=0, atol=1e-4 * np.abs(R["U"]).max())
    for l in range(m):
        for key in ("variance", "lengthscale", "mean"):
            assert G["gps"][l][key] == pytest.approx(R["gps"][l][key], rel=2e-3, abs=1e-2), (l, key)
    # the same answers as the Float64 HIP path, at that tolerance
    lmm.set_compute_dtype("f64")
    G64 = lmm.logpdf_and_gradient(fx, y)
    lmm.set_compute_dtype("f32")
    np.testing.assert_allclose(G["y"], G64["y"], rtol=0, atol=1e-4 * np.abs(G64["y"]).max())
    # predictive logpdf gradient (posterior built in fp32, two noise blocks)
    xs = np.sort(rng.uniform(0, 30, 150)); ys = rng.standard_normal(150 * p)
    fxs = lmm.posterior(fx, y)(lmm.MOInputIsotopicByOutputs(xs, p), 0.2)
    Gp = lmm.logpdf_and_gradient(fxs, ys)
    lmm.set_compute_dtype("f64")
    fxs64 = lmm.posterior(fx, y)(lmm.MOInputIsotopicByOutputs(xs, p), 0.2)
    Gp64 = lmm.logpdf_and_gradient(fxs64, ys)
    lmm.set_compute_dtype("f32")
    assert Gp["value"] == pytest.approx(Gp64["value"], rel=5e-4)
    np.testing.assert_allclose(Gp["y"], Gp64["y"], rtol=0, atol=5e-3 * np.abs(Gp64["y"]).max())
    assert Gp["sigma2"] == pytest.approx(Gp64["sigma2"], rel=2e-2, abs=2e-2)
    print("f32 gradient errors:", {k: (float(np.max(np.abs(np.asarray(G[k]) - np.asarray(R[k])))), float(np.max(np.abs(np.asarray(R[k]))))) for k in ("y", "S", "U")},
          "sigma2", G["sigma2"], R["sigma2"], "gps", [(G["gps"][l], R["gps"][l]) for l in range(m)])


def test_f32_full_covariance(lmm32):
    """Round 4: cov / mean_and_cov of independent latents (reference src/ilmm.jl:132-147, src/independent_mogp.jl:60-63) in the fp32
    compute mode -- prior OILMM, posterior OILMM and the IndependentMOGP -- against the Float64 oracle.  The latent covariances at x*
    (Gram minus the Schur complement R R' of a Float32 cross-solve) are Float32 matrices, the mixing sum_l H H' C_l is Float64.
    STATED TOLERANCE (sigma2 = 0.1, unit-scale kernels, n = 600): means within 2e-4 of the largest, covariance entries within 5e-5 of
    the largest (observed on the box: 2.8e-5 and 3.5e-6; tools/f32_cov_errors.py)."""
    lmm = lmm32
    rng = np.random.default_rng(23)
    n, ns, p, m, s2 = 600, 70, 4, 3, 0.1
    x, xs = np.sort(rng.uniform(0, 20, n)), np.sort(rng.uniform(0, 20, ns))
    gps = [{"kind": k, "variance": float(rng.uniform(0.7, 1.4)), "lengthscale": float(rng.uniform(0.7, 1.5)), "mean": float(rng.normal())}
           for k in ["matern52", "se", "matern32"]]
    U, _ = np.linalg.qr(rng.standard_normal((p, m)))
    S = np.linspace(1.5, 0.8, m)
    H = O.orthogonal_dense(U, S)
    y = rng.standard_normal(n * p)
    f = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
    xsin = lmm.MOInputIsotopicByOutputs(xs, p)
    # prior
    M0, C0 = lmm.mean_and_cov(f(xsin, s2))
    # posterior against the naive dense GP of the oracle
    post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x, p), s2), y)
    M1, C1 = lmm.mean_and_cov(post(xsin, s2))
    Mo, Co = O.naive_posterior_mean_cov(gps, H, x, s2, y, xs)
    Co = Co + s2 * np.eye(ns * p)
    np.testing.assert_allclose(M1, Mo, rtol=0, atol=2e-4 * np.abs(Mo).max())
    np.testing.assert_allclose(C1, Co, rtol=0, atol=5e-5 * np.abs(Co).max())
    # the prior: against the Float64 HIP path (itself pinned by tests/test_gpu_parity.py::test_cov_and_mean_and_cov)
    lmm.set_compute_dtype("f64")
    M64, C64 = lmm.mean_and_cov(f(xsin, s2))
    lmm.set_compute_dtype("f32")
    np.testing.assert_allclose(M0, M64, rtol=0, atol=1e-6 * max(1.0, np.abs(M64).max()))
    np.testing.assert_allclose(C0, C64, rtol=0, atol=5e-5 * np.abs(C64).max())
    assert np.allclose(C1, C1.T, atol=1e-12)
    # IndependentMOGP posterior covariance (block diagonal)
    ft = _model(lmm, gps)(lmm.MOInputIsotopicByOutputs(x, m), s2)
    pm = lmm.posterior(ft, y[:n * m])
    Cm = lmm.cov(pm(lmm.MOInputIsotopicByOutputs(xs, m), s2))
    po = O.mogp_posterior(gps, x, s2, y[:n * m])
    Cr = O.mogp_cov(po, xs) + s2 * np.eye(ns * m)
    np.testing.assert_allclose(Cm, Cr, rtol=0, atol=5e-5 * np.abs(Cr).max())


def test_f32_dense_gradients(lmm32):
    """Round 4: value and gradient of the dense-H ILMM logpdf (reference src/ilmm.jl:150-181 differentiated; test/ilmm.jl:31) and of
    the dense-H posterior's predictive logpdf (test/ilmm.jl:32) in the fp32 compute mode: Float32 (mn) x (mn) factor, triangular
    inverse and explicit inverse on v_mfma_f32, every reduction and the chain rule through project(H, sigma2) in Float64.  Against
    the Float64 HIP path (pinned to finite differences of the oracle in tests/test_gpu_parity.py).  STATED TOLERANCE (sigma2 = 0.1,
    m n ~ 10^3): value rtol 2e-5; d/dy within 1e-4, d/dy_train and d/dH within 5e-4 of their largest component; d/dsigma2 rtol 1e-4;
    kernel parameters rtol 2e-3 + 1e-2 absolute (differences of O(mn) terms).  Observed on the box: 6e-7 / 5e-6 / 4e-5 / 4e-5 / 5e-7 /
    <= 1e-3 absolute (tools/f32_dense_grad_errors.py)."""
    lmm = lmm32
    rng = np.random.default_rng(31)
    n, ns, p, m, s2 = 300, 60, 4, 3, 0.1
    x, xs = np.sort(rng.uniform(0, 12, n)), np.sort(rng.uniform(0, 12, ns))
    gps = [{"kind": k, "variance": float(rng.uniform(0.7, 1.4)), "lengthscale": float(rng.uniform(0.7, 1.5)), "mean": float(rng.normal())}
           for k in ["se", "matern52", "matern32"]]
    H = rng.uniform(0.2, 1.0, size=(p, m))
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    fx = lmm.ILMM(_model(lmm, gps), H)(lmm.MOInputIsotopicByOutputs(x, p), s2)

    def both(fn):
        g32 = fn()
        lmm.set_compute_dtype("f64")
        try:
            g64 = fn()
        finally:
            lmm.set_compute_dtype("f32")
        return g32, g64

    def compare(G, R, ykeys):
        assert G["value"] == pytest.approx(R["value"], rel=2e-5)
        for k in ykeys:
            np.testing.assert_allclose(G[k], R[k], rtol=0, atol=(1e-4 if k == "y" else 5e-4) * np.abs(R[k]).max())
        np.testing.assert_allclose(G["H"], R["H"], rtol=0, atol=5e-4 * np.abs(R["H"]).max())
        assert G["sigma2"] == pytest.approx(R["sigma2"], rel=1e-4)
        for l in range(m):
            for key in ("variance", "lengthscale", "mean"):
                assert G["gps"][l][key] == pytest.approx(R["gps"][l][key], rel=2e-3, abs=1e-2), (l, key)

    G, R = both(lambda: lmm.logpdf_and_gradient(fx, y))
    compare(G, R, ["y"])
    assert G["value"] != R["value"]                      # really computed in the other precision
    Gp, Rp = both(lambda: lmm.logpdf_and_gradient(lmm.posterior(fx, y)(lmm.MOInputIsotopicByOutputs(xs, p), 0.2), ys))
    compare(Gp, Rp, ["y", "y_train"])
