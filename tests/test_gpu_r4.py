"""-m gpu, round 4.
(a) Every posterior verb at test-point counts whose roundings to 64 and to 128 differ (n* = 9, 70, 130): the factor width of the
    covariance at x* is rounded to 128 columns, the cross-solve blocks used to keep rup(n*, 64) rows -- the round-3 out-of-bounds read
    (DESIGN.md "faults and aborts").  OILMM, IndependentMOGP and dense-H posteriors, values against the oracle.
(b) The allocation-extent guard of lmm_api.hip turns such a mismatch into LMM_ERR_ARG before anything is launched.
(c) The round-4 launch shapes of the factorisation against LAPACK: region launches whose row tasks mix 128- and 64-row tiles
    (region_plan), and update launches that carry the ragged last 64 rows as work items -- with and without the fused bulk tiles."""
import ctypes as C
import math

import numpy as np
import pytest

from oracle import lmm_oracle as O

pytestmark = pytest.mark.gpu

NS = [9, 70, 130]


@pytest.fixture(scope="module")
def lmm():
    import lmm_amd
    lmm_amd.init(0)
    return lmm_amd


def _model(lmm, gps):
    K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
    return lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])


def _gps(m, rng):
    kinds = ["matern52", "se", "matern32"]
    return [{"kind": kinds[l % 3], "variance": float(rng.uniform(0.6, 1.4)), "lengthscale": float(rng.uniform(0.7, 1.6)),
             "mean": float(rng.normal())} for l in range(m)]


# ---------------------------------------------------------------------------------------------------
# (a) posterior verbs at n* with rup(n*, 64) != rup(n*, 128)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ns", NS)
def test_oilmm_posterior_verbs_small_nstar(lmm, ns):
    """reference src/oilmm.jl:116-134 then :57-76 (marginals), :79-93 (logpdf of po(x*)), :40-54 (rand), src/ilmm.jl:132-147 (cov)."""
    rng = np.random.default_rng(900 + ns)
    n, m, p, s2 = 150, 3, 4, 0.15
    x, xs = np.sort(rng.uniform(0, 8, n)), np.sort(rng.uniform(0, 8, ns))
    gps = _gps(m, rng)
    U, _ = np.linalg.qr(rng.standard_normal((p, m)))
    S = np.linspace(1.4, 0.8, m)
    H = O.orthogonal_dense(U, S)
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    fx = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), s2)
    post = lmm.posterior(fx, y)
    pfx = post(lmm.MOInputIsotopicByOutputs(xs, p), s2)
    po = O.oilmm_posterior(gps, U, S, x, s2, y)
    mo, vo = O.oilmm_mean_var(po, U, S, xs, s2)
    mu, v = lmm.mean_and_var(pfx)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(v, vo, rtol=1e-8)
    assert lmm.logpdf(pfx, ys) == pytest.approx(O.oilmm_logpdf(po, U, S, xs, s2, ys), rel=1e-8)
    jit = (1e-9, 1e-6, 1e-6)
    s = lmm.rand(np.random.default_rng(5), pfx, jitters=jit)
    g2 = np.random.default_rng(5); z = g2.standard_normal(m * ns); eps = g2.standard_normal(ns * p)
    X = np.stack([O.gp_rand(g, xs, 1e-6, z[l * ns:(l + 1) * ns]) for l, g in enumerate(po)])
    np.testing.assert_allclose(s, (H @ X).reshape(-1) + math.sqrt(s2) * eps, rtol=1e-6, atol=1e-7)
    M, Cm = lmm.mean_and_cov(pfx)
    Mn, Cn = O.naive_posterior_mean_cov(gps, H, x, s2, y, xs)
    np.testing.assert_allclose(M, Mn, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(Cm, Cn + s2 * np.eye(ns * p), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("ns", NS)
def test_mogp_posterior_verbs_small_nstar(lmm, ns):
    """reference src/independent_mogp.jl:119-126 then :39-58 (marginals), :74-80 (logpdf), :83-86 (rand)."""
    rng = np.random.default_rng(1900 + ns)
    n, m, s2 = 140, 3, 0.2
    x, xs = np.sort(rng.uniform(0, 8, n)), np.sort(rng.uniform(0, 8, ns))
    gps = _gps(m, rng)
    y, ys = rng.standard_normal(n * m), rng.standard_normal(ns * m)
    ft = _model(lmm, gps)(lmm.MOInputIsotopicByOutputs(x, m), s2)
    post = lmm.posterior(ft, y)
    pfx = post(lmm.MOInputIsotopicByOutputs(xs, m), s2)
    po = O.mogp_posterior(gps, x, s2, y)
    mo, vo = O.mogp_mean_var(po, xs)
    mu, v = lmm.mean_and_var(pfx)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(v, vo + s2, rtol=1e-8)
    assert lmm.logpdf(pfx, ys) == pytest.approx(O.mogp_logpdf(po, xs, s2, ys), rel=1e-8)
    s = lmm.rand(np.random.default_rng(43), pfx)
    np.testing.assert_allclose(s, O.mogp_rand(po, xs, s2, np.random.default_rng(43).standard_normal(m * ns)), rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(lmm.cov(pfx), O.mogp_cov(po, xs) + s2 * np.eye(ns * m), rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("ns", NS)
def test_dense_posterior_verbs_small_nstar(lmm, ns):
    """reference src/ilmm.jl:184-198 then :108-129 (marginals), :150-163 (logpdf of pi(x*)), :78-87 (rand), :132-147 (cov)."""
    rng = np.random.default_rng(2900 + ns)
    n, m, p, s2 = 60, 3, 4, 0.1
    x, xs = np.sort(rng.uniform(0, 6, n)), np.sort(rng.uniform(0, 6, ns))
    gps = _gps(m, rng)
    H = rng.uniform(0.2, 1.0, size=(p, m))
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    fx = lmm.ILMM(_model(lmm, gps), H)(lmm.MOInputIsotopicByOutputs(x, p), s2)
    post = lmm.posterior(fx, y)
    pix = post(lmm.MOInputIsotopicByOutputs(xs, p), s2)
    po = O.ilmm_posterior(gps, H, x, s2, y)
    mo, vo = O.ilmm_mean_var(po, H, xs, s2)
    mu, v = lmm.mean_and_var(pix)
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(v, vo, rtol=1e-7)
    assert lmm.logpdf(pix, ys) == pytest.approx(O.ilmm_logpdf(po, H, xs, s2, ys), rel=1e-8)
    s = lmm.rand(np.random.default_rng(21), pix, jitters=(1e-9, 1e-8, 1e-8))
    g2 = np.random.default_rng(21); z = g2.standard_normal(m * ns); eps = g2.standard_normal(ns * p)
    mlat, Clat = O._ilmm_latent_joint(po, xs)
    lat = mlat + np.linalg.cholesky(Clat + 1e-8 * np.eye(m * ns)) @ z
    np.testing.assert_allclose(s, (H @ lat.reshape(m, ns)).reshape(-1) + math.sqrt(s2) * eps, rtol=1e-6, atol=1e-7)
    if ns * p <= 600:
        Mg, Cg = lmm.mean_and_cov(pix)
        Mo, Co = O.ilmm_mean_cov(po, H, xs, s2)
        np.testing.assert_allclose(Mg, Mo, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(Cg, Co, rtol=1e-7, atol=1e-9)
    # the latent PosteriorGP{IndependentMOGP} of the dense-H posterior (get_latent_gp, reference src/ilmm.jl:39 on :196-197)
    lat_fx = lmm.get_latent_gp(post)(lmm.MOInputIsotopicByOutputs(xs, m), 0.07)
    ml, vl = lmm.mean_and_var(lat_fx)
    np.testing.assert_allclose(ml, mlat, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(vl, np.diag(Clat) + 0.07, rtol=1e-7)


# ---------------------------------------------------------------------------------------------------
# (b) the extent guard
# ---------------------------------------------------------------------------------------------------
def test_extent_guard_refuses_a_short_buffer(lmm):
    """lmm_dev_extent_check (the guard every Gram / triangular-solve / Schur / factorisation launch site of lmm_api.hip goes through)
    on a POOLED block: a 64-row block passes, the 128 rows the round-3 Schur complement read from it are LMM_ERR_ARG -- nothing is
    launched either way."""
    from lmm_amd import _lib as L
    lib = lmm.load()
    rc_ok = lib.lmm_dev_extent_check(C.c_size_t(64 * 256 * 8), C.c_size_t(64), C.c_size_t(64), C.c_size_t(256))
    assert rc_ok == L.LMM_OK
    rc_bad = lib.lmm_dev_extent_check(C.c_size_t(64 * 256 * 8), C.c_size_t(128), C.c_size_t(64), C.c_size_t(256))
    assert rc_bad == L.LMM_ERR_ARG and b"extent check" in lib.lmm_last_error_string()
    rc_bad2 = lib.lmm_dev_extent_check(C.c_size_t(64 * 256 * 8), C.c_size_t(64), C.c_size_t(64), C.c_size_t(257))
    assert rc_bad2 == L.LMM_ERR_ARG
    # the library still works afterwards
    P = O.synthetic_problem(2, 3, 40, "se", True, s2=0.1, seed=3)
    f = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    got = lmm.logpdf(f(lmm.MOInputIsotopicByOutputs(P["x"], 3), 0.1), P["y"])
    assert got == pytest.approx(O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"]), rel=1e-10)


def test_posterior_handle_freed_without_gc(lmm):
    """Deleting a dense-H posterior whose latent view was taken releases the device state by reference count (no parent <-> view
    cycle): the next posterior of the same shape reuses the pooled blocks, so device memory in use does not grow."""
    import gc
    import torch
    rng = np.random.default_rng(8)
    n, m, p = 200, 3, 4
    x = np.sort(rng.uniform(0, 6, n))
    gps = _gps(m, rng)
    H = rng.uniform(0.2, 1.0, size=(p, m))
    y = rng.standard_normal(n * p)
    f = lmm.ILMM(_model(lmm, gps), H)
    gc.disable()
    try:
        used = []
        for it in range(6):
            post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x, p), 0.1), y)
            lat = lmm.get_latent_gp(post)
            lmm.mean_and_var(lat(lmm.MOInputIsotopicByOutputs(x[:9], m), 0.05))
            del post, lat
            free, total = torch.cuda.mem_get_info()
            used.append(total - free)
        assert used[-1] <= used[1] + (1 << 20), used       # steady after the first iteration's pool warm-up
    finally:
        gc.enable()


# ---------------------------------------------------------------------------------------------------
# (c) row-tile plans and in-launch ragged rows
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,m", [(1280, 8), (1792, 8), (2560, 4), (3072, 8), (4096, 8), (4096, 24)])
def test_logpdf_mid_sizes_vs_lapack(lmm, n, m):
    """OILMM logpdf (reference src/oilmm.jl:79-93) at sizes between the one-launch path and C2, against the oracle's LAPACK
    factorisations.  (1280, 8) .. (3072, 8): region launches with all-64 / mixed row tiles; (4096, 8): + K >= 1024 updates with the
    ragged rows as work items (dataflow base case); (4096, 24): the panel recursion with fused bulk tiles waiting for the strip's
    column-0 item."""
    import torch
    p = m + 2
    P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=n + m)
    fs = lmm.independent_mogp([lmm.GP(lmm.Matern52Kernel()) for _ in range(m)])
    fx = lmm.ILMM(fs, lmm.Orthogonal(P["U"], P["S"]))(lmm.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), p), 0.1)
    got = float(lmm.logpdf(fx, torch.from_numpy(P["y"]).cuda()))
    want = O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    assert got == pytest.approx(want, rel=1e-9)
    again = float(lmm.logpdf(fx, torch.from_numpy(P["y"]).cuda()))
    assert again == pytest.approx(got, rel=1e-12)      # (split-K atomics on the K >= 1024 levels: not bitwise)
