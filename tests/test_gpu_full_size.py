"""-m gpu: BASELINE.json configs at (one GPU's share of) their full sizes, checked through size-independent properties and,
where a host LAPACK run stays within seconds, against the oracle on ONE latent.  Float64 here; the variants configs[3] / [4]
name are tested as named elsewhere: the bf16 MFMA projection in tests/test_gpu_r3.py::test_c3_bf16_projection_full_size, the fp32
compute mode in tests/test_gpu_f32.py::test_f32_c4_share_full_size."""
import numpy as np
import pytest

from oracle import lmm_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lmm():
    import lmm_amd
    lmm_amd.init(0)
    return lmm_amd


def _model(lmm, gps):
    K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
    return lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])


def test_c1_dense_ilmm_full_size(lmm):
    """configs[1]: ILMM, dense H 16x8, 8 SE latents, n = 4096.  The reference's single (mn) x (mn) = 32768^2 factorisation
    and the decoupled shortcut (8 independent 4096^2 ones) are two different GPU computations of the same number; one
    rotated latent is also checked against host LAPACK."""
    import torch
    from lmm_amd import model as M
    P = O.synthetic_problem(8, 16, 4096, "se", False, s2=0.1, seed=0)
    fx = lmm.ILMM(_model(lmm, P["gps"]), P["H"])(lmm.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 16), 0.1)
    y = torch.from_numpy(P["y"]).cuda()
    try:
        M.ILMM_ALLOW_DECOUPLED = False
        dense = lmm.logpdf(fx, y)
        M.ILMM_ALLOW_DECOUPLED = True
        dec = lmm.logpdf(fx, y)
    finally:
        M.ILMM_ALLOW_DECOUPLED = True
    assert dense == pytest.approx(dec, rel=1e-9)
    # host check of the whole value through the same rotation: sum_a logN(Q'(Ty) ; K + lam_a I) + regulariser
    T, ST = O.project_dense(P["H"], 0.1)
    lam, Q = np.linalg.eigh(ST)
    D = Q.T @ (T @ O.reshape_y(P["y"], 4096))
    K = O.kernelmatrix(P["gps"][0], P["x"])
    ref = sum(O.gaussian_logpdf(np.zeros(4096), K + lam[a] * np.eye(4096), D[a]) for a in range(8))
    ref += O.regulariser_ilmm(P["H"], 0.1, O.reshape_y(P["y"], 4096))
    assert dec == pytest.approx(ref, rel=1e-8)


def test_c2_share_properties(lmm):
    """configs[2] on one GPU of eight: 4 Matern52 latents of the 64x32 OILMM at n = 16384: shard additivity, exact
    quadratic dependence on y, and latent 0 against host LAPACK."""
    import torch
    P = O.synthetic_problem(32, 64, 16384, "matern52", True, s2=0.1, seed=0)
    fs, H = _model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"])
    xin = lmm.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 64)
    y = torch.from_numpy(P["y"]).cuda()
    a = lmm.logpdf(lmm.ILMM(fs, H, shard=(0, 4))(xin, 0.1), y, True)
    b = lmm.logpdf(lmm.ILMM(fs, H, shard=(0, 2))(xin, 0.1), y, True) + lmm.logpdf(lmm.ILMM(fs, H, shard=(2, 4))(xin, 0.1), y, False)
    assert a == pytest.approx(b, rel=1e-12)
    f1 = lmm.ILMM(fs, H, shard=(0, 1))(xin, 0.1)
    l1, l0, l2 = lmm.logpdf(f1, y, False), lmm.logpdf(f1, torch.zeros_like(y), False), lmm.logpdf(f1, 2.0 * y, False)
    assert (l2 - l0) == pytest.approx(4.0 * (l1 - l0), rel=1e-10)
    T, ST = O.project_orthogonal(P["U"], P["S"], 0.1)
    ref = O.gp_logpdf(P["gps"][0], P["x"], ST[0], T[0] @ O.reshape_y(P["y"], 16384))     # 16384^3/3 on the host: seconds
    assert l1 == pytest.approx(ref, rel=1e-9)


def test_c3_shape_posterior_predictive(lmm):
    """configs[3] shape in Float64: posterior predictive with a 128x64 mixing matrix, n_train = n_test = 8192 (Orthogonal H,
    one GPU's 8 latents): prediction at the TRAINING inputs reproduces K alpha structure -- mean_l(x) = delta_l - s_l alpha_l
    i.e. H-mixed residual identity -- and latent 0's marginals match host LAPACK."""
    import torch
    n = 8192
    P = O.synthetic_problem(64, 128, n, "matern52", True, s2=0.1, seed=0)
    fs, H = _model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"])
    sh = (0, 8)
    f = lmm.ILMM(fs, H, shard=sh)
    xd = torch.from_numpy(P["x"]).cuda()
    post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(xd, 128), 0.1), torch.from_numpy(P["y"]).cuda())
    xs = P["x"] + 0.5 * 20.0 / 575.0
    lib = lmm.load()
    import ctypes as C
    ml, vl = np.empty(8 * n), np.empty(8 * n)
    from lmm_amd import _lib as L
    L.check(lib.lmm_latent_marginals(post.f._post.ptr, None, 8, L.Arr(xs).ptr, 1, n, L.Arr(ml, True).ptr, L.Arr(vl, True).ptr))
    T, ST = O.project_orthogonal(P["U"], P["S"], 0.1)
    po = O.gp_posterior(P["gps"][0], P["x"], ST[0], T[0] @ O.reshape_y(P["y"], n))
    mo, vo = O.gp_mean_var(po, xs)
    np.testing.assert_allclose(ml[:n], mo, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(vl[:n], vo, rtol=1e-6, atol=1e-10)
    assert np.all(vl > 0) and np.all(vl < 1.0 + 1e-12)          # 0 < posterior variance <= prior variance
    # mixed marginals = H (shard columns) applied to the latent marginals
    mu, v = lmm.mean_and_var(post(lmm.MOInputIsotopicByOutputs(xs, 128), 0.1), add_noise=True)
    Hs = O.orthogonal_dense(P["U"], P["S"])[:, :8]
    np.testing.assert_allclose(mu, (Hs @ ml.reshape(8, n)).reshape(-1), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v, ((Hs * Hs) @ (vl.reshape(8, n) + 1e-18)).reshape(-1) + 0.1, rtol=1e-10)


def test_c4_shape_rand_marginals(lmm):
    """configs[4] shape in Float64: p = 256 outputs, 128 latents, n = 32768 -- one full lock-step batch (8 of one GPU's 16
    latents): prior marginals (closed form); posterior marginals bounded by the prior AND latent 0's marginals against host
    LAPACK (32768^3/3 flops on the host cores: tens of seconds); a posterior sample given normals has the mixing structure
    (sample - mean lies in span(H_shard) before noise) and latent 0's sample equals the oracle's on the same normals."""
    import torch
    from lmm_amd import _lib as L
    n, p, m, nl, ns = 32768, 256, 128, 8, 2048
    P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
    fs, H = _model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"])
    f = lmm.ILMM(fs, H, shard=(0, nl))
    xd = torch.from_numpy(P["x"]).cuda()
    xin = lmm.MOInputIsotopicByOutputs(xd, p)
    mu, v = lmm.mean_and_var(f(xin, 0.1))                       # prior: mean 0, var = sum_l H[o,l]^2 (1 + 1e-18) + 0.1
    Hs = O.orthogonal_dense(P["U"], P["S"])[:, :nl]
    assert float(mu.abs().max()) == 0.0
    np.testing.assert_allclose(v.cpu().numpy().reshape(p, n)[:, 0], (Hs * Hs).sum(1) + 0.1, rtol=1e-12)
    yd = torch.from_numpy(P["y"]).cuda()
    post = lmm.posterior(f(xin, 0.1), yd)
    xs = P["x"][:ns] + 0.01
    xsin = lmm.MOInputIsotopicByOutputs(xs, p)
    mup, vp = lmm.mean_and_var(post(xsin, 0.1))
    assert np.all(np.isfinite(mup)) and np.all(vp > 0.1) and np.all(vp <= v.cpu().numpy().reshape(p, n)[:, :ns].reshape(-1) + 1e-12)
    # latent marginals of the batch; latent 0 against host LAPACK
    ml, vl = np.empty(nl * ns), np.empty(nl * ns)
    L.check(lmm.load().lmm_latent_marginals(post.f._post.ptr, None, nl, L.Arr(xs).ptr, 1, ns, L.Arr(ml, True).ptr, L.Arr(vl, True).ptr))
    T, ST = O.project_orthogonal(P["U"], P["S"], 0.1)
    po0 = O.gp_posterior(P["gps"][0], P["x"], ST[0], T[0] @ O.reshape_y(P["y"], n))
    mo, vo = O.gp_mean_var(po0, xs)
    np.testing.assert_allclose(ml[:ns], mo, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(vl[:ns], vo, rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(mup, (Hs @ ml.reshape(nl, ns)).reshape(-1), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(vp, ((Hs * Hs) @ (vl.reshape(nl, ns) + 1e-18)).reshape(-1) + 0.1, rtol=1e-10)
    # posterior sample of the batch given normals: mixing structure
    jit = (1e-9, 1e-8, 1e-8)
    s = lmm.rand(np.random.default_rng(0), post(xsin, 0.1), jitters=jit, add_noise=False)
    R = (s - mup).reshape(p, ns)
    resid = R - Hs @ np.linalg.lstsq(Hs, R, rcond=None)[0]
    assert np.abs(resid).max() < 1e-9 * max(1.0, np.abs(R).max())
    # latent 0's contribution on its own (a one-latent shard of the same model) against the oracle on the same normals
    del post
    post0 = lmm.posterior(lmm.ILMM(fs, H, shard=(0, 1))(xin, 0.1), yd)
    s0 = lmm.rand(np.random.default_rng(0), post0(xsin, 0.1), jitters=jit, add_noise=False)
    z0 = np.random.default_rng(0).standard_normal(m * ns)[:ns]                     # latent 0's block of the reference's draw order
    ref0 = O.gp_rand(po0, xs, 1e-8, z0)
    np.testing.assert_allclose(s0.reshape(p, ns), Hs[:, :1] * ref0[None, :], rtol=1e-6, atol=1e-8)
