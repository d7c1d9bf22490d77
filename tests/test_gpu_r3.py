"""-m gpu, round 3: (a) the bf16-MFMA covariance projection of BASELINE configs[3] (lmm_set_projection_dtype) against the
oracle at a small shape and against the Float64 path at configs[3]'s own shape, at the tolerance the header states;
(b) Gram assembly branches that no earlier test reached (lmm_kernels.hip gram_body): the separable-Matern guard fallback
(unsorted / widely spaced d = 1 inputs with a short lengthscale), strips that mix guarded and unguarded tiles, the LDS
fast path at its widest (d = 8) and the generic tiles beyond it (d = 9)."""
import ctypes as C

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


def _bf16(a):
    """Round-to-nearest-even bfloat16 image of a float64 array (through float32, as the kernel does)."""
    u = np.asarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return (u.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


# ---------------------------------------------------------------------------------------------------
# (a) bf16 projection
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["bf16", "bf16x2"])
def test_bf16_projection_small_vs_oracle(lmm, mode):
    """M = H M_lat, V = abs2.(H) V_lat .+ sigma2 (reference src/oilmm.jl:69-72) with bf16 operands on the matrix pipe:
    posterior AND prior marginals of a 7-output / 5-latent OILMM (ragged against the 16 x 16 x 32 tile in every direction)
    against the oracle, inside the header's bound 2^-7 sum_l |H^pw| |lat| (each bf16 operand carries 2^-8; bf16x2: 2^-15); and, for plain bf16, EQUAL (to
    Float32 accumulation error) to the same sum over bf16-rounded operands computed on the host."""
    from lmm_amd import _lib as L
    rng = np.random.default_rng(11)
    n, ns, p, m = 90, 37, 7, 5
    x = np.sort(rng.uniform(0, 6, n)); xs = rng.uniform(0, 6, ns)
    gps = [{"kind": k, "variance": float(rng.uniform(0.5, 2)), "lengthscale": float(rng.uniform(0.5, 2)), "mean": float(rng.normal())}
           for k in ["se", "matern32", "matern52", "se", "matern52"]]
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    S = np.linspace(2, 1, m)
    y = rng.standard_normal(n * p)
    f = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
    post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x, p), 0.1), y)
    pox = post(lmm.MOInputIsotopicByOutputs(xs, p), 0.1)
    mo, vo = O.oilmm_mean_var(O.oilmm_posterior(gps, U, S, x, 0.1, y), U, S, xs, 0.1)
    H = O.orthogonal_dense(U, S)
    lib = lmm.load()
    ml, vl = np.empty(m * ns), np.empty(m * ns)
    L.check(lib.lmm_latent_marginals(post.f._post.ptr, None, m, L.Arr(xs).ptr, 1, ns, L.Arr(ml, True).ptr, L.Arr(vl, True).ptr))
    ml, vl = ml.reshape(m, ns), vl.reshape(m, ns)
    try:
        lmm.set_projection_dtype(mode)
        assert lmm.get_projection_dtype() == mode
        mu, v = lmm.mean_and_var(pox)
        mu_only = lmm.mean(pox)
        pmu, pv = lmm.mean_and_var(f(lmm.MOInputIsotopicByOutputs(xs, p), 0.1))
    finally:
        lmm.set_projection_dtype("native")
    eps = 2.0 ** -7 if mode == "bf16" else 2.0 ** -15
    bm = (eps * (np.abs(H) @ np.abs(ml))).reshape(-1)
    bv = (eps * ((H * H) @ np.abs(vl))).reshape(-1)
    assert np.all(np.abs(mu - mo) <= 1.01 * bm + 1e-12)
    assert np.all(np.abs(v - vo) <= 1.01 * bv + 1e-12)
    assert np.max(np.abs(mu - mo)) > 1e-9 or mode == "bf16x2"       # the bf16 path really ran (Float64 agrees to 1e-12)
    # mean(fx) alone (mu + K(x*, x) alpha, no solve) takes the same projection; the latent means of the two forms agree to 1e-9
    assert np.all(np.abs(mu_only - mo) <= 1.01 * bm + 1e-7)
    if mode == "bf16":
        np.testing.assert_allclose(mu, (_bf16(H) @ _bf16(ml)).reshape(-1), rtol=0, atol=2e-6 * np.max(np.abs(H) @ np.abs(ml)))
        np.testing.assert_allclose(v, (_bf16(H * H) @ _bf16(vl + 1e-18)).reshape(-1) + 0.1, rtol=0, atol=2e-6 * np.max((H * H) @ vl))
    # prior marginals: mean = H mean_l, var = abs2.(H) (variance_l + 1e-18) + sigma2
    means = np.array([g["mean"] for g in gps]); vars_ = np.array([g["variance"] for g in gps])
    assert np.all(np.abs(pmu.reshape(p, ns) - (H @ means)[:, None]) <= 1.01 * eps * (np.abs(H) @ np.abs(means))[:, None] + 1e-12)
    assert np.all(np.abs(pv.reshape(p, ns) - ((H * H) @ vars_ + 0.1)[:, None]) <= 1.01 * eps * ((H * H) @ vars_)[:, None] + 1e-12)


def test_c3_bf16_projection_full_size(lmm):
    """configs[3] AS NAMED: posterior predictive with a 128 x 64 mixing matrix, n_train = n_test = 8192, bf16 MFMA covariance
    projection -- one GPU's share (8 of the 64 latents; the latent marginals stay Float64).  The Float64 path of the same
    handle is the reference value (itself checked against host LAPACK on latent 0 by test_c3_shape_posterior_predictive);
    tolerance: the header's 2^-7 sum_l |H^pw| |lat| bound, elementwise."""
    import torch
    from lmm_amd import _lib as L
    n = 8192
    P = O.synthetic_problem(64, 128, n, "matern52", True, s2=0.1, seed=0)
    f = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]), shard=(0, 8))
    xd = torch.from_numpy(P["x"]).cuda()
    post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(xd, 128), 0.1), torch.from_numpy(P["y"]).cuda())
    xs = P["x"] + 0.5 * 20.0 / 575.0
    pox = post(lmm.MOInputIsotopicByOutputs(xs, 128), 0.1)
    mu64, v64 = lmm.mean_and_var(pox, add_noise=True)
    try:
        lmm.set_projection_dtype("bf16")
        mu16, v16 = lmm.mean_and_var(pox, add_noise=True)
    finally:
        lmm.set_projection_dtype("native")
    ml, vl = np.empty(8 * n), np.empty(8 * n)
    L.check(lmm.load().lmm_latent_marginals(post.f._post.ptr, None, 8, L.Arr(xs).ptr, 1, n, L.Arr(ml, True).ptr, L.Arr(vl, True).ptr))
    Hs = O.orthogonal_dense(P["U"], P["S"])[:, :8]
    bm = (2.0 ** -7 * (np.abs(Hs) @ np.abs(ml.reshape(8, n)))).reshape(-1)
    bv = (2.0 ** -7 * ((Hs * Hs) @ vl.reshape(8, n))).reshape(-1)
    assert np.all(np.abs(mu16 - mu64) <= 1.01 * bm + 1e-12)
    assert np.all(np.abs(v16 - v64) <= 1.01 * bv + 1e-12)
    assert np.max(np.abs(mu16 - mu64)) > 1e-8                                 # not the Float64 kernel in disguise
    assert np.all(v16 > 0.1 - 1e-12)                                          # sigma2 is added in Float64, after the MFMA sum


# ---------------------------------------------------------------------------------------------------
# (b) Gram assembly branches (lmm_kernels.hip gram_body)
# ---------------------------------------------------------------------------------------------------
def _gram(lmm, gp, x, d, n, diag=0.25):
    import torch
    from lmm_amd import _lib as L
    NC = NR = (n + 63) // 64 * 64
    ld = NR + 16
    A = torch.full((NC, ld), float("nan"), dtype=torch.float64, device="cuda")
    xd = torch.from_numpy(np.ascontiguousarray(x.T if d > 1 else x)).cuda()
    torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
    rc = lmm.load().lmm_dev_gram(C.c_void_p(A.data_ptr()), ld, NR, NC, C.c_void_p(xd.data_ptr()), d, n, L.gps_array([gp]), C.c_double(diag))
    assert rc == 0, lmm.load().lmm_last_error_string()
    return A.T.cpu().numpy()[:NR]


@pytest.mark.parametrize("kind", ["matern32", "matern52"])
@pytest.mark.parametrize("layout", ["unsorted", "mixed", "sorted_wide"])
def test_gram_separable_matern_guard(lmm, kind, layout):
    """d = 1 Matern with a SHORT lengthscale (0.3) on [0, 200]: a |x - c| / l beyond 40 trips the guard of the separable
    exp(-a|xi - xj|) = min(E_i F_j, F_i E_j) form, and the tile falls back to the per-element exponential.
      unsorted    : every 64-point strip spans the whole range -> the guard trips in (almost) every interior tile;
      mixed       : the first 256 points are dense and sorted (guard holds), the rest unsorted -> strips whose tiles take
                    different branches, and row strips that pass the guard against column tiles that do not;
      sorted_wide : sorted with spacing 0.5 (a 64-strip spans 32 = 107 lengthscales, a sqrt(5)/0.3 x 32 > 40 -> trips) next to
                    sorted_dense strips that pass."""
    rng = np.random.default_rng({"unsorted": 1, "mixed": 2, "sorted_wide": 3}[layout])
    n = 640 + 37                                   # ragged last strip: border tiles take the generic routine
    if layout == "unsorted":
        x = rng.uniform(0, 200, n)
    elif layout == "mixed":
        x = np.concatenate([np.sort(rng.uniform(0, 1.5, 256)), rng.uniform(0, 200, n - 256)])
    else:
        x = np.concatenate([np.arange(320) * 0.5, 160.0 + np.sort(rng.uniform(0, 2.0, n - 320))])
    gp = {"kind": kind, "variance": 1.3, "lengthscale": 0.3, "mean": 0.0}
    got = _gram(lmm, gp, x, 1, n)
    ref = O.kernelmatrix(gp, x) + 0.25 * np.eye(n)
    il = np.tril_indices(n)
    # exp(-s) carries the rounding of its ARGUMENT (s = sqrt(5) |xi - xj| / l, formed as r / l by the oracle and r * (1 / l) by
    # the kernel: a few ulps of s, i.e. a relative 1e-15 s in the value): 2e-13 down to s ~ 100 (values >= 1e-40), 2e-12 for
    # the far tail; entries below 1e-290 approach the subnormal range of v_ldexp: absolute there
    g, r = got[:n, :n][il], ref[il]
    near = r > 1e-40
    np.testing.assert_allclose(g[near], r[near], rtol=2e-13, atol=0)
    np.testing.assert_allclose(g[~near], r[~near], rtol=2e-12, atol=1e-290)
    NC = got.shape[0]
    np.testing.assert_array_equal(np.tril(got[n:, n:NC]), np.eye(NC - n))


@pytest.mark.parametrize("kind", ["se", "matern32", "matern52"])
@pytest.mark.parametrize("d", [8, 9, 12])
def test_gram_high_dimensional_inputs(lmm, kind, d):
    """d = 8: the widest input dimension of the LDS fast path (column points pre-scaled in LDS, row points in registers);
    d = 9, 12: beyond it, every tile takes the generic routine.  n = 300 (interior AND border tiles)."""
    rng = np.random.default_rng(40 + d)
    n = 300
    x = rng.uniform(0, 2.5, size=(d, n))
    gp = {"kind": kind, "variance": 0.7, "lengthscale": 1.9, "mean": 0.0}
    got = _gram(lmm, gp, x, d, n)
    ref = O.kernelmatrix(gp, x) + 0.25 * np.eye(n)
    il = np.tril_indices(n)
    np.testing.assert_allclose(got[:n, :n][il], ref[il], rtol=2e-13, atol=1e-300)


# ---------------------------------------------------------------------------------------------------
# (c) get_latent_gp(posterior(dense-H ILMM)): the coupled latent PosteriorGP{IndependentMOGP}
# ---------------------------------------------------------------------------------------------------
def test_dense_posterior_latent_gp_vs_oracle(lmm):
    """reference src/ilmm.jl:39 applied to the ILMM that posterior(fx, y) returns (:196-197): get_latent_gp gives the latent
    PosteriorGP{IndependentMOGP}, which answers the AbstractGPs verbs on MOInputIsotopicByOutputs(x*, m) like any other FiniteGP:
    mean / var / cov = the latent posterior's joint moments + sigma2 I, logpdf = the generic Gaussian, rand = mean +
    chol(cov + sigma2 I).U' z, posterior = conditioning on further LATENT observations.  Oracle: O._ilmm_latent_joint on
    O.ilmm_posterior (the dense (mn) x (mn) restatement)."""
    import scipy.linalg as sla
    rng = np.random.default_rng(5)
    n, ns, p, m = 70, 9, 4, 3                       # m n = 210: four 64-column blocks in the coupled factor
    x = np.sort(rng.uniform(0, 6, n)); xs = rng.uniform(0, 6, ns)
    gps = [{"kind": k, "variance": float(rng.uniform(0.5, 2)), "lengthscale": float(rng.uniform(0.5, 2)), "mean": float(rng.normal())}
           for k in ["se", "matern32", "matern52"]]
    H = rng.uniform(size=(p, m)); y = rng.standard_normal(n * p)
    post = lmm.posterior(lmm.ILMM(_model(lmm, gps), H)(lmm.MOInputIsotopicByOutputs(x, p), 0.1), y)
    fl = lmm.get_latent_gp(post)
    assert isinstance(fl, lmm.IndependentMOGP)
    s2 = 0.07
    flx = fl(lmm.MOInputIsotopicByOutputs(xs, m), s2)
    po = O.ilmm_posterior(gps, H, x, 0.1, y)
    mo, Co = O._ilmm_latent_joint(po, xs)
    Co = Co + s2 * np.eye(m * ns)
    mu, v = lmm.mean_and_var(flx)
    np.testing.assert_allclose(mu, mo, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(v, np.diag(Co), rtol=1e-8)
    mu2, Cg = lmm.mean_and_cov(flx)
    np.testing.assert_allclose(mu2, mo, rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(Cg, Co, rtol=1e-7, atol=1e-10)
    ys = rng.standard_normal(m * ns)
    assert lmm.logpdf(flx, ys) == pytest.approx(O.gaussian_logpdf(mo, Co, ys), rel=1e-8)
    Ym = rng.standard_normal((m * ns, 2))            # matrix-Y: one value per column
    np.testing.assert_allclose(lmm.logpdf(flx, Ym), [O.gaussian_logpdf(mo, Co, Ym[:, c]) for c in range(2)], rtol=1e-8)
    z = np.random.default_rng(77).standard_normal(m * ns)
    smp = lmm.rand(np.random.default_rng(77), flx)
    np.testing.assert_allclose(smp, mo + np.linalg.cholesky(Co) @ z, rtol=1e-7, atol=1e-9)
    S2 = lmm.rand(np.random.default_rng(78), flx, 2)
    assert S2.shape == (m * ns, 2)
    # conditioning the latent GP on latent observations at x2: Gaussian conditioning of (mo, Co) -- checked through the
    # predictive mean at a third set of points, which is linear in the joint latent moments
    x3 = rng.uniform(0, 6, 5)
    pl2 = lmm.posterior(flx, ys)
    m3, v3 = lmm.mean_and_var(pl2(lmm.MOInputIsotopicByOutputs(x3, m), 0.0))
    xall = np.concatenate([xs, x3])
    ma, Ca = O._ilmm_latent_joint(po, xall)
    idx_s = np.concatenate([np.arange(ns) + l * (ns + 5) for l in range(m)])
    idx_3 = np.concatenate([np.arange(5) + ns + l * (ns + 5) for l in range(m)])
    Kss = Ca[np.ix_(idx_s, idx_s)] + s2 * np.eye(m * ns)
    K3s = Ca[np.ix_(idx_3, idx_s)]
    sol = sla.cho_solve(sla.cho_factor(Kss, lower=True), np.column_stack([ys - ma[idx_s], K3s.T]))
    np.testing.assert_allclose(m3, ma[idx_3] + K3s @ sol[:, 0], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(v3, np.diag(Ca[np.ix_(idx_3, idx_3)] - K3s @ sol[:, 1:]), rtol=1e-5, atol=1e-9)
    # destroying the ILMM posterior first must not invalidate the latent view (the base's release is deferred)
    view = fl._post.latent_view()
    del post, flx, pl2
    import gc; gc.collect()
    mu_again, _ = lmm.mean_and_var(fl(lmm.MOInputIsotopicByOutputs(xs, m), s2))
    np.testing.assert_allclose(mu_again, mo, rtol=1e-8, atol=1e-10)
    assert view.ptr


# ---------------------------------------------------------------------------------------------------
# (d) predictive-logpdf gradients across a tile boundary: n = 90, ns = 37 => N = 127 joint points, two 64-tiles, the training /
#     test split (nsplit = 90) unaligned with them, m = 3 latents (round-2 tests stopped at n + ns <= 25: one tile)
# ---------------------------------------------------------------------------------------------------
def _fd(f, h=1e-6):
    return (f(h) - f(-h)) / (2.0 * h)


def _gps3(rng):
    return [{"kind": k, "variance": float(rng.uniform(0.5, 2)), "lengthscale": float(rng.uniform(0.7, 2)), "mean": float(rng.normal())}
            for k in ["se", "matern32", "matern52"]]


def _check_gps_grad(G, F, gps, idx, rel, ab):
    for l in idx:
        for key in ("variance", "lengthscale", "mean"):
            def f1(t, l=l, key=key):
                g2 = [dict(g) for g in gps]; g2[l][key] += t
                return F(gps=g2)
            assert G["gps"][l][key] == pytest.approx(_fd(f1), rel=rel, abs=ab), (l, key)


def test_posterior_oilmm_gradient_two_tiles(lmm):
    rng = np.random.default_rng(71)
    n, ns, p, m = 90, 37, 4, 3
    x = np.sort(rng.uniform(0, 9, n)); xs = np.sort(rng.uniform(0, 9, ns))
    gps = _gps3(rng)
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    s2, s2s = 0.3, 0.2

    def F(gps=gps, U=U, S=S, s2=s2, s2s=s2s, y=y, ys=ys):
        return O.oilmm_logpdf(O.oilmm_posterior(gps, U, S, x, s2, y), U, S, xs, s2s, ys)

    f = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
    fxs = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x, p), s2), y)(lmm.MOInputIsotopicByOutputs(xs, p), s2s)
    G = lmm.logpdf_and_gradient(fxs, ys)
    assert G["value"] == pytest.approx(F(), rel=1e-9)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=2e-5, abs=1e-6)
    assert G["sigma2_train"] == pytest.approx(_fd(lambda t: F(s2=s2 + t)), rel=2e-5, abs=1e-6)
    for k in [0, ns + 30, ns * p - 1]:
        e = np.zeros(ns * p); e[k] = 1.0
        assert G["y"][k] == pytest.approx(_fd(lambda t: F(ys=ys + t * e)), rel=2e-5, abs=1e-6)
    for k in [0, n + 70, n * p - 1]:                    # a training point beyond the first tile
        e = np.zeros(n * p); e[k] = 1.0
        assert G["y_train"][k] == pytest.approx(_fd(lambda t: F(y=y + t * e)), rel=2e-5, abs=1e-6)
    _check_gps_grad(G, F, gps, [1], 2e-5, 1e-6)
    e = np.zeros(m); e[2] = 1.0
    assert G["S"][2] == pytest.approx(_fd(lambda t: F(S=S + t * e)), rel=2e-5, abs=1e-6)


def test_posterior_mogp_gradient_two_tiles(lmm):
    rng = np.random.default_rng(72)
    n, ns, m = 90, 37, 3
    x = np.sort(rng.uniform(0, 9, n)); xs = np.sort(rng.uniform(0, 9, ns))
    gps = _gps3(rng)
    y, ys = rng.standard_normal(n * m), rng.standard_normal(ns * m)
    s2, s2s = 0.4, 0.25

    def F(gps=gps, s2=s2, s2s=s2s, y=y, ys=ys):
        return O.mogp_logpdf(O.mogp_posterior(gps, x, s2, y), xs, s2s, ys)

    fxs = lmm.posterior(_model(lmm, gps)(lmm.MOInputIsotopicByOutputs(x, m), s2), y)(lmm.MOInputIsotopicByOutputs(xs, m), s2s)
    G = lmm.logpdf_and_gradient(fxs, ys)
    assert G["value"] == pytest.approx(F(), rel=1e-9)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=2e-5, abs=1e-6)
    assert G["sigma2_train"] == pytest.approx(_fd(lambda t: F(s2=s2 + t)), rel=2e-5, abs=1e-6)
    for k in [3, ns + 36, ns * m - 1]:
        e = np.zeros(ns * m); e[k] = 1.0
        assert G["y"][k] == pytest.approx(_fd(lambda t: F(ys=ys + t * e)), rel=2e-5, abs=1e-6)
    for k in [5, 2 * n + 80]:
        e = np.zeros(n * m); e[k] = 1.0
        assert G["y_train"][k] == pytest.approx(_fd(lambda t: F(y=y + t * e)), rel=2e-5, abs=1e-6)
    _check_gps_grad(G, F, gps, [0, 2], 2e-5, 1e-6)


def test_posterior_dense_ilmm_gradient_two_tiles(lmm):
    """the dense two-noise-block core at m N = 381 coupled unknowns (six 64-blocks), nsplit = 90 unaligned."""
    rng = np.random.default_rng(73)
    n, ns, p, m = 90, 37, 4, 3
    x = np.sort(rng.uniform(0, 9, n)); xs = np.sort(rng.uniform(0, 9, ns))
    gps = _gps3(rng)
    H = rng.uniform(0.2, 1.0, size=(p, m))
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    s2, s2s = 0.3, 0.2

    def F(gps=gps, H=H, s2=s2, s2s=s2s, y=y, ys=ys):
        return O.ilmm_logpdf(O.ilmm_posterior(gps, H, x, s2, y), H, xs, s2s, ys)

    f = lmm.ILMM(_model(lmm, gps), H)
    fxs = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x, p), s2), y)(lmm.MOInputIsotopicByOutputs(xs, p), s2s)
    G = lmm.logpdf_and_gradient(fxs, ys)
    assert G["value"] == pytest.approx(F(), rel=1e-7)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=5e-5, abs=1e-5)
    assert G["sigma2_train"] == pytest.approx(_fd(lambda t: F(s2=s2 + t)), rel=5e-5, abs=1e-5)
    for k in [0, ns + 30, ns * p - 1]:
        e = np.zeros(ns * p); e[k] = 1.0
        assert G["y"][k] == pytest.approx(_fd(lambda t: F(ys=ys + t * e)), rel=2e-5, abs=1e-6)
    for k in [0, n + 70, n * p - 1]:
        e = np.zeros(n * p); e[k] = 1.0
        assert G["y_train"][k] == pytest.approx(_fd(lambda t: F(y=y + t * e)), rel=2e-5, abs=1e-6)
    E = np.zeros((p, m)); E[1, 2] = 1.0
    assert G["H"][1, 2] == pytest.approx(_fd(lambda t: F(H=H + t * E)), rel=5e-5, abs=1e-5)
    _check_gps_grad(G, F, gps, [1], 5e-5, 1e-5)


# ---------------------------------------------------------------------------------------------------
# (e) the 128-column panel path of the factorisation (lmm_kernels.hip K2c: leaf128, bulk GEMM by the panel inverse, the trailing
#     update fused with the next panel's leaf) against host LAPACK, through the exported building block lmm_dev_potrf
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,nrider", [(128, 0), (128, 64), (192, 64), (256, 128), (500, 3), (576, 64), (1000, 1), (1536, 192), (2111, 5)])
def test_panel_factorisation_vs_lapack(lmm, n, nrider):
    """L (lower triangle, in place), the rider rows R L^-T and the 64 x 64 inverse diagonal blocks W_b = L_bb^-1 for widths that
    are 0 and 64 mod 128, with rider counts that make the row count an even and an odd multiple of 64 (half row tiles), one to
    twenty-four 64-column blocks (one to five recursion levels; n = 2111: updates of K >= 1024 take the two-tile-deep prefetch)."""
    import torch
    import scipy.linalg as sla
    rng = np.random.default_rng(n + nrider)
    NC = (n + 63) // 64 * 64
    NR = (NC + nrider + 63) // 64 * 64
    ld = NR + 16
    G = rng.standard_normal((n, n + 8))
    K = G @ G.T / (n + 8) + 0.5 * np.eye(n)
    R = rng.standard_normal((NR - NC, NC))
    full = np.zeros((NR, NC))
    full[:n, :n] = np.tril(K)
    full[n:NC, n:] = np.eye(NC - n)
    full[NC:, :] = R
    A = torch.full((NC, ld), float("nan"), dtype=torch.float64, device="cuda")      # column-major: A[col][row]; upper triangle stays NaN
    At = torch.from_numpy(np.ascontiguousarray(full.T))
    A[:, :NR] = At.cuda()
    iu = np.triu_indices(NC, 1)
    Ah = A.cpu().numpy(); Ah[iu[1], iu[0]] = np.nan; A = torch.from_numpy(Ah).cuda()   # never-read upper triangle: NaN
    W = torch.full((NC // 64, 64, 64), float("nan"), dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    lib = lmm.load()
    torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
    rc = lib.lmm_dev_potrf(C.c_void_p(A.data_ptr()), NR, NC, ld, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr()))
    assert rc == 0, lib.lmm_last_error_string()
    assert int(info.item()) == 0
    got = A.cpu().numpy().T[:NR]                                  # [row][col]
    Kp = np.eye(NC); Kp[:n, :n] = K
    Lref = np.linalg.cholesky(Kp)
    il = np.tril_indices(NC)
    np.testing.assert_allclose(got[:NC][il], Lref[il], rtol=0, atol=2e-12 * np.abs(Lref).max())
    if NR > NC:
        Xref = sla.solve_triangular(Lref, R.T, lower=True).T      # R L^-T
        np.testing.assert_allclose(got[NC:], Xref, rtol=0, atol=1e-10 * max(1.0, np.abs(Xref).max()))
    Wh = W.cpu().numpy()                                          # [block][col][row]
    for b in range(NC // 64):
        Lbb = Lref[64 * b:64 * b + 64, 64 * b:64 * b + 64]
        np.testing.assert_allclose(Wh[b].T @ Lbb, np.eye(64), rtol=0, atol=1e-10)
        assert np.all(np.triu(Wh[b].T, 1) == 0.0)


def test_panel_path_reports_the_failing_pivot(lmm):
    """a non-positive pivot inside a fused leaf (second panel, second 64-block) surfaces as the LAPACK-style 1-based info."""
    import torch
    n = 384
    K = np.eye(n) * 2.0
    K[300, 300] = -1.0
    ld = n + 16
    A = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    A[:, :n] = torch.from_numpy(np.ascontiguousarray(np.tril(K).T)).cuda()
    W = torch.zeros((n // 64, 64, 64), dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()      # raw pointers cross the ABI: torch's asynchronous producers of these tensors must be done
    rc = lmm.load().lmm_dev_potrf(C.c_void_p(A.data_ptr()), n, n, ld, C.c_void_p(W.data_ptr()), n, C.c_void_p(info.data_ptr()))
    assert rc == 0
    assert int(info.item()) == 301


# ---------------------------------------------------------------------------------------------------
# (f) gradient of the predictive logpdf after SEQUENTIAL conditioning (equal noise per batch: the batches merge), and
#     device inputs produced on a side stream named through lmm_amd.wait_stream
# ---------------------------------------------------------------------------------------------------
def test_gradient_after_sequential_conditioning(lmm):
    """Zygote differentiates logpdf(posterior(posterior(f(x1, s2), y1)(x2, s2), y2)(xs, s2s), ys) in the reference (src/oilmm.jl:116-134
    composed twice).  With equal noise on the batches the mirror merges them; value and TOTAL derivatives against central finite
    differences of the oracle's two-step conditioning, incl. d/dy of EACH batch."""
    rng = np.random.default_rng(81)
    n1, n2, ns, p, m = 40, 33, 11, 4, 3
    x1, x2, xs = np.sort(rng.uniform(0, 8, n1)), np.sort(rng.uniform(0, 8, n2)), np.sort(rng.uniform(0, 8, ns))
    gps = _gps3(rng)
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    y1, y2, ys = rng.standard_normal(n1 * p), rng.standard_normal(n2 * p), rng.standard_normal(ns * p)
    s2, s2s = 0.3, 0.2

    def F(gps=gps, s2=s2, s2s=s2s, y1=y1, y2=y2, ys=ys):
        return O.oilmm_logpdf(O.oilmm_posterior(O.oilmm_posterior(gps, U, S, x1, s2, y1), U, S, x2, s2, y2), U, S, xs, s2s, ys)

    f = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
    po2 = lmm.posterior(lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x1, p), s2), y1)(lmm.MOInputIsotopicByOutputs(x2, p), s2), y2)
    fxs = po2(lmm.MOInputIsotopicByOutputs(xs, p), s2s)
    G = lmm.logpdf_and_gradient(fxs, ys)
    assert G["value"] == pytest.approx(F(), rel=1e-8)
    assert G["value"] == pytest.approx(lmm.logpdf(fxs, ys), rel=1e-8)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=2e-5, abs=1e-6)
    assert G["sigma2_train"] == pytest.approx(_fd(lambda t: F(s2=s2 + t)), rel=2e-5, abs=1e-6)
    assert isinstance(G["y_train"], list) and [len(g) for g in G["y_train"]] == [n1 * p, n2 * p]
    for k in [0, n1 + 7, n1 * p - 1]:
        e = np.zeros(n1 * p); e[k] = 1.0
        assert G["y_train"][0][k] == pytest.approx(_fd(lambda t: F(y1=y1 + t * e)), rel=2e-5, abs=1e-6)
    for k in [3, 2 * n2 + 5]:
        e = np.zeros(n2 * p); e[k] = 1.0
        assert G["y_train"][1][k] == pytest.approx(_fd(lambda t: F(y2=y2 + t * e)), rel=2e-5, abs=1e-6)
    _check_gps_grad(G, F, gps, [2], 2e-5, 1e-6)
    po_mixed = lmm.posterior(lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x1, p), 0.1), y1)(lmm.MOInputIsotopicByOutputs(x2, p), 0.3), y2)
    # different variances per batch (refused until round 5): one noise block per batch; the per-batch derivatives are checked in
    # tests/test_gpu_r5.py, here the value
    Gm = lmm.logpdf_and_gradient(po_mixed(lmm.MOInputIsotopicByOutputs(xs, p), s2s), ys)
    ref = O.oilmm_logpdf(O.oilmm_posterior(O.oilmm_posterior(gps, U, S, x1, 0.1, y1), U, S, x2, 0.3, y2), U, S, xs, s2s, ys)
    assert Gm["value"] == pytest.approx(ref, rel=1e-8) and len(Gm["sigma2_train"]) == 2


def test_inputs_from_a_side_stream(lmm):
    """A device tensor produced on a stream that is NOT torch's current stream: the caller names it (lmm_amd.wait_stream ->
    lmm_stream_wait_caller), and the library's streams order themselves behind it."""
    import torch
    P = O.synthetic_problem(3, 5, 700, "matern52", True, s2=0.1, seed=3)
    f = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    ref = O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"])
    side = torch.cuda.Stream()
    yh = torch.from_numpy(P["y"]).pin_memory()
    big = torch.randn(1 << 24, device="cuda", dtype=torch.float64)
    with torch.cuda.stream(side):
        for _ in range(20):
            big = big * 1.0000001                          # keep the side stream busy ahead of the copy
        yd = yh.to("cuda", non_blocking=True)
        xd = torch.from_numpy(P["x"]).to("cuda", non_blocking=True)
    lmm.wait_stream(side)
    got = lmm.logpdf(f(lmm.MOInputIsotopicByOutputs(xd, 5), 0.1), yd)
    side.synchronize()
    assert got == pytest.approx(ref, rel=1e-9)


# ---------------------------------------------------------------------------------------------------
# (h) widths whose last 64-column block is padding only (the factor width is rounded up to 128 columns; potrf_region_kernel's walker
#     passes over such a block: L = [0 .. 0 I], W = I): logpdf, and the posterior verbs that go on to use the stored factor and
#     inverse blocks, against the oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,kind,d", [(130, "matern52", 1), (320, "se", 2), (552, "matern32", 1), (833, "matern52", 1)])
def test_all_padding_last_block(lmm, n, kind, d):
    rng = np.random.default_rng(n)
    m, p, s2 = 3, 4, 0.1
    x = np.sort(rng.uniform(0, 9, size=n)) if d == 1 else rng.uniform(0, 5, size=(d, n))
    gps = [{"kind": kind, "variance": 0.7 + 0.3 * l, "lengthscale": 0.8 + 0.2 * l, "mean": 0.1 * l} for l in range(m)]
    U, _ = np.linalg.qr(rng.standard_normal((p, m)))
    S = np.linspace(1.5, 0.8, m)
    y = rng.standard_normal(n * p)
    fx = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), s2)
    assert lmm.logpdf(fx, y) == pytest.approx(O.oilmm_logpdf(gps, U, S, x, s2, y), rel=1e-9)
    ns = 70
    xs = (np.sort(rng.uniform(0, 9, size=ns)) if d == 1 else rng.uniform(0, 5, size=(d, ns)))
    po = O.oilmm_posterior(gps, U, S, x, s2, y)
    mo, vo = O.oilmm_mean_var(po, U, S, xs, s2)
    post = lmm.posterior(fx, y)
    mu, v = lmm.mean_and_var(post(lmm.MOInputIsotopicByOutputs(xs, p), s2))
    np.testing.assert_allclose(mu, mo, rtol=1e-7, atol=1e-8)
    np.testing.assert_allclose(v, vo, rtol=1e-7)
    ys = rng.standard_normal(ns * p)
    assert lmm.logpdf(post(lmm.MOInputIsotopicByOutputs(xs, p), s2), ys) == pytest.approx(
        O.oilmm_logpdf(po, U, S, xs, s2, ys), rel=1e-8)


def test_two_concurrent_batches_with_region_base(lmm):
    """36 latents at n = 1100 (factor width 1152 > one region): two lock-step batches (32 + 4) on concurrent streams, each taking its own
    choice of base case for the recursion (the small one: 1024-column dataflow launches with assistants; the large one: by the residency
    rule) -- the sum over latents must still be the oracle's."""
    rng = np.random.default_rng(36)
    n, m, p, s2 = 1100, 36, 40, 0.2
    x = np.sort(rng.uniform(0, 30, size=n))
    kinds = ["matern52", "se", "matern32"]
    gps = [{"kind": kinds[l % 3], "variance": 0.6 + 0.02 * l, "lengthscale": 0.7 + 0.03 * l, "mean": 0.01 * l} for l in range(m)]
    U, _ = np.linalg.qr(rng.standard_normal((p, m)))
    S = np.linspace(1.6, 0.7, m)
    y = rng.standard_normal(n * p)
    fx = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), s2)
    assert lmm.logpdf(fx, y) == pytest.approx(O.oilmm_logpdf(gps, U, S, x, s2, y), rel=1e-9)


@pytest.mark.parametrize("n,m", [(552, 20), (1024, 8), (1100, 36), (2048, 16)])
def test_region_kernel_is_bitwise_reproducible(lmm, n, m):
    """potrf_region_kernel (walker, helpers, assistants, thin row streams; flags + write-through publishing) uses no atomics on data: a
    repeated evaluation must return the same bits -- a missed dependency would show as a differing value (tools/stress_region.py runs the
    long version, profiles/r03/stress_region.txt)."""
    import torch
    from lmm_amd import workloads as W
    P = W.synthetic_problem(m, 2 * m, n, "matern52", True, s2=0.1, seed=n + m)
    fs = lmm.independent_mogp([lmm.GP(lmm.Matern52Kernel()) for _ in range(m)])
    fx = lmm.ILMM(fs, lmm.Orthogonal(P["U"], P["S"]))(lmm.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1)
    yd = torch.from_numpy(P["y"]).cuda()
    # (1100, 36), (2048, 16): the panel recursion with the leaf inside the update launches, many matrices per launch -- the shapes on
    # which round 4's three-wave diagonal block first lost a hand-off (a clobbered exchange buffer: NaN once in ~10 evaluations)
    vals = {lmm.logpdf(fx, yd) for _ in range(60 if n <= 1024 else 24)}
    assert all(np.isfinite(v) for v in vals), vals
    if n <= 1024:
        assert len(vals) == 1, vals
    else:
        # above 1024 columns the factorisation has K >= 1024 update launches, whose split-K parts are combined with f64 atomics in
        # whatever order they finish (LMM_DETERMINISTIC=1 turns that off): the last bits may differ, a lost hand-off would not stop there
        v = sorted(vals)
        assert (v[-1] - v[0]) <= 1e-13 * abs(v[0]), vals


def test_alternating_problems_never_see_recycled_memory(lmm):
    """Different problems of the same shapes in turn (tools/stress_alternate.py, short form): whatever a recycled host or device buffer
    still holds from the previous call is wrong for the current one, so a pointer that outlives its array, or a consumer that reads
    ahead of its producer, changes the value.  (Found that way: the mirror handed the C ABI pointers into temporaries -- the transposed
    copy of a (d, n) input -- that died before the call; _lib._OwnedPtr keeps them alive.)"""
    def problem(seed, n, d, p=4, m=3):
        rng = np.random.default_rng(seed)
        x = np.sort(rng.uniform(0, 6, n)) if d == 1 else rng.uniform(0, 4, size=(d, n))
        kinds = [lmm.Matern52Kernel, lmm.SEKernel, lmm.Matern32Kernel]
        gps = [lmm.GP(float(rng.normal()), kinds[l % 3](float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.5, 2.0)))) for l in range(m)]
        U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
        return lmm.ILMM(lmm.independent_mogp(gps), lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x, p), 0.3), rng.standard_normal(n * p)
    probs = [problem(17 * i + n, n, d) for i, (n, d) in enumerate([(130, 2), (150, 1), (130, 1), (150, 2)])]
    ref = [(lmm.logpdf(fx, y), lmm.logpdf_and_gradient(fx, y)["value"]) for fx, y in probs]
    for fx, y in probs:                 # the two entry points agree to begin with
        assert lmm.logpdf(fx, y) == pytest.approx(lmm.logpdf_and_gradient(fx, y)["value"], rel=1e-12)
    for it in range(150):
        for k, (fx, y) in enumerate(probs):
            v = lmm.logpdf(fx, y) if it % 2 == 0 else lmm.logpdf_and_gradient(fx, y)["value"]
            assert v == ref[k][it % 2], (it, k, v, ref[k])
