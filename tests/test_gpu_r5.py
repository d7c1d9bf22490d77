"""-m gpu, round 5.
(a) cov(f::IndependentMOGP, x, y), the two-input cross-covariance (reference src/independent_mogp.jl:66-71 by outputs, :184-215 by
    features / mixed; the reference's own checks: test/independent_mogp.jl:136-141): lmm_mogp_cross_cov against the oracle's restatement
    and against the naive LinearMixingModelKernel with H = I, prior and posterior latents, all four input-order combinations.
(b) The headline number has a check: the full configs[2] value (32 latents, the N = 1 plan of bench.py: two concurrent 16-latent
    batches, 512-column dataflow base case) equals the sum of the eight latent_shard(32, r, 8) partials -- what the eight ranks of the
    8-GPU job compute, each through the 4-latent plan (one batch, 1024-column base case) -- and latent 0's term equals host LAPACK.
(c) The flag-epoch wrap-around clear of the dataflow kernels (lmm_kernels.hip next_flag_epoch) executed once: two region evaluations
    across the wrap against the oracle."""
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


def _gps(m, rng):
    kinds = ["matern52", "se", "matern32"]
    return [{"kind": kinds[l % 3], "variance": float(rng.uniform(0.6, 1.4)), "lengthscale": float(rng.uniform(0.7, 1.6)),
             "mean": float(rng.normal())} for l in range(m)]


def _inp(lmm, x, m, by_features):
    return (lmm.MOInputIsotopicByFeatures if by_features else lmm.MOInputIsotopicByOutputs)(x, m)


# ---------------------------------------------------------------------------------------------------
# (a) cov(f, x, y)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d", [1, 2])
@pytest.mark.parametrize("n,n2", [(4, 3), (70, 130), (200, 65)])
def test_mogp_cross_cov_prior(lmm, n, n2, d):
    """reference src/independent_mogp.jl:66-71, :184-215 on prior latents; relation test/independent_mogp.jl:140-141:
    cov(f, x, x') == cov(GP(LinearMixingModelKernel(kernels, I)), x, x')."""
    rng = np.random.default_rng(31 * n + n2 + d)
    m = 3
    gps = _gps(m, rng)
    x = rng.uniform(0, 6, n) if d == 1 else rng.uniform(0, 3, (d, n))
    y = rng.uniform(0, 6, n2) if d == 1 else rng.uniform(0, 3, (d, n2))
    f = _model(lmm, gps)
    naive = O.naive_cov(gps, np.eye(m), x, y)                      # by-outputs rows and columns
    for xf in (False, True):
        for yf in (False, True):
            got = lmm.cov(f, _inp(lmm, x, m, xf), _inp(lmm, y, m, yf))
            ref = O.mogp_cross_cov(gps, x, y, xf, yf)
            assert got.shape == (m * n, m * n2)
            np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-14)
            ri = O.reorder_indices_outputs_to_features(n, m) if xf else np.arange(m * n)
            ci = O.reorder_indices_outputs_to_features(n2, m) if yf else np.arange(m * n2)
            np.testing.assert_allclose(got, naive[np.ix_(ri, ci)], rtol=1e-12, atol=1e-14)
    # cov(f, x) = cov(f, x, x): reference src/independent_mogp.jl:60-63
    np.testing.assert_allclose(lmm.cov(f, _inp(lmm, x, m, False)), O.mogp_cov(gps, x), rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("n,n2", [(9, 70), (130, 40)])
def test_mogp_cross_cov_posterior(lmm, n, n2):
    """The same on PosteriorGP latents (posterior(f(x0, s2), y0): src/independent_mogp.jl:119-126; the posterior's latents answer
    cov(f_l, x, y) = K(x, y) - A_x' A_y), including a sequentially conditioned posterior and the latents of a posterior OILMM."""
    rng = np.random.default_rng(77 + n)
    m, n0, s2 = 3, 150, 0.2
    gps = _gps(m, rng)
    x0 = np.sort(rng.uniform(0, 8, n0))
    y0 = rng.standard_normal(n0 * m)
    x, y = rng.uniform(0, 8, n), rng.uniform(0, 8, n2)
    f = _model(lmm, gps)
    fp = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x0, m), s2), y0)
    po = O.mogp_posterior(gps, x0, s2, y0)
    for xf, yf in [(False, False), (True, False), (False, True), (True, True)]:
        got = lmm.cov(fp, _inp(lmm, x, m, xf), _inp(lmm, y, m, yf))
        np.testing.assert_allclose(got, O.mogp_cross_cov(po, x, y, xf, yf), rtol=1e-9, atol=1e-11)
    # symmetric use: cov(f, x, x) is the block-diagonal posterior covariance the existing cov(fx) path serves (without noise)
    cxx = lmm.cov(fp, lmm.MOInputIsotopicByOutputs(x, m))
    np.testing.assert_allclose(cxx, O.mogp_cov(po, x), rtol=1e-9, atol=1e-11)
    # sequential conditioning: posterior(fp(x1, s2b), y1)
    n1, s2b = 40, 0.3
    x1 = np.sort(rng.uniform(0, 8, n1)); y1 = rng.standard_normal(n1 * m)
    fp2 = lmm.posterior(fp(lmm.MOInputIsotopicByOutputs(x1, m), s2b), y1)
    po2 = [O.gp_posterior(g, x1, s2b, y1.reshape(m, n1)[l]) for l, g in enumerate(po)]
    np.testing.assert_allclose(lmm.cov(fp2, _inp(lmm, x, m, False), _inp(lmm, y, m, True)),
                               O.mogp_cross_cov(po2, x, y, False, True), rtol=1e-8, atol=1e-10)
    # latents of a posterior OILMM: get_latent_gp(posterior(fx, y)) (reference src/ilmm.jl:39 on src/oilmm.jl:133)
    p = 4
    U, _ = np.linalg.qr(rng.standard_normal((p, m)))
    S = np.linspace(1.5, 0.9, m)
    yo = rng.standard_normal(n0 * p)
    post = lmm.posterior(lmm.ILMM(f, lmm.Orthogonal(U, S))(lmm.MOInputIsotopicByOutputs(x0, p), s2), yo)
    lat = O.oilmm_posterior(gps, U, S, x0, s2, yo)
    np.testing.assert_allclose(lmm.cov(lmm.get_latent_gp(post), _inp(lmm, x, m, False), _inp(lmm, y, m, False)),
                               O.mogp_cross_cov(lat, x, y), rtol=1e-9, atol=1e-11)


def test_mogp_cross_cov_shards_and_errors(lmm):
    """Shards fill disjoint blocks and sum to the whole; wrong out_dim raises the reference's error text (src/ilmm.jl:52 wording)."""
    from lmm_amd import _lib as L
    rng = np.random.default_rng(5)
    m, n, n2 = 4, 33, 21
    gps = _gps(m, rng)
    x, y = rng.uniform(0, 5, n), rng.uniform(0, 5, n2)
    lib = lmm.load()
    acc = np.zeros((m * n2, m * n))
    for l0, l1 in [(0, 1), (1, 3), (3, 4)]:
        out = np.empty((m * n) * (m * n2))
        L.check(lib.lmm_mogp_cross_cov(None, L.gps_array(gps), m, l0, l1, L.Arr(x).ptr, 1, n, 0, L.Arr(y).ptr, n2, 1,
                                       L.Arr(out, True).ptr))
        acc += out.reshape(m * n2, m * n)
    np.testing.assert_allclose(acc.T, O.mogp_cross_cov(gps, x, y, False, True), rtol=1e-12, atol=1e-14)
    with pytest.raises(RuntimeError, match="out dim of x != out dim of f."):
        lmm.cov(_model(lmm, gps), lmm.MOInputIsotopicByOutputs(x, m + 1), lmm.MOInputIsotopicByOutputs(y, m))


# ---------------------------------------------------------------------------------------------------
# (b) the headline value
# ---------------------------------------------------------------------------------------------------
def test_c2_full_value_equals_sum_of_eight_shares(lmm):
    """bench.py's N = 1 evaluation of BASELINE configs[2] (the number in BENCH_rNN.json) against the sum of the eight per-rank partials
    of the 8-GPU job -- two different launch plans of the same arithmetic (reference src/oilmm.jl:79-93: sum(lmls) + regulariser)."""
    import torch
    from lmm_amd.workloads import synthetic_problem
    m, p, n, s2 = 32, 64, 16384, 0.1
    P = synthetic_problem(m, p, n, "matern52", True, s2=s2, seed=0)
    fs = lmm.independent_mogp([lmm.GP(lmm.Matern52Kernel()) for _ in range(m)])
    H = lmm.Orthogonal(P["U"], P["S"])
    xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
    xin = lmm.MOInputIsotopicByOutputs(xd, p)
    whole = lmm.logpdf(lmm.ILMM(fs, H)(xin, s2), yd, True)
    parts = [lmm.logpdf(lmm.ILMM(fs, H, shard=lmm.latent_shard(m, r, 8))(xin, s2), yd, r == 0) for r in range(8)]
    assert np.isfinite(whole)
    assert abs(sum(parts) - whole) <= 1e-12 * abs(whole), (whole, sum(parts))
    # anchor: latent 0's log marginal likelihood against host LAPACK (same check as test_c2_share_properties, on THIS plan's inputs)
    one = lmm.logpdf(lmm.ILMM(fs, H, shard=(0, 1))(xin, s2), yd, False)
    T, ST = O.project_orthogonal(P["U"], P["S"], s2)
    ref = O.gp_logpdf(P["gps"][0], P["x"], ST[0], T[0] @ O.reshape_y(P["y"], n))
    assert abs(one - ref) <= 1e-9 * abs(ref), (one, ref)
    lmm.load().lmm_release_cached_memory()


# ---------------------------------------------------------------------------------------------------
# (c) flag-epoch wrap-around
# ---------------------------------------------------------------------------------------------------
def test_flag_epoch_wraparound_clear(lmm):
    """The dependency flags of the dataflow kernels are tagged with a 26-bit launch epoch and never reset; when the epoch wraps every
    persistent flag word is cleared (lmm_kernels.hip next_flag_epoch).  Set the epoch just below the wrap and evaluate across it."""
    lib = lmm.load()
    rng = np.random.default_rng(11)
    m, p, n, s2 = 4, 6, 600, 0.1
    P = O.synthetic_problem(m, p, n, "matern52", True, s2=s2, seed=4)
    f = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    fx = f(lmm.MOInputIsotopicByOutputs(P["x"], p), s2)
    ref = O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], s2, P["y"])
    before = lmm.logpdf(fx, P["y"])
    old = C.c_int()
    assert lib.lmm_dev_flag_epoch(C.c_int((1 << 26) - 2), C.byref(old)) == 0
    vals = [lmm.logpdf(fx, P["y"]) for _ in range(4)]              # epochs 2^26 - 1, then the wrap to 1, 2, 3
    now = C.c_int()
    assert lib.lmm_dev_flag_epoch(C.c_int(-1), C.byref(now)) == 0
    assert 1 <= now.value <= 16, now.value                          # the counter wrapped
    for v in [before] + vals:
        assert abs(v - ref) <= 1e-9 * abs(ref), (v, ref)
    del rng


# ---------------------------------------------------------------------------------------------------
# (d) LMM_DETERMINISTIC=1: bitwise reproducibility above 1024 columns too
# ---------------------------------------------------------------------------------------------------
def test_deterministic_mode_is_bitwise_reproducible_above_1024_columns():
    """by default the K >= 1024 update launches split some tiles along K and combine the parts with f64 atomics, so above
    1024 columns repeated evaluations agree to ~1e-13, not bitwise (test_region_kernel_is_bitwise_reproducible had to be loosened there:
    a lost hand-off that perturbs only low bits would pass it).  With LMM_DETERMINISTIC=1 no tile is split: the same shapes -- dataflow
    block columns with assistants, fused update + leaf launches, many matrices per launch -- must return identical bits every time."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json; sys.path.insert(0, %r); import numpy as np, torch, lmm_amd; from lmm_amd.workloads import synthetic_problem as sp; "
            "lmm_amd.init(0); out = {}\n"
            "for n, m in [(1100, 36), (2048, 16), (3072, 8), (4096, 4)]:\n"
            "    P = sp(m, 2 * m, n, 'matern52', True, s2=0.1, seed=n + m)\n"
            "    fx = lmm_amd.ILMM(lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)]), lmm_amd.Orthogonal(P['U'], P['S']))("
            "lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P['x']).cuda(), 2 * m), 0.1)\n"
            "    yd = torch.from_numpy(P['y']).cuda()\n"
            "    out['%%d,%%d' %% (n, m)] = sorted({lmm_amd.logpdf(fx, yd).hex() for _ in range(16)})\n"
            "print('RESULT' + json.dumps(out))") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LMM_DETERMINISTIC="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    vals = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1][len("RESULT"):])
    for shape, v in vals.items():
        assert len(v) == 1, (shape, v)
        assert np.isfinite(float.fromhex(v[0])), (shape, v)


def test_mogp_cross_cov_fp32_mode(lmm):
    """cov(f, x, y) under lmm_set_compute_dtype(LMM_F32): Float32 cross-Gram / solve blocks, Float64 output; stated tolerance as for the
    other covariances of that mode (entries within 5e-5 of the largest, include/lmm_hip.h)."""
    rng = np.random.default_rng(41)
    m, n0, n, n2, s2 = 2, 300, 70, 45, 0.2
    gps = _gps(m, rng)
    x0 = np.sort(rng.uniform(0, 8, n0)); y0 = rng.standard_normal(n0 * m)
    x, y = rng.uniform(0, 8, n), rng.uniform(0, 8, n2)
    lmm.set_compute_dtype("f32")
    try:
        f = _model(lmm, gps)
        got_prior = lmm.cov(f, _inp(lmm, x, m, False), _inp(lmm, y, m, True))
        fp = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x0, m), s2), y0)
        got_post = lmm.cov(fp, _inp(lmm, x, m, True), _inp(lmm, y, m, False))
    finally:
        lmm.set_compute_dtype("f64")
    ref_prior = O.mogp_cross_cov(gps, x, y, False, True)
    ref_post = O.mogp_cross_cov(O.mogp_posterior(gps, x0, s2, y0), x, y, True, False)
    assert np.max(np.abs(got_prior - ref_prior)) <= 5e-6 * np.max(np.abs(ref_prior))      # kappa is computed in Float64, stored Float32
    assert np.max(np.abs(got_post - ref_post)) <= 5e-5 * np.max(np.abs(ref_prior))


# ---------------------------------------------------------------------------------------------------
# gradient of the predictive logpdf after SEQUENTIAL conditioning with a DIFFERENT noise variance per batch
# (lmm_oilmm_post_logpdf_grad_seq / lmm_ilmm_post_logpdf_grad_seq: one noise block per conditioning batch + one for the test points).
# The reference differentiates logpdf(posterior(posterior(f(x1, s1), y1)(x2, s2), y2)(xs, s2s), ys) with Zygote (src/oilmm.jl:116-134,
# src/ilmm.jl:184-198 composed); here against central finite differences of the oracle's step-by-step conditioning.
# ---------------------------------------------------------------------------------------------------
def _fd(f, h=1e-6):
    return (f(h) - f(-h)) / (2.0 * h)


def _check_seq_common(G, F, gps, ys, ybatches, rel, ab):
    ns_p = len(ys)
    for k in [0, ns_p // 2 + 1, ns_p - 1]:
        e = np.zeros(ns_p); e[k] = 1.0
        assert G["y"][k] == pytest.approx(_fd(lambda t: F(ys=ys + t * e)), rel=rel, abs=ab)
    assert isinstance(G["y_train"], list) and [len(g) for g in G["y_train"]] == [len(yb) for yb in ybatches]
    for b, yb in enumerate(ybatches):
        for k in [1, len(yb) - 2]:
            e = np.zeros(len(yb)); e[k] = 1.0
            def Fb(t, b=b, e=e):
                yy = [np.array(v) for v in ybatches]; yy[b] = yy[b] + t * e
                return F(ybs=yy)
            assert G["y_train"][b][k] == pytest.approx(_fd(Fb), rel=rel, abs=ab), (b, k)
    for l in range(len(gps)):
        for key in ("variance", "lengthscale", "mean"):
            def f1(t, l=l, key=key):
                g2 = [dict(g) for g in gps]; g2[l][key] += t
                return F(gps=g2)
            assert G["gps"][l][key] == pytest.approx(_fd(f1), rel=rel, abs=ab), (l, key)


@pytest.mark.parametrize("sizes,noises", [((40, 33), (0.1, 0.3)), ((70, 25, 48), (0.25, 0.1, 0.4)), ((20, 30, 25), (0.2, 0.2, 0.35))])
def test_oilmm_gradient_after_sequential_conditioning_with_per_batch_noise(lmm, sizes, noises):
    rng = np.random.default_rng(501 + len(sizes))
    ns, p, m = 11, 4, 3
    xb = [np.sort(rng.uniform(0, 8, nb)) for nb in sizes]
    xs = np.sort(rng.uniform(0, 8, ns))
    gps = _gps(m, rng)
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    yb = [rng.standard_normal(nb * p) for nb in sizes]
    ys = rng.standard_normal(ns * p)
    s2s = 0.15

    def F(gps=gps, S=S, U=U, s2b=noises, s2s=s2s, ybs=yb, ys=ys):
        post = gps
        for x_, s_, y_ in zip(xb, s2b, ybs):
            post = O.oilmm_posterior(post, U, S, x_, s_, y_)
        return O.oilmm_logpdf(post, U, S, xs, s2s, ys)

    f = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
    po = f
    for x_, s_, y_ in zip(xb, noises, yb):
        po = lmm.posterior(po(lmm.MOInputIsotopicByOutputs(x_, p), s_), y_)
    fxs = po(lmm.MOInputIsotopicByOutputs(xs, p), s2s)
    G = lmm.logpdf_and_gradient(fxs, ys)
    assert G["value"] == pytest.approx(F(), rel=1e-8)
    assert G["value"] == pytest.approx(lmm.logpdf(fxs, ys), rel=1e-8)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=2e-5, abs=1e-6)
    per_batch = []
    for b in range(len(sizes)):
        def Fs(t, b=b):
            s = list(noises); s[b] += t
            return F(s2b=s)
        per_batch.append(_fd(Fs))
    assert isinstance(G["sigma2_train"], list) and len(G["sigma2_train"]) == len(sizes)
    assert G["sigma2_train"] == pytest.approx(per_batch, rel=2e-5, abs=1e-6)
    _check_seq_common(G, F, gps, ys, yb, 2e-5, 1e-6)
    e = np.zeros(m); e[1] = 1.0
    assert G["S"][1] == pytest.approx(_fd(lambda t: F(S=S + t * e)), rel=2e-5, abs=1e-6)
    E = np.zeros((p, m)); E[2, 0] = 1.0
    # (U as an unconstrained matrix, as Zygote treats the field: the oracle's T = S^-1/2 U' uses U as given)
    assert G["U"][2, 0] == pytest.approx(_fd(lambda t: F(U=U + t * E)), rel=5e-5, abs=1e-5)


def test_mogp_gradient_after_sequential_conditioning_with_per_batch_noise(lmm):
    rng = np.random.default_rng(511)
    sizes, noises, ns, m = (45, 80), (0.3, 0.12), 13, 3
    xb = [np.sort(rng.uniform(0, 8, nb)) for nb in sizes]
    xs = np.sort(rng.uniform(0, 8, ns))
    gps = _gps(m, rng)
    yb = [rng.standard_normal(nb * m) for nb in sizes]
    ys = rng.standard_normal(ns * m)
    s2s = 0.2

    def F(gps=gps, s2b=noises, s2s=s2s, ybs=yb, ys=ys):
        post = gps
        for x_, s_, y_ in zip(xb, s2b, ybs):
            post = O.mogp_posterior(post, x_, s_, y_)
        return O.mogp_logpdf(post, xs, s2s, ys)

    po = _model(lmm, gps)
    for x_, s_, y_ in zip(xb, noises, yb):
        po = lmm.posterior(po(lmm.MOInputIsotopicByOutputs(x_, m), s_), y_)
    G = lmm.logpdf_and_gradient(po(lmm.MOInputIsotopicByOutputs(xs, m), s2s), ys)
    assert G["value"] == pytest.approx(F(), rel=1e-8)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=2e-5, abs=1e-6)
    assert G["sigma2_train"] == pytest.approx([_fd(lambda t: F(s2b=(noises[0] + t, noises[1]))),
                                               _fd(lambda t: F(s2b=(noises[0], noises[1] + t)))], rel=2e-5, abs=1e-6)
    _check_seq_common(G, F, gps, ys, yb, 2e-5, 1e-6)


def test_dense_ilmm_gradient_after_sequential_conditioning_with_per_batch_noise(lmm):
    rng = np.random.default_rng(521)
    sizes, noises, ns, p, m = (40, 33, 21), (0.3, 0.1, 0.2), 11, 4, 3
    xb = [np.sort(rng.uniform(0, 8, nb)) for nb in sizes]
    xs = np.sort(rng.uniform(0, 8, ns))
    gps = _gps(m, rng)
    H = rng.uniform(0.2, 1.0, size=(p, m))
    yb = [rng.standard_normal(nb * p) for nb in sizes]
    ys = rng.standard_normal(ns * p)
    s2s = 0.25

    def F(gps=gps, H=H, s2b=noises, s2s=s2s, ybs=yb, ys=ys):
        post = O.ilmm_posterior(gps, H, xb[0], s2b[0], ybs[0])
        for x_, s_, y_ in zip(xb[1:], s2b[1:], ybs[1:]):
            post = O.ilmm_posterior_condition(post, H, x_, s_, y_)
        return O.ilmm_logpdf(post, H, xs, s2s, ys)

    po = lmm.ILMM(_model(lmm, gps), H)
    for x_, s_, y_ in zip(xb, noises, yb):
        po = lmm.posterior(po(lmm.MOInputIsotopicByOutputs(x_, p), s_), y_)
    fxs = po(lmm.MOInputIsotopicByOutputs(xs, p), s2s)
    G = lmm.logpdf_and_gradient(fxs, ys)
    assert G["value"] == pytest.approx(F(), rel=1e-7)
    assert G["value"] == pytest.approx(lmm.logpdf(fxs, ys), rel=1e-7)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=5e-5, abs=1e-5)
    per_batch = []
    for b in range(len(sizes)):
        def Fs(t, b=b):
            s = list(noises); s[b] += t
            return F(s2b=s)
        per_batch.append(_fd(Fs))
    assert G["sigma2_train"] == pytest.approx(per_batch, rel=5e-5, abs=1e-5)
    _check_seq_common(G, F, gps, ys, yb, 5e-5, 1e-5)
    E = np.zeros((p, m)); E[1, 2] = 1.0
    assert G["H"][1, 2] == pytest.approx(_fd(lambda t: F(H=H + t * E)), rel=5e-5, abs=1e-5)


def test_post_logpdf_grad_seq_argument_checks(lmm):
    lib = lmm.load()
    rng = np.random.default_rng(531)
    n, ns, p, m = 30, 7, 3, 2
    x, xs = np.sort(rng.uniform(0, 5, n)), np.sort(rng.uniform(0, 5, ns))
    y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    Uc = np.asfortranarray(U).ravel(order="F").copy()
    from lmm_amd import _lib as L
    gps = L.gps_array(_gps(m, rng))
    val = C.c_double()

    def call(sizes, noises):
        k = len(sizes)
        return lib.lmm_oilmm_post_logpdf_grad_seq(x.ctypes.data_as(C.c_void_p), 1, n, (C.c_int * k)(*sizes), (C.c_double * k)(*noises), k,
                                                  y.ctypes.data_as(C.c_void_p), xs.ctypes.data_as(C.c_void_p), ns,
                                                  ys.ctypes.data_as(C.c_void_p), p, Uc.ctypes.data_as(C.c_void_p),
                                                  S.ctypes.data_as(C.c_void_p), m, C.c_double(0.1), gps, 0, m, 1, C.byref(val),
                                                  None, None, None, None, None, None, None)

    assert call([20, 10], [0.1, 0.2]) == 0
    assert call([20, 11], [0.1, 0.2]) != 0                     # sizes do not add up to n
    assert call([20, 10], [0.1, 0.0]) != 0                     # sigma2 must be > 0
    assert call([4, 4, 4, 4, 4, 4, 4, 2], [0.1] * 8) != 0      # more than 7 batches
    assert b"7 conditioning batches" in lib.lmm_last_error_string()


def test_per_batch_noise_gradient_fp32_mode(lmm):
    """The three-noise-block core under lmm_set_compute_dtype(LMM_F32) (Float32 factor, inverse and per-block traces read from it; Float64
    reductions) against the Float64 mode of the same call: the tolerances stated for that mode's gradients in include/lmm_hip.h."""
    rng = np.random.default_rng(541)
    sizes, noises, ns, p, m = (300, 260), (0.1, 0.25), 90, 4, 3
    xb = [np.sort(rng.uniform(0, 12, nb)) for nb in sizes]
    xs = np.sort(rng.uniform(0, 12, ns))
    gps = _gps(m, rng)
    U, S, _ = np.linalg.svd(rng.uniform(size=(p, m)), full_matrices=False)
    yb = [rng.standard_normal(nb * p) for nb in sizes]
    ys = rng.standard_normal(ns * p)

    def run():
        po = lmm.ILMM(_model(lmm, gps), lmm.Orthogonal(U, S))
        for x_, s_, y_ in zip(xb, noises, yb):
            po = lmm.posterior(po(lmm.MOInputIsotopicByOutputs(x_, p), s_), y_)
        return lmm.logpdf_and_gradient(po(lmm.MOInputIsotopicByOutputs(xs, p), 0.15), ys)

    G64 = run()
    lmm.set_compute_dtype("f32")
    try:
        G32 = run()
    finally:
        lmm.set_compute_dtype("f64")
    assert G32["value"] == pytest.approx(G64["value"], rel=2e-5)
    assert np.max(np.abs(np.asarray(G32["y"]) - np.asarray(G64["y"]))) <= 1e-4 * np.max(np.abs(G64["y"]))
    for b in range(2):
        assert np.max(np.abs(G32["y_train"][b] - G64["y_train"][b])) <= 5e-4 * np.max(np.abs(G64["y_train"][b]))
    assert G32["sigma2"] == pytest.approx(G64["sigma2"], rel=1e-4)
    assert G32["sigma2_train"] == pytest.approx(G64["sigma2_train"], rel=2e-3, abs=1e-2)
    for l in range(m):
        for key in ("variance", "lengthscale", "mean"):
            assert G32["gps"][l][key] == pytest.approx(G64["gps"][l][key], rel=2e-3, abs=1e-2), (l, key)


# ---------------------------------------------------------------------------------------------------
# strict forward progress (lmm_set_strict_progress, default ON): potrf_region_kernel's workgroups claim their task at entry instead of
# reading it from blockIdx.x (and the fused update launches are not used).  The tasks are the same, only which workgroup runs which
# changes: logpdf is bit-identical to the index-order mode on the one-launch region path (n <= 1024) and agrees to that mode's own
# run-to-run noise (split-K atomics) under the recursion.  Batch sizes: 3, 5, 20 take the global ticket; 2, 8, 16 the per-XCD claim.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n", [(3, 200), (20, 552), (8, 1024), (8, 2048), (5, 3000), (16, 4096), (2, 6500)])
def test_strict_progress_mode_gives_the_same_values(lmm, m, n):
    from lmm_amd.workloads import synthetic_problem
    p = m + 2
    P = synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=7)
    f = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))
    xin = lmm.MOInputIsotopicByOutputs(P["x"], p)
    xs = lmm.MOInputIsotopicByOutputs(np.linspace(0.0, 1.0, 37) * float(np.max(P["x"])), p)

    def run():
        v = [lmm.logpdf(f(xin, 0.1), P["y"]) for _ in range(3)]
        mu, var = lmm.mean_and_var(lmm.posterior(f(xin, 0.1), P["y"])(xs, 0.1))
        return v, np.asarray(mu), np.asarray(var)

    assert lmm.get_strict_progress()                   # the default
    v1, mu1, var1 = run()
    lmm.set_strict_progress(False)
    try:
        assert not lmm.get_strict_progress()
        v0, mu0, var0 = run()
    finally:
        lmm.set_strict_progress(True)
    if n <= 1024:                        # one region launch per batch: no split-K atomics anywhere, bitwise reproducible in both modes
        assert v0[0] == v0[1] == v0[2] and v1 == v0
        np.testing.assert_allclose(mu1, mu0, rtol=1e-11, atol=1e-12)      # (the cross-solve behind the marginals combines partial products
        np.testing.assert_allclose(var1, var0, rtol=1e-11, atol=1e-12)    #  with atomics in either mode)
        assert v0[0] == pytest.approx(O.oilmm_logpdf(P["gps"], P["U"], P["S"], P["x"], 0.1, P["y"]), rel=1e-9)
    else:                                # the update launches' split-K tails are combined with f64 atomics: run-to-run noise of the default mode
        assert v1 == pytest.approx(v0, rel=1e-12)
        np.testing.assert_allclose(mu1, mu0, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(var1, var0, rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("m,n", [(3, 200), (20, 552), (32, 1024), (8, 2048), (16, 4096), (4, 9000)])
def test_region_kernel_completes_when_workgroups_start_in_the_wrong_order(lmm, m, n):
    """lmm_dev_claim_scramble: every workgroup of the strict-progress region kernel asks for the task index that a REVERSED dispatch order
    would give it.  Launches that fit the device run with every role on the 'wrong' workgroup; larger ones (32 x 1024: 32 x 30 workgroups,
    16 x 4096, 4 x 9000) start with all slots held by workgroups whose turn cannot come, which time out after 200 us and take the next
    free index.  The evaluation must complete (no LMM_ERR_HIP) with the same value."""
    from lmm_amd.workloads import synthetic_problem
    lib = lmm.load()
    p = m + 1
    P = synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=11)
    fx = lmm.ILMM(_model(lmm, P["gps"]), lmm.Orthogonal(P["U"], P["S"]))(lmm.MOInputIsotopicByOutputs(P["x"], p), 0.1)
    v0 = lmm.logpdf(fx, P["y"])
    assert lib.lmm_dev_claim_scramble(1) == 0
    try:
        v1 = [lmm.logpdf(fx, P["y"]) for _ in range(3)]
    finally:
        assert lib.lmm_dev_claim_scramble(0) == 0
    v2 = lmm.logpdf(fx, P["y"])
    if n <= 1024:
        assert v1 == [v0, v0, v0] and v2 == v0
    else:
        assert v1 == pytest.approx([v0] * 3, rel=1e-12) and v2 == pytest.approx(v0, rel=1e-12)


# ---------------------------------------------------------------------------------------------------
# gradient of logpdf(get_latent_gp(posterior(dense-H ILMM ...))(xs, s2s), zs): the coupled latent PosteriorGP of a dense-H posterior
# (reference src/ilmm.jl:39 on the ILMM of :196-197) -- lmm_ilmm_post_latent_logpdf_grad_seq.  Oracle: the generic Gaussian on
# O._ilmm_latent_joint of O.ilmm_posterior / ilmm_posterior_condition, central finite differences.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sizes,noises", [((70,), (0.1,)), ((40, 33), (0.3, 0.12))])
def test_gradient_of_the_latent_view_of_a_dense_posterior(lmm, sizes, noises):
    rng = np.random.default_rng(601 + len(sizes))
    ns, p, m = 9, 5, 3
    xb = [np.sort(rng.uniform(0, 6, nb)) for nb in sizes]
    xs = np.sort(rng.uniform(0, 6, ns))
    gps = _gps(m, rng)
    H = rng.uniform(0.2, 1.0, size=(p, m))
    yb = [rng.standard_normal(nb * p) for nb in sizes]
    zs = rng.standard_normal(ns * m)
    s2s = 0.07

    def F(gps=gps, H=H, s2b=noises, s2s=s2s, ybs=yb, zs=zs):
        post = O.ilmm_posterior(gps, H, xb[0], s2b[0], ybs[0])
        for x_, s_, y_ in zip(xb[1:], s2b[1:], ybs[1:]):
            post = O.ilmm_posterior_condition(post, H, x_, s_, y_)
        mo, Co = O._ilmm_latent_joint(post, xs)
        return O.gaussian_logpdf(mo, Co + s2s * np.eye(m * ns), zs)

    po = lmm.ILMM(_model(lmm, gps), H)
    for x_, s_, y_ in zip(xb, noises, yb):
        po = lmm.posterior(po(lmm.MOInputIsotopicByOutputs(x_, p), s_), y_)
    flx = lmm.get_latent_gp(po)(lmm.MOInputIsotopicByOutputs(xs, m), s2s)
    G = lmm.logpdf_and_gradient(flx, zs)
    assert G["value"] == pytest.approx(F(), rel=1e-7)
    assert G["value"] == pytest.approx(lmm.logpdf(flx, zs), rel=1e-7)
    assert G["sigma2"] == pytest.approx(_fd(lambda t: F(s2s=s2s + t)), rel=5e-5, abs=1e-5)
    per_batch = []
    for b in range(len(sizes)):
        def Fs(t, b=b):
            s = list(noises); s[b] += t
            return F(s2b=s)
        per_batch.append(_fd(Fs))
    got = G["sigma2_train"] if isinstance(G["sigma2_train"], list) else [G["sigma2_train"]]
    assert got == pytest.approx(per_batch, rel=5e-5, abs=1e-5)
    for k in [0, ns + 4, ns * m - 1]:
        e = np.zeros(ns * m); e[k] = 1.0
        assert G["y"][k] == pytest.approx(_fd(lambda t: F(zs=zs + t * e)), rel=5e-5, abs=1e-5)
    ytr = G["y_train"] if isinstance(G["y_train"], list) else [G["y_train"]]
    for b, yv in enumerate(yb):
        for k in [2, len(yv) - 3]:
            e = np.zeros(len(yv)); e[k] = 1.0
            def Fb(t, b=b, e=e):
                yy = [np.array(v) for v in yb]; yy[b] = yy[b] + t * e
                return F(ybs=yy)
            assert np.asarray(ytr[b])[k] == pytest.approx(_fd(Fb), rel=5e-5, abs=1e-5), (b, k)
    for l in range(m):
        for key in ("variance", "lengthscale", "mean"):
            def f1(t, l=l, key=key):
                g2 = [dict(g) for g in gps]; g2[l][key] += t
                return F(gps=g2)
            assert G["gps"][l][key] == pytest.approx(_fd(f1), rel=5e-5, abs=1e-5), (l, key)
    for (o, l) in [(1, 2), (4, 0)]:
        E = np.zeros((p, m)); E[o, l] = 1.0
        assert G["H"][o, l] == pytest.approx(_fd(lambda t: F(H=H + t * E)), rel=5e-5, abs=1e-5)
    # conditioned ON latent observations: refused
    with pytest.raises(NotImplementedError):
        lmm.logpdf_and_gradient(lmm.posterior(flx, zs)(lmm.MOInputIsotopicByOutputs(xs, m), s2s), zs)
