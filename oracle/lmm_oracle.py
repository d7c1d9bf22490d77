"""CPU oracle for the ILMM/OILMM inference hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a Float64 NumPy/SciPy restatement of the reference algorithm
(LinearMixingModels.jl 0.1.11, /root/reference/src/*.jl) and of the generic
AbstractGPs / KernelFunctions arithmetic those files reach (SURVEY.md section 2,
"External arithmetic").  It is the *checker*: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path
(linearmixingmodels.jl_amd/) never imports, calls or falls back to anything here.

Parity pin: Julia is absent from the build container, so the
reference itself cannot be run.  The oracle is pinned by
  (1) the reference's own *relational* tests -- structured path == naive dense
      multivariate normal (test/ilmm.jl:10-14,23-26, test/oilmm.jl:10-14,23-26,
      test/independent_mogp.jl:40-43,53-60) -- restated in tests/test_oracle.py, and
  (2) the six literal Matern52 Gram / Cholesky numbers printed in the reference
      notebook (examples/oilmm_and_ilmm.ipynb:616), held in
      tests/golden/notebook_literals.json.
The Julia-RNG-seeded values in the notebook are not reproducible offline and are
not used.

Conventions (reference: src/ilmm.jl:43, KernelFunctions MOInputIsotopicByOutputs):
  * x is (n,) or (d, n) (ColVecs layout); y is the length n*p "by-outputs" vector:
    y[o*n:(o+1)*n] are the n observations of output o.
  * H is p x m.  Orthogonal H is given as (U p x m, S (m,)) with H = U*sqrt(S)
    (src/orthogonal_matrix.jl:11-30).
  * A latent GP is a dict: {"kind": "se"|"matern32"|"matern52", "variance": s2,
    "lengthscale": ell, "mean": c}  (ScaledKernel(s2) * kernel o ScaleTransform(1/ell),
    ConstMean(c)).  A *posterior* latent additionally carries "post": {"x", "alpha",
    "L"} (AbstractGPs PosteriorGP data: alpha = C\\delta, C = cholesky(K + Sigma)).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla

LOG2PI = math.log(2.0 * math.pi)

# Numerics constants hard-coded in the reference.
JITTER_PROJECT = 1e-9      # src/ilmm.jl:63
JITTER_ILMM_RAND = 1e-12   # src/ilmm.jl:84
JITTER_DEFAULT = 1e-18     # AbstractGPs default f(x) == f(x, 1e-18): src/oilmm.jl:47,61; src/ilmm.jl:115

KINDS = {"se": 0, "matern32": 1, "matern52": 2}


# ----------------------------------------------------------------------------------------
# KernelFunctions.jl / Distances.jl restatement (SURVEY.md section 2, row "KernelFunctions kernels")
# ----------------------------------------------------------------------------------------
def _as_cols(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    return x[None, :] if x.ndim == 1 else x


def pairwise_dist(x: np.ndarray, x2: Optional[np.ndarray] = None) -> np.ndarray:
    """Euclidean distances by direct differences (the device kernels do the same).

    The reference goes through Distances.jl's ||a||^2+||b||^2-2ab form (abs. error
    ~ eps*||x||^2 in d^2, visible at ipynb:616); SE/Matern32/Matern52 have zero slope at
    d = 0, so the two agree to ~1e-13 in the Gram entries (caveat).
    """
    a = _as_cols(x)
    b = a if x2 is None else _as_cols(x2)
    d2 = np.zeros((a.shape[1], b.shape[1]))
    for k in range(a.shape[0]):
        diff = a[k][:, None] - b[k][None, :]
        d2 += diff * diff
    return np.sqrt(d2)


def kernel_eval(kind: str, variance: float, lengthscale: float, r: np.ndarray) -> np.ndarray:
    """kappa(r) for the three stationary kernels the reference's tests/notebook use
    (test/ilmm.jl:46,52,76; test/oilmm.jl:47,54,61; ipynb:81,1060)."""
    d = r / lengthscale
    if kind == "se":
        k = np.exp(-0.5 * d * d)
    elif kind == "matern32":
        s = math.sqrt(3.0) * d
        k = (1.0 + s) * np.exp(-s)
    elif kind == "matern52":
        s = math.sqrt(5.0) * d
        k = (1.0 + s + 5.0 * d * d / 3.0) * np.exp(-s)
    else:
        raise ValueError(f"unknown kernel kind {kind!r}")
    return variance * k


def kernelmatrix(gp: Dict, x: np.ndarray, x2: Optional[np.ndarray] = None) -> np.ndarray:
    return kernel_eval(gp["kind"], gp.get("variance", 1.0), gp.get("lengthscale", 1.0),
                       pairwise_dist(x, x2))


def npoints(x: np.ndarray) -> int:
    return _as_cols(x).shape[1]


# ----------------------------------------------------------------------------------------
# AbstractGPs.jl generic single-output GP arithmetic (first five rows)
# ----------------------------------------------------------------------------------------
def gp_mean_cov(gp: Dict, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """mean(f, x), cov(f, x) for a prior GP or a PosteriorGP.

    PosteriorGP: mean = m(x*) + K(x*,x) alpha;  cov = K(x*,x*) - A'A, A = C.U' \\ K(x,x*).
    """
    n = npoints(x)
    m = np.full(n, float(gp.get("mean", 0.0)))
    K = kernelmatrix(gp, x)
    post = gp.get("post")
    if post is not None:
        Kxs = kernelmatrix(gp, post["x"], x)                      # n_train x n*
        m = m + Kxs.T @ post["alpha"]
        A = sla.solve_triangular(post["L"], Kxs, lower=True)
        K = K - A.T @ A
    return m, K


def gp_mean_var(gp: Dict, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """mean_and_var of f(x) (no noise): var = k(x*,x*) - colsumsq(C.U' \\ K(x,x*))."""
    n = npoints(x)
    m = np.full(n, float(gp.get("mean", 0.0)))
    v = np.full(n, float(gp.get("variance", 1.0)))               # kappa(0) = variance
    post = gp.get("post")
    if post is not None:
        Kxs = kernelmatrix(gp, post["x"], x)
        m = m + Kxs.T @ post["alpha"]
        A = sla.solve_triangular(post["L"], Kxs, lower=True)
        v = v - np.sum(A * A, axis=0)
    return m, v


def gaussian_logpdf(m: np.ndarray, C: np.ndarray, y: np.ndarray) -> float:
    """AbstractGPs generic logpdf(fx, y): -(n log 2pi + logdet C + ||U' \\ (y-m)||^2)/2."""
    L = np.linalg.cholesky(C)
    z = sla.solve_triangular(L, y - m, lower=True)
    return -0.5 * (len(y) * LOG2PI + 2.0 * np.sum(np.log(np.diag(L))) + z @ z)


def gp_logpdf(gp: Dict, x: np.ndarray, noise: float, y: np.ndarray) -> float:
    m, K = gp_mean_cov(gp, x)
    return gaussian_logpdf(m, K + noise * np.eye(len(m)), y)


def gp_posterior(gp: Dict, x: np.ndarray, noise: float, y: np.ndarray) -> Dict:
    """AbstractGPs generic posterior(fx, y): delta = y - m; alpha = C \\ delta.  On a PosteriorGP (sequential
    conditioning; AbstractGPs updates the Cholesky factor) the result is the posterior of the PRIOR given both data sets,
    each with its own noise -- restated here by conditioning the prior on the concatenation."""
    prior = {k: v for k, v in gp.items() if k != "post"}
    xa = _as_cols(x)
    delta = np.asarray(y, dtype=np.float64) - float(prior.get("mean", 0.0))
    nz = np.full(xa.shape[1], float(noise))
    old = gp.get("post")
    if old is not None:
        xa = np.concatenate([_as_cols(old["x"]), xa], axis=1)
        delta = np.concatenate([old["delta"], delta])
        nz = np.concatenate([old["noise"], nz])
    xall = xa[0] if np.asarray(x).ndim == 1 else xa
    K = kernelmatrix(prior, xall)
    L = np.linalg.cholesky(K + np.diag(nz))
    alpha = sla.cho_solve((L, True), delta)
    out = dict(prior)
    out["post"] = {"x": xall, "alpha": alpha, "L": L, "delta": delta, "noise": nz}
    return out


def gp_rand(gp: Dict, x: np.ndarray, jitter: float, z: np.ndarray) -> np.ndarray:
    """AbstractGPs rand(rng, f(x, jitter)) with the standard normals z supplied by the caller:
    m + cholesky(K + jitter I).U' * z."""
    m, K = gp_mean_cov(gp, x)
    L = np.linalg.cholesky(K + jitter * np.eye(len(m)))
    return m + L @ z


# ----------------------------------------------------------------------------------------
# src/ilmm.jl helpers
# ----------------------------------------------------------------------------------------
def reshape_y(y: np.ndarray, n: int) -> np.ndarray:
    """src/ilmm.jl:43  reshape(y, N, :)'  ->  p x n."""
    return np.asarray(y, dtype=np.float64).reshape(-1, n)


def check_out_dim(p_inputs: int, H: np.ndarray) -> None:
    """src/ilmm.jl:52."""
    if p_inputs != H.shape[0]:
        raise RuntimeError("out dim of x != out dim of f.")


def orthogonal_validate(U: np.ndarray) -> None:
    """src/orthogonal_matrix.jl:21-23: isapprox(U'U, I) with Julia's default rtol=sqrt(eps)
    on the Frobenius norm."""
    m = U.shape[1]
    G = U.T @ U
    if not np.linalg.norm(G - np.eye(m)) <= math.sqrt(np.finfo(np.float64).eps) * max(
            np.linalg.norm(G), math.sqrt(m)):
        raise ValueError("`U` is not an orthogonal matrix")


def orthogonal_dense(U: np.ndarray, S: np.ndarray) -> np.ndarray:
    """src/orthogonal_matrix.jl:27-30: H = U * sqrt(S)."""
    return U * np.sqrt(S)[None, :]


def project_orthogonal(U: np.ndarray, S: np.ndarray, s2: float) -> Tuple[np.ndarray, np.ndarray]:
    """src/oilmm.jl:20-30: T = sqrt(S) \\ U' (m x p); SigmaT = diag(s2 * inv(S)) (m)."""
    return U.T / np.sqrt(S)[:, None], s2 / S


def project_dense(H: np.ndarray, s2: float) -> Tuple[np.ndarray, np.ndarray]:
    """src/ilmm.jl:61-68."""
    m = H.shape[1]
    ST_inv = H.T @ H / s2 + JITTER_PROJECT * np.eye(m)
    T = sla.cho_solve((np.linalg.cholesky(ST_inv), True), H.T / s2)
    ST = T @ (s2 * T.T)
    return T, ST


def regulariser_oilmm(U: np.ndarray, S: np.ndarray, s2: float, Y: np.ndarray) -> float:
    """src/oilmm.jl:101-113."""
    p, m = U.shape
    n = Y.shape[1]
    R = Y - U @ (U.T @ Y)
    return -(n * (np.sum(np.log(S)) + (p - m) * math.log(2.0 * math.pi * s2)) + np.sum(R * R) / s2) / 2.0


def regulariser_ilmm(H: np.ndarray, s2: float, Y: np.ndarray) -> float:
    """src/ilmm.jl:171-181."""
    p, m = H.shape
    n = Y.shape[1]
    T, ST = project_dense(H, s2)
    _, logdet = np.linalg.slogdet(ST)
    R = Y - H @ (T @ Y)
    return -(n * ((p - m) * LOG2PI + (p * math.log(s2) - logdet)) + np.sum(R * R) / s2) / 2.0


# ----------------------------------------------------------------------------------------
# src/independent_mogp.jl:39-126  (MOInputIsotopicByOutputs half)
# ----------------------------------------------------------------------------------------
def mogp_logpdf(gps: Sequence[Dict], x: np.ndarray, s2: float, y: np.ndarray) -> float:
    """src/independent_mogp.jl:74-80: sum of per-latent logpdfs with scalar noise."""
    n = npoints(x)
    Y = np.asarray(y).reshape(len(gps), n)
    return float(sum(gp_logpdf(g, x, s2, Y[l]) for l, g in enumerate(gps)))


def mogp_logpdf_diag(gps: Sequence[Dict], x: np.ndarray, noise_diag: np.ndarray, y: np.ndarray) -> float:
    """logpdf of a by-outputs IndependentMOGP FiniteGP with a general Diagonal noise (what src/independent_mogp.jl:222-229
    reaches after reorder_by_outputs, :149-159): no specialised method exists for a non-Fill Diagonal, so AbstractGPs'
    generic dense logpdf runs on cov(f, x) (block diagonal, src/independent_mogp.jl:60-63) + Diagonal(noise_diag)."""
    n = npoints(x)
    blocks = [gp_mean_cov(g, x) for g in gps]
    mean = np.concatenate([b[0] for b in blocks])
    C = sla.block_diag(*[b[1] for b in blocks]) + np.diag(np.asarray(noise_diag, dtype=float))
    return gaussian_logpdf(mean, C, np.asarray(y, dtype=float).reshape(len(gps) * n))


def mogp_posterior(gps: Sequence[Dict], x: np.ndarray, s2: float, y: np.ndarray) -> List[Dict]:
    """src/independent_mogp.jl:119-126."""
    n = npoints(x)
    Y = np.asarray(y).reshape(len(gps), n)
    return [gp_posterior(g, x, s2, Y[l]) for l, g in enumerate(gps)]


def mogp_mean_var(gps: Sequence[Dict], x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """src/independent_mogp.jl:50,55: vcat of per-latent mean / var."""
    mv = [gp_mean_var(g, x) for g in gps]
    return np.concatenate([a for a, _ in mv]), np.concatenate([b for _, b in mv])


def mogp_cov(gps: Sequence[Dict], x: np.ndarray) -> np.ndarray:
    """src/independent_mogp.jl:60-63: dense block-diagonal."""
    return sla.block_diag(*[gp_mean_cov(g, x)[1] for g in gps])


def gp_cross_cov(gp: Dict, x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """AbstractGPs cov(f, x, y) for a prior GP (kernelmatrix(k, x, y)) or a PosteriorGP
    (K(x, y) - A_x' A_y with A_z = C.U' \\ K(x_train, z))."""
    K = kernelmatrix(gp, x, y)
    post = gp.get("post")
    if post is not None:
        Ax = sla.solve_triangular(post["L"], kernelmatrix(gp, post["x"], x), lower=True)
        Ay = sla.solve_triangular(post["L"], kernelmatrix(gp, post["x"], y), lower=True)
        K = K - Ax.T @ Ay
    return K


def reorder_indices_outputs_to_features(n: int, p: int) -> np.ndarray:
    """src/independent_mogp.jl:135-139 (0-based): applied to a by-outputs vector it orders it by features."""
    return np.arange(n * p).reshape(p, n).T.reshape(-1)


def mogp_cross_cov(gps: Sequence[Dict], x: np.ndarray, y: np.ndarray, x_by_features: bool = False,
                   y_by_features: bool = False) -> np.ndarray:
    """cov(f::IndependentMOGP, x, y): src/independent_mogp.jl:66-71 (both MOInputIsotopicByOutputs: the dense block
    diagonal of the per-latent cov(f_l, x.x, y.x)); :184-215 for the by-features / mixed forms (rows and / or columns
    permuted with indices_which_reorder_outputs_to_features)."""
    C = sla.block_diag(*[gp_cross_cov(g, x, y) for g in gps])
    m = len(gps)
    if x_by_features:
        C = C[reorder_indices_outputs_to_features(npoints(x), m), :]
    if y_by_features:
        C = C[:, reorder_indices_outputs_to_features(npoints(y), m)]
    return C


def mogp_rand(gps: Sequence[Dict], x: np.ndarray, s2: float, z: np.ndarray) -> np.ndarray:
    """src/independent_mogp.jl:83-86: vcat of rand(rng, f_l(x, s2)); z is m blocks of n normals."""
    n = npoints(x)
    Z = np.asarray(z).reshape(len(gps), n)
    return np.concatenate([gp_rand(g, x, s2, Z[l]) for l, g in enumerate(gps)])


# ----------------------------------------------------------------------------------------
# src/oilmm.jl
# ----------------------------------------------------------------------------------------
def oilmm_logpdf(gps: Sequence[Dict], U: np.ndarray, S: np.ndarray, x: np.ndarray, s2: float,
                 y: np.ndarray) -> float:
    """src/oilmm.jl:79-93."""
    n = npoints(x)
    Y = reshape_y(y, n)
    check_out_dim(Y.shape[0], U)
    T, ST = project_orthogonal(U, S, s2)
    Ty = T @ Y
    lmls = [gp_logpdf(g, x, ST[l], Ty[l]) for l, g in enumerate(gps)]
    return float(sum(lmls) + regulariser_oilmm(U, S, s2, Y))


def oilmm_posterior(gps: Sequence[Dict], U: np.ndarray, S: np.ndarray, x: np.ndarray, s2: float,
                    y: np.ndarray) -> List[Dict]:
    """src/oilmm.jl:116-134 -> list of posterior latents (H is unchanged)."""
    n = npoints(x)
    Y = reshape_y(y, n)
    check_out_dim(Y.shape[0], U)
    T, ST = project_orthogonal(U, S, s2)
    Ty = T @ Y
    return [gp_posterior(g, x, ST[l], Ty[l]) for l, g in enumerate(gps)]


def oilmm_mean_var(gps: Sequence[Dict], U: np.ndarray, S: np.ndarray, x: np.ndarray,
                   s2: float) -> Tuple[np.ndarray, np.ndarray]:
    """src/oilmm.jl:57-76 (marginals(f(x)) uses the default 1e-18 jitter: oilmm.jl:61)."""
    mv = [gp_mean_var(g, x) for g in gps]
    M_lat = np.stack([a for a, _ in mv])                      # m x n
    V_lat = np.stack([b + JITTER_DEFAULT for _, b in mv])
    H = orthogonal_dense(U, S)
    M = H @ M_lat
    V = (H * H) @ V_lat + s2
    return M.reshape(-1), V.reshape(-1)                       # vec(M'), by-outputs


def oilmm_rand(gps: Sequence[Dict], U: np.ndarray, S: np.ndarray, x: np.ndarray, s2: float,
               z_lat: np.ndarray, eps: np.ndarray) -> np.ndarray:
    """src/oilmm.jl:40-54; draw order: m blocks of n latent normals, then n*p noise normals."""
    n = npoints(x)
    Z = np.asarray(z_lat).reshape(len(gps), n)
    X = np.stack([gp_rand(g, x, JITTER_DEFAULT, Z[l]) for l, g in enumerate(gps)])   # m x n
    F = (orthogonal_dense(U, S) @ X).reshape(-1)
    return F + math.sqrt(s2) * np.asarray(eps)


# ----------------------------------------------------------------------------------------
# src/ilmm.jl  (general H: dense (mn) x (mn) path)
# ----------------------------------------------------------------------------------------
def _ilmm_latent_joint(gps_or_post, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """mean_and_cov of the latent multi-output GP at MOInputIsotopicByOutputs(x, m).

    gps_or_post is either a list of (prior) latents -> block-diagonal covariance
    (src/independent_mogp.jl:60-63), or the dict produced by ilmm_posterior (a PosteriorGP over
    the IndependentMOGP with a dense (mn) x (mn) Cholesky).
    """
    if isinstance(gps_or_post, dict):
        P = gps_or_post
        gps, xt = P["gps"], P["x"]
        mean = np.concatenate([np.full(npoints(x), float(g.get("mean", 0.0))) for g in gps])
        Kss = sla.block_diag(*[kernelmatrix(g, x) for g in gps])
        Kxs = sla.block_diag(*[kernelmatrix(g, xt, x) for g in gps])      # src/independent_mogp.jl:66-71
        mean = mean + Kxs.T @ P["alpha"]
        A = sla.solve_triangular(P["L"], Kxs, lower=True)
        return mean, Kss - A.T @ A
    gps = gps_or_post
    mean = np.concatenate([gp_mean_cov(g, x)[0] for g in gps])
    return mean, mogp_cov(gps, x)


def ilmm_logpdf(latent, H: np.ndarray, x: np.ndarray, s2: float, y: np.ndarray) -> float:
    """src/ilmm.jl:150-163: one dense (mn)x(mn) Gaussian + regulariser."""
    n = npoints(x)
    Y = reshape_y(y, n)
    check_out_dim(Y.shape[0], H)
    T, ST = project_dense(H, s2)
    Yproj = (T @ Y).reshape(-1)                               # by-outputs over latents
    mean, C = _ilmm_latent_joint(latent, x)
    C = C + np.kron(ST, np.eye(n))
    return float(gaussian_logpdf(mean, C, Yproj) + regulariser_ilmm(H, s2, Y))


def ilmm_posterior(gps: Sequence[Dict], H: np.ndarray, x: np.ndarray, s2: float, y: np.ndarray) -> Dict:
    """src/ilmm.jl:184-198."""
    n = npoints(x)
    Y = reshape_y(y, n)
    check_out_dim(Y.shape[0], H)
    T, ST = project_dense(H, s2)
    Yproj = (T @ Y).reshape(-1)
    mean, C = _ilmm_latent_joint(list(gps), x)
    L = np.linalg.cholesky(C + np.kron(ST, np.eye(n)))
    alpha = sla.cho_solve((L, True), Yproj - mean)
    return {"gps": list(gps), "x": np.asarray(x, dtype=np.float64), "alpha": alpha, "L": L,
            "delta": (Yproj - mean).reshape(len(gps), n), "noise": [(ST, n)]}


def ilmm_posterior_condition(post: Dict, H: np.ndarray, x2: np.ndarray, s2: float, y2: np.ndarray) -> Dict:
    """posterior(pi(x2, s2), y2) on the dense-H posterior ILMM: src/ilmm.jl:184-198 applied to the PosteriorGP latent
    (AbstractGPs updates the Cholesky factor).  The result is the posterior of the PRIOR latents given both projected data
    sets, batch k carrying the noise SigmaT_k (x) I -- restated here by refactorising the stacked system."""
    gps = post["gps"]
    m, n2 = len(gps), npoints(x2)
    Y2 = reshape_y(y2, n2)
    check_out_dim(Y2.shape[0], H)
    T, ST = project_dense(H, s2)
    x1 = post["x"]
    xall = np.concatenate([x1, np.asarray(x2, dtype=np.float64)], axis=-1 if np.ndim(x1) == 1 else 1)
    n = npoints(xall)
    mean2 = np.array([float(g.get("mean", 0.0)) for g in gps])[:, None]
    delta = np.concatenate([post["delta"], T @ Y2 - mean2], axis=1)            # m x n
    noise = post["noise"] + [(ST, n2)]
    _, C = _ilmm_latent_joint(list(gps), xall)
    for a in range(m):
        for b in range(m):
            dvec = np.concatenate([np.full(nk, STk[a, b]) for STk, nk in noise])
            C[a * n:(a + 1) * n, b * n:(b + 1) * n] += np.diag(dvec)
    L = np.linalg.cholesky(C)
    alpha = sla.cho_solve((L, True), delta.reshape(-1))
    return {"gps": list(gps), "x": xall, "alpha": alpha, "L": L, "delta": delta, "noise": noise}


def ilmm_mean_cov(latent, H: np.ndarray, x: np.ndarray, s2: float) -> Tuple[np.ndarray, np.ndarray]:
    """src/ilmm.jl:108-119,132-139, computed as H_full Sigma_lat H_full' + s2 I (the same numbers
    as the reference's Xt_A_X(cholesky(latent_cov), H_full') without its fragile Cholesky of a
    1e-18-jittered covariance)."""
    n = npoints(x)
    mean, C = _ilmm_latent_joint(latent, x)
    C = C + JITTER_DEFAULT * np.eye(len(mean))               # f(x_mo_input) default jitter, ilmm.jl:115
    Hf = np.kron(H, np.eye(n))
    return Hf @ mean, Hf @ C @ Hf.T + s2 * np.eye(Hf.shape[0])


def ilmm_mean_var(latent, H: np.ndarray, x: np.ndarray, s2: float) -> Tuple[np.ndarray, np.ndarray]:
    """src/ilmm.jl:122-129."""
    M, C = ilmm_mean_cov(latent, H, x, s2)
    return M, np.diag(C).copy()


def ilmm_rand(gps: Sequence[Dict], H: np.ndarray, x: np.ndarray, s2: float, z_lat: np.ndarray,
              eps: np.ndarray) -> np.ndarray:
    """src/ilmm.jl:78-87 for prior latents (latent jitter 1e-12 through the IndependentMOGP
    fan-out src/independent_mogp.jl:83-86)."""
    n = npoints(x)
    lat = mogp_rand(gps, x, JITTER_ILMM_RAND, z_lat).reshape(len(gps), n)    # m x n
    return (H @ lat).reshape(-1) + math.sqrt(s2) * np.asarray(eps)


# ----------------------------------------------------------------------------------------
# Naive dense Gaussian: the right-hand side of the reference's relational tests
# (GP(LinearMixingModelKernel(kernels, H')), test/ilmm.jl:5).
# ----------------------------------------------------------------------------------------
def naive_cov(gps: Sequence[Dict], H: np.ndarray, x: np.ndarray, x2: Optional[np.ndarray] = None) -> np.ndarray:
    """sum_l h_l h_l' (x) K_l, by-outputs ordering."""
    out = None
    for l, g in enumerate(gps):
        blk = np.kron(np.outer(H[:, l], H[:, l]), kernelmatrix({k: v for k, v in g.items() if k != "post"}, x, x2))
        out = blk if out is None else out + blk
    return out


def naive_mean(gps: Sequence[Dict], H: np.ndarray, x: np.ndarray) -> np.ndarray:
    n = npoints(x)
    mu = np.array([float(g.get("mean", 0.0)) for g in gps])
    return np.repeat(H @ mu, n)


def naive_logpdf(gps: Sequence[Dict], H: np.ndarray, x: np.ndarray, s2: float, y: np.ndarray) -> float:
    C = naive_cov(gps, H, x)
    return float(gaussian_logpdf(naive_mean(gps, H, x), C + s2 * np.eye(C.shape[0]), np.asarray(y)))


def naive_posterior_mean_cov(gps: Sequence[Dict], H: np.ndarray, x: np.ndarray, s2: float, y: np.ndarray,
                             xs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Posterior of the naive dense GP at xs (no observation noise added)."""
    C = naive_cov(gps, H, x)
    L = np.linalg.cholesky(C + s2 * np.eye(C.shape[0]))
    Kxs = naive_cov(gps, H, x, xs)
    alpha = sla.cho_solve((L, True), np.asarray(y) - naive_mean(gps, H, x))
    A = sla.solve_triangular(L, Kxs, lower=True)
    return naive_mean(gps, H, xs) + Kxs.T @ alpha, naive_cov(gps, H, xs) - A.T @ A


# ----------------------------------------------------------------------------------------
# Synthetic workloads: shared by tests, bench.py and the CPU baseline.
# ----------------------------------------------------------------------------------------
def synthetic_problem(m: int, p: int, n: int, kind: str, orthogonal: bool, s2: float = 0.1,
                      seed: int = 0) -> Dict:
    """x_i = i*20/575; unit kernels; H from svd(uniform(p,m)) with S = linspace(2,1,m) (OILMM) or
    dense uniform(0,1) (ILMM); y standard normal.  Seeds: H -> seed+2, y -> seed+3."""
    x = np.arange(n, dtype=np.float64) * (20.0 / 575.0)
    gps = [{"kind": kind, "variance": 1.0, "lengthscale": 1.0, "mean": 0.0} for _ in range(m)]
    A = np.random.default_rng(seed + 2).uniform(0.0, 1.0, (p, m))
    out = {"x": x, "gps": gps, "s2": s2, "m": m, "p": p, "n": n}
    if orthogonal:
        U, _, _ = np.linalg.svd(A, full_matrices=False)
        out["U"], out["S"] = np.ascontiguousarray(U), np.linspace(2.0, 1.0, m)
        out["H"] = orthogonal_dense(out["U"], out["S"])
    else:
        out["H"] = A
    out["y"] = np.random.default_rng(seed + 3).standard_normal(n * p)
    return out


# ----------------------------------------------------------------------------------------
# Gradients of the OILMM logpdf (SURVEY.md section 8f "next" #1: what Zygote.gradient(logpdf, fx, y) differentiates,
# reference test/oilmm.jl:31-32).  Analytic, dense; checked against central finite differences in tests/test_oracle.py.
# ----------------------------------------------------------------------------------------
def kernel_dlengthscale(kind: str, variance: float, lengthscale: float, r: np.ndarray) -> np.ndarray:
    """d kappa(r/ell) / d ell."""
    d = r / lengthscale
    if kind == "se":
        return variance * np.exp(-0.5 * d * d) * d * d / lengthscale
    if kind == "matern32":
        s = math.sqrt(3.0) * d
        return variance * s * s * np.exp(-s) / lengthscale
    s = math.sqrt(5.0) * d
    return variance * np.exp(-s) * (s * s / 3.0) * (1.0 + s) / lengthscale


def oilmm_logpdf_grad(gps: Sequence[Dict], U: np.ndarray, S: np.ndarray, x: np.ndarray, s2: float, y: np.ndarray) -> Dict:
    """Value and gradients of src/oilmm.jl:79-93 w.r.t. y, sigma2, S, U (treated as an unconstrained p x m matrix, as
    Zygote treats the field) and each latent's (variance, lengthscale, mean)."""
    n = npoints(x)
    p, m = U.shape
    Y = reshape_y(y, n)
    T, ST = project_orthogonal(U, S, s2)
    Ty = T @ Y
    gY = np.zeros_like(Y)
    gU = np.zeros_like(U)
    gS = np.zeros(m)
    gs2 = 0.0
    ggps = []
    val = 0.0
    R = pairwise_dist(x)
    for l, g in enumerate(gps):
        v, ell, mu = g.get("variance", 1.0), g.get("lengthscale", 1.0), g.get("mean", 0.0)
        K = kernel_eval(g["kind"], v, ell, R)
        Kt = K + ST[l] * np.eye(n)
        L = np.linalg.cholesky(Kt)
        delta = Ty[l] - mu
        alpha = sla.cho_solve((L, True), delta)
        Kinv = sla.cho_solve((L, True), np.eye(n))
        val += -0.5 * (n * LOG2PI + 2.0 * np.sum(np.log(np.diag(L))) + delta @ alpha)
        A = np.outer(alpha, alpha) - Kinv
        g_s = 0.5 * np.trace(A)                                 # d lml / d noise_l
        ggps.append({"variance": 0.5 * np.sum(A * K) / v,
                     "lengthscale": 0.5 * np.sum(A * kernel_dlengthscale(g["kind"], v, ell, R)),
                     "mean": float(np.sum(alpha))})
        gs2 += g_s / S[l]
        gS[l] += -g_s * s2 / S[l] ** 2 + 0.5 * (alpha @ Ty[l]) / S[l]
        gU[:, l] += -(Y @ alpha) / math.sqrt(S[l])
        gY += -np.outer(T[l], alpha)
    Pm = np.eye(p) - U @ U.T
    PY = Pm @ Y
    Rn = np.sum(PY * PY)
    val += -(n * (np.sum(np.log(S)) + (p - m) * math.log(2.0 * math.pi * s2)) + Rn / s2) / 2.0
    gS += -n / (2.0 * S)
    gs2 += -0.5 * (n * (p - m) / s2 - Rn / s2 ** 2)
    M2 = Y @ Y.T
    gU += (Pm @ M2 @ U + M2 @ Pm @ U) / s2                     # -(1/(2 s2)) dR/dU, dR/dU = -2 (P M2 U + M2 P U)
    gY += -(Pm.T @ PY) / s2
    return {"value": float(val), "y": gY.reshape(-1), "sigma2": float(gs2), "S": gS, "U": gU, "gps": ggps}
