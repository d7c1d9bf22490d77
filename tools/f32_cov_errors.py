"""Observed errors of the fp32 full covariance against the Float64 oracle (the shapes of tests/test_gpu_f32.py::test_f32_full_covariance)."""
import sys; sys.path.insert(0, '.')
import numpy as np, lmm_amd as lmm
from oracle import lmm_oracle as O
lmm.init(0)
K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
def model(gps): return lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])
rng = np.random.default_rng(23)
n, ns, p, m, s2 = 600, 70, 4, 3, 0.1
x, xs = np.sort(rng.uniform(0, 20, n)), np.sort(rng.uniform(0, 20, ns))
gps = [{"kind": k, "variance": float(rng.uniform(0.7, 1.4)), "lengthscale": float(rng.uniform(0.7, 1.5)), "mean": float(rng.normal())} for k in ["matern52", "se", "matern32"]]
U, _ = np.linalg.qr(rng.standard_normal((p, m))); S = np.linspace(1.5, 0.8, m); H = O.orthogonal_dense(U, S)
y = rng.standard_normal(n * p)
f = lmm.ILMM(model(gps), lmm.Orthogonal(U, S)); xsin = lmm.MOInputIsotopicByOutputs(xs, p)
lmm.set_compute_dtype("f32")
post = lmm.posterior(f(lmm.MOInputIsotopicByOutputs(x, p), s2), y)
M1, C1 = lmm.mean_and_cov(post(xsin, s2))
Mo, Co = O.naive_posterior_mean_cov(gps, H, x, s2, y, xs); Co = Co + s2 * np.eye(ns * p)
print("posterior mean err / max", np.abs(M1 - Mo).max() / np.abs(Mo).max(), " cov err / max", np.abs(C1 - Co).max() / np.abs(Co).max())
M0, C0 = lmm.mean_and_cov(f(xsin, s2))
lmm.set_compute_dtype("f64"); M64, C64 = lmm.mean_and_cov(f(xsin, s2))
print("prior cov err / max", np.abs(C0 - C64).max() / np.abs(C64).max())
