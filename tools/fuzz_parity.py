"""Randomised parity sweep of the HIP path against the oracle: shapes, kernel kinds, input dimensions, shards, test-point counts.
    python tools/fuzz_parity.py [ncases] [seed]"""
import sys
sys.path.insert(0, '.')
import numpy as np, lmm_amd
from oracle import lmm_oracle as O
lmm_amd.init(0)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
K = {"se": lmm_amd.SEKernel, "matern32": lmm_amd.Matern32Kernel, "matern52": lmm_amd.Matern52Kernel}
worst = {}
def upd(name, got, ref, scale=None):
    got, ref = np.asarray(got, dtype=float), np.asarray(ref, dtype=float)
    err = float(np.max(np.abs(got - ref) / (np.abs(ref) + (scale if scale is not None else 1e-9))))
    worst[name] = max(worst.get(name, 0.0), err)
    return err
for c in range(ncases):
    m = int(rng.integers(1, 6)); p = m + int(rng.integers(0, 4)); d = int(rng.integers(1, 4))
    n = int(rng.choice([1, 2, 3, 7, 63, 64, 65, 127, 130, 200, 257, 400, 640])); ns = int(rng.choice([1, 2, 5, 63, 64, 65, 150]))
    s2 = float(rng.choice([0.05, 0.1, 0.5]))
    x = rng.uniform(0, 6, n) if d == 1 else rng.uniform(0, 3, (d, n))
    xs = rng.uniform(0, 6, ns) if d == 1 else rng.uniform(0, 3, (d, ns))
    gps = [{"kind": str(rng.choice(list(K))), "variance": float(rng.uniform(0.5, 1.5)), "lengthscale": float(rng.uniform(0.5, 1.5)),
            "mean": float(rng.uniform(-0.3, 0.3))} for _ in range(m)]
    A = rng.uniform(0.1, 1.0, (p, m))
    U, _, _ = np.linalg.svd(A, full_matrices=False); S = np.linspace(2.0, 1.0, m)
    y = rng.standard_normal(n * p); ys = rng.standard_normal(ns * p)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])
    f = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(U, S))
    xin, xsin = lmm_amd.MOInputIsotopicByOutputs(x, p), lmm_amd.MOInputIsotopicByOutputs(xs, p)
    e1 = upd("oilmm logpdf", lmm_amd.logpdf(f(xin, s2), y), O.oilmm_logpdf(gps, U, S, x, s2, y), 1e-6)
    post = lmm_amd.posterior(f(xin, s2), y); po = O.oilmm_posterior(gps, U, S, x, s2, y)
    mu, var = lmm_amd.mean_and_var(post(xsin, s2)); mo, vo = O.oilmm_mean_var(po, U, S, xs, s2)
    e2 = upd("post mean", mu, mo, 1e-6); e3 = upd("post var", var, vo)
    e4 = upd("post logpdf", lmm_amd.logpdf(post(xsin, s2), ys), O.oilmm_logpdf(po, U, S, xs, s2, ys), 1e-6)
    # dense-H ILMM on the same data (distinct code path) when small
    e5 = 0.0
    if m * n <= 1500:
        fd = lmm_amd.ILMM(fs, np.ascontiguousarray(A))
        e5 = upd("ilmm logpdf", lmm_amd.logpdf(fd(xin, s2), y), O.ilmm_logpdf(gps, A, x, s2, y), 1e-6)
    print(f"case {c:3d}: m={m} p={p} d={d} n={n:4d} ns={ns:4d} s2={s2}: rel errs {e1:.1e} {e2:.1e} {e3:.1e} {e4:.1e} {e5:.1e}", flush=True)
print("worst relative errors:", {k: f"{v:.2e}" for k, v in worst.items()})
assert all(v < 1e-6 for v in worst.values()), "parity bar rtol 1e-6 exceeded"
print("fuzz parity OK")
