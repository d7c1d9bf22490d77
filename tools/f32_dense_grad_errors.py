import sys; sys.path.insert(0,'/root/repo')
import numpy as np, lmm_amd as lmm
from oracle import lmm_oracle as O
lmm.init(0)
K = {"se": lmm.SEKernel, "matern32": lmm.Matern32Kernel, "matern52": lmm.Matern52Kernel}
def model(gps): return lmm.independent_mogp([lmm.GP(g["mean"], K[g["kind"]](g["variance"], g["lengthscale"])) for g in gps])
rng = np.random.default_rng(31)
n, ns, p, m, s2 = 300, 60, 4, 3, 0.1
x, xs = np.sort(rng.uniform(0, 12, n)), np.sort(rng.uniform(0, 12, ns))
gps = [{"kind": k, "variance": float(rng.uniform(0.7, 1.4)), "lengthscale": float(rng.uniform(0.7, 1.5)), "mean": float(rng.normal())} for k in ["se", "matern52", "matern32"]]
H = rng.uniform(0.2, 1.0, size=(p, m)); y, ys = rng.standard_normal(n * p), rng.standard_normal(ns * p)
fx = lmm.ILMM(model(gps), H)(lmm.MOInputIsotopicByOutputs(x, p), s2)
def both(fn):
    lmm.set_compute_dtype("f32"); a = fn(); lmm.set_compute_dtype("f64"); b = fn(); return a, b
for name, fn, ykeys in [("prior", lambda: lmm.logpdf_and_gradient(fx, y), ["y"]), ("post", lambda: lmm.logpdf_and_gradient(lmm.posterior(fx, y)(lmm.MOInputIsotopicByOutputs(xs, p), 0.2), ys), ["y", "y_train"])]:
    G, R = both(fn)
    print(name, "value rel", abs(G["value"]-R["value"])/abs(R["value"]), "sigma2", G["sigma2"], R["sigma2"])
    for k in ykeys + ["H"]: print("  ", k, np.abs(np.asarray(G[k])-np.asarray(R[k])).max()/np.abs(np.asarray(R[k])).max())
    for l in range(m): print("   gp", l, {k: (round(G["gps"][l][k],4), round(R["gps"][l][k],4)) for k in ("variance","lengthscale","mean")})
