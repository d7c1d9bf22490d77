"""Size check at n = 32768 (BASELINE configs[4] scale, here in Float64): logpdf, posterior, marginals, rand on 2 latents."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
n, p, m, ml = 32768, 8, 4, 2
P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
f = lmm_amd.ILMM(fs, H, shard=(0, ml))
fx = f(lmm_amd.MOInputIsotopicByOutputs(xd, p), 0.1)
t0 = time.perf_counter(); v = lmm_amd.logpdf(fx, yd, True); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"logpdf n={n} ({ml} latents): {v:.6f} in {1e3*(t1-t0):.0f} ms ({ml*n**3/3/(t1-t0)/1e12:.1f} TF)", flush=True)
l0 = lmm_amd.logpdf(fx, torch.zeros_like(yd), True); l2 = lmm_amd.logpdf(fx, 2 * yd, True)
print("quadratic-in-y property rel err:", abs((l2 - l0) - 4 * (v - l0)) / abs(l2 - l0), flush=True)
post = lmm_amd.posterior(fx, yd)
xs = xd[:4096] + 0.01
mu, var = lmm_amd.mean_and_var(post(lmm_amd.MOInputIsotopicByOutputs(xs, p), 0.1))
print("posterior marginals: mean finite", bool(torch.isfinite(mu).all()), "var range", float(var.min()), float(var.max()), flush=True)
s = lmm_amd.rand(np.random.default_rng(0), post(lmm_amd.MOInputIsotopicByOutputs(xs.cpu().numpy(), p), 0.1), jitters=(1e-9, 1e-8, 1e-8))
print("posterior sample finite:", bool(np.isfinite(s).all()), "len", len(s), flush=True)
