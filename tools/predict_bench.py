"""C3-shaped posterior predictive on one GPU's share: OILMM, m_local latents, n = n* = 8192, Float64."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
m, p, n, ns, ml = 64, 128, 8192, 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 8
P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
xs = torch.from_numpy(P["x"] + 0.5 * 20.0 / 575.0).cuda()
f = lmm_amd.ILMM(fs, H, shard=(0, ml))
fx = f(lmm_amd.MOInputIsotopicByOutputs(xd, p), 0.1)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    post = lmm_amd.posterior(fx, yd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    mu, v = lmm_amd.mean_and_var(post(lmm_amd.MOInputIsotopicByOutputs(xs, p), 0.1))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    fl_post = ml * n ** 3 / 3; fl_pred = ml * n * n * ns
    print(f"rep {rep}: posterior {1e3 * (t1 - t0):.1f} ms ({fl_post / (t1 - t0) / 1e12:.1f} TF), mean_and_var {1e3 * (t2 - t1):.1f} ms "
          f"({fl_pred / (t2 - t1) / 1e12:.1f} TF on n^2 n* TRSM flops); marginals/s {ns * p / (t2 - t1):.3e}", flush=True)
    del post
