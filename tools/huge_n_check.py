"""Memory-plan check at n = 65536 (one factor matrix = 34.4 GB): logpdf of an 8-latent OILMM must shrink its batch plan to fit
288 GB instead of failing in hipMalloc; quadratic-in-y property as the size-independent parity check."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
n, p, m = 65536, 8, 8
P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
fx = lmm_amd.ILMM(fs, H)(lmm_amd.MOInputIsotopicByOutputs(xd, p), 0.1)
t0 = time.perf_counter(); v = lmm_amd.logpdf(fx, yd); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"logpdf n={n} ({m} latents): {v:.6f} in {1e3*(t1-t0):.0f} ms ({m*n**3/3/(t1-t0)/1e12:.1f} TF)", flush=True)
t0 = time.perf_counter(); v2 = lmm_amd.logpdf(fx, yd); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"second call: {1e3*(t1-t0):.0f} ms ({m*n**3/3/(t1-t0)/1e12:.1f} TF), same value: {v2 == v}", flush=True)
l0 = lmm_amd.logpdf(fx, torch.zeros_like(yd)); l2 = lmm_amd.logpdf(fx, 2 * yd)
print("quadratic-in-y property rel err:", abs((l2 - l0) - 4 * (v - l0)) / abs(l2 - l0), flush=True)
print("free/total GB:", [round(b / 2**30, 1) for b in torch.cuda.mem_get_info()])
