"""logpdf latency vs n for a mid-size OILMM (m = 8, p = 16): where the leaf chain, not the MFMA rate, sets the time."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
m, p = 8, 16
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
for n in (256, 512, 1024, 2048, 4096, 8192):
    P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
    fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), p), 0.1)
    y = torch.from_numpy(P["y"]).cuda()
    for _ in range(3): lmm_amd.logpdf(fx, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20 if n <= 2048 else 5
    for _ in range(reps): lmm_amd.logpdf(fx, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"n={n:5d}: {dt*1e3:8.3f} ms/eval  ({m*n**3/3/dt/1e12:6.2f} TFLOP/s; leaves {((n+63)//64)})", flush=True)
