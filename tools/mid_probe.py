"""Wall time of an OILMM logpdf at mid sizes (one batch of m latents):  python tools/mid_probe.py n m [n m ...]   (compare LMM_REGION_ALL=1
LMM_REGION=512 / 1024 with the default panel recursion)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O
lmm_amd.init(0)
args = [int(a) for a in sys.argv[1:]]
for n, m in zip(args[0::2], args[1::2]):
    P = O.synthetic_problem(m, 2 * m, n, "matern52", True, s2=0.1, seed=0)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
    fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1)
    yd = torch.from_numpy(P["y"]).cuda()
    for _ in range(3): lmm_amd.logpdf(fx, yd, False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20 if n <= 4096 else 5
    for _ in range(reps): v = lmm_amd.logpdf(fx, yd, False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"n={n:6d} m={m:3d}: {dt * 1e3:9.3f} ms/eval  ({m * n ** 3 / 3 / dt / 1e12:6.2f} TFLOP/s)  logpdf {v:.6f}", flush=True)
