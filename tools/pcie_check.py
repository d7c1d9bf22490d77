"""Host-pointer vs device-pointer logpdf at C2 (PCIe-inclusive rate)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
P = O.synthetic_problem(32, 64, 16384, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(32)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
for name, x, y in (("device", torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()), ("host", P["x"], P["y"])):
    fx = lmm_amd.ILMM(fs, H)(lmm_amd.MOInputIsotopicByOutputs(x, 64), 0.1)
    lmm_amd.logpdf(fx, y); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): v = lmm_amd.logpdf(fx, y)
    torch.cuda.synchronize()
    print(f"{name:6s} pointers: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms/eval  logpdf {v:.6f}", flush=True)
