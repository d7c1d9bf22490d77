"""One OILMM logpdf at (n, m) with the region kernel traced:  LMM_REGION_TRACE=1 python tools/region_one.py 552 4 2> trace.txt"""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O
n, m = int(sys.argv[1]), int(sys.argv[2])
lmm_amd.init(0)
P = O.synthetic_problem(m, 2 * m, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1)
yd = torch.from_numpy(P["y"]).cuda()
for _ in range(3):
    print(lmm_amd.logpdf(fx, yd, False))
