"""Kernel-class times (serial instrumented pass) and wall time of an OILMM logpdf for small / mid n: what the factorisation of a
matrix that is ONE region (n <= 1024) costs.   python tools/region_probe.py [n ...]   (LMM_REGION=0 / 1024, LMM_PANEL128=0 to compare)"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
from lmm_amd import workloads as O
lmm_amd.init(0); lib = lmm_amd.load()
for n in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048]:
    for m in (1, 4, 16):
        P = O.synthetic_problem(m, 2 * m, n, "matern52", True, s2=0.1, seed=0)
        fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
        fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1)
        yd = torch.from_numpy(P["y"]).cuda()
        for _ in range(5): lmm_amd.logpdf(fx, yd, False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): lmm_amd.logpdf(fx, yd, False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
        L.check(lib.lmm_profile_begin(1)); lmm_amd.logpdf(fx, yd, False)
        ent = (L.ProfEntryT * len(L.PROF_CLASSES))(); L.check(lib.lmm_profile_end(ent))
        cls = {c: round(ent[i].ms * 1e3, 1) for i, c in enumerate(L.PROF_CLASSES) if ent[i].launches}
        print(f"n={n:5d} m={m:2d}: {dt * 1e6:8.1f} us/eval   classes (us): {cls}", flush=True)
