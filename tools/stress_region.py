"""Repeat logpdf evaluations that take the dataflow paths (region kernel whole / as base case, assistants, thin row streams, fused bulk
rows) and require every repetition to return the SAME bits: these paths use no atomics, so any difference is a missed dependency.
   python tools/stress_region.py [reps]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O
lmm_amd.init(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for (n, m) in [(200, 3), (552, 20), (640, 8), (1024, 4), (1024, 16), (1024, 32), (1100, 36), (1536, 8), (2048, 8), (2048, 16), (3000, 5), (4096, 8)]:
    P = O.synthetic_problem(m, 2 * m, n, "matern52", True, s2=0.1, seed=n + m)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
    fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1)
    yd = torch.from_numpy(P["y"]).cuda()
    r = max(10, reps // (1 + n // 1024) // (1 + m // 16))
    vals = [lmm_amd.logpdf(fx, yd) for _ in range(r)]
    uniq = sorted(set(vals))
    print(f"n={n:5d} m={m:3d}: {r:4d} evaluations, {len(uniq)} distinct value(s): {uniq[:3]}", flush=True)
    bad += len(uniq) > 1
sys.exit(1 if bad else 0)
