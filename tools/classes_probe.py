"""Kernel-class times (HIP events, serial pass) and wall time of one OILMM logpdf at (n, m):  python tools/classes_probe.py n m [n m ...]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
from lmm_amd import workloads as O
lmm_amd.init(0)
lib = lmm_amd.load()
args = [int(a) for a in sys.argv[1:]]
for n, m in zip(args[0::2], args[1::2]):
    P = O.synthetic_problem(m, 2 * m, n, "matern52", True, s2=0.1, seed=0)
    fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
    fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 2 * m), 0.1)
    yd = torch.from_numpy(P["y"]).cuda()
    for _ in range(3): lmm_amd.logpdf(fx, yd, False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20 if n <= 4096 else 5
    for _ in range(reps): lmm_amd.logpdf(fx, yd, False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    L.check(lib.lmm_profile_begin(1))
    lmm_amd.logpdf(fx, yd, False)
    ent = (L.ProfEntryT * len(L.PROF_CLASSES))(); L.check(lib.lmm_profile_end(ent))
    cls = {c: (int(ent[i].launches), round(float(ent[i].ms), 3), round(ent[i].work / max(ent[i].ms, 1e-9) / 1e9, 1)) for i, c in enumerate(L.PROF_CLASSES) if ent[i].launches}
    val = lmm_amd.logpdf(fx, yd, False)
    print(f"n={n} m={m}: logpdf {float(val)!r}  {dt * 1e3:.3f} ms/eval; classes (launches, ms, TFLOP/s or GB/s): {cls}", flush=True)
