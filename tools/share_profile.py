"""Kernel-class times (HIP events, serial pass) of one rank's share of C2 as if the job ran on W GPUs, on a single GPU."""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch
import lmm_amd
from lmm_amd import _lib as L
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
lib = lmm_amd.load()
P = O.synthetic_problem(32, 64, 16384, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(32)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
xin = lmm_amd.MOInputIsotopicByOutputs(xd, 64)
for W in [int(a) for a in sys.argv[1:]] or [8, 4, 2]:
    fx = lmm_amd.ILMM(fs, H, shard=lmm_amd.latent_shard(32, 0, W))(xin, 0.1)
    lmm_amd.logpdf(fx, yd, True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): lmm_amd.logpdf(fx, yd, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    L.check(lib.lmm_profile_begin(1))
    lmm_amd.logpdf(fx, yd, True)
    ent = (L.ProfEntryT * len(L.PROF_CLASSES))()
    L.check(lib.lmm_profile_end(ent))
    cls = {c: (int(ent[i].launches), round(float(ent[i].ms), 3)) for i, c in enumerate(L.PROF_CLASSES)}
    tot = sum(v[1] for v in cls.values())
    ideal = (32 // W) * 16384 ** 3 / 3 / 78.6e12 * 1e3
    print(f"world={W:2d} latents/gpu={32 // W:2d}  {dt * 1e3:8.2f} ms/eval (at FP64 peak: {ideal:.1f} ms)  classes (launches, ms): {cls}  sum {tot:.2f}", flush=True)
