"""Time rank 0's 4-latent share of C2 (the 8-GPU job's per-rank work) on one GPU.  Environment knobs (LMM_BATCH, ...) are read by the
library: run once per setting."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
P = O.synthetic_problem(32, 64, 16384, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(32)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
xin = lmm_amd.MOInputIsotopicByOutputs(xd, 64)
fx = lmm_amd.ILMM(fs, H, shard=lmm_amd.latent_shard(32, 0, 8))(xin, 0.1)
v = lmm_amd.logpdf(fx, yd, True); torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("REPS", "5"))):
    t0 = time.perf_counter(); lmm_amd.logpdf(fx, yd, True); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"LMM_BATCH={os.environ.get('LMM_BATCH', '-')} share: min {min(ts):.2f} median {sorted(ts)[len(ts) // 2]:.2f} ms  value {v:.6f}", flush=True)
