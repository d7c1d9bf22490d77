"""Time one rank's share of C2 as if the job ran on W GPUs (latent shard of rank 0), on a single GPU."""
import sys, time
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
for W in (1, 2, 4, 8, 16, 32):
    fx = lmm_amd.ILMM(fs, H, shard=lmm_amd.latent_shard(32, 0, W))(xin, 0.1)
    lmm_amd.logpdf(fx, yd, True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): lmm_amd.logpdf(fx, yd, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(f"world={W:2d} latents/gpu={32 // W:2d}  {dt * 1e3:8.1f} ms/step  -> speedup vs W=1 shown after", flush=True)
