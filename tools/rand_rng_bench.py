import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
m, p, n, share = 64, 128, 2048, 8
P = O.synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
f = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]), shard=(0, share))
xin = lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), p)
for name, rng in (("host numpy Generator", np.random.default_rng(0)), ("DeviceNormals", lmm_amd.DeviceNormals(0))):
    for _ in range(3): lmm_amd.rand(rng, f(xin, 0.1), jitters=(1e-9, 1e-8, 1e-8))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): lmm_amd.rand(rng, f(xin, 0.1), jitters=(1e-9, 1e-8, 1e-8))
    torch.cuda.synchronize(); print(f"rand(prior) n={n}, 8 latents, {name}: {(time.perf_counter()-t0)/10*1e3:.2f} ms")
