import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0)
ml = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
P = O.synthetic_problem(32, 64, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(32)])
fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]), shard=(0, ml))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 64), 0.1)
yd = torch.from_numpy(P["y"]).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); v = lmm_amd.logpdf(fx, yd); torch.cuda.synchronize(); t1 = time.perf_counter()
    G = lmm_amd.logpdf_and_gradient(fx, yd); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: n={n} latents={ml}: logpdf {1e3*(t1-t0):.1f} ms, logpdf+gradient {1e3*(t2-t1):.1f} ms (x{(t2-t1)/(t1-t0):.2f}); value diff {abs(v-G['value'])/abs(v):.1e}", flush=True)
