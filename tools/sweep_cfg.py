"""Sweep LMM_BATCH / LMM_NSTREAMS for a given number of latents per GPU (C2 shapes), each config in a fresh process."""
import os, subprocess, sys
code = r'''
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from oracle import lmm_oracle as O
lmm_amd.init(0)
ml = int(sys.argv[1])
P = O.synthetic_problem(32, 64, 16384, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(32)])
H = lmm_amd.Orthogonal(P["U"], P["S"])
xd, yd = torch.from_numpy(P["x"]).cuda(), torch.from_numpy(P["y"]).cuda()
fx = lmm_amd.ILMM(fs, H, shard=(0, ml))(lmm_amd.MOInputIsotopicByOutputs(xd, 64), 0.1)
lmm_amd.logpdf(fx, yd, True); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): lmm_amd.logpdf(fx, yd, True)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 3 * 1e3:.1f}")
'''
for ml in map(int, sys.argv[1].split(",")):
    for b, ns in [(1, 4), (2, 2), (2, 4), (4, 2), (4, 4), (8, 2), (8, 4)]:
        if b > ml: continue
        env = dict(os.environ, LMM_BATCH=str(b), LMM_NSTREAMS=str(ns), LMM_BATCH_FORCE="1")
        r = subprocess.run([sys.executable, "-c", code, str(ml)], env=env, capture_output=True, text=True)
        print(f"latents={ml:2d} batch={b} streams={ns}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:]} ms", flush=True)
