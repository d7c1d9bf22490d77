"""Kernel-class times (serial-stream instrumented pass) for one batch of B latents of C2: what a rank's batch costs alone."""
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
from lmm_amd import workloads as O      # input generation only
lmm_amd.init(0); lib = lmm_amd.load()
B = int(sys.argv[1])
P = O.synthetic_problem(32, 64, 16384, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(32)])
fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]), shard=(0, B))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), 64), 0.1)
yd = torch.from_numpy(P["y"]).cuda()
lmm_amd.logpdf(fx, yd, False)
L.check(lib.lmm_profile_begin(1)); lmm_amd.logpdf(fx, yd, False)
ent = (L.ProfEntryT * len(L.PROF_CLASSES))(); L.check(lib.lmm_profile_end(ent))
print(f"batch of {B}:", {c: round(ent[i].ms, 2) for i, c in enumerate(L.PROF_CLASSES)}, "update TF", round(ent[1].work / ent[1].ms / 1e9, 1), flush=True)
