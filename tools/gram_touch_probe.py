"""Gram assembly rate on freshly allocated against already-touched factor matrices (the 4.9 vs 5.7 TB/s spread between the rocprofv3 run and
the driver-style run of round 4): one 16-latent batch at n = 16384, class 'gram' of an instrumented pass, (a) right after
lmm_release_cached_memory (the pool hands out blocks hipMalloc has just created: first touch), (b) again (pooled, touched blocks)."""
import sys
sys.path.insert(0, '.')
import ctypes as C
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
from lmm_amd.workloads import synthetic_problem
lmm_amd.init(0)
lib = lmm_amd.load()
m, p, n = 16, 32, 16384
P = synthetic_problem(m, p, n, "matern52", True, s2=0.1, seed=0)
fs = lmm_amd.independent_mogp([lmm_amd.GP(lmm_amd.Matern52Kernel()) for _ in range(m)])
fx = lmm_amd.ILMM(fs, lmm_amd.Orthogonal(P["U"], P["S"]))(lmm_amd.MOInputIsotopicByOutputs(torch.from_numpy(P["x"]).cuda(), p), 0.1)
yd = torch.from_numpy(P["y"]).cuda()


def gram_rate():
    L.check(lib.lmm_profile_begin(1))
    lmm_amd.logpdf(fx, yd, False)
    ent = (L.ProfEntryT * len(L.PROF_CLASSES))(); L.check(lib.lmm_profile_end(ent))
    return ent[0].work / (ent[0].ms * 1e-3) / 1e9, ent[0].ms


for rep in range(2):
    lib.lmm_release_cached_memory()
    a = gram_rate(); b = gram_rate(); c = gram_rate()
    print(f"round {rep}: fresh blocks {a[0]:.0f} GB/s ({a[1]:.3f} ms) | pooled, touched {b[0]:.0f} GB/s ({b[1]:.3f} ms) | again {c[0]:.0f} GB/s ({c[1]:.3f} ms)")
wr = C.c_double()
for bytes_ in (1 << 30, 8 << 30, 17 << 30):
    L.check(lib.lmm_dev_write_rate(C.c_size_t(bytes_), 3, C.byref(wr)))
    print(f"hipMemsetAsync {bytes_ >> 30} GiB x3: {wr.value:.0f} GB/s")
