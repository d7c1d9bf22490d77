"""Gram assembly rate by input dimension and kernel kind (lmm_dev_gram, n = 16384): is the d > 1 path off the d = 1 rate?"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
lmm_amd.init(0); lib = lmm_amd.load()
n = 16384; NC = n; NR = n + 64; ld = NR
A = torch.empty(ld * NC, dtype=torch.float64, device='cuda')
rng = np.random.default_rng(0)
for d in (1, 2, 3, 8):
    x = torch.from_numpy(np.ascontiguousarray(rng.uniform(0, 20, (n, d)))).cuda()      # d x n column-major == (n, d) C-order
    for kind in (0, 1, 2):
        gp = L.GpT(kind, 1.0, 1.0, 0.0)
        args = (C.c_void_p(A.data_ptr()), ld, NR, NC, C.c_void_p(x.data_ptr()), d, n, C.byref(gp), C.c_double(0.1))
        L.check(lib.lmm_dev_gram(*args)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): L.check(lib.lmm_dev_gram(*args))
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"d={d} kind={kind}: {dt*1e3:7.3f} ms  {n*(n+1)/2*8/dt/1e12:5.2f} TB/s (lower-triangle bytes)", flush=True)
