import ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, lmm_amd
from lmm_amd import _lib as L
lmm_amd.init(0); lib = lmm_amd.load()
n = 16384; NC = 16384; NR = 16448; ld = NR
A = torch.empty(NC * ld, dtype=torch.float64, device="cuda")
x = torch.arange(n, dtype=torch.float64, device="cuda") * (20.0 / 575.0)
for kind in ("matern52", "se", "matern32"):
    g = L.gps_array([{"kind": kind}])
    for _ in range(2): lib.lmm_dev_gram(C.c_void_p(A.data_ptr()), ld, NR, NC, C.c_void_p(x.data_ptr()), 1, n, g, C.c_double(0.1))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): lib.lmm_dev_gram(C.c_void_p(A.data_ptr()), ld, NR, NC, C.c_void_p(x.data_ptr()), 1, n, g, C.c_double(0.1))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{kind:9s} n={n}: {dt*1e3:.3f} ms  {n*(n+1)/2*8/dt/1e9:.0f} GB/s (lower-tri algorithmic bytes)")
