import ctypes as C, sys
sys.path.insert(0, '.')
import lmm_amd
lmm_amd.init(0)
t = C.c_double()
for _ in range(3):
    rc = lmm_amd.load().lmm_dev_mfma_f64_peak(C.byref(t)); print("mfma f64 16x16x4 peak TFLOP/s:", rc, t.value, flush=True)
