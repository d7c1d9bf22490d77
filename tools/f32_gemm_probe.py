"""fp32 wide update rate through the exported building block lmm_dev_gemm_nt_sub:  LMM_F32_TILE256=0|1 python tools/f32_gemm_probe.py [M N K lower ...]"""
import sys, time, ctypes as C
sys.path.insert(0, '.')
import torch, lmm_amd
lmm_amd.init(0); lmm_amd.set_compute_dtype("f32")
lib = lmm_amd.load()
args = [int(a) for a in sys.argv[1:]] or [16384, 16384, 8192, 1, 16384, 8192, 16384, 1, 8192, 8192, 4096, 0, 32768, 4096, 4096, 1]
for M, N, K, lower in zip(args[0::4], args[1::4], args[2::4], args[3::4]):
    ldc, lda, ldb = M + 16, M + 4, N + 8
    g = torch.Generator(device="cuda").manual_seed(1)
    Ct = torch.randn(N, ldc, generator=g, device="cuda", dtype=torch.float32)
    At = torch.randn(K, lda, generator=g, device="cuda", dtype=torch.float32)
    Bt = At if (lower and M == N) else torch.randn(K, ldb, generator=g, device="cuda", dtype=torch.float32)
    if Bt is At: ldb = lda
    torch.cuda.synchronize()
    call = lambda: lib.lmm_dev_gemm_nt_sub(C.c_void_p(Ct.data_ptr()), ldc, C.c_void_p(At.data_ptr()), lda, C.c_void_p(Bt.data_ptr()), ldb, M, N, K, lower)
    assert call() == 0
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps): call()
    dt = (time.perf_counter() - t0) / reps
    outs = (N * (N + 1) / 2 + (M - N) * N) if lower else M * N
    print(f"M={M} N={N} K={K} lower={lower}: {dt * 1e3:8.3f} ms  {2 * K * outs / dt / 1e12:7.2f} TFLOP/s (algorithmic)", flush=True)
