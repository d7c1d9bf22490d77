"""Shader clock and socket power while the fp32 (256 x 256 tile, v_mfma_f32_32x32x2_f32) and the f64 (v_mfma_f64_16x16x4_f64) update kernels run
back to back for a few seconds each -- is the fp32 kernel's 0.82-0.85 of the 157.3-TFLOP/s datasheet peak an issue-rate limit of the
loop or a clock the device does not hold under that load?   python tools/f32_clock_probe.py [seconds]"""
import sys, time, threading, subprocess, re, ctypes as C
sys.path.insert(0, '.')
import torch, lmm_amd
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
lmm_amd.init(0)
lib = lmm_amd.load()


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)
            pw = re.findall(r"Socket Graphics Package Power \(W\): ([\d.]+)", txt) or re.findall(r"Power \(W\): ([\d.]+)", txt)
            out.append((time.perf_counter(), int(sclk[0]) if sclk else None, float(pw[0]) if pw else None))
        except Exception as e:      # noqa: BLE001
            out.append((time.perf_counter(), None, None))
        time.sleep(0.05)


def run(dtype, M, N, K):
    lmm_amd.set_compute_dtype(dtype)
    tdt = torch.float32 if dtype == "f32" else torch.float64
    ldc, lda = M + 16, M + 4
    g = torch.Generator(device="cuda").manual_seed(1)
    Ct = torch.randn(N, ldc, generator=g, device="cuda", dtype=tdt)
    At = torch.randn(K, lda, generator=g, device="cuda", dtype=tdt)
    torch.cuda.synchronize()
    call = lambda: lib.lmm_dev_gemm_nt_sub(C.c_void_p(Ct.data_ptr()), ldc, C.c_void_p(At.data_ptr()), lda, C.c_void_p(At.data_ptr()), lda, M, N, K, 1)
    assert call() == 0
    idle = []
    st0 = threading.Event(); th = threading.Thread(target=sample, args=(st0, idle)); th.start(); time.sleep(1.0); st0.set(); th.join()
    samples = []
    stop = threading.Event(); th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    t0 = time.perf_counter(); reps = 0; rates = []
    outs = N * (N + 1) / 2 + (M - N) * N
    while time.perf_counter() - t0 < secs:
        t1 = time.perf_counter(); call(); dt = time.perf_counter() - t1
        rates.append(2 * K * outs / dt / 1e12); reps += 1
    stop.set(); th.join()
    clk = [s[1] for s in samples if s[1]]; pw = [s[2] for s in samples if s[2]]
    iclk = [s[1] for s in idle if s[1]]; ipw = [s[2] for s in idle if s[2]]
    med = lambda v: sorted(v)[len(v) // 2] if v else None
    print(f"{dtype} SYRK M=N={M} K={K}: {reps} launches in {secs:.0f} s, TFLOP/s first {rates[0]:.1f} median {med(rates):.1f} last {rates[-1]:.1f} | "
          f"sclk MHz idle {med(iclk)} under load min {min(clk) if clk else None} median {med(clk)} max {max(clk) if clk else None} ({len(clk)} samples) | "
          f"power W idle {med(ipw)} under load median {med(pw)} max {max(pw) if pw else None}", flush=True)


run("f32", 16384, 16384, 8192)
run("f64", 16384, 16384, 8192)
run("f32", 16384, 16384, 8192)
