// Do two under-filled update kernels on two streams overlap on the chip?
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
#include <chrono>
int main() {
  const int M = 8192, N = 512, K = 2048, ld = M + 16;   // 64 x 4 = 256 tiles: half the chip's workgroup slots
  setenv("LMM_DETERMINISTIC", "1", 1);
  double *C[4], *A[4];
  for (int i = 0; i < 4; ++i) { hipMalloc(&C[i], (size_t)ld * N * 8); hipMalloc(&A[i], (size_t)ld * K * 8); hipMemset(C[i], 0, (size_t)ld * N * 8); hipMemset(A[i], 0, (size_t)ld * K * 8); }
  hipStream_t st[4]; for (int i = 0; i < 4; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ns : {1, 2, 4}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipDeviceSynchronize();
      hipEventRecord(e0, 0); hipStreamSynchronize(0);
      auto t0 = std::chrono::high_resolution_clock::now();
      for (int r = 0; r < 8; ++r) for (int i = 0; i < ns; ++i) launch_gemm_nt(C[i], ld, A[i], ld, A[i], ld, M, N, K, 0, false, st[i]);
      hipDeviceSynchronize();
      auto t1 = std::chrono::high_resolution_clock::now();
      double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
      if (rep) printf("streams=%d: %d kernels (256 tiles each) in %.3f ms -> %.3f ms per kernel, %.1f TF aggregate\n", ns, 8 * ns, ms, ms / (8 * ns), 8.0 * ns * 2.0 * M * N * K / ms / 1e9);
    }
  }
  return 0;
}
