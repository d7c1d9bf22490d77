// Times the trailing-update kernel alone (and its ablations: -DLMM_ABLATE_NOLOAD / -DLMM_ABLATE_NOMFMA).
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
__global__ void fill_rand(double* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    p[i] = ((double)(z & 0xFFFFFFFFFFFFFull) / 4503599627370496.0) - 0.5;
  }
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
  const int lower = argc > 4 ? atoi(argv[4]) : 1;
  const int ld = M + 16;
  double *C, *A;
  hipMalloc(&C, (size_t)ld * N * 8); hipMalloc(&A, (size_t)ld * K * 8);
  const int rnd = argc > 5 ? atoi(argv[5]) : 1;
  const int reps = argc > 6 ? atoi(argv[6]) : 3;
  hipMemset(C, 0, (size_t)ld * N * 8); hipMemset(A, 0, (size_t)ld * K * 8);
  if (rnd) { fill_rand<<<2048, 256>>>(A, (size_t)ld * K, 1u); fill_rand<<<2048, 256>>>(C, (size_t)ld * N, 2u); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    if (which) break;
    launch_gemm_nt(C, ld, A, ld, A, ld, M, N, K, lower, false, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch_gemm_nt(C, ld, A, ld, A, ld, M, N, K, lower, false, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    double fl = lower ? 2.0 * K * ((double)N * (N + 1) / 2 + (double)(M - N) * N) : 2.0 * M * N * (double)K;
#ifdef LMM_CLOCK_PROBE
    { unsigned long long h[4]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_probe), sizeof h);
      printf("  tile of block 300: %llu shader clocks in %llu ticks of 100 MHz -> %.0f MHz\n", h[0], h[1], h[1] ? 100.0 * h[0] / h[1] : 0.0); }
#endif
    printf("%s rnd=%d M=%d N=%d K=%d lower=%d: %.3f ms  %.2f TFLOP/s (algorithmic)\n", which ? "mfma16x16x4" : "mfma4x4x4_4b", rnd, M, N, K, lower, ms, fl / ms / 1e9);
  }
#ifdef LMM_CLOCK_PROBE
  { double* o; hipMalloc(&o, 256 * 256 * 8);
    for (int iters : {20000, 200000}) {
      hipEventRecord(e0); launch_mfma_peak(o, 256, iters, 0); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[4]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk_probe), sizeof h);
      printf("mfma peak probe iters=%d: %.3f ms  %.2f TF;  %llu shader clocks in %llu ticks -> %.0f MHz\n", iters, ms,
             256.0 * 4 * iters * 64.0 * 512.0 / ms / 1e9, h[2], h[3], h[3] ? 100.0 * h[2] / h[3] : 0.0);
    } }
#endif
  return 0;
}
