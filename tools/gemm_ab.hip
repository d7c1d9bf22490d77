// A/B of the trailing-update kernel variants in ONE process (interleaved rounds; cdna_hip_programming.md 5.4 rule 24):
//   variant 0: gemm44_kernel (v_mfma_f64_4x4x4_4b)        variant 1: gemm16_kernel (v_mfma_f64_16x16x4, VGPR accumulators)
//   (the LDS-flag vs s_barrier comparison of the 4x4x4 kernel is in git history: profiles/r02/gemm_ab_flags_vs_barrier_*.log)
// Also checks the two variants against each other on random data (same inputs -> same C up to split-K atomics order).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/gemm_ab.hip -o tools/gemm_ab && tools/gemm_ab [nb]
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
__global__ void fill_rand(double* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    p[i] = ((double)(z & 0xFFFFFFFFFFFFFull) / 4503599627370496.0) - 0.5;
  }
}
__global__ void max_abs_diff(const double* a, const double* b, size_t n, double* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  double m = 0.0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) { double d = fabs(a[i] - b[i]); if (!(d <= m)) m = d; }
  for (int o = 32; o > 0; o >>= 1) { double x = __shfl_down(m, o, 64); if (!(x <= m)) m = x; }
  if ((threadIdx.x & 63) == 0) { unsigned long long* p = (unsigned long long*)out; atomicMax(p, (unsigned long long)__double_as_longlong(m)); }
}
extern int g_gemm_flags, g_gemm_m16;
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 8;
  const int only_shape = argc > 2 ? atoi(argv[2]) : -1;      // profile runs: one shape ...
  const int only_variant = argc > 3 ? atoi(argv[3]) : -1;    // ... one variant, no correctness pass
  struct Shape { int M, N, K, lower; };
  std::vector<Shape> shapes = {{8192 + 64, 8192, 8192, 1}, {12288 + 64, 4096, 4096, 1}, {4096 + 64, 4096, 4096, 1}, {14336 + 64, 2048, 2048, 1},
                               {15360 + 64, 1024, 1024, 1}, {15872 + 64, 512, 512, 1}, {16128 + 64, 256, 256, 1}, {16256 + 64, 128, 128, 1},
                               {8192, 8192, 8192, 0}};
  const int ldmax = 16384 + 64 + 16;
  std::vector<double*> Cs(nb), As(nb), C2(nb);
  for (int b = 0; b < nb; ++b) {
    if (hipMalloc(&Cs[b], (size_t)ldmax * 8192 * 8) != hipSuccess || hipMalloc(&As[b], (size_t)ldmax * 8192 * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    fill_rand<<<2048, 256>>>(As[b], (size_t)ldmax * 8192, 1u + b);
  }
  hipMalloc(&C2[0], (size_t)ldmax * 8192 * 8);
  double* dmax; hipMalloc(&dmax, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t si = 0; si < shapes.size(); ++si) {
    if (only_shape >= 0 && (int)si != only_shape) continue;
    const Shape& s = shapes[si];
    const int ld = s.M + 16;
    BatchPtr C{}, A{};
    for (int b = 0; b < nb; ++b) { C.p[b] = Cs[b]; A.p[b] = As[b]; }
    const double outs = s.lower ? ((double)s.N * (s.N + 1) / 2 + (double)(s.M - s.N) * s.N) : (double)s.M * s.N;
    const double fl = 2.0 * s.K * outs * nb;
    // correctness: variants 1, 2 vs variant 0 on matrix 0
    double hmaxv[3] = {0, 0, 0};
    if (only_variant < 0)
    for (int v = 0; v < 3; ++v) {
      g_gemm_flags = 0; g_gemm_m16 = v == 0 ? 0 : (v == 1 ? 1 : 2);
      double* Cv = v ? C2[0] : Cs[0];
      fill_rand<<<2048, 256>>>(Cv, (size_t)ld * s.N, 77u);
      launch_gemm_nt(Cv, ld, As[0], ld, As[0], ld, s.M, s.N, s.K, s.lower, false, 0);
      if (v) {
        hipMemset(dmax, 0, 8);
        max_abs_diff<<<1024, 256>>>(Cs[0], C2[0], (size_t)ld * s.N, dmax);
        hipMemcpy(&hmaxv[v], dmax, 8, hipMemcpyDeviceToHost);
      }
    }
    const double hmax = hmaxv[1] > hmaxv[2] ? hmaxv[1] : hmaxv[2];
    double best[3] = {1e30, 1e30, 1e30}, med[3][5];
    for (int round = 0; round < 5; ++round)
      for (int v = 0; v < 3; ++v) {
        if (only_variant >= 0 && v != only_variant) continue;
        g_gemm_flags = 0; g_gemm_m16 = v == 0 ? 0 : (v == 1 ? 1 : 2);
        launch_gemm_nt(C, 0, ld, A, 0, ld, A, 0, ld, s.M, s.N, s.K, s.lower, false, nb, 0);   // warm
        hipEventRecord(e0);
        const int reps = s.K >= 4096 ? 2 : 6;
        for (int r = 0; r < reps; ++r) launch_gemm_nt(C, 0, ld, A, 0, ld, A, 0, ld, s.M, s.N, s.K, s.lower, false, nb, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        med[v][round] = ms; if (ms < best[v]) best[v] = ms;
      }
    printf("M=%5d N=%5d K=%5d lower=%d nb=%d | 4x4x4 (round 1): %.3f ms %.2f TF | 16x16x4 in the round-1 loop: %.3f ms %.2f TF | 16x16x4 pipelined (shipped): %.3f ms %.2f TF (x%.3f) | maxdiff %.3e\n",
           s.M, s.N, s.K, s.lower, nb, best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9, best[2], fl / best[2] / 1e9, best[0] / best[2], hmax);
    fflush(stdout);
  }
  return 0;
}
