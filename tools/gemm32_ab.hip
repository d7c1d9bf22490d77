// A/B of the fp32 update kernel variants in ONE process (C4 level shapes, n = 32768).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/gemm32_ab.hip -o tools/gemm32_ab -lrccl && tools/gemm32_ab [nb]
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
#include <vector>
__global__ void fill_randf(float* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    p[i] = (float)((double)(z & 0xFFFFFFull) / 16777216.0 - 0.5);
  }
}
extern int g_f32_sched;
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 8;
  g_f32 = 1;
  struct Shape { int M, N, K; };
  std::vector<Shape> shapes = {{16384 + 64, 16384, 16384}, {8192 + 64, 8192, 8192}, {12288 + 64, 4096, 4096}, {14336 + 64, 2048, 2048}, {15360 + 64, 1024, 1024}, {16128 + 64, 256, 256}};
  const int ldmax = 16384 + 64 + 16;
  std::vector<float*> Cs(nb), As(nb);
  for (int b = 0; b < nb; ++b) {
    if (hipMalloc(&Cs[b], (size_t)ldmax * 16384 * 4) != hipSuccess || hipMalloc(&As[b], (size_t)ldmax * 16384 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    fill_randf<<<2048, 256>>>(As[b], (size_t)ldmax * 16384, 1u + b);
    fill_randf<<<2048, 256>>>(Cs[b], (size_t)ldmax * 16384, 100u + b);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (const Shape& s : shapes) {
    const int ld = s.M + 16;
    BatchPtr C{}, A{};
    for (int b = 0; b < nb; ++b) { C.p[b] = reinterpret_cast<double*>(Cs[b]); A.p[b] = reinterpret_cast<double*>(As[b]); }
    const double outs = (double)s.N * (s.N + 1) / 2 + (double)(s.M - s.N) * s.N;
    const double fl = 2.0 * s.K * outs * nb;
    float best[2] = {1e30f, 1e30f};
    for (int round = 0; round < 4; ++round)
      for (int v = 0; v < 2; ++v) {
        g_f32_sched = v;
        launch_gemm_nt(C, 0, ld, A, 0, ld, A, 0, ld, s.M, s.N, s.K, 1, false, nb, 0);
        hipEventRecord(e0);
        const int reps = s.K >= 4096 ? 2 : 6;
        for (int r = 0; r < reps; ++r) launch_gemm_nt(C, 0, ld, A, 0, ld, A, 0, ld, s.M, s.N, s.K, 1, false, nb, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        if (ms < best[v]) best[v] = ms;
      }
    printf("f32 M=%5d N=%5d K=%5d nb=%d | compiler-scheduled (shipped): %.3f ms %.2f TF | pinned interleave: %.3f ms %.2f TF\n", s.M, s.N, s.K, nb,
           best[0], fl / best[0] / 1e9, best[1], fl / best[1] / 1e9);
    fflush(stdout);
  }
  return 0;
}
