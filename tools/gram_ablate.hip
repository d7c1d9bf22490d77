// Times gram_kernel alone with HIP events (10 launches back to back).
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 16384;
  const int NC = (n + 63) / 64 * 64, NR = NC + 64, ld = NR;
  double *A, *x; hipMalloc(&A, (size_t)ld * NC * 8); hipMalloc(&x, n * 8);
  std::vector<double> hx(n); for (int i = 0; i < n; ++i) hx[i] = i * (20.0 / 575.0);
  hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int kind = 0; kind < 3; ++kind) {
    GramArgs a{}; a.A = A; a.ld = ld; a.nrows = NR; a.ncols = NC; a.x = x; a.d = 1; a.n = n; a.kind = kind; a.var = 1.0; a.inv_ls = 1.0;
    a.diag_add = 0.1; a.pad_diag = 1.0;
    launch_gram(a, 0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) launch_gram(a, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("kind %d n=%d: %.4f ms  %.0f GB/s (n(n+1)/2*8 B)\n", kind, n, ms, (double)n * (n + 1) / 2 * 8 / ms / 1e6);
  }
  return 0;
}
