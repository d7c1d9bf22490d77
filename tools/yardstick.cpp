// Same-box yardstick (round-1 review item 8): the vendor libraries on the shapes this library's kernels run -- NEVER on the product
// path, only a comparison recorded under profiles/.
//   rocblas_dgemm NT / rocblas_dsyrk at the K = 8192 level shape of the C2 factorisation, rocblas_sgemm for the fp32 mode,
//   rocsolver_dpotrf (one matrix) and rocsolver_dpotrf_batched (8 matrices) at n = 16384.
//   hipcc --offload-arch=gfx950 -O2 tools/yardstick.cpp -o tools/yardstick -lrocblas -lrocsolver && tools/yardstick
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("%s failed: %d (line %d)\n", #x, (int)e_, __LINE__); return 1; } } while (0)
__global__ void fill_rand(double* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    p[i] = ((double)(z & 0xFFFFFFFFFFFFFull) / 4503599627370496.0) - 0.5;
  }
}
__global__ void fill_randf(float* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32;
    p[i] = (float)(((double)(z & 0xFFFFFFFFFFFFFull) / 4503599627370496.0) - 0.5);
  }
}
// SPD: A = 0.25 everywhere-random symmetric-ish + n on the diagonal (diagonally dominant)
__global__ void make_spd(double* A, int n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < (size_t)n) A[i * n + i] = (double)n;
}
int main() {
  rocblas_handle h; CK(rocblas_create_handle(&h));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int n = 8192, k = 8192;
  double *A, *B, *C; CK(hipMalloc(&A, (size_t)n * k * 8)); CK(hipMalloc(&B, (size_t)n * k * 8)); CK(hipMalloc(&C, (size_t)n * n * 8));
  fill_rand<<<2048, 256>>>(A, (size_t)n * k, 1); fill_rand<<<2048, 256>>>(B, (size_t)n * k, 2); fill_rand<<<2048, 256>>>(C, (size_t)n * n, 3);
  const double alpha = -1.0, beta = 1.0; float ms;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) CK(rocblas_dgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &alpha, A, n, B, n, &beta, C, n));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    if (rep) printf("rocblas_dgemm NT  %d x %d x %d: %.3f ms  %.2f TFLOP/s\n", n, n, k, ms, 2.0 * n * n * (double)k / ms / 1e9);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) CK(rocblas_dsyrk(h, rocblas_fill_lower, rocblas_operation_none, n, k, &alpha, A, n, &beta, C, n));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    if (rep) printf("rocblas_dsyrk lower n = %d, k = %d: %.3f ms  %.2f TFLOP/s (n (n + 1) k flops)\n", n, k, ms, (double)n * (n + 1) * k / ms / 1e9);
  }
  {
    float *Af = (float*)A, *Bf = (float*)B, *Cf = (float*)C; const float al = -1.f, be = 1.f;
    fill_randf<<<2048, 256>>>(Af, (size_t)n * k, 1); fill_randf<<<2048, 256>>>(Bf, (size_t)n * k, 2); fill_randf<<<2048, 256>>>(Cf, (size_t)n * n, 3);
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int r = 0; r < 3; ++r) CK(rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_transpose, n, n, k, &al, Af, n, Bf, n, &be, Cf, n));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      if (rep) printf("rocblas_sgemm NT  %d x %d x %d: %.3f ms  %.2f TFLOP/s\n", n, n, k, ms, 2.0 * n * n * (double)k / ms / 1e9);
    }
  }
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  const int N = 16384, NB = 8;
  std::vector<double*> mats(NB);
  for (int b = 0; b < NB; ++b) { CK(hipMalloc(&mats[b], (size_t)N * N * 8)); }
  double** dptr; CK(hipMalloc(&dptr, NB * sizeof(double*))); CK(hipMemcpy(dptr, mats.data(), NB * sizeof(double*), hipMemcpyHostToDevice));
  rocblas_int* info; CK(hipMalloc(&info, NB * sizeof(rocblas_int)));
  auto reset = [&]() { for (int b = 0; b < NB; ++b) { fill_rand<<<2048, 256>>>(mats[b], (size_t)N * N, 7 + b); make_spd<<<(N + 255) / 256, 256>>>(mats[b], N); } return hipDeviceSynchronize(); };
  for (int rep = 0; rep < 2; ++rep) {
    CK(reset());
    CK(hipEventRecord(e0));
    CK(rocsolver_dpotrf(h, rocblas_fill_lower, N, mats[0], N, info));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("rocsolver_dpotrf n = %d (1 matrix): %.2f ms  %.2f TFLOP/s (n^3/3)\n", N, ms, (double)N * N * N / 3.0 / ms / 1e9);
    CK(reset());
    CK(hipEventRecord(e0));
    CK(rocsolver_dpotrf_batched(h, rocblas_fill_lower, N, dptr, N, info, NB));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("rocsolver_dpotrf_batched n = %d x %d: %.2f ms  %.2f TFLOP/s (batch n^3/3)\n", N, NB, ms, NB * (double)N * N * N / 3.0 / ms / 1e9);
  }
  return 0;
}
