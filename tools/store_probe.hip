// Achievable write-only HBM bandwidth with 16-byte coalesced stores (calibration for the Gram-assembly roofline).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
__global__ void fill(d2* p, size_t n, double v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  d2 x = {v, v + 1.0};
  for (; i < n; i += stride) p[i] = x;
}
// column-strip pattern like gram_kernel: each wave writes 2 columns x 512 B; columns are ld*8 bytes apart
__global__ void strips(double* A, int ld, int nrows, int ncols) {
  const int ti = blockIdx.x, sy = blockIdx.y, t = threadIdx.x;
  const int i0 = ti * 64 + 2 * (t & 31), cg = t >> 5;
  for (int c4 = 0; c4 < 4; ++c4) {
    const int tj = sy * 4 + c4;
    if (tj * 64 >= ncols || ti < tj) break;
    double* out = A + (size_t)(tj * 64 + cg) * ld + i0;
    for (int q = 0; q < 8; ++q) { d2 v = {(double)q, (double)t}; *reinterpret_cast<d2*>(out + (size_t)(8 * q) * ld) = v; }
  }
}
int main() {
  const size_t bytes = 2ull << 30;
  const int n = 16384, ld = n + 64;
  const size_t alloc = (size_t)ld * n * 8 + (1u << 20);     // the strip pattern spans ld*n doubles (> 2 GiB)
  d2* p; if (hipMalloc(&p, alloc) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {2048, 8192, 65536}) {
    fill<<<blocks, 256>>>(p, bytes / 16, 1.0); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 10; ++r) fill<<<blocks, 256>>>(p, bytes / 16, 2.0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("linear 16-B stores, %6d blocks: %.0f GB/s\n", blocks, 10.0 * bytes / ms / 1e6);
  }
  double* A = (double*)p;
  strips<<<dim3(n / 64, n / 256), 256>>>(A, ld, n, n); hipDeviceSynchronize();
  hipEventRecord(e0); for (int r = 0; r < 10; ++r) strips<<<dim3(n / 64, n / 256), 256>>>(A, ld, n, n); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("lower-triangle strips n=%d (no math): %.4f ms/launch  %.0f GB/s\n", n, ms / 10, 10.0 * n * (n + 1.0) / 2 * 8 / ms / 1e6);
  return 0;
}
