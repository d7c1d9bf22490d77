// Probe 2: f64 MFMA issue rate with distinct operands / more accumulators / the 4x4x4 form.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int NOPS>
__global__ __launch_bounds__(256) void probe16(double* out, int iters) {
  d4 acc[NACC];
  double a[NOPS], b[NOPS];
#pragma unroll
  for (int q = 0; q < NACC; ++q) acc[q] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < NOPS; ++q) { a[q] = 1.0 + (threadIdx.x + q) * 1e-9; b[q] = 1.0 - (threadIdx.x + 3 * q) * 1e-9; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      acc[q % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q % NOPS], b[(q / 4) % NOPS], acc[q % NACC], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void probe4x4(double* out, int iters) {
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[q], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  double* out; hipMalloc(&out, 2048 * 256 * 8);
  const int it = 10000;
  for (int blocks : {512, 1024, 2048}) {
    float ms = timeit([&] { probe16<16, 4><<<blocks, 256>>>(out, it); });
    printf("16x16x4 16acc 4+4 operands blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 16 * 2048.0 / ms / 1e9);
    ms = timeit([&] { probe16<8, 1><<<blocks, 256>>>(out, it); });
    printf("16x16x4  8acc same operands blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 16 * 2048.0 / ms / 1e9);
    ms = timeit([&] { probe4x4<<<blocks, 256>>>(out, it); });
    printf("4x4x4_4b 16acc               blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 16 * 512.0 / ms / 1e9);
  }
  return 0;
}
