// Probe 3: realistic operand patterns (4 A-frags x 4 B-frags -> 16 accumulators) for both f64 MFMA forms.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NA, int NB>
__global__ __launch_bounds__(256) void p16(double* out, const double* in, int iters) {
  d4 acc[NA * NB];
  double a[NA], b[NB];
#pragma unroll
  for (int q = 0; q < NA * NB; ++q) acc[q] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < NA; ++q) a[q] = in[threadIdx.x + 64 * q];
#pragma unroll
  for (int q = 0; q < NB; ++q) b[q] = in[threadIdx.x + 64 * q + 1024];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i * NB + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * NB + j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int q = 0; q < NA * NB; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NA, int NB>
__global__ __launch_bounds__(256) void p4(double* out, const double* in, int iters) {
  double acc[NA * NB];
  double a[NA], b[NB];
#pragma unroll
  for (int q = 0; q < NA * NB; ++q) acc[q] = 0;
#pragma unroll
  for (int q = 0; q < NA; ++q) a[q] = in[threadIdx.x + 64 * q];
#pragma unroll
  for (int q = 0; q < NB; ++q) b[q] = in[threadIdx.x + 64 * q + 1024];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j)
        acc[i * NB + j] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[i], b[j], acc[i * NB + j], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int q = 0; q < NA * NB; ++q) s += acc[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
  double *out, *in; hipMalloc(&out, 2048 * 256 * 8); hipMalloc(&in, 4096 * 8);
  hipMemset(in, 0, 4096 * 8);
  const int it = 10000;
  for (int blocks : {256, 512, 1024}) {
    float ms;
    ms = timeit([&] { p16<4, 4><<<blocks, 256>>>(out, in, it); });
    printf("16x16x4  4x4 frags (16 acc)  blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 16 * 2048.0 / ms / 1e9);
    ms = timeit([&] { p16<2, 4><<<blocks, 256>>>(out, in, it); });
    printf("16x16x4  2x4 frags ( 8 acc)  blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 8 * 2048.0 / ms / 1e9);
    ms = timeit([&] { p16<1, 8><<<blocks, 256>>>(out, in, it); });
    printf("16x16x4  1x8 frags ( 8 acc)  blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 8 * 2048.0 / ms / 1e9);
    ms = timeit([&] { p4<4, 16><<<blocks, 256>>>(out, in, it); });
    printf("4x4x4_4b 4x16 frags (64 acc) blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 64 * 512.0 / ms / 1e9);
    ms = timeit([&] { p4<4, 4><<<blocks, 256>>>(out, in, it); });
    printf("4x4x4_4b 4x4 frags (16 acc)  blocks=%4d: %7.3f ms  %6.2f TF\n", blocks, ms, blocks * 4.0 * it * 16 * 512.0 / ms / 1e9);
  }
  return 0;
}
