// Probe 4 (round 2): v_mfma_f64_16x16x4_f64 issue rate as a function of the OPERAND ORDER.  rocBLAS' gfx950 dgemm kernels
// (Tensile MT128x128x16 MI16x16x4) walk the 8 x 2 fragment grid of a wave in serpentine order, so that every MFMA shares its A
// or its B source registers with the previous one; round 1 measured the same instruction at 36 TF (16 acc, 4 + 4 operands in
// row-major order) and 58 TF (one operand pair).  This measures: serpentine 8x2, row-major 8x2, serpentine 4x4, 4x4x4_4b.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0)
template <int MODE>
__global__ __launch_bounds__(256) void probe(double* out, int iters) {
  d4 c[16];
  double a[8], b[4];
#pragma unroll
  for (int q = 0; q < 16; ++q) c[q] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 8; ++q) a[q] = 1.0 + (threadIdx.x + 7 * q) * 1e-9;
#pragma unroll
  for (int q = 0; q < 4; ++q) b[q] = 1.0 - (threadIdx.x + 3 * q) * 1e-9;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // serpentine over an 8 (A) x 2 (B) grid: consecutive MFMAs share A or B
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if ((i & 1) == 0) { MF(c[2 * i], a[i], b[0]); MF(c[2 * i + 1], a[i], b[1]); }
        else { MF(c[2 * i + 1], a[i], b[1]); MF(c[2 * i], a[i], b[0]); }
      }
    } else if (MODE == 1) {   // row-major 8 x 2, B alternates every instruction, A changes every second one
#pragma unroll
      for (int i = 0; i < 8; ++i) { MF(c[2 * i], a[i], b[0]); MF(c[2 * i + 1], a[i], b[1]); }
    } else if (MODE == 2) {   // serpentine over a 4 x 4 grid
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { const int j = (i & 1) ? 3 - jj : jj; MF(c[4 * i + j], a[i], b[j]); }
    } else {                  // no sharing at all: both operands change every instruction
#pragma unroll
      for (int q = 0; q < 16; ++q) MF(c[q], a[q & 7], b[(q + (q >> 2)) & 3]);
    }
  }
  double s = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += c[q][0] + c[q][1] + c[q][2] + c[q][3];
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
  const int it = 20000;
  const char* names[4] = {"serpentine 8x2", "row-major 8x2 ", "serpentine 4x4", "no sharing    "};
  for (int blocks : {256, 512}) {
    float ms[4];
    ms[0] = timeit([&] { probe<0><<<blocks, 256>>>(out, it); });
    ms[1] = timeit([&] { probe<1><<<blocks, 256>>>(out, it); });
    ms[2] = timeit([&] { probe<2><<<blocks, 256>>>(out, it); });
    ms[3] = timeit([&] { probe<3><<<blocks, 256>>>(out, it); });
    for (int m = 0; m < 4; ++m)
      printf("16x16x4 f64 %s  %d wave(s)/SIMD: %8.3f ms  %6.2f TFLOP/s\n", names[m], blocks / 256, ms[m], blocks * 4.0 * it * 16 * 2048.0 / ms[m] / 1e9);
  }
  return 0;
}
