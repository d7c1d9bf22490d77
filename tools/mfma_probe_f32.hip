// Issue rate of the FP32 MFMAs on gfx950 (peak 157.3 TFLOP/s = 64 FLOP/clk/SIMD): 32x32x2 and 16x16x4, 4 independent accumulators per wave,
// one wave per SIMD (256 threads, 1 workgroup per CU) and two (2 workgroups per CU).  Build twice: with and without
// -mllvm -amdgpu-mfma-vgpr-form=1 (architectural-VGPR vs AccVGPR accumulators).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k32(float* out, int iters) {
  f16v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
  }
  float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(float* out, int iters) {
  f4v a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0}, a4 = {0}, a5 = {0}, a6 = {0}, a7 = {0};
  const float x = threadIdx.x * 1e-3f, y = 1.0f + x;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
    a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a5, 0, 0, 0);
    a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a6, 0, 0, 0);
    a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a7, 0, 0, 0);
  }
  float s = 0; for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r] + a4[r] + a5[r] + a6[r] + a7[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  float* out; hipMalloc(&out, 4 * 256 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs : {256, 512, 1024}) {
    for (int which = 0; which < 2; ++which) {
      auto run = [&]() { if (which == 0) k32<<<wgs, 256>>>(out, iters); else k16<<<wgs, 256>>>(out, iters); };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (which == 0 ? 4.0 * 32 * 32 * 2 * 2 : 8.0 * 16 * 16 * 4 * 2) * iters * (double)wgs * 4;
      printf("%s  %4d workgroups of 4 waves: %.1f TFLOP/s\n", which == 0 ? "v_mfma_f32_32x32x2_f32" : "v_mfma_f32_16x16x4_f32", wgs, flops / ms / 1e9);
    }
  }
  return 0;
}
