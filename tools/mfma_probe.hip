// Probe: FP64 MFMA / VALU issue rates and the clock the chip holds (gfx950).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void probe(double* out, unsigned long long* stamps, int iters) {
  d4 acc[8];
  double v[16];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = threadIdx.x * 1e-3 + q;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < NM; ++q) acc[q & 7] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q & 7], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < NV; ++q) v[q & 15] = __builtin_fma(v[q & 15], a, b);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
#pragma unroll
  for (int q = 0; q < 16; ++q) s += v[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
}

template <int NM, int NV>
void run(const char* name, int blocks, int iters) {
  double* out; unsigned long long* st;
  hipMalloc(&out, blocks * 256 * 8); hipMalloc(&st, blocks * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<NM, NV><<<blocks, 256>>>(out, st, iters / 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<NM, NV><<<blocks, 256>>>(out, st, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), st, blocks * 8 * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w]); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  double waves = blocks * 4.0;
  double mf = waves * iters * NM * 2048.0, vf = waves * iters * NV * 128.0;
  printf("%-28s blocks=%4d  %8.3f ms  mfma %6.2f TF  valu %6.2f TF  total %6.2f TF | median wave cycles/iter %.1f  clock %.0f MHz\n",
         name, blocks, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, cyc[cyc.size() / 2] / iters, clk[clk.size() / 2]);
  hipFree(out); hipFree(st);
}

int main() {
  const int it = 20000;
  run<8, 0>("mfma only, 1 wave/SIMD", 256, it);
  run<8, 0>("mfma only, 2 waves/SIMD", 512, it);
  run<8, 0>("mfma only, 4 waves/SIMD", 1024, it);
  run<0, 32>("valu fma only, 2 waves/SIMD", 512, it);
  run<0, 32>("valu fma only, 4 waves/SIMD", 1024, it);
  run<8, 16>("mfma 8 + valu 16, 2 w/SIMD", 512, it);
  run<8, 32>("mfma 8 + valu 32, 2 w/SIMD", 512, it);
  run<8, 64>("mfma 8 + valu 64, 2 w/SIMD", 512, it);
  run<8, 128>("mfma 8 + valu 128, 2 w/SIMD", 512, it);
  run<8, 128>("mfma 8 + valu 128, 4 w/SIMD", 1024, it);
  return 0;
}
