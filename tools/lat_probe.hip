// Latency probe for the diag64 design (gfx950): cycles per DEPENDENT operation in one wave (s_memtime deltas / chain length).
//   hipcc --offload-arch=gfx950 -O3 tools/lat_probe.hip -o tools/lat_probe && tools/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
__global__ void probe(double* out, long long* cyc, double seed) {
  __shared__ double lds[1024];
  const int t = threadIdx.x;
  double x = seed + t * 1e-3, y = 1.0000001, acc = 0.0;
  long long t0, t1; int k = 0;
  // 0: dependent f64 FMA
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = __builtin_fma(x, y, 1e-9);
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 1: dependent f64 mul
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = x * y;
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 2: dependent v_rsq_f64
  x = 1.5 + t;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = __builtin_amdgcn_rsq(x) + 1.0;      // rsq + add: subtract the add (row 0) when reading
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 3: dependent v_rcp_f64 (+ add)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = __builtin_amdgcn_rcp(x) + 1.0;
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 4: full rsq with two Newton steps (as diag64 does), dependent
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      double r = __builtin_amdgcn_rsq(x); const double h = 0.5 * x;
      r = r * __builtin_fma(-h * r, r, 1.5); r = r * __builtin_fma(-h * r, r, 1.5);
      x = r + 1.0;
    }
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 5: readlane (uniform) of a double + fma dependent
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
      x = __builtin_fma(__hiloint2double(hi, lo), 1e-3, x);
    }
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 6: ds_bpermute (shfl) of a double, dependent
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = __shfl(x, (t + 17) & 63, 64) + 1e-9;
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 7: LDS write -> read round trip within the wave (dependent)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lds[t] = x; __builtin_amdgcn_wave_barrier(); x = lds[(t + 1) & 63] + 1e-9; __builtin_amdgcn_wave_barrier(); }
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 8: LDS write -> __syncthreads -> read (whole workgroup)
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { lds[t] = x; __syncthreads(); x = lds[(t + 64) & (blockDim.x - 1)] + 1e-9; __syncthreads(); }
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  // 9: DPP row broadcast style: __builtin_amdgcn_mov_dpp / use ds_swizzle? measure v_permlane via __shfl_xor 16 (bpermute again) -> skip
  // 9: 8 INDEPENDENT f64 FMA chains (issue rate)
  double z[8]; for (int j = 0; j < 8; ++j) z[j] = x + j;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = __builtin_fma(z[j], y, 1e-9);
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; for (int j = 0; j < 8; ++j) acc += z[j];
  // 10: dependent MFMA 16x16x4 f64 (D feeds C)
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c = {x, x, x, x};
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, c, 0, 0, 0);
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += c[0] + c[1] + c[2] + c[3];
  // 11: dependent MFMA 4x4x4 f64
  double c1 = x;
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(y, y, c1, 0, 0, 0);
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += c1;
  // 12: f32 rsq + convert + one... f64 sqrt via v_sqrt_f64 dependent
  t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N / 8; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x = __builtin_amdgcn_sqrt(x) + 1.0;
  }
  t1 = __builtin_readcyclecounter(); if (t == 0) cyc[k] = t1 - t0; k++; acc += x;
  out[blockIdx.x * blockDim.x + t] = acc;
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8 * 32);
  const char* names[] = {"dep fma f64", "dep mul f64", "dep v_rsq_f64 + add", "dep v_rcp_f64 + add", "dep rsq + 2 Newton + add", "readlane x2 + fma", "ds_bpermute x2 + add",
                         "LDS write->read (wave) + add", "LDS write->barrier->read->barrier (wg) + add", "8 independent fma chains (per fma)", "dep mfma 16x16x4 f64", "dep mfma 4x4x4 f64", "dep v_sqrt_f64 + add"};
  for (int threads : {64, 256}) {
    hipMemset(cyc, 0, 8 * 32);
    probe<<<1, threads>>>(out, cyc, 1.25); hipDeviceSynchronize();
    probe<<<1, threads>>>(out, cyc, 1.25); hipDeviceSynchronize();
    long long h[32]; hipMemcpy(h, cyc, 8 * 32, hipMemcpyDeviceToHost);
    printf("threads=%d (s_memtime ticks at 100 MHz? -> reported raw and per op)\n", threads);
    for (int k = 0; k < 13; ++k) printf("  %-46s %8lld ticks total, %.2f per op\n", names[k], h[k], (double)h[k] / N);
  }
  return 0;
}
