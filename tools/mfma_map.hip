// Empirical lane map of v_mfma_f64_4x4x4_4b_f64: one-hot A lane x one-hot B lane -> nonzero D lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out) {   // out[la*64*64 + lb*64 + lane]
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      double a = (lane == la) ? 1.0 : 0.0, b = (lane == lb) ? 1.0 : 0.0;
      double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}
int main() {
  double* d; hipMalloc(&d, 64 * 64 * 64 * 8);
  k<<<1, 64>>>(d); hipDeviceSynchronize();
  static double h[64 * 64 * 64];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  // For block 0..: print for la in 0..15, lb in 0..15 the output lane (or -)
  for (int blk = 0; blk < 2; ++blk) {
    printf("block %d: rows = A lane (la), cols = B lane (lb); entry = D lane that became 1\n      ", blk);
    for (int lb = 0; lb < 16; ++lb) printf("%3d ", 16 * blk + lb);
    printf("\n");
    for (int la = 0; la < 16; ++la) {
      printf("la=%2d: ", 16 * blk + la);
      for (int lb = 0; lb < 16; ++lb) {
        int found = -1, cnt = 0;
        for (int l = 0; l < 64; ++l) if (h[((16 * blk + la) * 64 + 16 * blk + lb) * 64 + l] != 0.0) { found = l; cnt++; }
        if (cnt == 0) printf("  . "); else if (cnt == 1) printf("%3d ", found); else printf(" m%d ", cnt);
      }
      printf("\n");
    }
  }
  // cross-block check: A in block 0, B in block 1 -> expect nothing
  int cross = 0;
  for (int la = 0; la < 16; ++la) for (int lb = 16; lb < 64; ++lb) for (int l = 0; l < 64; ++l) if (h[(la * 64 + lb) * 64 + l] != 0.0) cross++;
  printf("cross-block nonzeros: %d\n", cross);
  return 0;
}
