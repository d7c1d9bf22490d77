// Times diag64_kernel alone (1 and 8 matrices per launch).
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
#include <vector>
template <bool WPART, bool SYNC, bool RECIP>
__global__ __launch_bounds__(256) void diag64_exp(BatchPtr Ab, size_t offA, int ld, BatchPtr Wb, size_t offW,
                                                     int gcol0, int n_real, BatchInfo infob) {
  double* __restrict__ A = Ab.p[blockIdx.x] + offA;
  double* __restrict__ W = Wb.p[blockIdx.x] + offW;
  int* __restrict__ info = infob.p[blockIdx.x];
  // Register-resident elimination: thread (i = t & 63, q = t >> 6) owns row i, columns 16q..16q+15 of both the
  // S part (the block being factored) and the W part (identity -> L1^-1).  Per pivot j the owners publish, through
  // double-buffered LDS, column j (colb: every row's multiplier numerator; cmsk: the same masked to rows > j, which by
  // symmetry is the pivot row of the S part) and row j of the W part; one barrier per pivot; all register indices are
  // compile-time (inner 16 pivots unrolled), so nothing spills to scratch.
  __shared__ __attribute__((aligned(16))) double colb[2][64];
  __shared__ __attribute__((aligned(16))) double cmsk[2][64];
  __shared__ __attribute__((aligned(16))) double roww[2][64];
  __shared__ double dd[64];
  const int t = threadIdx.x, i = t & 63, q = t >> 6;
  double s[16], w[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int k = 16 * q + c;
    s[c] = (i >= k) ? A[(size_t)k * ld + i] : 0.0;
    w[c] = (i == k) ? 1.0 : 0.0;
  }
  for (int jb = 0; jb < 4; ++jb) {
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
      const int j = 16 * jb + jj, b = jj & 1;
      if (q == jb) {                       // owners of column j (wave-uniform)
        colb[b][i] = s[jj];
        cmsk[b][i] = (i > j) ? s[jj] : 0.0;
      }
      if (WPART && i == j) {                        // owners of row j of the W part
#pragma unroll
        for (int c = 0; c < 16; ++c) roww[b][16 * q + c] = w[c];
      }
      if (SYNC) __syncthreads();
      const double dj = colb[b][j];
      const double ci = colb[b][i];
      const d2* rs = reinterpret_cast<const d2*>(&cmsk[b][16 * q]);
      const d2* rw = reinterpret_cast<const d2*>(&roww[b][16 * q]);
      d2 ps[8], pw[8];
#pragma unroll
      for (int c2 = 0; c2 < 8; ++c2) { ps[c2] = rs[c2]; pw[c2] = rw[c2]; }     // issued before the reciprocal: overlaps it
      if (t == 0) dd[j] = dj;
      // 1/d_j by v_rcp_f64 + two Newton steps (<= 1-2 ulp; an IEEE division is ~3x the dependent latency on the
      // pivot-to-pivot critical path); non-positive pivots are detected after the loop from dd[].
      double rinv = RECIP ? __builtin_amdgcn_rcp(dj) : 1.0;
      rinv = __builtin_fma(__builtin_fma(-dj, rinv, 1.0), rinv, rinv);
      rinv = __builtin_fma(__builtin_fma(-dj, rinv, 1.0), rinv, rinv);
      const double mult = (i > j) ? ci * rinv : 0.0;
#pragma unroll
      for (int c2 = 0; c2 < 8; ++c2) {
        s[2 * c2] = __builtin_fma(-mult, ps[c2].x, s[2 * c2]);
        s[2 * c2 + 1] = __builtin_fma(-mult, ps[c2].y, s[2 * c2 + 1]);
        if (WPART) { w[2 * c2] = __builtin_fma(-mult, pw[c2].x, w[2 * c2]);
        w[2 * c2 + 1] = __builtin_fma(-mult, pw[c2].y, w[2 * c2 + 1]); }
      }
    }
  }
  __syncthreads();
  if (t < 64) {                                    // LAPACK-style info: first non-positive (or NaN) pivot, 1-based
    const bool bad = !(dd[t] > 0.0) && (gcol0 + t < n_real);
    const unsigned long long mask = __ballot(bad);
    if (t == 0 && mask != 0ull) atomicCAS(info, 0, gcol0 + __builtin_ctzll(mask) + 1);
  }
  __syncthreads();
  const double rsi = 1.0 / sqrt(dd[i]);           // row scale of W = D^-1/2 L1^-1
#pragma unroll
  for (int c = 0; c < 16; ++c) {
    const int k = 16 * q + c;
    if (i >= k) {
      const double lk = sqrt(dd[k]);
      A[(size_t)k * ld + i] = (i == k) ? lk : s[c] / lk;
    }
    W[k * 64 + i] = (i >= k) ? w[c] * rsi : 0.0;
  }
}


template <bool W_, bool S_, bool R_> void run_exp(const char* name, BatchPtr a, BatchPtr w, BatchInfo inf, double* A, int* info, const std::vector<double>& h) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0;
  for (int r = 0; r < 20; ++r) {
    hipMemcpy(A, h.data(), 8 * 4096 * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 32);
    hipEventRecord(e0); hipLaunchKernelGGL((diag64_exp<W_, S_, R_>), dim3(1), dim3(256), 0, 0, a, (size_t)0, 64, w, (size_t)0, 0, 64, inf); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 5) tot += ms;
  }
  printf("%-40s %.2f us\n", name, tot / 15 * 1e3);
}
int main() {
  const int ld = 64;
  double *A, *W; int* info;
  hipMalloc(&A, 8 * 4096 * 8); hipMalloc(&W, 8 * 4096 * 8); hipMalloc(&info, 8 * 4);
  std::vector<double> h(8 * 4096, 0.0);
  for (int b = 0; b < 8; ++b) for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) h[b * 4096 + j * 64 + i] = (i == j) ? 70.0 : 1.0 / (1 + i - j);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nb : {1, 8}) {
    BatchPtr a{}, w{}; BatchInfo inf{};
    for (int b = 0; b < nb; ++b) { a.p[b] = A + b * 4096; w.p[b] = W + b * 4096; inf.p[b] = info + b; }
    float tot = 0;
    for (int r = 0; r < 20; ++r) {
      hipMemcpy(A, h.data(), 8 * 4096 * 8, hipMemcpyHostToDevice); hipMemset(info, 0, 32);
      hipEventRecord(e0); launch_diag64(a, 0, ld, w, 0, 0, 64, inf, nb, 0); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 5) tot += ms;
    }
    printf("diag64 x%d: %.2f us per launch\n", nb, tot / 15 * 1e3);
  }
  {
    BatchPtr a{}, w{}; BatchInfo inf{}; a.p[0] = A; w.p[0] = W; inf.p[0] = info;
    run_exp<true, true, true>("full (W part, barriers, reciprocal)", a, w, inf, A, info, h);
    run_exp<false, true, true>("no W part", a, w, inf, A, info, h);
    run_exp<true, false, true>("no barriers (wrong results)", a, w, inf, A, info, h);
    run_exp<true, true, false>("no reciprocal (wrong results)", a, w, inf, A, info, h);
    run_exp<false, false, false>("no W, no barriers, no reciprocal", a, w, inf, A, info, h);
  }
  return 0;
}
