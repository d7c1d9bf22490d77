// lmm-hip: the fp32 wide update on a 256 x 256 block tile, in its own translation unit because it must be compiled WITHOUT
// -amdgpu-mfma-vgpr-form=1 (the rest of the library needs that flag for v_mfma_f64_16x16x4_f64; here the 16 accumulator blocks of a wave
// -- 256 registers -- have to live in the AccVGPRs, where fp32 MFMAs issue at full rate).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "lmm_internal.h"
#include "lmm_work_item.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------
// fp32 wide update on a 256 x 256 block tile (round 3).  gemm32_kernel<128> stops at 0.79 of the 157-TFLOP/s fp32 matrix peak: per
// MFMA it reads one operand fragment from LDS and stages 1/4 of a float4 from global memory.  Here ONE workgroup of EIGHT waves per CU
// owns a 256 x 256 tile, each wave a 128 x 64 piece (2 x 4 waves): 8 accumulator blocks of v_mfma_f32_32x32x2_f32 in AccVGPRs (for fp32
// the AccVGPR form issues at full rate -- unlike f64), 6 fragment reads per 8 MFMAs and half the global bytes per flop.
// Same software pipeline as gemm32_kernel (BK = 32 per LDS stage), in GROUPS of two k-pairs: the fragments of the next group are read
// and two staging instructions ride along (the 8 ds_write_b128 of tile t+1 in the first half of a stage, the 8 global_load_dwordx4 of
// tile t+2 in the second), then a burst of 16 MFMAs; the barrier sits before the last group.  133 120 bytes of dynamic LDS.
// C -= A B' on the lower trapezoid (lower) or the full M x N (rows / columns beyond M / N masked per 32 x 32 block; M, N multiples
// of 64); split-K parts combine with fp32 atomics.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void gemm32w_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
                                                         BatchPtr Bb, size_t goffB, int ldb, int M, int N, int K, int lower, int MT,
                                                         int full_items, int splitk) {
  extern __shared__ __attribute__((aligned(16))) float w32_lds[];
  float* C = reinterpret_cast<float*>(Cb.p[blockIdx.y]) + goffC;
  const float* A = reinterpret_cast<const float*>(Ab.p[blockIdx.y]) + goffA;
  const float* B = reinterpret_cast<const float*>(Bb.p[blockIdx.y]) + goffB;
  constexpr int BM = 256, BN = 256, BK = 32, WM = 128, WN = 64, TU = 4, TV = 2;       // 8 waves: 2 (rows) x 4 (columns) of 128 x 64
  constexpr int SA = BM + 4, SB = BN + 4;
  constexpr int NL = 4, KS = 8;                // thread t stages rows 4 (t % 64).. of k-columns t / 64 + 8 q, q < 4, of either operand
  float (*As)[BK * SA] = reinterpret_cast<float (*)[BK * SA]>(w32_lds);
  float (*Bs)[BK * SB] = reinterpret_cast<float (*)[BK * SB]>(w32_lds + 2 * BK * SA);
  int part = 0, nparts = 1, tj = 0, ti = 0;
  gemm_work_item(BM, BN, N, lower, MT, full_items, splitk, part, nparts, ti, tj);
  const int bm = ti * BM, bn = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * WM, wc = (w >> 1) * WN;
  const bool active = (bm + wr < M) && (bn + wc < N) && !(lower && bm + wr + WM - 1 < bn + wc);
  const int nk_all = (K + BK - 1) / BK;
  const int kc0 = (int)((long long)nk_all * part / nparts);
  const int kc1 = (int)((long long)nk_all * (part + 1) / nparts);
  const int nk = kc1 - kc0;
  const int kmax = K - 1;
  int rowa = bm + 4 * (t & 63); if (rowa > M - 4) rowa = M - 4;
  int rowb = bn + 4 * (t & 63); if (rowb > N - 4) rowb = N - 4;
  const int kq = t >> 6;
  const float* gA = A + rowa;
  const float* gB = B + rowb;
  const int sa0 = kq * SA + 4 * (t & 63);
  const int sb0 = kq * SB + 4 * (t & 63);
  float4 ra[NL], rb[NL];
  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < NL; ++q) {
      const int k = k0 + kq + KS * q;
      const int kk = k <= kmax ? k : kmax;
      ra[q] = *reinterpret_cast<const float4*>(gA + (size_t)kk * lda);
      rb[q] = *reinterpret_cast<const float4*>(gB + (size_t)kk * ldb);
      if (k > kmax) { ra[q] = make_float4(0.f, 0.f, 0.f, 0.f); rb[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
  };
  load_tile(kc0);
#pragma unroll
  for (int q = 0; q < NL; ++q) {
    *reinterpret_cast<float4*>(&As[0][sa0 + KS * q * SA]) = ra[q];
    *reinterpret_cast<float4*>(&Bs[0][sb0 + KS * q * SB]) = rb[q];
  }
  load_tile(kc0 + (nk > 1 ? 1 : 0));
  __syncthreads();

  f32x16 acc[TV][TU];
#pragma unroll
  for (int v = 0; v < TV; ++v)
#pragma unroll
    for (int u = 0; u < TU; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[v][u][r] = 0.f;
  const int l31 = lane & 31, lh = lane >> 5;
  const int offA = lh * SA + wr + l31, offB = lh * SB + wc + l31;
  // fragments of TWO k-pairs per set: the loop body is "read the next set (8 ds_read2), 4 staging instructions, then a burst of 32 MFMAs"
  float fu[2][2][TU], fv[2][2][TV];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int u = 0; u < TU; ++u) fu[0][h][u] = As[0][offA + 2 * h * SA + 32 * u];
#pragma unroll
    for (int v = 0; v < TV; ++v) fv[0][h][v] = Bs[0][offB + 2 * h * SB + 32 * v];
  }

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const float* as = &As[buf][0];
    const float* bs = &Bs[buf][0];
    float* asn = &As[buf ^ 1][0];
    float* bsn = &Bs[buf ^ 1][0];
    const int kn2 = kc0 + ((kt + 2 < nk) ? kt + 2 : nk - 1);
#pragma unroll
    for (int kg = 0; kg < BK / 4; ++kg) {          // groups of two k-pairs
      const int cur = kg & 1, nxt = cur ^ 1;
      if (kg == BK / 4 - 1) __syncthreads();       // every fragment of this buffer has been read; tile t+1 is complete in the other
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (kg + 1 < BK / 4) {
#pragma unroll
          for (int u = 0; u < TU; ++u) fu[nxt][h][u] = as[offA + 2 * (2 * (kg + 1) + h) * SA + 32 * u];
#pragma unroll
          for (int v = 0; v < TV; ++v) fv[nxt][h][v] = bs[offB + 2 * (2 * (kg + 1) + h) * SB + 32 * v];
        } else {
#pragma unroll
          for (int u = 0; u < TU; ++u) fu[nxt][h][u] = asn[offA + 2 * h * SA + 32 * u];
#pragma unroll
          for (int v = 0; v < TV; ++v) fv[nxt][h][v] = bsn[offB + 2 * h * SB + 32 * v];
        }
      }
      // two staging instructions per group: slots 0-3 write A of tile t+1, 4-7 write B, 8-11 load A of tile t+2, 12-15 load B
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int sl = 2 * kg + h;
        if (sl < 4) *reinterpret_cast<float4*>(&asn[sa0 + KS * sl * SA]) = ra[sl];
        else if (sl < 8) *reinterpret_cast<float4*>(&bsn[sb0 + KS * (sl - 4) * SB]) = rb[sl - 4];
        else if (sl < 12) {
          const int q = sl - 8, k = kn2 * BK + kq + KS * q;
          ra[q] = *reinterpret_cast<const float4*>(gA + (size_t)(k <= kmax ? k : kmax) * lda);
          if (k > kmax) ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          const int q = sl - 12, k = kn2 * BK + kq + KS * q;
          rb[q] = *reinterpret_cast<const float4*>(gB + (size_t)(k <= kmax ? k : kmax) * ldb);
          if (k > kmax) rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      // (the branch also keeps the MFMAs of a group in a block of their own, behind the group's LDS reads and staging instructions:
      // every instruction that sits between two MFMAs of a wave costs -- see DESIGN.md 4.3)
      if (active) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int v = 0; v < TV; ++v)
#pragma unroll
            for (int uu = 0; uu < TU; ++uu) {
              const int u = (v & 1) ? TU - 1 - uu : uu;
              acc[v][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[cur][h][v], fu[cur][h][u], acc[v][u], 0, 0, 0);
            }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int v = 0; v < TV; ++v)
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      const int r0 = bm + wr + 32 * u, c0 = bn + wc + 32 * v;
      if (r0 >= M || c0 >= N || (lower && r0 + 31 < c0)) continue;       // outside, or a 32 x 32 block strictly above the diagonal
      float* cp = C + (size_t)(c0 + 4 * lh) * ldc + r0 + l31;
      if (nparts == 1) {
        float cv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) cv[r] = cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc];
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = cv[r] - acc[v][u][r];
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) unsafeAtomicAdd(cp + (size_t)((r & 3) + 8 * (r >> 2)) * ldc, -acc[v][u][r]);
      }
    }
}


// true: launched.  false: the caller's 128 x 128 kernel should take the update (too few 256-tiles to fill the device, or switched off).
bool launch_gemm32w(const BatchPtr& C, size_t offC, int ldc, const BatchPtr& A, size_t offA, int lda, const BatchPtr& B, size_t offB, int ldb,
                    int M, int N, int K, int lower, int nb, int cus, bool deterministic, hipStream_t st) {
  static int tile256 = -1;
  constexpr int lds_bytes = 2 * 32 * (260 + 260) * 4;
  if (tile256 < 0) {
    const char* e = getenv("LMM_F32_TILE256"); tile256 = e ? atoi(e) : 1;      // 0: never, 2: whenever the shape allows
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm32w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  if (!tile256 || M < 512 || N < 256 || K < 256 || nb <= 0) return false;
  const int MTw = (M + 255) / 256, NTw = (N + 255) / 256;
  long long Tw = 0;
  for (int tj = 0; tj < NTw; ++tj) Tw += lower ? (MTw - tj) : MTw;
  if (Tw * nb < cus && tile256 != 2) return false;
  const int slw = (cus / nb) > 0 ? (cus / nb) : 1;
  int fullw = (int)(Tw / slw) * slw, skw = 1;
  const int Rw = (int)(Tw - fullw), nk32 = K / 32;
  if (!deterministic && Rw > 0 && Rw <= slw / 2 && nk32 >= 8) { skw = slw / Rw; if (skw > nk32 / 4) skw = nk32 / 4; if (skw < 1) skw = 1; }
  if (skw == 1) fullw = (int)Tw;
  const int itemsw = fullw + (int)(Tw - fullw) * skw;
  hipLaunchKernelGGL(gemm32w_kernel, dim3(itemsw, nb), dim3(512), lds_bytes, st, C, offC, ldc, A, offA, lda, B, offB, ldb, M, N, K, lower, MTw,
                     fullw, skw);
  return true;
}
