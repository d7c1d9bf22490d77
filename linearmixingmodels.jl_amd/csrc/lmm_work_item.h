// Device-side work-item scheduler of the update kernels, shared by lmm_kernels.hip (compiled with MFMA accumulators in architectural
// VGPRs: the f64 form that issues at full rate) and lmm_kernels_f32w.hip (compiled WITHOUT that flag: the 256 accumulators of the
// 256 x 256 fp32 tile live in AccVGPRs).
#pragma once
#include <hip/hip_runtime.h>

// Work item -> (tile, k-part) of the update kernels (shared by the f64 and f32 variants).
__device__ __forceinline__ void gemm_work_item_from(int item, int c0, int BM, int BN, int N, int lower, int MT, int full_items, int splitk,
                                                    int& part, int& nparts, int& ti, int& tj, int xcds = 8);
__device__ __forceinline__ void gemm_work_item(int BM, int BN, int N, int lower, int MT, int full_items, int splitk, int& part,
                                               int& nparts, int& ti, int& tj) {
  gemm_work_item_from(blockIdx.x, 0, BM, BN, N, lower, MT, full_items, splitk, part, nparts, ti, tj);
}
// item: work-item index; c0: first column tile of the enumeration (the fused node kernel hands column tile 0 out separately)
// xcds: over how many XCDs consecutive items of ONE matrix are dealt (8 when blockIdx.x = item; the node kernel, whose dispatch order
// is matrix-fastest, passes 8 / gcd(batch size, 8): with 8 or 16 matrices per batch a matrix stays on one XCD and no remap is needed)
__device__ __forceinline__ void gemm_work_item_from(int item, int c0, int BM, int BN, int N, int lower, int MT, int full_items, int splitk,
                                                    int& part, int& nparts, int& ti, int& tj, int xcds) {
  // Work item -> (tile, k-part).  Tiles on/below the diagonal are enumerated column by column; the first
  // `full_items` tiles run their whole K range, the remaining ones (the last, partial round of workgroups over the
  // chip) are split `splitk`-ways along K and combined with f64 atomics, so the launch ends without a long tail.
  int tile = item;
  part = 0; nparts = 1;
  {
    // XCD-aware order (speed only): workgroups are dealt round-robin over the 8 XCDs, so workgroup 8q + g runs on the XCD
    // of group g.  Give group g the contiguous logical tiles [32g, 32g + 32) of every round of 256: with bands of 8 row
    // tiles that is an 8-row x 4-column patch of C per XCD (12 operand panels through that XCD's L2 instead of 18+;
    // measured FETCH_SIZE of the K = 8192 SYRK: 11.9 -> 7.3 GB; 64-tile patches measured no better).
    const int win = 32 * xcds, nfull = (full_items / win) * win;
    if (tile < nfull && xcds > 1) {
      const int g8 = tile % xcds, q = tile / xcds;
      tile = (q >> 5) * win + g8 * 32 + (q & 31);
    }
  }
  if (tile >= full_items) {
    const int r = tile - full_items;
    tile = full_items + r / splitk; part = r - (r / splitk) * splitk; nparts = splitk;
  }
  // Tile order: bands of 8 row tiles, column-major inside a band, so that the ~256 tiles in flight form a compact
  // 8 x 32 patch of C (40 operand panels instead of ~68 for plain column-major order) and 32 consecutive logical
  // tiles are an 8 x 4 patch (see the XCD remap above): fewer re-reads of A/B.
  tj = 0; ti = 0;
  {
    const int NTc = (N + BN - 1) / BN;
    int rem = tile;
    for (int r0 = 0; r0 < MT; r0 += 8) {
      const int r1 = (r0 + 8 < MT) ? r0 + 8 : MT;             // band rows [r0, r1)
      bool found = false;
      for (int c = c0; c < NTc; ++c) {
        int first = lower ? (c * BN) / BM : 0;                 // first active row tile of column c
        if (first < r0) first = r0;
        const int cnt = r1 - first;
        if (cnt <= 0) break;                                   // columns further right are above the diagonal for this band
        if (rem < cnt) { tj = c; ti = first + rem; found = true; break; }
        rem -= cnt;
      }
      if (found) break;
    }
  }
}
