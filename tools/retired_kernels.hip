// tools/retired_kernels.hip -- kernel families that are NO LONGER part of liblmm_hip.so (moved out of csrc/lmm_kernels.hip in round 4).
// Kept as the record the measurements under profiles/r01 .. r03 refer to; not compiled by __graft_entry__.build().  To A/B one of them
// again, paste it back behind the helpers it uses (MatIO, BatchPtr, gemm_work_item) -- tools/gemm_ab.hip, tools/diag_ab.hip and
// tools/gemm16_ablate.sh were written against the file as it stood at the end of round 3 (git tag of that tree: commit f371326).
//   diag64_kernel     round 1: 64 x 64 diagonal block, 256 threads, 4x4 register blocks, four pivots per barrier   (26.5 us per launch)
//   diag64v2_kernel   round 2: the same with owner-only pivot-block work                                          (40.5 us: slower)
//   gemm44_kernel<128, false[, FLAGS]>  round 1 wide trailing update on v_mfma_f64_4x4x4_4b_f64 (62.7 TFLOP/s); FLAGS: LDS-flag synchronised
//                     main loop instead of s_barrier (3-6 % slower).  The 64-column instantiations of gemm44_kernel are still product code.
//   gemm16_kernel     round 2: v_mfma_f64_16x16x4 dropped into the round-1 loop structure (64.0 TFLOP/s: no gain without the pipeline)
// Product replacements: diag64m_kernel (15.3 us), gemm16p_kernel / potrf_node_kernel (72-73 TFLOP/s).

// ===== diag64_kernel, diag64v2_kernel =====
// ---------------------------------------------------------------------------------------------------
// K2a: 64x64 diagonal block: Cholesky factor L and its inverse W = L^-1 in one symmetric Gaussian elimination of [A | I]
// held in registers: after eliminating the columns,  [A | I] -> [D L1' | L1^-1];  L = L1 D^1/2,  W = D^-1/2 L1^-1.
// One workgroup per matrix of the batch, latency bound.
// ---------------------------------------------------------------------------------------------------
// Rank-4, 4x4-register-block form: four pivots per barrier, and thread (a, b) owns the 4x4 blocks
// (rows 4a..4a+3, columns 4b..4b+3) of the S part and of the W part, so a step needs only 4 + 4 + 4 + 4 four-double LDS rows
// per thread (its rows of the pivot columns, the pivot block, its columns of the pivot rows of both parts) instead of the
// 64 + 64 values of a row-per-thread layout.  The owners of block column J/4 publish the four current columns J..J+3 (raw,
// and masked to the rows below the block: by symmetry the pivot rows); the owners of block row J/4 publish rows J..J+3 of the
// W part; every thread factors the 4x4 pivot block P = Lp Lp' redundantly, forms y = B Lp^-T for its four rows (the final L
// entries in these columns) and the multipliers m = y Lp^-1, and applies one rank-4 update to its 16 + 16 register values.
// Rows inside the block finish their W rows as Lp^-1 Wtop.  s holds L sqrt(d), w holds sqrt(d) L^-1, dd the pivots.
template <typename TS>
__global__ __launch_bounds__(256) void diag64_kernel(BatchPtr Ab, size_t offA, int ld, BatchPtr Wb, size_t offW,
                                                     int gcol0, int n_real, BatchInfo infob) {
  void* __restrict__ A = Ab.p[blockIdx.x];        // element offsets offA / offW are applied in units of TS
  void* __restrict__ W = Wb.p[blockIdx.x];
  int* __restrict__ info = infob.p[blockIdx.x];
  __shared__ __attribute__((aligned(16))) double cb[2][4][64];     // raw columns J..J+3 (all rows)
  __shared__ __attribute__((aligned(16))) double cm[2][4][64];     // the same, zero for rows <= J+3
  __shared__ __attribute__((aligned(16))) double rw[2][4][64];     // rows J..J+3 of the W part
  __shared__ double dd[64];
  const int t = threadIdx.x, a = t & 15, b = t >> 4;
  double s[4][4], w[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = 4 * b + c;
    double col[4] = {0.0, 0.0, 0.0, 0.0};
    if (a >= b) MatIO<TS>::ld4(A, offA + (size_t)k * ld + 4 * a, col);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[r][c] = (4 * a + r >= k) ? col[r] : 0.0;
      w[r][c] = (4 * a + r == k) ? 1.0 : 0.0;
    }
  }
  auto rsq = [](double x) {          // 1/sqrt(x): v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
  };
  for (int st = 0; st < 16; ++st) {
    const int J = 4 * st, bf = st & 1;
    if (b == st) {                       // owners of columns J..J+3: column r' holds s[.][r'] for rows 4a..4a+3
#pragma unroll
      for (int rp = 0; rp < 4; ++rp) {
        d2* o = reinterpret_cast<d2*>(&cb[bf][rp][4 * a]);
        d2* om = reinterpret_cast<d2*>(&cm[bf][rp][4 * a]);
        const d2 lo = mk2(s[0][rp], s[1][rp]), hi = mk2(s[2][rp], s[3][rp]);
        const d2 z = mk2(0.0, 0.0);
        o[0] = lo; o[1] = hi;
        om[0] = (a > st) ? lo : z; om[1] = (a > st) ? hi : z;
      }
    }
    if (a == st) {                       // owners of rows J..J+3 of the W part
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2* o = reinterpret_cast<d2*>(&rw[bf][r][4 * b]);
        o[0] = mk2(w[r][0], w[r][1]); o[1] = mk2(w[r][2], w[r][3]);
      }
    }
    __syncthreads();
    double Bv[4][4], Pv[4][4], Rv[4][4], Wv[4][4];     // [column or pivot r'][row / col within my block]
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
      const d2* pb = reinterpret_cast<const d2*>(&cb[bf][rp][4 * a]);
      const d2* pp = reinterpret_cast<const d2*>(&cb[bf][rp][J]);
      const d2* pr = reinterpret_cast<const d2*>(&cm[bf][rp][4 * b]);
      const d2* pw = reinterpret_cast<const d2*>(&rw[bf][rp][4 * b]);
      const d2 b0 = pb[0], b1 = pb[1], p0 = pp[0], p1 = pp[1], r0 = pr[0], r1 = pr[1], w0 = pw[0], w1 = pw[1];
      Bv[rp][0] = b0.x; Bv[rp][1] = b0.y; Bv[rp][2] = b1.x; Bv[rp][3] = b1.y;
      Pv[rp][0] = p0.x; Pv[rp][1] = p0.y; Pv[rp][2] = p1.x; Pv[rp][3] = p1.y;
      Rv[rp][0] = r0.x; Rv[rp][1] = r0.y; Rv[rp][2] = r1.x; Rv[rp][3] = r1.y;
      Wv[rp][0] = w0.x; Wv[rp][1] = w0.y; Wv[rp][2] = w1.x; Wv[rp][3] = w1.y;
    }
    // 4x4 Cholesky of the pivot block (Pv[c][r] = P[r][c], lower part used)
    const double d0 = Pv[0][0], r0 = rsq(d0);
    const double l10 = Pv[0][1] * r0, l20 = Pv[0][2] * r0, l30 = Pv[0][3] * r0;
    const double d1 = __builtin_fma(-l10, l10, Pv[1][1]), r1 = rsq(d1);
    const double l21 = __builtin_fma(-l20, l10, Pv[1][2]) * r1, l31 = __builtin_fma(-l30, l10, Pv[1][3]) * r1;
    const double d2v = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, Pv[2][2])), r2 = rsq(d2v);
    const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, Pv[2][3])) * r2;
    const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, Pv[3][3]))), r3 = rsq(d3);
    const double l00 = d0 * r0, l11 = d1 * r1, l22 = d2v * r2, l33 = d3 * r3;
    if (t == 0) { dd[J] = d0; dd[J + 1] = d1; dd[J + 2] = d2v; dd[J + 3] = d3; }
    const bool below = (a > st), inblk = (a == st);
    // coefficients of the W-part update of my four rows: below the block -m; inside it l_rr * (Lp^-1)[r][.]; above: none
    double cf[4][4], keep = 1.0;
    double y[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      // y = B_row Lp^-T (forward), m = y Lp^-1 (backward)
      y[r][0] = Bv[0][r] * r0;
      y[r][1] = __builtin_fma(-y[r][0], l10, Bv[1][r]) * r1;
      y[r][2] = __builtin_fma(-y[r][1], l21, __builtin_fma(-y[r][0], l20, Bv[2][r])) * r2;
      y[r][3] = __builtin_fma(-y[r][2], l32, __builtin_fma(-y[r][1], l31, __builtin_fma(-y[r][0], l30, Bv[3][r]))) * r3;
      const double m3 = y[r][3] * r3;
      const double m2 = __builtin_fma(-m3, l32, y[r][2]) * r2;
      const double m1 = __builtin_fma(-m3, l31, __builtin_fma(-m2, l21, y[r][1])) * r1;
      const double m0 = __builtin_fma(-m3, l30, __builtin_fma(-m2, l20, __builtin_fma(-m1, l10, y[r][0]))) * r0;
      cf[r][0] = below ? -m0 : 0.0; cf[r][1] = below ? -m1 : 0.0; cf[r][2] = below ? -m2 : 0.0; cf[r][3] = below ? -m3 : 0.0;
    }
    // S part: only rows below the block change (cf = -m there, 0 elsewhere)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        s[r][c] = __builtin_fma(cf[r][3], Rv[3][c], __builtin_fma(cf[r][2], Rv[2][c], __builtin_fma(cf[r][1], Rv[1][c],
                  __builtin_fma(cf[r][0], Rv[0][c], s[r][c]))));
    if (inblk) {                         // rows J..J+3: w <- l_rr * (Lp^-1 Wtop)[r]
      keep = 0.0;
      const double i10 = -l10 * r0 * r1;
      const double i20 = -(l20 * r0 + l21 * i10) * r2, i21 = -l21 * r1 * r2;
      const double i30 = -(l30 * r0 + l31 * i10 + l32 * i20) * r3, i31 = -(l31 * r1 + l32 * i21) * r3, i32 = -l32 * r2 * r3;
      cf[0][0] = l00 * r0; cf[0][1] = 0.0; cf[0][2] = 0.0; cf[0][3] = 0.0;
      cf[1][0] = l11 * i10; cf[1][1] = l11 * r1; cf[1][2] = 0.0; cf[1][3] = 0.0;
      cf[2][0] = l22 * i20; cf[2][1] = l22 * i21; cf[2][2] = l22 * r2; cf[2][3] = 0.0;
      cf[3][0] = l33 * i30; cf[3][1] = l33 * i31; cf[3][2] = l33 * i32; cf[3][3] = l33 * r3;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        w[r][c] = __builtin_fma(cf[r][3], Wv[3][c], __builtin_fma(cf[r][2], Wv[2][c], __builtin_fma(cf[r][1], Wv[1][c],
                  __builtin_fma(cf[r][0], Wv[0][c], keep * w[r][c]))));
    if (b == st) {                       // my rows' final entries in columns J..J+3, in the s = L sqrt(d) convention
#pragma unroll
      for (int r = 0; r < 4; ++r) { s[r][0] = y[r][0] * l00; s[r][1] = y[r][1] * l11; s[r][2] = y[r][2] * l22; s[r][3] = y[r][3] * l33; }
    }
  }
  __syncthreads();
  if (t < 64) {                                    // LAPACK-style info: first non-positive (or NaN) pivot, 1-based
    const bool bad = !(dd[t] > 0.0) && (gcol0 + t < n_real);
    const unsigned long long mask = __ballot(bad);
    if (t == 0 && mask != 0ull) atomicCAS(info, 0, gcol0 + __builtin_ctzll(mask) + 1);
  }
  __syncthreads();
  double rsr[4];                                   // row scales of W = D^-1/2 L1^-1
#pragma unroll
  for (int r = 0; r < 4; ++r) rsr[r] = rsq(dd[4 * a + r]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = 4 * b + c;
    const double dk = dd[k], lki = rsq(dk);
    double ao[4], wo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * a + r;
      ao[r] = (i == k) ? dk * lki : s[r][c] * lki;
      wo[r] = (i >= k) ? w[r][c] * rsr[r] : 0.0;
    }
    if (a > b) {
      MatIO<TS>::st4(A, offA + (size_t)k * ld + 4 * a, ao);
    } else if (a == b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r >= c) MatIO<TS>::st1(A, offA + (size_t)k * ld + 4 * a + r, ao[r]);
    }
    MatIO<TS>::st4(W, offW + (size_t)k * 64 + 4 * a, wo);
  }
}

// ---------------------------------------------------------------------------------------------------
// K2a, second form (round 2): the same rank-4 / 4x4-register-block elimination, but the pivot-block work is done ONCE per step by
// the 16 threads that own the current block column instead of redundantly by all 256:
//   phase A (owners of block column st: one quarter-wave): pivot block through wave shuffles, 4x4 Cholesky Lp and Lp^-1, then for
//           the thread's four rows y = B Lp^-T and the multipliers m = y Lp^-1 as short dot products (no substitution chains);
//           publish m (rows below the block), the raw pivot columns (= pivot rows by symmetry), Lp^-1 scaled for the W rows, pivots;
//   barrier;
//   phase B (all threads): 3 x 4 sixteen-byte LDS rows, one rank-4 update of the 16 + 16 registers.
// A step costs ~1.8k clocks instead of ~3.5k (every thread used to run the ~600-clock 4x4 Cholesky chain and ~470 clocks of
// forward/backward substitution for its rows, and to read 512 instead of 384 bytes of pivot data).
// ---------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void diag64v2_kernel(BatchPtr Ab, size_t offA, int ld, BatchPtr Wb, size_t offW,
                                                       int gcol0, int n_real, BatchInfo infob) {
  void* __restrict__ A = Ab.p[blockIdx.x];
  void* __restrict__ W = Wb.p[blockIdx.x];
  int* __restrict__ info = infob.p[blockIdx.x];
  __shared__ __attribute__((aligned(16))) double mm[2][4][64];     // multipliers m[k][row] (0 for rows not below the block)
  __shared__ __attribute__((aligned(16))) double cm[2][4][64];     // raw pivot columns k, rows below the block (0 elsewhere)
  __shared__ __attribute__((aligned(16))) double rw[2][4][64];     // rows J..J+3 of the W part
  __shared__ __attribute__((aligned(16))) double li[2][16];        // l_rr (Lp^-1)[r][k]: the W rows inside the block
  __shared__ double dd[64];
  const int t = threadIdx.x, a = t & 15, b = t >> 4, wv = t >> 6;
  double s[4][4], w[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = 4 * b + c;
    double col[4] = {0.0, 0.0, 0.0, 0.0};
    if (a >= b) MatIO<TS>::ld4(A, offA + (size_t)k * ld + 4 * a, col);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s[r][c] = (4 * a + r >= k) ? col[r] : 0.0;
      w[r][c] = (4 * a + r == k) ? 1.0 : 0.0;
    }
  }
  auto rsq = [](double x) {          // 1/sqrt(x): v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
  };
  for (int st = 0; st < 16; ++st) {
    const int J = 4 * st, bf = st & 1;
    if (wv == (st >> 2)) {               // the wave holding block column st (wave-uniform branch: shuffles are safe)
      const int src = (st & 3) * 16 + st;            // lane of thread (a = st, b = st) inside this wave
      // pivot block P (lower part), broadcast from its owner
      const double p00 = __shfl(s[0][0], src), p10 = __shfl(s[1][0], src), p20 = __shfl(s[2][0], src), p30 = __shfl(s[3][0], src);
      const double p11 = __shfl(s[1][1], src), p21 = __shfl(s[2][1], src), p31 = __shfl(s[3][1], src);
      const double p22 = __shfl(s[2][2], src), p32 = __shfl(s[3][2], src), p33 = __shfl(s[3][3], src);
      if (b == st) {
        // 4x4 Cholesky P = Lp Lp'
        const double d0 = p00, r0 = rsq(d0);
        const double l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
        const double d1 = __builtin_fma(-l10, l10, p11), r1 = rsq(d1);
        const double l21 = __builtin_fma(-l20, l10, p21) * r1, l31 = __builtin_fma(-l30, l10, p31) * r1;
        const double d2v = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, p22)), r2 = rsq(d2v);
        const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, p32)) * r2;
        const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, p33))), r3 = rsq(d3);
        const double l00 = d0 * r0, l11 = d1 * r1, l22 = d2v * r2, l33 = d3 * r3;
        // Lp^-1 (lower): diagonal r0..r3
        const double i10 = -l10 * r0 * r1;
        const double i20 = -(l20 * r0 + l21 * i10) * r2, i21 = -l21 * r1 * r2;
        const double i30 = -(l30 * r0 + l31 * i10 + l32 * i20) * r3, i31 = -(l31 * r1 + l32 * i21) * r3, i32 = -l32 * r2 * r3;
        const bool below = (a > st);
        double mk[4][4], ck[4][4];       // [k][row]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double b0 = s[r][0], b1 = s[r][1], b2 = s[r][2], b3 = s[r][3];
          // y = B Lp^-T: y_k = sum_{j <= k} B_j (Lp^-1)[k][j]
          const double y0 = b0 * r0;
          const double y1 = __builtin_fma(b1, r1, b0 * i10);
          const double y2 = __builtin_fma(b2, r2, __builtin_fma(b1, i21, b0 * i20));
          const double y3 = __builtin_fma(b3, r3, __builtin_fma(b2, i32, __builtin_fma(b1, i31, b0 * i30)));
          // m = y Lp^-1: m_k = sum_{j >= k} y_j (Lp^-1)[j][k]
          const double m3 = y3 * r3;
          const double m2 = __builtin_fma(y2, r2, y3 * i32);
          const double m1 = __builtin_fma(y1, r1, __builtin_fma(y2, i21, y3 * i31));
          const double m0 = __builtin_fma(y0, r0, __builtin_fma(y1, i10, __builtin_fma(y2, i20, y3 * i30)));
          mk[0][r] = below ? m0 : 0.0; mk[1][r] = below ? m1 : 0.0; mk[2][r] = below ? m2 : 0.0; mk[3][r] = below ? m3 : 0.0;
          ck[0][r] = below ? b0 : 0.0; ck[1][r] = below ? b1 : 0.0; ck[2][r] = below ? b2 : 0.0; ck[3][r] = below ? b3 : 0.0;
          // my rows' final entries in columns J..J+3, in the s = L sqrt(d) convention (rows inside the block: Lp itself)
          if (a >= st) { s[r][0] = y0 * l00; s[r][1] = y1 * l11; s[r][2] = y2 * l22; s[r][3] = y3 * l33; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          d2* om = reinterpret_cast<d2*>(&mm[bf][k][4 * a]);
          d2* oc = reinterpret_cast<d2*>(&cm[bf][k][4 * a]);
          om[0] = mk2(mk[k][0], mk[k][1]); om[1] = mk2(mk[k][2], mk[k][3]);
          oc[0] = mk2(ck[k][0], ck[k][1]); oc[1] = mk2(ck[k][2], ck[k][3]);
        }
        if (a == st) {
          dd[J] = d0; dd[J + 1] = d1; dd[J + 2] = d2v; dd[J + 3] = d3;
          double* q = &li[bf][0];          // row r of l_rr Lp^-1
          q[0] = l00 * r0;  q[1] = 0.0;        q[2] = 0.0;        q[3] = 0.0;
          q[4] = l11 * i10; q[5] = l11 * r1;   q[6] = 0.0;        q[7] = 0.0;
          q[8] = l22 * i20; q[9] = l22 * i21;  q[10] = l22 * r2;  q[11] = 0.0;
          q[12] = l33 * i30; q[13] = l33 * i31; q[14] = l33 * i32; q[15] = l33 * r3;
        }
      }
    }
    if (a == st) {                       // owners of rows J..J+3 of the W part
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        d2* o = reinterpret_cast<d2*>(&rw[bf][r][4 * b]);
        o[0] = mk2(w[r][0], w[r][1]); o[1] = mk2(w[r][2], w[r][3]);
      }
    }
    __syncthreads();
    double Mv[4][4], Rv[4][4], Wv[4][4];     // [pivot k][row r of mine] / [pivot k][col c of mine]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const d2* pm = reinterpret_cast<const d2*>(&mm[bf][k][4 * a]);
      const d2* pr = reinterpret_cast<const d2*>(&cm[bf][k][4 * b]);
      const d2* pw = reinterpret_cast<const d2*>(&rw[bf][k][4 * b]);
      const d2 m0 = pm[0], m1 = pm[1], r0 = pr[0], r1 = pr[1], w0 = pw[0], w1 = pw[1];
      Mv[k][0] = m0.x; Mv[k][1] = m0.y; Mv[k][2] = m1.x; Mv[k][3] = m1.y;
      Rv[k][0] = r0.x; Rv[k][1] = r0.y; Rv[k][2] = r1.x; Rv[k][3] = r1.y;
      Wv[k][0] = w0.x; Wv[k][1] = w0.y; Wv[k][2] = w1.x; Wv[k][3] = w1.y;
    }
    // S part: rows below the block, columns right of it (Mv / Rv are zero elsewhere)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        s[r][c] = __builtin_fma(-Mv[3][r], Rv[3][c], __builtin_fma(-Mv[2][r], Rv[2][c], __builtin_fma(-Mv[1][r], Rv[1][c],
                  __builtin_fma(-Mv[0][r], Rv[0][c], s[r][c]))));
    if (a == st) {                       // rows J..J+3: w <- l_rr (Lp^-1 Wtop)[r]
      const double* q = &li[bf][0];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          w[r][c] = __builtin_fma(q[4 * r + 3], Wv[3][c], __builtin_fma(q[4 * r + 2], Wv[2][c], __builtin_fma(q[4 * r + 1], Wv[1][c],
                    q[4 * r] * Wv[0][c])));
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          w[r][c] = __builtin_fma(-Mv[3][r], Wv[3][c], __builtin_fma(-Mv[2][r], Wv[2][c], __builtin_fma(-Mv[1][r], Wv[1][c],
                    __builtin_fma(-Mv[0][r], Wv[0][c], w[r][c]))));
    }
  }
  __syncthreads();
  if (t < 64) {                                    // LAPACK-style info: first non-positive (or NaN) pivot, 1-based
    const bool bad = !(dd[t] > 0.0) && (gcol0 + t < n_real);
    const unsigned long long mask = __ballot(bad);
    if (t == 0 && mask != 0ull) atomicCAS(info, 0, gcol0 + __builtin_ctzll(mask) + 1);
  }
  __syncthreads();
  double rsr[4];                                   // row scales of W = D^-1/2 L1^-1
#pragma unroll
  for (int r = 0; r < 4; ++r) rsr[r] = rsq(dd[4 * a + r]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = 4 * b + c;
    const double dk = dd[k], lki = rsq(dk);
    double ao[4], wo[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * a + r;
      ao[r] = (i == k) ? dk * lki : s[r][c] * lki;
      wo[r] = (i >= k) ? w[r][c] * rsr[r] : 0.0;
    }
    if (a > b) {
      MatIO<TS>::st4(A, offA + (size_t)k * ld + 4 * a, ao);
    } else if (a == b) {
#pragma unroll
      for (int r = 0; r < 4; ++r) if (r >= c) MatIO<TS>::st1(A, offA + (size_t)k * ld + 4 * a + r, ao[r]);
    }
    MatIO<TS>::st4(W, offW + (size_t)k * 64 + 4 * a, wo);
  }
}


// ===== gemm44_kernel with the FLAGS variant and the timing-ablation macros (as of round 3) =====
// ---- LDS-flag synchronisation (FLAGS variant of gemm44_kernel) ----------------------------------------------------------
// s_barrier makes the four waves of a workgroup meet once per k-stage, so every stage pays the arrival skew of waves whose
// SIMD partners (the other resident workgroup) progress unevenly.  The FLAGS main loop replaces it by two monotonic LDS
// counters: "published" (a wave has written its share of the NEXT stage's operands) and "retired" (a wave has finished
// reading the CURRENT stage).  A wave publishes in the middle of a stage and polls the counters half a stage later, so up to
// half a stage of skew between the waves costs nothing.  LDS operations of one wave execute in issue order (they return in
// order: lgkmcnt), so a ds_add issued after the wave's ds_writes / ds_reads is performed after them.
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_signal(unsigned off) {
  asm volatile("ds_add_u32 %0, %1" ::"v"(off), "v"(1u) : "memory");
}
__device__ __forceinline__ void lds_wait_ge(unsigned off, unsigned target) {
  for (;;) {
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(off) : "memory");
    if ((int)(__builtin_amdgcn_readfirstlane(v) - target) >= 0) break;
    __builtin_amdgcn_s_sleep(1);
  }
}

template <int BN, bool SET, bool FLAGS = false>
__global__ __launch_bounds__(256, 2) void gemm44_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
                                                         BatchPtr Bb, size_t goffB, int ldb,
                                                         int M, int N, int K, int lower, int MT, int full_items,
                                                         int splitk, int kfrom_row) {
#ifdef LMM_CLOCK_PROBE
  const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  double* C = Cb.p[blockIdx.y] + goffC;
  const double* A = Ab.p[blockIdx.y] + goffA;
  const double* B = Bb.p[blockIdx.y] + goffB;
  constexpr int BM = 128, BK = 16;
  constexpr int WN = BN / 2;
  constexpr int TM = 4, TN = WN / 16;
  constexpr int SA = BM + 16, SB = BN + 16;
  constexpr int NLA = (BM * BK / 2) / 256;     // 4: thread t stages rows 2(t%64).. of k-columns t/64 + 4q
  constexpr int NLB = (BN * BK / 2) / 256;     // 4 (BN=128) or 2 (BN=64)
  constexpr int KSB = 256 / (BN / 2);          // k-columns covered per pass of the B staging (4 or 8)
  constexpr int STAGE = 2 * BK * SA + 2 * BK * SB, EPI = 4 * 32 * 65;
  __shared__ __attribute__((aligned(16))) double smem[STAGE > EPI ? STAGE : EPI];   // staging, then epilogue transpose
  double (*As)[BK * SA] = reinterpret_cast<double (*)[BK * SA]>(smem);
  double (*Bs)[BK * SB] = reinterpret_cast<double (*)[BK * SB]>(smem + 2 * BK * SA);
  __shared__ unsigned sync_cnt[FLAGS ? 2 : 1];          // FLAGS: [0] stages published, [1] stages retired (x 4 waves)
  if (FLAGS && threadIdx.x == 0) { sync_cnt[0] = 0; sync_cnt[1] = 0; }        // visible after the prologue barrier

  int part = 0, nparts = 1, tj = 0, ti = 0;
  gemm_work_item(BM, BN, N, lower, MT, full_items, splitk, part, nparts, ti, tj);
  const int bm = ti * BM, bn = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * WN;
  const bool active = (bm + wr < M) && (bn + wc < N) && !(lower && bm + wr + 63 < bn + wc);
  const int nk_all = K / BK;
  int kc0 = (int)((long long)nk_all * part / nparts);
  const int kc1 = (int)((long long)nk_all * (part + 1) / nparts);
  // kfrom_row: the operands are upper triangular (X[i,k] = 0 for k < i), so the product over k starts at the tile's
  // first row (LAUUM-like X X' for the inverse from its Cholesky factor)
  if (kfrom_row && kc0 < bm / BK) kc0 = bm / BK;
  A += (size_t)kc0 * BK * lda;
  B += (size_t)kc0 * BK * ldb;

  // staging addresses: one base pointer per operand; the NLA / NLB passes differ by a uniform k offset
  int rowa = bm + 2 * (t % (BM / 2)); if (rowa > M - 2) rowa = M - 2;
  int rowb = bn + 2 * (t % (BN / 2)); if (rowb > N - 2) rowb = N - 2;
  const double* ga0 = A + (size_t)(t / (BM / 2)) * lda + rowa;
  const double* gb0 = B + (size_t)(t / (BN / 2)) * ldb + rowb;
  const int sa0 = (t / (BM / 2)) * SA + 2 * (t % (BM / 2));
  const int sb0 = (t / (BN / 2)) * SB + 2 * (t % (BN / 2));
  d2 ra[NLA], rb[NLB];
#pragma unroll
  for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(4 * q) * lda);
#pragma unroll
  for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(KSB * q) * ldb);
#pragma unroll
  for (int q = 0; q < NLA; ++q) *reinterpret_cast<d2*>(&As[0][sa0 + 4 * q * SA]) = ra[q];
#pragma unroll
  for (int q = 0; q < NLB; ++q) *reinterpret_cast<d2*>(&Bs[0][sb0 + KSB * q * SB]) = rb[q];
  __syncthreads();

  double acc[TM][TN][4];
#pragma unroll
  for (int u = 0; u < TM; ++u)
#pragma unroll
    for (int v = 0; v < TN; ++v)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[u][v][s] = 0.0;

  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + wr + l15;
  int offB[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) offB[s] = lk * SB + wc + ((l15 + 4 * s) & 15);

  const int nk = kc1 - kc0;
  if constexpr (FLAGS) {
    const unsigned off_pub = lds_off(&sync_cnt[0]), off_ret = lds_off(&sync_cnt[1]);
    if (nk > 1) {                                  // stage 1 into registers (stage 0 is in LDS, published by the prologue barrier)
      const double* pa = ga0 + (size_t)BK * lda;
      const double* pb = gb0 + (size_t)BK * ldb;
#pragma unroll
      for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(4 * q) * lda);
#pragma unroll
      for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(KSB * q) * ldb);
    }
    if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      if (kt > 0) lds_wait_ge(off_pub, 4u * kt);   // every wave has published its share of stage kt (written during stage kt-1)
      const double* as = &As[buf][0];
      const double* bs = &Bs[buf][0];
#pragma unroll
      for (int s4 = 0; s4 < BK / 4; ++s4) {
        if (active) {
          double fa[TM];
#pragma unroll
          for (int u = 0; u < TM; ++u) fa[u] = as[offA + 4 * s4 * SA + 16 * u];
#pragma unroll
          for (int v = 0; v < TN; ++v) {
            double fb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) fb[s] = bs[offB[s] + 4 * s4 * SB + 16 * v];
#pragma unroll
            for (int u = 0; u < TM; ++u)
#pragma unroll
              for (int s = 0; s < 4; ++s)
                acc[u][v][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[u], fb[s], acc[u][v][s], 0, 0, 0);
          }
        }
        if (s4 == 1 && kt + 1 < nk) {
          // mid-stage: the other buffer was read during stage kt-1 -- wait until all four waves retired it, then write stage
          // kt+1 (in registers since the middle of stage kt-1), publish, and start the loads of stage kt+2
          lds_wait_ge(off_ret, 4u * kt);
#pragma unroll
          for (int q = 0; q < NLA; ++q) *reinterpret_cast<d2*>(&As[buf ^ 1][sa0 + 4 * q * SA]) = ra[q];
#pragma unroll
          for (int q = 0; q < NLB; ++q) *reinterpret_cast<d2*>(&Bs[buf ^ 1][sb0 + KSB * q * SB]) = rb[q];
          if (lane == 0) lds_signal(off_pub);
          if (kt + 2 < nk) {
            const double* pa = ga0 + (size_t)(kt + 2) * BK * lda;
            const double* pb = gb0 + (size_t)(kt + 2) * BK * ldb;
#pragma unroll
            for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(4 * q) * lda);
#pragma unroll
            for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(KSB * q) * ldb);
          }
        }
      }
      if (lane == 0) lds_signal(off_ret);          // this wave's reads of stage kt are issued (LDS runs a wave's ops in order)
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();                               // the epilogue reuses the staging memory as transpose scratch
  } else
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
#ifndef LMM_ABLATE_NOLOAD
    if (kt + 1 < nk) {
#ifdef LMM_ABLATE_L2HOT
      const int ktl = (kt + 1) & 3;      // ablation: operand loads always hit the same 4 k-stages (L2-resident)
#else
      const int ktl = kt + 1;
#endif
      const double* pa = ga0 + (size_t)ktl * BK * lda;
      const double* pb = gb0 + (size_t)ktl * BK * ldb;
#pragma unroll
      for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(4 * q) * lda);
#pragma unroll
      for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(KSB * q) * ldb);
    }
#endif
#ifndef LMM_ABLATE_NOMFMA
    if (active) {
      // two workgroups share each SIMD: the one in its MFMA phase issues first (+2 %); odd work items one level higher, so
      // that two co-resident workgroups do not trade the pipe instruction by instruction (+0.8 %)
      if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
      const double* as = &As[buf][0];
      const double* bs = &Bs[buf][0];
#pragma unroll
      for (int s4 = 0; s4 < BK / 4; ++s4) {
        double fa[TM];
#pragma unroll
#ifdef LMM_ABLATE_NOLDSREAD
        for (int u = 0; u < TM; ++u) fa[u] = __builtin_amdgcn_readfirstlane(kt) * 1e-9 + u + s4;   // operands from registers (ablation)
#else
        for (int u = 0; u < TM; ++u) fa[u] = as[offA + 4 * s4 * SA + 16 * u];
#endif
#pragma unroll
        for (int v = 0; v < TN; ++v) {
          double fb[4];
#pragma unroll
#ifdef LMM_ABLATE_NOLDSREAD
          for (int s = 0; s < 4; ++s) fb[s] = lane * 1e-9 + s + v;
#else
          for (int s = 0; s < 4; ++s) fb[s] = bs[offB[s] + 4 * s4 * SB + 16 * v];
#endif
#pragma unroll
          for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[u][v][s] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[u], fb[s], acc[u][v][s], 0, 0, 0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
    }
#endif
    if (kt + 1 < nk) {
#pragma unroll
      for (int q = 0; q < NLA; ++q) *reinterpret_cast<d2*>(&As[buf ^ 1][sa0 + 4 * q * SA]) = ra[q];
#pragma unroll
      for (int q = 0; q < NLB; ++q) *reinterpret_cast<d2*>(&Bs[buf ^ 1][sb0 + KSB * q * SB]) = rb[q];
    }
#ifndef LMM_ABLATE_NOBARRIER
    __syncthreads();
#endif
  }

  if (!active) return;
  // epilogue: lane (i = lane>>4, blk = (lane>>2)&3, j = lane&3) holds C[row 16u + 4 blk + i, col 16v + 4((blk+s)&3) + j].
  // Transposed through a wave-private LDS region (32 columns x 64 rows at a time) so that every global access of the
  // read-modify-write (or f64 atomic) is one contiguous 512-byte row segment per wave instruction.
  constexpr int ES = 65;
  double* ep = smem + w * (32 * ES);
  const int li = lane >> 4, lb = (lane >> 2) & 3, lj = lane & 3;
#pragma unroll
  for (int h = 0; h < TN / 2; ++h) {
    double* cp = C + (size_t)(bn + wc + 32 * h) * ldc + bm + wr + lane;
    // all 32 C loads of this half are issued before the LDS transpose (the accumulators they replace are dead by
    // then), so the read-modify-write pays ONE memory round trip per half instead of one per few columns
    double cv[32];
    if (!SET && nparts == 1) {
#pragma unroll
      for (int c = 0; c < 32; ++c) cv[c] = cp[(size_t)c * ldc];
    }
#pragma unroll
    for (int u = 0; u < TM; ++u)
#pragma unroll
      for (int vv = 0; vv < 2; ++vv)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          ep[(16 * vv + 4 * ((lb + s) & 3) + lj) * ES + 16 * u + 4 * lb + li] = acc[u][2 * h + vv][s];
    __builtin_amdgcn_wave_barrier();
    if (SET) {
#pragma unroll
      for (int c = 0; c < 32; ++c) cp[(size_t)c * ldc] = ep[c * ES + lane];
    } else if (nparts == 1) {
#pragma unroll
      for (int c = 0; c < 32; ++c) cp[(size_t)c * ldc] = cv[c] - ep[c * ES + lane];
    } else {
#pragma unroll
      for (int c = 0; c < 32; ++c) unsafeAtomicAdd(cp + (size_t)c * ldc, -ep[c * ES + lane]);
    }
    __builtin_amdgcn_wave_barrier();
  }
#ifdef LMM_CLOCK_PROBE
  if (blockIdx.x == 300 && threadIdx.x == 0) {
    g_clk_probe[0] = __builtin_readcyclecounter() - clk0; g_clk_probe[1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

// ===== gemm16_kernel =====
// ---------------------------------------------------------------------------------------------------
// K2b, second form (round 2): the same update on v_mfma_f64_16x16x4_f64 with the accumulators in ARCHITECTURAL VGPRs.
// Round 1 measured this instruction at 36-59 TFLOP/s and chose the 4x4x4 form; tools/mfma_probe4 shows why: with AccVGPR
// accumulators (what hipcc allocates by default once a kernel holds many of them) v_mfma_f64_16x16x4 issues at 36 TFLOP/s,
// with VGPR accumulators (-mllvm -amdgpu-mfma-vgpr-form=1, the form rocBLAS' gfx950 dgemm kernels use) at 77.7 TFLOP/s
// = 98.9 % of the FP64 peak, whatever the operand order.  Per k-step of 4 a wave's 64 x 64 tile then needs 4 + 4 ds_read_b64
// feeding 16 MFMAs of 64 cycles (the 4x4x4 form: 4 + 16 reads feeding 64 MFMAs of 16 cycles) -- a fifth of the instruction
// stream.  Same block tile, staging and LDS image as gemm44_kernel (the A fragment of the 4x4x4 form IS the 16x16x4 operand
// read; B needs no rotations).  As in the fp32 kernel the MFMA's A operand is fed from the B matrix and its B operand from
// the A matrix, so D's lane index runs along the rows of C:
//     acc[v][u][r] (lane l)  <->  C[bm + wr + 16 u + (l & 15),  bn + wc + 16 v + (l >> 4) + 4 r]
// and every global access of the epilogue is four contiguous 128-byte row segments.
// ---------------------------------------------------------------------------------------------------
template <int BN, bool SET>
__global__ __launch_bounds__(256, 2) void gemm16_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
                                                         BatchPtr Bb, size_t goffB, int ldb,
                                                         int M, int N, int K, int lower, int MT, int full_items,
                                                         int splitk, int kfrom_row) {
  double* C = Cb.p[blockIdx.y] + goffC;
  const double* A = Ab.p[blockIdx.y] + goffA;
  const double* B = Bb.p[blockIdx.y] + goffB;
  constexpr int BM = 128, BK = 16;
  constexpr int WN = BN / 2;
  constexpr int TM = 4, TN = WN / 16;
  constexpr int SA = BM + 16, SB = BN + 16;
  constexpr int NLA = (BM * BK / 2) / 256;
  constexpr int NLB = (BN * BK / 2) / 256;
  constexpr int KSB = 256 / (BN / 2);
  __shared__ __attribute__((aligned(16))) double As[2][BK * SA];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * SB];

  int part = 0, nparts = 1, tj = 0, ti = 0;
  gemm_work_item(BM, BN, N, lower, MT, full_items, splitk, part, nparts, ti, tj);
  const int bm = ti * BM, bn = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * WN;
  const bool active = (bm + wr < M) && (bn + wc < N) && !(lower && bm + wr + 63 < bn + wc);
  const int nk_all = K / BK;
  int kc0 = (int)((long long)nk_all * part / nparts);
  const int kc1 = (int)((long long)nk_all * (part + 1) / nparts);
  if (kfrom_row && kc0 < bm / BK) kc0 = bm / BK;
  A += (size_t)kc0 * BK * lda;
  B += (size_t)kc0 * BK * ldb;

  int rowa = bm + 2 * (t % (BM / 2)); if (rowa > M - 2) rowa = M - 2;
  int rowb = bn + 2 * (t % (BN / 2)); if (rowb > N - 2) rowb = N - 2;
  const double* ga0 = A + (size_t)(t / (BM / 2)) * lda + rowa;
  const double* gb0 = B + (size_t)(t / (BN / 2)) * ldb + rowb;
  const int sa0 = (t / (BM / 2)) * SA + 2 * (t % (BM / 2));
  const int sb0 = (t / (BN / 2)) * SB + 2 * (t % (BN / 2));
  d2 ra[NLA], rb[NLB];
#pragma unroll
  for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(4 * q) * lda);
#pragma unroll
  for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(KSB * q) * ldb);
#pragma unroll
  for (int q = 0; q < NLA; ++q) *reinterpret_cast<d2*>(&As[0][sa0 + 4 * q * SA]) = ra[q];
#pragma unroll
  for (int q = 0; q < NLB; ++q) *reinterpret_cast<d2*>(&Bs[0][sb0 + KSB * q * SB]) = rb[q];
  __syncthreads();

  d4 acc[TN][TM];
#pragma unroll
  for (int v = 0; v < TN; ++v)
#pragma unroll
    for (int u = 0; u < TM; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};

  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + wr + l15, offB = lk * SB + wc + l15;
  const int nk = kc1 - kc0;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      const double* pa = ga0 + (size_t)(kt + 1) * BK * lda;
      const double* pb = gb0 + (size_t)(kt + 1) * BK * ldb;
#pragma unroll
      for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(4 * q) * lda);
#pragma unroll
      for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(KSB * q) * ldb);
    }
    if (active) {
      if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
      const double* as = &As[buf][0];
      const double* bs = &Bs[buf][0];
#pragma unroll
      for (int s4 = 0; s4 < BK / 4; ++s4) {
        double fa[TM], fb[TN];
#pragma unroll
        for (int u = 0; u < TM; ++u) fa[u] = as[offA + 4 * s4 * SA + 16 * u];      // rows of C: the MFMA's B operand
#pragma unroll
        for (int v = 0; v < TN; ++v) fb[v] = bs[offB + 4 * s4 * SB + 16 * v];      // columns of C: the MFMA's A operand
#pragma unroll
        for (int v = 0; v < TN; ++v)
#pragma unroll
          for (int uu = 0; uu < TM; ++uu) {
            const int u = (v & 1) ? TM - 1 - uu : uu;                              // serpentine: one operand is reused each time
            acc[v][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[v], fa[u], acc[v][u], 0, 0, 0);
          }
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if (kt + 1 < nk) {
#pragma unroll
      for (int q = 0; q < NLA; ++q) *reinterpret_cast<d2*>(&As[buf ^ 1][sa0 + 4 * q * SA]) = ra[q];
#pragma unroll
      for (int q = 0; q < NLB; ++q) *reinterpret_cast<d2*>(&Bs[buf ^ 1][sb0 + KSB * q * SB]) = rb[q];
    }
    __syncthreads();
  }
  if (!active) return;
#pragma unroll
  for (int v = 0; v < TN; ++v) {
    double* cpv = C + (size_t)(bn + wc + 16 * v + lk) * ldc + bm + wr + l15;
    if (SET) {
#pragma unroll
      for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * ldc + 16 * u] = acc[v][u][r];
    } else if (nparts == 1) {
      double cv[TM][4];
#pragma unroll
      for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * ldc + 16 * u];
#pragma unroll
      for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * ldc + 16 * u] = cv[u][r] - acc[v][u][r];
    } else {
#pragma unroll
      for (int u = 0; u < TM; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(cpv + (size_t)(4 * r) * ldc + 16 * u, -acc[v][u][r]);
    }
  }
}

