// lmm_kernels.hip -- CDNA4 (gfx950) device kernels for the ILMM/OILMM inference hot path.
//
// Everything is Float64 and column-major.  A latent's "factor matrix" is an NR x NC panel (leading
// dimension ld) whose first n columns hold the lower triangle of K_l + noise*I, columns n..NC-1 are an
// identity pad (NC = roundup(n, 64)), and rows >= NC are "rider" rows: right-hand sides (delta, or
// cross-Gram rows K(x*, x)) that ride along the factorisation and come out as  rider * L^-T, i.e. the
// forward-substituted vectors  (L^-1 delta)'  -- the triangular solve costs no extra kernel.
//
// Kernels:
//   K1 gram_kernel / ilmm_dense_assemble_kernel   HBM-write bound   (A17, A6, A7)
//   K2 diag64m_kernel (64x64 factor + inverse, one wave, MFMA rank-4 steps) + gemm44_kernel<64, SET> (TRSM by inverse) +
//      gemm16p_kernel / gemm16h_kernel (v_mfma_f64_16x16x4_f64 SYRK/GEMM trailing update; gemm44_kernel: the round-1 4x4x4 form),
//      batched over latents   MFMA-f64 bound (A6, A7, A10, A11, A14)
//   K3/K6 lml_reduce_kernel (logdet + quadratic form, wavefront shuffles), backsolve_step_kernel
//   K4/K5 tall_skinny_kernel (T*Y projection, H*T*Y residual norm), mix_kernel (H unprojection)
//   K7 trmv_lower_kernel (sample transform), axpy noise
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#include <map>
#include <queue>
#include <tuple>
#include <functional>
#include <mutex>
#include <type_traits>
#include <cstdlib>
#include "lmm_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define LMM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
__device__ __forceinline__ d2 mk2(double x, double y) { d2 v; v.x = x; v.y = y; return v; }

#define LOG2PI 1.8378770664093453

// ---------------------------------------------------------------------------------------------------
// Storage type of MATRICES (factor matrices, inverse diagonal blocks, cross-solve blocks): double, or float for the fp32
// compute mode (BASELINE configs[4]; lmm_set_compute_dtype).  Vectors (inputs, observations, normals, alpha, outputs) and all
// scalar arithmetic outside the MFMA update stay double in both modes.  Element indices are in elements of TS.
// ---------------------------------------------------------------------------------------------------
template <typename TS> struct MatIO;
template <> struct MatIO<double> {
  static __device__ __forceinline__ double ld1(const void* b, size_t i) { return reinterpret_cast<const double*>(b)[i]; }
  static __device__ __forceinline__ void st1(void* b, size_t i, double v) { reinterpret_cast<double*>(b)[i] = v; }
  static __device__ __forceinline__ void st2(void* b, size_t i, d2 v) { *reinterpret_cast<d2*>(reinterpret_cast<double*>(b) + i) = v; }
  static __device__ __forceinline__ void ld4(const void* b, size_t i, double* o) {
    const d2* q = reinterpret_cast<const d2*>(reinterpret_cast<const double*>(b) + i);
    const d2 u = q[0], v = q[1];
    o[0] = u.x; o[1] = u.y; o[2] = v.x; o[3] = v.y;
  }
  static __device__ __forceinline__ void st4(void* b, size_t i, const double* v) {
    d2* q = reinterpret_cast<d2*>(reinterpret_cast<double*>(b) + i);
    q[0] = mk2(v[0], v[1]); q[1] = mk2(v[2], v[3]);
  }
};
template <> struct MatIO<float> {
  static __device__ __forceinline__ double ld1(const void* b, size_t i) { return (double)reinterpret_cast<const float*>(b)[i]; }
  static __device__ __forceinline__ void st1(void* b, size_t i, double v) { reinterpret_cast<float*>(b)[i] = (float)v; }
  static __device__ __forceinline__ void st2(void* b, size_t i, d2 v) {
    *reinterpret_cast<float2*>(reinterpret_cast<float*>(b) + i) = make_float2((float)v.x, (float)v.y);
  }
  static __device__ __forceinline__ void ld4(const void* b, size_t i, double* o) {
    const float4 u = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(b) + i);
    o[0] = u.x; o[1] = u.y; o[2] = u.z; o[3] = u.w;
  }
  static __device__ __forceinline__ void st4(void* b, size_t i, const double* v) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(b) + i) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
};

// ---------------------------------------------------------------------------------------------------
// math helpers
// ---------------------------------------------------------------------------------------------------
// exp(x) for x <= 0, Float64, < 1 ulp-ish: Cody-Waite reduction + degree-13 Taylor on |r| <= ln2/2
// (truncation 4e-18) + v_ldexp_f64 (handles the subnormal tail; exp(-inf) -> 0).
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -800.0);
  const double k = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;            // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);    // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);   // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);   // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);  // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);    // 1/8!
  p = __builtin_fma(p, r, 1.984126984126984e-04);   // 1/7!
  p = __builtin_fma(p, r, 1.388888888888889e-03);   // 1/6!
  p = __builtin_fma(p, r, 8.333333333333333e-03);   // 1/5!
  p = __builtin_fma(p, r, 4.1666666666666664e-02);  // 1/4!
  p = __builtin_fma(p, r, 1.6666666666666666e-01);  // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)k);
}

// kappa(r) of KernelFunctions' SEKernel / Matern32Kernel / Matern52Kernel,
// r = |x - x'| / lengthscale, r2 = r^2.
__device__ __forceinline__ double kappa(int kind, double var, double r, double r2) {
  if (kind == LMM_KERNEL_SE) return var * exp_nonpos(-0.5 * r2);
  if (kind == LMM_KERNEL_MATERN32) {
    const double s = 1.7320508075688772 * r;
    return var * (1.0 + s) * exp_nonpos(-s);
  }
  const double s = 2.23606797749979 * r;
  return var * (1.0 + s + (5.0 / 3.0) * r2) * exp_nonpos(-s);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// Block-wide sum for 256-thread blocks; result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* sh4) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh4[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) r = sh4[0] + sh4[1] + sh4[2] + sh4[3];
  __syncthreads();
  return r;
}

// ---------------------------------------------------------------------------------------------------
// K1: Gram assembly (lower triangle of K + diag_add*I, identity pad, rider rows).  HBM-write bound:
// each thread produces two consecutive rows (one 16-byte store) for 8 columns of a 64x64 tile.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double scaled_dist2(const double* __restrict__ a, const double* __restrict__ b,
                                               int d, double inv_ls) {
  double s = 0.0;
  for (int k = 0; k < d; ++k) {
    const double t = (a[k] - b[k]) * inv_ls;
    s = __builtin_fma(t, t, s);
  }
  return s;
}

// sqrt for the d > 1 Matern distances: v_rsq_f64 seed, two Newton steps on 1/sqrt, one Heron correction (<= 1 ulp-ish for
// normal arguments; 0 -> 0 exactly).  The library sqrt carries scaling for subnormal / huge arguments that scaled squared
// distances never need and costs about twice as much.
__device__ __forceinline__ double sqrt_dist(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = y * __builtin_fma(-h * y, y, 1.5);
  y = y * __builtin_fma(-h * y, y, 1.5);
  double r = x * y;
  r = __builtin_fma(0.5 * y, __builtin_fma(-r, r, x), r);
  return (x > 1e-300) ? r : 0.0;
}

// 16-byte store of two consecutive rows of a Gram column (non-temporal stores measured no different: 4.7-4.9 TB/s either way, round 2)
template <typename TS>
__device__ __forceinline__ void gram_store(void* base, size_t idx, d2 v) { MatIO<TS>::st2(base, idx, v); }

template <int KIND>
__device__ __forceinline__ double kappa_t(double var, double r, double r2) {
  if (KIND == LMM_KERNEL_SE) return var * exp_nonpos(-0.5 * r2);
  if (KIND == LMM_KERNEL_MATERN32) {
    const double s = 1.7320508075688772 * r;
    return var * (1.0 + s) * exp_nonpos(-s);
  }
  const double s = 2.23606797749979 * r;
  return var * __builtin_fma(5.0 / 3.0, r2, 1.0 + s) * exp_nonpos(-s);
}

template <typename TS>
__device__ __forceinline__ void gram_tile_generic(const GramArgs& a, int ti, int tj) {
  const int t = threadIdx.x;
  const int i0 = ti * 64 + 2 * (t & 31);
  const int cg = t >> 5;
  // Classify the two rows this thread owns (constant over the column loop).
  // type 0: data point of x; 1: zero row (pad); 2: rider from buffer; 3: cross-Gram row of xs.
  int rtype[2];
  const double* rpt[2];
  double rx[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = i0 + e;
    rtype[e] = 1; rpt[e] = nullptr; rx[e] = 0.0;
    if (i < a.n) { rtype[e] = 0; rpt[e] = a.x + (size_t)i * a.d; }
    else if (i >= a.ncols) {
      const int r = i - a.ncols;
      if (a.xs != nullptr && r < a.ns) { rtype[e] = 3; rpt[e] = a.xs + (size_t)r * a.d; }
      else if (a.rider != nullptr && r < a.nrider) { rtype[e] = 2; rpt[e] = a.rider + (size_t)r * a.rider_ld; }
    }
    if ((rtype[e] == 0 || rtype[e] == 3) && a.d == 1) rx[e] = rpt[e][0];
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = tj * 64 + cg + 8 * q;
    d2 v;
    if (j >= a.n) {                                        // identity pad column
      v.x = (i0 == j) ? a.pad_diag : 0.0;
      v.y = (i0 + 1 == j) ? a.pad_diag : 0.0;
    } else {
      double xj = 0.0;
      if (a.d == 1) xj = a.x[j];
      double out[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        double val = 0.0;
        if (rtype[e] == 0 || rtype[e] == 3) {
          double r, r2;
          if (a.d == 1) { r = fabs(rx[e] - xj) * a.inv_ls; r2 = r * r; }
          else { r2 = scaled_dist2(rpt[e], a.x + (size_t)j * a.d, a.d, a.inv_ls); r = sqrt(r2); }
          val = kappa(a.kind, a.var, r, r2);
          if (rtype[e] == 0 && i0 + e == j) val += a.diag_add + (a.diag_vec ? a.diag_vec[j] : 0.0);
        } else if (rtype[e] == 2) {
          val = rpt[e][j] - a.rider_sub;
        }
        out[e] = val;
      }
      v.x = out[0]; v.y = out[1];
    }
    MatIO<TS>::st2(a.A, (size_t)j * a.ld + (i0 - a.row_shift), v);
  }
}

// exp for moderate arguments of either sign (|x| <= ~700): same reduction and polynomial as exp_nonpos.
__device__ __forceinline__ double exp_any(double x) {
  const double k = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(-k, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;
  p = __builtin_fma(p, r, 2.08767569878681e-09);
  p = __builtin_fma(p, r, 2.505210838544172e-08);
  p = __builtin_fma(p, r, 2.755731922398589e-07);
  p = __builtin_fma(p, r, 2.7557319223985893e-06);
  p = __builtin_fma(p, r, 2.48015873015873e-05);
  p = __builtin_fma(p, r, 1.984126984126984e-04);
  p = __builtin_fma(p, r, 1.388888888888889e-03);
  p = __builtin_fma(p, r, 8.333333333333333e-03);
  p = __builtin_fma(p, r, 4.1666666666666664e-02);
  p = __builtin_fma(p, r, 1.6666666666666666e-01);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)k);
}

// One workgroup assembles a 64-row x 256-column strip (4 tiles).  Interior tiles (all rows and columns are data points,
// d == 1) take a branch-free path specialised on the kernel kind; border / rider / pad tiles and d > 1 use the generic
// tile routine.  For the Matern kernels on d == 1 the exponential factor is SEPARABLE:
//     exp(-a |xi - xj|) = min( E_i F_j , F_i E_j ),   E = exp(a (x - c)),  F = exp(-a (x - c)),
// with per-strip reference points c (rows: c_r, each column tile: c_c, and the scalar exp(+-a (c_r - c_c)) folded into the
// column factors), so the per-element cost drops from a full exp (~20 f64 ops) to 2 multiplies and a min; the 4 row and
// 2 x 64 column exponentials are amortised over 64 elements per thread.  Guard: |a (x - c)| <= 40 inside the strip (else
// the direct per-element exp is used, e.g. for unsorted inputs).  Relative error of the product form <= ~1e-14.
template <int KIND, bool ND, typename TS>      // ND: the 1 < d <= 8 fast path is compiled in (kept out of the d == 1 kernel's register budget)
__device__ __forceinline__ void gram_body(const GramArgs& a) {
  constexpr int DMAX = 8;                                     // input dimensions with a fast path (d > DMAX: generic tiles)
  __shared__ double colE[64], colF[64], colX[64];
  __shared__ double colP[ND ? 64 * DMAX : 1];
  const int ti = blockIdx.x + a.row_tile0, sy = blockIdx.y;   // (a 1-D grid over the non-empty strips measured 4 % slower)
  const int t = threadIdx.x;
  const int i0 = ti * 64 + 2 * (t & 31);
  const int cg = t >> 5;
  // Row source of this strip when all 64 rows are points: training inputs x (rows < n) or the cross-Gram inputs xs
  // (rider rows ncols .. ncols + ns - 1 of the predict path); otherwise the strip mixes kinds and takes the generic tiles.
  const double* rsrc = nullptr;
  int rbase = 0;
  if (ti * 64 + 63 < a.n) { rsrc = a.x; }
  else if (a.xs != nullptr && ti * 64 >= a.ncols && ti * 64 + 63 - a.ncols < a.ns) { rsrc = a.xs; rbase = a.ncols; }
  const bool rows_interior = (a.d == 1) && rsrc != nullptr;
  const bool rows_nd = ND && (a.d > 1) && (a.d <= DMAX) && rsrc != nullptr;
  double xr0[DMAX], xr1[DMAX];                                // d > 1: this thread's two row points, pre-scaled by 1/lengthscale
  if (ND && rows_nd) {
#pragma unroll
    for (int k = 0; k < DMAX; ++k) {
      xr0[k] = (k < a.d) ? rsrc[(size_t)(i0 - rbase) * a.d + k] * a.inv_ls : 0.0;
      xr1[k] = (k < a.d) ? rsrc[(size_t)(i0 + 1 - rbase) * a.d + k] * a.inv_ls : 0.0;
    }
  }
  constexpr bool SEP = (KIND != LMM_KERNEL_SE);
  const double aS = (KIND == LMM_KERNEL_MATERN32 ? 1.7320508075688772 : 2.23606797749979) * a.inv_ls;
  double x0 = 0.0, x1 = 0.0, cr = 0.0, E0 = 0.0, F0 = 0.0, E1 = 0.0, F1 = 0.0;
  bool rows_ok = false;
  if (rows_interior) {
    x0 = rsrc[i0 - rbase]; x1 = rsrc[i0 + 1 - rbase]; cr = rsrc[ti * 64 - rbase];
    if (SEP) {
      const double u0 = aS * (x0 - cr), u1 = aS * (x1 - cr);
      rows_ok = fabs(u0) <= 40.0 && fabs(u1) <= 40.0;
      if (rows_ok) { E0 = exp_any(u0); F0 = exp_any(-u0); E1 = exp_any(u1); F1 = exp_any(-u1); }
    }
  }
  for (int c4 = 0; c4 < a.cpw; ++c4) {
    const int tj = sy * a.cpw + c4;
    if (tj * 64 >= a.ncols) break;
    if (!a.full && ti < tj) break;                          // lower tiles only
    const bool cols_in = (tj * 64 + 63 < a.n);
    const size_t out = (size_t)(tj * 64 + cg) * a.ld + (i0 - a.row_shift);      // element index of this thread's first store
    if (ND && rows_nd && cols_in) {                         // d > 1 interior tile: column points staged (pre-scaled) in LDS
      __syncthreads();
      for (int e = t; e < 64 * a.d; e += 256) colP[e] = a.x[(size_t)tj * 64 * a.d + e] * a.inv_ls;
      __syncthreads();
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jl = cg + 8 * q;
        const double* cp = colP + jl * a.d;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int k = 0; k < DMAX; ++k) {
          if (k < a.d) {
            const double c = cp[k];
            const double t0 = xr0[k] - c, t1 = xr1[k] - c;
            s0 = __builtin_fma(t0, t0, s0); s1 = __builtin_fma(t1, t1, s1);
          }
        }
        d2 v;
        v.x = kappa_t<KIND>(a.var, KIND == LMM_KERNEL_SE ? 0.0 : sqrt_dist(s0), s0);
        v.y = kappa_t<KIND>(a.var, KIND == LMM_KERNEL_SE ? 0.0 : sqrt_dist(s1), s1);
        if (ti == tj) {
          const int j = tj * 64 + jl;
          const double da = a.diag_add + (a.diag_vec ? a.diag_vec[j] : 0.0);
          if (i0 == j) v.x += da;
          if (i0 + 1 == j) v.y += da;
        }
        gram_store<TS>(a.A, out + (size_t)(8 * q) * a.ld, v);
      }
      continue;
    }
    const bool interior = rows_interior && cols_in;
    if (!interior) { gram_tile_generic<TS>(a, ti, tj); continue; }
    bool sep = false;
    if (SEP) {
      __syncthreads();                                      // previous tile's readers are done with colE/colF/colX
      bool ok = true;
      if (t < 64) {
        const double xj = a.x[tj * 64 + t], cc = a.x[tj * 64];
        const double vj = aS * (xj - cc);
        double D = aS * (cr - cc);
        D = fmin(fmax(D, -700.0), 700.0);
        ok = fabs(vj) <= 40.0;
        const double vc = ok ? vj : 0.0;
        // a (xi - xj) = u_i - v_j + D  (u = a (xi - c_r), v = a (xj - c_c), D = a (c_r - c_c)):
        colE[t] = exp_any(vc) * exp_any(-D);                // F_i * colE[j] = exp(-a (xi - xj))
        colF[t] = exp_any(-vc) * exp_any(D);                // E_i * colF[j] = exp(+a (xi - xj));  min picks the one <= 1
        colX[t] = xj;
      }
      sep = __syncthreads_and(ok && rows_ok) != 0;
    }
    if (SEP && sep) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int jl = cg + 8 * q;
        const double xj = colX[jl], cE = colE[jl], cF = colF[jl];
        const double s0 = aS * fabs(x0 - xj), s1 = aS * fabs(x1 - xj);
        const double e0 = fmin(E0 * cF, F0 * cE), e1 = fmin(E1 * cF, F1 * cE);
        d2 v;
        if (KIND == LMM_KERNEL_MATERN32) { v.x = a.var * (1.0 + s0) * e0; v.y = a.var * (1.0 + s1) * e1; }
        else {
          v.x = a.var * __builtin_fma(s0 * s0, 1.0 / 3.0, 1.0 + s0) * e0;
          v.y = a.var * __builtin_fma(s1 * s1, 1.0 / 3.0, 1.0 + s1) * e1;
        }
        if (ti == tj) {
          const int j = tj * 64 + jl;
          const double da = a.diag_add + (a.diag_vec ? a.diag_vec[j] : 0.0);     // per-point noise (sequential conditioning)
          if (i0 == j) v.x += da;
          if (i0 + 1 == j) v.y += da;
        }
        gram_store<TS>(a.A, out + (size_t)(8 * q) * a.ld, v);
      }
    } else {
      const double* xc = a.x + tj * 64 + cg;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const double xj = xc[8 * q];
        const double r0 = fabs(x0 - xj) * a.inv_ls, r1 = fabs(x1 - xj) * a.inv_ls;
        d2 v;
        v.x = kappa_t<KIND>(a.var, r0, r0 * r0);
        v.y = kappa_t<KIND>(a.var, r1, r1 * r1);
        if (ti == tj) {
          const int j = tj * 64 + cg + 8 * q;
          const double da = a.diag_add + (a.diag_vec ? a.diag_vec[j] : 0.0);
          if (i0 == j) v.x += da;
          if (i0 + 1 == j) v.y += da;
        }
        gram_store<TS>(a.A, out + (size_t)(8 * q) * a.ld, v);
      }
    }
  }
}

template <int KIND, bool ND, typename TS>
__global__ __launch_bounds__(256) void gram_kernel(GramArgs a) {
  if (a.info_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *a.info_zero = 0;
  gram_body<KIND, ND, TS>(a);
}

template <int KIND, bool ND, typename TS>
__global__ __launch_bounds__(256) void gram_batch_kernel(GramBatchArgs b) {
  GramArgs a = b.base;
  const int z = blockIdx.z;
  a.A = b.A[z]; a.var = b.var[z]; a.inv_ls = b.inv_ls[z]; a.diag_add = b.diag_add[z]; a.diag_vec = b.diag_vec[z]; a.rider = b.rider[z]; a.rider_sub = b.rider_sub[z];
  if (b.info_zero[z] && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *b.info_zero[z] = 0;
  gram_body<KIND, ND, TS>(a);
}

// K8: dense ILMM latent covariance  blockdiag(K_1..K_m) + SigmaT (x) I_n  (+ mean-free rider row):
// element (i, j), i = li*n + ii, j = lj*n + jj  ->  [li == lj] kappa_li(x_ii, x_jj) + [ii == jj] SigmaT[li, lj].
// Reference: src/ilmm.jl:160 kron(SigmaT, I) + src/independent_mogp.jl:60-63 BlockDiagonal.
template <typename TS>
__global__ __launch_bounds__(256) void ilmm_dense_assemble_kernel(DenseArgs a) {
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (ti < tj) return;
  const int t = threadIdx.x;
  const int i0 = ti * 64 + 2 * (t & 31);
  const int cg = t >> 5;
  const int N = a.m * a.n;
  int li[2], ii[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) { li[e] = (i0 + e) / a.n; ii[e] = (i0 + e) - li[e] * a.n; }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int j = tj * 64 + cg + 8 * q;
    double out[2] = {0.0, 0.0};
    if (j >= N) {
      out[0] = (i0 == j) ? 1.0 : 0.0;
      out[1] = (i0 + 1 == j) ? 1.0 : 0.0;
    } else {
      const int lj = j / a.n, jj = j - lj * a.n;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int i = i0 + e;
        double val = 0.0;
        if (i < N) {
          if (li[e] == lj) {
            const LatentDev g = a.lat[lj];
            double r, r2;
            if (a.d == 1) { r = fabs(a.x[ii[e]] - a.x[jj]) * g.inv_ls; r2 = r * r; }
            else { r2 = scaled_dist2(a.x + (size_t)ii[e] * a.d, a.x + (size_t)jj * a.d, a.d, g.inv_ls); r = sqrt(r2); }
            val = kappa(g.kind, g.var, r, r2);
          }
          if (ii[e] == jj) val += a.sigmaT[(size_t)(a.sig_idx ? a.sig_idx[jj] : 0) * a.m * a.m + li[e] + lj * a.m];
        } else if (i >= a.ncols) {
          const int r = i - a.ncols;
          if (a.rider != nullptr && r < a.nrider) val = a.rider[(size_t)r * a.rider_ld + j];
        }
        out[e] = val;
      }
    }
    d2 v; v.x = out[0]; v.y = out[1];
    MatIO<TS>::st2(a.A, (size_t)j * a.ld + i0, v);
  }
}

// Dense-ILMM cross-covariance riders: row (l, s) of R (m*ns rows) is K_l(xs_s, x) placed in the column block of
// latent l (reference src/independent_mogp.jl:66-71: block-diagonal cov(f, x, y)); zero elsewhere and in the pad.
template <typename TS>
__global__ __launch_bounds__(256) void dense_cross_kernel(void* __restrict__ R, int ldr, int nrows, int /*ncols*/,
                                                          const double* __restrict__ xs, int ns,
                                                          const double* __restrict__ x, int n, int d, int m,
                                                          const LatentDev* __restrict__ lat) {
  const int r = blockIdx.x * 256 + threadIdx.x;      // row (l, s)
  const int j = blockIdx.y;                          // column
  if (r >= nrows) return;
  double val = 0.0;
  const int l = r / ns, s = r - l * ns;
  if (l < m && j < m * n) {
    const int lj = j / n, jj = j - lj * n;
    if (lj == l) {
      const LatentDev g = lat[l];
      double rr, r2;
      if (d == 1) { rr = fabs(xs[s] - x[jj]) * g.inv_ls; r2 = rr * rr; }
      else { r2 = scaled_dist2(xs + (size_t)s * d, x + (size_t)jj * d, d, g.inv_ls); rr = sqrt(r2); }
      val = kappa(g.kind, g.var, rr, r2);
    }
  }
  MatIO<TS>::st1(R, (size_t)j * ldr + r, val);
}

// Dense-ILMM posterior variance (reference src/ilmm.jl:122-129 on a PosteriorGP latent):
//   V[o, s] = sum_l H[o,l]^2 (k_l(0) + jitter) + sigma2 - sum_k ( sum_l H[o,l] R[(l,s), k] )^2,   R = Kxs' L^-T.
// Thread per output element e = o * ns + s, the k range cut into chunks of kc columns (blockIdx.y) whose partial sums
// dense_var_finish_kernel subtracts from the prior term in a fixed order.
template <typename TS>
__global__ __launch_bounds__(256) void dense_var_kernel(const void* __restrict__ R, int ldr, int ns, int m, int Ncols, int kc,
                                                        const double* __restrict__ Hm, int p, double* __restrict__ partial) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= p * ns) return;
  const int o = e / ns, s = e - o * ns;
  const int k0 = blockIdx.y * kc;
  int k1 = k0 + kc; if (k1 > Ncols) k1 = Ncols;
  double q = 0.0;
  for (int k = k0; k < k1; ++k) {
    const size_t col = (size_t)k * ldr + s;
    double t = 0.0;
#pragma unroll 4
    for (int l = 0; l < m; ++l) t = __builtin_fma(Hm[o + (size_t)l * p], MatIO<TS>::ld1(R, col + (size_t)l * ns), t);
    q = __builtin_fma(t, t, q);
  }
  partial[(size_t)blockIdx.y * ((size_t)p * ns) + e] = q;
}

__global__ void dense_var_finish_kernel(const double* __restrict__ partial, int nch, int ns, int m, const double* __restrict__ Hm,
                                        int p, const LatentDev* __restrict__ lat, double jitter, double sigma2,
                                        double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p * ns) return;
  const int o = e / ns;
  double base = sigma2, q = 0.0;
  for (int l = 0; l < m; ++l) { const double h = Hm[o + (size_t)l * p]; base = __builtin_fma(h * h, lat[l].var + jitter, base); }
  for (int c = 0; c < nch; ++c) q += partial[(size_t)c * ((size_t)p * ns) + e];
  out[e] = base - q;
}

// ---------------------------------------------------------------------------------------------------
// K2a, third form (round 2): ONE WAVE per 64x64 block, rank-4 steps on the matrix pipe.
// tools/lat_probe (profiles/r02/lat_probe.log): a wave issues one f64 VALU instruction per 8 cycles whether or not it depends on
// the previous one, so the first two forms are bound by their per-wave INSTRUCTION COUNT (~260 f64 ops per rank-4 step in every
// wave: redundant pivot factorisation, substitution for the multipliers, 128 FMAs of rank-4 update), not by latency; one
// v_mfma_f64_16x16x4_f64 does a 16x16 block of a rank-4 update in 64 cycles (the same FMAs cost 128 cycles of VALU issue in a
// wave, and the MFMA leaves the VALU free).  So:
//   * the block and the inverse being built live as 16x16 MFMA accumulator blocks of one wave (lower block triangle of each:
//     20 blocks x 4 doubles per lane; element (4r+g, c) of a block is register r of lane 16g+c);
//   * per step (pivots J..J+3): the four pivot columns and the four pivot rows of the W part go through 2 + 2 KB of LDS so that
//     every lane gets the 4x4 pivot block (uniform reads) and its own row of the panel / column of the W rows (lane 16g+c: row c
//     of each 16-row block, column c of each 16-column block);
//   * every lane factors the pivot block (true Cholesky, v_rsq_f64 + Newton) and inverts the 4x4 factor; lane group g keeps row g
//     of that inverse, so Y[row][g] = panel row . Linv[g][:] (the final L entries, stored straight to global) and
//     Z[g][col] = Linv[g][:] . Wtop[:][col] (the final rows J..J+3 of W) come out ALREADY in the MFMA operand layouts
//     (A: lane (i = c, k = g); B: lane (k = g, j = c));
//   * trailing update S -= Y Y', W_below -= Y Z: one MFMA per live 16x16 block (10 per step on average).
// No barrier (one wave), no D^1/2 rescaling at the end.  Compile-time register indices throughout (16 steps unrolled).
// ---------------------------------------------------------------------------------------------------
struct Diag64mState {
  d4 S[4][4], V[4][4];                                             // [block row][block col], block col <= block row
  double Ym[2][4], nY[2][4], Z[2][4];                              // operands of step s live in set s & 1 (the lagging MFMAs of step
                                                                   // s are issued inside step s+1, under its VALU chain)
  double e0, e1, e2, e3;                                           // lane-group indicator (row g of the 4x4 inverse)
  unsigned long long badmask;                                      // bit k: pivot k is not > 0 (uniform)
};
// The MFMAs of step T in three groups, by when the NEXT step (pivot block column jn) needs their block:
//   0: S[jn][jn], 1: S[rb > jn][jn] and V[jn][*]  -- the next panel / the next pivot rows of W: issued at the end of step T
//   2: everything else                              -- issued inside step T+1, between its LDS reads and its VALU chain, so that the
//                                                      LDS round trip of step T+1 is covered by matrix-pipe work
// (f64 MFMAs and f64 VALU instructions of one wave do NOT overlap on gfx950 -- interleaving them one-for-one changed nothing,
// profiles/r02/diag_ab_*.log -- so the kernel is bound by its instruction count: ~9 MFMAs of 64 cycles and ~60 f64 VALU
// instructions of 8 cycles per step.)
template <int T, int GROUP>
__device__ __forceinline__ void diag64m_mfmas(Diag64mState& st) {
  constexpr int jb = T >> 2, q = T & 3, jn = (T + 1) >> 2, o = T & 1;
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    if (rb == jb && q == 3) continue;                              // no rows below the pivots in this block row
#pragma unroll
    for (int cb = jb; cb <= rb; ++cb) {
      if (cb == jb && q == 3) continue;                            // block column jb is finished
      if ((cb == jn ? (rb == jn ? 0 : 1) : 2) != GROUP) continue;
      st.S[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(st.nY[o][rb], st.Ym[o][cb], st.S[rb][cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb <= jb; ++cb) {
      if ((rb == jn ? 1 : 2) != GROUP) continue;
      st.V[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(st.nY[o][rb], st.Z[o][cb], st.V[rb][cb], 0, 0, 0);
    }
  }
}
constexpr int DIAG_SP = 18;                                        // row stride of Sp (doubles): 16-byte aligned rows, conflict-free b128 reads
constexpr int DIAG_LS = 68;                                        // column stride of Lo / Wl
// DIRECT: the finished L entries of a step leave for global memory straight from the registers (four columns x sixteen consecutive
// rows per store instruction: four 128-byte segments) instead of being collected in the Lo image for a coalesced epilogue --
// the form leaf128 uses, which has no LDS to spare for Lo.
template <int s, bool DIRECT = false, typename TS = double>
__device__ __forceinline__ void diag64m_step(Diag64mState& st, double* __restrict__ Sp, double* __restrict__ Wt, double* __restrict__ Lo,
                                             int c, int g, void* __restrict__ Ag = nullptr, size_t offAg = 0, int ldg = 0) {
  constexpr int SP = DIAG_SP, LS = DIAG_LS;
  constexpr int J = 4 * s, jb = s >> 2, q = s & 3, o = s & 1;
  __builtin_amdgcn_sched_barrier(0);
  // one wave: LDS executes its instructions in order, and the compiler keeps may-aliasing LDS accesses in program order
  // the current block column of S and the pivot rows of the W part -> LDS
#pragma unroll
  for (int rb = jb; rb < 4; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) Sp[(16 * rb + 4 * r + g) * SP + c] = st.S[rb][jb][r];
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb) Wt[(16 * cb + c) * 4 + g] = st.V[jb][cb][q];
  // -> the 4x4 pivot block (uniform), my row of each 16-row block of the panel, my column of each block of the W rows
  const d2 pr0 = *reinterpret_cast<const d2*>(&Sp[(J + 0) * SP + 4 * q]);
  const d2 pr1 = *reinterpret_cast<const d2*>(&Sp[(J + 1) * SP + 4 * q]);
  const d2 pr2a = *reinterpret_cast<const d2*>(&Sp[(J + 2) * SP + 4 * q]), pr2b = *reinterpret_cast<const d2*>(&Sp[(J + 2) * SP + 4 * q + 2]);
  const d2 pr3a = *reinterpret_cast<const d2*>(&Sp[(J + 3) * SP + 4 * q]), pr3b = *reinterpret_cast<const d2*>(&Sp[(J + 3) * SP + 4 * q + 2]);
  d2 bo[4][2], wo[4][2];
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    bo[rb][0] = *reinterpret_cast<const d2*>(&Sp[(16 * rb + c) * SP + 4 * q]);
    bo[rb][1] = *reinterpret_cast<const d2*>(&Sp[(16 * rb + c) * SP + 4 * q + 2]);
  }
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb) {
    wo[cb][0] = *reinterpret_cast<const d2*>(&Wt[(16 * cb + c) * 4]);
    wo[cb][1] = *reinterpret_cast<const d2*>(&Wt[(16 * cb + c) * 4 + 2]);
  }
  if constexpr (s > 0) diag64m_mfmas<s - 1, 2>(st);                // the previous step's lagging blocks
  auto rsq = [](double x) {          // 1/sqrt(x): v_rsq_f64 + two Newton steps
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
  };
  // 4x4 Cholesky P = Lp Lp'
  const double d0 = pr0.x, r0 = rsq(d0);
  const double l10 = pr1.x * r0, l20 = pr2a.x * r0, l30 = pr3a.x * r0;
  const double d1 = __builtin_fma(-l10, l10, pr1.y), r1 = rsq(d1);
  const double l21 = __builtin_fma(-l20, l10, pr2a.y) * r1, l31 = __builtin_fma(-l30, l10, pr3a.y) * r1;
  const double d2v = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, pr2b.x)), r2 = rsq(d2v);
  const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, pr3b.x)) * r2;
  const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, pr3b.y))), r3 = rsq(d3);
  // row g of Lp^-1: x' Lp = e_g' by back substitution (zero above the diagonal by construction)
  const double k3 = st.e3 * r3;
  const double k2 = __builtin_fma(-k3, l32, st.e2) * r2;
  const double k1 = __builtin_fma(-k2, l21, __builtin_fma(-k3, l31, st.e1)) * r1;
  const double k0 = __builtin_fma(-k1, l10, __builtin_fma(-k2, l20, __builtin_fma(-k3, l30, st.e0))) * r0;
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    const double y = __builtin_fma(bo[rb][1].y, k3, __builtin_fma(bo[rb][1].x, k2, __builtin_fma(bo[rb][0].y, k1, bo[rb][0].x * k0)));
    const int row = 16 * rb + c;
    if constexpr (DIRECT) { if (row >= J + g) MatIO<TS>::st1(Ag, offAg + (size_t)(J + g) * ldg + row, y); }
    else Lo[(J + g) * LS + row] = y;                               // L[row][J+g], final for row >= J+g (the rest is never stored)
    st.Ym[o][rb] = (rb > jb || row > J + 3) ? y : 0.0;             // rows of and above the pivot block take no part in the update
    st.nY[o][rb] = -st.Ym[o][rb];
  }
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb)
    st.Z[o][cb] = __builtin_fma(wo[cb][1].y, k3, __builtin_fma(wo[cb][1].x, k2, __builtin_fma(wo[cb][0].y, k1, wo[cb][0].x * k0)));
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb) st.V[jb][cb][q] = st.Z[o][cb];  // rows J..J+3 of W = Lp^-1 Wtop, final (the MFMAs of this step
                                                                   // add 0 * Z to them: rows of the pivot block have Ym = 0)
  diag64m_mfmas<s, 0>(st); diag64m_mfmas<s, 1>(st);                // the blocks the next step reads
  __builtin_amdgcn_sched_barrier(0);
  // non-positive (or NaN) pivots: the values are uniform, so any lane's comparison will do
  if (__builtin_amdgcn_ballot_w64(!(d0 > 0.0) | !(d1 > 0.0) | !(d2v > 0.0) | !(d3 > 0.0)) != 0ull) {
    const unsigned m4 = (!(d0 > 0.0) ? 1u : 0u) | (!(d1 > 0.0) ? 2u : 0u) | (!(d2v > 0.0) ? 4u : 0u) | (!(d3 > 0.0) ? 8u : 0u);
    st.badmask |= (unsigned long long)__builtin_amdgcn_readfirstlane(m4) << J;
  }
}
template <int s>
__device__ __forceinline__ void diag64m_steps(Diag64mState& st, double* __restrict__ Sp, double* __restrict__ Wt, double* __restrict__ Lo,
                                              int c, int g) {
  if constexpr (s < 16) { diag64m_step<s>(st, Sp, Wt, Lo, c, g); diag64m_steps<s + 1>(st, Sp, Wt, Lo, c, g); }
}
template <int s, typename TS>
__device__ __forceinline__ void diag64m_steps_direct(Diag64mState& st, double* __restrict__ Sp, double* __restrict__ Wt, int c, int g,
                                                     void* __restrict__ Ag, size_t offAg, int ldg) {
  if constexpr (s < 16) {
    diag64m_step<s, true, TS>(st, Sp, Wt, nullptr, c, g, Ag, offAg, ldg);
    diag64m_steps_direct<s + 1, TS>(st, Sp, Wt, c, g, Ag, offAg, ldg);
  }
}

// One wave factors the 64 x 64 block at (A, offA, ld) in place (lower triangle) and leaves W = L^-1 in the LDS image
// Wl[col * DIAG_LS + row] (zeros above the diagonal).  Sp: 64 * DIAG_SP doubles, Wt: 256 doubles of LDS work area.  Returns the mask of
// non-positive (or NaN) pivots.  l = lane.  The last step's lagging MFMAs (group 2 of step 15) do not exist: nothing is pending.
// Wl MAY overlap Sp / Wt (leaf128 does that to stay inside the update kernel's LDS footprint): it is written only after the last step.
// src != nullptr: the block is taken from the LDS image src[col * DIAG_LS + row] instead of global memory (L still goes to A).
template <typename TS>
__device__ __forceinline__ unsigned long long diag64m_wave(void* __restrict__ A, size_t offA, int ld, double* Sp, double* Wt, double* Wl, int l,
                                                           const double* src = nullptr) {
  const int c = l & 15, g = l >> 4;
  Diag64mState st;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rb + 4 * r + g, col = 16 * cb + c;
        st.S[rb][cb][r] = (row >= col) ? (src ? src[col * DIAG_LS + row] : MatIO<TS>::ld1(A, offA + (size_t)col * ld + row)) : 0.0;
        st.V[rb][cb][r] = (row == col) ? 1.0 : 0.0;
      }
  st.e0 = g == 0 ? 1.0 : 0.0; st.e1 = g == 1 ? 1.0 : 0.0; st.e2 = g == 2 ? 1.0 : 0.0; st.e3 = g == 3 ? 1.0 : 0.0;
  st.badmask = 0ull;
  diag64m_steps_direct<0, TS>(st, Sp, Wt, c, g, A, offA, ld);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Wl[(16 * cb + c) * DIAG_LS + 16 * rb + 4 * r + g] = cb <= rb ? st.V[rb][cb][r] : 0.0;
  return st.badmask;
}
// ---------------------------------------------------------------------------------------------------
// K2a, two-wave form (round 4): the same elimination with the FACTOR and the INVERSE on different waves (different SIMDs).
// Nothing in the factorisation depends on the inverse being built: per step the factor wave needs the block column of S, the 4 x 4
// pivot factor, its panel rows Y and the S updates; the inverse wave needs only Y and the rows k of the 4 x 4 inverse from it, and
// owns everything about W (the pivot rows through Wt, Z = Lp^-1 Wtop, the V updates).  So wave 0 runs the factor chain -- its step
// sheds the Z dot products, the Wt round trip and 4-6 of its 9-14 MFMAs -- and hands (Y, k) of each step to wave 1 through a
// double-buffered LDS exchange area; wave 1 follows one step behind and writes the W image at the end.  Every accumulator sees the
// same operations in the same order as in the one-wave form: the results are bitwise identical.  (A split of the MFMAs over four waves
// would leave the ~120 f64 VALU instructions of a step -- 8 issue cycles each -- in every wave: priced at 1.15x; this split removes
// instructions from the critical wave instead.)
//   xch: 2 x 8 x 64 doubles ([parity][v][lane]: v < 4 the panel rows -Y of block row v, v >= 4 the inverse row entries k_{v-4}),
//   flags (LDS ints, monotonic over the kernel's lifetime, zeroed once at kernel start): [0] steps produced, [1] steps consumed by the
//   inverse wave, [2] by the store wave (wave 2 takes the global stores of the finished L entries -- their address arithmetic and
//   execution masks -- off the factor wave as well);
//   `base` = 16 x (number of two-wave factorisations this workgroup has run before): uniform over the workgroup.
// ---------------------------------------------------------------------------------------------------
struct DiagFState {
  d4 S[4][4];
  double Ym[2][4], nY[2][4];
  double e0, e1, e2, e3;
  unsigned long long badmask;
};
struct DiagWState { d4 V[4][4]; };
constexpr int DIAG_XCH = 2 * 8 * 64;                               // doubles of the exchange area
__device__ __forceinline__ void diag_flag_wait(const int* f, int need) {
  while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need) __builtin_amdgcn_s_sleep(0);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void diag_flag_set(int* f, int v, int l) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");            // this wave's LDS writes (or reads: s_waitcnt lgkmcnt(0)) are complete
  if (l == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The producer's form: the LDS executes one wave's instructions in issue order, so a flag store issued behind the data stores is
// performed behind them -- no s_waitcnt between them (it sat on the factor wave's chain, 16 times per block); the asm statements only
// keep the compiler from reordering.
__device__ __forceinline__ void diag_flag_set_inorder(int* f, int v, int l) {
  asm volatile("" ::: "memory");
  if (l == 0) __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  asm volatile("" ::: "memory");
}
template <int T, int GROUP>
__device__ __forceinline__ void diagf_mfmas(DiagFState& st) {
  constexpr int jb = T >> 2, q = T & 3, jn = (T + 1) >> 2, o = T & 1;
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    if (rb == jb && q == 3) continue;
#pragma unroll
    for (int cb = jb; cb <= rb; ++cb) {
      if (cb == jb && q == 3) continue;
      if ((cb == jn ? (rb == jn ? 0 : 1) : 2) != GROUP) continue;
      st.S[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(st.nY[o][rb], st.Ym[o][cb], st.S[rb][cb], 0, 0, 0);
    }
  }
}
template <int s, typename TS>
__device__ __forceinline__ void diagf_step(DiagFState& st, double* __restrict__ Sp, double* __restrict__ xch, int* __restrict__ flags, int base,
                                           int c, int g, int l, void* __restrict__ Ag, size_t offAg, int ldg) {
  constexpr int SP = DIAG_SP;
  constexpr int J = 4 * s, jb = s >> 2, q = s & 3, o = s & 1;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int rb = jb; rb < 4; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) Sp[(16 * rb + 4 * r + g) * SP + c] = st.S[rb][jb][r];
  // the consumers' counters, read together with the pivot data (their LDS latency then costs nothing): the exchange buffer of this
  // step was last read for step s - 2, and both consumers are normally long past it
  int cons1 = 0, cons2 = 0;
  if constexpr (s >= 2) {
    cons1 = __hip_atomic_load(flags + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    cons2 = __hip_atomic_load(flags + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  const d2 pr0 = *reinterpret_cast<const d2*>(&Sp[(J + 0) * SP + 4 * q]);
  const d2 pr1 = *reinterpret_cast<const d2*>(&Sp[(J + 1) * SP + 4 * q]);
  const d2 pr2a = *reinterpret_cast<const d2*>(&Sp[(J + 2) * SP + 4 * q]), pr2b = *reinterpret_cast<const d2*>(&Sp[(J + 2) * SP + 4 * q + 2]);
  const d2 pr3a = *reinterpret_cast<const d2*>(&Sp[(J + 3) * SP + 4 * q]), pr3b = *reinterpret_cast<const d2*>(&Sp[(J + 3) * SP + 4 * q + 2]);
  d2 bo[4][2];
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    bo[rb][0] = *reinterpret_cast<const d2*>(&Sp[(16 * rb + c) * SP + 4 * q]);
    bo[rb][1] = *reinterpret_cast<const d2*>(&Sp[(16 * rb + c) * SP + 4 * q + 2]);
  }
  if constexpr (s > 0) diagf_mfmas<s - 1, 2>(st);                  // the previous step's lagging blocks
  auto rsq = [](double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double h = 0.5 * x;
    y = y * __builtin_fma(-h * y, y, 1.5);
    y = y * __builtin_fma(-h * y, y, 1.5);
    return y;
  };
  const double d0 = pr0.x, r0 = rsq(d0);
  const double l10 = pr1.x * r0, l20 = pr2a.x * r0, l30 = pr3a.x * r0;
  const double d1 = __builtin_fma(-l10, l10, pr1.y), r1 = rsq(d1);
  const double l21 = __builtin_fma(-l20, l10, pr2a.y) * r1, l31 = __builtin_fma(-l30, l10, pr3a.y) * r1;
  const double d2v = __builtin_fma(-l21, l21, __builtin_fma(-l20, l20, pr2b.x)), r2 = rsq(d2v);
  const double l32 = __builtin_fma(-l31, l21, __builtin_fma(-l30, l20, pr3b.x)) * r2;
  const double d3 = __builtin_fma(-l32, l32, __builtin_fma(-l31, l31, __builtin_fma(-l30, l30, pr3b.y))), r3 = rsq(d3);
  const double k3 = st.e3 * r3;
  const double k2 = __builtin_fma(-k3, l32, st.e2) * r2;
  const double k1 = __builtin_fma(-k2, l21, __builtin_fma(-k3, l31, st.e1)) * r1;
  const double k0 = __builtin_fma(-k1, l10, __builtin_fma(-k2, l20, __builtin_fma(-k3, l30, st.e0))) * r0;
  double yy[4];
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    const double y = __builtin_fma(bo[rb][1].y, k3, __builtin_fma(bo[rb][1].x, k2, __builtin_fma(bo[rb][0].y, k1, bo[rb][0].x * k0)));
    const int row = 16 * rb + c;
    yy[rb] = y;
    st.Ym[o][rb] = (rb > jb || row > J + 3) ? y : 0.0;
    st.nY[o][rb] = -st.Ym[o][rb];
  }
  // hand (Y, k) to the inverse wave and to the store wave: buffer o was last read for step s - 2
  if constexpr (s >= 2) {
    if (cons1 < base + s - 1) diag_flag_wait(flags + 1, base + s - 1);      // (uniform; the slow path, rarely taken)
    if (cons2 < base + s - 1) diag_flag_wait(flags + 2, base + s - 1);
  }
  double* xo = xch + o * 512 + l;
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) xo[rb * 64] = yy[rb];
  xo[4 * 64] = k0; xo[5 * 64] = k1; xo[6 * 64] = k2; xo[7 * 64] = k3;
  diag_flag_set_inorder(flags, base + s + 1, l);
  diagf_mfmas<s, 0>(st); diagf_mfmas<s, 1>(st);
  __builtin_amdgcn_sched_barrier(0);
  if (__builtin_amdgcn_ballot_w64(!(d0 > 0.0) | !(d1 > 0.0) | !(d2v > 0.0) | !(d3 > 0.0)) != 0ull) {
    const unsigned m4 = (!(d0 > 0.0) ? 1u : 0u) | (!(d1 > 0.0) ? 2u : 0u) | (!(d2v > 0.0) ? 4u : 0u) | (!(d3 > 0.0) ? 8u : 0u);
    st.badmask |= (unsigned long long)__builtin_amdgcn_readfirstlane(m4) << J;
  }
}
template <int s>
__device__ __forceinline__ void diagw_step(DiagWState& st, double* __restrict__ Wt, const double* __restrict__ xch, int* __restrict__ flags,
                                           int base, int c, int g, int l) {
  constexpr int jb = s >> 2, q = s & 3, o = s & 1;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb) Wt[(16 * cb + c) * 4 + g] = st.V[jb][cb][q];
  d2 wo[4][2];
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb) {
    wo[cb][0] = *reinterpret_cast<const d2*>(&Wt[(16 * cb + c) * 4]);
    wo[cb][1] = *reinterpret_cast<const d2*>(&Wt[(16 * cb + c) * 4 + 2]);
  }
  diag_flag_wait(flags, base + s + 1);
  const double* xo = xch + o * 512 + l;
  double nY[4];
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {                                // Y of the step: rows of and above the pivot block take no part in the update
    const double y = xo[rb * 64];
    nY[rb] = -((rb > jb || 16 * rb + c > 4 * s + 3) ? y : 0.0);
  }
  const double k0 = xo[4 * 64], k1 = xo[5 * 64], k2 = xo[6 * 64], k3 = xo[7 * 64];
  diag_flag_set(flags + 1, base + s + 1, l);                       // (its release waits for the reads above)
  double Z[4];
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb)
    Z[cb] = __builtin_fma(wo[cb][1].y, k3, __builtin_fma(wo[cb][1].x, k2, __builtin_fma(wo[cb][0].y, k1, wo[cb][0].x * k0)));
#pragma unroll
  for (int cb = 0; cb <= jb; ++cb) st.V[jb][cb][q] = Z[cb];
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    if (rb == jb && q == 3) continue;
#pragma unroll
    for (int cb = 0; cb <= jb; ++cb) st.V[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(nY[rb], Z[cb], st.V[rb][cb], 0, 0, 0);
  }
  __builtin_amdgcn_sched_barrier(0);
}
// store wave: the finished L entries of step s -- L[row][J + g] = Y of lane (g, c) of row block rb, final for row >= J + g -- leave for
// global memory from the exchange area (four columns x sixteen consecutive rows per store instruction: four 128-byte segments)
template <int s, typename TS>
__device__ __forceinline__ void diags_step(const double* __restrict__ xch, int* __restrict__ flags, int base, int c, int g, int l,
                                           void* __restrict__ Ag, size_t offAg, int ldg) {
  constexpr int J = 4 * s, jb = s >> 2, o = s & 1;
  diag_flag_wait(flags, base + s + 1);
  const double* xo = xch + o * 512 + l;
  double y[4];
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) y[rb] = xo[rb * 64];
  diag_flag_set(flags + 2, base + s + 1, l);
#pragma unroll
  for (int rb = jb; rb < 4; ++rb) {
    const int row = 16 * rb + c;
    if (row >= J + g) MatIO<TS>::st1(Ag, offAg + (size_t)(J + g) * ldg + row, y[rb]);
  }
}
template <int s, typename TS>
__device__ __forceinline__ void diags_steps(const double* xch, int* flags, int base, int c, int g, int l, void* __restrict__ Ag, size_t offAg, int ldg) {
  if constexpr (s < 16) { diags_step<s, TS>(xch, flags, base, c, g, l, Ag, offAg, ldg); diags_steps<s + 1, TS>(xch, flags, base, c, g, l, Ag, offAg, ldg); }
}
template <int s, typename TS>
__device__ __forceinline__ void diagf_steps(DiagFState& st, double* Sp, double* xch, int* flags, int base, int c, int g, int l,
                                            void* __restrict__ Ag, size_t offAg, int ldg) {
  if constexpr (s < 16) { diagf_step<s, TS>(st, Sp, xch, flags, base, c, g, l, Ag, offAg, ldg); diagf_steps<s + 1, TS>(st, Sp, xch, flags, base, c, g, l, Ag, offAg, ldg); }
}
template <int s>
__device__ __forceinline__ void diagw_steps(DiagWState& st, double* Wt, const double* xch, int* flags, int base, int c, int g, int l) {
  if constexpr (s < 16) { diagw_step<s>(st, Wt, xch, flags, base, c, g, l); diagw_steps<s + 1>(st, Wt, xch, flags, base, c, g, l); }
}
// Waves 0, 1 and 2 of the workgroup call this (w = wave index, uniform: 0 factor, 1 inverse, 2 stores of L); wave 3 does not.  work: 64 * DIAG_SP + 256 + DIAG_XCH
// doubles of LDS (Sp, Wt, exchange area); Wl: the W image (MAY overlap `work`: it is written after the last step, by wave 1, which
// finishes behind wave 0); src as in diag64m_wave.  Returns the mask of non-positive pivots in wave 0 (0 in wave 1).  The caller
// follows with a workgroup barrier before anybody reads Wl.
constexpr int DIAG_PAIR_WORK = 64 * DIAG_SP + 256 + DIAG_XCH;
template <typename TS>
__device__ __forceinline__ unsigned long long diag64_pair(void* __restrict__ A, size_t offA, int ld, double* work, double* Wl, int* flags, int base,
                                                          int w, int l, const double* src = nullptr) {
  const int c = l & 15, g = l >> 4;
  double* Sp = work;
  double* Wt = work + 64 * DIAG_SP;
  double* xch = Wt + 256;
  if (w == 0) {
    DiagFState st;
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * rb + 4 * r + g, col = 16 * cb + c;
          st.S[rb][cb][r] = (row >= col) ? (src ? src[col * DIAG_LS + row] : MatIO<TS>::ld1(A, offA + (size_t)col * ld + row)) : 0.0;
        }
    st.e0 = g == 0 ? 1.0 : 0.0; st.e1 = g == 1 ? 1.0 : 0.0; st.e2 = g == 2 ? 1.0 : 0.0; st.e3 = g == 3 ? 1.0 : 0.0;
    st.badmask = 0ull;
    diagf_steps<0, TS>(st, Sp, xch, flags, base, c, g, l, A, offA, ld);
    return st.badmask;
  }
  if (w == 2) { diags_steps<0, TS>(xch, flags, base, c, g, l, A, offA, ld); return 0ull; }
  DiagWState st;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) st.V[rb][cb][r] = (16 * rb + 4 * r + g == 16 * cb + c) ? 1.0 : 0.0;
  diagw_steps<0>(st, Wt, xch, flags, base, c, g, l);
  __builtin_amdgcn_sched_barrier(0);
  // Wl may overlap the exchange area (leaf128 does that): the store wave must have read its last buffer before the image is written
  // (the factor wave is done with Sp by then: it published step 15 before either consumer could finish).  Found by tools/stress_region.py:
  // without this wait the last columns of L came out of a clobbered buffer once in ~10 evaluations of the panel path.
  diag_flag_wait(flags + 2, base + 16);
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Wl[(16 * cb + c) * DIAG_LS + 16 * rb + 4 * r + g] = cb <= rb ? st.V[rb][cb][r] : 0.0;
  return 0ull;
}

// first non-positive pivot among the real columns gcol0 .. of a 64-pivot mask -> LAPACK-style info (0: none)
__device__ __forceinline__ int diag_info_of(unsigned long long badmask, int gcol0, int n_real) {
  const int lim = n_real - gcol0;                                  // pivots k < lim are real columns
  const unsigned long long m = lim >= 64 ? badmask : (lim <= 0 ? 0ull : (badmask & ((1ull << lim) - 1ull)));
  return m != 0ull ? gcol0 + __builtin_ctzll(m) + 1 : 0;
}

template <typename TS>
__global__ __launch_bounds__(64) void diag64m_kernel(BatchPtr Ab, size_t offA, int ld, BatchPtr Wb, size_t offW,
                                                     int gcol0, int n_real, BatchInfo infob) {
  void* __restrict__ A = Ab.p[blockIdx.x];
  void* __restrict__ W = Wb.p[blockIdx.x];
  int* __restrict__ info = infob.p[blockIdx.x];
  constexpr int LS = DIAG_LS;
  __shared__ __attribute__((aligned(16))) double Sp[64 * DIAG_SP];  // current block column of S: Sp[row][16]
  __shared__ __attribute__((aligned(16))) double Wt[64 * 4];        // pivot rows of the W part, transposed: Wt[col][k]
  __shared__ __attribute__((aligned(16))) double Lo[64 * LS];       // finished L entries: Lo[col][row]
  __shared__ __attribute__((aligned(16))) double Wl[64 * LS];       // W for the coalesced store: Wl[col][row]
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  Diag64mState st;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb <= rb; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * rb + 4 * r + g, col = 16 * cb + c;
        st.S[rb][cb][r] = (row >= col) ? MatIO<TS>::ld1(A, offA + (size_t)col * ld + row) : 0.0;
        st.V[rb][cb][r] = (row == col) ? 1.0 : 0.0;
      }
  st.e0 = g == 0 ? 1.0 : 0.0; st.e1 = g == 1 ? 1.0 : 0.0; st.e2 = g == 2 ? 1.0 : 0.0; st.e3 = g == 3 ? 1.0 : 0.0;
  st.badmask = 0ull;
  diag64m_steps<0>(st, Sp, Wt, Lo, c, g);
  __builtin_amdgcn_sched_barrier(0);
  if (l == 0) {
    const int lim = n_real - gcol0;                                // pivots k < lim are real columns
    const unsigned long long m = lim >= 64 ? st.badmask : (lim <= 0 ? 0ull : (st.badmask & ((1ull << lim) - 1ull)));
    if (m != 0ull) atomicCAS(info, 0, gcol0 + __builtin_ctzll(m) + 1);
  }
  // epilogue.  W: the register blocks go through LDS (blocks above the diagonal as zeros) and leave as 16-byte-per-lane row
  // segments, two columns per store instruction.  L (lower triangle only -- the upper triangle of A is never written): column c
  // has 64 - c entries, so columns c and 64 - c together fill exactly one wave; no execution masks, one 8-byte store per lane.
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) Wl[(16 * cb + c) * LS + 16 * rb + 4 * r + g] = cb <= rb ? st.V[rb][cb][r] : 0.0;
  const int r2 = 2 * (l & 31), ch = l >> 5;
#pragma unroll
  for (int it0 = 0; it0 < 32; it0 += 8) {
    d2 wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wv[i] = *reinterpret_cast<const d2*>(&Wl[(2 * (it0 + i) + ch) * LS + r2]);
#pragma unroll
    for (int i = 0; i < 8; ++i) MatIO<TS>::st2(W, offW + (size_t)(2 * (it0 + i) + ch) * 64 + r2, wv[i]);
  }
  MatIO<TS>::st1(A, offA + l, Lo[l]);                                        // column 0: all 64 rows
  if (l >= 32) MatIO<TS>::st1(A, offA + (size_t)32 * ld + l, Lo[32 * LS + l]);   // column 32: rows 32..63
#pragma unroll
  for (int c0 = 1; c0 < 32; c0 += 8) {
    double lv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (c0 + i < 32) {
        const int ca = c0 + i, cz = 64 - ca;                                 // lanes 0 .. 63-ca: column ca, rows ca .. 63; the rest: column cz, rows cz .. 63
        const bool first = l < 64 - ca;
        lv[i] = Lo[(first ? ca : cz) * LS + (first ? ca + l : l)];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (c0 + i < 32) {
        const int ca = c0 + i, cz = 64 - ca;
        const bool first = l < 64 - ca;
        MatIO<TS>::st1(A, offA + (size_t)(first ? ca : cz) * ld + (first ? ca + l : l), lv[i]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K2b: C (M x N) {-=, =} A (M x K) * B (N x K)^T  (column-major; M, N multiples of 64, K of 16), batched over up to
// LMM_MAX_BATCH independent matrices (blockIdx.y), on v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction).
// Round 1's update kernel; what is still product code are its SHORT instantiations, which are latency- not throughput-bound:
//   gemm44_kernel<64, true>   the 64-column solve by a stored inverse diagonal block (trsm_rec leaves, round-2 factorisation path)
//   gemm44_kernel<64, false>  the 64-column trailing update of the round-2 path
//   gemm44_kernel<128, true>  C = X X' on the upper-triangular operand of the explicit inverse (gradient paths; kfrom_row)
// (the wide update <128, false>, its LDS-flag variant and gemm16_kernel moved to tools/retired_kernels.hip in round 4.)
//   128 x BN block tile, 4 waves (2 x 2), wave tile 64 x BN/2, BK = 16 k-columns per LDS stage, two stages, one
//   barrier per stage.  LDS image is k-major ([k][row], row stride BM+16 doubles): a k-column of the tile is one
//   contiguous 1-KiB global segment (coalesced 16-B loads) and an operand read is one conflict-free ds_read_b64.
//   lower != 0: only tiles on/below the diagonal of the (common-origin) region (SYRK on the lower triangle).  Measured lane map (tools/mfma_map):
//     A[blk][i][k] : lane 16k + 4 blk + i      B[blk][k][j] : lane 16k + 4 blk + j      D[blk][i][j] : lane 16i + 4 blk + j
// A wave's 64 x WN tile is covered by fragments  fa[u] (rows 16u + (lane&15), k = lane>>4 -- the same LDS read as the
// 16x16x4 operand) and fb[v][s] (columns 16v + ((lane&15) + 4s) mod 16: the 4-row groups rotated by s), so that block
// blk of MFMA (u, v, s) is the 4x4 product of row group blk with column group (blk + s) mod 4: 4 rotations cover a
// 16 x 16 tile.  Per k-step of 4: 4 + 4 TN ds_read_b64 feed 16 TN MFMAs (16 cycles each).
// ---------------------------------------------------------------------------------------------------
#include "lmm_work_item.h"

template <int BN, bool SET>
__global__ __launch_bounds__(256, 2) void gemm44_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
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
  constexpr int NLA = (BM * BK / 2) / 256;     // 4: thread t stages rows 2(t%64).. of k-columns t/64 + 4q
  constexpr int NLB = (BN * BK / 2) / 256;     // 4 (BN=128) or 2 (BN=64)
  constexpr int KSB = 256 / (BN / 2);          // k-columns covered per pass of the B staging (4 or 8)
  constexpr int STAGE = 2 * BK * SA + 2 * BK * SB, EPI = 4 * 32 * 65;
  __shared__ __attribute__((aligned(16))) double smem[STAGE > EPI ? STAGE : EPI];   // staging, then epilogue transpose
  double (*As)[BK * SA] = reinterpret_cast<double (*)[BK * SA]>(smem);
  double (*Bs)[BK * SB] = reinterpret_cast<double (*)[BK * SB]>(smem + 2 * BK * SA);

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
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      const int ktl = kt + 1;
      const double* pa = ga0 + (size_t)ktl * BK * lda;
      const double* pb = gb0 + (size_t)ktl * BK * ldb;
#pragma unroll
      for (int q = 0; q < NLA; ++q) ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(4 * q) * lda);
#pragma unroll
      for (int q = 0; q < NLB; ++q) rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(KSB * q) * ldb);
    }
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
}

// ---------------------------------------------------------------------------------------------------
// K2b, third form (round 2): ONE workgroup per CU (one wave per SIMD, up to 256 VGPRs), software-pipelined by hand the way the
// vendor's gfx950 dgemm kernels are (rocBLAS reaches 76.8 TFLOP/s on this box, tools/yardstick): v_mfma_f64_16x16x4 lasts 64
// cycles, so a wave that keeps one MFMA in flight can spend the next ~60 cycles on LDS reads, LDS writes, global loads,
// waits and even the workgroup barrier without starving the matrix pipe -- no second workgroup is needed to cover them.
//   per k-tile (BK = 16 = four k-steps of 16 MFMAs):
//     k-step 0..2 : 16 MFMAs each, interleaved one-for-one with the 8 fragment reads of the NEXT k-step, the 8 ds_write_b128 of
//                   tile t+1 (its global loads were issued a whole tile earlier) and the 8 global loads of tile t+2
//     k-step 3    : 10 MFMAs, s_barrier (tile t+1 is complete in the other LDS buffer, everybody is done reading this one),
//                   6 MFMAs interleaved with the 8 fragment reads of tile t+1's k-step 0
//   __builtin_amdgcn_sched_group_barrier pins the interleave (hipcc otherwise clusters the loads ahead of the MFMAs).
// ---------------------------------------------------------------------------------------------------
#define LMM_MFMA16(SET, V, U) acc[V][U] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[SET][V], fa[SET][U], acc[V][U], 0, 0, 0)
#define LMM_MFMA16_ALL(SET)                                                                                   \
    LMM_MFMA16(SET, 0, 0); LMM_MFMA16(SET, 0, 1); LMM_MFMA16(SET, 0, 2); LMM_MFMA16(SET, 0, 3);               \
    LMM_MFMA16(SET, 1, 3); LMM_MFMA16(SET, 1, 2); LMM_MFMA16(SET, 1, 1); LMM_MFMA16(SET, 1, 0);               \
    LMM_MFMA16(SET, 2, 0); LMM_MFMA16(SET, 2, 1); LMM_MFMA16(SET, 2, 2); LMM_MFMA16(SET, 2, 3);               \
    LMM_MFMA16(SET, 3, 3); LMM_MFMA16(SET, 3, 2); LMM_MFMA16(SET, 3, 1); LMM_MFMA16(SET, 3, 0)
// One k-tile of the pipeline.  RA / RB: the staging registers that hold tile t+1 on entry; they are written to the other LDS
// buffer and immediately reloaded with tile `KLOAD` (each global load one instruction behind the ds_write that frees its
// register), so a load has a whole tile (DEPTH 1) or two tiles (DEPTH 2, two register sets) to arrive.
#define LMM_TILE_BODY(RA, RB, KLOAD)                                                                                              \
  {                                                                                                                               \
    const double* pa = ga0 + (size_t)(KLOAD) * BK * lda;                                                                          \
    const double* pb = gb0 + (size_t)(KLOAD) * BK * ldb;                                                                          \
    /* k-step 0: MFMAs on set 0; reads of k-step 1 into set 1; A half: ds_write of tile t+1, reload */                             \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { fa[1][u] = as[offA + 4 * SA + 16 * u]; fb[1][u] = bs[offB + 4 * SB + 16 * u]; } \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                               \
      *reinterpret_cast<d2*>(&asn[sa0 + 4 * q * SA]) = RA[q];                                                                     \
      RA[q] = *reinterpret_cast<const d2*>(pa + (size_t)(4 * q) * lda);                                                           \
    }                                                                                                                             \
    LMM_MFMA16_ALL(0);                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x008, 1); LMM_SGB(0x020, 1); } \
    /* k-step 1: MFMAs on set 1; reads of k-step 2 into set 0; B half: ds_write, reload */                                        \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { fa[0][u] = as[offA + 8 * SA + 16 * u]; fb[0][u] = bs[offB + 8 * SB + 16 * u]; } \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                               \
      *reinterpret_cast<d2*>(&bsn[sb0 + 4 * q * SB]) = RB[q];                                                                     \
      RB[q] = *reinterpret_cast<const d2*>(pb + (size_t)(4 * q) * ldb);                                                           \
    }                                                                                                                             \
    LMM_MFMA16_ALL(1);                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x008, 1); LMM_SGB(0x020, 1); } \
    /* k-step 2: MFMAs on set 0; reads of k-step 3 into set 1 */                                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { fa[1][u] = as[offA + 12 * SA + 16 * u]; fb[1][u] = bs[offB + 12 * SB + 16 * u]; } \
    LMM_MFMA16_ALL(0);                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }                                       \
    LMM_SGB(0x008, 8);                                                                                                            \
    /* k-step 3, first part: 10 MFMAs on set 1, then the barrier */                                                               \
    LMM_MFMA16(1, 0, 0); LMM_MFMA16(1, 0, 1); LMM_MFMA16(1, 0, 2); LMM_MFMA16(1, 0, 3);                                           \
    LMM_MFMA16(1, 1, 3); LMM_MFMA16(1, 1, 2); LMM_MFMA16(1, 1, 1); LMM_MFMA16(1, 1, 0);                                           \
    LMM_MFMA16(1, 2, 0); LMM_MFMA16(1, 2, 1);                                                                                     \
    __syncthreads();                                                                                                 \
    /* k-step 3, second part: 6 MFMAs on set 1, interleaved with the reads of tile t+1's k-step 0 into set 0 */                   \
    _Pragma("unroll") for (int u = 0; u < 4; ++u) { fa[0][u] = asn[offA + 16 * u]; fb[0][u] = bsn[offB + 16 * u]; }               \
    LMM_MFMA16(1, 2, 2); LMM_MFMA16(1, 2, 3);                                                                                     \
    LMM_MFMA16(1, 3, 3); LMM_MFMA16(1, 3, 2); LMM_MFMA16(1, 3, 1); LMM_MFMA16(1, 3, 0);                                           \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }                                       \
    LMM_SGB(0x100, 2);                                                                                                            \
  }

template <int DEPTH>
__global__ __launch_bounds__(256, 1) void gemm16p_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
                                                          BatchPtr Bb, size_t goffB, int ldb,
                                                          int M, int N, int K, int lower, int MT, int full_items,
                                                          int splitk, int kfrom_row) {
  double* C = Cb.p[blockIdx.y] + goffC;
  const double* A = Ab.p[blockIdx.y] + goffA;
  const double* B = Bb.p[blockIdx.y] + goffB;
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int SA = BM + 16, SB = BN + 16;
  __shared__ __attribute__((aligned(16))) double As[2][BK * SA];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * SB];

  int part = 0, nparts = 1, tj = 0, ti = 0;
  gemm_work_item(BM, BN, N, lower, MT, full_items, splitk, part, nparts, ti, tj);
  const int bm = ti * BM, bn = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * 64;
  const bool active = (bm + wr < M) && (bn + wc < N) && !(lower && bm + wr + 63 < bn + wc);
  const int nk_all = K / BK;
  int kc0 = (int)((long long)nk_all * part / nparts);
  const int kc1 = (int)((long long)nk_all * (part + 1) / nparts);
  if (kfrom_row && kc0 < bm / BK) kc0 = bm / BK;
  A += (size_t)kc0 * BK * lda;
  B += (size_t)kc0 * BK * ldb;
  const int nk = kc1 - kc0;

  int rowa = bm + 2 * (t & 63); if (rowa > M - 2) rowa = M - 2;
  int rowb = bn + 2 * (t & 63); if (rowb > N - 2) rowb = N - 2;
  const double* ga0 = A + (size_t)(t >> 6) * lda + rowa;       // thread t stages rows 2(t%64).. of k-columns t/64 + 4q
  const double* gb0 = B + (size_t)(t >> 6) * ldb + rowb;
  const int sa0 = (t >> 6) * SA + 2 * (t & 63);
  const int sb0 = (t >> 6) * SB + 2 * (t & 63);
  d2 ra[4], rb[4], ra2[DEPTH == 2 ? 4 : 1], rb2[DEPTH == 2 ? 4 : 1];
#pragma unroll
  for (int q = 0; q < 4; ++q) { ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(4 * q) * lda); rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(4 * q) * ldb); }
#pragma unroll
  for (int q = 0; q < 4; ++q) { *reinterpret_cast<d2*>(&As[0][sa0 + 4 * q * SA]) = ra[q]; *reinterpret_cast<d2*>(&Bs[0][sb0 + 4 * q * SB]) = rb[q]; }
  {                                                  // tile 1 (and, DEPTH 2, tile 2) into registers; clamped indices reload valid tiles
    const int k1 = nk > 1 ? 1 : 0, k2 = nk > 2 ? 2 : nk - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)k1 * BK * lda + (size_t)(4 * q) * lda);
      rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)k1 * BK * ldb + (size_t)(4 * q) * ldb);
      if (DEPTH == 2) {
        ra2[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)k2 * BK * lda + (size_t)(4 * q) * lda);
        rb2[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)k2 * BK * ldb + (size_t)(4 * q) * ldb);
      }
    }
  }
  __syncthreads();

  d4 acc[4][4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + wr + l15, offB = lk * SB + wc + l15;
  double fa[2][4], fb[2][4];                          // two fragment sets: the k-step being multiplied and the next one
#pragma unroll
  for (int u = 0; u < 4; ++u) { fa[0][u] = As[0][offA + 16 * u]; fb[0][u] = Bs[0][offB + 16 * u]; }

  if (DEPTH == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const double* as = &As[buf][0];
      const double* bs = &Bs[buf][0];
      double* asn = &As[buf ^ 1][0];
      double* bsn = &Bs[buf ^ 1][0];
      const int kn = (kt + 2 < nk) ? kt + 2 : nk - 1;              // clamped: the surplus loads / writes of the last tiles are unused
      LMM_TILE_BODY(ra, rb, kn)
    }
  } else {
    for (int kt = 0; kt < nk; kt += 2) {                           // two tiles per trip: the register sets alternate statically
      {
        const double* as = &As[0][0]; const double* bs = &Bs[0][0];
        double* asn = &As[1][0]; double* bsn = &Bs[1][0];
        const int kn = (kt + 3 < nk) ? kt + 3 : nk - 1;
        LMM_TILE_BODY(ra, rb, kn)
      }
      if (kt + 1 < nk) {
        const double* as = &As[1][0]; const double* bs = &Bs[1][0];
        double* asn = &As[0][0]; double* bsn = &Bs[0][0];
        const int kn = (kt + 4 < nk) ? kt + 4 : nk - 1;
        LMM_TILE_BODY(ra2, rb2, kn)
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    double* cpv = C + (size_t)(bn + wc + 16 * v + lk) * ldc + bm + wr + l15;
    if (nparts == 1) {
      double cv[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * ldc + 16 * u];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * ldc + 16 * u] = cv[u][r] - acc[v][u][r];
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) unsafeAtomicAdd(cpv + (size_t)(4 * r) * ldc + 16 * u, -acc[v][u][r]);
    }
  }
}

// (the direct-to-LDS variant of this kernel, gemm16d_kernel, measured slower; it lives in profiles/r02/gemm16d_direct_to_lds_rejected.hip.txt)

// Half-height companion of gemm16p_kernel for the RAGGED last 64 rows of an update (factor matrices carry 64 rider rows, so the
// row count of every trailing update is an odd multiple of 64: a 128-row tile there has two idle waves for a whole tile time,
// ~1.5 % of the update).  64 x 128 block tile, 4 waves side by side (64 x 32 each: 4 x 2 MFMA blocks), same LDS image and
// pipeline; launched behind the main grid on the same stream for the rows [M - 64, M) when they lie below every column.
#define LMM_MFMA16H_ALL(SET)                                                                                  \
    LMM_MFMA16(SET, 0, 0); LMM_MFMA16(SET, 0, 1); LMM_MFMA16(SET, 0, 2); LMM_MFMA16(SET, 0, 3);               \
    LMM_MFMA16(SET, 1, 3); LMM_MFMA16(SET, 1, 2); LMM_MFMA16(SET, 1, 1); LMM_MFMA16(SET, 1, 0)
// STRIP = true: the ragged strip (blockIdx.x = column tile; A, C already point at its 64 rows).  STRIP = false (round 2): ANY 64-row
// tile of a lower-trapezoid update -- blockIdx.x enumerates the tiles (ti64, tj128) with 64 ti + 63 >= 128 tj column tile by column
// tile; used for launches whose 128 x 128 tiles would leave CUs idle (few latents per GPU, low recursion levels): twice the
// workgroups, half the time per workgroup.
template <bool STRIP>
__global__ __launch_bounds__(256, 2) void gemm16h_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
                                                          BatchPtr Bb, size_t goffB, int ldb, int N, int K, int MT64) {
  double* C = Cb.p[blockIdx.y] + goffC;              // STRIP: row 0 = first of the 64 rows
  const double* A = Ab.p[blockIdx.y] + goffA;
  const double* B = Bb.p[blockIdx.y] + goffB;
  constexpr int BM = 64, BN = 128, BK = 16;
  constexpr int SA = BM + 16, SB = BN + 16;
  __shared__ __attribute__((aligned(16))) double As[2][BK * SA];
  __shared__ __attribute__((aligned(16))) double Bs[2][BK * SB];
  int tjx = blockIdx.x, ti64 = 0;
  if (!STRIP) {                                      // column tile tj holds the row tiles 2 tj .. MT64 - 1
    int rem = blockIdx.x; tjx = 0;
    while (rem >= MT64 - 2 * tjx) { rem -= MT64 - 2 * tjx; ++tjx; }
    ti64 = 2 * tjx + rem;
    A += (size_t)64 * ti64; C += (size_t)64 * ti64;
  }
  const int bn = tjx * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wc = w * 32;
  const int nk = K / BK;
  // staging: A tile 64 x 16 = 512 d2 (2 per thread: rows 2(t%32).., k-columns t/32 + 8q); B tile 128 x 16 (4 per thread)
  int rowb = bn + 2 * (t & 63); if (rowb > N - 2) rowb = N - 2;
  const double* ga0 = A + (size_t)(t >> 5) * lda + 2 * (t & 31);
  const double* gb0 = B + (size_t)(t >> 6) * ldb + rowb;
  const int sa0 = (t >> 5) * SA + 2 * (t & 31);
  const int sb0 = (t >> 6) * SB + 2 * (t & 63);
  d2 ra[2], rb[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(8 * q) * lda);
#pragma unroll
  for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(4 * q) * ldb);
#pragma unroll
  for (int q = 0; q < 2; ++q) *reinterpret_cast<d2*>(&As[0][sa0 + 8 * q * SA]) = ra[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<d2*>(&Bs[0][sb0 + 4 * q * SB]) = rb[q];
  {
    const int k1 = nk > 1 ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)k1 * BK * lda + (size_t)(8 * q) * lda);
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)k1 * BK * ldb + (size_t)(4 * q) * ldb);
  }
  __syncthreads();
  d4 acc[2][4];
#pragma unroll
  for (int v = 0; v < 2; ++v)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + l15, offB = lk * SB + wc + l15;
  double fa[2][4], fb[2][2];
#pragma unroll
  for (int u = 0; u < 4; ++u) fa[0][u] = As[0][offA + 16 * u];
#pragma unroll
  for (int v = 0; v < 2; ++v) fb[0][v] = Bs[0][offB + 16 * v];
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const double* as = &As[buf][0];
    const double* bs = &Bs[buf][0];
    double* asn = &As[buf ^ 1][0];
    double* bsn = &Bs[buf ^ 1][0];
    const int kn = (kt + 2 < nk) ? kt + 2 : nk - 1;
    const double* pa = ga0 + (size_t)kn * BK * lda;
    const double* pb = gb0 + (size_t)kn * BK * ldb;
    // k-step 0: A half of the staging
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[1][u] = as[offA + 4 * SA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[1][v] = bs[offB + 4 * SB + 16 * v];
#pragma unroll
    for (int q = 0; q < 2; ++q) { *reinterpret_cast<d2*>(&asn[sa0 + 8 * q * SA]) = ra[q]; ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(8 * q) * lda); }
    LMM_MFMA16H_ALL(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
    LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1); LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
    // k-step 1 and 2: B half of the staging, two pieces each
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[0][u] = as[offA + 8 * SA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[0][v] = bs[offB + 8 * SB + 16 * v];
#pragma unroll
    for (int q = 0; q < 2; ++q) { *reinterpret_cast<d2*>(&bsn[sb0 + 4 * q * SB]) = rb[q]; rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(4 * q) * ldb); }
    LMM_MFMA16H_ALL(1);
#pragma unroll
    for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
    LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1); LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[1][u] = as[offA + 12 * SA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[1][v] = bs[offB + 12 * SB + 16 * v];
#pragma unroll
    for (int q = 2; q < 4; ++q) { *reinterpret_cast<d2*>(&bsn[sb0 + 4 * q * SB]) = rb[q]; rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(4 * q) * ldb); }
    LMM_MFMA16H_ALL(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
    LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1); LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
    // k-step 3: 5 MFMAs, barrier, 3 MFMAs covering the first reads of tile t+1
    LMM_MFMA16(1, 0, 0); LMM_MFMA16(1, 0, 1); LMM_MFMA16(1, 0, 2); LMM_MFMA16(1, 0, 3); LMM_MFMA16(1, 1, 3);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[0][u] = asn[offA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[0][v] = bsn[offB + 16 * v];
    LMM_MFMA16(1, 1, 2); LMM_MFMA16(1, 1, 1); LMM_MFMA16(1, 1, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 2); }
  }
  if (bn + wc >= N) return;
  if (!STRIP && 64 * ti64 + 63 < bn + wc) return;     // this wave's 64 x 32 piece lies wholly above the diagonal
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    double* cpv = C + (size_t)(bn + wc + 16 * v + lk) * ldc + l15;
    double cv[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * ldc + 16 * u];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * ldc + 16 * u] = cv[u][r] - acc[v][u][r];
  }
}
// ---------------------------------------------------------------------------------------------------
// K2c (round 3): the 128-column PANEL as part of the update launch that precedes it.
// Until round 2 every 128 columns cost five latency-bound launches behind each trailing update (diag64, TRSM, 64-column update,
// diag64, TRSM: 11.5 ms of the 96 ms a rank's 4-latent share of the C2 job takes, all of it serial).  Now:
//   leaf128 (device function, one workgroup): the 128 x 128 diagonal block D = [A11 .; A21 A22] -> L (in place), the two 64 x 64
//       inverse blocks W11, W22 (what the solves downstream use) AND the full inverse Dinv = [W11 0; W21 W22],
//       W21 = -W22 L21 W11, into a 128 x 128 panel of the W2 scratch:
//         A  wave 0: diag64m on A11 (L11 straight to global, W11 -> LDS X)   | waves 1-3: A21 -> LDS Y
//         B  all   : L21 = A21 W11'  -> global and Y;   T = L21 W11 -> X;   A22 -= L21 L21' -> global
//         C  wave 0: diag64m on A22' (L22 to global, W22 -> LDS Y)
//         D  all   : W21 = -W22 T -> W2
//       LDS: X + Y (the work areas Sp / Wt of each diag64m sit inside the image it will overwrite) = 69 632 bytes < the 73 728 of the
//       update's staging buffers, so the node kernel has exactly gemm16p_kernel's footprint (two workgroups per CU).
//   bulk (the rows below the block): X = P Dinv' in place -- ONE pipelined MFMA GEMM with K = 128 (two passes over the panel rows
//       instead of seven).
//   potrf_node_kernel: the trailing update  C -= A B'  of gemm16p_kernel, in which the workgroup that owns the top-left tile of
//       the target region -- the diagonal block of the NEXT panel -- goes on to factor it (leaf128) while the other workgroups of
//       the launch are still updating: for every update of more than one scheduling round the leaf disappears from the critical
//       path.  Column tile 0 of the region is handed out first (items 0 .. MT-1, never split along K); the remaining tiles follow
//       in gemm_work_item's band order with its split-K tail.  mode NODE_BULK: the same kernel runs the bulk tiles.
// ---------------------------------------------------------------------------------------------------
// D (64 x 64) = A R on four waves (wave w: rows 32 (w & 1).., columns 32 (w >> 1)..), operands in LDS with element strides
// (A[i][k] = As[i sai + k sak], R[k][j] = Rs[k srk + j srj]); emit(u, v, r, row, col, value) receives every entry once, (u, v, r)
// being compile-time after unrolling (so that callers can keep per-entry registers).
// acc += A R over K = 64 nblk (operand pointers advance with sak / srk per k), in chunks of 32 k: the 32 operand values a lane needs
// for a chunk are loaded (independent loads -- the operands may sit in global memory) while the 32 MFMAs of earlier chunks run.
// DEEP: two chunks in flight ahead of the one being multiplied (three register sets; for the one-workgroup-per-CU build, which has
// the registers: a global load takes ~2 us, a chunk's MFMAs 0.85 us); else one.
struct MM64Chunk { double fa[8][2], fr[8][2]; };
// A load of data that ANOTHER workgroup of the same launch has published (write-through ST_PUB stores, drained, then a flag): an
// agent-scope relaxed atomic load = `global_load ... sc1`, which bypasses this CU's vector L1 (MI355X_MICROARCH.md, "Valid forms").  With
// EVERY load of handed-off bytes in this form the consumer needs no agent-scope acquire fence after its flag poll -- 1.7 us each
// (buffer_inv sc1 + its wait), twice per 64-column block on the chain walker -> helper -> walker.
__device__ __forceinline__ double ld_pub(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// SC1 bit 0: the A operand is handed-off global data (ld_pub), bit 1: the R operand is
template <int SC1 = 0>
__device__ __forceinline__ void mm64_load(MM64Chunk& b, const double* pa, size_t sai, size_t sak, const double* pr, size_t srk, size_t srj) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
    for (int u = 0; u < 2; ++u) b.fa[ks][u] = (SC1 & 1) ? ld_pub(pa + 16 * u * sai + 4 * ks * sak) : pa[16 * u * sai + 4 * ks * sak];
#pragma unroll
    for (int v = 0; v < 2; ++v) b.fr[ks][v] = (SC1 & 2) ? ld_pub(pr + 4 * ks * srk + 16 * v * srj) : pr[4 * ks * srk + 16 * v * srj];
  }
}
__device__ __forceinline__ void mm64_mul(d4 (&acc)[2][2], const MM64Chunk& b) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(b.fa[ks][u], b.fr[ks][v], acc[u][v], 0, 0, 0);
}
template <bool DEEP = false, int SC1 = 0>
__device__ __forceinline__ void wg_mm64_core(d4 (&acc)[2][2], const double* As, size_t sai, size_t sak, const double* Rs, size_t srk, size_t srj,
                                             int w, int l, int nblk = 1) {
  const int c = l & 15, g = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
  const double* pa = As + (wi + c) * sai + g * sak;
  const double* pr = Rs + g * srk + (wj + c) * srj;
  const int nch = 2 * nblk;
  const size_t da = 32 * sak, dr = 32 * srk;
  if (DEEP && nch > 2) {
    MM64Chunk b0, b1, b2;
    mm64_load<SC1>(b0, pa, sai, sak, pr, srk, srj);
    mm64_load<SC1>(b1, pa + da, sai, sak, pr + dr, srk, srj);
    for (int ch = 0; ch < nch; ch += 3) {
      if (ch + 2 < nch) mm64_load<SC1>(b2, pa + (ch + 2) * da, sai, sak, pr + (ch + 2) * dr, srk, srj);
      mm64_mul(acc, b0);
      if (ch + 1 < nch) {
        if (ch + 3 < nch) mm64_load<SC1>(b0, pa + (ch + 3) * da, sai, sak, pr + (ch + 3) * dr, srk, srj);
        mm64_mul(acc, b1);
      }
      if (ch + 2 < nch) {
        if (ch + 4 < nch) mm64_load<SC1>(b1, pa + (ch + 4) * da, sai, sak, pr + (ch + 4) * dr, srk, srj);
        mm64_mul(acc, b2);
      }
    }
    return;
  }
  MM64Chunk b0, b1;
  mm64_load<SC1>(b0, pa, sai, sak, pr, srk, srj);
  for (int ch = 0; ch < nch; ch += 2) {
    if (ch + 1 < nch) mm64_load<SC1>(b1, pa + (ch + 1) * da, sai, sak, pr + (ch + 1) * dr, srk, srj);
    mm64_mul(acc, b0);
    if (ch + 1 < nch) {
      if (ch + 2 < nch) mm64_load<SC1>(b0, pa + (ch + 2) * da, sai, sak, pr + (ch + 2) * dr, srk, srj);
      mm64_mul(acc, b1);
    }
  }
}

template <typename Emit>
__device__ __forceinline__ void wg_mm64(const double* __restrict__ As, int sai, int sak, const double* __restrict__ Rs, int srk, int srj,
                                        int w, int l, Emit emit) {
  const int c = l & 15, g = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
  d4 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = (d4){0.0, 0.0, 0.0, 0.0};
  // all 64 operand values of the wave first (the reads are independent of the MFMAs), then 64 MFMAs back to back
  double fa[16][2], fr[16][2];
  const double* pa = As + (wi + c) * sai + g * sak;
  const double* pr = Rs + g * srk + (wj + c) * srj;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
    for (int u = 0; u < 2; ++u) fa[ks][u] = pa[16 * u * sai + 4 * ks * sak];
#pragma unroll
    for (int v = 0; v < 2; ++v) fr[ks][v] = pr[4 * ks * srk + 16 * v * srj];
  }
#pragma unroll
  for (int ks = 0; ks < 16; ++ks)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[ks][u], fr[ks][v], acc[u][v], 0, 0, 0);
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) emit(u, v, r, wi + 16 * u + 4 * r + g, wj + 16 * v + c, acc[u][v][r]);
}

// ---- flags of a region launch (per matrix, ints; every value is epoch * 32 + count, so words left by earlier launches never match):
//   [0] abort word (epoch * 32 + 1 when raised)      [1] wk: blocks the WALKER has finished -- count r means W_0 .. W_{r-1} and L[c, c-1], c <= r, are final
//   [2 + r]  trs[r]:  64-column blocks of square row r that its HELPER has solved (count k + 1: L[r, 0 .. k] final), r >= 2
//   [18 + r] upd[r]:  steps whose updates helper r has applied to its two rightmost tiles (r, r-1), (r, r) -- what the walker waits for
//   [34 + j] dinv[j]: the 128 x 128 inverse of panel j is in the W2 scratch (count 0)
// A wait spins on thread 0 (bounded: LMM_REGION_SPIN_TICKS = 4 s of the 100 MHz wall clock, or until another workgroup raised the abort word -- the grid
// always drains), then an agent-scope acquire fence makes the producer's data visible to the whole workgroup.
//   [42 + r] asst[r]: column blocks whose first-half partial product the ASSISTANT of square row r has left in the scratch (count c)
#define REGION_FLAG_INTS (34 + LMM_REGION_MAX_PANELS + 16)
// SLEEP: s_sleep argument between polls (64 clocks each).  1 on the region kernel's chain (a handful of pollers, every 30 ns counts); the
// hundreds of bulk workgroups of a fused node launch poll the same few words and use 16 (~0.5 us), or they slow the leaf they wait for.
// The abort word is epoch-tagged like every other flag (epoch * 32 + 1): a word raised by an EARLIER launch on the same slice of the
// persistent flag array never matches, so a timeout in one launch cannot make a later launch leave its waits early.
// Strict-progress build of potrf_region_kernel.  HIP promises no dispatch order, and the deadlock-freedom argument of the dataflow
// needs one: "task (b, idx) waits only for tasks (b, idx' < idx), which are running or finished" (the walker excepted, see the kernel).
// So a workgroup does not TAKE its task index from blockIdx.x, it takes its TURN with it: claim[b] counts the tasks of matrix b
// (b = blockIdx.x % nb) that have been handed out, in index order, to workgroups that have STARTED.
//   * Normally: workgroup (b, i = blockIdx.x / nb) waits until claim[b] == i and moves it to i + 1 (one compare-and-swap when it is
//     already its turn -- every workgroup but those of the first burst, which sort themselves in ~1-2 us per index while the
//     walker, index 0, goes at once).  It then runs exactly the task it has in the index-order build, on the same CU at the same time:
//     that matters, because the hardware refills freed CU slots strictly in workgroup order (one shader engine after the other, the
//     queue's head waiting for ITS engine: profiles/r05/strict_progress_cost.txt), so WHICH first-burst workgroup runs which role
//     decides when the later ones get in -- arrival-order claims scramble the roles and cost configs[1] 10 %, a rank's share 2 %.
//   * Should its turn not come within LMM_CLAIM_TURN_TICKS (a workgroup with a lower index has not started: the hardware did NOT
//     dispatch in order), or has its index been handed out already, it takes the next free index instead (atomic add).  Either way
//     indices go out in order to started workgroups -- the argument holds in any dispatch order; only the speed assumed one.
// The counters come in two sets used by alternate launches of a stream: the workgroup that claims task (0, 0) zeroes the OTHER set
// for the next launch (which cannot start before this one has ended, nor while the launch before, the other set's last user, is
// running) -- no memset between launches, nothing for the host to track but the parity.
#define LMM_CLAIM_INTS LMM_MAX_BATCH
#define LMM_CLAIM_TURN_TICKS 20000LL          // 200 us of the 100-MHz wall clock
__device__ __forceinline__ bool region_claim(unsigned* claim, unsigned* claim_next, int nb, int ntasks, int b, int& idx, int scramble) {
  __shared__ unsigned claim_sh;
  if (threadIdx.x == 0) {
    // (scramble: the test hook's stand-in for a device that starts the LAST workgroups first -- every index is then asked for by the
    // wrong workgroup, and launches larger than the device go through the time-out path)
    const unsigned own = scramble ? (unsigned)(ntasks - 1 - idx) : (unsigned)idx;
    unsigned* c = claim + b;
    unsigned got = 0xffffffffu;
    const long long t0 = wall_clock64();
    for (;;) {
      unsigned v = own;
      if (__hip_atomic_compare_exchange_strong(c, &v, own + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { got = own; break; }
      if (v > own || wall_clock64() - t0 > LMM_CLAIM_TURN_TICKS) { got = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(2);
    }
    claim_sh = got;
  }
  __syncthreads();
  const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)claim_sh);
  idx = (int)c;
  if (b == 0 && c == 0u && threadIdx.x < LMM_CLAIM_INTS) __hip_atomic_store(claim_next + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return c < (unsigned)ntasks;
}
__device__ __forceinline__ bool region_aborted(const int* abort_word, int epoch) {
  const int v = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (v >> 5) == epoch && (v & 31) != 0;
}
// ACQ = false: no acquire fence -- for consumers whose EVERY load of the handed-off bytes is an sc1 load (ld_pub; one workgroup per CU)
template <int SLEEP = 1, bool ACQ = true>
__device__ __forceinline__ void region_wait_ge(const int* f, int epoch, int need, int* abort_word, int* info) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int polls = 0;
    for (;;) {
      const int v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((v >> 5) == epoch && (v & 31) >= need) break;
      __builtin_amdgcn_s_sleep(SLEEP);
      if ((++polls & 63) == 0) {
        if (region_aborted(abort_word, epoch)) break;
        if (wall_clock64() - t0 > LMM_REGION_SPIN_TICKS) {
          __hip_atomic_store(abort_word, epoch * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicCAS(info, 0, LMM_INFO_SYNC_TIMEOUT);               // surfaces through the host's check of the pivot info word
          break;
        }
      }
    }
  }
  __syncthreads();
  if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// Publishing.  An agent-scope release fence writes back EVERY dirty line of the XCD's L2 -- including the megabytes the row streams
// of the same launch keep producing -- and costs 4-13 us under load (tools/fence_probe, profiles/r03), twice per 64-column block of
// the chain.  So everything the square publishes (L blocks, inverse blocks, the pair tiles) is stored WRITE-THROUGH (ST_PUB: agent-
// scope relaxed atomic stores, `global_store ... sc1`), and publishing only waits for those stores to be acknowledged
// (s_waitcnt vmcnt(0)) before the flag goes out.  Consumers still take an agent-scope ACQUIRE fence after seeing the flag (an L2
// invalidate: 0.03-1.4 us), so their ordinary loads fetch the written-through data.
#define ST_PUB(PTR, VAL) __hip_atomic_store((PTR), (VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ void region_publish(int* f, int epoch, int count) {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // every thread: its write-through stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(f, epoch * 32 + count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// LDS of the node kernel: the update's staging (2 x 2 x 16 x 144 doubles = 73 728 bytes, as gemm16p_kernel) >= the leaf's two 64 x 64
// images X, Y (2 x 64 x 68 doubles); the diagonal-block work areas Sp / Wt live inside whichever image is dead in that phase.
constexpr int LEAF_LDS_DOUBLES = 4 * 16 * 144;
static_assert(2 * 64 * DIAG_LS <= LEAF_LDS_DOUBLES && DIAG_PAIR_WORK <= 64 * DIAG_LS, "leaf128 LDS layout");
// lds: LEAF_LDS_DOUBLES doubles.  A: the matrix (double), offD: element offset of the diagonal block; W: 64 x 64 inverse blocks
// (offW: block of the first 64 columns; the second follows at + 4096); W2p: this panel's 128 x 128 inverse (column-major, ld 128).
// All 256 threads of the workgroup call it; it starts and ends with everything in LDS free for reuse.
// PUB: the panel inverse is stored write-through (ST_PUB), for consumers in the SAME launch (NODE_FUSE bulk items).
template <bool PUB = false>
__device__ __forceinline__ void leaf128_dev(double* __restrict__ lds, double* __restrict__ A, size_t offD, int ld,
                                            double* __restrict__ W, size_t offW, double* __restrict__ W2p, int gcol0, int n_real,
                                            int* __restrict__ info, int* __restrict__ dflags) {
  constexpr int LS = DIAG_LS;
  double* X = lds;
  double* Y = X + 64 * LS;
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const size_t off21 = offD + 64, off22 = offD + (size_t)64 * ld + 64;
  unsigned long long bad1 = 0ull, bad2 = 0ull;
  // A   (work areas of the diagonal-block factorisation inside X, which is dead until W11 lands in it after the last step)
  if (w < 3) {                                                     // waves 0-2: factor, inverse, L stores of A11
    bad1 = diag64_pair<double>(A, offD, ld, X, X, dflags, 0, w, l);
  } else {
    for (int e = t - 192; e < 4096; e += 64) { const int row = e & 63, col = e >> 6; Y[col * LS + row] = A[off21 + (size_t)col * ld + row]; }
  }
  __syncthreads();
  // W11 (X) -> the 64 x 64 inverse block and the top-left block of the panel inverse; zeros into the panel's top-right block
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = t + 256 * i, row = e & 63, col = e >> 6;
    const double v = X[col * LS + row];
    W[offW + (size_t)col * 64 + row] = v;
    if (PUB) { ST_PUB(&W2p[(size_t)col * 128 + row], v); ST_PUB(&W2p[(size_t)(64 + col) * 128 + row], 0.0); }
    else { W2p[(size_t)col * 128 + row] = v; W2p[(size_t)(64 + col) * 128 + row] = 0.0; }
  }
  // B: L21 = A21 W11'   (A[i][k] = Y[k][i]; R[k][j] = W11[j][k] = X[k][j])
  double l21[2][2][4];
  wg_mm64(Y, 1, LS, X, LS, 1, w, l, [&](int u, int v, int r, int, int, double x) { l21[u][v][r] = x; });
  // this lane's entries of A22 (the D layout of wg_mm64), requested before the barrier
  double a22[2][2][4];
  {
    const int c_ = l & 15, g_ = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wi + 16 * u + 4 * r + g_, col = wj + 16 * v + c_;
          a22[u][v][r] = row >= col ? A[off22 + (size_t)col * ld + row] : 0.0;
        }
  }
  __syncthreads();                                               // every wave has read A21 from Y
  {
    const int c_ = l & 15, g_ = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = wi + 16 * u + 4 * r + g_, col = wj + 16 * v + c_;
          Y[col * LS + row] = l21[u][v][r];
          A[off21 + (size_t)col * ld + row] = l21[u][v][r];
        }
  }
  __syncthreads();                                               // Y = L21
  // T = L21 W11   (A[i][k] = Y[k][i]; R[k][j] = W11[k][j] = X[j][k])   and   A22 -= L21 L21'   (R[k][j] = L21[j][k] = Y[k][j])
  double tt[2][2][4];
  wg_mm64(Y, 1, LS, X, 1, LS, w, l, [&](int u, int v, int r, int, int, double x) { tt[u][v][r] = x; });
  wg_mm64(Y, 1, LS, Y, LS, 1, w, l, [&](int u, int v, int r, int row, int col, double x) {
    if (row >= col) A[off22 + (size_t)col * ld + row] = a22[u][v][r] - x;
  });
  __syncthreads();                                               // reads of X (W11) and Y (L21) done; A22' is in global memory
  {
    const int c_ = l & 15, g_ = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(wj + 16 * v + c_) * LS + wi + 16 * u + 4 * r + g_] = tt[u][v][r];
  }
  // C
  if (w < 3) bad2 = diag64_pair<double>(A, off22, ld, Y, Y, dflags, 16, w, l);            // work areas inside Y (L21 is in global memory)
  __syncthreads();                                               // X = T, Y = W22
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = t + 256 * i, row = e & 63, col = e >> 6;
    const double v = Y[col * LS + row];
    W[offW + 4096 + (size_t)col * 64 + row] = v;
    if (PUB) ST_PUB(&W2p[(size_t)(64 + col) * 128 + 64 + row], v); else W2p[(size_t)(64 + col) * 128 + 64 + row] = v;
  }
  // D: W21 = -W22 T   (A[i][k] = W22[i][k] = Y[k][i]; R[k][j] = T[k][j] = X[j][k])
  wg_mm64(Y, 1, LS, X, 1, LS, w, l, [&](int, int, int, int row, int col, double x) {
    if (PUB) ST_PUB(&W2p[(size_t)col * 128 + 64 + row], -x); else W2p[(size_t)col * 128 + 64 + row] = -x;
  });
  if (t == 0) {
    int i = diag_info_of(bad1, gcol0, n_real);
    if (i == 0) i = diag_info_of(bad2, gcol0 + 64, n_real);
    if (i) atomicCAS(info, 0, i);
  }
  __syncthreads();                                               // LDS free again
}

#define NODE_UPDATE 1
#define NODE_LEAF 2
#define NODE_BULK 4
#define NODE_FUSE 8
__global__ __launch_bounds__(256) void leaf128_kernel(BatchPtr Ab, size_t offD, int ld, BatchPtr Wb, size_t offW, BatchPtr W2b, size_t offW2,
                                                      int gcol0, int n_real, BatchInfo infob) {
  extern __shared__ __attribute__((aligned(16))) double node_lds[];
  __shared__ int dflags[3];                                        // hand-off counters of the two-wave diagonal-block factorisation
  if (threadIdx.x == 0) { dflags[0] = 0; dflags[1] = 0; dflags[2] = 0; }
  __syncthreads();
  leaf128_dev(node_lds, Ab.p[blockIdx.x], offD, ld, Wb.p[blockIdx.x], offW, W2b.p[blockIdx.x] + offW2, gcol0, n_real, infob.p[blockIdx.x], dflags);
}

// The 64-row form (gemm16h_kernel's pipeline: 64 x 128 output, 4 waves side by side, 64 x 32 each = 4 x 2 MFMA blocks):
// acc += A (64 rows x 16 nk k-columns) * B' (128 rows likewise); lane (l15, lk) of wave w holds acc[v][u][r] = entry
// (row 16 u + l15, column 32 w + 16 v + 4 r + lk).  LDS: 2 x 16 x 80 + 2 x 16 x 144 doubles.
__device__ __forceinline__ void pipe64_accumulate(d4 (&acc)[2][4], double* __restrict__ lds, const double* __restrict__ A, int lda,
                                                  const double* __restrict__ B, int ldb, int nk = 8) {
  constexpr int BK = 16, SA = 80, SB = 144;
  double (*As)[BK * SA] = reinterpret_cast<double (*)[BK * SA]>(lds);
  double (*Bs)[BK * SB] = reinterpret_cast<double (*)[BK * SB]>(lds + 2 * BK * SA);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wc = w * 32;
  const double* ga0 = A + (size_t)(t >> 5) * lda + 2 * (t & 31);
  const double* gb0 = B + (size_t)(t >> 6) * ldb + 2 * (t & 63);
  const int sa0 = (t >> 5) * SA + 2 * (t & 31);
  const int sb0 = (t >> 6) * SB + 2 * (t & 63);
  d2 ra[2], rb[4];
#pragma unroll
  for (int q = 0; q < 2; ++q) ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(8 * q) * lda);
#pragma unroll
  for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(4 * q) * ldb);
  __syncthreads();                                       // the previous user of the staging buffers is done with them
#pragma unroll
  for (int q = 0; q < 2; ++q) *reinterpret_cast<d2*>(&As[0][sa0 + 8 * q * SA]) = ra[q];
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<d2*>(&Bs[0][sb0 + 4 * q * SB]) = rb[q];
  {
    const int k1 = nk > 1 ? 1 : 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)k1 * BK * lda + (size_t)(8 * q) * lda);
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)k1 * BK * ldb + (size_t)(4 * q) * ldb);
  }
  __syncthreads();
  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + l15, offB = lk * SB + wc + l15;
  double fa[2][4], fb[2][2];
#pragma unroll
  for (int u = 0; u < 4; ++u) fa[0][u] = As[0][offA + 16 * u];
#pragma unroll
  for (int v = 0; v < 2; ++v) fb[0][v] = Bs[0][offB + 16 * v];
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const double* as = &As[buf][0];
    const double* bs = &Bs[buf][0];
    double* asn = &As[buf ^ 1][0];
    double* bsn = &Bs[buf ^ 1][0];
    const int kn = (kt + 2 < nk) ? kt + 2 : nk - 1;
    const double* pa = ga0 + (size_t)kn * BK * lda;
    const double* pb = gb0 + (size_t)kn * BK * ldb;
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[1][u] = as[offA + 4 * SA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[1][v] = bs[offB + 4 * SB + 16 * v];
#pragma unroll
    for (int q = 0; q < 2; ++q) { *reinterpret_cast<d2*>(&asn[sa0 + 8 * q * SA]) = ra[q]; ra[q] = *reinterpret_cast<const d2*>(pa + (size_t)(8 * q) * lda); }
    LMM_MFMA16H_ALL(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
    LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1); LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[0][u] = as[offA + 8 * SA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[0][v] = bs[offB + 8 * SB + 16 * v];
#pragma unroll
    for (int q = 0; q < 2; ++q) { *reinterpret_cast<d2*>(&bsn[sb0 + 4 * q * SB]) = rb[q]; rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(4 * q) * ldb); }
    LMM_MFMA16H_ALL(1);
#pragma unroll
    for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
    LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1); LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[1][u] = as[offA + 12 * SA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[1][v] = bs[offB + 12 * SB + 16 * v];
#pragma unroll
    for (int q = 2; q < 4; ++q) { *reinterpret_cast<d2*>(&bsn[sb0 + 4 * q * SB]) = rb[q]; rb[q] = *reinterpret_cast<const d2*>(pb + (size_t)(4 * q) * ldb); }
    LMM_MFMA16H_ALL(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
    LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1); LMM_SGB(0x008, 1); LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
    LMM_MFMA16(1, 0, 0); LMM_MFMA16(1, 0, 1); LMM_MFMA16(1, 0, 2); LMM_MFMA16(1, 0, 3); LMM_MFMA16(1, 1, 3);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) fa[0][u] = asn[offA + 16 * u];
#pragma unroll
    for (int v = 0; v < 2; ++v) fb[0][v] = bsn[offB + 16 * v];
    LMM_MFMA16(1, 1, 2); LMM_MFMA16(1, 1, 1); LMM_MFMA16(1, 1, 0);
#pragma unroll
    for (int i = 0; i < 3; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 2); }
  }
}

// FUSE: the NODE_FUSE work items exist (their waits and write-through stores cost the hot loop 3-4 spilled registers: 0.3 % on the
// K >= 4096 launches, which is why it is a template parameter and the long launches run without it)
template <int DEPTH, bool FUSE = false>
__global__ __launch_bounds__(256, 2) void potrf_node_kernel(NodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) double node_lds[];
  __shared__ int dflags[3];                                        // two-wave diagonal-block factorisation of the leaf (zero before its first barrier)
  if (threadIdx.x == 0) { dflags[0] = 0; dflags[1] = 0; dflags[2] = 0; }
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int SA = BM + 16, SB = BN + 16;
  double (*As)[BK * SA] = reinterpret_cast<double (*)[BK * SA]>(node_lds);
  double (*Bs)[BK * SB] = reinterpret_cast<double (*)[BK * SB]>(node_lds + 2 * BK * SA);
  // 1-D grid, dispatched in index order:
  //   update mode: [column tile 0 of every matrix, row tile by row tile: item = ti nb + b]  then  [matrix by matrix, the remaining tiles
  //     in gemm_work_item's band order with its XCD-aware remap and split-K tail].  Every matrix's tile (0, 0) -- and with it the
  //     leaf of the next panel -- thus starts in the FIRST scheduling round (with the column-0 tiles inside each matrix's own range
  //     the last matrix's unsplit K-long tiles ran in the tail of the launch: +1.6 ms per launch at C2 sizes), while the bulk of the
  //     tiles keeps the matrix-after-matrix order of gemm16p_kernel (interleaving the matrices of a batch instead measured 2-3 %
  //     slower on the K >= 2048 levels: sixteen operand panels of 0.5 GB compete for the caches instead of one).
  //   bulk mode: matrix by matrix, row tiles 1 .. MT-1.
  //   NODE_FUSE (update mode): after all of the above, the bulk tiles of the panel whose diagonal block this launch's leaf factors
  //     (row tile by row tile, item = bulk0 + (ti - 1) nb + b).  Bulk tile (b, ti) waits for the leaf of matrix b and for column-0
  //     tile (b, ti) of this launch -- workgroups dispatched before it, which wait for nothing: no deadlock however few are resident.
  //     Their results reach it write-through (ST_PUB) + flag, as in the region kernel.
  int part = 0, nparts = 1, tj = 0, ti = 0, bidx = 0;
  bool bulk = false, col0 = false;
  constexpr bool fuse = FUSE;
  {
    const int item = blockIdx.x;
    if (a.mode & NODE_UPDATE) {
      const int n0 = a.nb * a.MT;
      if (item < n0) { ti = item / a.nb; bidx = item - ti * a.nb; col0 = true; }
      else if (item >= a.strip0 && item < a.strip0 + a.nb * a.strip_n) {
        // the ragged last 64 rows of the region (rows a.M .. a.M + 63 below every column), one item per column tile: the 64 x 128 pipeline
        // (a separate gemm16h_kernel launch behind this one until round 4: N / 128 workgroups per matrix alone on the device for half
        // a tile time -- 55 us of a 1.08 ms K = 1024 launch, 435 us of the 30 ms K = 8192 one)
        const int q = item - a.strip0;                  // column tile by column tile, matrices interleaved: every matrix's column 0 first
        tj = q / a.nb; bidx = q - tj * a.nb;
        double* Am = a.A.p[bidx];
        const size_t r0s = (size_t)a.j0 + a.h;
        const double* Ap = Am + (size_t)a.j0 * a.ld + r0s;
        d4 acc[2][4];
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
          for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
        pipe64_accumulate(acc, node_lds, Ap + a.M, a.ld, Ap + 128 * tj, a.ld, a.h / 16);
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          double* cpv = Am + (r0s + 128 * tj + 32 * w + 16 * v + lk) * a.ld + r0s + a.M + l15;
          double cv[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * a.ld + 16 * u];
          if (fuse && tj == 0) {                          // the bulk tile of these rows reads column tile 0 in this launch
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int r = 0; r < 4; ++r) ST_PUB(cpv + (size_t)(4 * r) * a.ld + 16 * u, cv[u][r] - acc[v][u][r]);
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * a.ld + 16 * u] = cv[u][r] - acc[v][u][r];
          }
        }
        if (fuse && tj == 0) region_publish(a.nflags + (size_t)bidx * a.nf_stride + 2 + a.MT, a.epoch, 1);
        return;
      }
      else if (fuse && item >= a.bulk0) { bulk = true; const int q = item - a.bulk0; ti = q / a.nb; bidx = q - ti * a.nb; ti += 1; }
      else {
        const int q = item - n0 - a.nb * a.strip_n;
        bidx = q / a.rest_items;
        if (bidx < a.nb - 1) gemm_work_item_from(q - bidx * a.rest_items, 1, BM, BN, a.N, 1, a.MT, a.full_items, a.splitk, part, nparts, ti, tj);
        else {                                                     // the last matrix carries the launch's split-K tail
          bidx = a.nb - 1;
          gemm_work_item_from(q - bidx * a.rest_items, 1, BM, BN, a.N, 1, a.MT, a.full_items_last, a.splitk_last, part, nparts, ti, tj);
        }
      }
    } else { bulk = true; bidx = item / (a.MT - 1); ti = item - bidx * (a.MT - 1) + 1; }
  }
  double* Am = a.A.p[bidx];
  const int r0 = a.j0 + a.h;
  const int M = bulk ? a.Mb : a.M;
  int* nf = fuse ? a.nflags + (size_t)bidx * a.nf_stride : nullptr;
  if (bulk && fuse) {
    region_wait_ge<16>(nf + 1, a.epoch, 1, nf, a.info.p[bidx]);                          // the panel inverse
    if (ti < a.MT + (a.strip_n ? 1 : 0)) region_wait_ge<16>(nf + 2 + ti, a.epoch, 1, nf, a.info.p[bidx]);      // this tile's rows of the panel, updated (ti = MT: by the ragged-row item)
  }
  double* C = Am + (size_t)r0 * a.ld + r0;
  const double* A = bulk ? C : Am + (size_t)a.j0 * a.ld + r0;
  const double* B = bulk ? a.W2.p[bidx] + (size_t)(r0 / 128) * 16384 : A;
  const int lda = a.ld, ldb = bulk ? 128 : a.ld, ldc = a.ld;
  const int N = bulk ? 128 : a.N, K = bulk ? 128 : a.h;
  const int bm = ti * BM, bn = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * 64;
  const bool active = (bm + wr < M) && (bn + wc < N) && (bulk || !(bm + wr + 63 < bn + wc));
  const int nk_all = K / BK;
  const int kc0 = (int)((long long)nk_all * part / nparts);
  const int kc1 = (int)((long long)nk_all * (part + 1) / nparts);
  A += (size_t)kc0 * BK * lda;
  B += (size_t)kc0 * BK * ldb;
  const int nk = kc1 - kc0;

  int rowa = bm + 2 * (t & 63); if (rowa > M - 2) rowa = M - 2;
  int rowb = bn + 2 * (t & 63); if (rowb > N - 2) rowb = N - 2;
  const double* ga0 = A + (size_t)(t >> 6) * lda + rowa;       // thread t stages rows 2(t%64).. of k-columns t/64 + 4q
  const double* gb0 = B + (size_t)(t >> 6) * ldb + rowb;
  const int sa0 = (t >> 6) * SA + 2 * (t & 63);
  const int sb0 = (t >> 6) * SB + 2 * (t & 63);
  d2 ra[4], rb[4], ra2[DEPTH == 2 ? 4 : 1], rb2[DEPTH == 2 ? 4 : 1];
#pragma unroll
  for (int q = 0; q < 4; ++q) { ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(4 * q) * lda); rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(4 * q) * ldb); }
#pragma unroll
  for (int q = 0; q < 4; ++q) { *reinterpret_cast<d2*>(&As[0][sa0 + 4 * q * SA]) = ra[q]; *reinterpret_cast<d2*>(&Bs[0][sb0 + 4 * q * SB]) = rb[q]; }
  {
    const int k1 = nk > 1 ? 1 : 0, k2 = nk > 2 ? 2 : nk - 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)k1 * BK * lda + (size_t)(4 * q) * lda);
      rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)k1 * BK * ldb + (size_t)(4 * q) * ldb);
      if (DEPTH == 2) {
        ra2[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)k2 * BK * lda + (size_t)(4 * q) * lda);
        rb2[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)k2 * BK * ldb + (size_t)(4 * q) * ldb);
      }
    }
  }
  __syncthreads();

  d4 acc[4][4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + wr + l15, offB = lk * SB + wc + l15;
  double fa[2][4], fb[2][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { fa[0][u] = As[0][offA + 16 * u]; fb[0][u] = Bs[0][offB + 16 * u]; }

  if (DEPTH == 1) {
    for (int kt = 0; kt < nk; ++kt) {
      const int buf = kt & 1;
      const double* as = &As[buf][0];
      const double* bs = &Bs[buf][0];
      double* asn = &As[buf ^ 1][0];
      double* bsn = &Bs[buf ^ 1][0];
      const int kn = (kt + 2 < nk) ? kt + 2 : nk - 1;
      LMM_TILE_BODY(ra, rb, kn)
    }
  } else {
    for (int kt = 0; kt < nk; kt += 2) {
      {
        const double* as = &As[0][0]; const double* bs = &Bs[0][0];
        double* asn = &As[1][0]; double* bsn = &Bs[1][0];
        const int kn = (kt + 3 < nk) ? kt + 3 : nk - 1;
        LMM_TILE_BODY(ra, rb, kn)
      }
      if (kt + 1 < nk) {
        const double* as = &As[1][0]; const double* bs = &Bs[1][0];
        double* asn = &As[0][0]; double* bsn = &Bs[0][0];
        const int kn = (kt + 4 < nk) ? kt + 4 : nk - 1;
        LMM_TILE_BODY(ra2, rb2, kn)
      }
    }
  }
  if (active) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      double* cpv = C + (size_t)(bn + wc + 16 * v + lk) * ldc + bm + wr + l15;
      if (bulk) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * ldc + 16 * u] = acc[v][u][r];
      } else if (nparts == 1) {
        double cv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * ldc + 16 * u];
        if (fuse && col0) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) ST_PUB(cpv + (size_t)(4 * r) * ldc + 16 * u, cv[u][r] - acc[v][u][r]);
        } else {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * ldc + 16 * u] = cv[u][r] - acc[v][u][r];
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) unsafeAtomicAdd(cpv + (size_t)(4 * r) * ldc + 16 * u, -acc[v][u][r]);
      }
    }
  }
  if (fuse && col0 && ti >= 1) region_publish(nf + 2 + ti, a.epoch, 1);        // uniform over the workgroup
  if ((a.mode & NODE_LEAF) && !bulk && ti == 0 && tj == 0) {       // uniform over the workgroup
    __syncthreads();                                               // the tile's stores are issued; the staging LDS is free
    if (fuse) {
      leaf128_dev<true>(node_lds, Am, (size_t)r0 * a.ld + r0, a.ld, a.W.p[bidx], (size_t)(r0 / 64) * 4096,
                        a.W2.p[bidx] + (size_t)(r0 / 128) * 16384, r0, a.n_real, a.info.p[bidx], dflags);
      region_publish(nf + 1, a.epoch, 1);
    } else {
      leaf128_dev(node_lds, Am, (size_t)r0 * a.ld + r0, a.ld, a.W.p[bidx], (size_t)(r0 / 64) * 4096,
                  a.W2.p[bidx] + (size_t)(r0 / 128) * 16384, r0, a.n_real, a.info.p[bidx], dflags);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// K2d (round 3): a whole block column of panels in ONE launch -- potrf_region_kernel.
// The columns [c0, c0 + 128 P) of the factor matrix (P <= 8 panels: everything the recursion used to do below its K = 1024 level,
// i.e. per 1024 columns 8 bulk launches + 7 update launches whose K <= 512 products ran at 15-55 TFLOP/s with the leaf of every
// panel exposed; or the WHOLE factorisation of a matrix of up to 1024 columns) as a dataflow over 128 x 128 tiles:
//   task (i, j), 0 <= j < P, j <= i < R (R row tiles from row c0 down, rider rows included), one workgroup each, LEFT-LOOKING with the
//   tile in registers: acc = sum_{p < j} X[i, p] X[j, p]'  (one K = 128 pass of the pipelined MFMA loop per finished panel p, as
//   soon as the two operand tiles are flagged ready), then  C[i, j] -= acc  (the ONLY read-modify-write of the tile), then
//     i == j :  leaf128 (factor the diagonal block, its 64 x 64 inverse blocks and the panel inverse)            -> flag F[j][j]
//     i >  j :  wait F[j][j];  X[i, j] = C[i, j] Dinv_j'  (one more K = 128 pass, in place)                       -> flag F[j][i]
//   Flags are per matrix and carry the launch's epoch (no reset between launches); release = agent-scope fence + store, acquire =
//   spin on thread 0 + agent-scope fence (tools/fence_probe: 1-4 us per hand-off, profiles/r03).
//   Order of the 1-D grid = order of dispatch: wavefronts s = i + j, inside a wavefront the tile nearest the diagonal first, matrices
//   interleaved.  Every dependency of a task -- (i, p), (j, p) for p < j and (j, j) -- lies on an earlier wavefront, so a waiting
//   workgroup only ever waits for workgroups dispatched before it (already resident or finished): no deadlock however many are
//   resident; and the critical path leaf(j) -> X[j+1, j] -> tile (j+1, j+1) -> leaf(j+1) is at the FRONT of each wavefront.
//   Every spin is bounded (4 s): on a timeout the abort word is raised, all spinners leave, the grid drains and the host reports it.
// ---------------------------------------------------------------------------------------------------
// acc += A (128 rows x 128 k-columns, from row pointer A, leading dimension lda; rows >= rows_a are clamped) * B' (likewise), through
// the LDS staging buffers `lds` (4 x 16 x 144 doubles) with gemm16p_kernel's one-tile-ahead pipeline.  All 256 threads.
__device__ __forceinline__ void pipe128_accumulate(d4 (&acc)[4][4], double* __restrict__ lds, const double* __restrict__ A, int lda, int rows_a,
                                                   const double* __restrict__ B, int ldb, int rows_b, int nk = 8) {
  constexpr int BK = 16, SA = 144, SB = 144;
  double (*As)[BK * SA] = reinterpret_cast<double (*)[BK * SA]>(lds);
  double (*Bs)[BK * SB] = reinterpret_cast<double (*)[BK * SB]>(lds + 2 * BK * SA);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * 64;
  int rowa = 2 * (t & 63); if (rowa > rows_a - 2) rowa = rows_a - 2;
  int rowb = 2 * (t & 63); if (rowb > rows_b - 2) rowb = rows_b - 2;
  const double* ga0 = A + (size_t)(t >> 6) * lda + rowa;
  const double* gb0 = B + (size_t)(t >> 6) * ldb + rowb;
  const int sa0 = (t >> 6) * SA + 2 * (t & 63);
  const int sb0 = (t >> 6) * SB + 2 * (t & 63);
  d2 ra[4], rb[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)(4 * q) * lda); rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)(4 * q) * ldb); }
  __syncthreads();                                       // the previous user of the staging buffers is done with them
#pragma unroll
  for (int q = 0; q < 4; ++q) { *reinterpret_cast<d2*>(&As[0][sa0 + 4 * q * SA]) = ra[q]; *reinterpret_cast<d2*>(&Bs[0][sb0 + 4 * q * SB]) = rb[q]; }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ra[q] = *reinterpret_cast<const d2*>(ga0 + (size_t)BK * lda + (size_t)(4 * q) * lda);
    rb[q] = *reinterpret_cast<const d2*>(gb0 + (size_t)BK * ldb + (size_t)(4 * q) * ldb);
  }
  __syncthreads();
  const int l15 = lane & 15, lk = lane >> 4;
  const int offA = lk * SA + wr + l15, offB = lk * SB + wc + l15;
  double fa[2][4], fb[2][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { fa[0][u] = As[0][offA + 16 * u]; fb[0][u] = Bs[0][offB + 16 * u]; }
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const double* as = &As[buf][0];
    const double* bs = &Bs[buf][0];
    double* asn = &As[buf ^ 1][0];
    double* bsn = &Bs[buf ^ 1][0];
    const int kn = (kt + 2 < nk) ? kt + 2 : nk - 1;
    LMM_TILE_BODY(ra, rb, kn)
  }
}

// the same for up to three flags at once (one barrier, one fence); f2 / f3 may be nullptr
template <bool ACQ = true>
__device__ __forceinline__ void region_wait3(const int* f1, int n1, const int* f2, int n2, const int* f3, int n3, int epoch, int* abort_word, int* info) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int polls = 0;
    for (;;) {
      const int v1 = __hip_atomic_load(f1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int v2 = f2 ? __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch * 32 + 31;
      const int v3 = f3 ? __hip_atomic_load(f3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch * 32 + 31;
      if ((v1 >> 5) == epoch && (v1 & 31) >= n1 && (v2 >> 5) == epoch && (v2 & 31) >= n2 && (v3 >> 5) == epoch && (v3 & 31) >= n3) break;
      __builtin_amdgcn_s_sleep(1);
      if ((++polls & 63) == 0) {
        if (region_aborted(abort_word, epoch)) break;
        if (wall_clock64() - t0 > LMM_REGION_SPIN_TICKS) {
          __hip_atomic_store(abort_word, epoch * 32 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          atomicCAS(info, 0, LMM_INFO_SYNC_TIMEOUT);
          break;
        }
      }
    }
  }
  __syncthreads();
  if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}
// 64 x 64 block G (column-major, leading dimension ldg) -> LDS image img[col * DIAG_LS + row]; all 256 threads, 512-byte row runs
__device__ __forceinline__ void img_load(double* __restrict__ img, const double* __restrict__ G, int ldg) {
  const int t = threadIdx.x;
  double v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int e = t + 256 * i; v[i] = G[(size_t)(e >> 6) * ldg + (e & 63)]; }
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int e = t + 256 * i; img[(e >> 6) * DIAG_LS + (e & 63)] = v[i]; }
}

// The region's diagonal square (Q = 2P blocks of 64) is factored RIGHT-LOOKING at 64-column granularity -- the granularity that
// keeps the serial chain of a Cholesky short -- by one WALKER workgroup per matrix plus one HELPER per block row:
//   WALKER, block r:  [r > 0:  wait upd[r] >= r - 1;  L[r, r-1] = A[r, r-1] W_{r-1}'  (W_{r-1} is still in LDS from the previous
//                      diagonal block);  A[r, r] -= L[r, r-1] L[r, r-1]';  publish wk = r]   diag64m -> L[r, r], W_r.
//       The chain from one diagonal block to the next crosses NO flag and NO fence: ~10 us of diag64m + two 64^3 products.
//   HELPER r (r >= 2), step k = 0 .. r-2 as soon as wk > k:  L[r, k] = A[r, k] W_k';  publish trs[r] = k + 1;  then its tiles
//       A[r, c] -= L[r, k] L[c, k]'  -- (r, r-1) and (r, r) first (publish upd[r] = k + 1: the walker needs them at block r), then
//       c = k+1 .. r-2.  L[c, k] comes from the walker (c = k + 1) or helper c (wait trs[c] > k).
//   HELPER of an odd row b = 2j + 1 finally forms the panel inverse Dinv_j = [W_a 0; -W_b (L[b, a] W_a)  W_b] for the row streams.
// Operands of every 64^3 product are read straight from global memory into MFMA fragments where they are only used once; the
// operand reused across a step's updates (L[r, k]) sits in an LDS image.  Products are formed TRANSPOSED (D' = R' A') so that D's lane
// index runs along the rows of the stored tile: four 128-byte segments per store / read-modify-write instruction.
__device__ __forceinline__ void region_store_W(const RegionArgs& a, int b, const double* __restrict__ X, size_t grow, bool second) {
  constexpr int LS = DIAG_LS;
  double* Wm = a.W.p[b];
  double* W2p = a.W2.p[b] + (size_t)(grow / 128) * 16384;
  const int t = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = t + 256 * i, row = e & 63, col = e >> 6;
    const double v = X[col * LS + row];
    ST_PUB(&Wm[(size_t)(grow / 64) * 4096 + (size_t)col * 64 + row], v);
    if (!second) { ST_PUB(&W2p[(size_t)col * 128 + row], v); ST_PUB(&W2p[(size_t)(64 + col) * 128 + row], 0.0); }
    else ST_PUB(&W2p[(size_t)(64 + col) * 128 + 64 + row], v);
  }
}

#define MM64_ZERO(ACC) _Pragma("unroll") for (int u_ = 0; u_ < 2; ++u_) _Pragma("unroll") for (int v_ = 0; v_ < 2; ++v_) ACC[u_][v_] = (d4){0.0, 0.0, 0.0, 0.0}
// lane (c_, g_) of wave w holds D[i][j], i = wi + 16 u + 4 q + g_, j = wj + 16 v + c_, of a wg_mm64 product in acc[u][v][q]
#define MM64_FOREACH(BODY) _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int v = 0; v < 2; ++v) _Pragma("unroll") \
    for (int q = 0; q < 4; ++q) { const int i = wi + 16 * u + 4 * q + g_, j = wj + 16 * v + c_; BODY }

// NF ("no fence", the one-workgroup-per-CU build): hand-offs on the chain are consumed through sc1 loads instead of acquire fences
template <bool NF>
__device__ __forceinline__ void potrf_region_walker(const RegionArgs& a, double* __restrict__ lds, double* __restrict__ Am, int b, int* __restrict__ dflags) {
  int dbase = 0;                             // 16 x two-wave diagonal-block factorisations run so far (uniform)
  constexpr int LS = DIAG_LS;
  int* fl = a.flags.p[b];
  int* abort_word = fl; int* wk = fl + 1; int* upd = fl + 18;
  int* info = a.info.p[b];
  double* X = lds;                          // image of W_{r-1} (left there by diag64m), then of the updated diagonal tile
  double* Y = lds + 64 * LS;                // image of L[r, r-1], then diag64m's work areas
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int c_ = l & 15, g_ = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
  const int Q = 2 * a.P;
  int r0 = 0;
  if (a.first_done) {                        // panel 0 came factored out of the preceding update launch (leaf128)
    r0 = 2;
    img_load(X, a.W.p[b] + (size_t)((a.c0 + 64) / 64) * 4096, 64);     // W_1
    region_publish(wk, a.epoch, 1);            // W_0, L[1, 0]; wk = 2 (W_1 AND L[2, 1]) follows from block 2 below, as always
  }
  long long* tr2 = a.trace ? a.trace + 2 * (size_t)a.ntasks * a.nb + 64 * (size_t)b : nullptr;      // per-block stamps (debug)
  for (int r = r0; r < Q; ++r) {
    const size_t grow = (size_t)a.c0 + 64 * (size_t)r;
    const double* src = nullptr;
    if (tr2 && t == 0) tr2[r] = wall_clock64();
    if (r > 0 && (int)grow >= a.n_real) {
      // A block of padding columns only (the width is rounded up to 128: n = 552 -> 640): the assembly left L[r, :] = [0 .. 0 I]
      // there, which IS the factor, and W_r = I.  Nothing to wait for and nothing to compute -- a whole step of the chain less.
      __syncthreads();                                                   // readers of X / Y of the previous block are done
#pragma unroll
      for (int i = 0; i < 16; ++i) { const int e = t + 256 * i, row = e & 63, col = e >> 6; X[col * LS + row] = row == col ? 1.0 : 0.0; }
      region_publish(wk, a.epoch, r);                                    // W_{r-1} is stored; L[r, r-1] = 0 has been final all along
      region_store_W(a, b, X, grow, (r & 1) != 0);                       // (region_publish's barrier completed the image)
      continue;
    }
    if (r > 0) {
      const size_t gcol = grow - 64;
      if (r >= 2) region_wait_ge<1, !NF>(upd + r, a.epoch, r - 1, abort_word, info);     // tiles (r, r-1), (r, r) updated through block r - 2
      if (tr2 && t == 0) tr2[16 + r] = wall_clock64();
      // the diagonal tile (row j, column i) in the lanes that will hold its update D[i][j]
      double dt[2][2][4];
      const double* Ct = Am + grow * a.ld + grow;
      MM64_FOREACH(dt[u][v][q] = (j >= i) ? (NF ? ld_pub(&Ct[(size_t)i * a.ld + j]) : Ct[(size_t)i * a.ld + j]) : 0.0;)
      // L[r, r-1]' = W_{r-1} A[r, r-1]'   (A-operand W = X: (1, LS); R[k'][j] = A[r, r-1][j][k'], straight from global: (ld, 1))
      d4 acc[2][2];
      MM64_ZERO(acc);
      wg_mm64_core<false, NF ? 2 : 0>(acc, X, 1, LS, Am + gcol * a.ld + grow, a.ld, 1, w, l);
      __syncthreads();                                                   // previous readers of Y (work areas of the last diag64m) are done
      MM64_FOREACH(Y[i * LS + j] = acc[u][v][q]; ST_PUB(&Am[(gcol + i) * a.ld + grow + j], acc[u][v][q]);)        // D[i][j] = L[r, r-1][j][i]
      region_publish(wk, a.epoch, r);                                    // W_{r-1} (stored last iteration) and L[r, r-1] are final;
                                                                         // its barrier also completes Y = L[r, r-1] and ends the reads of X
      // (A[r, r] update)' = L L'   (A-operand Y: (1, LS); R[k'][j] = L[j][k'] = Y[k' LS + j]: (LS, 1)) -> the tile, as an image in X
      MM64_ZERO(acc);
      wg_mm64_core(acc, Y, 1, LS, Y, LS, 1, w, l);
      MM64_FOREACH(X[i * LS + j] = dt[u][v][q] - acc[u][v][q];)          // element (row j, column i); entries above the diagonal are never read
      src = X;
    }
    __syncthreads();                                                     // the image (r > 0) is complete, Y is free
    unsigned long long bad = 0ull;
    if (tr2 && t == 0) tr2[32 + r] = wall_clock64();
    if (w < 3) bad = diag64_pair<double>(Am, grow * a.ld + grow, a.ld, Y, X, dflags, dbase, w, l, src);      // L[r, r] -> global, W_r -> X
    dbase += 16;
    __syncthreads();
    if (tr2 && t == 0) tr2[48 + r] = wall_clock64();
    region_store_W(a, b, X, grow, (r & 1) != 0);
    if (t == 0) { const int i = diag_info_of(bad, (int)grow, a.n_real); if (i) atomicCAS(info, 0, i); }
  }
  region_publish(wk, a.epoch, Q);
}

// HELPER of square row r >= 2, LEFT-LOOKING: column blocks c = 0 .. r-2 in turn,
//     tile = A[r, c] - L[r, 0:c] L[c, 0:c]'    (ONE product over K = 64 c, operands straight from global, formed before W_c is needed)
//     L[r, c] = tile W_c'                       (as soon as the walker publishes W_c);  publish trs[r] = c + 1
//     pair accumulators (registers):  pa += L[r, c] L[r-1, c]',  pd += L[r, c] L[r, c]'
// and once c = r - 2 is done:  A[r, r-1] -= pa,  A[r, r] -= pd  (their only read-modify-write);  publish upd[r] = r - 1.
// The walker's request "tiles (r, r-1), (r, r) updated through block r - 2" thus costs, after W_{r-2} arrives, one solve, two 64^3
// products in registers and one write -- about the time the walker spends in diag64m of block r - 1.
template <bool DEEP, bool ASST, bool NF>
__device__ __forceinline__ void potrf_region_helper(const RegionArgs& a, double* __restrict__ lds, double* __restrict__ Am, int b, int r) {
  constexpr int LS = DIAG_LS;
  int* fl = a.flags.p[b];
  int* abort_word = fl; int* wk = fl + 1; int* trs = fl + 2; int* upd = fl + 18; int* dinv = fl + 34;
  int* info = a.info.p[b];
  double* Y = lds + 64 * LS;                // image of the tile, then of L[r, c]
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int c_ = l & 15, g_ = l >> 4, wi = 32 * (w & 1), wj = 32 * (w >> 1);
  const size_t grow = (size_t)a.c0 + 64 * (size_t)r;
  const size_t col0 = (size_t)a.c0 * a.ld;
  const double* Wm = a.W.p[b];
  const bool skip = a.first_done && r < 2;
  if (r >= 2) {
    d4 pa[2][2], pd[2][2];
    MM64_ZERO(pa); MM64_ZERO(pd);
    double* C1 = Am + (grow - 64) * a.ld + grow;
    double* C2 = Am + grow * a.ld + grow;
    // one column block of the row; LAST (c = r - 2, peeled out of the loop so that the registers below are live in it alone): the two
    // pair tiles' current values are requested BEFORE the wait for W_c -- nobody else writes them -- so their read-modify-write at the
    // end, on the path the walker waits for, costs no memory round trip (walker wait at rows 2-5: 4-5 us -> 0.4 us)
    auto column = [&](const int c, auto last_tag) {
      constexpr bool LAST = decltype(last_tag)::value;
      d4 pc1[2][2], pc2[2][2];
      const size_t gcol = (size_t)a.c0 + 64 * (size_t)c;
      d4 acc[2][2];
      MM64_ZERO(acc);
      if (c > 0) {
        // row c final through block c - 1: its helper's blocks (c >= 2) and the walker's subdiagonal block
        if (c >= 2) region_wait_ge<1, !NF>(trs + c, a.epoch, c - 1, abort_word, info);
        region_wait_ge<1, !NF>(wk, a.epoch, c, abort_word, info);
        // (tile update)' = L[c, 0:c] L[r, 0:c]'   (A[i][k'] = L[c][i][k']: (1, ld); R[k'][j] = L[r][j][k']: (ld, 1)), K = 64 c.
        // From column block LMM_REGION_ASST_MIN_C on, this row's ASSISTANT has formed the part over the blocks [0, cs) ahead of time
        // (its inputs are final several steps earlier); this workgroup multiplies the blocks [cs, c) and adds the assistant's tile.
        const int cs = (ASST && a.na > 0 && r >= LMM_REGION_ASST_MIN_R && c >= LMM_REGION_ASST_MIN_C) ? LMM_REGION_ASST_SPLIT(c) : 0;
        const size_t koff = (size_t)cs * 64 * a.ld;
        wg_mm64_core<DEEP, NF ? 1 : 0>(acc, Am + col0 + koff + gcol, 1, a.ld, Am + col0 + koff + grow, a.ld, 1, w, l, c - cs);      // row c: handed off; row r: own
        if (ASST && cs > 0) {
          region_wait_ge<1, !NF>(fl + 42 + r, a.epoch, c, abort_word, info);
          const double* St = a.S.p[b] + ((size_t)(r - LMM_REGION_ASST_MIN_R) * 16 + c) * 4096;
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[u][v][q] += NF ? ld_pub(&St[((u * 2 + v) * 4 + q) * 256 + t]) : St[((u * 2 + v) * 4 + q) * 256 + t];
        }
      }
      const double* Ct = Am + gcol * a.ld + grow;
      __syncthreads();                                                   // the previous column's readers of Y are done
      MM64_FOREACH(Y[i * LS + j] = Ct[(size_t)i * a.ld + j] - acc[u][v][q];)       // tile element (row j, column i) -> image Y[col][row]
      if constexpr (LAST) {
        MM64_FOREACH(pc1[u][v][q] = C1[(size_t)i * a.ld + j]; pc2[u][v][q] = (j >= i) ? C2[(size_t)i * a.ld + j] : 0.0;)
      }
      region_wait_ge<1, !NF>(wk, a.epoch, c + 1, abort_word, info);      // W_c (its barrier completes the image)
      // L[r, c]' = W_c tile'   (A[i][k'] = W_c[i][k'] global: (1, 64); R[k'][j] = tile[j][k'] = Y[k' LS + j]: (LS, 1))
      MM64_ZERO(acc);
      wg_mm64_core<false, NF ? 1 : 0>(acc, Wm + (size_t)(gcol / 64) * 4096, 1, 64, Y, LS, 1, w, l);
      __syncthreads();                                                   // reads of the tile image done
      MM64_FOREACH(Y[i * LS + j] = acc[u][v][q]; ST_PUB(&Am[(gcol + i) * a.ld + grow + j], acc[u][v][q]);)        // D[i][j] = L[r, c][j][i]
      region_publish(trs + r, a.epoch, c + 1);                           // (release; its barrier completes Y = L[r, c].  The rows below need
                                                                         // L[r, c] for THEIR pair tiles: delaying this flag to the end of the
                                                                         // last column -- one drain for both flags -- cost 5-10 us of walker
                                                                         // wait per block, round 4)
      // pair accumulators:  (r, r-1): D[i][j] += L[r-1, c][i][k'] L[r, c][j][k'];   (r, r): D[i][j] += L[r, c][i][k'] L[r, c][j][k']
      if (c <= r - 3) region_wait_ge<1, !NF>(trs + r - 1, a.epoch, c + 1, abort_word, info);      // helper r-1's L[r-1, c]; c = r-2: the walker's
      wg_mm64_core<false, NF ? 1 : 0>(pa, Am + gcol * a.ld + grow - 64, 1, a.ld, Y, LS, 1, w, l);
      wg_mm64_core(pd, Y, 1, LS, Y, LS, 1, w, l);
      if constexpr (LAST) {
        MM64_FOREACH(ST_PUB(&C1[(size_t)i * a.ld + j], pc1[u][v][q] - pa[u][v][q]);
                     if (j >= i) ST_PUB(&C2[(size_t)i * a.ld + j], pc2[u][v][q] - pd[u][v][q]);)
      }
    };
    for (int c = 0; c < r - 2; ++c) column(c, std::false_type{});
    column(r - 2, std::true_type{});
    region_publish(upd + r, a.epoch, r - 1);
  }
  if ((r & 1) && !skip) {
    // Dinv of panel j = r / 2 (a = r - 1, b = r), once the walker has published W_b:  T = L[b, a] W_a;  W21 = -W_b T
    region_wait_ge<1, !NF>(wk, a.epoch, r + 1, abort_word, info);
    const size_t gcol = grow - 64;
    d4 acc[2][2];
    MM64_ZERO(acc);
    // T = L[b, a] W_a   (A[i][k'] = L[i][k'] global: (1, ld); R[k'][j] = W_a[k'][j] global: (1, 64))
    wg_mm64_core<false, NF ? 3 : 0>(acc, Am + gcol * a.ld + grow, 1, a.ld, Wm + (size_t)(gcol / 64) * 4096, 1, 64, w, l);
    __syncthreads();
    MM64_FOREACH(Y[j * LS + i] = acc[u][v][q];)                          // Y = image of T: T[row i][col j] -> Y[col * LS + row]
    __syncthreads();
    double* W2p = a.W2.p[b] + (size_t)(grow / 128) * 16384;
    // D = T' W_b'  (A[i][k'] = T[k'][i] = Y[i LS + k']: (LS, 1); R[k'][j] = W_b[j][k'] global: (64, 1))  ->  W21[j][i] = -D[i][j]
    MM64_ZERO(acc);
    wg_mm64_core<false, NF ? 2 : 0>(acc, Y, LS, 1, Wm + (size_t)(grow / 64) * 4096, 64, 1, w, l);
    MM64_FOREACH(ST_PUB(&W2p[(size_t)i * 128 + 64 + j], -acc[u][v][q]);)
    region_publish(dinv + (r >> 1), a.epoch, 0);
  } else if (skip && r == 1) {
    region_publish(dinv + 0, a.epoch, 0);                                // the fused leaf left Dinv_0 in the scratch
  }
}

// ASSISTANT of square row r >= LMM_REGION_ASST_MIN_R.  A helper's left-looking product for column block c is K = 64 c long at ONE
// workgroup's rate (2.5 us per 64 k) and cannot start before row c is final through block c - 1; from row 8 on it no longer fits into
// the walker's 22-us block period, and at 16 blocks the chain ran at the helpers' pace (walker waits of 10-23 us per block,
// profiles/r03).  The first half of that product, over the blocks [0, c / 2), depends only on blocks that were final c / 2 steps
// earlier: the assistant forms it ahead of time into a scratch tile (in the helper's own lane layout: element ((u 2 + v) 4 + q) 256 + t,
// coalesced both ways) and flags it; the helper multiplies the blocks [c / 2, c) and adds the tile.
template <bool NF>
__device__ __forceinline__ void potrf_region_assistant(const RegionArgs& a, double* __restrict__ Am, int b, int r) {
  int* fl = a.flags.p[b];
  int* abort_word = fl; int* trs = fl + 2; int* asst = fl + 42;
  int* info = a.info.p[b];
  const int t = threadIdx.x, l = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const size_t grow = (size_t)a.c0 + 64 * (size_t)r;
  const size_t col0 = (size_t)a.c0 * a.ld;
  for (int c = LMM_REGION_ASST_MIN_C; c <= r - 2; ++c) {
    const int cs = LMM_REGION_ASST_SPLIT(c);
    const size_t gcol = (size_t)a.c0 + 64 * (size_t)c;
    region_wait3<!NF>(trs + c, cs, trs + r, cs, nullptr, 0, a.epoch, abort_word, info);       // L[c, 0:cs], L[r, 0:cs] final
    d4 acc[2][2];
    MM64_ZERO(acc);
    wg_mm64_core<false, NF ? 3 : 0>(acc, Am + col0 + gcol, 1, a.ld, Am + col0 + grow, a.ld, 1, w, l, cs);
    double* St = a.S.p[b] + ((size_t)(r - LMM_REGION_ASST_MIN_R) * 16 + c) * 4096;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int q = 0; q < 4; ++q) ST_PUB(&St[((u * 2 + v) * 4 + q) * 256 + t], acc[u][v][q]);
    region_publish(asst + r, a.epoch, c);
  }
}

// ROW task of potrf_region_kernel: row tile i >= P of matrix b (128 rows below the square, riders included), columns j = 0 .. P-1
// left to right.  It consumes only the square's rows 2j, 2j+1 (final through block column 2j - 1) and the panel inverses, produces
// only its own row -- which nobody else reads inside this launch -- so it signals nothing: no release fence, no flag traffic beyond
// 3 P waits.
//   column j:  acc = X[i, 0:j] X[j, 0:j]'  as ONE pass over K = 128 j (own earlier outputs + the square's rows);  C[i, j] -= acc;
//              wait Dinv_j;  X[i, j] = C[i, j] Dinv_j'.
__device__ __forceinline__ void potrf_region_row(const RegionArgs& a, double* __restrict__ lds, double* __restrict__ Am, int b, int ti) {
  int* abort_word = a.flags.p[b];
  int* wk = abort_word + 1; int* trs = abort_word + 2; int* dinv = abort_word + 34;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * 64;
  const int l15 = lane & 15, lk = lane >> 4;
  const int rows_i = min(128, a.M - 128 * ti);
  const size_t row_i = (size_t)a.c0 + 128 * (size_t)ti;
  const bool active = wr < rows_i;
  for (int tj = 0; tj < a.P; ++tj) {
    const size_t row_j = (size_t)a.c0 + 128 * (size_t)tj;
    double* C = Am + row_j * a.ld + row_i;
    d4 acc[4][4];
    if (tj > 0) {
#pragma unroll
      for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
      // the square's rows 2 tj, 2 tj + 1 final through block column 2 tj - 1: helpers' blocks + the walker's subdiagonal block
      region_wait3(trs + 2 * tj, 2 * tj - 1, wk, 2 * tj, trs + 2 * tj + 1, 2 * tj, a.epoch, abort_word, a.info.p[b]);
      const size_t col0 = (size_t)a.c0 * a.ld;
      pipe128_accumulate(acc, lds, Am + col0 + row_i, a.ld, rows_i, Am + col0 + row_j, a.ld, 128, 8 * tj);
      if (active) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          double* cpv = C + (size_t)(wc + 16 * v + lk) * a.ld + wr + l15;
          double cv[4][4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * a.ld + 16 * u];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * a.ld + 16 * u] = cv[u][r] - acc[v][u][r];
        }
      }
      __syncthreads();
    }
    region_wait_ge(dinv + tj, a.epoch, 0, abort_word, a.info.p[b]);                          // Dinv_j
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
    pipe128_accumulate(acc, lds, C, a.ld, rows_i, a.W2.p[b] + (size_t)(row_j / 128) * 16384, 128, 128);
    if (active) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        double* cpv = C + (size_t)(wc + 16 * v + lk) * a.ld + wr + l15;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * a.ld + 16 * u] = acc[v][u][r];
      }
    }
    __syncthreads();                                               // X[i, j] is in memory for this workgroup's own later passes
  }
}

// The same row task on a 64-row tile (rows c0 + roff .. + 63 of matrix b): half the work per workgroup, twice the workgroups -- for
// launches whose 128-row tiles leave CUs idle or come to a little more than a whole number of waves (launch_region's rule).
__device__ __forceinline__ void potrf_region_row64(const RegionArgs& a, double* __restrict__ lds, double* __restrict__ Am, int b, int roff) {
  int* abort_word = a.flags.p[b];
  int* wk = abort_word + 1; int* trs = abort_word + 2; int* dinv = abort_word + 34;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wc = 32 * w;
  const int l15 = lane & 15, lk = lane >> 4;
  const size_t row_i = (size_t)a.c0 + (size_t)roff;
  for (int tj = 0; tj < a.P; ++tj) {
    const size_t row_j = (size_t)a.c0 + 128 * (size_t)tj;
    double* C = Am + row_j * a.ld + row_i;
    d4 acc[2][4];
    if (tj > 0) {
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
      region_wait3(trs + 2 * tj, 2 * tj - 1, wk, 2 * tj, trs + 2 * tj + 1, 2 * tj, a.epoch, abort_word, a.info.p[b]);
      const size_t col0 = (size_t)a.c0 * a.ld;
      pipe64_accumulate(acc, lds, Am + col0 + row_i, a.ld, Am + col0 + row_j, a.ld, 8 * tj);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        double* cpv = C + (size_t)(wc + 16 * v + lk) * a.ld + l15;
        double cv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) cv[u][r] = cpv[(size_t)(4 * r) * a.ld + 16 * u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * a.ld + 16 * u] = cv[u][r] - acc[v][u][r];
      }
      __syncthreads();
    }
    region_wait_ge(dinv + tj, a.epoch, 0, abort_word, a.info.p[b]);                          // Dinv_j
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
    pipe64_accumulate(acc, lds, C, a.ld, a.W2.p[b] + (size_t)(row_j / 128) * 16384, 128, 8);
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      double* cpv = C + (size_t)(wc + 16 * v + lk) * a.ld + l15;
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) cpv[(size_t)(4 * r) * a.ld + 16 * u] = acc[v][u][r];
    }
    __syncthreads();                                               // X[i, j] is in memory for this workgroup's own later passes
  }
}

// THIN row task: a row tile with at most 16 real rows -- the rider rows of a logpdf (ONE projected observation vector per latent,
// zero-padded to 64 rows).  The full-tile stream above multiplies 128 x 128 tiles whatever the row count and, alone on its CU, runs
// ~60 us per panel behind the square at P = 5 (notebook shape: walker done at 200 us, row stream at 306).  Here the 16 rows are ONE
// MFMA row block: products are formed transposed, c' (128 panel columns x 16 rows), wave w owning panel columns 32 w .. 32 w + 31,
//   c' = C[i, j]' - X[j, 0:128 j] X[i, 0:128 j]'     operands straight from global memory (32-k chunks, one chunk prefetched), as soon
//                                                    as the square's rows of each earlier panel p are final (progressive waits);
//   X[i, j]' = Dinv_j c'                             c' through LDS (16 KB): the D layout of a 16x16x4 product IS the B-operand layout
//                                                    of the next one (register s of lane (g, c) = row 4 s + g), so nothing is transposed;
//                                                    Dinv is lower triangular: wave w stops at k = 32 (w + 1).
// Rows 16 .. of the tile are padding (zeros before and after).  Nothing is published: nobody reads a row stream inside the launch.
struct ThinChunk { double fb[8], fa[8][2]; };
__device__ __forceinline__ void potrf_region_row_thin(const RegionArgs& a, double* __restrict__ lds, double* __restrict__ Am, int b, int roff) {
  int* abort_word = a.flags.p[b];
  int* wk = abort_word + 1; int* trs = abort_word + 2; int* dinv = abort_word + 34;
  int* info = a.info.p[b];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, lk = lane >> 4;
  const size_t row_i = (size_t)a.c0 + (size_t)roff;       // the tile's first row (its rows 16 .. are padding)
  double* cT = lds;                                      // c'[k][row]: 128 x 16 doubles
  for (int tj = 0; tj < a.P; ++tj) {
    const size_t row_j = (size_t)a.c0 + 128 * (size_t)tj;
    const size_t jw = row_j + 32 * (size_t)w;            // this wave's panel columns = rows jw .. jw + 31 of the square
    d4 acc[2];
    acc[0] = acc[1] = (d4){0.0, 0.0, 0.0, 0.0};
    const int R = 2 * tj;
    for (int p = 0; p < tj; ++p) {
      // square rows R, R + 1 final in the columns of panel p (64-column blocks 2p, 2p + 1): helpers' blocks; the walker's
      // subdiagonal block L[R, R - 1] when 2p + 1 = R - 1
      if (p < tj - 1) region_wait3(trs + R, 2 * p + 2, trs + R + 1, 2 * p + 2, nullptr, 0, a.epoch, abort_word, info);
      else region_wait3(trs + R + 1, 2 * p + 2, wk, R, R >= 2 ? trs + R : nullptr, R - 1, a.epoch, abort_word, info);
      auto load_chunk = [&](ThinChunk& ch, int k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const size_t kc = ((size_t)a.c0 + k0 + 4 * q + lk) * a.ld;
          ch.fb[q] = Am[kc + row_i + l15];
          ch.fa[q][0] = Am[kc + jw + l15];
          ch.fa[q][1] = Am[kc + jw + 16 + l15];
        }
      };
      auto mul_chunk = [&](const ThinChunk& ch) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ch.fa[q][0], ch.fb[q], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ch.fa[q][1], ch.fb[q], acc[1], 0, 0, 0);
        }
      };
      ThinChunk c0_, c1_;
      load_chunk(c0_, 128 * p);
      load_chunk(c1_, 128 * p + 32);
      mul_chunk(c0_);
      load_chunk(c0_, 128 * p + 64);
      mul_chunk(c1_);
      load_chunk(c1_, 128 * p + 96);
      mul_chunk(c0_);
      mul_chunk(c1_);
    }
    // c' = C' - acc  ->  LDS   (lane (g, c), register r of block v: panel column 32 w + 16 v + 4 r + g, tile row c)
    __syncthreads();                                     // the previous panel's readers of cT are done
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 32 * w + 16 * v + 4 * r + lk;
        cT[col * 16 + l15] = Am[(row_j + col) * a.ld + row_i + l15] - acc[v][r];
      }
    region_wait_ge(dinv + tj, a.epoch, 0, abort_word, info);                          // Dinv_j (its barrier completes cT)
    const double* Dv = a.W2.p[b] + (size_t)(row_j / 128) * 16384;                      // Dinv[out][k] = Dv[k * 128 + out]
    d4 xo[2];
    xo[0] = xo[1] = (d4){0.0, 0.0, 0.0, 0.0};
    const int nch = w + 1;                               // 32-k chunks: k < 32 (w + 1) (lower triangular)
    for (int ch = 0; ch < nch; ++ch) {
      double fa[8][2], fb[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 32 * ch + 4 * q + lk;
        fa[q][0] = Dv[(size_t)k * 128 + 32 * w + l15];
        fa[q][1] = Dv[(size_t)k * 128 + 32 * w + 16 + l15];
        fb[q] = cT[k * 16 + l15];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        xo[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[q][0], fb[q], xo[0], 0, 0, 0);
        xo[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[q][1], fb[q], xo[1], 0, 0, 0);
      }
    }
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) Am[(row_j + 32 * w + 16 * v + 4 * r + lk) * a.ld + row_i + l15] = xo[v][r];
    __syncthreads();                                     // X[i, j] is in memory for this workgroup's own later passes
  }
}

// 1-D grid in dispatch order (matrices interleaved): the walkers, the helpers of square rows 1 .. 2P-1, then the row streams.
// Deadlock freedom: helper r waits only for the walker and for helpers of rows above it, a row stream only for the square -- all
// dispatched before it.  The walker is the one workgroup that waits for a LATER one (helper r, at block r); but by then it has
// published everything the helpers of rows < r need to finish (they never wait beyond wk = r - 1), so they complete and free their
// slots however few workgroups are resident, helper r gets dispatched and runs.  One workgroup per CU (414 registers per lane).
// OCC 1: one workgroup per CU (no register spills in the walker); 2: two (the row streams' natural occupancy)
// STRICT: the task's index is CLAIMED at entry (region_claim), so "dispatched before it" in the argument above reads "started before
// it" and holds in any dispatch order
template <int OCC, bool STRICT = false>
__global__ __launch_bounds__(256, OCC) void potrf_region_kernel(RegionArgs a) {
  extern __shared__ __attribute__((aligned(16))) double node_lds[];
  __shared__ int dflags[3];                  // the walker's two-wave diagonal-block factorisation (zero before the walker's first barrier)
  if (threadIdx.x == 0) { dflags[0] = 0; dflags[1] = 0; dflags[2] = 0; }
  const int b = blockIdx.x % a.nb;
  int idx = blockIdx.x / a.nb;
  if (STRICT && !region_claim(a.claim, a.claim_next, a.nb, a.ntasks, b, idx, a.claim_scramble)) {      // counters out of step (never expected): report, do nothing
    if (threadIdx.x == 0) atomicCAS(a.info.p[0], 0, LMM_INFO_SYNC_TIMEOUT);
    return;
  }
  const int task = idx * a.nb + b;
  double* Am = a.A.p[b];
  const int Q = 2 * a.P;
  if (a.trace && threadIdx.x == 0) a.trace[2 * task] = wall_clock64();
  // square tasks in dispatch order: walker, helper 1, ..., helper MIN_R - 1, then (helper r, assistant r) pairs -- a helper's assistant is
  // the NEXT workgroup to be dispatched, so a helper never waits for a workgroup that other waiting workgroups keep out of the device
  int role = 1, r = idx;                      // 0 walker, 1 helper, 2 assistant
  if (idx == 0) role = 0;
  else if (OCC == 1 && a.na > 0 && idx >= LMM_REGION_ASST_MIN_R && idx < Q + a.na) {
    const int k = idx - LMM_REGION_ASST_MIN_R;
    r = LMM_REGION_ASST_MIN_R + (k >> 1); role = 1 + (k & 1);
  }
  if (idx >= Q + a.na) role = 3;
  if (role == 0) potrf_region_walker<OCC == 1>(a, node_lds, Am, b, dflags);
  else if (role == 1) potrf_region_helper<false, OCC == 1, OCC == 1>(a, node_lds, Am, b, r);      // DEEP (two chunks ahead) spills even at one workgroup per CU
  else if (role == 2) { if (OCC == 1) potrf_region_assistant<true>(a, Am, b, r); }       // (the two-per-CU build has no assistants)
  else {
    const int k = idx - Q - a.na;                        // row tasks in dispatch order: n128 tiles of 128 rows, then 64-row tiles
    const bool tall = k < a.n128;
    const int roff = tall ? 128 * (a.P + k) : 128 * (a.P + a.n128) + 64 * (k - a.n128);
    const int real = a.M_real - roff;                    // rows of this tile that hold data (the rest is zero padding, before and after)
    if (real > 16) { if (tall) potrf_region_row(a, node_lds, Am, b, a.P + k); else potrf_region_row64(a, node_lds, Am, b, roff); }
    else if (real > 0) potrf_region_row_thin(a, node_lds, Am, b, roff);
  }
  if (a.trace && threadIdx.x == 0) a.trace[2 * task + 1] = wall_clock64();
}

#undef LMM_MFMA16H_ALL
#undef LMM_TILE_BODY
#undef LMM_MFMA16_ALL
#undef LMM_MFMA16

// ---------------------------------------------------------------------------------------------------
// K2b (fp32 compute mode): the same C {-=, =} A B' on v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = 157 TFLOP/s peak:
// twice the FP64 matrix rate; MI355X_MICROARCH.md "Matrix cores").  128 x BN block tile, 4 waves (2 x 2), wave tile
// 64 x BN/2 as 32x32 MFMA blocks, BK = 16, two LDS stages, one barrier per stage, k-major LDS image (a k-column of the
// tile is one contiguous 512-byte global segment; a fragment read is one conflict-free ds_read_b32: 32 consecutive floats
// per half-wave).  The MFMA's A operand is fed from the B matrix and its B operand from the A matrix, so the result comes
// out transposed: the lane index of D runs along the ROWS of C (contiguous in memory) and the read-modify-write epilogue is
// coalesced 128-byte segments without an LDS transpose:
//     acc[tv][tu][r] (lane l)  <->  C[bm + wr + 32 tu + (l & 31),  bn + wc + 32 tv + (r & 3) + 8 (r >> 2) + 4 (l >> 5)].
// ---------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Software-pipelined like gemm16p_kernel (one MFMA of 64 cycles in flight covers the LDS reads, LDS writes, global loads and the
// barrier issued behind it): BK = 32 per LDS stage = 16 k-pairs of TU x TV MFMAs; per k-pair the fragments of the NEXT pair are
// read, and one staging instruction (ds_write_b128 of tile t+1 in the first half, global_load_dwordx4 of tile t+2 in the second)
// rides along; the barrier sits before the last k-pair, whose MFMAs cover the first fragment reads of tile t+1.
template <int BN, bool SET, int SCHED = 0>
__global__ __launch_bounds__(256, SCHED == 1 ? 1 : 2) void gemm32_kernel(BatchPtr Cb, size_t goffC, int ldc, BatchPtr Ab, size_t goffA, int lda,
                                                         BatchPtr Bb, size_t goffB, int ldb,
                                                         int M, int N, int K, int lower, int MT, int full_items,
                                                         int splitk, int kfrom_row) {
  float* C = reinterpret_cast<float*>(Cb.p[blockIdx.y]) + goffC;
  const float* A = reinterpret_cast<const float*>(Ab.p[blockIdx.y]) + goffA;
  const float* B = reinterpret_cast<const float*>(Bb.p[blockIdx.y]) + goffB;
  constexpr int BM = 128, BK = 32;
  constexpr int WN = BN / 2;
  constexpr int TU = 2, TV = WN / 32;
  constexpr int SA = BM + 4, SB = BN + 4;
  constexpr int NLA = (BM * BK / 4) / 256;     // 4: thread t stages rows 4(t%32).. of k-columns t/32 + 8q
  constexpr int NLB = (BN * BK / 4) / 256;     // 4 (BN=128) or 2 (BN=64)
  constexpr int KSB = 256 / (BN / 4);          // k-columns covered per pass of the B staging (8 or 16)
  __shared__ __attribute__((aligned(16))) float As[2][BK * SA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * SB];

  int part = 0, nparts = 1, tj = 0, ti = 0;
  gemm_work_item(BM, BN, N, lower, MT, full_items, splitk, part, nparts, ti, tj);
  const int bm = ti * BM, bn = tj * BN;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wr = (w & 1) * 64, wc = (w >> 1) * WN;
  const bool active = (bm + wr < M) && (bn + wc < N) && !(lower && bm + wr + 63 < bn + wc);
  // K is a multiple of 16 (callers: multiples of 64); a trailing half stage (K % 32 == 16) is handled by zero-weighting: the
  // k range of a part is cut on 32-column boundaries and K % 32 != 0 falls back to one extra half-filled stage
  const int nk_all = (K + BK - 1) / BK;
  int kc0 = (int)((long long)nk_all * part / nparts);
  const int kc1 = (int)((long long)nk_all * (part + 1) / nparts);
  if (kfrom_row && kc0 < bm / BK) kc0 = bm / BK;
  const int nk = kc1 - kc0;
  const int kmax = K - 1;                      // last valid k-column (loads are clamped to it; clamped duplicates are zeroed below)

  int rowa = bm + 4 * (t % (BM / 4)); if (rowa > M - 4) rowa = M - 4;
  int rowb = bn + 4 * (t % (BN / 4)); if (rowb > N - 4) rowb = N - 4;
  const int ka = t / (BM / 4), kb = t / (BN / 4);          // this thread's k-column inside a pass
  const float* gA = A + rowa;
  const float* gB = B + rowb;
  const int sa0 = ka * SA + 4 * (t % (BM / 4));
  const int sb0 = kb * SB + 4 * (t % (BN / 4));
  float4 ra[NLA], rb[NLB];
  auto load_tile = [&](int kt) {                // global -> registers for k-stage kt (absolute stage index), zero beyond K
    const int k0 = kt * BK;
#pragma unroll
    for (int q = 0; q < NLA; ++q) {
      const int k = k0 + ka + 8 * q;
      ra[q] = *reinterpret_cast<const float4*>(gA + (size_t)(k <= kmax ? k : kmax) * lda);
      if (k > kmax) ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < NLB; ++q) {
      const int k = k0 + kb + KSB * q;
      rb[q] = *reinterpret_cast<const float4*>(gB + (size_t)(k <= kmax ? k : kmax) * ldb);
      if (k > kmax) rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  load_tile(kc0);
#pragma unroll
  for (int q = 0; q < NLA; ++q) *reinterpret_cast<float4*>(&As[0][sa0 + 8 * q * SA]) = ra[q];
#pragma unroll
  for (int q = 0; q < NLB; ++q) *reinterpret_cast<float4*>(&Bs[0][sb0 + KSB * q * SB]) = rb[q];
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
  float fu[2][TU], fv[2][TV];
#pragma unroll
  for (int u = 0; u < TU; ++u) fu[0][u] = As[0][offA + 32 * u];
#pragma unroll
  for (int v = 0; v < TV; ++v) fv[0][v] = Bs[0][offB + 32 * v];

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const float* as = &As[buf][0];
    const float* bs = &Bs[buf][0];
    float* asn = &As[buf ^ 1][0];
    float* bsn = &Bs[buf ^ 1][0];
    const int kn2 = kc0 + ((kt + 2 < nk) ? kt + 2 : nk - 1);
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      const int cur = kp & 1, nxt = cur ^ 1;
      // before the LAST pair: every fragment of this buffer has been read (pair 15's were issued during pair 14) and tile t+1 is
      // complete in the other buffer
      if (kp == BK / 2 - 1) __syncthreads();
      // fragments of the next k-pair (the last pair reads the first pair of tile t+1 from the other buffer)
      if (kp + 1 < BK / 2) {
#pragma unroll
        for (int u = 0; u < TU; ++u) fu[nxt][u] = as[offA + 2 * (kp + 1) * SA + 32 * u];
#pragma unroll
        for (int v = 0; v < TV; ++v) fv[nxt][v] = bs[offB + 2 * (kp + 1) * SB + 32 * v];
      } else {
#pragma unroll
        for (int u = 0; u < TU; ++u) fu[nxt][u] = asn[offA + 32 * u];
#pragma unroll
        for (int v = 0; v < TV; ++v) fv[nxt][v] = bsn[offB + 32 * v];
      }
      // one staging instruction per k-pair: ds_write of tile t+1 (pairs 0 .. NLA+NLB-1), then the global loads of tile t+2
      if (kp < NLA) *reinterpret_cast<float4*>(&asn[sa0 + 8 * kp * SA]) = ra[kp];
      else if (kp < NLA + NLB) *reinterpret_cast<float4*>(&bsn[sb0 + KSB * (kp - NLA) * SB]) = rb[kp - NLA];
      else if (kp < 2 * NLA + NLB) {
        const int q = kp - NLA - NLB, k = kn2 * BK + ka + 8 * q;
        ra[q] = *reinterpret_cast<const float4*>(gA + (size_t)(k <= kmax ? k : kmax) * lda);
        if (k > kmax) ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (kp < 2 * NLA + 2 * NLB) {
        const int q = kp - 2 * NLA - NLB, k = kn2 * BK + kb + KSB * q;
        rb[q] = *reinterpret_cast<const float4*>(gB + (size_t)(k <= kmax ? k : kmax) * ldb);
        if (k > kmax) rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (active) {
#pragma unroll
        for (int v = 0; v < TV; ++v)
#pragma unroll
          for (int uu = 0; uu < TU; ++uu) {
            const int u = (v & 1) ? TU - 1 - uu : uu;
            acc[v][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fv[cur][v], fu[cur][u], acc[v][u], 0, 0, 0);
          }
      }
      if (SCHED == 1) {          // pin: one fragment read behind each MFMA, the staging instruction behind the last
#pragma unroll
        for (int i = 0; i < TU + TV; ++i) { LMM_SGB(0x008, 1); LMM_SGB(0x100, 1); }
        LMM_SGB(0x200, 1); LMM_SGB(0x020, 1);
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int v = 0; v < TV; ++v)
#pragma unroll
    for (int u = 0; u < TU; ++u) {
      float* cp = C + (size_t)(bn + wc + 32 * v + 4 * lh) * ldc + bm + wr + 32 * u + l31;
      if (SET) {
#pragma unroll
        for (int r = 0; r < 16; ++r) cp[(size_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[v][u][r];
      } else if (nparts == 1) {
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

// ---------------------------------------------------------------------------------------------------
// K3/K6: per-latent log marginal likelihood from the factor:  -(n log 2pi + 2 sum log L_kk + |z|^2)/2,
// z = rider row `rider_row` (= (L^-1 delta)').  One workgroup; wavefront shuffle reductions.
// Also used with nrhs > 1 riders (matrix-Y): out[r].
// ---------------------------------------------------------------------------------------------------
// One workgroup per matrix of the batch (blockIdx.x); matrix b writes out[b * nrhs + r].
template <typename TS>
__global__ __launch_bounds__(256) void lml_reduce_kernel(BatchPtr Ab, int ld, int n,
                                                         int rider_row0, int nrhs, double* __restrict__ out, BatchInfo infob, int* __restrict__ info_out) {
  const void* __restrict__ A = Ab.p[blockIdx.x];
  if (info_out && threadIdx.x == 0) info_out[blockIdx.x] = *infob.p[blockIdx.x];
  out += (size_t)blockIdx.x * nrhs;
  __shared__ double sh[4];
  double sl = 0.0;
  for (int k = threadIdx.x; k < n; k += 256) sl += log(MatIO<TS>::ld1(A, (size_t)k * ld + k));
  const double sumlog = block_sum_256(sl, sh);
  __shared__ double bc;
  if (threadIdx.x == 0) bc = sumlog;
  __syncthreads();
  for (int r = 0; r < nrhs; ++r) {
    double q = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) {
      const double v = MatIO<TS>::ld1(A, (size_t)k * ld + rider_row0 + r);
      q = __builtin_fma(v, v, q);
    }
    const double quad = block_sum_256(q, sh);
    if (threadIdx.x == 0) out[r] = -0.5 * ((double)n * LOG2PI + 2.0 * bc + quad);
  }
}

// Extract rider row r (length n) of a factor matrix into a contiguous vector.
template <typename TS>
__global__ void extract_row_kernel(const void* __restrict__ A, int ld, int row, int n, double* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = MatIO<TS>::ld1(A, (size_t)k * ld + row);
}

// The same for the matrices of a batch (blockIdx.y), zero-filled up to nfill, into two destinations per matrix (alpha, which
// the back substitution then overwrites, and the kept copy z = L^-1 delta).
template <typename TS>
__global__ void extract_rows_kernel(BatchPtr Ab, int ld, int row, int n, int nfill, BatchPtr o1, BatchPtr o2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nfill) return;
  const double v = (k < n) ? MatIO<TS>::ld1(Ab.p[blockIdx.y], (size_t)k * ld + row) : 0.0;
  o1.p[blockIdx.y][k] = v;
  o2.p[blockIdx.y][k] = v;
}

// Strip reductions over a column-major block M (rows r contiguous in memory, columns k, stride ld): for the 64-row strip
// blockIdx.x and the k-chunk blockIdx.y (kc columns) of this workgroup
//   dot[r] = sum_k M[r,k] v[k]        and, SQ:  sq[r] = sum_k M[r,k]^2       (TRI: only k <= r, i.e. the product L z)
// into partial[(chunk * NV + val) * nrp + r] (NV = SQ ? 2 : 1, nrp = 64 gridDim.x); strip_finish_kernel adds the chunks
// in a fixed order.  Serves the posterior marginals (mean = mu + R' z, var = k(x*,x*) - colsumsq(R) with R = L^-1 K(x,x*)
// -- AbstractGPs PosteriorGP mean/var) and the sample transform L z.  The four waves take
// interleaved 16-column slabs so that 16 independent coalesced 512-byte loads per wave are in flight.
template <bool SQ, bool TRI, typename TS>
__global__ __launch_bounds__(256) void strip_reduce_kernel(const void* __restrict__ M, int ld, int nk, int kc,
                                                           const double* __restrict__ v, double* __restrict__ partial) {
  __shared__ double red[2][3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + lane;
  const int kbeg = blockIdx.y * kc;
  int kend = kbeg + kc; if (kend > nk) kend = nk;
  if (TRI) { const int rowmax = blockIdx.x * 64 + 63; if (kend > rowmax + 1) kend = rowmax + 1; if (kbeg > rowmax) return; }
  double dot = 0.0, sq = 0.0;
  for (int k0 = kbeg + w * 16; k0 < kend; k0 += 64) {
    if (k0 + 16 <= kend) {
      double a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = MatIO<TS>::ld1(M, (size_t)(k0 + u) * ld + r);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const double av = (TRI && (k0 + u > r)) ? 0.0 : a[u];
        dot = __builtin_fma(av, v[k0 + u], dot);
        if (SQ) sq = __builtin_fma(av, av, sq);
      }
    } else {
      for (int k = k0; k < kend; ++k) {
        double av = MatIO<TS>::ld1(M, (size_t)k * ld + r);
        if (TRI && k > r) av = 0.0;
        dot = __builtin_fma(av, v[k], dot);
        if (SQ) sq = __builtin_fma(av, av, sq);
      }
    }
  }
  if (w > 0) { red[0][w - 1][lane] = dot; if (SQ) red[1][w - 1][lane] = sq; }
  __syncthreads();
  if (w == 0) {
    const int NV = SQ ? 2 : 1;
    const size_t nrp = (size_t)gridDim.x * 64;
    dot += red[0][0][lane] + red[0][1][lane] + red[0][2][lane];
    partial[((size_t)blockIdx.y * NV) * nrp + r] = dot;
    if (SQ) {
      sq += red[1][0][lane] + red[1][1][lane] + red[1][2][lane];
      partial[((size_t)blockIdx.y * NV + 1) * nrp + r] = sq;
    }
  }
}

// out_dot[r] = add_dot + sum_chunks dot;  out_sq[r] = base_sq - sum_chunks sq.   tri_kc > 0: chunks above the diagonal were
// skipped by strip_reduce_kernel<.,true> and are skipped here too.
__global__ void strip_finish_kernel(const double* __restrict__ partial, int nrp, int nch, int nv, int nr, int tri_kc,
                                    double add_dot, double base_sq, double* __restrict__ out_dot,
                                    double* __restrict__ out_sq) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nr) return;
  double d = 0.0, q = 0.0;
  for (int c = 0; c < nch; ++c) {
    if (tri_kc > 0 && c * tri_kc > (r | 63)) break;
    d += partial[((size_t)c * nv) * nrp + r];
    if (nv > 1) q += partial[((size_t)c * nv + 1) * nrp + r];
  }
  if (out_dot) out_dot[r] = add_dot + d;
  if (out_sq) out_sq[r] = base_sq - q;
}

// ---------------------------------------------------------------------------------------------------
// Back substitution  L' alpha = z  (alpha = C \ delta second half), one launch per 64-block from the last block to the
// first, for every matrix of the batch (blockIdx.y).  Step b: alpha_b = W_bb' z_b (every workgroup recomputes it from an
// LDS copy of the 64x64 inverse block), then z_i -= sum_{j in b} L[j, i] alpha_j for the 256 columns i < 64 b of this
// workgroup: 16 lanes share a column (4 consecutive rows each, so a wave load covers 4 whole 512-byte column segments).
// ---------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void backsolve_step_kernel(BatchPtr Lb, int ld, BatchPtr Wb_, int b, BatchPtr zb_) {
  __shared__ double Ws[64 * 65];
  __shared__ double zb[64];
  __shared__ double ps[4][64];
  __shared__ double ab[64];
  const void* __restrict__ L = Lb.p[blockIdx.y];
  const void* __restrict__ Wb = Wb_.p[blockIdx.y];
  const size_t woff = (size_t)b * 4096;
  double* __restrict__ z = zb_.p[blockIdx.y];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  // issue this workgroup's 32 panel loads first: they do not depend on alpha_b
  const int c0 = blockIdx.x * 256 + w * 64;
  const int jq = lane & 15, cq = lane >> 4;
  double2 pa[16], pb[16];
#pragma unroll
  for (int gI = 0; gI < 16; ++gI) {
    const int col = c0 + gI * 4 + cq;
    if (col < b * 64) {
      double q4[4];
      MatIO<TS>::ld4(L, (size_t)col * ld + b * 64 + jq * 4, q4);
      pa[gI] = make_double2(q4[0], q4[1]); pb[gI] = make_double2(q4[2], q4[3]);
    } else { pa[gI] = make_double2(0.0, 0.0); pb[gI] = pa[gI]; }
  }
  for (int e = t; e < 4096; e += 256) Ws[(e >> 6) * 65 + (e & 63)] = MatIO<TS>::ld1(Wb, woff + e);     // Wb[c*64 + j] = W[j, c]
  if (t < 64) zb[t] = z[b * 64 + t];
  __syncthreads();
  {
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int j = w * 16 + u;
      if (j >= lane) s = __builtin_fma(Ws[lane * 65 + j], zb[j], s);          // (W')[t, j] = W[j, t], j >= t
    }
    ps[w][lane] = s;
  }
  __syncthreads();
  if (t < 64) ab[t] = (ps[0][t] + ps[1][t]) + (ps[2][t] + ps[3][t]);
  __syncthreads();
  if (blockIdx.x == 0 && t < 64) z[b * 64 + t] = ab[t];
  const double a0 = ab[jq * 4], a1 = ab[jq * 4 + 1], a2 = ab[jq * 4 + 2], a3 = ab[jq * 4 + 3];
#pragma unroll
  for (int gI = 0; gI < 16; ++gI) {
    double s = __builtin_fma(pa[gI].x, a0, __builtin_fma(pa[gI].y, a1, __builtin_fma(pb[gI].x, a2, pb[gI].y * a3)));
    s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
    const int col = c0 + gI * 4 + cq;
    if (jq == 0 && col < b * 64) z[col] -= s;
  }
}

// ---------------------------------------------------------------------------------------------------
// K4/K5: tall-skinny products with a thread per row i of an n-row operand.
//   val[i, c] = sum_k Mx[c + k*ldm] * In[i + k*ldi]            (c in this block's chunk of CH outputs)
//   mode 0: Out[i + c*ldo] = val - sub[c]                       (T*Y projection, minus latent mean)
//   mode 1: partial[block] = sum (Ref[i + c*ldr] - val)^2       (regulariser residual |Y - H T Y|_F^2)
// ---------------------------------------------------------------------------------------------------
template <int CH, int NW>
__global__ __launch_bounds__(64 * NW) void tall_skinny_kernel(const double* __restrict__ In, int ldi, int n, int K,
                                                          const double* __restrict__ Mx, int ldm, int C,
                                                          double* __restrict__ Out, int ldo,
                                                          const double* __restrict__ sub,
                                                          const double* __restrict__ Ref, int ldr,
                                                          double* __restrict__ partial, int mode) {
  // 64 rows per workgroup; the NW waves split the K loop (k = wave, wave + NW, ...) and combine through LDS, so that
  // short-and-wide problems (n of a few hundred, K = p of several hundred) still expose enough parallelism: NW = 16 when the
  // grid alone would leave most CUs idle (the reference notebook's projection, n = 552, p = 600, m = 20: 27 workgroups of 4 waves
  // each walking 150 dependent loads took 114 us of a 570-us evaluation).
  __shared__ double red[NW - 1][CH][64];
  __shared__ double sh[4];
  // ks is wave-uniform: as an SGPR it makes every Mx address scalar, so the CH coefficients of a k-step arrive by scalar loads (one
  // s_load per k-step) instead of CH vector loads of 64 identical addresses -- the kernel was bound by the vector-memory issue rate
  // of the few CUs it runs on (notebook shape: 9 loads per k-step, 8 of them these: 45 us)
  const int lane = threadIdx.x & 63, ks = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int i = blockIdx.x * 64 + lane;
  const int c0 = blockIdx.y * CH;
  double acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = 0.0;
  if (i < n) {
    // UB k-steps per batch with ALL their loads issued before the first FMA (UB + UB * CH independent loads in flight): left to the
    // compiler's unrolling, every k-step waited for its own loads -- 38 dependent round trips of ~1.2 us for the notebook shape
    // (45 us of a 515-us evaluation).  16 waves per workgroup leave 128 registers per lane: UB = 4 there.
    constexpr int UB = NW >= 16 ? 4 : 8;
    int cc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) cc[c] = (c0 + c < C) ? (c0 + c) : (C - 1);
    for (int k0 = ks; k0 < K; k0 += UB * NW) {
      double v[UB], mx[UB][CH];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = k0 + u * NW;
        const int kk = k < K ? k : K - 1;
        v[u] = In[(size_t)kk * ldi + i];
        if (k >= K) v[u] = 0.0;
#pragma unroll
        for (int c = 0; c < CH; ++c) mx[u][c] = Mx[(size_t)kk * ldm + cc[c]];
      }
#pragma unroll
      for (int u = 0; u < UB; ++u)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_fma(mx[u][c], v[u], acc[c]);
    }
  }
  if (ks > 0) {
#pragma unroll
    for (int c = 0; c < CH; ++c) red[ks - 1][c][lane] = acc[c];
  }
  __syncthreads();
  double s = 0.0;
  if (ks == 0 && i < n) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      double val = acc[c];
#pragma unroll
      for (int w = 0; w < NW - 1; ++w) val += red[w][c][lane];
      if (c0 + c < C) {
        if (mode == 0) Out[(size_t)(c0 + c) * ldo + i] = val - (sub ? sub[c0 + c] : 0.0);
        else { const double r = Ref[(size_t)(c0 + c) * ldr + i] - val; s = __builtin_fma(r, r, s); }
      }
    }
  }
  if constexpr (NW == 4) {
    if (mode != 0) {
      const double tot = block_sum_256(s, sh);
      if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = tot;
    }
  }
}

// Deterministic final sum of `count` partials into out[0] (+ add).
__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int count,
                                                           double* __restrict__ out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int k = threadIdx.x; k < count; k += 256) s += partial[k];
  const double tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[0] = tot;
}

// ---------------------------------------------------------------------------------------------------
// Latent mean at xs from the weights:  mean[s] = mu + sum_i kappa(xs_s, x_i) alpha_i   (cross-Gram fused with the GEMV;
// never materialised).  The i range is cut into chunks of ichunk (blockIdx.y) whose partial sums strip_finish_kernel adds.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void post_mean_kernel(const double* __restrict__ xs, int ns,
                                                        const double* __restrict__ x, int n, int d, int ichunk,
                                                        const double* __restrict__ alpha, LatentDev g,
                                                        double* __restrict__ partial) {
  __shared__ double xa[256 * 2];
  const int s = blockIdx.x * 256 + threadIdx.x;
  const int ibeg = blockIdx.y * ichunk;
  int iend = ibeg + ichunk; if (iend > n) iend = n;
  double acc = 0.0;
  if (d == 1) {
    const double xv = (s < ns) ? xs[s] : 0.0;
    for (int i0 = ibeg; i0 < iend; i0 += 256) {
      __syncthreads();
      const int i = i0 + threadIdx.x;
      xa[threadIdx.x] = (i < iend) ? x[i] : 0.0;
      xa[256 + threadIdx.x] = (i < iend) ? alpha[i] : 0.0;
      __syncthreads();
      const int lim = (iend - i0 < 256) ? (iend - i0) : 256;
#pragma unroll 4
      for (int k = 0; k < lim; ++k) {
        const double r = fabs(xv - xa[k]) * g.inv_ls;
        acc = __builtin_fma(kappa(g.kind, g.var, r, r * r), xa[256 + k], acc);
      }
    }
  } else if (s < ns) {
    for (int i = ibeg; i < iend; ++i) {
      const double r2 = scaled_dist2(xs + (size_t)s * d, x + (size_t)i * d, d, g.inv_ls);
      acc = __builtin_fma(kappa(g.kind, g.var, sqrt(r2), r2), alpha[i], acc);
    }
  }
  partial[(size_t)blockIdx.y * ((size_t)gridDim.x * 256) + s] = acc;
}

// ---------------------------------------------------------------------------------------------------
// K4: H unprojection of latent marginals / samples.  out[s + o*ns] (+)= sum_l Hm[o,l]^pw * lat[s + l*ns] (+ add)
//   pw = 1: means / samples;  pw = 2: variances (abs2.(H) * V).   Reference src/oilmm.jl:69,72, src/ilmm.jl:86.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mix_kernel(const double* __restrict__ lat, int ns, int ml,
                                                  const double* __restrict__ Hm, int p, int pw,
                                                  double lat_add, double out_add,
                                                  const double* __restrict__ eps, double eps_scale,
                                                  double* __restrict__ out) {
  const int s = blockIdx.x * 256 + threadIdx.x;
  const int o = blockIdx.y;
  if (s >= ns) return;
  double acc = out_add;
  for (int l = 0; l < ml; ++l) {
    double h = Hm[o + (size_t)l * p];
    if (pw == 2) h = h * h;
    acc = __builtin_fma(h, lat[(size_t)l * ns + s] + lat_add, acc);
  }
  if (eps != nullptr) acc = __builtin_fma(eps_scale, eps[(size_t)o * ns + s], acc);
  out[(size_t)o * ns + s] = acc;
}

// K4, bf16 projection (BASELINE configs[3]: "bf16 MFMA covariance projection"; lmm_set_projection_dtype): the same unprojection
//     out[s + o*ns] = out_add + sum_l Hm[o,l]^pw * (lat[s + l*ns] + lat_add)          (reference src/oilmm.jl:69,72)
// on v_mfma_f32_16x16x32_bf16: both operands are rounded to bfloat16 (round-to-nearest-even of the Float32 image; for pw = 2 the
// SQUARED entry abs2(H) is what is rounded, as abs2.(H) * V in the reference's order of operations), products and the sum over l
// accumulate in Float32 on the matrix pipe, out_add (sigma2) is added in Float64.  TERMS = 1: plain bf16 (relative input error
// <= 2^-8 each, so |error| <= ~2^-7 sum_l |H^pw| |lat|); TERMS = 2: each operand split hi + lo into two bf16 values and the three
// leading products hi hi + hi lo + lo hi summed (~2^-16: Float32-class), for callers who want the bf16 pipe without the loss.
// Lane map (cdna_hip_programming.md section 3): lane l holds A[row l&15][k = 8(l>>4) + j], B[k = 8(l>>4) + j][col l&15], j = 0..7,
// D[row 4(l>>4) + r][col l&15].  A = H^pw (16 outputs x 32 latents), B = lat (32 latents x 16 points): D's lanes run along the
// points, so every store is eight 128-byte segments.  One wave per 16 points, looping over the output tiles.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
template <int TERMS>
__global__ __launch_bounds__(256) void mix_bf16_kernel(const double* __restrict__ lat, int ns, int ml,
                                                       const double* __restrict__ Hm, int p, int pw,
                                                       double lat_add, double out_add, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int s0 = (blockIdx.x * 4 + w) * 16;
  if (s0 >= ns) return;                                            // whole wave
  const int c = lane & 15, kg = lane >> 4;
  const int s = s0 + c;
  for (int o0 = 0; o0 < p; o0 += 16) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < ml; k0 += 32) {
      bf16x8 a[TERMS], b[TERMS];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int l = k0 + 8 * kg + j;
        double hv = 0.0, bv = 0.0;
        if (l < ml) {
          if (o0 + c < p) { hv = Hm[(o0 + c) + (size_t)l * p]; if (pw == 2) hv = hv * hv; }
          if (s < ns) bv = lat[(size_t)l * ns + s] + lat_add;
        }
        const unsigned short ah = f32_to_bf16_rne((float)hv), bh = f32_to_bf16_rne((float)bv);
        a[0][j] = (short)ah; b[0][j] = (short)bh;
        if (TERMS == 2) {
          a[1][j] = (short)f32_to_bf16_rne((float)(hv - (double)bf16_to_f32(ah)));
          b[1][j] = (short)f32_to_bf16_rne((float)(bv - (double)bf16_to_f32(bh)));
        }
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
      if (TERMS == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
      }
    }
    if (s < ns) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + 4 * kg + r;
        if (o < p) out[(size_t)o * ns + s] = out_add + (double)acc[r];
      }
    }
  }
}

// Dense-H posterior full covariance (reference src/ilmm.jl:132-139 with coupled latents): C = H_full S H_full' + sigma2 I
// for the latent joint covariance S ((m ns) x (m ns), lower triangle of a factor-layout buffer, index l*ns + i), in two
// passes:  T[(o,i), (l',j)] = sum_l H[o,l] S[(l,i),(l',j)]   then   C[(o,i),(o',j)] = sum_l' T[(o,i),(l',j)] H[o',l'].
template <typename TS>
__global__ __launch_bounds__(256) void dense_cov_half_kernel(const void* __restrict__ S, int lds, int ns, int m,
                                                             const double* __restrict__ Hm, int p, double jitter,
                                                             double* __restrict__ T) {
  const int a = blockIdx.x * 256 + threadIdx.x;      // (o, i)
  const int c = blockIdx.y;                          // (l', j)
  if (a >= p * ns) return;
  const int o = a / ns, i = a - o * ns;
  double acc = 0.0;
  for (int l = 0; l < m; ++l) {
    const int r = l * ns + i;
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    double v = MatIO<TS>::ld1(S, (size_t)lo * lds + hi);
    if (r == c) v += jitter;
    acc = __builtin_fma(Hm[o + (size_t)l * p], v, acc);
  }
  T[(size_t)c * ((size_t)p * ns) + a] = acc;
}

__global__ __launch_bounds__(256) void dense_cov_full_kernel(const double* __restrict__ T, int ns, int m,
                                                             const double* __restrict__ Hm, int p, double sigma2,
                                                             double* __restrict__ out) {
  const int a = blockIdx.x * 256 + threadIdx.x;      // (o, i): row
  const int b = blockIdx.y;                          // (o', j): column
  if (a >= p * ns) return;
  const int o2 = b / ns, j = b - o2 * ns;
  double acc = (a == b) ? sigma2 : 0.0;
  for (int l = 0; l < m; ++l) acc = __builtin_fma(T[(size_t)(l * ns + j) * ((size_t)p * ns) + a], Hm[o2 + (size_t)l * p], acc);
  out[(size_t)b * ((size_t)p * ns) + a] = acc;
}

// Full mixed covariance (reference src/ilmm.jl:132-139 / AbstractGPs cov):
//   out[(o,i),(o',j)] (+)= sum_{l in chunk} H[o,l] H[o',l] (C_l[i,j] + jitter [i==j])  (+ sigma2 [o==o', i==j] on init)
// C_l is the lower triangle of a factor-matrix-layout buffer (mirrored here).  out is (p ns) x (p ns) column-major.
template <typename TS>
__global__ __launch_bounds__(256) void cov_mix_kernel(BatchPtr Cl, int ldcl, int nl, const double* __restrict__ Hs, int p,
                                                      int ns, double jitter, double sigma2, int init,
                                                      double* __restrict__ out) {
  const int i = blockIdx.x * 16 + (threadIdx.x & 15), j = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (i >= ns || j >= ns) return;
  const int o = blockIdx.z / p, o2 = blockIdx.z - o * p;
  double acc = 0.0;
  const int hi = i > j ? i : j, lo = i > j ? j : i;
  for (int l = 0; l < nl; ++l) {
    double c = MatIO<TS>::ld1(Cl.p[l], (size_t)lo * ldcl + hi);
    if (i == j) c += jitter;
    acc = __builtin_fma(Hs[o + (size_t)l * p] * Hs[o2 + (size_t)l * p], c, acc);
  }
  double* q = out + ((size_t)o2 * ns + j) * ((size_t)p * ns) + (size_t)o * ns + i;
  if (init) *q = acc + ((o == o2 && i == j) ? sigma2 : 0.0);
  else *q += acc;
}

// K7: the sample transform  out = mu + L z  is strip_reduce_kernel<false, true> + strip_finish_kernel (above).

__global__ void vec_lin_kernel(const double* a, const double* b, double sb, int n, double* out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = a[k] + sb * b[k];
}
__global__ void vec_axpby_kernel(const double* a, double sa, const double* b, double sb, size_t n, double* out) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = sa * a[k] + sb * b[k];
}

// R = I on an nc x nc block (ld), zero elsewhere: the riders whose triangular solve gives L^-T.
template <typename TS>
__global__ void set_identity_kernel(void* __restrict__ R, int ld, int nc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  if (i < nc) MatIO<TS>::st1(R, (size_t)j * ld + i, (i == j) ? 1.0 : 0.0);
}

// d kappa / d lengthscale at scaled distance r (r2 = r^2):  SE v e^{-r2/2} r2 / ell;  Matern32 v s^2 e^{-s} / ell;
// Matern52 v e^{-s} (s^2/3)(1+s) / ell   (s = sqrt(3) r, sqrt(5) r).
__device__ __forceinline__ double dkappa_dell(int kind, double var, double inv_ls, double r, double r2) {
  if (kind == LMM_KERNEL_SE) return var * exp_nonpos(-0.5 * r2) * r2 * inv_ls;
  if (kind == LMM_KERNEL_MATERN32) { const double s = 1.7320508075688772 * r; return var * s * s * exp_nonpos(-s) * inv_ls; }
  const double s = 2.23606797749979 * r;
  return var * exp_nonpos(-s) * (s * s / 3.0) * (1.0 + s) * inv_ls;
}

// Gradient contractions of one latent: per 64x64 lower tile (ti >= tj) of Kinv
//   partial[NG*tile + 0] = sum_{i>j in tile} (alpha_i alpha_j - Kinv_ij) dK_ij/d ell        (lengthscale)
//   partial[NG*tile + 1], [5] = sum_i Kinv_ii over the tile's rows i < nsplit, i >= nsplit    (trace of the inverse, per noise block)
//   partial[NG*tile + 2], [6] = alpha.alpha over the same two row ranges
//   partial[NG*tile + 3..4]   = alpha.delta, sum alpha over the tile's rows                   (diagonal tiles only)
//   partial[NG*tile + 7]      = sum_{i>j in tile} (alpha_i alpha_j - Kinv_ij) K_ij            (variance, when Kinv is a block of a larger inverse)
// The split at nsplit serves the predictive logpdf (joint of training and test points, each block with its own noise).
#define LMM_NG 8
template <typename TS>      // storage type of the inverse Kinv (a MATRIX: Float32 in the fp32 compute mode); all sums in Float64
__global__ __launch_bounds__(256) void grad_reduce_kernel(const void* __restrict__ Kinv, int ld, int n, int nsplit,
                                                          const double* __restrict__ alpha, const double* __restrict__ delta,
                                                          const double* __restrict__ x, int d, LatentDev g, int nt,
                                                          double* __restrict__ partial) {
  __shared__ double sh[4];
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (ti < tj) return;
  const int t = threadIdx.x;
  const int i0 = ti * 64 + (t & 63);
  const int cg = t >> 6;
  double acc = 0.0, acck = 0.0;
  if (i0 < n) {
    const double ai = alpha[i0];
    for (int q = 0; q < 16; ++q) {
      const int j = tj * 64 + cg + 4 * q;
      if (j < i0 && j < n) {
        double r, r2;
        if (d == 1) { r = fabs(x[i0] - x[j]) * g.inv_ls; r2 = r * r; }
        else { r2 = scaled_dist2(x + (size_t)i0 * d, x + (size_t)j * d, d, g.inv_ls); r = sqrt(r2); }
        const double w = ai * alpha[j] - MatIO<TS>::ld1(Kinv, (size_t)j * ld + i0);
        acc = __builtin_fma(w, dkappa_dell(g.kind, g.var, g.inv_ls, r, r2), acc);
        acck = __builtin_fma(w, kappa(g.kind, g.var, r, r2), acck);
      }
    }
  }
  const int tile = ti * nt + tj;
  const double tl = block_sum_256(acc, sh);
  const double tk = block_sum_256(acck, sh);
  double tra = 0.0, aaa = 0.0, trb = 0.0, aab = 0.0, ad = 0.0, sa = 0.0;
  if (ti == tj && t < 64 && i0 < n) {
    const double ai = alpha[i0], kii = MatIO<TS>::ld1(Kinv, (size_t)i0 * ld + i0);
    if (i0 < nsplit) { tra = kii; aaa = ai * ai; } else { trb = kii; aab = ai * ai; }
    ad = ai * delta[i0]; sa = ai;
  }
  const double s1 = block_sum_256(tra, sh), s2 = block_sum_256(aaa, sh), s3 = block_sum_256(ad, sh), s4 = block_sum_256(sa, sh);
  const double s5 = block_sum_256(trb, sh), s6 = block_sum_256(aab, sh);
  if (t == 0) {
    double* o = partial + (size_t)LMM_NG * tile;
    o[0] = tl; o[1] = s1; o[2] = s2; o[3] = s3; o[4] = s4; o[5] = s5; o[6] = s6; o[7] = tk;
  }
}

// out[c] = sum over the nt*nt tile partials of component c (c < LMM_NG); upper tiles were never written -> skip them.
__global__ __launch_bounds__(256) void grad_finish_kernel(const double* __restrict__ partial, int nt, double* __restrict__ out) {
  __shared__ double sh[4];
  for (int c = 0; c < LMM_NG; ++c) {
    double s = 0.0;
    for (int k = threadIdx.x; k < nt * nt; k += 256) {
      const int ti = k / nt, tj = k - ti * nt;
      if (ti >= tj) s += partial[(size_t)LMM_NG * k + c];
    }
    const double tot = block_sum_256(s, sh);
    if (threadIdx.x == 0) out[c] = tot;
  }
}

// out[l + l2*m] = sum_i Minv[(l n + i), (l2 n + i)]  for l >= l2 (mirrored into l < l2): the m x m matrix of traces of the diagonals of
// the n x n blocks of a symmetric (m n) x (m n) matrix whose lower triangle is stored (dense-H ILMM gradient: dL/dSigmaT).
template <typename TS>
__global__ __launch_bounds__(256) void block_trace_kernel(const void* __restrict__ Minv, int ld, int n, int m, int i0, int i1,
                                                          double* __restrict__ out) {
  __shared__ double sh[4];
  const int l = blockIdx.x, l2 = blockIdx.y;
  if (l < l2) return;
  double s = 0.0;
  for (int i = i0 + threadIdx.x; i < i1; i += 256) s += MatIO<TS>::ld1(Minv, (size_t)(l2 * n + i) * ld + (l * n + i));
  const double tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) { out[l + (size_t)l2 * m] = tot; out[l2 + (size_t)l * m] = tot; }
}

// out[k] = a[k] + (num / s2[block of row(k)]) * b[k]  with row(k) = k mod N  (column-major N x p operands; blocks: NoiseBlocks)
__global__ void vec_lin_blocks_kernel(const double* a, const double* b, NoiseBlocks nb, double num, int N, size_t count, double* out) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  const int row = (int)(k % N);
  int blk = 0;
  while (blk + 1 < nb.nblk && row >= nb.off[blk + 1]) ++blk;
  out[k] = a[k] + (num / nb.s2[blk]) * b[k];
}

// out[a + b*na] = sum_i X[i + a*ldx] Z[i + b*ldz]   (X' Z for tall-skinny X (n x na), Z (n x nb)); one block per (a, b).
__global__ __launch_bounds__(256) void atb_kernel(const double* __restrict__ X, int ldx, const double* __restrict__ Z, int ldz,
                                                  int n, int na, double* __restrict__ out) {
  __shared__ double sh[4];
  const int a = blockIdx.x, b = blockIdx.y;
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s = __builtin_fma(X[(size_t)a * ldx + i], Z[(size_t)b * ldz + i], s);
  const double tot = block_sum_256(s, sh);
  if (threadIdx.x == 0) out[a + (size_t)b * na] = tot;
}

__global__ void fill_kernel(double* __restrict__ p, int n, double v) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) p[k] = v;
}

// by-features <-> by-outputs reordering (an n x p transpose): reference src/independent_mogp.jl:135-159
__global__ void reorder_kernel(const double* __restrict__ in, int n, int p, int to_outputs, double* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n * p) return;
  if (to_outputs) { const int o = k / n, i = k - o * n; out[k] = in[(size_t)i * p + o]; }
  else { const int i = k / p, o = k - i * p; out[k] = in[(size_t)o * n + i]; }
}

// One diagonal block of cov(f::IndependentMOGP, x, y) (reference src/independent_mogp.jl:66-71; :184-215 for the by-features orders):
// out[(row0 + i rs) + (col0 + j cs) ldo] = src[i + j lds], i < nr, j < nc.  by-outputs: row0 = l nr, rs = 1; by-features: row0 = l, rs = m.
template <typename TS>
__global__ __launch_bounds__(256) void block_scatter_kernel(const void* __restrict__ src, int lds, int nr, int nc, double* __restrict__ out,
                                                            size_t ldo, size_t row0, int rs, size_t col0, int cs) {
  const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (i < nr && j < nc) out[(row0 + (size_t)i * rs) + (col0 + (size_t)j * cs) * ldo] = MatIO<TS>::ld1(src, (size_t)j * lds + i);
}

// ---------------------------------------------------------------------------------------------------
// K7 (optional): standard normals on the device -- Philox4x32-10 counter RNG (Salmon et al. 2011) + Box-Muller in Float64.
// Normal 2j and 2j+1 come from counter (j, stream) under key `seed`, so a buffer is reproducible for (seed, stream) whatever
// the launch shape.  The reference draws with Julia's MersenneTwister on the host; this is for callers that do not need
// the reference's random stream (the draw ORDER of rand is kept by the caller: latent normals first, then noise normals).
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(unsigned int c[4], unsigned int k0, unsigned int k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned int hi0 = (unsigned int)(p0 >> 32), lo0 = (unsigned int)p0, hi1 = (unsigned int)(p1 >> 32), lo1 = (unsigned int)p1;
    const unsigned int n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void normals_kernel(unsigned long long seed, unsigned long long stream, size_t count, double* __restrict__ out) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;       // pair index
  if (2 * j >= count) return;
  unsigned int c[4] = {(unsigned int)j, (unsigned int)(j >> 32), (unsigned int)stream, (unsigned int)(stream >> 32)};
  philox4x32_10(c, (unsigned int)seed, (unsigned int)(seed >> 32));
  // two uniforms in (0, 1] / [0, 1) with 53 random bits each
  const double u1 = ((double)(((unsigned long long)(c[0] >> 5) << 26) | (c[1] >> 6)) + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = (double)(((unsigned long long)(c[2] >> 5) << 26) | (c[3] >> 6)) * (1.0 / 9007199254740992.0);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  out[2 * j] = rad * cs;
  if (2 * j + 1 < count) out[2 * j + 1] = rad * sn;
}

// ---------------------------------------------------------------------------------------------------
// f64 MFMA issue-rate microbenchmark (the guide gives no FP64 matrix peak).
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double* out, int iters) {
  // v_mfma_f64_16x16x4_f64 exactly as the update kernels issue it: 16 accumulators of 16 x 16 in architectural VGPRs (the library is
  // compiled with -amdgpu-mfma-vgpr-form=1), 4 + 4 operand fragments, serpentine order; one wave per SIMD is enough (64-cycle MFMAs)
  d4 acc[4][4];
  double fa[4], fb[4];
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[v][u] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int q = 0; q < 4; ++q) { fa[q] = 1.0 + (threadIdx.x + 7 * q) * 1e-3; fb[q] = 1.0 - (threadIdx.x + 3 * q) * 1e-3; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        const int u = (v & 1) ? 3 - uu : uu;
        acc[v][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[v], fa[u], acc[v][u], 0, 0, 0);
      }
  }
  double s = 0.0;
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int u = 0; u < 4; ++u) s += acc[v][u][0] + acc[v][u][1] + acc[v][u][2] + acc[v][u][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------
// launch wrappers (host)
// ---------------------------------------------------------------------------------------------------
// Compute dtype of the matrices the launches below touch: 0 = Float64, 1 = Float32 (lmm_set_compute_dtype; set by the API
// layer under its context lock).  Every matrix pointer (BatchPtr entries, GramArgs.A, ...) is then a float buffer; offsets and
// leading dimensions stay in elements.
int g_f32 = 0;
#define LMM_TS_LAUNCH(KERNEL_T, ...)                                                   \
  do {                                                                                 \
    if (g_f32) { using TS = float; hipLaunchKernelGGL(KERNEL_T, __VA_ARGS__); }        \
    else { using TS = double; hipLaunchKernelGGL(KERNEL_T, __VA_ARGS__); }             \
  } while (0)

// Column tiles per workgroup: 4 (a strip's row data and row exponentials are reused across them), or 1 when that grid would leave
// most of the device idle (small matrices: C0's 22-us Gram was 15 workgroups walking up to four tiles each).
static int gram_cpw(int row_tiles, int col_tiles, int nmat) {
  return (long long)row_tiles * ((col_tiles + 3) / 4) * nmat < 2048 ? 1 : 4;
}
void launch_gram(const GramArgs& a0, hipStream_t st) {
  GramArgs a = a0;
  a.cpw = gram_cpw(a.nrows / 64 - a.row_tile0, a.ncols / 64, 1);
  dim3 grid(a.nrows / 64 - a.row_tile0, (a.ncols / 64 + a.cpw - 1) / a.cpw);
  const bool nd = (a.d > 1 && a.d <= 8);
#define LMM_GRAM_LAUNCH(K)                                                                          \
  do {                                                                                              \
    if (nd) LMM_TS_LAUNCH((gram_kernel<K, true, TS>), grid, dim3(256), 0, st, a);                   \
    else LMM_TS_LAUNCH((gram_kernel<K, false, TS>), grid, dim3(256), 0, st, a);                     \
  } while (0)
  if (a.kind == LMM_KERNEL_SE) LMM_GRAM_LAUNCH(LMM_KERNEL_SE);
  else if (a.kind == LMM_KERNEL_MATERN32) LMM_GRAM_LAUNCH(LMM_KERNEL_MATERN32);
  else LMM_GRAM_LAUNCH(LMM_KERNEL_MATERN52);
#undef LMM_GRAM_LAUNCH
}

void launch_gram_batch(const GramArgs* args, int nb, hipStream_t st) {
  int j0 = 0;
  while (j0 < nb) {
    int j1 = j0 + 1;
    while (j1 < nb && args[j1].kind == args[j0].kind) ++j1;
    if (j1 - j0 == 1) { launch_gram(args[j0], st); j0 = j1; continue; }
    GramBatchArgs b{};
    b.base = args[j0];
    for (int j = j0; j < j1; ++j) {
      const GramArgs& a = args[j];
      b.A[j - j0] = a.A; b.var[j - j0] = a.var; b.inv_ls[j - j0] = a.inv_ls; b.diag_add[j - j0] = a.diag_add;
      b.diag_vec[j - j0] = a.diag_vec; b.rider[j - j0] = a.rider; b.rider_sub[j - j0] = a.rider_sub; b.info_zero[j - j0] = a.info_zero;
    }
    b.base.cpw = gram_cpw(b.base.nrows / 64 - b.base.row_tile0, b.base.ncols / 64, j1 - j0);
    const GramArgs& a = b.base;
    dim3 grid(a.nrows / 64 - a.row_tile0, (a.ncols / 64 + a.cpw - 1) / a.cpw, j1 - j0);
    const bool nd = (a.d > 1 && a.d <= 8);
#define LMM_GRAM_LAUNCH(K)                                                                          \
    do {                                                                                            \
      if (nd) LMM_TS_LAUNCH((gram_batch_kernel<K, true, TS>), grid, dim3(256), 0, st, b);           \
      else LMM_TS_LAUNCH((gram_batch_kernel<K, false, TS>), grid, dim3(256), 0, st, b);             \
    } while (0)
    if (a.kind == LMM_KERNEL_SE) LMM_GRAM_LAUNCH(LMM_KERNEL_SE);
    else if (a.kind == LMM_KERNEL_MATERN32) LMM_GRAM_LAUNCH(LMM_KERNEL_MATERN32);
    else LMM_GRAM_LAUNCH(LMM_KERNEL_MATERN52);
#undef LMM_GRAM_LAUNCH
    j0 = j1;
  }
}

void launch_dense_cross(double* R, int ldr, int nrows, int ncols, const double* xs, int ns, const double* x, int n, int d,
                        int m, const LatentDev* lat, hipStream_t st) {
  dim3 grid((nrows + 255) / 256, ncols);
  LMM_TS_LAUNCH((dense_cross_kernel<TS>), grid, dim3(256), 0, st, (void*)R, ldr, nrows, ncols, xs, ns, x, n, d, m, lat);
}

int dense_var_kc(int Ncols) { int kc = ((Ncols + 127) / 128 + 63) / 64 * 64; return kc < 64 ? 64 : kc; }   // <= 128 chunks
size_t dense_var_partial_elems(int ns, int p, int Ncols) {
  const int kc = dense_var_kc(Ncols);
  return (size_t)((Ncols + kc - 1) / kc) * p * ns;
}

void launch_dense_var(const double* R, int ldr, int ns, int m, int Ncols, const double* Hm, int p, const LatentDev* lat,
                      double jitter, double sigma2, double* partial, double* out, hipStream_t st) {
  const int kc = dense_var_kc(Ncols), nch = (Ncols + kc - 1) / kc, ne = p * ns;
  LMM_TS_LAUNCH((dense_var_kernel<TS>), dim3((ne + 255) / 256, nch), dim3(256), 0, st, (const void*)R, ldr, ns, m, Ncols, kc, Hm, p, partial);
  hipLaunchKernelGGL(dense_var_finish_kernel, dim3((ne + 255) / 256), dim3(256), 0, st, partial, nch, ns, m, Hm, p, lat, jitter,
                     sigma2, out);
}

// C (p ns x p ns, column-major, by-outputs index o*ns + i) from the latent joint covariance S; T: (p ns) x (m ns) scratch.
void launch_dense_cov(const double* S, int lds, int ns, int m, const double* Hm, int p, double jitter, double sigma2, double* T,
                      double* out, hipStream_t st) {
  const int na = p * ns;
  LMM_TS_LAUNCH((dense_cov_half_kernel<TS>), dim3((na + 255) / 256, m * ns), dim3(256), 0, st, (const void*)S, lds, ns, m, Hm, p, jitter, T);
  hipLaunchKernelGGL(dense_cov_full_kernel, dim3((na + 255) / 256, na), dim3(256), 0, st, T, ns, m, Hm, p, sigma2, out);
}

void launch_dense_assemble(const DenseArgs& a, hipStream_t st) {
  dim3 grid(a.nrows / 64, a.ncols / 64);
  LMM_TS_LAUNCH((ilmm_dense_assemble_kernel<TS>), grid, dim3(256), 0, st, a);      // fp32 compute mode: Float32 matrix (the dense logpdf paths)
}

// ---- round 3: 128-column panels (leaf128 / bulk) and the update fused with the next panel's leaf (K2c) ----
static void node_lds_attr() {
  static bool done = false;
  if (done) return;
  const int bytes = LEAF_LDS_DOUBLES * 8;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(leaf128_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_node_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_node_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_node_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_node_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_region_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_region_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_region_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_region_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  done = true;
}
// Strict-progress mode (default): two sets of claim counters per stream, used by alternate region launches of that stream (region_claim).
int g_strict_progress = 1;
int g_claim_scramble = 0;
static std::mutex g_claim_mu;
struct ClaimSets { unsigned* dev; int parity; };
static std::map<hipStream_t, ClaimSets> g_claims;
// after an error drained the device (lmm_api.hip's drain_after_error: a launch that never ran zeroed nothing) and at lmm_shutdown (the
// streams the counters are keyed by go away): drop the counters; the next strict launch of a stream makes fresh ones
void strict_ticket_reset() {
  std::lock_guard<std::mutex> lock(g_claim_mu);
  for (auto& kv : g_claims) (void)hipFree(kv.second.dev);              // (the caller has synchronised the device)
  g_claims.clear();
}
static bool strict_claim_sets(hipStream_t st, unsigned** cur, unsigned** next) {
  std::lock_guard<std::mutex> lock(g_claim_mu);
  auto it = g_claims.find(st);
  if (it == g_claims.end()) {
    unsigned* d = nullptr;
    if (hipMalloc((void**)&d, 2 * LMM_CLAIM_INTS * sizeof(unsigned)) != hipSuccess) { (void)hipGetLastError(); return false; }
    (void)hipMemsetAsync(d, 0, 2 * LMM_CLAIM_INTS * sizeof(unsigned), st);      // on the stream whose launches use it: ordered before the first
    it = g_claims.emplace(st, ClaimSets{d, 0}).first;
  }
  *cur = it->second.dev + LMM_CLAIM_INTS * it->second.parity;
  *next = it->second.dev + LMM_CLAIM_INTS * (it->second.parity ^ 1);
  it->second.parity ^= 1;
  return true;
}
void launch_leaf128(const BatchPtr& A, size_t offD, int ld, const BatchPtr& W, size_t offW, const BatchPtr& W2, size_t offW2,
                    int gcol0, int n_real, const BatchInfo& info, int nb, hipStream_t st) {
  node_lds_attr();
  hipLaunchKernelGGL(leaf128_kernel, dim3(nb), dim3(256), LEAF_LDS_DOUBLES * 8, st, A, offD, ld, W, offW, W2, offW2, gcol0, n_real, info);
}
void launch_panel_bulk(const BatchPtr& A, const BatchPtr& W2, int ld, int NR, int r0, int nb, hipStream_t st) {
  const int M = NR - r0, MT = (M + 127) / 128;
  if (MT <= 1 || nb <= 0) return;
  node_lds_attr();
  NodeArgs a{};
  a.A = A; a.W2 = W2; a.ld = ld; a.M = M; a.Mb = M; a.j0 = r0; a.h = 0; a.N = 128; a.MT = MT; a.MTb = MT; a.nb = nb; a.mode = NODE_BULK;
  hipLaunchKernelGGL(potrf_node_kernel<1>, dim3((unsigned)nb * (MT - 1)), dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
}
static int g_region_epoch = 0;
static int* g_region_flags_base = nullptr;       // the persistent flag array of the context (lmm_init), for the wrap-around clear
static size_t g_region_flags_ints = 0;
void region_flags_register(int* base, size_t ints) { g_region_flags_base = base; g_region_flags_ints = ints; }
// Every flag-carrying launch (region, fused node) takes a fresh tag.  When the tags wrap, every persistent flag word is cleared, so that
// no stale tag can match again (the per-call NODE_FUSE flags are zeroed at the start of their factorisation anyway).
static int next_flag_epoch() {
  if (++g_region_epoch >= (1 << 26)) {
    g_region_epoch = 1;
    if (g_region_flags_base) { (void)hipDeviceSynchronize(); (void)hipMemset(g_region_flags_base, 0, g_region_flags_ints * sizeof(int)); }
  }
  return g_region_epoch;
}
// test hook (lmm_dev_flag_epoch): read the launch-epoch counter, and set it when set_to >= 0 -- to just below 2^26, so that a test
// executes the wrap-around clear above
int region_flag_epoch(int set_to) {
  const int old = g_region_epoch;
  if (set_to >= 0) g_region_epoch = set_to;
  return old;
}
int g_concurrent_batches = 1;        // batches in flight on the slot streams (lmm_api.hip's fork_slots)
size_t node_flag_ints(int NR) { return (size_t)(2 + (NR + 127) / 128 + 1); }
bool launch_update_leaf(const BatchPtr& A, const BatchPtr& W, const BatchPtr& W2, const BatchInfo& info, int ld, int NR, int j0, int h,
                        int N, int n_real, int nb, hipStream_t st, int* nflags, int nf_stride) {
  const int r0 = j0 + h, NT = (N + 127) / 128;     // N may end in a 64-column half tile
  if (nb <= 0 || N <= 0 || h <= 0) return false;
  node_lds_attr();
  // ragged last row tile (the rider rows make the row count an odd multiple of 64): on the long products the node kernel takes the
  // full 128-row tiles and gemm16h_kernel the last 64 rows, as in launch_gemm_nt (two idle waves for a whole tile time otherwise)
  int M = NR - r0;
  const bool strip = (M % 128) == 64 && M - 64 >= N && (N % 128) == 0 && h >= 1024;
  const int Mfull = M;
  if (strip) M -= 64;
  const int MT = (M + 127) / 128;
  const int MTb = (Mfull + 127) / 128;
  // the ragged rows first when the bulk tiles ride along: they cover those rows too (the two launches touch disjoint entries)
  const bool fuse = nflags != nullptr && N >= 128 && MTb > 1;
  auto launch_strip = [&]() {
    const size_t offA = (size_t)j0 * ld + r0 + (size_t)M;
    hipLaunchKernelGGL((gemm16h_kernel<true>), dim3(N / 128, nb), dim3(256), 0, st, A, (size_t)r0 * ld + r0 + (size_t)M, ld, A, offA, ld,
                       A, (size_t)j0 * ld + r0, ld, N, h, 0);
  };
  static int strip_items = -1;                     // LMM_STRIP_ITEMS=0: the ragged rows as a separate launch (the round-2/3 form)
  if (strip_items < 0) { const char* e = getenv("LMM_STRIP_ITEMS"); strip_items = e ? atoi(e) : 1; }
  // ... and only while no other batch runs beside this one: with two batches in flight (C2 at N = 1) the other stream's kernels fill the
  // CUs a separate strip launch leaves idle, and the in-launch items measured slower (687.2 -> 689.9 ms per step)
  const bool strip_in = strip && strip_items && (g_concurrent_batches <= 1 || strip_items == 2);
  if (strip && !strip_in && fuse) launch_strip();
  static int cus = 0;
  if (cus == 0) { int dev = 0; cus = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
  long long T = 0;                                            // tiles of the column tiles 1 .. NT-1 (lower trapezoid)
  for (int tj = 1; tj < NT; ++tj) T += MT - tj;
  // Split-K tail.  The matrices of the batch run back to back (the next matrix's tiles fill the CUs as the previous one's drain), so
  // only the END of the launch can leave CUs idle: up to a whole tile time with unsplit tiles.  cap = workgroups resident at once
  // (two per CU).  (a) a launch that does not even fill half of that is split uniformly; (b) otherwise the LAST cap / s tiles of
  // the last matrix are split s ways, so that the final cap work items are short (drain ~ a tile time / s) and nothing else pays
  // for atomics.  LMM_DETERMINISTIC=1: no split (bitwise reproducible sums).
  const int cap = 2 * cus, nk = h / 16;
  static int deterministic = -1;
  if (deterministic < 0) { const char* e = getenv("LMM_DETERMINISTIC"); deterministic = (e && atoi(e) != 0) ? 1 : 0; }
  int full_a = (int)T, split_a = 1, full_l = (int)T, split_l = 1;
  if (!deterministic && T > 0 && nk >= 8) {
    const long long all = (long long)nb * T + (long long)nb * MT;
    if (all <= cap / 2) {
      int sk = (int)(cap / all); if (sk > nk / 4) sk = nk / 4; if (sk < 1) sk = 1;
      if (sk > 1) { full_a = full_l = 0; split_a = split_l = sk; }
    } else if (all <= 2LL * cap && nk >= 32) {
      // (a') up to two rounds of whole tiles: every tile in two K halves -- whole tiles pair up on some CUs (each at half rate) while
      // other CUs idle or run one; halves fill the slots evenly (8 x 2048, K = 1024: 268 -> 220 us; 8 x 3072: 588 -> 554; 8 x 4096's
      // three launches 2.06 -> 2.01 ms)
      full_a = full_l = 0; split_a = split_l = 2;
    } else {                                // (b): always the last cap / s tiles
      // (splitting exactly the tiles of the last partial "round" instead measured no better: equal tiles do not stay in lock-step --
      // of the two workgroups of a CU the older one gets the matrix pipe -- and the slots are 94-96 % busy either way; profiles/r03)
      int sk = nk >= 16 ? 4 : 2;
      long long tail = cap / sk; if (tail > T) tail = T;
      full_l = (int)(T - tail); split_l = sk;
    }
  }
  NodeArgs a{};
  a.A = A; a.W = W; a.W2 = W2; a.info = info; a.ld = ld; a.M = M; a.j0 = j0; a.h = h; a.N = N; a.n_real = n_real; a.MT = MT; a.nb = nb;
  a.full_items = full_a; a.splitk = split_a; a.rest_items = full_a + (int)(T - full_a) * split_a;
  a.full_items_last = full_l; a.splitk_last = split_l;
  const int rest_last = full_l + (int)(T - full_l) * split_l;
  if (a.rest_items < 1) a.rest_items = 1;          // divisor in the kernel's item decode (no item reaches it when T = 0)
  a.mode = NODE_UPDATE | NODE_LEAF;
  // the ragged rows' half-height items follow the column-0 tiles (at the END of the update part they lengthen the launch's tail:
  // 4 latents at n = 16384 90.15 ms against 89.69 here)
  a.strip_n = strip_in ? N / 128 : 0; a.strip0 = nb * MT;
  long long items = nb * MT + (long long)nb * a.strip_n + (T > 0 ? (long long)(nb - 1) * a.rest_items + rest_last : 0);
  a.Mb = Mfull; a.MTb = MTb; a.bulk0 = (int)items;
  if (fuse) {
    a.mode |= NODE_FUSE; a.nflags = nflags; a.nf_stride = nf_stride; a.epoch = next_flag_epoch();
    items += (long long)nb * (MTb - 1);
  }
  const dim3 grid((unsigned)items);
  if (fuse) {
    if (h >= 1024) hipLaunchKernelGGL((potrf_node_kernel<2, true>), grid, dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
    else hipLaunchKernelGGL((potrf_node_kernel<1, true>), grid, dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
  } else {
    if (h >= 1024) hipLaunchKernelGGL(potrf_node_kernel<2>, grid, dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
    else hipLaunchKernelGGL(potrf_node_kernel<1>, grid, dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
  }
  if (strip && !strip_in && !fuse) launch_strip();
  return fuse;
}

// Plan of a region launch's row tasks.  One workgroup per row tile, dispatched in order behind the square's workgroups, one per CU:
// with 128-row tiles throughout, a launch of a little more than a whole number of "waves" ends with most CUs idle for a chain's
// length (4 latents, n = 16384, first block column: 484 chains of ~640 us on 192 + 64 CUs: 1.50 ms against 1.21 ms of work), and a
// launch that fills less than the device is as long as one chain however few there are.  64-row tiles (half the chain, ~7 % more
// time per row: the square's rows are streamed twice as often) fix both when used for the LAST tasks only.  The split is chosen by
// simulating the dispatch (greedy list scheduling on `cus` CUs) with the measured task lengths for nine candidate splits.
struct RegionPlan { int n128, row_tasks, na; };
static RegionPlan region_plan(int P, int nb, int Mb, int Mb_real, int cus, int na_full, bool asst_always, int force_th) {
  const int T128 = (Mb + 127) / 128;
  struct Key { int P, nb, Mb, Mr, na, f, cus; bool operator<(const Key& o) const { return std::tie(P, nb, Mb, Mr, na, f, cus) < std::tie(o.P, o.nb, o.Mb, o.Mr, o.na, o.f, o.cus); } };
  static std::map<Key, RegionPlan> cache;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  const Key key{P, nb, Mb, Mb_real, na_full, force_th * 2 + (asst_always ? 1 : 0), cus};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const double unit = 17.4;                                 // us per 128^3 product on one CU inside this kernel (profiles/r04: chains of 36 units take 590-640 us)
  const double d128 = unit * P * (P + 1) / 2.0, d64 = 0.525 * d128;
  RegionPlan best{T128, T128, 0}; double best_t = 1e30;
  std::vector<int> xs;                                       // candidate splits, tallest first (ties keep the fewer workgroups)
  if (force_th == 128) xs.push_back(T128);
  else if (force_th == 64) xs.push_back(0);
  else { const int step = std::max(1, T128 / 32); for (int x = T128; x > 0; x -= step) xs.push_back(x); xs.push_back(0); }
  for (int pass = 0; pass < (asst_always ? 1 : 2); ++pass)        // pass 1: assistants although the rows are not all resident
  for (int x : xs) {
    const int rest = Mb - 128 * x;
    const int t64 = rest > 0 ? (rest + 63) / 64 : 0;
    if (x < T128 && t64 == 0) continue;
    const int ntask = x + t64;
    int full = 0;
    for (int k = 0; k < ntask; ++k) { const int roff = k < x ? 128 * k : 128 * x + 64 * (k - x); if (Mb_real - roff > 16) ++full; }
    const bool resident = (2LL * P + na_full + full) * nb <= cus;
    if (pass == 1 && (resident || na_full == 0 || (2LL * P + na_full) * nb > cus)) continue;
    const int na = (na_full > 0 && (asst_always || pass == 1 || resident)) ? na_full : 0;
    const double sq_end = 2.0 * P * (na ? 17.7 : 29.0);
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    const long long nsq = (2LL * P + na) * nb;
    for (int i = 0; i < cus; ++i) {
      if (i < nsq) { const int idx = i / nb; const int r = idx < LMM_REGION_ASST_MIN_R || !na ? idx : LMM_REGION_ASST_MIN_R + (idx - LMM_REGION_ASST_MIN_R) / 2; free_at.push(sq_end * (std::min(r, 2 * P - 1) + 1) / (2.0 * P)); }
      else free_at.push(0.0);
    }
    // (more square workgroups than CUs: the rule of potrf_batch keeps such launches on the panel path; treat the excess as row time)
    double end = sq_end;
    for (int k = 0; k < ntask; ++k) {
      const int roff = k < x ? 128 * k : 128 * x + 64 * (k - x);
      const int real = Mb_real - roff;
      if (real <= 0) continue;
      const double d = real > 16 ? (k < x ? d128 : d64) : 0.1 * d128;
      const double tail = real > 16 ? 2.0 * d / (P + 1) : 0.05 * d128;          // the last column follows the square's end
      for (int b = 0; b < nb; ++b) {
        const double t0 = free_at.top(); free_at.pop();
        const double t1 = std::max(t0 + d, sq_end + tail);
        free_at.push(t1);
        if (t1 > end) end = t1;
      }
    }
    if (end < best_t - 1e-9) { best_t = end; best = RegionPlan{x, ntask, na}; }
  }
  cache[key] = best;
  return best;
}

void region_plan_probe(int P, int nb, int Mb, int Mb_real, int cus, int na_full, int out[3]) {
  const RegionPlan r = region_plan(P, nb, Mb, Mb_real, cus, na_full, false, (Mb % 64) != 0 ? 128 : 0);
  out[0] = r.n128; out[1] = r.row_tasks; out[2] = r.na;
}
size_t region_flag_ints(int) { return REGION_FLAG_INTS; }
void launch_region(const BatchPtr& A, const BatchPtr& W, const BatchPtr& W2, const BatchInfo& info, const BatchInfo& flags, int ld, int NR,
                   int c0, int width, int n_real, int nb, bool first_done, hipStream_t st, int rows_real, const BatchPtr* S) {
  const int P = width / 128, M = NR - c0, R = (M + 127) / 128;
  if (nb <= 0 || P <= 0) return;
  node_lds_attr();
  RegionArgs a{};
  a.A = A; a.W = W; a.W2 = W2; a.info = info; a.flags = flags; a.ld = ld; a.M = M; a.c0 = c0; a.P = P; a.R = R; a.n_real = n_real; a.nb = nb;
  a.epoch = next_flag_epoch(); a.first_done = first_done ? 1 : 0;
  a.M_real = (rows_real >= 0 && rows_real <= NR) ? rows_real - c0 : M;
  // assistants (LMM_REGION_ASST=0 disables them): one per square row from LMM_REGION_ASST_MIN_R on, when the caller gave scratch
  static int asst_env = -1;
  if (asst_env < 0) { const char* e = getenv("LMM_REGION_ASST"); asst_env = e ? atoi(e) : 1; }       // 2: also when they are not resident from the start (measured slower)
  static int cus = 0;
  if (cus == 0) { int dev = 0; cus = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); }
  a.na = (asst_env && S != nullptr && S->p[0] != nullptr && 2 * P > LMM_REGION_ASST_MIN_R) ? 2 * P - LMM_REGION_ASST_MIN_R : 0;
  // ... and only while the launch with them still fits one workgroup per CU: pushed into the two-per-CU build (where the walker spills)
  // they cost more than they bring (8 latents, n = 2048: 1.60 -> 1.87 ms)
  // Row tiles that take the thin stream (or hold padding only) need not be resident from the start: they wait for the square, never
  // the square for them, and catch up within a panel's time.  What must fit one workgroup per CU for that build: square + full rows.
  // LMM_REGION_OCC=1 / 2 forces a build; default: one workgroup per CU (414 registers per lane, no spills in the walker)
  static int occ_env = -1;
  if (occ_env < 0) { const char* e = getenv("LMM_REGION_OCC"); occ_env = e ? atoi(e) : 0; }
  // (Until late in round 3 the default went to the two-per-CU build whenever the launch did not fit one workgroup per CU.  Measured
  // again with the thin row stream in place, the one-per-CU build wins at EVERY size: 32 x 1024: 0.82 against 1.05 ms, 8 x 3072: 2.98
  // against 3.42, 8 x 4096: 5.23 against 5.65, 8 x 8192: 26.5 against 28.6 -- the walker's 432 spilled registers sit on the chain every
  // other task waits for, while tasks that are dispatched late (helpers of high rows, row streams) are also needed late.)
  int occ = 1;
  if (occ_env) occ = occ_env;
  if (occ != 1) a.na = 0;
  // Row tiles: the first n128 are 128 rows high, the rest 64 (region_plan: a list-scheduling estimate of the launch for a few splits;
  // LMM_REGION_TH=128 / 64 forces all-128 / all-64).  Assistants only while everything is resident from the start.
  static int th_env = -1;
  if (th_env < 0) { const char* e = getenv("LMM_REGION_TH"); th_env = e ? atoi(e) : 0; }
  const RegionPlan plan = region_plan(P, nb, M - 128 * P, a.M_real - 128 * P, cus, a.na, asst_env == 2, (M % 64) != 0 || occ != 1 ? 128 : th_env);
  a.n128 = plan.n128; a.na = plan.na;
  if (a.na > 0) a.S = *S;
  const int row_tasks = plan.row_tasks;
  const long long tasks = 2LL * P + a.na + row_tasks;     // the square's 64-row blocks + assistants + one task per row tile below it
  // LMM_REGION_TRACE=1: per-workgroup start / end ticks (100 MHz) of every region launch, printed to stderr (a debugging aid: it
  // synchronises the stream after each launch)
  static int trace_env = -1;
  if (trace_env < 0) { const char* e = getenv("LMM_REGION_TRACE"); trace_env = e ? atoi(e) : 0; }
  long long* tr = nullptr;
  if (trace_env) { if (hipMalloc((void**)&tr, ((size_t)tasks * nb * 2 + 64 * (size_t)nb) * sizeof(long long)) != hipSuccess) tr = nullptr; }
  a.trace = tr; a.ntasks = (int)tasks;
  a.claim_scramble = g_claim_scramble;
  if (g_strict_progress && strict_claim_sets(st, &a.claim, &a.claim_next)) {
    if (occ == 1) hipLaunchKernelGGL((potrf_region_kernel<1, true>), dim3((unsigned)(tasks * nb)), dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
    else hipLaunchKernelGGL((potrf_region_kernel<2, true>), dim3((unsigned)(tasks * nb)), dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
  } else if (occ == 1) hipLaunchKernelGGL(potrf_region_kernel<1>, dim3((unsigned)(tasks * nb)), dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
  else hipLaunchKernelGGL(potrf_region_kernel<2>, dim3((unsigned)(tasks * nb)), dim3(256), LEAF_LDS_DOUBLES * 8, st, a);
  if (tr) {
    (void)hipStreamSynchronize(st);
    std::vector<long long> h((size_t)tasks * nb * 2 + 64 * (size_t)nb);
    (void)hipMemcpy(h.data(), tr, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    long long t0 = h[0];
    for (size_t i = 0; i < (size_t)tasks * nb * 2; i += 2) if (h[i] < t0) t0 = h[i];
    fprintf(stderr, "[region-trace] c0=%d P=%d R=%d nb=%d occ=%d square=%d n128=%d\n", c0, P, R, nb, occ, 2 * P + a.na, a.n128);
    for (long long i = 0; i < tasks * nb; ++i)
      fprintf(stderr, "[region-trace] wg=%lld b=%lld idx=%lld start_us=%.2f end_us=%.2f\n", i, i % nb, i / nb, (h[2 * i] - t0) / 100.0, (h[2 * i + 1] - t0) / 100.0);
    {                                                             // matrix 0's walker: per block [arrive, flag seen, diag start, diag end]
      const long long* w0 = h.data() + (size_t)tasks * nb * 2;
      for (int r = first_done ? 2 : 0; r < 2 * P; ++r)
        fprintf(stderr, "[region-walker] r=%d arrive=%.2f flag=%.2f diag0=%.2f diag1=%.2f\n", r, (w0[r] - t0) / 100.0, r ? (w0[16 + r] - t0) / 100.0 : 0.0,
                (w0[32 + r] - t0) / 100.0, (w0[48 + r] - t0) / 100.0);
    }
    (void)hipFree(tr);
  }
}

void launch_diag64(const BatchPtr& A, size_t offA, int ld, const BatchPtr& W, size_t offW, int gcol0, int n_real,
                   const BatchInfo& info, int nb, hipStream_t st) {
  // diag64m_kernel: one wave per block, rank-4 steps on the matrix pipe (the 256-thread forms of rounds 1-2: tools/retired_kernels.hip)
  LMM_TS_LAUNCH((diag64m_kernel<TS>), dim3(nb), dim3(64), 0, st, A, offA, ld, W, offW, gcol0, n_real, info);
}

int g_f32_sched = 0;      // tools/gemm32_ab: 1 = sched_group_barrier-pinned interleave in the fp32 update kernel
void launch_gemm_nt(const BatchPtr& C, size_t offC, int ldc, const BatchPtr& A, size_t offA, int lda, const BatchPtr& B,
                    size_t offB, int ldb, int M, int N, int K, int lower, bool set, int nb, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || nb <= 0) return;
  const bool narrow = (N <= 64);
  const int MT = (M + 127) / 128;
  if (set) {   // in-place TRSM by inverse: one block column, no K split
    if (g_f32) hipLaunchKernelGGL((gemm32_kernel<64, true>), dim3(MT, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB, ldb, M, N,
                                  K, 0, MT, MT, 1, 0);
    else hipLaunchKernelGGL((gemm44_kernel<64, true>), dim3(MT, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB, ldb, M, N,
                            K, 0, MT, MT, 1, 0);
    return;
  }
  // f64 wide update with a ragged last row tile (M an odd multiple of 64, the last 64 rows below every column): the main grid takes
  // the full 128-row tiles, gemm16h_kernel the last 64 rows
  if (!g_f32 && !narrow && (M % 128) == 64 && M - 64 >= N && M > 64 && (N % 128) == 0 && K >= 1024) {   // below K ~ 1000 the extra launch costs more than the idle waves
    launch_gemm_nt(C, offC, ldc, A, offA, lda, B, offB, ldb, M - 64, N, K, lower, false, nb, st);
    hipLaunchKernelGGL((gemm16h_kernel<true>), dim3(N / 128, nb), dim3(256), 0, st, C, offC + (size_t)(M - 64), ldc, A, offA + (size_t)(M - 64), lda,
                       B, offB, ldb, N, K, 0);
    return;
  }
  // underfilled wide update (the 128 x 128 tiles of all nb matrices together do not fill the CUs once): 64 x 128 tiles instead
  if (!g_f32 && !narrow && lower && M >= N && (N % 128) == 0 && (M % 64) == 0 && K >= 64 && K < 1024) {
    static int cus_h = 0;
    if (cus_h == 0) { int dev = 0; cus_h = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus_h, hipDeviceAttributeMultiprocessorCount, dev); }
    long long T128 = 0;
    for (int tj = 0; tj < N / 128; ++tj) T128 += MT - tj;
    const int MT64 = M / 64;
    long long T64 = 0;
    for (int tj = 0; tj < N / 128; ++tj) T64 += MT64 - 2 * tj;
    if (T128 * nb <= cus_h && T64 > 0) {
      hipLaunchKernelGGL((gemm16h_kernel<false>), dim3((unsigned)T64, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB, ldb, N, K, MT64);
      return;
    }
  }
  const int BNsel = narrow ? 64 : 128;
  const int NT = narrow ? 1 : (N + 127) / 128;
  long long T = 0;
  for (int tj = 0; tj < NT; ++tj) T += lower ? (MT - (tj * BNsel) / 128) : MT;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0; cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  }
  // Scheduling round = one tile per CU: the MFMA pipe, not residency, is the resource (one workgroup per CU already runs
  // at ~80 % of the CU's f64 rate, tools/stream_overlap).  With nb matrices in the batch a round holds cus / nb tiles of
  // each.  Tiles of the last, partial round are split along K (f64 atomics) so the launch ends without a long tail.
  const int slots = (cus / nb) > 0 ? (cus / nb) : 1;
  const int nk = K / 16;
  int full_items = (int)(T / slots) * slots, splitk = 1;
  const int R = (int)(T - full_items);
  static int deterministic = -1;          // LMM_DETERMINISTIC=1: no split-K atomics (bitwise reproducible, slower tail)
  if (deterministic < 0) { const char* e = getenv("LMM_DETERMINISTIC"); deterministic = (e && atoi(e) != 0) ? 1 : 0; }
  if (!deterministic && R > 0 && R <= slots / 2 && nk >= 8) {
    splitk = slots / R; if (splitk > nk / 4) splitk = nk / 4; if (splitk < 1) splitk = 1;
  }
  if (splitk == 1) full_items = (int)T;
  const int items = full_items + (int)(T - full_items) * splitk;
  if (g_f32) {
    // 256 x 256 tiles, one workgroup per CU (gemm32w_kernel, lmm_kernels_f32w.hip), where there are enough of them to fill the device
    if (!narrow && launch_gemm32w(C, offC, ldc, A, offA, lda, B, offB, ldb, M, N, K, lower, nb, cus, deterministic != 0, st)) return;
    if (narrow) hipLaunchKernelGGL((gemm32_kernel<64, false>), dim3(items, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB,
                                   ldb, M, N, K, lower, MT, full_items, splitk, 0);
    else if (g_f32_sched) hipLaunchKernelGGL((gemm32_kernel<128, false, 1>), dim3(items, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB,
                                             ldb, M, N, K, lower, MT, full_items, splitk, 0);
    else hipLaunchKernelGGL((gemm32_kernel<128, false>), dim3(items, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB,
                            ldb, M, N, K, lower, MT, full_items, splitk, 0);
    return;
  }
  if (narrow) hipLaunchKernelGGL((gemm44_kernel<64, false>), dim3(items, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB,
                                 ldb, M, N, K, lower, MT, full_items, splitk, 0);
  // staging loads two tiles ahead on the long products (+0.4 %), one below K = 1024 (less prologue)
  else if (K >= 1024) hipLaunchKernelGGL((gemm16p_kernel<2>), dim3(items, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB, ldb, M, N, K,
                                         lower, MT, full_items, splitk, 0);
  else hipLaunchKernelGGL((gemm16p_kernel<1>), dim3(items, nb), dim3(256), 0, st, C, offC, ldc, A, offA, lda, B, offB, ldb, M, N, K, lower, MT,
                          full_items, splitk, 0);
}

// C (lower triangle, N x N) = X X' for an upper-triangular X (N x N): the inverse K^-1 = L^-T L^-1 from X = L^-T.
void launch_syrk_upper_set(const BatchPtr& C, int ldc, const BatchPtr& X, int ldx, int N, int nb, hipStream_t st) {
  if (N <= 0 || nb <= 0) return;
  const int MT = (N + 127) / 128, NT = (N + 127) / 128;
  long long T = 0;
  for (int tj = 0; tj < NT; ++tj) T += MT - tj;
  if (g_f32) hipLaunchKernelGGL((gemm32_kernel<128, true>), dim3((int)T, nb), dim3(256), 0, st, C, (size_t)0, ldc, X, (size_t)0, ldx, X, (size_t)0, ldx,
                                N, N, N, 1, MT, (int)T, 1, 1);
  else hipLaunchKernelGGL((gemm44_kernel<128, true>), dim3((int)T, nb), dim3(256), 0, st, C, (size_t)0, ldc, X, (size_t)0, ldx, X, (size_t)0, ldx,
                          N, N, N, 1, MT, (int)T, 1, 1);
}
void launch_syrk_upper_set(double* C, int ldc, const double* X, int ldx, int N, hipStream_t st) {
  BatchPtr c{}, a{};
  c.p[0] = C; a.p[0] = const_cast<double*>(X);
  launch_syrk_upper_set(c, ldc, a, ldx, N, 1, st);
}

void launch_gemm_nt(double* C, int ldc, const double* A, int lda, const double* B, int ldb, int M, int N, int K,
                    int lower, bool set, hipStream_t st) {
  BatchPtr c{}, a{}, b{};
  c.p[0] = C; a.p[0] = const_cast<double*>(A); b.p[0] = const_cast<double*>(B);
  launch_gemm_nt(c, 0, ldc, a, 0, lda, b, 0, ldb, M, N, K, lower, set, 1, st);
}

void launch_lml_reduce(const BatchPtr& A, int nb, int ld, int n, int rider_row0, int nrhs, double* out, hipStream_t st,
                       const BatchInfo* info, int* info_out) {
  BatchInfo ib{};
  if (info && info_out) ib = *info; else info_out = nullptr;
  LMM_TS_LAUNCH((lml_reduce_kernel<TS>), dim3(nb), dim3(256), 0, st, A, ld, n, rider_row0, nrhs, out, ib, info_out);
}
void launch_lml_reduce(const double* A, int ld, int n, int rider_row0, int nrhs, double* out, hipStream_t st) {
  BatchPtr b{};
  b.p[0] = const_cast<double*>(A);
  launch_lml_reduce(b, 1, ld, n, rider_row0, nrhs, out, st);
}

void launch_extract_row(const double* A, int ld, int row, int n, double* out, hipStream_t st) {
  LMM_TS_LAUNCH((extract_row_kernel<TS>), dim3((n + 255) / 256), dim3(256), 0, st, (const void*)A, ld, row, n, out);
}

void launch_extract_rows(const BatchPtr& A, int nb, int ld, int row, int n, int nfill, const BatchPtr& o1, const BatchPtr& o2,
                         hipStream_t st) {
  LMM_TS_LAUNCH((extract_rows_kernel<TS>), dim3((nfill + 255) / 256, nb), dim3(256), 0, st, A, ld, row, n, nfill, o1, o2);
}

// k-chunk width of the strip reductions: at most 64 chunks, a multiple of 256 columns
int strip_kc(int nk) { int kc = ((nk + 63) / 64 + 255) / 256 * 256; return kc < 256 ? 256 : kc; }
size_t strip_partial_elems(int nr, int nk, int nv) {
  const int kc = strip_kc(nk);
  return (size_t)((nk + kc - 1) / kc) * nv * ((nr + 63) / 64 * 64);
}

// mean_out[r] = mu + sum_k R[r,k] z[k],  var_out[r] = base - sum_k R[r,k]^2  for the nr riders of R (ld, nk columns).
// R == nullptr (prior): mean = mu, var = base.
void launch_rider_stats(const double* R, int ld, int nr, int nk, const double* z, double mu, double base, double* partial,
                        double* mean_out, double* var_out, hipStream_t st) {
  if (R == nullptr || nk == 0) {
    if (mean_out) hipLaunchKernelGGL(fill_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, mean_out, nr, mu);
    if (var_out) hipLaunchKernelGGL(fill_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, var_out, nr, base);
    return;
  }
  const int kc = strip_kc(nk), nch = (nk + kc - 1) / kc, nrp = (nr + 63) / 64 * 64;
  dim3 grid(nrp / 64, nch);
  if (var_out) LMM_TS_LAUNCH((strip_reduce_kernel<true, false, TS>), grid, dim3(256), 0, st, (const void*)R, ld, nk, kc, z, partial);
  else LMM_TS_LAUNCH((strip_reduce_kernel<false, false, TS>), grid, dim3(256), 0, st, (const void*)R, ld, nk, kc, z, partial);
  hipLaunchKernelGGL(strip_finish_kernel, dim3((nr + 255) / 256), dim3(256), 0, st, partial, nrp, nch, var_out ? 2 : 1, nr, 0,
                     mu, base, mean_out, var_out);
}

void launch_backsolve(const BatchPtr& L, int ld, const BatchPtr& W, int nblk, const BatchPtr& z, int nb, hipStream_t st) {
  for (int b = nblk - 1; b >= 0; --b) {
    int grid = (b * 64 + 255) / 256; if (grid < 1) grid = 1;
    LMM_TS_LAUNCH((backsolve_step_kernel<TS>), dim3(grid, nb), dim3(256), 0, st, L, ld, W, b, z);
  }
}

void launch_tall_skinny(const double* In, int ldi, int n, int K, const double* Mx, int ldm, int C, double* Out, int ldo,
                        const double* sub, const double* Ref, int ldr, double* partial, int mode, hipStream_t st) {
  dim3 grid((n + 63) / 64, (C + 7) / 8);
  if (mode == 0 && K >= 128 && (size_t)grid.x * grid.y < 256)      // few workgroups, long K loop: 16 waves split it
    hipLaunchKernelGGL((tall_skinny_kernel<8, 16>), grid, dim3(1024), 0, st, In, ldi, n, K, Mx, ldm, C, Out, ldo, sub, Ref, ldr,
                       partial, mode);
  else
    hipLaunchKernelGGL((tall_skinny_kernel<8, 4>), grid, dim3(256), 0, st, In, ldi, n, K, Mx, ldm, C, Out, ldo, sub, Ref, ldr,
                       partial, mode);
}

int tall_skinny_partials(int n, int C) { return ((n + 63) / 64) * ((C + 7) / 8); }

void launch_sum_partials(const double* partial, int count, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, count, out);
}

int post_mean_ichunk(int n) { return n > 512 ? 512 : (n < 1 ? 1 : n); }
size_t post_mean_partial_elems(int ns, int n) {
  const int ic = post_mean_ichunk(n);
  return (size_t)((n + ic - 1) / ic) * ((ns + 255) / 256 * 256);
}

// out[s] = g.mean + sum_i kappa(xs_s, x_i) alpha_i;  alpha == nullptr: the prior mean.
void launch_post_mean(const double* xs, int ns, const double* x, int n, int d, const double* alpha, LatentDev g,
                      double* partial, double* out, hipStream_t st) {
  if (alpha == nullptr || n == 0) {
    hipLaunchKernelGGL(fill_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, out, ns, g.mean);
    return;
  }
  const int ic = post_mean_ichunk(n), nch = (n + ic - 1) / ic, nsp = (ns + 255) / 256 * 256;
  hipLaunchKernelGGL(post_mean_kernel, dim3(nsp / 256, nch), dim3(256), 0, st, xs, ns, x, n, d, ic, alpha, g, partial);
  hipLaunchKernelGGL(strip_finish_kernel, dim3((ns + 255) / 256), dim3(256), 0, st, partial, nsp, nch, 1, ns, 0, g.mean, 0.0,
                     out, (double*)nullptr);
}

void launch_mix(const double* lat, int ns, int ml, const double* Hm, int p, int pw, double lat_add, double out_add,
                const double* eps, double eps_scale, double* out, hipStream_t st) {
  dim3 grid((ns + 255) / 256, p);
  hipLaunchKernelGGL(mix_kernel, grid, dim3(256), 0, st, lat, ns, ml, Hm, p, pw, lat_add, out_add, eps, eps_scale, out);
}

// terms: 1 = plain bf16 operands, 2 = hi + lo split (three products)
void launch_mix_bf16(const double* lat, int ns, int ml, const double* Hm, int p, int pw, double lat_add, double out_add,
                     int terms, double* out, hipStream_t st) {
  dim3 grid((ns + 63) / 64);
  if (terms == 2) hipLaunchKernelGGL((mix_bf16_kernel<2>), grid, dim3(256), 0, st, lat, ns, ml, Hm, p, pw, lat_add, out_add, out);
  else hipLaunchKernelGGL((mix_bf16_kernel<1>), grid, dim3(256), 0, st, lat, ns, ml, Hm, p, pw, lat_add, out_add, out);
}

void launch_cov_mix(const BatchPtr& Cl, int ldcl, int nl, const double* Hs, int p, int ns, double jitter, double sigma2,
                    int init, double* out, hipStream_t st) {
  dim3 grid((ns + 15) / 16, (ns + 15) / 16, p * p);
  LMM_TS_LAUNCH((cov_mix_kernel<TS>), grid, dim3(256), 0, st, Cl, ldcl, nl, Hs, p, ns, jitter, sigma2, init, out);
}

// out = mu + L z for the leading n x n lower triangle of L (ld); partial: strip_partial_elems(n, n, 1) doubles.
void launch_trmv_lower(const double* L, int ld, int n, const double* z, double mu, double* partial, double* out,
                       hipStream_t st) {
  const int kc = strip_kc(n), nch = (n + kc - 1) / kc, nrp = (n + 63) / 64 * 64;
  LMM_TS_LAUNCH((strip_reduce_kernel<false, true, TS>), dim3(nrp / 64, nch), dim3(256), 0, st, (const void*)L, ld, n, kc, z, partial);
  hipLaunchKernelGGL(strip_finish_kernel, dim3((n + 255) / 256), dim3(256), 0, st, partial, nrp, nch, 1, n, kc, mu, 0.0, out,
                     (double*)nullptr);
}

void launch_set_identity(double* R, int ld, int nc, hipStream_t st) {
  LMM_TS_LAUNCH((set_identity_kernel<TS>), dim3((nc + 255) / 256, nc), dim3(256), 0, st, (void*)R, ld, nc);
}

int grad_partials(int n) { const int nt = (n + 63) / 64; return LMM_NG * nt * nt; }

// out8: [dl/d ell, tr Kinv (rows < nsplit), a.a (rows < nsplit), a.delta, sum a, tr Kinv (rows >= nsplit), a.a (rows >= nsplit),
//        sum_{i>j} (a_i a_j - Kinv_ij) K_ij]
void launch_grad_reduce(const double* Kinv, int ld, int n, int nsplit, const double* alpha, const double* delta, const double* x, int d,
                        LatentDev g, double* partial, double* out7, hipStream_t st) {
  const int nt = (n + 63) / 64;
  LMM_TS_LAUNCH((grad_reduce_kernel<TS>), dim3(nt, nt), dim3(256), 0, st, (const void*)Kinv, ld, n, nsplit, alpha, delta, x, d, g, nt, partial);
  hipLaunchKernelGGL(grad_finish_kernel, dim3(1), dim3(256), 0, st, partial, nt, out7);
}

void launch_vec_axpby(const double* a, double sa, const double* b, double sb, size_t n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(vec_axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, sa, b, sb, n, out);
}

void launch_block_trace(const double* Minv, int ld, int n, int m, int i0, int i1, double* out, hipStream_t st) {
  LMM_TS_LAUNCH((block_trace_kernel<TS>), dim3(m, m), dim3(256), 0, st, (const void*)Minv, ld, n, m, i0, i1, out);
}

void launch_vec_lin_blocks(const double* a, const double* b, const NoiseBlocks& nb, double num, int N, size_t count, double* out, hipStream_t st) {
  hipLaunchKernelGGL(vec_lin_blocks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, a, b, nb, num, N, count, out);
}

void launch_atb(const double* X, int ldx, const double* Z, int ldz, int n, int na, int nb, double* out, hipStream_t st) {
  hipLaunchKernelGGL(atb_kernel, dim3(na, nb), dim3(256), 0, st, X, ldx, Z, ldz, n, na, out);
}

void launch_fill(double* p, int n, double v, hipStream_t st) {
  hipLaunchKernelGGL(fill_kernel, dim3((n + 255) / 256), dim3(256), 0, st, p, n, v);
}

void launch_reorder(const double* in, int n, int p, int to_outputs, double* out, hipStream_t st) {
  hipLaunchKernelGGL(reorder_kernel, dim3((n * p + 255) / 256), dim3(256), 0, st, in, n, p, to_outputs, out);
}

void launch_block_scatter(const double* src, int lds, int nr, int nc, double* out, size_t ldo, size_t row0, int rs, size_t col0, int cs,
                          hipStream_t st) {
  if (nr <= 0 || nc <= 0) return;
  LMM_TS_LAUNCH((block_scatter_kernel<TS>), dim3((nr + 255) / 256, nc), dim3(256), 0, st, (const void*)src, lds, nr, nc, out, ldo, row0, rs, col0, cs);
}

void launch_vec_lin(const double* a, const double* b, double sb, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(vec_lin_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, b, sb, n, out);
}

void launch_normals(unsigned long long seed, unsigned long long stream, size_t count, double* out, hipStream_t st) {
  if (count == 0) return;
  const size_t pairs = (count + 1) / 2;
  hipLaunchKernelGGL(normals_kernel, dim3((unsigned int)((pairs + 255) / 256)), dim3(256), 0, st, seed, stream, count, out);
}

void launch_mfma_peak(double* out, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, st, out, iters);
}
