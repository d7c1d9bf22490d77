// A/B of the 64x64 diagonal-block kernels in ONE process: form 1 = diag64_kernel (round 1), 2 = diag64v2_kernel, 3 = diag64m_kernel
// (one wave, MFMA rank-4 steps).  Checks L and W = L^-1 of every form against a host long-double Cholesky, then times them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 tools/diag_ab.hip -o tools/diag_ab && tools/diag_ab [nb]
#include "../linearmixingmodels.jl_amd/csrc/lmm_kernels.hip"
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
extern int g_diag_form;
int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 16;
  const int ld = 1024 + 16, off = 128;                 // the block sits at (off, off) of a larger matrix
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  std::vector<std::vector<double>> host(nb, std::vector<double>((size_t)ld * 256, 0.0));
  std::vector<std::vector<long double>> Lref(nb, std::vector<long double>(64 * 64, 0.0L)), Wref(nb, std::vector<long double>(64 * 64, 0.0L));
  for (int b = 0; b < nb; ++b) {
    std::vector<double> G(64 * 64);
    for (auto& v : G) v = nd(rng);
    std::vector<long double> Am(64 * 64);
    for (int i = 0; i < 64; ++i)
      for (int j = 0; j <= i; ++j) {
        long double sacc = (i == j) ? 1.0L + b : 0.0L;
        for (int k = 0; k < 64; ++k) sacc += (long double)G[i * 64 + k] * G[j * 64 + k];
        Am[i * 64 + j] = (long double)(double)sacc;
        host[b][(size_t)(off + j) * ld + off + i] = (double)sacc;
        if (i != j) host[b][(size_t)(off + i) * ld + off + j] = NAN;       // upper triangle must never be read
      }
    auto& L = Lref[b]; auto& Wi = Wref[b];
    for (int j = 0; j < 64; ++j) {
      long double d = Am[j * 64 + j];
      for (int k = 0; k < j; ++k) d -= L[j * 64 + k] * L[j * 64 + k];
      L[j * 64 + j] = sqrtl(d);
      for (int i = j + 1; i < 64; ++i) {
        long double v = Am[i * 64 + j];
        for (int k = 0; k < j; ++k) v -= L[i * 64 + k] * L[j * 64 + k];
        L[i * 64 + j] = v / L[j * 64 + j];
      }
    }
    for (int j = 0; j < 64; ++j)
      for (int i = j; i < 64; ++i) {
        long double v = (i == j) ? 1.0L : 0.0L;
        for (int k = j; k < i; ++k) v -= L[i * 64 + k] * Wi[k * 64 + j];
        Wi[i * 64 + j] = v / L[i * 64 + i];
      }
  }
  BatchPtr A{}, W{}; BatchInfo info{};
  std::vector<double*> dA(nb), dW(nb); std::vector<int*> dI(nb);
  for (int b = 0; b < nb; ++b) {
    hipMalloc(&dA[b], (size_t)ld * 256 * 8); hipMalloc(&dW[b], 64 * 64 * 8); hipMalloc(&dI[b], 4);
    A.p[b] = dA[b]; W.p[b] = dW[b]; info.p[b] = dI[b];
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form : {1, 2, 3}) {
    g_diag_form = form;
    for (int b = 0; b < nb; ++b) { hipMemcpy(dA[b], host[b].data(), (size_t)ld * 256 * 8, hipMemcpyHostToDevice); hipMemset(dW[b], 0xff, 64 * 64 * 8); hipMemset(dI[b], 0, 4); }
    launch_diag64(A, (size_t)off * ld + off, ld, W, 0, off, 100000, info, nb, 0);
    hipDeviceSynchronize();
    double eL = 0, eW = 0, eU = 0; int inf = 0;
    for (int b = 0; b < nb; ++b) {
      std::vector<double> o((size_t)ld * 256), w(64 * 64); int ii;
      hipMemcpy(o.data(), dA[b], o.size() * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), dW[b], 64 * 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(&ii, dI[b], 4, hipMemcpyDeviceToHost);
      inf |= ii;
      for (int i = 0; i < 64; ++i)
        for (int j = 0; j < 64; ++j) {
          const double lv = o[(size_t)(off + j) * ld + off + i], wv = w[j * 64 + i];
          if (j <= i) {
            eL = fmax(eL, fabs(lv - (double)Lref[b][i * 64 + j]) / fabs((double)Lref[b][i * 64 + i]));
            const double ew = fabs(wv - (double)Wref[b][i * 64 + j]) * fabs((double)Lref[b][j * 64 + j]);
            if (!(ew <= eW)) eW = ew;
          } else {
            if (!std::isnan(lv)) eU = 1.0;                 // upper triangle of A must be left alone
            if (wv != 0.0) eU = fmax(eU, 2.0);             // upper triangle of W must be exactly zero
          }
        }
    }
    // non-positive pivot: info must name it
    std::vector<double> badm = host[0]; badm[(size_t)(off + 37) * ld + off + 37] = -5.0;
    hipMemcpy(dA[0], badm.data(), badm.size() * 8, hipMemcpyHostToDevice); hipMemset(dI[0], 0, 4);
    launch_diag64(A, (size_t)off * ld + off, ld, W, 0, off, 100000, info, 1, 0);
    int ib; hipMemcpy(&ib, dI[0], 4, hipMemcpyDeviceToHost);
    float best = 1e30f;
    for (int round = 0; round < 5; ++round) {
      hipEventRecord(e0);
      for (int r = 0; r < 200; ++r) launch_diag64(A, (size_t)off * ld + off, ld, W, 0, off, 100000, info, nb, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms / 200);
    }
    printf("form %d nb=%d: %.2f us per launch | max rel err L %.2e, W %.2e, upper-triangle flags %.0f, info(clean) %d, info(bad pivot at %d) %d\n",
           form, nb, best * 1e3, eL, eW, eU, inf, off + 37 + 1, ib);
    fflush(stdout);
#ifdef LMM_DIAG_TIMING
    if (form == 3) {
      long long ts[8]; hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_diag_ts), sizeof(ts));
      printf("  form 3 cycles (block 0): prologue %lld, 16 steps %lld, epilogue %lld\n", ts[0] - ts[3], ts[1] - ts[0], ts[2] - ts[1]);
    }
#endif
  }
  return 0;
}
