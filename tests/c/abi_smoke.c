/* Plain-C consumer of include/lmm_hip.h (what a Julia `ccall`, or any FFI, sees): no Python, no torch.
 * Builds with gcc against liblmm_hip.so; run on a GPU box it evaluates a tiny OILMM logpdf + gradient + posterior marginals
 * and checks them against closed forms (m = p = 1, U = 1, S = 1: a single GP).  tests/test_abi_c.py drives it. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "lmm_hip.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != LMM_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, lmm_last_error_string()); return 2; } } while (0)

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'l') {   /* "link": only prove that every symbol resolves; no GPU touched */
    void* syms[] = {(void*)lmm_init, (void*)lmm_oilmm_logpdf, (void*)lmm_ilmm_logpdf, (void*)lmm_mogp_logpdf, (void*)lmm_mogp_logpdf_diag,
                    (void*)lmm_oilmm_posterior_create, (void*)lmm_post_destroy, (void*)lmm_oilmm_mean_and_var,
                    (void*)lmm_lmm_rand, (void*)lmm_oilmm_logpdf_grad, (void*)lmm_lmm_mean_and_cov,
                    (void*)lmm_ilmm_post_mean_and_cov, (void*)lmm_ilmm_post_condition, (void*)lmm_normals};
    printf("linked %zu symbols\n", sizeof syms / sizeof *syms);
    return 0;
  }
  CHECK(lmm_init(0));
  /* n = 2 points at distance 1, SE kernel, noise 0.5:  K + s I = [[1.5, e^-1/2], [e^-1/2, 1.5]] */
  const double x[2] = {0.0, 1.0}, y[2] = {0.3, -0.7}, U[1] = {1.0}, S[1] = {1.0};
  lmm_gp_t gp = {LMM_KERNEL_SE, 1.0, 1.0, 0.0};
  double lp = 0.0;
  CHECK(lmm_oilmm_logpdf(x, 1, 2, y, 1, U, S, 1, 0.5, &gp, 0, 1, 1, &lp));
  const double a = 1.5, b = exp(-0.5), det = a * a - b * b;
  const double quad = (a * y[0] * y[0] - 2 * b * y[0] * y[1] + a * y[1] * y[1]) / det;
  const double ref = -0.5 * (2 * log(2 * M_PI) + log(det) + quad);      /* regulariser is 0: log S = 0, p = m, U U' = I */
  printf("logpdf %.15g ref %.15g\n", lp, ref);
  if (fabs(lp - ref) > 1e-12 * fabs(ref)) return 3;
  double gy[2], gs2, gS[1], gU[1];
  lmm_gp_grad_t gg[1];
  CHECK(lmm_oilmm_logpdf_grad(x, 1, 2, y, 1, U, S, 1, 0.5, &gp, 0, 1, 1, &lp, gy, &gs2, gS, gU, gg));
  /* d/dy = -K~^-1 y */
  const double al0 = (a * y[0] - b * y[1]) / det, al1 = (a * y[1] - b * y[0]) / det;
  printf("grad_y %.15g %.15g ref %.15g %.15g\n", gy[0], gy[1], -al0, -al1);
  if (fabs(gy[0] + al0) > 1e-12 || fabs(gy[1] + al1) > 1e-12) return 4;
  if (fabs(gg[0].mean - (al0 + al1)) > 1e-12) return 5;
  lmm_post_t* post = NULL;
  CHECK(lmm_oilmm_posterior_create(x, 1, 2, y, 1, U, S, 1, 0.5, &gp, 0, 1, &post));
  double mu[2], var[2];
  CHECK(lmm_oilmm_mean_and_var(post, NULL, U, S, 1, 1, 0, 1, 0.5, 1, x, 1, 2, NULL, mu, var));
  /* posterior mean at the training points: K alpha */
  const double m0 = 1.0 * al0 + b * al1;
  printf("post mean %.15g ref %.15g var %.15g\n", mu[0], m0, var[0]);
  if (fabs(mu[0] - m0) > 1e-12) return 6;
  CHECK(lmm_post_destroy(post));
  /* error path: out-dim mismatch is LMM_ERR_DIM with the reference's message */
  const double U2[2] = {1.0, 0.0};
  int rc = lmm_oilmm_logpdf(x, 1, 2, y, 1, U2, S, 2, 0.5, &gp, 0, 1, 1, &lp);
  if (rc != LMM_ERR_DIM && rc != LMM_ERR_ARG) { fprintf(stderr, "expected an error, got %d\n", rc); return 7; }
  CHECK(lmm_shutdown());
  printf("C ABI smoke OK\n");
  return 0;
}
