// Accuracy of v_rsq_f64 and of one / two Newton steps from it (the 4 x 4 pivot factorisation of the diagonal-block kernel):
// max relative error over random inputs against a long-double reference.   hipcc --offload-arch=gfx950 -O2 tools/rsq_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i], y0 = __builtin_amdgcn_rsq(v), h = 0.5 * v;
  double y1 = y0 * __builtin_fma(-h * y0, y0, 1.5);
  double y2 = y1 * __builtin_fma(-h * y1, y1, 1.5);
  // one step in the residual form: e = 1 - v y0^2 (one fma after y0^2), y = y0 + y0 * e / 2
  double e = __builtin_fma(-v * y0, y0, 1.0);
  double y1r = __builtin_fma(y0 * 0.5, e, y0);
  o[4 * i] = y0; o[4 * i + 1] = y1; o[4 * i + 2] = y2; o[4 * i + 3] = y1r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(4 * n);
  std::mt19937_64 g(1); std::uniform_real_distribution<double> u(-12.0, 12.0);
  for (auto& v : x) v = std::exp(u(g));
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, 4 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(o.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
  double m[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    long double r = 1.0L / sqrtl((long double)x[i]);
    for (int j = 0; j < 4; ++j) { double e = (double)fabsl(((long double)o[4 * i + j] - r) / r); if (e > m[j]) m[j] = e; }
  }
  printf("max rel err: v_rsq_f64 %.3e | one Newton step %.3e | two steps %.3e | one step, residual form %.3e  (eps = 1.1e-16)\n", m[0], m[1], m[2], m[3]);
  return 0;
}
