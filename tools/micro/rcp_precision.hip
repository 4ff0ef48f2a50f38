// micro-benchmark: accuracy of v_rcp_f64 / v_rsq_f64 hardware estimates and of 1 / 2 Newton steps
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(int n, const double* x, double* o) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double e = fma(-v, r0, 1.0); double r1 = fma(r0, e, r0);
  e = fma(-v, r1, 1.0); double r2 = fma(r1, e, r1);
  double s0 = __builtin_amdgcn_rsq(v);
  double f = fma(-v * s0, 0.5 * s0, 0.5); double s1 = fma(s0, f, s0);
  f = fma(-v * s1, 0.5 * s1, 0.5); double s2 = fma(s1, f, s1);
  o[6 * i] = r0; o[6 * i + 1] = r1; o[6 * i + 2] = r2; o[6 * i + 3] = s0; o[6 * i + 4] = s1; o[6 * i + 5] = s2;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), o(6 * n);
  unsigned long long st = 88172645463325252ull;
  for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = std::ldexp(1.0 + (st >> 11) * (1.0 / 9007199254740992.0), (int)(st % 41) - 20); }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, n, dx, dout);
  hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    long double rc = 1.0L / x[i], rs = 1.0L / sqrtl((long double)x[i]);
    for (int k = 0; k < 3; ++k) { double e = fabs((double)((o[6 * i + k] - rc) / rc)); if (e > m[k]) m[k] = e; }
    for (int k = 3; k < 6; ++k) { double e = fabs((double)((o[6 * i + k] - rs) / rs)); if (e > m[k]) m[k] = e; }
  }
  printf("max rel err: rcp hw %.3e, +1NR %.3e, +2NR %.3e | rsq hw %.3e, +1NR %.3e, +2NR %.3e\n", m[0], m[1], m[2], m[3], m[4], m[5]);
  return 0;
}
