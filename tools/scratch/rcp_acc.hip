// one-off: accuracy of v_rcp_f64 / v_rsq_f64 estimates on gfx950 and after 1 / 2 Newton steps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = r0 * fma(-v, r0, 2.0);
  double r2 = r1 * fma(-v, r1, 2.0);
  double q0 = __builtin_amdgcn_rsq(v);
  double q1 = q0 * fma(-0.5 * v * q0, q0, 1.5);
  double q2 = q1 * fma(-0.5 * v * q1, q1, 1.5);
  out[6 * i + 0] = r0; out[6 * i + 1] = r1; out[6 * i + 2] = r2;
  out[6 * i + 3] = q0; out[6 * i + 4] = q1; out[6 * i + 5] = q2;
}
int main() {
  const int n = 1 << 20;
  double* hx = (double*)malloc(n * 8); double* ho = (double*)malloc(n * 48);
  srand(1);
  for (int i = 0; i < n; ++i) hx[i] = 0.25 + 2.5 * (rand() / (double)RAND_MAX);
  double *dx, *dout; hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 48);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(ho, dout, n * 48, hipMemcpyDeviceToHost);
  double e[6] = {0};
  for (int i = 0; i < n; ++i) {
    long double rr = 1.0L / hx[i], qq = 1.0L / sqrtl((long double)hx[i]);
    for (int j = 0; j < 3; ++j) { double d = fabs((double)((ho[6*i+j] - rr) / rr)); if (d > e[j]) e[j] = d; }
    for (int j = 3; j < 6; ++j) { double d = fabs((double)((ho[6*i+j] - qq) / qq)); if (d > e[j]) e[j] = d; }
  }
  printf("rcp: est %.3e  1 newton %.3e  2 newton %.3e\n", e[0], e[1], e[2]);
  printf("rsq: est %.3e  1 newton %.3e  2 newton %.3e\n", e[3], e[4], e[5]);
  return 0;
}
