// Rate of v_mfma_f64_16x16x4_f64 on gfx950 (the MI355X microarch guide has no f64 row): back-to-back issue with
// independent accumulators in VGPRs, one to four waves per SIMD, operands constant or random (DVFS: the clock the
// chip holds depends on the toggling data).  hipcc -O3 --offload-arch=gfx950 mfma_f64_rate.hip -o _bin/mfma64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024) k(const double* __restrict__ in, double* out, int iters) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f64x4){0, 0, 0, 0};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(threadIdx.x * 8 + i) & 4095];
    b[i] = in[(threadIdx.x * 8 + 4 + i) & 4095];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double *in, *out;
  double* h = (double*)malloc(4096 * 8);
  hipMalloc(&in, 4096 * 8);
  hipMalloc(&out, 1024 * 1024 * 8);
  const int iters = 40000;
  for (int random = 0; random < 2; ++random) {
    for (int i = 0; i < 4096; ++i) h[i] = random ? (double)rand() / RAND_MAX - 0.5 : 1.0;
    hipMemcpy(in, h, 4096 * 8, hipMemcpyHostToDevice);
    for (int cfg = 0; cfg < 4; ++cfg) {
      const int blocks = cfg == 0 ? 1 : 256, threads = cfg <= 1 ? 256 : cfg == 2 ? 512 : 1024;
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      k<<<blocks, threads>>>(in, out, 100);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k<<<blocks, threads>>>(in, out, iters);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double waves_per_simd = threads / 256.0;
      const double tf = (double)blocks * (threads / 64) * iters * 8 * 2048 / (ms * 1e-3) / 1e12;
      printf("%s operands, %3d CU(s), %.0f wave(s) per SIMD: %7.3f ms, %5.1f TFLOP/s, %.1f ns per MFMA and SIMD\n",
             random ? "random  " : "constant", blocks, waves_per_simd, ms, tf, ms * 1e6 / (iters * 8.0 * waves_per_simd));
    }
  }
  return 0;
}
