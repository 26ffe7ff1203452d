// Scratch: can an fp64-MFMA kernel of <= 64 registers per lane run BESIDE the resident tridiagonalisation (448 of the 512
// registers of every SIMD lane, two workgroups per CU)?  A burner with the instruction mix of a Gram tile loop at a 32 x 32
// wave tile: per k-step of four rows two + two operand reads from LDS (fp32, converted) and four independent
// v_mfma_f64_16x16x4_f64.  One workgroup of four waves per CU (its dynamic LDS keeps a second one out).
// hipcc -O3 --offload-arch=gfx950 -shared -fPIC corun_probe.hip -o _bin/libcorun.so
#include <hip/hip_runtime.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

extern "C" __global__ void __launch_bounds__(256, 8) burn_kernel(const float* __restrict__ in, double* __restrict__ out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 32 * 128; i += 256) lds[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane >> 4, lc = lane & 15;
  f64x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  const int a_col = (wave >> 1) * 32, b_col = 64 + (wave & 1) * 32;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float* p = lds + (4 * ks + lr) * 128 + lc;
      const double a0 = p[a_col], a1 = p[a_col + 16], b0 = p[b_col], b1 = p[b_col + 16];
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[3], 0, 0, 0);
    }
    asm volatile("" ::: "memory");  // the operands are read again every trip
  }
  double s = 0.0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// the same with fp32 MFMA (v_mfma_f32_32x32x2_f32): the instruction mix of a projection's tile loop at a 32 x 64 wave tile
typedef float f32x16 __attribute__((ext_vector_type(16)));
extern "C" __global__ void __launch_bounds__(256, 8) burn32_kernel(const float* __restrict__ in, double* __restrict__ out, int iters) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 32 * 128; i += 256) lds[i] = in[i & 4095];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const int a_col = (wave >> 1) * 32, b_col = 64 + (wave & 1) * 32;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {  // 32 rows in k-steps of 2
      const float* p = lds + (2 * ks + lk) * 128 + li;
      const float a0 = p[a_col], b0 = p[b_col], b1 = p[(b_col + 32) & 127];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
    }
    asm volatile("" ::: "memory");
  }
  double s = 0.0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

extern "C" int corun_burn32(int wgs, int iters, int lds_bytes, const float* in, double* out, void* stream) {
  static int set = 0;
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&burn32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess)
      return 1;
    set = 1;
  }
  hipLaunchKernelGGL(burn32_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, in, out, iters);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int corun_burn(int wgs, int iters, int lds_bytes, const float* in, double* out, void* stream) {
  static int set = 0;
  if (!set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&burn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024) != hipSuccess)
      return 1;
    set = 1;
  }
  hipLaunchKernelGGL(burn_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, in, out, iters);
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
