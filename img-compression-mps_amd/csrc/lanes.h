// Cross-lane sums of doubles on the VALU (gfx950): DPP moves inside a row of 16 lanes, v_permlane16_swap /
// v_permlane32_swap between rows.  __shfl_xor compiles to ds_bpermute_b32 -- one round trip through the LDS crossbar per
// dword and step, twelve dependent round trips for a wave-wide sum of doubles -- and these sums sit on the critical
// path of every column of the tridiagonalisation kernels (eig_tridiag.hip, eig_sym.inc).
//
// Every function returns the sum in ALL lanes of the group it sums over, with a fixed association (deterministic).
#pragma once
#include <hip/hip_runtime.h>

namespace ndmps_lanes {

constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E;            // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int kDppHalfMirror = 0x141, kDppMirror = 0x140;  // lane i <-> 7 - i of its 8, i <-> 15 - i of its row
constexpr int kDppRor4 = 0x124, kDppRor8 = 0x128;          // rotation inside the row of 16

template <int CTRL>
__device__ __forceinline__ double dpp_get(double v) {
  const long long bits = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(bits >> 32), CTRL, 0xF, 0xF, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

typedef unsigned int lanes_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double make_f64(unsigned lo, unsigned hi) {
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// v[l] + v[l ^ 16]: the swap leaves rows (0, 0, 2, 2) of v in one register and rows (1, 1, 3, 3) in the other
__device__ __forceinline__ double xor16_sum(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const lanes_u32x2 lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
  const lanes_u32x2 hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  return make_f64(lo[0], hi[0]) + make_f64(lo[1], hi[1]);
}
// v[l] + v[l ^ 32]
__device__ __forceinline__ double xor32_sum(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const lanes_u32x2 lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
  const lanes_u32x2 hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
  return make_f64(lo[0], hi[0]) + make_f64(lo[1], hi[1]);
}

// sum over N ADJACENT lanes (N = 2, 4, ..., 64; groups aligned to N)
template <int N>
__device__ __forceinline__ double sum_adjacent(double v) {
  static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16 || N == 32 || N == 64, "power of two up to the wave");
  if (N >= 2) v += dpp_get<kDppXor1>(v);
  if (N >= 4) v += dpp_get<kDppXor2>(v);
  if (N >= 8) v += dpp_get<kDppHalfMirror>(v);  // the quads are uniform by now: any lane of the other quad will do
  if (N >= 16) v += dpp_get<kDppMirror>(v);
  if (N >= 32) v = xor16_sum(v);
  if (N >= 64) v = xor32_sum(v);
  return v;
}

// sum over the lanes with the same (lane % STRIDE) (STRIDE = 1, 2, ..., 32)
template <int STRIDE>
__device__ __forceinline__ double sum_strided(double v) {
  static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8 || STRIDE == 16 || STRIDE == 32, "power of two");
  if (STRIDE <= 1) v += dpp_get<kDppXor1>(v);
  if (STRIDE <= 2) v += dpp_get<kDppXor2>(v);
  if (STRIDE <= 4) v += dpp_get<kDppRor4>(v);  // with the next step: the orbit i, i + 4, i + 8, i + 12 of the row
  if (STRIDE <= 8) v += dpp_get<kDppRor8>(v);
  if (STRIDE <= 16) v = xor16_sum(v);
  if (STRIDE <= 32) v = xor32_sum(v);
  return v;
}

}  // namespace ndmps_lanes
