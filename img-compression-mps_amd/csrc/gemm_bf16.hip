// bf16 storage path (BASELINE config 5: bf16 volumes, chi = 128, contraction-bound):
//
//   ndmps_gemm_bf16   C = A B or A B^T with bf16 operands / result and fp32 accumulation on
//                     v_mfma_f32_32x32x16_bf16 (reference: the tensordot chain of `mps ^ ...`, core/ndmps.py:140,
//                     and the U S product carried left by from_dense, core/ndmps.py:74, at the dtype the
//                     reference keeps at core/ndmps.py:56)
//   ndmps_convert_*   element-type conversions used where a bf16 tensor meets an fp32 / fp64 kernel
//
// The kernel multiplies K-contiguous operands: A (m x K, row-major) and Bt (n x K, row-major), C = A Bt^T.
// A (K x n) right operand -- a core viewed as (chi, d chi') -- is transposed first; cores are at most a few
// MB, the tensors they multiply hundreds.  128 x 128 tile per workgroup, 4 waves (2 x 2) of 64 x 64, K in
// steps of 64 through LDS (16-byte chunks XOR-swizzled by row: conflict-free ds_read_b128 fragment reads);
// the result tile goes through the same LDS as bf16 and leaves in 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "common.h"

namespace {

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int CHUNKS = BK / 8;  // 16-byte chunks per LDS row

__device__ __forceinline__ bf16x8 zero8() {
  bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
  return z;
}

// 8 consecutive elements of row `row` starting at column k (zero beyond the matrix); vec: 16-byte load allowed
__device__ __forceinline__ bf16x8 load8(const bf16* __restrict__ base, int64_t ld, int64_t row, int64_t rows, int64_t k,
                                        int64_t K, bool vec) {
  if (row >= rows || k >= K) return zero8();
  const bf16* p = base + row * ld + k;
  if (vec && k + 8 <= K) return *reinterpret_cast<const bf16x8*>(p);
  bf16x8 v = zero8();
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (k + i < K) v[i] = p[i];
  return v;
}

__global__ void __launch_bounds__(256, 2)
gemm_bf16_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ Bt, int64_t ldb,
                 bf16* __restrict__ C, int64_t ldc, int64_t m, int64_t n, int64_t K, int vec_in, int vec_out) {
  __shared__ __attribute__((aligned(16))) bf16 smem[(BM + BN) * BK];  // 32 KB: A tile, B tile; then the C tile
  bf16x8* As = reinterpret_cast<bf16x8*>(smem);
  bf16x8* Bs = As + BM * CHUNKS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order is not needed here: consecutive workgroups share the (small) B operand through L2
  const int64_t m0 = (int64_t)blockIdx.y * BM, n0 = (int64_t)blockIdx.x * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.0f;

  bf16x8 pa[4], pb[4];
  auto fetch = [&](int64_t k0) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = tid + 256 * v, row = e / CHUNKS, ch = e % CHUNKS;
      pa[v] = load8(A, lda, m0 + row, m, k0 + 8 * ch, K, vec_in);
      pb[v] = load8(Bt, ldb, n0 + row, n, k0 + 8 * ch, K, vec_in);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = tid + 256 * v, row = e / CHUNKS, ch = e % CHUNKS;
      As[row * CHUNKS + (ch ^ (row & 7))] = pa[v];
      Bs[row * CHUNKS + (ch ^ (row & 7))] = pb[v];
    }
  };

  fetch(0);
  for (int64_t k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    stash();
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);  // next tile in flight under the MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int ra = wr * 64 + 32 * t + r, rb = wc * 64 + 32 * t + r;
        af[t] = As[ra * CHUNKS + ((2 * ks + h) ^ (ra & 7))];
        bfr[t] = Bs[rb * CHUNKS + ((2 * ks + h) ^ (rb & 7))];
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
  }
  // ---- C tile through LDS: element (row, col) of the 128 x 128 tile at smem[row * 128 + col]
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = wr * 64 + 32 * a + (i & 3) + 8 * (i >> 2) + 4 * h, col = wc * 64 + 32 * b + r;
        smem[row * BN + col] = (bf16)acc[a][b][i];
      }
  __syncthreads();
  const bf16x8* Cs = reinterpret_cast<const bf16x8*>(smem);
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int e = tid + 256 * v, row = e / (BN / 8), ch = e % (BN / 8);
    const int64_t gr = m0 + row, gc = n0 + 8 * ch;
    if (gr >= m || gc >= n) continue;
    const bf16x8 val = Cs[e];
    bf16* dst = C + gr * ldc + gc;
    if (vec_out && gc + 8 <= n) {
      *reinterpret_cast<bf16x8*>(dst) = val;
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (gc + i < n) dst[i] = val[i];
    }
  }
}

// out (cols x rows) = in (rows x cols)^T, 32 x 32 tiles through LDS
__global__ void __launch_bounds__(256)
transpose_bf16_kernel(const bf16* __restrict__ in, int64_t rows, int64_t cols, int64_t ld_in, bf16* __restrict__ out,
                      int64_t ld_out) {
  __shared__ bf16 tile[32][34];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  for (int y = ty; y < 32; y += 8)
    if (r0 + y < rows && c0 + tx < cols) tile[y][tx] = in[(r0 + y) * ld_in + c0 + tx];
  __syncthreads();
  for (int y = ty; y < 32; y += 8)
    if (c0 + y < cols && r0 + tx < rows) out[(c0 + y) * ld_out + r0 + tx] = tile[tx][y];
}

template <typename TI, typename TO>
__global__ void __launch_bounds__(256) convert_kernel(const TI* __restrict__ x, int64_t n, TO* __restrict__ y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = (TO)(float)x[i];
}

inline int grid1d(int64_t n) {
  return (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(n, 256), 1), (int64_t)ndmps::kNumCU * 8);
}

}  // namespace

extern "C" int64_t ndmps_gemm_bf16_workspace_bytes(int transB, int64_t n, int64_t k) {
  return transB ? 0 : ndmps::round_up(n * k * 2, 256) + 256;  // transposed copy of the right operand
}

// C (m, n) = A (m, k) op(B), op(B) = B (k, n) for transB == 0, B^T with B (n, k) for transB == 1; all bf16
// row-major, fp32 accumulation.  d_ws: ndmps_gemm_bf16_workspace_bytes(transB, n, k) bytes (may be NULL when 0).
extern "C" int ndmps_gemm_bf16(int transB, int64_t m, int64_t n, int64_t k, const void* d_A, int64_t lda,
                               const void* d_B, int64_t ldb, void* d_C, int64_t ldc, void* d_ws, int64_t ws_bytes,
                               ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_A && d_B && d_C, "NULL GEMM operand");
  NDMPS_REQUIRE(m > 0 && n > 0 && k > 0, "bad GEMM extents m=%lld n=%lld k=%lld", (long long)m, (long long)n,
                (long long)k);
  NDMPS_REQUIRE(lda >= k && ldc >= n && ldb >= (transB ? k : n), "leading dimension smaller than a row");
  NDMPS_REQUIRE(ndmps::ceil_div(m, BM) < 65536 * 32768LL, "too many row tiles");
  hipStream_t s = (hipStream_t)stream;
  const bf16* Bt = (const bf16*)d_B;
  int64_t ldbt = ldb;
  if (!transB) {
    const int64_t need = ndmps_gemm_bf16_workspace_bytes(0, n, k);
    if (!d_ws || ws_bytes < need) {
      ndmps::set_error("bf16 GEMM workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
      return NDMPS_EWORKSPACE;
    }
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)ndmps::ceil_div(n, 32), (unsigned)ndmps::ceil_div(k, 32)),
                       dim3(256), 0, s, (const bf16*)d_B, k, n, ldb, (bf16*)d_ws, k);
    Bt = (const bf16*)d_ws;
    ldbt = k;
  }
  const int vec_in = (lda % 8 == 0 && ldbt % 8 == 0 && ((uintptr_t)d_A % 16) == 0 && ((uintptr_t)Bt % 16) == 0) ? 1 : 0;
  const int vec_out = (ldc % 8 == 0 && ((uintptr_t)d_C % 16) == 0) ? 1 : 0;
  // grid.y is limited to 65535: fold very tall problems into several launches
  const int64_t row_tiles = ndmps::ceil_div(m, BM);
  for (int64_t t0 = 0; t0 < row_tiles; t0 += 65535) {
    const int64_t tiles = std::min<int64_t>(65535, row_tiles - t0);
    const int64_t r0 = t0 * BM;
    hipLaunchKernelGGL(gemm_bf16_kernel, dim3((unsigned)ndmps::ceil_div(n, BN), (unsigned)tiles), dim3(256), 0, s,
                       (const bf16*)d_A + r0 * lda, lda, Bt, ldbt, (bf16*)d_C + r0 * ldc, ldc, std::min(m - r0, tiles * BM),
                       n, k, vec_in, vec_out);
  }
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_convert_bf16_to_f32(const void* d_x, int64_t n, float* d_y, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && d_y && n >= 0, "bad convert argument");
  if (n == 0) return NDMPS_OK;
  hipLaunchKernelGGL((convert_kernel<bf16, float>), dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (const bf16*)d_x, n,
                     d_y);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_convert_f32_to_bf16(const float* d_x, int64_t n, void* d_y, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && d_y && n >= 0, "bad convert argument");
  if (n == 0) return NDMPS_OK;
  hipLaunchKernelGGL((convert_kernel<float, bf16>), dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, d_x, n,
                     (bf16*)d_y);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
