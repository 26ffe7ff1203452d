// Dense building blocks on the CDNA4 matrix cores.
//
//   ndmps_sgemm  : C = op(A) op(B), fp32 in / fp32 accumulate, v_mfma_f32_32x32x2_f32
//   ndmps_dgemm  : same in fp64, v_mfma_f64_16x16x4_f64
//   ndmps_gram_f32 : G = A^T A with fp32 A and fp64 products/accumulation (exact products,
//                  one rounding per add) -- the small-side Gram of the per-site SVD.
//
// They stand in for the LAPACK/BLAS calls NumPy makes inside quimb for the reference
// (dgemm via tensordot in `mps ^ ...`, core/ndmps.py:140; the SVD's internal products in
// from_dense, core/ndmps.py:74).  GEMMs here are genuine dense GEMMs (bond x bond x phys);
// nothing is reshaped to reach the matrix cores.
//
// Layout notes (wave64): for the f32 32x32x2 MFMA lane l feeds A[i=l&31][k=l>>5] and
// B[k=l>>5][j=l&31]; for the f64 16x16x4 MFMA A[i=l&15][k=l>>4], B[k=l>>4][j=l&15].  Both
// operand tiles are therefore staged k-major in LDS (As[k][m], Bs[k][n]) so a fragment
// read is 32 (16) consecutive words per half (quarter) wave: conflict-free ds_read.
#include <stdlib.h>

#include <algorithm>

#include <mutex>
#include <type_traits>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;

template <>
struct Mfma<float> {
  static constexpr int MT = 32;   // tile edge
  static constexpr int KS = 2;    // k per instruction
  static constexpr int NACC = 16; // accumulator registers per lane
  typedef f32x16 acc_t;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int reg, int lane) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
  }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 31; }
  static __device__ __forceinline__ int frag_idx(int lane) { return lane & 31; }
  static __device__ __forceinline__ int frag_k(int lane) { return lane >> 5; }
};

template <>
struct Mfma<double> {
  static constexpr int MT = 16;
  static constexpr int KS = 4;
  static constexpr int NACC = 4;
  typedef f64x4 acc_t;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int reg, int lane) { return (lane >> 4) + 4 * reg; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_idx(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_k(int lane) { return lane >> 4; }
};

// ----------------------------------------------------------------------------------
// Generic tiled GEMM.  256 threads = 4 waves laid out WAVES_M x WAVES_N over a BM x BN
// block tile; BK-deep k-tiles staged through LDS (k-major for both operands).  The next
// k-tile is fetched into registers while the current one is consumed (one LDS buffer, two
// barriers per k-tile).  VEC: every operand row is a multiple of 4 elements and 16-byte
// (32-byte for fp64) aligned, so a thread moves 4 consecutive elements per global access;
// otherwise element-wise guarded loads (any shape, any leading dimension).
// ----------------------------------------------------------------------------------
template <typename T>
struct alignas(sizeof(T) * 4) Quad {
  T v[4];
};

// Optional table-driven addressing (IDX): element (m, k) of A at A[a_row[m] + a_col[k]], element (m, n) of C
// at C[c_row[m] + c_col[n]]; a NULL pair means dense row-major for that operand.  This is how the index
// permutation of the reshape stage rides on a product: the offset of a site-order element is additive over
// sites (permute.hip), so for any split of the sites into a row part and a column part it is
// RowOff[r] + ColOff[c].  With VEC, a_col must come in aligned runs of 4 consecutive offsets.
struct GemmIndex {
  const int64_t* a_row;
  const int64_t* a_col;
  const int64_t* c_row;
  const int64_t* c_col;
  int guarded;  // A/B (NDMPS_GEMM_GUARDED): every tile through the guarded fetch
};
inline int gemm_guarded_env() { return getenv("NDMPS_GEMM_GUARDED") ? 1 : 0; }

// operands of a batch of products of one shape: product blockIdx.z uses a[z], b[z], c[z] (kernel arguments)
constexpr int kGemmMaxBatch = 64;
struct GemmBatchPtrs {
  const void* a[kGemmMaxBatch];
  const void* b[kGemmMaxBatch];
  void* c[kGemmMaxBatch];
};

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, bool VEC, bool IDX>
__device__ __forceinline__ void
gemm_body(int64_t M, int64_t N, int64_t K, const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
          T* __restrict__ C, int64_t ldc, const GemmIndex& ix) {
  using MF = Mfma<T>;
  constexpr int BK = 16;
  constexpr int MT = MF::MT;
  constexpr int TM = BM / (WAVES_M * MT);
  constexpr int TN = BN / (WAVES_N * MT);
  constexpr int PAD = 4;
  constexpr int A_PER = BM * BK / 256, B_PER = BN * BK / 256;  // elements per thread and k-tile
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(TM >= 1 && TN >= 1, "tile too small");
  constexpr bool VA = VEC && A_PER % 4 == 0, VB = VEC && B_PER % 4 == 0;  // per-operand vector staging

  __shared__ T As[BK][BM + PAD];
  __shared__ T Bs[BK][BN + PAD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N;
  const int wn = wave % WAVES_N;
  const int64_t m0 = (int64_t)blockIdx.x * BM;  // x: row blocks (can exceed 65535)
  const int64_t n0 = (int64_t)blockIdx.y * BN;

  typename MF::acc_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < MF::NACC; ++r) acc[i][j][r] = (T)0;

  T ra[A_PER], rb[B_PER];

  // table-addressed A (the volume read through the index permutation): the row offsets of this thread's quads
  // are loop-invariant (registers) and the column offsets of the whole K range sit in LDS -- looked up in global
  // memory per k-tile they put two dependent L2 round trips in front of every tile (first projection of a group:
  // 1.14 ms for 2.1 GB)
  constexpr int kIdxK = 1024;
  __shared__ int64_t acol_s[(IDX && !TA && VEC) ? kIdxK : 1];
  int64_t arow_q[(A_PER / 4) > 0 ? (A_PER / 4) : 1];
  const bool idx_fast = IDX && !TA && VA && ix.a_row != nullptr && K <= kIdxK;
  if (IDX && !TA && VA) {
    if (idx_fast) {
      for (int64_t kk = tid; kk < K; kk += 256) acol_s[kk] = ix.a_col[kk];
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) {
        const int m = (tid + 256 * i) / (BK / 4);
        arow_q[i] = m0 + m < M ? ix.a_row[m0 + m] : 0;
      }
      __syncthreads();
    }
  }

  // Tiles that lie inside the matrices (and k-tiles inside K) are fetched by straight-line code: a load inside a
  // per-lane guard sits in an exec-masked block behind its own s_waitcnt vmcnt(0) (see gram128_kernel), so the
  // guarded quads of a k-tile went out one memory round trip after the other.
  const bool inner_a = VA && m0 + BM <= M && !ix.guarded, inner_b = VB && n0 + BN <= N && !ix.guarded;
  // op(A)[m][k]: stored (M, K) unless TA (then (K, M)).  contiguous axis: k unless TA (then m)
  auto fetch_a = [&](int64_t k0) {
    if (VA && inner_a && k0 + BK <= K) {
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) {
        const int e = tid + 256 * i;  // quad index
        const T* src;
        if (TA) {
          src = A + (k0 + e / (BM / 4)) * lda + m0 + (e % (BM / 4)) * 4;
        } else {
          const int m = e / (BK / 4), k = (e % (BK / 4)) * 4;
          if (IDX && idx_fast) src = A + arow_q[i] + acol_s[k0 + k];
          else if (IDX && ix.a_row) src = A + ix.a_row[m0 + m] + ix.a_col[k0 + k];
          else src = A + (m0 + m) * lda + k0 + k;
        }
        const Quad<T> q = *reinterpret_cast<const Quad<T>*>(src);
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[4 * i + j] = q.v[j];
      }
    } else if (VA) {
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) {
        const int e = tid + 256 * i;  // quad index
        Quad<T> q;
        q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
        if (TA) {
          const int k = e / (BM / 4), m = (e % (BM / 4)) * 4;
          if (k0 + k < K && m0 + m < M) q = *reinterpret_cast<const Quad<T>*>(A + (k0 + k) * lda + m0 + m);
        } else {
          const int m = e / (BK / 4), k = (e % (BK / 4)) * 4;
          if (k0 + k < K && m0 + m < M) {
            if (IDX && idx_fast) q = *reinterpret_cast<const Quad<T>*>(A + arow_q[i] + acol_s[k0 + k]);
            else if (IDX && ix.a_row) q = *reinterpret_cast<const Quad<T>*>(A + ix.a_row[m0 + m] + ix.a_col[k0 + k]);
            else q = *reinterpret_cast<const Quad<T>*>(A + (m0 + m) * lda + k0 + k);
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[4 * i + j] = q.v[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int e = tid + 256 * i;
        int m, k;
        if (TA) {
          k = e / BM;
          m = e % BM;
        } else {
          m = e / BK;
          k = e % BK;
        }
        const int64_t gk = k0 + k, gm = m0 + m;
        if (IDX && ix.a_row) ra[i] = (gk < K && gm < M) ? A[ix.a_row[gm] + ix.a_col[gk]] : (T)0;
        else ra[i] = (gk < K && gm < M) ? (TA ? A[gk * lda + gm] : A[gm * lda + gk]) : (T)0;
      }
    }
  };
  auto fetch_b = [&](int64_t k0) {
    if (VB && inner_b && k0 + BK <= K) {
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) {
        const int e = tid + 256 * i;
        const T* src = TB ? B + (n0 + e / (BK / 4)) * ldb + k0 + (e % (BK / 4)) * 4
                          : B + (k0 + e / (BN / 4)) * ldb + n0 + (e % (BN / 4)) * 4;
        const Quad<T> q = *reinterpret_cast<const Quad<T>*>(src);
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[4 * i + j] = q.v[j];
      }
    } else if (VB) {
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) {
        const int e = tid + 256 * i;
        Quad<T> q;
        q.v[0] = q.v[1] = q.v[2] = q.v[3] = (T)0;
        if (TB) {
          const int n = e / (BK / 4), k = (e % (BK / 4)) * 4;
          if (k0 + k < K && n0 + n < N) q = *reinterpret_cast<const Quad<T>*>(B + (n0 + n) * ldb + k0 + k);
        } else {
          const int k = e / (BN / 4), n = (e % (BN / 4)) * 4;
          if (k0 + k < K && n0 + n < N) q = *reinterpret_cast<const Quad<T>*>(B + (k0 + k) * ldb + n0 + n);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[4 * i + j] = q.v[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int e = tid + 256 * i;
        int n, k;
        if (TB) {
          n = e / BK;
          k = e % BK;
        } else {
          k = e / BN;
          n = e % BN;
        }
        const int64_t gk = k0 + k, gn = n0 + n;
        rb[i] = (gk < K && gn < N) ? (TB ? B[gn * ldb + gk] : B[gk * ldb + gn]) : (T)0;
      }
    }
  };
  auto commit = [&]() {
    if (VA) {
#pragma unroll
      for (int i = 0; i < A_PER / 4; ++i) {
        const int e = tid + 256 * i;
        if (TA) {
          const int k = e / (BM / 4), m = (e % (BM / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) As[k][m + j] = ra[4 * i + j];
        } else {
          const int m = e / (BK / 4), k = (e % (BK / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) As[k + j][m] = ra[4 * i + j];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int e = tid + 256 * i;
        if (TA) As[e / BM][e % BM] = ra[i];
        else As[e % BK][e / BK] = ra[i];
      }
    }
    if (VB) {
#pragma unroll
      for (int i = 0; i < B_PER / 4; ++i) {
        const int e = tid + 256 * i;
        if (TB) {
          const int n = e / (BK / 4), k = (e % (BK / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[k + j][n] = rb[4 * i + j];
        } else {
          const int k = e / (BN / 4), n = (e % (BN / 4)) * 4;
#pragma unroll
          for (int j = 0; j < 4; ++j) Bs[k][n + j] = rb[4 * i + j];
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int e = tid + 256 * i;
        if (TB) Bs[e % BK][e / BK] = rb[i];
        else Bs[e / BN][e % BN] = rb[i];
      }
    }
  };

  fetch_a(0);
  fetch_b(0);
  const int fi = MF::frag_idx(lane);
  const int fk = MF::frag_k(lane);
  for (int64_t k0 = 0; k0 < K; k0 += BK) {
    __syncthreads();  // previous k-tile fully consumed
    commit();
    __syncthreads();
    if (k0 + BK < K) {  // next k-tile in flight under the MFMAs
      fetch_a(k0 + BK);
      fetch_b(k0 + BK);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += MF::KS) {
      T a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kk + fk][(wm * TM + i) * MT + fi];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kk + fk][(wn * TN + j) * MT + fi];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = MF::mma(a[i], b[j], acc[i][j]);
    }
  }

  // ---- epilogue
  if constexpr (IDX && MF::MT == 32 && BM == 128 && BN == 128 && sizeof(T) == 4) {
    if (ix.c_row) {
      // table-addressed C of the big tile (the last chain product: the reconstructed volume, written once): every
      // 32 x 32 accumulator tile goes through a wave-private LDS patch so that a lane owns FOUR consecutive columns
      // of a row -- 16-byte stores wherever the four column offsets are consecutive (64-byte runs at 256^3)
      // instead of 4-byte ones.  All offsets are requested before the first store.
      __shared__ float cstage[4][32][36];
      float (*st)[36] = cstage[wave];
      const int rl0 = lane >> 3, cl = (lane & 7) * 4;
      int64_t roff[TM][4], coff[TN][4];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int64_t col = n0 + (wn * TN + j) * MT + cl;
#pragma unroll
        for (int e = 0; e < 4; ++e) coff[j][e] = col + e < N ? ix.c_col[col + e] : -1;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int64_t row = m0 + (wm * TM + i) * MT + rl0 + 8 * ps;
          roff[i][ps] = row < M ? ix.c_row[row] : -1;
        }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
          for (int r = 0; r < MF::NACC; ++r) st[MF::acc_row(r, lane)][MF::acc_col(lane)] = acc[i][j][r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const bool run = coff[j][0] >= 0 && coff[j][3] == coff[j][0] + 3 && coff[j][1] == coff[j][0] + 1 &&
                           coff[j][2] == coff[j][0] + 2;
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const float4 v = *reinterpret_cast<const float4*>(&st[rl0 + 8 * ps][cl]);
            const int64_t ro = roff[i][ps];
            if (ro < 0) continue;
            if (run && ((ro + coff[j][0]) & 3) == 0) {
              *reinterpret_cast<float4*>(C + ro + coff[j][0]) = v;
            } else {
              if (coff[j][0] >= 0) C[ro + coff[j][0]] = v.x;
              if (coff[j][1] >= 0) C[ro + coff[j][1]] = v.y;
              if (coff[j][2] >= 0) C[ro + coff[j][2]] = v.z;
              if (coff[j][3] >= 0) C[ro + coff[j][3]] = v.w;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();  // the patch is rewritten by the next tile
        }
      return;
    }
  }
  if (IDX && ix.c_row) {
    // table-addressed C: every offset this lane needs is requested first (TM * NACC row offsets, TN column offsets),
    // then the stores go out -- one dependent table load in front of every store made the epilogue a chain of
    // L2 round trips (the last chain product wrote the volume at 1 TB/s)
    int64_t roff[TM][MF::NACC], coff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + (wn * TN + j) * MT + MF::acc_col(lane);
      coff[j] = col < N ? ix.c_col[col] : -1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < MF::NACC; ++r) {
        const int64_t row = m0 + (wm * TM + i) * MT + MF::acc_row(r, lane);
        roff[i][r] = row < M ? ix.c_row[row] : -1;
      }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < MF::NACC; ++r)
          if (roff[i][r] >= 0 && coff[j] >= 0) C[roff[i][r] + coff[j]] = acc[i][j][r];
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int64_t col = n0 + (wn * TN + j) * MT + MF::acc_col(lane);
#pragma unroll
      for (int r = 0; r < MF::NACC; ++r) {
        const int64_t row = m0 + (wm * TM + i) * MT + MF::acc_row(r, lane);
        if (row < M && col < N) C[row * ldc + col] = acc[i][j][r];
      }
    }
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, bool VEC, bool IDX = false>
__global__ void __launch_bounds__(256, 2)
gemm_kernel(int64_t M, int64_t N, int64_t K, const T* __restrict__ A, int64_t lda,
            const T* __restrict__ B, int64_t ldb, T* __restrict__ C, int64_t ldc, GemmIndex ix = GemmIndex()) {
  gemm_body<T, BM, BN, WAVES_M, WAVES_N, TA, TB, VEC, IDX>(M, N, K, A, lda, B, ldb, C, ldc, ix);
}

// the same product for every operand triple of a batch (grid.z): what a lockstep group of volumes needs at a site
// of the sweep or a stage of the chain -- one launch instead of one per volume
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, bool TA, bool TB, bool VEC, bool IDX = false>
__global__ void __launch_bounds__(256, 2)
gemm_batched_kernel(int64_t M, int64_t N, int64_t K, GemmBatchPtrs p, int64_t lda, int64_t ldb, int64_t ldc,
                    GemmIndex ix = GemmIndex()) {
  gemm_body<T, BM, BN, WAVES_M, WAVES_N, TA, TB, VEC, IDX>(M, N, K, static_cast<const T*>(p.a[blockIdx.z]), lda,
                                                           static_cast<const T*>(p.b[blockIdx.z]), ldb,
                                                           static_cast<T*>(p.c[blockIdx.z]), ldc, ix);
}

template <typename T, int BM, int BN, int WAVES_M, int WAVES_N>
int launch_gemm(int transA, int transB, int64_t m, int64_t n, int64_t k, const T* A, int64_t lda,
                const T* B, int64_t ldb, T* C, int64_t ldc, hipStream_t stream, const GemmBatchPtrs* bp = nullptr,
                int batch = 1) {
  dim3 grid((unsigned)ndmps::ceil_div(m, BM), (unsigned)ndmps::ceil_div(n, BN), (unsigned)batch);
  dim3 block(256);
  GemmIndex gi{nullptr, nullptr, nullptr, nullptr, gemm_guarded_env()};
  // vector path: whole quads are either inside or outside every bound, and 4-element aligned
  const uintptr_t al = sizeof(T) * 4;
  bool vec = lda % 4 == 0 && ldb % 4 == 0 && k % 4 == 0 && (!transA || m % 4 == 0) && (transB || n % 4 == 0);
  if (bp) {
    for (int z = 0; z < batch; ++z) vec = vec && (uintptr_t)bp->a[z] % al == 0 && (uintptr_t)bp->b[z] % al == 0;
  } else {
    vec = vec && (uintptr_t)A % al == 0 && (uintptr_t)B % al == 0;
  }
#define NDMPS_GEMM_LAUNCH(TA_, TB_, V_)                                                                            \
  do {                                                                                                              \
    if (bp)                                                                                                         \
      hipLaunchKernelGGL((gemm_batched_kernel<T, BM, BN, WAVES_M, WAVES_N, TA_, TB_, V_>), grid, block, 0, stream,  \
                         m, n, k, *bp, lda, ldb, ldc, gi);                                                          \
    else                                                                                                            \
      hipLaunchKernelGGL((gemm_kernel<T, BM, BN, WAVES_M, WAVES_N, TA_, TB_, V_>), grid, block, 0, stream, m, n, k,  \
                         A, lda, B, ldb, C, ldc, gi);                                                               \
  } while (0)
#define NDMPS_GEMM_TRANS(V_)                                     \
  do {                                                           \
    if (transA && transB) NDMPS_GEMM_LAUNCH(true, true, V_);     \
    else if (transA) NDMPS_GEMM_LAUNCH(true, false, V_);         \
    else if (transB) NDMPS_GEMM_LAUNCH(false, true, V_);         \
    else NDMPS_GEMM_LAUNCH(false, false, V_);                    \
  } while (0)
  if (vec) NDMPS_GEMM_TRANS(true);
  else NDMPS_GEMM_TRANS(false);
#undef NDMPS_GEMM_TRANS
#undef NDMPS_GEMM_LAUNCH
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

template <typename T>
int gemm_check(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, int transA, const T* B,
               int64_t ldb, int transB, T* C, int64_t ldc) {
  NDMPS_REQUIRE(m >= 0 && n >= 0 && k >= 0, "negative GEMM extent");
  NDMPS_REQUIRE(A && B && C, "NULL GEMM operand");
  NDMPS_REQUIRE(lda >= (transA ? m : k) && ldb >= (transB ? k : n) && ldc >= n,
                "leading dimension too small (lda=%lld ldb=%lld ldc=%lld)", (long long)lda,
                (long long)ldb, (long long)ldc);
  NDMPS_REQUIRE(ndmps::ceil_div(n, 16) < 65536, "GEMM n=%lld exceeds grid.y", (long long)n);
  NDMPS_REQUIRE(ndmps::ceil_div(m, 32) < 2147483647LL, "GEMM m=%lld exceeds grid.x", (long long)m);
  return NDMPS_OK;
}

// ----------------------------------------------------------------------------------
// Gram matrix G = A^T A, A (m, n) fp32 row-major, G fp64.
// Lane l of a wave reads A[r + (l>>4)][c + (l&15)] -- straight row-major segments, no
// transposition -- converts to f64 and feeds v_mfma_f64_16x16x4_f64 as both operands.
// A workgroup owns one (T*16)^2 tile of the upper triangle over a slab of rows; its 4
// waves interleave 4-row k-steps and fold their accumulators through LDS; the slab result
// goes to a partial buffer, reduced in fixed order (deterministic) by gram_reduce_kernel.
// ----------------------------------------------------------------------------------
template <int T, typename TIN>
__global__ void __launch_bounds__(256, 2)
gram_partial_kernel(const TIN* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                    double* __restrict__ partial, int n_tiles_1d, int64_t rows_per_slab) {
  constexpr int TS = 16 * T;  // tile edge
  __shared__ double red[TS][TS + 1];

  // decode upper-triangular tile index
  int tile = blockIdx.x, ti = 0;
  while (tile >= n_tiles_1d - ti) {
    tile -= n_tiles_1d - ti;
    ++ti;
  }
  const int tj = ti + tile;
  const int64_t i0 = (int64_t)ti * TS, j0 = (int64_t)tj * TS;
  const int64_t slab = blockIdx.y;
  const int64_t r_begin = slab * rows_per_slab;
  const int64_t r_end = min(m, r_begin + rows_per_slab);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane >> 4, lc = lane & 15;

  f64x4 acc[T][T];
#pragma unroll
  for (int a = 0; a < T; ++a)
#pragma unroll
    for (int b = 0; b < T; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};

  for (int64_t r = r_begin + 4 * wave; r < r_end; r += 16) {
    const int64_t row = r + lr;
    const bool row_ok = row < r_end;
    double av[T], bv[T];
#pragma unroll
    for (int a = 0; a < T; ++a) {
      const int64_t ca = i0 + 16 * a + lc;
      const int64_t cb = j0 + 16 * a + lc;
      av[a] = (row_ok && ca < n) ? ndmps::to_f64(A[row * lda + ca]) : 0.0;
      bv[a] = (row_ok && cb < n) ? ndmps::to_f64(A[row * lda + cb]) : 0.0;
    }
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int b = 0; b < T; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
  }

  // fold the 4 waves through LDS, one after the other (fixed order)
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < T; ++a)
#pragma unroll
        for (int b = 0; b < T; ++b)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int rr = 16 * a + lr + 4 * reg, cc = 16 * b + lc;
            if (w == 0) red[rr][cc] = acc[a][b][reg];
            else red[rr][cc] += acc[a][b][reg];
          }
    }
    __syncthreads();
  }
  double* out = partial + ((int64_t)slab * gridDim.x + blockIdx.x) * (TS * TS);
  for (int e = tid; e < TS * TS; e += 256) out[e] = red[e / TS][e % TS];
}

// Slab reduce, fixed summation order (deterministic), in up to two levels so that no thread
// walks more than ~32 slabs: grid (n_tiles, TS*TS/256, n_groups).  Group g sums slabs
// [g*per_group, (g+1)*per_group); `final` writes the (mirrored) tile into G, otherwise the group
// sums go to `out` laid out like a partial buffer with n_groups slabs.
template <int TS>
__global__ void __launch_bounds__(256)
tile_reduce_kernel(const double* __restrict__ partial, int n_slabs, int per_group, int n_tiles_1d,
                   int n_tiles, double* __restrict__ out, double* __restrict__ G, int64_t n, int final,
                   const int32_t* __restrict__ perm = nullptr) {  // perm: element (r, c) goes to (perm[r], perm[c])
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= TS * TS) return;
  const int s0 = blockIdx.z * per_group, s1 = min(n_slabs, s0 + per_group);
  const double* src = partial + (int64_t)blockIdx.x * (TS * TS) + e;
  const int64_t slab_stride = (int64_t)n_tiles * (TS * TS);
  double s = 0.0;
  int sl = s0;
  for (; sl + 4 <= s1; sl += 4) {
    const double v0 = src[(sl + 0) * slab_stride], v1 = src[(sl + 1) * slab_stride];
    const double v2 = src[(sl + 2) * slab_stride], v3 = src[(sl + 3) * slab_stride];
    s = (((s + v0) + v1) + v2) + v3;
  }
  for (; sl < s1; ++sl) s += src[sl * slab_stride];
  if (!final) {
    out[((int64_t)blockIdx.z * n_tiles + blockIdx.x) * (TS * TS) + e] = s;
    return;
  }
  int tile = blockIdx.x, ti = 0;
  while (tile >= n_tiles_1d - ti) {
    tile -= n_tiles_1d - ti;
    ++ti;
  }
  const int tj = ti + tile;
  const int64_t r = (int64_t)ti * TS + e / TS, c = (int64_t)tj * TS + e % TS;
  if (r < n && c < n && (ti != tj || c >= r)) {  // diagonal tiles: upper part mirrored -> exactly symmetric
    const int64_t pr = perm ? perm[r] : r, pc = perm ? perm[c] : c;
    G[pr * n + pc] = s;
    G[pc * n + pr] = s;
  }
}

constexpr int kReduceGroup = 16;

template <int TS>
int launch_tile_reduce(double* partial, int n_slabs, int n_tiles_1d, int n_tiles, double* G, int64_t n,
                       hipStream_t s, const int32_t* perm = nullptr) {
  const unsigned ey = (TS * TS + 255) / 256;
  if (n_slabs <= 2 * kReduceGroup) {
    hipLaunchKernelGGL(tile_reduce_kernel<TS>, dim3(n_tiles, ey, 1), dim3(256), 0, s, partial, n_slabs, n_slabs,
                       n_tiles_1d, n_tiles, (double*)nullptr, G, n, 1, perm);
  } else {
    // level 1 writes its group sums behind the slabs (the workspace has room for them)
    const int groups = (n_slabs + kReduceGroup - 1) / kReduceGroup;
    double* lvl = partial + (int64_t)n_slabs * n_tiles * (TS * TS);
    hipLaunchKernelGGL(tile_reduce_kernel<TS>, dim3(n_tiles, ey, groups), dim3(256), 0, s, partial, n_slabs,
                       kReduceGroup, n_tiles_1d, n_tiles, lvl, G, n, 0);
    hipLaunchKernelGGL(tile_reduce_kernel<TS>, dim3(n_tiles, ey, 1), dim3(256), 0, s, lvl, groups, groups,
                       n_tiles_1d, n_tiles, (double*)nullptr, G, n, 1, perm);
  }
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

// ----------------------------------------------------------------------------------
// Gram for wide column counts (n >= 128): a workgroup owns a 128 x 128 tile of the upper
// triangle over a slab of rows; 32-row chunks of the two 128-column panels are staged in LDS
// (16-byte global loads, each element of A fetched once per tile row/column instead of once
// per 64-wide tile and per lane), the next chunk is prefetched into registers while the four
// waves (2 x 2, 64 x 64 each, 16 f64 MFMA accumulators) consume the current one.
// LDS rows are padded by 16 floats: a fragment read (4 rows x 16 columns) is conflict-free.
// ----------------------------------------------------------------------------------
constexpr int GW_KB = 32;    // rows per staged chunk

// GW_TS: tile edge, 128 (n >= 128) or 64 (64 <= n < 128); each of the 2 x 2 waves owns a
// (GW_TS/2)^2 sub-tile = (GW_TS/32)^2 MFMA accumulators
// four consecutive elements as fp32 (16-byte load for fp32 input, 8-byte load for bf16 input)
__device__ __forceinline__ float4 load4_as_f32(const float* p) { return *reinterpret_cast<const float4*>(p); }
// read-once streams: non-temporal (the lines are not kept in the caches in front of data that is read again)
__device__ __forceinline__ float4 load4_stream_f32(const float* p) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 load4_stream_f32(const __bf16* p) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const bf16x4 v = __builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p));
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ float4 load4_as_f32(const __bf16* p) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// row_off / col_off (both or neither): element (r, c) of A at A[row_off[r] + col_off[c]] instead of A[r lda + c] --
// the matrix is the C-order volume read through the index permutation (see GemmIndex); with vec_ok every aligned
// group of four columns has consecutive offsets.
template <int GW_TS, typename TIN>
__device__ __forceinline__ void gram_wide_body(const TIN* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                                               double* __restrict__ out, int n_tiles_1d, int64_t rows_per_slab, int vec_ok,
                                               const int64_t* __restrict__ row_off, const int64_t* __restrict__ col_off,
                                               int tile_id, int slab_id) {
  constexpr int GW_LD = GW_TS + 16;
  constexpr int NT = GW_TS / 32;       // MFMA tiles per wave and direction
  constexpr int SUB = GW_TS / 2;       // wave sub-tile edge
  constexpr int QPR = GW_TS / 4;       // float4 per staged row
  constexpr int VPT = GW_KB * QPR / 256;  // float4 per thread and panel
  __shared__ float Ai[GW_KB][GW_LD];
  __shared__ float Aj[GW_KB][GW_LD];

  int tile = tile_id, ti = 0;
  while (tile >= n_tiles_1d - ti) {
    tile -= n_tiles_1d - ti;
    ++ti;
  }
  const int tj = ti + tile;
  const bool diag = ti == tj;
  const int64_t i0 = (int64_t)ti * GW_TS, j0 = (int64_t)tj * GW_TS;
  const int64_t r_begin = (int64_t)slab_id * rows_per_slab;
  const int64_t r_end = min(m, r_begin + rows_per_slab);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane >> 4, lc = lane & 15;

  f64x4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};

  // staging map: GW_KB rows x QPR float4 per panel, VPT per thread
  float4 pi[VPT], pj[VPT];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int e = tid + 256 * v;
      const int rr = e / QPR, c4 = (e % QPR) * 4;
      const int64_t row = r0 + rr;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
      if (row < r_end && row_off) {  // gathered: offsets additive in (row, column); vec_ok guaranteed by the host
        const TIN* base = A + row_off[row];
        if (i0 + c4 + 3 < n) x = load4_as_f32(base + col_off[i0 + c4]);
        if (!diag && j0 + c4 + 3 < n) y = load4_as_f32(base + col_off[j0 + c4]);
      } else if (row < r_end) {
        const TIN* base = A + row * lda;
        if (vec_ok && i0 + c4 + 3 < n) x = load4_as_f32(base + i0 + c4);
        else {
          if (i0 + c4 + 0 < n) x.x = (float)base[i0 + c4 + 0];
          if (i0 + c4 + 1 < n) x.y = (float)base[i0 + c4 + 1];
          if (i0 + c4 + 2 < n) x.z = (float)base[i0 + c4 + 2];
          if (i0 + c4 + 3 < n) x.w = (float)base[i0 + c4 + 3];
        }
        if (!diag) {
          if (vec_ok && j0 + c4 + 3 < n) y = load4_as_f32(base + j0 + c4);
          else {
            if (j0 + c4 + 0 < n) y.x = (float)base[j0 + c4 + 0];
            if (j0 + c4 + 1 < n) y.y = (float)base[j0 + c4 + 1];
            if (j0 + c4 + 2 < n) y.z = (float)base[j0 + c4 + 2];
            if (j0 + c4 + 3 < n) y.w = (float)base[j0 + c4 + 3];
          }
        }
      }
      pi[v] = x;
      pj[v] = y;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int e = tid + 256 * v;
      const int rr = e / QPR, c4 = (e % QPR) * 4;
      *reinterpret_cast<float4*>(&Ai[rr][c4]) = pi[v];
      if (!diag) *reinterpret_cast<float4*>(&Aj[rr][c4]) = pj[v];
    }
  };

  fetch(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += GW_KB) {
    __syncthreads();  // previous chunk fully consumed
    stash();
    __syncthreads();
    if (r0 + GW_KB < r_end) fetch(r0 + GW_KB);  // prefetch under the MFMAs
    const float (*Bj)[GW_LD] = diag ? Ai : Aj;
#pragma unroll
    for (int k0 = 0; k0 < GW_KB; k0 += 4) {
      double av[NT], bv[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        av[a] = (double)Ai[k0 + lr][wr * SUB + 16 * a + lc];
        bv[a] = (double)Bj[k0 + lr][wc * SUB + 16 * a + lc];
      }
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }

#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int rr = wr * SUB + 16 * a + lr + 4 * reg, cc = wc * SUB + 16 * b + lc;
        out[rr * GW_TS + cc] = acc[a][b][reg];
      }
}

template <int GW_TS, typename TIN>
__global__ void __launch_bounds__(256, 2)
gram_wide_kernel(const TIN* __restrict__ A, int64_t m, int64_t n, int64_t lda,
                 double* __restrict__ partial, int n_tiles_1d, int64_t rows_per_slab, int vec_ok,
                 const int64_t* __restrict__ row_off = nullptr, const int64_t* __restrict__ col_off = nullptr) {
  gram_wide_body<GW_TS, TIN>(A, m, n, lda, partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (GW_TS * GW_TS),
                             n_tiles_1d, rows_per_slab, vec_ok, row_off, col_off, blockIdx.x, blockIdx.y);
}

// The same tiles for a whole lockstep group in ONE launch (blockIdx.z = matrix): the 64-column raw Gram of a bond cap
// of 32 (BASELINE configs 2 and 4) went out as one launch per volume, each cut into 512 short slabs to fill the GPU by
// itself (16.8 MB of partial tiles written and read back per 64 MB volume); a group shares the GPU, so slabs are long.
struct GramBatchPtrs {
  const void* a[64];
};
template <int GW_TS, typename TIN>
__global__ void __launch_bounds__(256, 4)  // a 64-column Gram is a stream: four workgroups per CU keep more of it in flight
gram_wide_batched_kernel(GramBatchPtrs ptrs, int64_t m, int64_t n, int64_t lda, double* __restrict__ partial, int n_tiles_1d,
                         int64_t rows_per_slab, int vec_ok, const int64_t* __restrict__ row_off,
                         const int64_t* __restrict__ col_off) {
  double* out = partial + (((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (GW_TS * GW_TS);
  gram_wide_body<GW_TS, TIN>(static_cast<const TIN*>(ptrs.a[blockIdx.z]), m, n, lda, out, n_tiles_1d, rows_per_slab, vec_ok,
                             row_off, col_off, blockIdx.x, blockIdx.y);
}
// ----------------------------------------------------------------------------------
// n == 64 exactly (the raw Gram of a bond cap of 32: d chi = 8 x 8 columns, 64 MB per 256^3 volume): a STREAM.
// The tile kernel above stages 32-row chunks in LDS behind two barriers and computes the full 64 x 64 tile (16 MFMAs
// per four rows): 1.26 ms per lockstep group of 32 alone, 2.2 ms inside a step -- 1 TB/s.  Here a wave reads four rows
// with ONE coalesced 16-byte load per lane (lane (i, k) = (lane % 16, lane / 16) takes columns 4i .. 4i + 3 of row
// k) and that float4 IS the four MFMA operands: component a belongs to the column block {4i + a}, so
// v_mfma_f64_16x16x4_f64(x_a, x_b) accumulates G[4i + a][4j + b] -- a column permutation that is undone when the tile is
// written.  Ten MFMAs per four rows (blocks a <= b), no LDS, no barrier in the loop, eight loads in flight per lane
// (offsets of a gathered operand one block further ahead).  The four waves of a workgroup take rows 4w .. 4w + 3 of
// every 16 and add their accumulators in a fixed order at the end; the partial tile has the layout
// tile_reduce_batched_kernel<64> expects.
template <typename TIN, bool GATHER>
__global__ void __launch_bounds__(256, 3)  // three waves per SIMD: one waiting for its loads leaves the MFMA pipe to two
gram64_stream_kernel(GramBatchPtrs ptrs, int64_t m, int64_t lda, double* __restrict__ partial, int64_t rows_per_slab,
                     const int64_t* __restrict__ row_off, const int64_t* __restrict__ col_off) {
  constexpr int U = 8;  // k-steps (of four rows per wave) per block
  const TIN* A = static_cast<const TIN*>(ptrs.a[blockIdx.z]);
  double* out = partial + ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * 4096;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, kk = lane >> 4;
  const int64_t r_begin = (int64_t)blockIdx.y * rows_per_slab;
  const int64_t r_end = min(m, r_begin + rows_per_slab);
  const int64_t n_steps = (r_end - r_begin + 15) / 16;
  const int64_t n_blocks = (n_steps + U - 1) / U;
  const int64_t coff = GATHER ? col_off[4 * li] : 4 * li;
  const int64_t row0 = r_begin + 4 * wave + kk;  // this lane's row at step 0; + 16 per step

  f64x4 acc[10];
#pragma unroll
  for (int q = 0; q < 10; ++q) acc[q] = (f64x4){0.0, 0.0, 0.0, 0.0};
#define NDMPS_GRAM64_STEP(v)                                                                                     \
  {                                                                                                              \
    const double x0 = (double)(v).x, x1 = (double)(v).y, x2 = (double)(v).z, x3 = (double)(v).w;                 \
    NDMPS_GRAM64_MFMAS(x0, x1, x2, x3)                                                                           \
  }
#define NDMPS_GRAM64_MFMAS(x0, x1, x2, x3)                                                                       \
  {                                                                                                              \
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, acc[0], 0, 0, 0);                                      \
    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x1, acc[1], 0, 0, 0);                                      \
    acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x2, acc[2], 0, 0, 0);                                      \
    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x3, acc[3], 0, 0, 0);                                      \
    acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, acc[4], 0, 0, 0);                                      \
    acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x2, acc[5], 0, 0, 0);                                      \
    acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x3, acc[6], 0, 0, 0);                                      \
    acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x2, acc[7], 0, 0, 0);                                      \
    acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, x3, acc[8], 0, 0, 0);                                      \
    acc[9] = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, x3, acc[9], 0, 0, 0);                                      \
  }
  // Full blocks (every row of every step exists): straight-line loads -- a load under a branch makes the compiler
  // wait for ALL outstanding loads (s_waitcnt vmcnt(0)) in front of the MFMAs, prefetched ones included.  Blocks
  // fetched ahead beyond the last full one repeat it (valid addresses, never used).
  const int64_t n_full = (r_end - r_begin) / (16 * U);
  if (n_full > 0) {
    // a ring of U loads per lane: step s takes slot s % U and refills it at once with step s + U (its offset -- the row's
    // entry of the table of a gathered operand -- was fetched U steps before that): a constant distance of U steps
    // (80 MFMAs) between a load and its use, no drain at block boundaries
    const int64_t last = n_full * U - 1;  // steps beyond the last full one repeat it (valid addresses, never used)
    auto offset_of = [&](int64_t step) {
      const int64_t row = row0 + 16 * min(step, last);
      return GATHER ? row_off[row] : row * lda;
    };
    float4 ring[U];
    int64_t roff[U];
#pragma unroll
    for (int u = 0; u < U; ++u) roff[u] = offset_of(u);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ring[u] = load4_stream_f32(A + roff[u] + coff);
      roff[u] = offset_of(U + u);
    }
    auto block = [&](int64_t blk) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        // the step's operands are converted BEFORE its slot is reloaded in place, and nothing moves across steps:
        // left to itself the scheduler hoists the block's eight loads to its top into fresh registers and copies them
        // into the loop-carried ones -- a copy that waits for the load it follows
        const double x0 = (double)ring[u].x, x1 = (double)ring[u].y, x2 = (double)ring[u].z, x3 = (double)ring[u].w;
        __builtin_amdgcn_sched_barrier(0);
        ring[u] = load4_stream_f32(A + roff[u] + coff);   // step (blk + 1) U + u
        roff[u] = offset_of((blk + 2) * U + u);
        NDMPS_GRAM64_MFMAS(x0, x1, x2, x3)
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // four blocks per trip: at a loop header the compiler waits for EVERY outstanding load (the newest was issued ten
    // MFMAs earlier), once per 32 steps then
    int64_t blk = 0;
    for (; blk + 4 <= n_full; blk += 4) {
      block(blk);
      block(blk + 1);
      block(blk + 2);
      block(blk + 3);
    }
    for (; blk < n_full; ++blk) block(blk);
  }
  // the ragged end of the slab, row by row
  for (int64_t row = row0 + 16 * U * n_full; row < r_end + 16; row += 16) {  // uniform trip count over the wave
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < r_end) v = load4_as_f32(A + (GATHER ? row_off[row] : row * lda) + coff);
    if (__builtin_amdgcn_ballot_w64(row < r_end) != 0) NDMPS_GRAM64_STEP(v)
  }
#undef NDMPS_GRAM64_STEP
#undef NDMPS_GRAM64_MFMAS

  // waves 3, 2, 1 hand their sums to wave 0 through LDS, one after the other (fixed order)
  __shared__ double hand[10 * 256];
#pragma unroll 1
  for (int src = 3; src >= 1; --src) {
    if (wave == src) {
#pragma unroll
      for (int q = 0; q < 10; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) hand[(q * 4 + reg) * 64 + lane] = acc[q][reg];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int q = 0; q < 10; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc[q][reg] += hand[(q * 4 + reg) * 64 + lane];
    }
    __syncthreads();
  }
  if (wave == 0) {
    // block q = (a, b), a <= b; accumulator element (i, j) = (kk + 4 reg, li) is G[4i + a][4j + b]
    constexpr int qa[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, qb[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
#pragma unroll
    for (int q = 0; q < 10; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 4 * (kk + 4 * reg) + qa[q], c = 4 * li + qb[q];
        out[r * 64 + c] = acc[q][reg];
        if (qa[q] != qb[q]) out[c * 64 + r] = acc[q][reg];
      }
  }
}

// G of matrix blockIdx.z: its slabs of tile blockIdx.x summed in order; upper part mirrored; optional un-permutation
template <int TS>
__global__ void __launch_bounds__(256)
tile_reduce_batched_kernel(const double* __restrict__ partial, int n_slabs, int n_tiles_1d, int n_tiles, double* __restrict__ G,
                           int64_t stride_G, int64_t n, const int32_t* __restrict__ perm) {
  const int e = blockIdx.y * 256 + threadIdx.x;
  if (e >= TS * TS) return;
  const double* src = partial + ((int64_t)blockIdx.z * n_slabs * n_tiles + blockIdx.x) * (TS * TS) + e;
  double s = 0.0;
  for (int sl = 0; sl < n_slabs; ++sl) s += src[(int64_t)sl * n_tiles * (TS * TS)];
  int tile = blockIdx.x, ti = 0;
  while (tile >= n_tiles_1d - ti) {
    tile -= n_tiles_1d - ti;
    ++ti;
  }
  const int tj = ti + tile;
  const int64_t r = (int64_t)ti * TS + e / TS, c = (int64_t)tj * TS + e % TS;
  if (r < n && c < n && (ti != tj || c >= r)) {
    double* Gb = G + (int64_t)blockIdx.z * stride_G;
    const int64_t pr = perm ? perm[r] : r, pc = perm ? perm[c] : c;
    Gb[pr * n + pc] = s;
    Gb[pc * n + pr] = s;
  }
}

// ----------------------------------------------------------------------------------
// Gram for n >= 128, the dominant kernel of the bond-capped sweep (fp64 MFMA bound: 2 m n (n + 1) / 2 flops).
// Same tiling as gram_wide_kernel<128> (128 x 128 tiles of the upper triangle x row slabs, 32-row chunks of
// the two column panels in LDS, four waves of 64 x 64), rebuilt around what kept that kernel at 67 % of the
// 78 TFLOP/s this GPU sustains on v_mfma_f64_16x16x4_f64 (tools/scratch/mfma_f64_rate.hip):
//   * the chunks are double-buffered in LDS: the global loads of chunk c + 1 fly under the MFMAs of chunk c
//     and are written to the other buffer afterwards -- ONE barrier per chunk, not two around a stall;
//   * the operands of k-step s + 1 are read from LDS and converted before the MFMAs of step s are issued;
//   * a DIAGONAL tile computes only the 16 x 16 tiles of its upper triangle (36 of 64): wave 0 / wave 3 the
//     ten of a diagonal 64 x 64 block, waves 1 and 2 half of the off-diagonal block each -- 10 MFMAs per step
//     on the critical wave instead of 16; diagonal tiles get 1.6 x longer slabs so all workgroups last alike;
//   * one launch serves a whole batch of matrices (blockIdx.y): the small Grams of later sites fill the GPU
//     and the slabs can be long (fewer partial tiles to write and reduce).
// Partial tile of workgroup id of matrix b: partial[(b slots + id) 128^2 ..], off-diagonal tiles first
// (tile-major, S_off slabs each), then the diagonal ones (S_diag slabs each); gram128_reduce_kernel sums a
// tile's slabs in order (deterministic) and writes it mirrored.
// ----------------------------------------------------------------------------------
constexpr int G128_LD = 128 + 16;                 // padded LDS row (floats): fragment reads are conflict-free
constexpr int G128_PANEL = GW_KB * G128_LD;       // floats per panel and buffer
constexpr size_t kGram128Lds = (size_t)4 * G128_PANEL * sizeof(float);  // 2 buffers x 2 panels = 73.7 KB
constexpr int kGram128MaxBatch = 48;              // matrices per launch (pointers travel as kernel arguments)

struct Gram128Geom {
  int tiles_1d, n_off, n_diag;
  int slabs_off, slabs_diag;
  int64_t rows_off, rows_diag;
  int slots;  // workgroups = partial tiles per matrix
  // XCD-aware order (xcd != 0, 1-D grid): a GROUP = the tiles_1d^2 workgroups that read the same two off-diagonal slabs
  // (= one diagonal slab, rows_diag = 2 rows_off) of one matrix, i.e. the same rows of every 128-column panel.
  // Workgroups are dealt round-robin over the 8 XCDs, so the group's members take hardware ids that are 8 apart and
  // consecutive on their XCD: they start together on one XCD and stream the same rows through ONE L2 (each panel
  // was fetched by the four tiles sharing it from four different XCDs otherwise: 4 x the fabric traffic).
  int xcd, members, groups_per_matrix, groups_total;
};
struct Gram128Ptrs {
  const void* a[kGram128MaxBatch];
};

template <int ROLE>  // 0: 4 x 4 tiles; 1: upper triangle of a diagonal block (10 tiles); 2: 2 x 4 tiles
__device__ __forceinline__ void gram128_load(const float* pa, const float* pb, int s, double (&av)[4], double (&bv)[4]) {
  const int off = 4 * s * G128_LD;
  if (ROLE == 1) {
#pragma unroll
    for (int a = 0; a < 4; ++a) av[a] = bv[a] = (double)pa[off + 16 * a];
  } else {
#pragma unroll
    for (int a = 0; a < (ROLE == 2 ? 2 : 4); ++a) av[a] = (double)pa[off + 16 * a];
#pragma unroll
    for (int b = 0; b < 4; ++b) bv[b] = (double)pb[off + 16 * b];
  }
}
template <int ROLE>
__device__ __forceinline__ void gram128_mfma(const double (&av)[4], const double (&bv)[4], f64x4 (&acc)[16]) {
#pragma unroll
  for (int a = 0; a < (ROLE == 2 ? 2 : 4); ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (ROLE != 1 || a <= b) acc[4 * a + b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[4 * a + b], 0, 0, 0);
}
// the eight k-steps of one chunk; pa / pb: this lane's element of row 0 of the two panels (LDS)
template <int ROLE>
__device__ __forceinline__ void gram128_chunk(const float* pa, const float* pb, f64x4 (&acc)[16]) {
  double av[2][4], bv[2][4];
  gram128_load<ROLE>(pa, pb, 0, av[0], bv[0]);
#pragma unroll
  for (int s = 0; s < GW_KB / 4; ++s) {
    if (s + 1 < GW_KB / 4) gram128_load<ROLE>(pa, pb, s + 1, av[(s + 1) & 1], bv[(s + 1) & 1]);
    gram128_mfma<ROLE>(av[s & 1], bv[s & 1], acc);
  }
}

// MODE 0: any shape (guarded, per-lane branches in the fetch).  MODE 1 (plain rows) / 2 (rows and columns through
// the offset tables): every chunk is interior -- whole 128-column panels, slabs of whole 32-row chunks, 16-byte
// loads -- and the fetch is eight straight-line loads.  In MODE 0 every load sits in an exec-masked block and carries
// its own s_waitcnt vmcnt(0) (the registers it overwrites may still be the target of a load of the previous trip,
// and a wait inside a skipped block clears nothing at the join): the eight loads of a chunk go out one after the
// other, each waiting for the one before, in front of the chunk's MFMAs.
template <typename TIN, int MODE>
__global__ void __launch_bounds__(256, 2)
gram128_kernel(Gram128Ptrs ptrs, int64_t m, int64_t n, int64_t lda, double* __restrict__ partial, Gram128Geom g,
               int vec_ok, const int64_t* __restrict__ row_off, const int64_t* __restrict__ col_off) {
  extern __shared__ __attribute__((aligned(16))) float g128_lds[];  // [2 buffers][2 panels][GW_KB][G128_LD]
  int id = blockIdx.x, vol = blockIdx.y;
  if (g.xcd) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int group = (q / g.members) * 8 + x, mem = q % g.members;
    if (group >= g.groups_total) return;
    vol = group / g.groups_per_matrix;
    const int sp = group % g.groups_per_matrix;
    if (g.xcd == 2) {  // one slab per group, diagonal slabs as long as the others
      id = mem < g.n_off ? mem * g.slabs_off + sp : g.n_off * g.slabs_off + (mem - g.n_off) * g.slabs_diag + sp;
    } else if (mem < 2 * g.n_off) {
      const int sl = 2 * sp + mem / g.n_off;
      if (sl >= g.slabs_off) return;
      id = (mem % g.n_off) * g.slabs_off + sl;
    } else {
      id = g.n_off * g.slabs_off + (mem - 2 * g.n_off) * g.slabs_diag + sp;
    }
  }
  const TIN* __restrict__ A = static_cast<const TIN*>(ptrs.a[vol]);
  int ti, tj, slab;
  int64_t rows;
  if (id < g.n_off * g.slabs_off) {
    int t = id / g.slabs_off;
    slab = id % g.slabs_off;
    rows = g.rows_off;
    ti = 0;
    while (t >= g.tiles_1d - 1 - ti) {
      t -= g.tiles_1d - 1 - ti;
      ++ti;
    }
    tj = ti + 1 + t;
  } else {
    const int t = (id - g.n_off * g.slabs_off) / g.slabs_diag;
    slab = (id - g.n_off * g.slabs_off) % g.slabs_diag;
    rows = g.rows_diag;
    ti = tj = t;
  }
  const bool diag = ti == tj;
  const int64_t i0 = (int64_t)ti * 128, j0 = (int64_t)tj * 128;
  const int64_t r_begin = (int64_t)slab * rows;
  const int64_t r_end = min(m, r_begin + rows);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane >> 4, lc = lane & 15;

  // staging map: GW_KB rows x 32 float4 per panel, 4 per thread
  float4 pi[4], pj[4];
  // gathered reads: the two column offsets of this thread never change (its column quad is fixed) and the row
  // offsets of a chunk are requested one chunk ahead -- looked up inside fetch() they put a table round trip in
  // front of the data loads, and the wave sat through it before it could start the chunk's MFMAs (+18 %)
  int64_t coli = 0, colj = 0, ro[4] = {0, 0, 0, 0};
  auto row_offsets = [&](int64_t r0) {
    if (MODE == 2 || (MODE == 0 && row_off)) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int64_t row = r0 + (tid + 256 * v) / 32;
        ro[v] = row < r_end ? row_off[row] : 0;
      }
    }
  };
  if (MODE == 2 || (MODE == 0 && row_off)) {
    const int c4 = (tid % 32) * 4;
    if (i0 + c4 + 3 < n) coli = col_off[i0 + c4];
    if (j0 + c4 + 3 < n) colj = col_off[j0 + c4];
  }
  auto fetch = [&](int64_t r0) {
    if constexpr (MODE != 0) {
      // interior chunk: straight-line loads, the raw elements converted on their way to LDS (stash)
      const int c4 = (tid % 32) * 4;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const TIN* src = MODE == 2 ? A + ro[v] + coli : A + (r0 + (tid + 256 * v) / 32) * lda + i0 + c4;
        pi[v] = load4_as_f32(src);
      }
      if (!diag) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const TIN* src = MODE == 2 ? A + ro[v] + colj : A + (r0 + (tid + 256 * v) / 32) * lda + j0 + c4;
          pj[v] = load4_as_f32(src);
        }
      }
      return;
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = tid + 256 * v;
      const int rr = e / 32, c4 = (e % 32) * 4;
      const int64_t row = r0 + rr;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f), y = x;
      if (row < r_end && row_off) {  // gathered: offsets additive in (row, column); vec_ok guaranteed by the host
        const TIN* base = A + ro[v];
        if (i0 + c4 + 3 < n) x = load4_as_f32(base + coli);
        if (!diag && j0 + c4 + 3 < n) y = load4_as_f32(base + colj);
      } else if (row < r_end) {
        const TIN* base = A + row * lda;
        if (vec_ok && i0 + c4 + 3 < n) x = load4_as_f32(base + i0 + c4);
        else {
          if (i0 + c4 + 0 < n) x.x = (float)base[i0 + c4 + 0];
          if (i0 + c4 + 1 < n) x.y = (float)base[i0 + c4 + 1];
          if (i0 + c4 + 2 < n) x.z = (float)base[i0 + c4 + 2];
          if (i0 + c4 + 3 < n) x.w = (float)base[i0 + c4 + 3];
        }
        if (!diag) {
          if (vec_ok && j0 + c4 + 3 < n) y = load4_as_f32(base + j0 + c4);
          else {
            if (j0 + c4 + 0 < n) y.x = (float)base[j0 + c4 + 0];
            if (j0 + c4 + 1 < n) y.y = (float)base[j0 + c4 + 1];
            if (j0 + c4 + 2 < n) y.z = (float)base[j0 + c4 + 2];
            if (j0 + c4 + 3 < n) y.w = (float)base[j0 + c4 + 3];
          }
        }
      }
      pi[v] = x;
      pj[v] = y;
    }
  };
  auto stash = [&](int buf) {
    float* Pi = g128_lds + (2 * buf) * G128_PANEL;
    float* Pj = Pi + G128_PANEL;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int e = tid + 256 * v;
      const int rr = e / 32, c4 = (e % 32) * 4;
      *reinterpret_cast<float4*>(Pi + rr * G128_LD + c4) = pi[v];
      if (!diag) *reinterpret_cast<float4*>(Pj + rr * G128_LD + c4) = pj[v];
    }
  };

  // operand columns of this wave inside the two panels
  //   off-diagonal tile : wave (wr, wc) -> rows wr 64 of panel i, columns wc 64 of panel j
  //   diagonal tile     : wave 0 -> block (0, 0), wave 3 -> block (1, 1) (upper triangles), waves 1 / 2 -> rows
  //                       0..31 / 32..63 of block (0, 1); both operands from panel i
  int a_col, b_col;
  if (!diag) {
    a_col = (wave >> 1) * 64;
    b_col = (wave & 1) * 64;
  } else if (wave == 0 || wave == 3) {
    a_col = b_col = wave == 0 ? 0 : 64;
  } else {
    a_col = wave == 1 ? 0 : 32;
    b_col = 64;
  }

  double* out = partial + ((int64_t)vol * g.slots + id) * (128 * 128);
  // one copy of the chunk loop per role (the role is fixed for the life of the wave; a branch per chunk made
  // the register allocator keep the accumulators three times)
  auto run = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    f64x4 acc[16];
#pragma unroll
    for (int a = 0; a < 16; ++a) acc[a] = (f64x4){0.0, 0.0, 0.0, 0.0};
    if (r_begin < r_end) {
      row_offsets(r_begin);
      fetch(r_begin);
      stash(0);
      row_offsets(r_begin + GW_KB);
    }
    __syncthreads();
    int buf = 0;
    for (int64_t r0 = r_begin; r0 < r_end; r0 += GW_KB, buf ^= 1) {
      const bool more = r0 + GW_KB < r_end;
      if (more) {
        fetch(r0 + GW_KB);          // in flight under the MFMAs
        row_offsets(r0 + 2 * GW_KB);  // for the fetch of the next iteration
      }
      const float* Pi = g128_lds + (2 * buf) * G128_PANEL + lr * G128_LD + lc;
      const float* Pj = ROLE == 0 ? Pi + G128_PANEL : Pi;
      gram128_chunk<ROLE>(Pi + a_col, Pj + b_col, acc);
      if (more) stash(buf ^ 1);
      __syncthreads();  // the other buffer is complete and nobody reads this one any more
    }
#pragma unroll
    for (int a = 0; a < (ROLE == 2 ? 2 : 4); ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (ROLE == 1 && a > b) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int rr = a_col + 16 * a + lr + 4 * reg, cc = b_col + 16 * b + lc;
          out[rr * 128 + cc] = acc[4 * a + b][reg];
        }
      }
  };
  if (!diag) run(std::integral_constant<int, 0>{});
  else if (wave == 0 || wave == 3) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 2>{});
}

// G of matrix blockIdx.z from its partial tiles: tile t = blockIdx.x (off-diagonal ones first), element
// e = blockIdx.y 256 + threadIdx.x; slabs summed in order, four in flight.
__global__ void __launch_bounds__(256)
gram128_reduce_kernel(const double* __restrict__ partial, Gram128Geom g, double* __restrict__ G, int64_t stride_G, int64_t n,
                      const int32_t* __restrict__ perm) {  // perm (or NULL): element (r, c) goes to (perm[r], perm[c])
  const int e = blockIdx.y * 256 + threadIdx.x;
  const int t = blockIdx.x;
  int ti, tj, base, count;
  if (t < g.n_off) {
    int u = t;
    ti = 0;
    while (u >= g.tiles_1d - 1 - ti) {
      u -= g.tiles_1d - 1 - ti;
      ++ti;
    }
    tj = ti + 1 + u;
    base = t * g.slabs_off;
    count = g.slabs_off;
  } else {
    ti = tj = t - g.n_off;
    base = g.n_off * g.slabs_off + (t - g.n_off) * g.slabs_diag;
    count = g.slabs_diag;
  }
  const int64_t r = (int64_t)ti * 128 + e / 128, c = (int64_t)tj * 128 + e % 128;
  if (r >= n || c >= n || (ti == tj && c < r)) return;  // diagonal tiles: upper part only (the rest was not computed)
  const double* src = partial + ((int64_t)blockIdx.z * g.slots + base) * (128 * 128) + e;
  double s = 0.0;
  int sl = 0;
  for (; sl + 4 <= count; sl += 4) {
    const double v0 = src[(int64_t)(sl + 0) * 16384], v1 = src[(int64_t)(sl + 1) * 16384];
    const double v2 = src[(int64_t)(sl + 2) * 16384], v3 = src[(int64_t)(sl + 3) * 16384];
    s = (((s + v0) + v1) + v2) + v3;
  }
  for (; sl < count; ++sl) s += src[(int64_t)sl * 16384];
  double* Gb = G + (int64_t)blockIdx.z * stride_G;
  const int64_t pr = perm ? perm[r] : r, pc = perm ? perm[c] : c;
  Gb[pr * n + pc] = s;
  Gb[pc * n + pr] = s;
}

// ----------------------------------------------------------------------------------
// Gram for very narrow matrices (n <= 8, the first site of the sweep: m = N / d rows of d
// voxels): pure streaming.  One thread per row (grid-stride), the 36 products of a row go to
// fp64 registers; wave shuffle + LDS fold; one partial per workgroup, summed in fixed order.
// ----------------------------------------------------------------------------------
template <typename TIN>
__global__ void __launch_bounds__(256)
gram_small_kernel(const TIN* __restrict__ A, int64_t m, int n, int64_t lda, double* __restrict__ partial,
                  int vec_ok) {
  __shared__ double red[4][36];
  double acc[36];
#pragma unroll
  for (int i = 0; i < 36; ++i) acc[i] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < m; r += stride) {
    float x[8];
    const TIN* row = A + r * lda;
    if (vec_ok) {
      const float4 a = load4_as_f32(row), b = load4_as_f32(row + 4);
      x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) x[c] = c < n ? (float)row[c] : 0.f;
    }
    int idx = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = i; j < 8; ++j) acc[idx++] += (double)x[i] * (double)x[j];
  }
#pragma unroll
  for (int i = 0; i < 36; ++i) {
    double v = acc[i];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 36)
    partial[(int64_t)blockIdx.x * 36 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ void __launch_bounds__(256)
gram_small_reduce_kernel(const double* __restrict__ partial, int n_blocks, double* __restrict__ G, int n) {
  __shared__ double part[7][36];
  const int e = threadIdx.x % 36, grp = threadIdx.x / 36;  // 7 groups of 36 threads (252 used)
  if (grp < 7) {
    double acc = 0.0;
    for (int b = grp; b < n_blocks; b += 7) acc += partial[(int64_t)b * 36 + e];
    part[grp][e] = acc;
  }
  __syncthreads();
  if (threadIdx.x >= 36) return;
  double s = 0.0;
  for (int k = 0; k < 7; ++k) s += part[k][e];
  int i = 0, rem = e;  // unrank e -> (i, j), j >= i
  while (rem >= 8 - i) {
    rem -= 8 - i;
    ++i;
  }
  const int j = i + rem;
  if (i < n && j < n) {
    G[i * n + j] = s;
    G[j * n + i] = s;
  }
}

constexpr int kGramSmallBlocks = 256;

struct GramGeom {
  int T;          // 16-wide sub-tiles per tile edge
  int tiles_1d;
  int n_tiles;    // upper triangle incl. diagonal
  int n_slabs;
  int64_t rows_per_slab;
};

GramGeom gram_geometry(int64_t m, int64_t n) {
  GramGeom g;
  g.T = n <= 16 ? 1 : (n <= 32 ? 2 : 4);
  const int ts = 16 * g.T;
  g.tiles_1d = (int)ndmps::ceil_div(n, ts);
  g.n_tiles = g.tiles_1d * (g.tiles_1d + 1) / 2;
  // aim at ~4 workgroups per CU; every slab is a multiple of 16 rows (4 waves x 4 rows)
  int64_t want = std::max<int64_t>(1, (2 * ndmps::kNumCU) / g.n_tiles);
  int64_t rows = ndmps::round_up(std::max<int64_t>(ndmps::ceil_div(m, want), 64), 16);
  g.rows_per_slab = rows;
  g.n_slabs = (int)std::max<int64_t>(1, ndmps::ceil_div(m, rows));
  return g;
}

}  // namespace

extern "C" int ndmps_sgemm(int transA, int transB, int64_t m, int64_t n, int64_t k, const float* d_A,
                           int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                           ndmps_stream_t stream) {
  NDMPS_TRY(gemm_check(m, n, k, d_A, lda, transA, d_B, ldb, transB, d_C, ldc));
  if (m == 0 || n == 0) return NDMPS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (n <= 32) return launch_gemm<float, 128, 32, 4, 1>(transA, transB, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, s);
  if (n <= 64 || m <= 64)
    return launch_gemm<float, 64, 64, 2, 2>(transA, transB, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, s);
  return launch_gemm<float, 128, 128, 2, 2>(transA, transB, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, s);
}

// C = A B (no transposes) with table-driven addressing of A and / or C (see GemmIndex): the reshape stage fused
// into a product of the sweep (A = the volume read through the permutation) or of the chain (C = the
// reconstructed volume written through the inverse permutation).  a_vec4: every aligned group of four
// consecutive k has consecutive offsets in d_a_col (16-byte loads allowed).
extern "C" int ndmps_sgemm_indexed(int64_t m, int64_t n, int64_t k, const float* d_A, int64_t lda,
                                   const int64_t* d_a_row, const int64_t* d_a_col, int a_vec4, const float* d_B,
                                   int64_t ldb, float* d_C, int64_t ldc, const int64_t* d_c_row,
                                   const int64_t* d_c_col, ndmps_stream_t stream) {
  NDMPS_REQUIRE(m >= 0 && n >= 0 && k >= 0 && d_A && d_B && d_C, "bad indexed GEMM argument");
  NDMPS_REQUIRE((d_a_row == nullptr) == (d_a_col == nullptr) && (d_c_row == nullptr) == (d_c_col == nullptr),
                "offset tables come in (row, column) pairs");
  NDMPS_REQUIRE(ldb >= n && (d_a_row || lda >= k) && (d_c_row || ldc >= n), "leading dimension too small");
  if (m == 0 || n == 0) return NDMPS_OK;
  hipStream_t s = (hipStream_t)stream;
  GemmIndex ix{d_a_row, d_a_col, d_c_row, d_c_col, gemm_guarded_env()};
  const uintptr_t al = 16;
  const bool vec = ldb % 4 == 0 && k % 4 == 0 && n % 4 == 0 && (uintptr_t)d_A % al == 0 && (uintptr_t)d_B % al == 0 &&
                   (d_a_row ? a_vec4 != 0 : lda % 4 == 0);
  const dim3 block(256);
  if (n <= 64 || m <= 64) {
    const dim3 grid((unsigned)ndmps::ceil_div(m, 64), (unsigned)ndmps::ceil_div(n, 64));
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<float, 64, 64, 2, 2, false, false, true, true>), grid, block, 0, s, m, n, k, d_A,
                         lda, d_B, ldb, d_C, ldc, ix);
    else
      hipLaunchKernelGGL((gemm_kernel<float, 64, 64, 2, 2, false, false, false, true>), grid, block, 0, s, m, n, k, d_A,
                         lda, d_B, ldb, d_C, ldc, ix);
  } else {
    const dim3 grid((unsigned)ndmps::ceil_div(m, 128), (unsigned)ndmps::ceil_div(n, 128));
    if (vec)
      hipLaunchKernelGGL((gemm_kernel<float, 128, 128, 2, 2, false, false, true, true>), grid, block, 0, s, m, n, k,
                         d_A, lda, d_B, ldb, d_C, ldc, ix);
    else
      hipLaunchKernelGGL((gemm_kernel<float, 128, 128, 2, 2, false, false, false, true>), grid, block, 0, s, m, n, k,
                         d_A, lda, d_B, ldb, d_C, ldc, ix);
  }
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_dgemm(int transA, int transB, int64_t m, int64_t n, int64_t k, const double* d_A,
                           int64_t lda, const double* d_B, int64_t ldb, double* d_C, int64_t ldc,
                           ndmps_stream_t stream) {
  NDMPS_TRY(gemm_check(m, n, k, d_A, lda, transA, d_B, ldb, transB, d_C, ldc));
  if (m == 0 || n == 0) return NDMPS_OK;
  hipStream_t s = (hipStream_t)stream;
  if (n <= 16) return launch_gemm<double, 64, 16, 4, 1>(transA, transB, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, s);
  if (n <= 32 || m <= 32)
    return launch_gemm<double, 32, 32, 2, 2>(transA, transB, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, s);
  return launch_gemm<double, 64, 64, 2, 2>(transA, transB, m, n, k, d_A, lda, d_B, ldb, d_C, ldc, s);
}

namespace {
template <typename T>
int gemm_batched_check(int batch, const T* const* h_A, const T* const* h_B, T* const* h_C, GemmBatchPtrs& bp) {
  NDMPS_REQUIRE(batch >= 1 && batch <= kGemmMaxBatch && h_A && h_B && h_C, "batched GEMM: batch=%d outside [1, %d] or NULL array",
                batch, kGemmMaxBatch);
  for (int z = 0; z < batch; ++z) {
    NDMPS_REQUIRE(h_A[z] && h_B[z] && h_C[z], "batched GEMM: NULL operand %d", z);
    bp.a[z] = h_A[z];
    bp.b[z] = h_B[z];
    bp.c[z] = h_C[z];
  }
  return NDMPS_OK;
}
}  // namespace

// `batch` (<= ndmps_gemm_batched_max()) products of one shape in one launch; h_A / h_B / h_C: HOST arrays of device
// pointers.  Same tiles and arithmetic as ndmps_sgemm / ndmps_dgemm on each triple.
extern "C" int ndmps_gemm_batched_max(void) { return kGemmMaxBatch; }

extern "C" int ndmps_sgemm_batched(int batch, int transA, int transB, int64_t m, int64_t n, int64_t k,
                                   const float* const* h_A, int64_t lda, const float* const* h_B, int64_t ldb,
                                   float* const* h_C, int64_t ldc, ndmps_stream_t stream) {
  GemmBatchPtrs bp;
  NDMPS_TRY(gemm_batched_check(batch, h_A, h_B, h_C, bp));
  NDMPS_TRY(gemm_check(m, n, k, h_A[0], lda, transA, h_B[0], ldb, transB, h_C[0], ldc));
  if (m == 0 || n == 0) return NDMPS_OK;
  hipStream_t s = (hipStream_t)stream;
  const float* A = nullptr;
  float* Cn = nullptr;
  if (n <= 32) return launch_gemm<float, 128, 32, 4, 1>(transA, transB, m, n, k, A, lda, A, ldb, Cn, ldc, s, &bp, batch);
  if (n <= 64 || m <= 64)
    return launch_gemm<float, 64, 64, 2, 2>(transA, transB, m, n, k, A, lda, A, ldb, Cn, ldc, s, &bp, batch);
  return launch_gemm<float, 128, 128, 2, 2>(transA, transB, m, n, k, A, lda, A, ldb, Cn, ldc, s, &bp, batch);
}

extern "C" int ndmps_dgemm_batched(int batch, int transA, int transB, int64_t m, int64_t n, int64_t k,
                                   const double* const* h_A, int64_t lda, const double* const* h_B, int64_t ldb,
                                   double* const* h_C, int64_t ldc, ndmps_stream_t stream) {
  GemmBatchPtrs bp;
  NDMPS_TRY(gemm_batched_check(batch, h_A, h_B, h_C, bp));
  NDMPS_TRY(gemm_check(m, n, k, h_A[0], lda, transA, h_B[0], ldb, transB, h_C[0], ldc));
  if (m == 0 || n == 0) return NDMPS_OK;
  hipStream_t s = (hipStream_t)stream;
  const double* A = nullptr;
  double* Cn = nullptr;
  if (n <= 16) return launch_gemm<double, 64, 16, 4, 1>(transA, transB, m, n, k, A, lda, A, ldb, Cn, ldc, s, &bp, batch);
  if (n <= 32 || m <= 32)
    return launch_gemm<double, 32, 32, 2, 2>(transA, transB, m, n, k, A, lda, A, ldb, Cn, ldc, s, &bp, batch);
  return launch_gemm<double, 64, 64, 2, 2>(transA, transB, m, n, k, A, lda, A, ldb, Cn, ldc, s, &bp, batch);
}

// ----------------------------------------------------------------------------------
// C = A W for A = a lockstep group's volumes read through the permutation tables as (m x 64) matrices -- the first
// projection of a bond cap of 32 (BASELINE configs 2 and 4): a 64 MB stream per 256^3 volume, 96 MB with the result.
// The tile kernel reads it at 2.8 TB/s (1.1 ms per group of 32).  Here, as in gram64_stream_kernel, a lane's 16-byte
// load is four MFMA operands: lane (i, h) = (lane % 32, lane / 32) of v_mfma_f32_32x32x2_f32 takes columns
// 8q + 4h .. 8q + 4h + 3 of row i of a 32-row tile (columns in memory order: aligned quads are contiguous), its
// component c is the A operand of the k-pair {8q + c, 8q + 4 + c}; the matching W rows sit in LDS as 16-byte records
// [q][h][j] = (W[8q + 4h + c][j])_c.  Rows are visited in ascending order of their offsets (d_row_sorted): the 32
// lanes of a half-wave read 512 contiguous bytes; row s of that order is row d_row_order[s] of C.  No barrier in the
// loop; the eight loads of the next tile are issued slot by slot as the current tile's steps consume theirs.
// NB = n / 32 (n = 32 or 64 columns of W).
namespace {
template <int NB>
__global__ void __launch_bounds__(256, 2)
proj64_stream_kernel(GemmBatchPtrs ptrs, int64_t m, int64_t ldb, int64_t ldc, const int64_t* __restrict__ row_sorted,
                     const int32_t* __restrict__ row_order, const int64_t* __restrict__ col_off, int64_t tiles_per_wg) {
  constexpr int N = 32 * NB, Q = 8;
  const float* A = static_cast<const float*>(ptrs.a[blockIdx.z]);
  const float* W = static_cast<const float*>(ptrs.b[blockIdx.z]);
  float* C = static_cast<float*>(ptrs.c[blockIdx.z]);
  __shared__ __attribute__((aligned(16))) float Wp[Q * 2 * N * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 64 * N; e += 256) {
    const int kidx = e / N, j = e % N;
    const int q = kidx >> 3, h = (kidx >> 2) & 1, c = kidx & 3;
    Wp[((q * 2 + h) * N + j) * 4 + c] = W[(int64_t)kidx * ldb + j];
  }
  __syncthreads();
  const int li = lane & 31, h = lane >> 5;
  int64_t coff[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) coff[q] = col_off[8 * q + 4 * h];
  const int64_t n_tiles = (m + 31) / 32;
  const int64_t t_begin = (int64_t)blockIdx.y * tiles_per_wg, t_end = min(n_tiles, t_begin + tiles_per_wg);
  // this wave's tiles: t_begin + wave, + 4, ...; a tile index beyond the end repeats the last row (never stored)
  auto row_offset = [&](int64_t t) { return row_sorted[min(32 * t + li, m - 1)]; };
  int64_t t = t_begin + wave;
  if (t >= t_end) return;
  float4 ring[Q];
  int64_t roff = row_offset(t), roff_next = row_offset(t + 4);
#pragma unroll
  for (int q = 0; q < Q; ++q) ring[q] = load4_stream_f32(A + roff + coff[q]);
  for (; t < t_end; t += 4) {
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;
    const int64_t roff_after = row_offset(t + 8);
    // the tile's sixteen output rows of this lane, requested before the next tile's loads (a load issued behind them,
    // inside the guarded stores, would make every store wait for the whole ring)
    int crow_of[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) crow_of[r] = row_order[min(32 * t + (r & 3) + 8 * (r >> 2) + 4 * h, m - 1)];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      // the step's operands leave the slot BEFORE it is reloaded in place (opaque copies: used from the slot itself, the
      // last MFMAs of the step would read registers an already issued load may overwrite, and the compiler answers
      // with a second set of registers and copies that wait for the loads they follow)
      float4 a = ring[q];
      asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
      float4 bq[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bq[nb] = *reinterpret_cast<const float4*>(&Wp[((q * 2 + h) * N + 32 * nb + li) * 4]);
      __builtin_amdgcn_sched_barrier(0);
      ring[q] = load4_stream_f32(A + roff_next + coff[q]);  // the same step of the wave's next tile
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[nb].x, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[nb].y, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[nb].z, acc[nb], 0, 0, 0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[nb].w, acc[nb], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    roff_next = roff_after;
    // accumulator register r of lane (j, h) is row (r & 3) + 8 (r >> 2) + 4 h of the tile, column j
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t s = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (s < m) {
        float* crow = C + (int64_t)crow_of[r] * ldc + li;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) crow[32 * nb] = acc[nb][r];
      }
    }
  }
}

}  // namespace

// batched C = A B with table-driven addressing (one set of tables for the whole batch: the volumes of a lockstep
// group share the index permutation); see ndmps_sgemm_indexed
extern "C" int ndmps_sgemm_indexed_batched(int batch, int64_t m, int64_t n, int64_t k, const float* const* h_A,
                                           int64_t lda, const int64_t* d_a_row, const int64_t* d_a_col, int a_vec4,
                                           const float* const* h_B, int64_t ldb, float* const* h_C, int64_t ldc,
                                           const int64_t* d_c_row, const int64_t* d_c_col, ndmps_stream_t stream) {
  GemmBatchPtrs bp;
  NDMPS_TRY(gemm_batched_check(batch, h_A, h_B, h_C, bp));
  NDMPS_REQUIRE(m >= 0 && n >= 0 && k >= 0, "bad indexed GEMM extent");
  NDMPS_REQUIRE((d_a_row == nullptr) == (d_a_col == nullptr) && (d_c_row == nullptr) == (d_c_col == nullptr),
                "offset tables come in (row, column) pairs");
  NDMPS_REQUIRE(ldb >= n && (d_a_row || lda >= k) && (d_c_row || ldc >= n), "leading dimension too small");
  if (m == 0 || n == 0) return NDMPS_OK;
  hipStream_t s = (hipStream_t)stream;
  GemmIndex ix{d_a_row, d_a_col, d_c_row, d_c_col, gemm_guarded_env()};
  const uintptr_t al = 16;
  bool vec = ldb % 4 == 0 && k % 4 == 0 && n % 4 == 0 && (d_a_row ? a_vec4 != 0 : lda % 4 == 0);
  for (int z = 0; z < batch; ++z) vec = vec && (uintptr_t)bp.a[z] % al == 0 && (uintptr_t)bp.b[z] % al == 0;
  const dim3 block(256);
  if (n <= 64 || m <= 64) {
    const dim3 grid((unsigned)ndmps::ceil_div(m, 64), (unsigned)ndmps::ceil_div(n, 64), (unsigned)batch);
    if (vec)
      hipLaunchKernelGGL((gemm_batched_kernel<float, 64, 64, 2, 2, false, false, true, true>), grid, block, 0, s, m, n, k,
                         bp, lda, ldb, ldc, ix);
    else
      hipLaunchKernelGGL((gemm_batched_kernel<float, 64, 64, 2, 2, false, false, false, true>), grid, block, 0, s, m, n,
                         k, bp, lda, ldb, ldc, ix);
  } else {
    const dim3 grid((unsigned)ndmps::ceil_div(m, 128), (unsigned)ndmps::ceil_div(n, 128), (unsigned)batch);
    if (vec)
      hipLaunchKernelGGL((gemm_batched_kernel<float, 128, 128, 2, 2, false, false, true, true>), grid, block, 0, s, m, n,
                         k, bp, lda, ldb, ldc, ix);
    else
      hipLaunchKernelGGL((gemm_batched_kernel<float, 128, 128, 2, 2, false, false, false, true>), grid, block, 0, s, m,
                         n, k, bp, lda, ldb, ldc, ix);
  }
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

// The first projection of the fused sweep for 64 gathered columns, as a stream (proj64_stream_kernel): C[b] (m x n,
// n = 32 or 64, row-major with leading dimension ldc) = A[b] W[b], element (r, c) of A[b] at
// h_A[b][d_row_off[r] + d_col_off[c]].  d_row_sorted: d_row_off in ascending order; d_row_order[s]: the row whose offset
// is d_row_sorted[s].  NDMPS_EINVAL for shapes the kernel does not take (the caller keeps ndmps_sgemm_indexed_batched).
extern "C" int ndmps_sgemm_gathered64_stream_batched(int batch, int64_t m, int64_t n, const float* const* h_A,
                                                     const int64_t* d_row_sorted, const int32_t* d_row_order,
                                                     const int64_t* d_col_off, const float* const* h_B, int64_t ldb,
                                                     float* const* h_C, int64_t ldc, ndmps_stream_t stream) {
  GemmBatchPtrs bp;
  NDMPS_TRY(gemm_batched_check(batch, h_A, h_B, h_C, bp));
  NDMPS_REQUIRE(d_row_sorted && d_row_order && d_col_off, "NULL table");
  NDMPS_REQUIRE(m >= 1 && (n == 32 || n == 64) && ldb >= n && ldc >= n, "the stream takes 32 or 64 result columns");
  for (int z = 0; z < batch; ++z) NDMPS_REQUIRE((uintptr_t)bp.a[z] % 16 == 0, "operand %d is not 16-byte aligned", z);
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = ndmps::ceil_div(m, 32);
  // ~4 workgroups per CU over the batch (two resident), whole multiples of the four tiles a workgroup's waves take
  const int64_t want = std::max<int64_t>(1, ndmps::ceil_div((int64_t)4 * ndmps::kNumCU, batch));
  const int64_t tiles_per_wg = ndmps::round_up(std::max<int64_t>(ndmps::ceil_div(n_tiles, want), 4), 4);
  const dim3 grid(1, (unsigned)ndmps::ceil_div(n_tiles, tiles_per_wg), (unsigned)batch);
  if (n == 32)
    hipLaunchKernelGGL(proj64_stream_kernel<1>, grid, dim3(256), 0, s, bp, m, ldb, ldc, d_row_sorted, d_row_order, d_col_off,
                       tiles_per_wg);
  else
    hipLaunchKernelGGL(proj64_stream_kernel<2>, grid, dim3(256), 0, s, bp, m, ldb, ldc, d_row_sorted, d_row_order, d_col_off,
                       tiles_per_wg);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

// geometry of the 128-wide path: ~2 workgroups per CU, slabs a multiple of the 32-row chunk
inline int gram_wide_tile(int64_t n) { return n >= 128 ? 128 : 64; }

GramGeom gram_wide_geometry(int64_t m, int64_t n) {
  GramGeom g;
  const int GW_TS = gram_wide_tile(n);
  g.T = GW_TS / 16;
  g.tiles_1d = (int)ndmps::ceil_div(n, GW_TS);
  g.n_tiles = g.tiles_1d * (g.tiles_1d + 1) / 2;
  const int64_t want = std::max<int64_t>(1, (2 * ndmps::kNumCU) / g.n_tiles);
  g.rows_per_slab = ndmps::round_up(std::max<int64_t>(ndmps::ceil_div(m, want), 4 * GW_KB), GW_KB);
  g.n_slabs = (int)std::max<int64_t>(1, ndmps::ceil_div(m, g.rows_per_slab));
  return g;
}

inline bool gram_use_wide(int64_t m, int64_t n) { return n >= 64 && m >= 256; }
inline bool gram_use_small(int64_t n) { return n <= 8; }

// the 128-tile kernel: n >= 128 and at least 8 chunks of rows
inline bool gram_use_128(int64_t m, int64_t n) { return n >= 128 && m >= 256; }

// Slabs of gram128_kernel.  An off-diagonal workgroup costs 16 MFMAs per k-step, a diagonal one 10, so the
// diagonal tiles get 1.6 x longer slabs.  One matrix alone fills the GPU once (~2 workgroups per CU); a batch is
// cut into ~12 rounds of 512 workgroups (the tail of the last round is what is lost), never below 512 rows per
// workgroup (128 for a lone matrix).  The geometry depends on (m, n, batch) only: a given call sequence is reproducible bit for bit.
Gram128Geom gram128_geometry(int64_t m, int64_t n, int batch) {
  Gram128Geom g;
  g.tiles_1d = (int)ndmps::ceil_div(n, 128);
  g.n_off = g.tiles_1d * (g.tiles_1d - 1) / 2;
  g.n_diag = g.tiles_1d;
  const double weight = g.n_off + 0.625 * g.n_diag;  // workgroups per off-diagonal slab count
  const double want = 2.0 * ndmps::kNumCU * (batch > 1 ? 12.0 : 1.0) / std::max(batch, 1) / weight;
  const int64_t floor_rows = batch > 1 ? 512 : 128;  // a lone small matrix still spreads over the GPU
  int64_t s_off = std::max<int64_t>(1, std::min<int64_t>((int64_t)(want + 0.5), std::max<int64_t>(m / floor_rows, 1)));
  g.rows_off = ndmps::round_up(ndmps::ceil_div(m, s_off), GW_KB);
  g.rows_diag = ndmps::round_up((g.rows_off * 8 + 4) / 5, GW_KB);
  g.slabs_off = (int)ndmps::ceil_div(m, g.rows_off);
  g.slabs_diag = (int)ndmps::ceil_div(m, g.rows_diag);
  g.xcd = 0;
  g.members = g.groups_per_matrix = g.groups_total = 0;
  // big batched launches (the ones that take the device-side turn): XCD-aware groups, diagonal slabs of exactly two
  // off-diagonal slabs (a diagonal workgroup then does 2 x 10 / 16 of an off-diagonal one's MFMA work)
  const char* xe = getenv("NDMPS_GRAM_XCD");
  const int xmode = xe ? atoi(xe) : 0;
  if (batch > 1 && g.tiles_1d >= 2 && g.slabs_off >= 4 && xmode == 1) {
    g.rows_diag = 2 * g.rows_off;
    g.slabs_diag = (int)ndmps::ceil_div(m, g.rows_diag);
    g.xcd = 1;
    g.members = 2 * g.n_off + g.n_diag;
    g.groups_per_matrix = g.slabs_diag;
    g.groups_total = batch * g.groups_per_matrix;
  } else if (batch > 1 && g.tiles_1d >= 2 && g.slabs_off >= 4 && xmode == 2) {
    // as many workgroups as before: slabs longer by (n_off + 0.625 n_diag) / (n_off + n_diag)
    const int64_t s2 = std::max<int64_t>(1, (int64_t)(g.slabs_off * weight / (g.n_off + g.n_diag) + 0.5));
    g.rows_off = g.rows_diag = ndmps::round_up(ndmps::ceil_div(m, s2), GW_KB);
    g.slabs_off = g.slabs_diag = (int)ndmps::ceil_div(m, g.rows_off);
    g.xcd = 2;
    g.members = g.n_off + g.n_diag;
    g.groups_per_matrix = g.slabs_off;
    g.groups_total = batch * g.groups_per_matrix;
  }
  g.slots = g.n_off * g.slabs_off + g.n_diag * g.slabs_diag;
  return g;
}

inline int64_t gram128_workspace(int64_t m, int64_t n, int batch) {
  return (int64_t)batch * gram128_geometry(m, n, batch).slots * 128 * 128 * (int64_t)sizeof(double) + 256;
}

int gram128_opt_in() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64 || done[dev]) return NDMPS_OK;
#define NDMPS_GRAM128_OPT_IN(T, MODE)                                                                          \
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gram128_kernel<T, MODE>),                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGram128Lds))
  NDMPS_GRAM128_OPT_IN(float, 0);
  NDMPS_GRAM128_OPT_IN(float, 1);
  NDMPS_GRAM128_OPT_IN(float, 2);
  NDMPS_GRAM128_OPT_IN(__bf16, 0);
  NDMPS_GRAM128_OPT_IN(__bf16, 1);
  NDMPS_GRAM128_OPT_IN(__bf16, 2);
#undef NDMPS_GRAM128_OPT_IN
  done[dev] = true;
  return NDMPS_OK;
}

// G[b] = A[b]^T A[b] for `batch` matrices of one shape (h_A: host array of device pointers)
template <typename TIN>
int gram128_batched(int batch, const TIN* const* h_A, int64_t m, int64_t n, int64_t lda, double* d_G, int64_t stride_G,
                    void* d_ws, int64_t ws_bytes, hipStream_t s, const int64_t* d_row_off, const int64_t* d_col_off,
                    const int32_t* d_perm = nullptr) {
  NDMPS_REQUIRE(batch >= 1 && h_A && d_G && gram_use_128(m, n) && lda >= n && stride_G >= n * n,
                "bad batched Gram argument (batch=%d m=%lld n=%lld)", batch, (long long)m, (long long)n);
  const int64_t need = gram128_workspace(m, n, batch);
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("Gram workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  NDMPS_TRY(gram128_opt_in());
  const Gram128Geom g = gram128_geometry(m, n, batch);
  int vec_ok = (lda % 4 == 0 && n % 4 == 0) ? 1 : 0;
  for (int b = 0; b < batch; ++b) {
    NDMPS_REQUIRE(h_A[b], "NULL Gram operand %d", b);
    if ((uintptr_t)h_A[b] % (4 * sizeof(TIN)) != 0) vec_ok = 0;
  }
  if (d_row_off)
    NDMPS_REQUIRE(d_col_off && vec_ok, "gathered Gram needs n %% 4 == 0 and aligned bases");
  double* partial = (double*)d_ws;
  // a launch that fills the GPU for milliseconds (a lockstep group's raw Gram) takes its turn with those of other
  // streams: two of them at once gain nothing (both MFMA-bound) and keep each other's groups in phase
  const bool turn = (int64_t)batch * g.slots >= 4096 && !getenv("NDMPS_GRAM_NO_TURN");
  // NDMPS_ONE_TURN=1 (A/B): the Gram launches take the resident tridiagonalisations' lock, whole -- with several batches in
  // flight (core/batch.py lanes) a Gram launch and a resident launch of another batch otherwise run together and slow each
  // other (both fp64: one pipeline)
  static const bool one_turn = getenv("NDMPS_ONE_TURN") != nullptr;
  ndmps::Turn gram_turn(s, one_turn ? ndmps::kTurnTeam : ndmps::kTurnGram, one_turn ? 2u : 1u, one_turn ? 2u : 1u);
  if (turn) NDMPS_TRY(gram_turn.begin());
  void* span = ndmps::span_begin(s);
  for (int base = 0; base < batch; base += kGram128MaxBatch) {
    const int count = std::min(kGram128MaxBatch, batch - base);
    Gram128Ptrs ptrs;
    for (int t = 0; t < count; ++t) ptrs.a[t] = h_A[base + t];
    // every chunk interior (whole panels, whole 32-row chunks, 16-byte loads): the straight-line fetch
    const bool interior = vec_ok && n % 128 == 0 && m % GW_KB == 0 && g.rows_off % GW_KB == 0 && g.rows_diag % GW_KB == 0 &&
                          !getenv("NDMPS_GRAM_GENERAL");
    auto kernel = !interior ? gram128_kernel<TIN, 0> : (d_row_off ? gram128_kernel<TIN, 2> : gram128_kernel<TIN, 1>);
    if (g.xcd) {
      Gram128Geom gc = g;  // this launch's matrices
      gc.groups_total = count * g.groups_per_matrix;
      const unsigned wgs = (unsigned)(ndmps::ceil_div(gc.groups_total, 8) * g.members * 8);
      hipLaunchKernelGGL(kernel, dim3(wgs), dim3(256), kGram128Lds, s, ptrs, m, n, lda,
                         partial + (int64_t)base * g.slots * 16384, gc, vec_ok, d_row_off, d_col_off);
    } else {
      hipLaunchKernelGGL(kernel, dim3(g.slots, count), dim3(256), kGram128Lds, s, ptrs, m, n, lda,
                         partial + (int64_t)base * g.slots * 16384, g, vec_ok, d_row_off, d_col_off);
    }
  }
  // algorithmic work of the span: the upper triangle incl. the diagonal, 2 flops per product
  ndmps::span_end(span, s, turn ? ndmps::kSpanGram : ndmps::kSpanGramSmall, (batch + kGram128MaxBatch - 1) / kGram128MaxBatch,
                  (int64_t)batch * m * n * (n + 1));
  NDMPS_TRY(gram_turn.end());
  hipLaunchKernelGGL(gram128_reduce_kernel, dim3(g.n_off + g.n_diag, 64, batch), dim3(256), 0, s, partial, g, d_G, stride_G, n,
                     d_perm);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int64_t ndmps_gram_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  if (gram_use_small(n)) return (int64_t)kGramSmallBlocks * 36 * 8 + 256;
  if (gram_use_128(m, n)) return gram128_workspace(m, n, 1);
  if (gram_use_wide(m, n)) {
    GramGeom g = gram_wide_geometry(m, n);
    const int64_t ts = gram_wide_tile(n);
    return (int64_t)(g.n_slabs + g.n_slabs / kReduceGroup + 2) * g.n_tiles * ts * ts * (int64_t)sizeof(double) + 256;
  }
  GramGeom g = gram_geometry(m, n);
  const int ts = 16 * g.T;
  return (int64_t)(g.n_slabs + g.n_slabs / kReduceGroup + 2) * g.n_tiles * ts * ts * (int64_t)sizeof(double) + 256;
}

namespace {
template <typename TIN>
int gram_any(const TIN* d_A, int64_t m, int64_t n, int64_t lda, double* d_G, void* d_ws, int64_t ws_bytes,
             ndmps_stream_t stream, const int64_t* d_row_off = nullptr, const int64_t* d_col_off = nullptr,
             const int32_t* d_perm = nullptr) {
  NDMPS_REQUIRE(d_A && d_G, "NULL Gram operand");
  NDMPS_REQUIRE(m > 0 && n > 0 && lda >= n, "bad Gram extents m=%lld n=%lld lda=%lld", (long long)m,
                (long long)n, (long long)lda);
  if (ws_bytes < ndmps_gram_workspace_bytes(m, n) || d_ws == nullptr) {
    ndmps::set_error("Gram workspace too small: %lld < %lld", (long long)ws_bytes,
                     (long long)ndmps_gram_workspace_bytes(m, n));
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  double* partial = (double*)d_ws;
  // four elements per load: 16 bytes of fp32, 8 bytes of bf16
  const int vec_ok = (lda % 4 == 0 && n % 4 == 0 && ((uintptr_t)d_A % (4 * sizeof(TIN))) == 0) ? 1 : 0;
  NDMPS_REQUIRE(!d_perm || d_row_off, "a column permutation comes with the offset tables");
  if (d_row_off) {
    NDMPS_REQUIRE(d_col_off && gram_use_wide(m, n) && n % 4 == 0 && ((uintptr_t)d_A % (4 * sizeof(TIN))) == 0,
                  "gathered Gram needs the wide path (n >= 64, m >= 256), n %% 4 == 0 and an aligned base");
  }
  if (gram_use_small(n)) {
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(m, 256 * 8), 1), kGramSmallBlocks);
    hipLaunchKernelGGL(gram_small_kernel<TIN>, dim3(blocks), dim3(256), 0, s, d_A, m, (int)n, lda, partial,
                       (vec_ok && n == 8) ? 1 : 0);
    hipLaunchKernelGGL(gram_small_reduce_kernel, dim3(1), dim3(256), 0, s, partial, blocks, d_G, (int)n);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }
  if (gram_use_128(m, n))
    return gram128_batched<TIN>(1, &d_A, m, n, lda, d_G, n * n, d_ws, ws_bytes, s, d_row_off, d_col_off, d_perm);
  if (gram_use_wide(m, n)) {
    GramGeom gw = gram_wide_geometry(m, n);
    NDMPS_REQUIRE(gw.n_slabs < 65536, "Gram slab count %d exceeds grid.y", gw.n_slabs);
    hipLaunchKernelGGL((gram_wide_kernel<64, TIN>), dim3(gw.n_tiles, gw.n_slabs), dim3(256), 0, s, d_A, m, n, lda,
                       partial, gw.tiles_1d, gw.rows_per_slab, vec_ok, d_row_off, d_col_off);
    NDMPS_LAUNCH_CHECK();
    return launch_tile_reduce<64>(partial, gw.n_slabs, gw.tiles_1d, gw.n_tiles, d_G, n, s, d_perm);
  }
  GramGeom g = gram_geometry(m, n);
  NDMPS_REQUIRE(g.n_slabs < 65536, "Gram slab count %d exceeds grid.y", g.n_slabs);
  dim3 grid(g.n_tiles, g.n_slabs);
#define NDMPS_GRAM(TT)                                                                              \
  do {                                                                                              \
    hipLaunchKernelGGL((gram_partial_kernel<TT, TIN>), grid, dim3(256), 0, s, d_A, m, n, lda, partial, \
                       g.tiles_1d, g.rows_per_slab);                                                \
    NDMPS_LAUNCH_CHECK();                                                                           \
    return launch_tile_reduce<16 * TT>(partial, g.n_slabs, g.tiles_1d, g.n_tiles, d_G, n, s);       \
  } while (0)
  if (g.T == 1) NDMPS_GRAM(1);
  else if (g.T == 2) NDMPS_GRAM(2);
  else NDMPS_GRAM(4);
#undef NDMPS_GRAM
  return NDMPS_OK;
}
}  // namespace

extern "C" int ndmps_gram_f32(const float* d_A, int64_t m, int64_t n, int64_t lda, double* d_G,
                              void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  return gram_any<float>(d_A, m, n, lda, d_G, d_ws, ws_bytes, stream);
}

// G = A^T A where element (r, c) of A is d_base[d_row_off[r] + d_col_off[c]] (the C-order volume read through the
// index permutation; d_col_off in aligned runs of four consecutive offsets); wide path only (n >= 64, m >= 256).
// d_col_perm (may be NULL): the columns were visited in another order than the caller numbers them (memory order
// of the volume); entry (a, b) of the product is stored at G[d_col_perm[a]][d_col_perm[b]] by the slab reduction
// itself (a separate pass over a group's 512 x 512 matrices took 0.47 ms per launch of 32).
extern "C" int ndmps_gram_indexed_f32(const float* d_base, int64_t m, int64_t n, const int64_t* d_row_off,
                                      const int64_t* d_col_off, const int32_t* d_col_perm, double* d_G, void* d_ws,
                                      int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_row_off && d_col_off, "NULL offset table");
  return gram_any<float>(d_base, m, n, n, d_G, d_ws, ws_bytes, stream, d_row_off, d_col_off, d_col_perm);
}

// Batched Gram of `batch` matrices of one shape (n >= 128, m >= 256): h_A[b] (host array of device pointers) ->
// d_G + b stride_G.  One launch for the whole batch (what a lockstep group of volumes needs at a site).
namespace {
// 64 <= n < 128: 64 x 64 tiles (gram_wide_body<64>), slabs sized for ~10 workgroups per CU over the whole group (two rounds at five resident)
inline bool gram_use_wide_batched(int batch, int64_t m, int64_t n) { return batch >= 2 && n >= 64 && n < 128 && m >= 256; }
struct GramWideBatchGeom {
  int tiles_1d, n_tiles, n_slabs;
  int64_t rows_per_slab;
  bool stream;  // n == 64: gram64_stream_kernel (the launch also wants 16-byte groups of columns: checked there)
};
GramWideBatchGeom gram_wide_batched_geometry(int batch, int64_t m, int64_t n) {
  GramWideBatchGeom g;
  g.tiles_1d = (int)ndmps::ceil_div(n, 64);
  g.n_tiles = g.tiles_1d * (g.tiles_1d + 1) / 2;
  g.stream = n == 64 && !getenv("NDMPS_GRAM64_TILES");
  // tiles: ~10 workgroups per CU over the group (five resident); stream: 6 per CU (three resident, long slabs)
  const int64_t want = std::max<int64_t>(1, ndmps::ceil_div((int64_t)(g.stream ? 6 : 10) * ndmps::kNumCU, (int64_t)batch * g.n_tiles));
  g.rows_per_slab = ndmps::round_up(std::max<int64_t>(ndmps::ceil_div(m, want), 4 * GW_KB), GW_KB);
  g.n_slabs = (int)std::max<int64_t>(1, ndmps::ceil_div(m, g.rows_per_slab));
  return g;
}
template <typename TIN>
int gram_wide_batched(int batch, const TIN* const* h_A, int64_t m, int64_t n, int64_t lda, double* d_G, int64_t stride_G,
                      void* d_ws, int64_t ws_bytes, hipStream_t s, const int64_t* d_row_off, const int64_t* d_col_off,
                      const int32_t* d_perm) {
  NDMPS_REQUIRE(h_A && d_G && lda >= n && stride_G >= n * n, "bad batched Gram argument");
  const GramWideBatchGeom g = gram_wide_batched_geometry(batch, m, n);
  const int64_t need = (int64_t)batch * g.n_slabs * g.n_tiles * 4096 * 8 + 256;
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("Gram workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  int vec_ok = (lda % 4 == 0 && n % 4 == 0) ? 1 : 0;
  for (int b = 0; b < batch; ++b) {
    NDMPS_REQUIRE(h_A[b], "NULL Gram operand %d", b);
    if ((uintptr_t)h_A[b] % (4 * sizeof(TIN)) != 0) vec_ok = 0;
  }
  if (d_row_off) NDMPS_REQUIRE(d_col_off && vec_ok, "gathered Gram needs n %% 4 == 0 and aligned bases");
  NDMPS_REQUIRE(!g.stream || vec_ok, "the 64-column Gram stream needs lda %% 4 == 0 and 16-byte aligned operands");
  double* partial = (double*)d_ws;
  void* span = ndmps::span_begin(s);
  for (int base = 0; base < batch; base += 64) {
    const int count = std::min(64, batch - base);
    GramBatchPtrs ptrs;
    for (int t = 0; t < count; ++t) ptrs.a[t] = h_A[base + t];
    double* part = partial + (int64_t)base * g.n_slabs * g.n_tiles * 4096;
    if (g.stream) {
      if (d_row_off)
        hipLaunchKernelGGL((gram64_stream_kernel<TIN, true>), dim3(1, g.n_slabs, count), dim3(256), 0, s, ptrs, m, lda, part,
                           g.rows_per_slab, d_row_off, d_col_off);
      else
        hipLaunchKernelGGL((gram64_stream_kernel<TIN, false>), dim3(1, g.n_slabs, count), dim3(256), 0, s, ptrs, m, lda, part,
                           g.rows_per_slab, d_row_off, d_col_off);
    } else
    hipLaunchKernelGGL((gram_wide_batched_kernel<64, TIN>), dim3(g.n_tiles, g.n_slabs, count), dim3(256), 0, s, ptrs, m, n, lda,
                       part, g.tiles_1d, g.rows_per_slab, vec_ok, d_row_off, d_col_off);
    hipLaunchKernelGGL(tile_reduce_batched_kernel<64>, dim3(g.n_tiles, 16, count), dim3(256), 0, s, part, g.n_slabs, g.tiles_1d,
                       g.n_tiles, d_G + (int64_t)base * stride_G, stride_G, n, d_perm);
  }
  ndmps::span_end(span, s, ndmps::kSpanGramSmall, (batch + 63) / 64, (int64_t)batch * m * n * (n + 1));
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
}  // namespace

extern "C" int64_t ndmps_gram_batched_workspace_bytes(int batch, int64_t m, int64_t n) {
  if (batch <= 0) return 0;
  if (gram_use_128(m, n)) return gram128_workspace(m, n, batch);
  if (gram_use_wide_batched(batch, m, n)) {
    const GramWideBatchGeom g = gram_wide_batched_geometry(batch, m, n);
    return (int64_t)batch * g.n_slabs * g.n_tiles * 4096 * 8 + 256;
  }
  return 0;
}
extern "C" int ndmps_gram_batched_f32(int batch, const float* const* h_A, int64_t m, int64_t n, int64_t lda,
                                      double* d_G, int64_t stride_G, void* d_ws, int64_t ws_bytes,
                                      ndmps_stream_t stream) {
  if (!gram_use_128(m, n) && gram_use_wide_batched(batch, m, n))
    return gram_wide_batched<float>(batch, h_A, m, n, lda, d_G, stride_G, d_ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr,
                                    nullptr);
  return gram128_batched<float>(batch, h_A, m, n, lda, d_G, stride_G, d_ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr);
}
extern "C" int ndmps_gram_batched_bf16(int batch, const void* const* h_A, int64_t m, int64_t n, int64_t lda,
                                       double* d_G, int64_t stride_G, void* d_ws, int64_t ws_bytes,
                                       ndmps_stream_t stream) {
  if (!gram_use_128(m, n) && gram_use_wide_batched(batch, m, n))
    return gram_wide_batched<__bf16>(batch, (const __bf16* const*)h_A, m, n, lda, d_G, stride_G, d_ws, ws_bytes,
                                     (hipStream_t)stream, nullptr, nullptr, nullptr);
  return gram128_batched<__bf16>(batch, (const __bf16* const*)h_A, m, n, lda, d_G, stride_G, d_ws, ws_bytes,
                                 (hipStream_t)stream, nullptr, nullptr);
}
extern "C" int ndmps_gram_batched_indexed_f32(int batch, const float* const* h_base, int64_t m, int64_t n,
                                              const int64_t* d_row_off, const int64_t* d_col_off,
                                              const int32_t* d_col_perm, double* d_G, int64_t stride_G, void* d_ws,
                                              int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_row_off && d_col_off, "NULL offset table");
  if (!gram_use_128(m, n) && gram_use_wide_batched(batch, m, n))
    return gram_wide_batched<float>(batch, h_base, m, n, n, d_G, stride_G, d_ws, ws_bytes, (hipStream_t)stream, d_row_off,
                                    d_col_off, d_col_perm);
  return gram128_batched<float>(batch, h_base, m, n, n, d_G, stride_G, d_ws, ws_bytes, (hipStream_t)stream, d_row_off,
                                d_col_off, d_col_perm);
}

// fp64 storage (the reference's own element type, core/ndmps.py:56): the matrix is read as fp64 straight from global
// memory by the tile kernel without LDS staging (64 x 64 tiles of the upper triangle x row slabs) -- a fidelity
// mode, not the throughput path.  Products of two fp64 numbers are rounded: G carries ~sqrt(m) eps relative error.
extern "C" int64_t ndmps_gram_f64_workspace_bytes(int64_t m, int64_t n) {
  if (m <= 0 || n <= 0) return 0;
  GramGeom g = gram_geometry(m, n);
  const int ts = 16 * g.T;
  return (int64_t)(g.n_slabs + g.n_slabs / kReduceGroup + 2) * g.n_tiles * ts * ts * (int64_t)sizeof(double) + 256;
}
extern "C" int ndmps_gram_f64(const double* d_A, int64_t m, int64_t n, int64_t lda, double* d_G, void* d_ws,
                              int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_A && d_G, "NULL Gram operand");
  NDMPS_REQUIRE(m > 0 && n > 0 && lda >= n, "bad Gram extents m=%lld n=%lld lda=%lld", (long long)m, (long long)n,
                (long long)lda);
  if (ws_bytes < ndmps_gram_f64_workspace_bytes(m, n) || d_ws == nullptr) {
    ndmps::set_error("Gram workspace too small: %lld < %lld", (long long)ws_bytes,
                     (long long)ndmps_gram_f64_workspace_bytes(m, n));
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  double* partial = (double*)d_ws;
  GramGeom g = gram_geometry(m, n);
  NDMPS_REQUIRE(g.n_slabs < 65536, "Gram slab count %d exceeds grid.y", g.n_slabs);
  dim3 grid(g.n_tiles, g.n_slabs);
#define NDMPS_GRAM64(TT)                                                                               \
  do {                                                                                                 \
    hipLaunchKernelGGL((gram_partial_kernel<TT, double>), grid, dim3(256), 0, s, d_A, m, n, lda, partial, \
                       g.tiles_1d, g.rows_per_slab);                                                   \
    NDMPS_LAUNCH_CHECK();                                                                              \
    return launch_tile_reduce<16 * TT>(partial, g.n_slabs, g.tiles_1d, g.n_tiles, d_G, n, s);          \
  } while (0)
  if (g.T == 1) NDMPS_GRAM64(1);
  else if (g.T == 2) NDMPS_GRAM64(2);
  else NDMPS_GRAM64(4);
#undef NDMPS_GRAM64
  return NDMPS_OK;
}

// same with a bf16 matrix (products of two bf16 numbers are exact in fp32, let alone fp64)
extern "C" int ndmps_gram_bf16(const void* d_A, int64_t m, int64_t n, int64_t lda, double* d_G,
                               void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  return gram_any<__bf16>((const __bf16*)d_A, m, n, lda, d_G, d_ws, ws_bytes, stream);
}
