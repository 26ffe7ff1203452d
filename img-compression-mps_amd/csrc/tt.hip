// MPS composites: the SVD sweep, bond truncation, chain contraction and overlap.
//
//   ndmps_tt_sweep_f32        <- quimb MatrixProductState.from_dense   (core/ndmps.py:74)
//   ndmps_compress_bond_f32   <- quimb tensor_compress_bond            (core/ndmps.py:104-106)
//   ndmps_chain_contract_f32  <- `mps ^ ...`                           (core/ndmps.py:140)
//   ndmps_overlap_f32         <- `mps @ mps`                           (core/ndmps.py:76,86)
//
// SVD strategy (per site, unfolding A of m rows x n cols, fp32 in HBM):
//   n <= m : G = A^T A in fp64 (exact products), G = V diag(w) V^T by Jacobi; sigma = sqrt(w);
//            site core = V_k^T (k x n), carry = A V_k (m x k, fp32 MFMA GEMM).
//   n >  m : G = A A^T in fp64, G = U diag(w) U^T; carry = U_k diag(sigma_k),
//            core = diag(1/sigma_k) U_k^T A (fp64 GEMM, then fp32).
// The carried matrix is exact whatever the accuracy of sigma (it is an orthogonal projection
// of the data); sigma is accurate to ~1e-8 sigma_0 (fp64 Gram of fp32 data).  Singular values
// below kCutoffFloor * sigma_0 are representation noise of fp32 input and are dropped even
// when the caller's cutoff is smaller (the reference's 1e-10 presumes fp64 data).
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

constexpr double kCutoffFloor = 1e-6;
// fp64 storage: the Gram route resolves singular values down to ~sqrt(eps) s_0
constexpr double kCutoffFloorF64 = 1e-8;
template <typename T>
constexpr double cutoff_floor() { return sizeof(T) == 8 ? kCutoffFloorF64 : kCutoffFloor; }
// Jacobi convergence threshold of the sweep's eigenproblems, relative to the largest eigenvalue.
// The data is fp32: off-diagonal couplings below 1e-13 lambda_0 move the kept subspace by less than
// fp32 rounding even inside a noise-floor cluster (gaps ~1e-9 lambda_0); override for experiments with
// NDMPS_SWEEP_EIG_TOL.
constexpr double kSweepEigTol = 1e-13;

inline double sweep_eig_tol(bool f64_storage = false) {
  const char* e = getenv("NDMPS_SWEEP_EIG_TOL");
  if (e) {
    const double v = atof(e);
    if (v >= 1e-16 && v <= 1e-6) return v;
  }
  return f64_storage ? 1e-15 : kSweepEigTol;  // fp64 data: iterate down to the rounding of the Gram matrix
}

using ndmps::arena_bytes;
using ndmps::Arena;
using ndmps::ceil_div;

inline int grid1d(int64_t n) {
  return (int)std::min<int64_t>(std::max<int64_t>(ceil_div(n, 256), 1), (int64_t)ndmps::kNumCU * 8);
}

// ----------------------------------------------------------------------------- small kernels
// |A v_j|^2 for the eigenvectors v_j = V[:, i0 + j], j < t, of a Gram matrix of A: the singular values behind the SMALLEST
// eigenvalues, which the eigenvalues themselves resolve only to ~1e-15 |G| (3e-8 of the largest singular value).
// A(r, c) = A[r rs + c cs] (rs = n, cs = 1 for G = A^T A; the transposed strides for G = A A^T).  Thread (row, j):
// 4 rows x 64 columns per workgroup; partial[blockIdx.x][j] = sum over the workgroup's rows; tail_norm_reduce_kernel
// adds the row blocks in fixed order.
template <typename T>
__global__ void __launch_bounds__(256)
tail_norm_partial_kernel(const T* __restrict__ A, int64_t rs, int64_t cs, int rows, int cols, const double* __restrict__ V,
                         int ldv, int i0, int t, double* __restrict__ partial) {
  __shared__ double red[4][64];
  const int jl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int j = blockIdx.y * 64 + jl, r = blockIdx.x * 4 + rl;
  double y = 0.0;
  if (r < rows && j < t) {
    const T* a = A + (int64_t)r * rs;
    const double* v = V + i0 + j;
    for (int c = 0; c < cols; ++c) y = fma(ndmps::to_f64(a[(int64_t)c * cs]), v[(int64_t)c * ldv], y);
  }
  red[rl][jl] = y * y;
  __syncthreads();
  if (rl == 0 && j < t) partial[(int64_t)blockIdx.x * t + j] = (red[0][jl] + red[1][jl]) + (red[2][jl] + red[3][jl]);
}
__global__ void tail_norm_reduce_kernel(const double* __restrict__ partial, int nblk, int t, double* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= t) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += partial[(int64_t)b * t + j];
  out[j] = s;
}

__global__ void set_scalar_f64_kernel(double* p, double v) { *p = v; }

template <typename T = float>
__global__ void __launch_bounds__(256) f32_to_f64_kernel(const T* __restrict__ x, int64_t n, double* y) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    y[i] = ndmps::to_f64(x[i]);
}

// core (k x n) fp32 <- first k columns of V (n x n fp64), transposed
template <typename T = float>
__global__ void __launch_bounds__(256)
core_from_vectors_kernel(const double* __restrict__ V, int64_t n, int64_t k, T* __restrict__ core) {
  const int64_t total = k * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / n, c = e % n;
    core[e] = ndmps::from_f64<T>(V[c * n + i]);
  }
}

// out (rows x k) fp32 <- M (rows x ldm fp64)[:, :k] * scale[col]^power
template <typename T = float>
__global__ void __launch_bounds__(256)
scale_cols_to_f32_kernel(const double* __restrict__ M, int64_t rows, int64_t ldm, int64_t k,
                         const double* __restrict__ sigma, double power, T* __restrict__ out) {
  const int64_t total = rows * k;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / k, c = e % k;
    const double sg = sigma[c];
    out[e] = ndmps::from_f64<T>(sg > 0.0 ? M[r * ldm + c] * pow(sg, power) : 0.0);
  }
}

// out (k x n) fp32 <- M (k x n fp64) with row i scaled by sigma[i]^power
template <typename T = float>
__global__ void __launch_bounds__(256)
scale_rows_to_f32_kernel(const double* __restrict__ M, int64_t k, int64_t n, const double* __restrict__ sigma,
                         double power, T* __restrict__ out) {
  const int64_t total = k * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
  {
    const double sg = sigma[e / n];
    out[e] = ndmps::from_f64<T>(sg > 0.0 ? M[e] * pow(sg, power) : 0.0);
  }
}

// in-place: M (rows x cols fp64), column c scaled by sqrt(max(w[c], 0))
__global__ void __launch_bounds__(256)
scale_cols_sqrt_kernel(double* __restrict__ M, int64_t rows, int64_t cols, const double* __restrict__ w) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    M[e] *= sqrt(fmax(w[e % cols], 0.0));
}

__global__ void __launch_bounds__(256) sqrt_clamp_kernel(const double* __restrict__ w, int64_t n, double* s) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    s[i] = sqrt(fmax(w[i], 0.0));
}

// ---- the same small kernels for a whole lockstep group (blockIdx.y = volume): fp64 side arrays are strided in the
//      workspace, volumes / carried matrices / cores come as pointers in the kernel arguments
constexpr int kSmallBatch = 64;
struct BatchOps {
  const void* in[kSmallBatch];
  void* out[kSmallBatch];
};
template <typename T>
__global__ void __launch_bounds__(256) f32_to_f64_batched_kernel(BatchOps ops, int64_t n, double* __restrict__ y, int64_t y_stride) {
  const T* x = static_cast<const T*>(ops.in[blockIdx.y]);
  double* yb = y + (int64_t)blockIdx.y * y_stride;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) yb[i] = ndmps::to_f64(x[i]);
}
template <typename T>
__global__ void __launch_bounds__(256)
core_from_vectors_batched_kernel(const double* __restrict__ V, int64_t v_stride, int64_t n, int64_t k, BatchOps ops) {
  const double* Vb = V + (int64_t)blockIdx.y * v_stride;
  T* core = static_cast<T*>(ops.out[blockIdx.y]);
  const int64_t total = k * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t i = e / n, c = e % n;
    core[e] = ndmps::from_f64<T>(Vb[c * n + i]);
  }
}
__global__ void __launch_bounds__(256)
sqrt_clamp_batched_kernel(const double* __restrict__ w, int64_t stride, int64_t n, double* __restrict__ sg) {
  const double* wb = w + (int64_t)blockIdx.y * stride;
  double* sb = sg + (int64_t)blockIdx.y * stride;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) sb[i] = sqrt(fmax(wb[i], 0.0));
}
template <typename T>
__global__ void __launch_bounds__(256)
scale_cols_to_f32_batched_kernel(const double* __restrict__ M, int64_t m_stride, int64_t rows, int64_t ldm, int64_t k,
                                 const double* __restrict__ sigma, int64_t s_stride, double power, BatchOps ops) {
  const double* Mb = M + (int64_t)blockIdx.y * m_stride;
  const double* sb = sigma + (int64_t)blockIdx.y * s_stride;
  T* out = static_cast<T*>(ops.out[blockIdx.y]);
  const int64_t total = rows * k;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / k, c = e % k;
    const double sg = sb[c];
    out[e] = ndmps::from_f64<T>(sg > 0.0 ? Mb[r * ldm + c] * pow(sg, power) : 0.0);
  }
}
template <typename T>
__global__ void __launch_bounds__(256)
scale_rows_to_f32_batched_kernel(const double* __restrict__ M, int64_t m_stride, int64_t k, int64_t n,
                                 const double* __restrict__ sigma, int64_t s_stride, double power, BatchOps ops) {
  const double* Mb = M + (int64_t)blockIdx.y * m_stride;
  const double* sb = sigma + (int64_t)blockIdx.y * s_stride;
  T* out = static_cast<T*>(ops.out[blockIdx.y]);
  const int64_t total = k * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const double sg = sb[e / n];
    out[e] = ndmps::from_f64<T>(sg > 0.0 ? Mb[e] * pow(sg, power) : 0.0);
  }
}

// ------------------------------------------------------------ merged trailing sites (bond-capped sweep)
// While a bond is exact (the product N_{i+1} of the site dims to its right does not exceed the cap) site i's
// unfolding is the RAW unfolding A_i (M_i x N_i) times a block-diagonal basis:  A_i (I_{d_i} (x) W_{i+1}),
// W_{i+1} (N_{i+1} x k_{i+1}) the accumulated right bases.  Its Gram matrix is therefore a congruence of the
// raw one, and the raw Gram matrices of all those sites are block sums of the largest:
//     G_i = B^T Graw_i B,  B = I (x) W_{i+1},   Graw_{i+1} = sum of the d_i diagonal blocks of Graw_i.
// One Gram pass over the raw tensor (order N_{i0}) thus serves every site i >= i0, and one projection
// A_{i0} W_{i0} replaces the per-site projections: the tensor is read twice instead of once per Gram and
// once per projection of every site, and the cores are the same SVD cores (same arithmetic, fewer roundings).
struct MergeRanks {  // per-volume ranks of one launch (kernel argument)
  int k_right[64];   // k_{i+1}
  int k_here[64];    // k_i (w_update only)
};

// T (N_i x n_i) = Graw_i B:  T[r][(b, q)] = sum_c' Graw_i[r][b N' + c'] W[c'][q],
// Graw_i[r][c] = sum_t Graw[(t N_i + r)][(t N_i + c)]  (t over the N_top / N_i diagonal blocks of the top Gram)
__global__ void __launch_bounds__(256)
merge_stage1_kernel(const double* __restrict__ Gtop, int64_t stride_top, int n_top, const double* __restrict__ W,
                    int64_t stride_w, int ldw, double* __restrict__ T, int64_t stride_t, int n_i, int n_right,
                    int d_i, MergeRanks rk) {
  const int b = blockIdx.y;
  const int k = rk.k_right[b];
  const int cols = d_i * k;  // n_i of this volume
  const double* G = Gtop + (int64_t)b * stride_top;
  const double* Wb = W + (int64_t)b * stride_w;
  double* Tb = T + (int64_t)b * stride_t;
  const int reps = n_top / n_i;
  const int64_t total = (int64_t)n_i * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / cols), col = (int)(e % cols);
    const int bb = col / k, q = col % k;
    double acc = 0.0;
    for (int t = 0; t < reps; ++t) {
      const double* grow = G + (int64_t)(t * n_i + r) * n_top + t * n_i + bb * n_right;
      for (int c = 0; c < n_right; ++c) acc = fma(grow[c], Wb[(int64_t)c * ldw + q], acc);
    }
    Tb[e] = acc;
  }
}

// G_i (n_i x n_i, ld n_i) = B^T T:  G[(a, p)][j] = sum_c' W[c'][p] T[(a N' + c')][j]
__global__ void __launch_bounds__(256)
merge_stage2_kernel(const double* __restrict__ T, int64_t stride_t, const double* __restrict__ W, int64_t stride_w,
                    int ldw, double* __restrict__ Gout, int64_t stride_g, int n_right, int d_i, MergeRanks rk) {
  const int b = blockIdx.y;
  const int k = rk.k_right[b];
  const int n = d_i * k;
  const double* Tb = T + (int64_t)b * stride_t;
  const double* Wb = W + (int64_t)b * stride_w;
  double* Gb = Gout + (int64_t)b * stride_g;
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int row = (int)(e / n), j = (int)(e % n);
    const int a = row / k, pp = row % k;
    double acc = 0.0;
    for (int c = 0; c < n_right; ++c) acc = fma(Wb[(int64_t)c * ldw + pp], Tb[(int64_t)(a * n_right + c) * n + j], acc);
    Gb[e] = acc;
  }
}

// W_i (N_i x k_i, ld ldw) = B V_i:  W_i[(a, c')][p] = sum_q W_{i+1}[c'][q] V[(a k_{i+1} + q)][p]   (V: n_i x n_i, ld n_i)
// optionally also as fp32 (ld = k_i, compact) for the projection GEMM
template <typename T>
__global__ void __launch_bounds__(256)
merge_basis_kernel(const double* __restrict__ Wr, int64_t stride_w, int ldw, const double* __restrict__ V,
                   int64_t stride_v, double* __restrict__ Wout, T* __restrict__ W32, int64_t stride_w32,
                   int n_right, int d_i, MergeRanks rk) {
  const int b = blockIdx.y;
  const int kr = rk.k_right[b], kh = rk.k_here[b];
  const int n = d_i * kr;
  const double* Wb = Wr + (int64_t)b * stride_w;
  const double* Vb = V + (int64_t)b * stride_v;
  double* Ob = Wout + (int64_t)b * stride_w;
  T* O32 = W32 ? W32 + (int64_t)b * stride_w32 : nullptr;
  const int64_t total = (int64_t)d_i * n_right * kh;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int row = (int)(e / kh), pp = (int)(e % kh);
    const int a = row / n_right, c = row % n_right;
    double acc = 0.0;
    for (int q = 0; q < kr; ++q) acc = fma(Wb[(int64_t)c * ldw + q], Vb[(int64_t)(a * kr + q) * n + pp], acc);
    Ob[(int64_t)row * ldw + pp] = acc;
    if (O32) O32[(int64_t)row * kh + pp] = ndmps::from_f64<T>(acc);
  }
}

__global__ void merge_basis_init_kernel(double* __restrict__ W, int64_t stride_w) {
  W[(int64_t)blockIdx.x * stride_w] = 1.0;  // W_L = [1]
}

// Fused reshape stage: the raw Gram pass reads the volume with its columns in MEMORY order (perm[c'] = site-order
// column of the c'-th smallest offset) and its slab reduction stores G'[a][b] at G[perm[a]][perm[b]]
// (ndmps_gram_indexed_f32) ...
// ... and the projection multiplies by the basis with its rows in memory order: out[c'] = W[perm[c']]
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float* __restrict__ W, int64_t rows, int64_t cols, const int32_t* __restrict__ perm,
                   float* __restrict__ out, int64_t w_stride = 0, int64_t out_stride = 0) {  // volume blockIdx.y
  W += (int64_t)blockIdx.y * w_stride;
  out += (int64_t)blockIdx.y * out_stride;
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    out[e] = W[(int64_t)perm[e / cols] * cols + e % cols];
}

struct SweepSource {           // the volume read through the index permutation (fp32, merged run only)
  const int64_t* row_off;      // [numel / n_cols]
  const int64_t* row_sorted;   // the same offsets in ascending order: the Gram kernels -- a sum over rows -- read the volume
                               // front to back (16-byte pieces of adjacent rows are neighbours in memory); may equal row_off
  const int32_t* row_order;    // row_sorted[s] = row_off[row_order[s]] (NULL: not given); the streamed projection
  const int64_t* col_off;      // [n_cols], ascending, aligned runs of four consecutive offsets
  const int32_t* col_perm;     // [n_cols]
  int64_t n_cols;
};

constexpr int64_t kMergeMax = 512;  // largest raw Gram order of the merged sites

// first site of the merged run, or L when nothing is merged (see the comment above)
int merge_start(int L, const int64_t* dims, int64_t numel, int64_t max_bond) {
  if (max_bond <= 0 || getenv("NDMPS_SWEEP_NO_MERGE")) return L;
  int best = L;
  int64_t right = 1;  // N_{i+1}
  for (int i = L - 1; i >= 1; --i) {
    const int64_t n_i = right * dims[i];
    if (right > max_bond || n_i > kMergeMax || numel / n_i < n_i) break;
    best = i;
    right = n_i;
  }
  return best >= L - 1 ? L : best;  // a run of one site is the ordinary path
}

// keep s_k > cutoff * s_0 (at least one), at most max_bond
int64_t kept_rank(const std::vector<double>& sigma, double cutoff, int64_t max_bond, double floor = kCutoffFloor) {
  const int64_t n = (int64_t)sigma.size();
  const double c = std::max(cutoff, floor);
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) k += sigma[i] > c * sigma[0];
  k = std::max<int64_t>(k, 1);
  if (max_bond > 0) k = std::min(k, max_bond);
  return std::min(k, n);
}

// The bond-capped sweep wants chi <= 128 of the n eigenpairs of every site: the direct solver
// (eig_tridiag.hip).  Exact sweeps (all eigenpairs above the cutoff) and orders beyond its limit stay on
// the block Jacobi.
inline bool use_topk(int64_t n_max, int64_t max_bond) {
  if (getenv("NDMPS_SWEEP_JACOBI")) return false;  // A/B timing
  return max_bond > 0 && max_bond <= ndmps_syevd_topk_max_k() && n_max <= ndmps_syevd_topk_max_n();
}
// Sweeps that want every eigenpair above a cutoff (no bond cap, or one beyond the direct solver's 128) take the direct
// solver too, with the eigenvectors orthonormalised across the chip (eig_wide.inc), while its workspace -- four
// n x n planes of LU factors per matrix -- stays below kDirectFullBytes; orders beyond its limit stay on the block Jacobi.
constexpr int64_t kDirectFullBytes = (int64_t)48 << 30;
inline bool use_direct_full(int64_t n_max, int batch, int64_t max_bond) {
  if (getenv("NDMPS_SWEEP_JACOBI") || getenv("NDMPS_EXACT_JACOBI")) return false;  // A/B timing
  if (max_bond > 0 && max_bond <= ndmps_syevd_topk_max_k()) return false;            // the bond-capped path
  if (n_max > ndmps_syevd_topk_max_n()) return false;
  const int64_t need = ndmps_syevd_topk_workspace_bytes(n_max, batch, n_max);
  return need > 0 && need <= kDirectFullBytes;
}
inline int64_t eig_workspace_bytes(int64_t n_max, int batch, int64_t max_bond) {
  const int64_t jac = ndmps_syevj_batched_workspace_bytes(n_max, batch);
  if (max_bond > 0 && max_bond <= ndmps_syevd_topk_max_k() && n_max <= ndmps_syevd_topk_max_n())
    return std::max(jac, ndmps_syevd_topk_workspace_bytes(n_max, batch, std::min(max_bond, n_max)));
  if (use_direct_full(n_max, batch, max_bond)) return std::max(jac, ndmps_syevd_topk_workspace_bytes(n_max, batch, n_max));
  return jac;
}
// The direct solver leaves eigenvalue errors of up to ~1e-14 |G| (measured 6e-16 at order 4096, tools/full_probe.py):
// eigenvalues within kDirectDoubt |G| of the threshold c^2 w_0 cannot decide a rank -- under the tiny cutoffs of exact
// sweeps that is the whole lower end of the spectrum of a square noisy unfolding (its smallest singular values reach
// zero) and every numerically zero eigenvalue of a rank-deficient matrix.  First index (descending order) from which
// the eigenvalues are in doubt; n: none.
constexpr double kDirectDoubt = 1e-13;
inline int64_t direct_doubt_from(const double* w_desc, int64_t n, double c) {
  if (n < 1) return n;
  const double hi = c * c * w_desc[0] + kDirectDoubt * fabs(w_desc[0]);
  int64_t i = n;
  while (i > 0 && w_desc[i - 1] <= hi) --i;
  return i;
}
inline bool direct_rank_is_safe(const double* w_desc, int64_t n, double c) {  // nothing in doubt around the threshold
  if (n < 1) return true;
  const double thr = c * c * w_desc[0], delta = kDirectDoubt * fabs(w_desc[0]);
  for (int64_t i = 0; i < n; ++i)
    if (fabs(w_desc[i] - thr) <= delta) return false;
  return true;
}

// upper bound of ndmps_gram_workspace_bytes(m, n') over every n' <= n (the actual bond may
// come out smaller than the worst case the layout is sized for): slabs * tiles <=
// max(1024, tiles(n)), 64 x 64 doubles each.
int64_t gram_ws_bound(int64_t n, int batch = 1) {
  const int64_t t1 = ceil_div(n, 64);
  // (slabs + slabs/16 + 2) * tiles tiles of 64 x 64 doubles, slabs * tiles <= max(512, tiles(n))
  const int64_t nt1 = t1 * (t1 + 1) / 2;
  const int64_t narrow = (std::max<int64_t>(512, nt1) * 17 / 16 + 3 * nt1) * 4096 * 8 + 256;
  // 128-wide path: same bound with 128 x 128 tiles
  const int64_t t2 = ceil_div(n, 128);
  const int64_t nt2 = t2 * (t2 + 1) / 2;
  const int64_t wide = (std::max<int64_t>(512, nt2) * 17 / 16 + 3 * nt2) * 16384 * 8 + 256;
  // a lockstep group in one launch (ndmps_gram_batched_*): ~12 rounds of 512 partial tiles in total, plus the
  // rounding of the slab counts per matrix
  const int64_t batched = batch > 1 ? (12 * 512 + 3 * nt2 * batch + 64) * 16384 * 8 + 256 : 0;
  return std::max(std::max(narrow, wide), batched);
}

// ------------------------------------------------------------------ sweep layout (worst case)
struct SweepLayout {
  std::vector<int64_t> max_bonds, core_off, spec_off;
  int64_t numel = 1;
  int64_t small_max = 1;      // largest eigenproblem
  int64_t gram_ws = 0;        // largest Gram workspace
  int64_t wide_elems = 0;     // largest wide unfolding (m < n), elements
  int merge_from = 0;         // first site of the merged trailing run (== L: none)
  int64_t merge_n = 0;        // order of its raw Gram matrix
  int64_t merge_w = 0;        // leading dimension of the accumulated bases
  int64_t transpose_bytes = 0;  // bf16 path: transposed copy of the merged basis
  bool device_rank = false;   // every site on the direct solver: ranks decided on the device, padded cores
  int64_t spec_stride = 0;    // singular values kept per site and volume on the device
  int64_t workspace = 0;      // for the batch size it was computed for
};

int sweep_layout(int L, const int64_t* dims, int64_t max_bond, int batch, SweepLayout& out, int elem_bytes = 4) {
  NDMPS_REQUIRE(L >= 1 && L <= 64, "L=%d outside [1, 64]", L);
  NDMPS_REQUIRE(batch >= 1 && batch <= 4096, "batch=%d outside [1, 4096]", batch);
  out.numel = 1;
  for (int i = 0; i < L; ++i) {
    NDMPS_REQUIRE(dims[i] >= 1, "dims[%d]=%lld must be positive", i, (long long)dims[i]);
    out.numel *= dims[i];
  }
  out.max_bonds.assign(L + 1, 1);
  std::vector<int64_t> left(L + 1, 1), right(L + 1, 1);
  for (int i = 0; i < L; ++i) left[i + 1] = left[i] * dims[i];
  for (int i = L - 1; i >= 0; --i) right[i] = right[i + 1] * dims[i];
  for (int i = 1; i < L; ++i) {
    int64_t b = std::min(left[i], right[i]);
    if (max_bond > 0) b = std::min(b, max_bond);
    out.max_bonds[i] = b;
  }
  out.core_off.assign(L + 1, 0);
  out.spec_off.assign(L + 1, 0);
  for (int i = 0; i < L; ++i) {
    out.core_off[i + 1] = out.core_off[i] + ndmps::round_up(out.max_bonds[i] * dims[i] * out.max_bonds[i + 1], 64);
    const int64_t m = left[i], n = dims[i] * out.max_bonds[i + 1];
    out.spec_off[i + 1] = out.spec_off[i] + (i == 0 ? 0 : std::min(m, n));
    if (i >= 1) {
      const int64_t small = std::min(m, n);
      out.small_max = std::max(out.small_max, small);
      if (n <= m) out.gram_ws = std::max(out.gram_ws, gram_ws_bound(n, batch));
      else out.wide_elems = std::max(out.wide_elems, m * n);
    }
  }
  out.merge_from = merge_start(L, dims, out.numel, max_bond);
  out.merge_n = out.merge_from < L ? right[out.merge_from] : 0;
  out.merge_w = out.merge_from < L ? std::min(out.merge_n, max_bond) : 0;
  if (out.merge_from < L) {
    out.small_max = std::max(out.small_max, out.merge_n);
    out.gram_ws = std::max(out.gram_ws, gram_ws_bound(out.merge_n, batch));
  }
  // Rank decision on the device: possible when every site's eigenproblem (order min(rows, d_i cap_{i+1}) with
  // the bonds at their caps) goes to the direct top-k solver.  The sweep then sizes everything by the caps,
  // zero-fills the columns beyond a volume's rank and never waits for the host between sites.
  out.device_rank = max_bond > 0 && max_bond <= ndmps_syevd_topk_max_k() && !getenv("NDMPS_SWEEP_HOST_RANK") &&
                    !getenv("NDMPS_SWEEP_JACOBI");
  for (int i = 1; i < L && out.device_rank; ++i)
    if (std::min(left[i], dims[i] * out.max_bonds[i + 1]) > ndmps_syevd_topk_max_n()) out.device_rank = false;
  out.spec_stride = max_bond > 0 ? std::min<int64_t>(max_bond, out.small_max) : 0;
  const int64_t sq = out.small_max * out.small_max;
  int64_t used = 0;
  used = arena_bytes(used, elem_bytes, (int64_t)batch * out.numel);     // second carry buffer per volume
  used = arena_bytes(used, 8, (int64_t)batch * sq);                     // G
  used = arena_bytes(used, 8, (int64_t)batch * sq);                     // V / U
  used = arena_bytes(used, 8, (int64_t)batch * out.small_max);          // w
  used = arena_bytes(used, 8, (int64_t)batch * out.small_max);          // sigma
  used = arena_bytes(used, 1, eig_workspace_bytes(out.small_max, batch, max_bond));
  used = arena_bytes(used, 1, out.gram_ws);                             // shared, stream-ordered
  used = arena_bytes(used, 8, (int64_t)batch * out.wide_elems);         // A64 per volume
  used = arena_bytes(used, 8, (int64_t)batch * out.wide_elems);         // U_k^T A64 per volume
  used = arena_bytes(used, 8, (int64_t)batch * out.merge_n * out.merge_n);      // raw Gram of the merged run
  used = arena_bytes(used, 8, (int64_t)batch * out.merge_n * out.merge_n);      // T = Graw B
  used = arena_bytes(used, 8, (int64_t)batch * out.merge_n * out.merge_w);      // accumulated basis W (ping)
  used = arena_bytes(used, 8, (int64_t)batch * out.merge_n * out.merge_w);      // (pong)
  used = arena_bytes(used, elem_bytes, (int64_t)batch * out.merge_n * out.merge_w);  // copy in the storage type for the projection
  out.transpose_bytes = ndmps_gemm_bf16_workspace_bytes(0, std::max<int64_t>(out.merge_w, 1), std::max<int64_t>(out.merge_n, 1));
  used = arena_bytes(used, 1, out.transpose_bytes);
  used = arena_bytes(used, 4, (int64_t)2 * L * batch);                          // device ranks, status per site
  used = arena_bytes(used, 8, (int64_t)L * batch * out.spec_stride);            // device spectra per site
  out.workspace = ndmps::round_up(used, 256) + 256;
  return NDMPS_OK;
}

}  // namespace

extern "C" int ndmps_tt_layout(int L, const int64_t* h_dims, int64_t max_bond, int64_t* h_max_bonds,
                               int64_t* h_core_offsets, int64_t* h_spec_offsets,
                               int64_t* h_workspace_bytes) {
  NDMPS_REQUIRE(h_dims, "NULL dims");
  SweepLayout lay;
  NDMPS_TRY(sweep_layout(L, h_dims, max_bond, 1, lay));
  for (int i = 0; i <= L; ++i) {
    if (h_max_bonds) h_max_bonds[i] = lay.max_bonds[i];
    if (h_core_offsets) h_core_offsets[i] = lay.core_off[i];
    if (h_spec_offsets) h_spec_offsets[i] = lay.spec_off[i];
  }
  if (h_workspace_bytes) *h_workspace_bytes = lay.workspace;
  return NDMPS_OK;
}

extern "C" int64_t ndmps_tt_sweep_batched_workspace_bytes(int batch, int L, const int64_t* h_dims,
                                                          int64_t max_bond) {
  SweepLayout lay;
  if (!h_dims || sweep_layout(L, h_dims, max_bond, batch, lay) != NDMPS_OK) return -1;
  return lay.workspace;
}
// the same for fp64 storage (ndmps_tt_sweep_batched_f64): carried matrices of 8-byte elements
extern "C" int64_t ndmps_tt_sweep_batched_workspace_bytes_f64(int batch, int L, const int64_t* h_dims,
                                                              int64_t max_bond) {
  SweepLayout lay;
  if (!h_dims || sweep_layout(L, h_dims, max_bond, batch, lay, 8) != NDMPS_OK) return -1;
  return lay.workspace;
}

// All volumes of the batch have the same site dims; they advance through the sites in lockstep
// so that every site's eigenproblems are solved by ONE batched Jacobi (its sequential depth is
// the cost of the path); Gram / projection launches stay per volume (they fill the chip alone).
namespace {
// element-type dispatch of the two streaming products of the sweep
inline int gram_T(const float* A, int64_t m, int64_t n, int64_t lda, double* G, void* ws, int64_t wsb, hipStream_t s) {
  return ndmps_gram_f32(A, m, n, lda, G, ws, wsb, s);
}
inline int gram_T(const __bf16* A, int64_t m, int64_t n, int64_t lda, double* G, void* ws, int64_t wsb, hipStream_t s) {
  return ndmps_gram_bf16(A, m, n, lda, G, ws, wsb, s);
}
inline int gram_T(const double* A, int64_t m, int64_t n, int64_t lda, double* G, void* ws, int64_t wsb, hipStream_t s) {
  return ndmps_gram_f64(A, m, n, lda, G, ws, wsb, s);
}
inline int64_t gram_need(const float*, int64_t m, int64_t n) { return ndmps_gram_workspace_bytes(m, n); }
inline int64_t gram_need(const __bf16*, int64_t m, int64_t n) { return ndmps_gram_workspace_bytes(m, n); }
inline int64_t gram_need(const double*, int64_t m, int64_t n) { return ndmps_gram_f64_workspace_bytes(m, n); }
// the group-wide Gram launch (LDS-staged fp32 panels) exists for fp32 / bf16 storage only
template <typename T>
inline int64_t gram_batched_need(int batch, int64_t m, int64_t n) {
  return sizeof(T) == 8 ? 0 : ndmps_gram_batched_workspace_bytes(batch, m, n);
}
// one launch for the Gram matrices of a lockstep group (same shape; n >= 128, m >= 256)
inline int gram_batched_T(int batch, const float* const* A, int64_t m, int64_t n, double* G, int64_t stride, void* ws,
                          int64_t wsb, hipStream_t s) {
  return ndmps_gram_batched_f32(batch, A, m, n, n, G, stride, ws, wsb, s);
}
inline int gram_batched_T(int batch, const __bf16* const* A, int64_t m, int64_t n, double* G, int64_t stride, void* ws,
                          int64_t wsb, hipStream_t s) {
  return ndmps_gram_batched_bf16(batch, (const void* const*)A, m, n, n, G, stride, ws, wsb, s);
}
inline int gram_batched_T(int, const double* const*, int64_t, int64_t, double*, int64_t, void*, int64_t, hipStream_t) {
  ndmps::set_error("internal: no group-wide Gram launch for fp64 storage");
  return NDMPS_EINVAL;
}
inline int gram_batched_src(int, const double* const*, int64_t, int64_t, const SweepSource&, double*, int64_t, void*,
                            int64_t, hipStream_t) {
  ndmps::set_error("the fused reshape stage is fp32 only");
  return NDMPS_EINVAL;
}
inline int gram_batched_src(int batch, const float* const* vol, int64_t m, int64_t n, const SweepSource& src, double* G,
                            int64_t stride, void* ws, int64_t wsb, hipStream_t s) {
  return ndmps_gram_batched_indexed_f32(batch, vol, m, n, src.row_sorted, src.col_off, src.col_perm, G, stride, ws, wsb, s);
}
inline int gram_batched_src(int, const __bf16* const*, int64_t, int64_t, const SweepSource&, double*, int64_t, void*,
                            int64_t, hipStream_t) {
  ndmps::set_error("the fused reshape stage is fp32 only");
  return NDMPS_EINVAL;
}
// C (m, n) = A (m, k) op(B);  tws: scratch of the bf16 path (transposed copy of a (k, n) right operand)
inline int gemm_T(int transB, int64_t m, int64_t n, int64_t k, const float* A, const float* B, int64_t ldb, float* C,
                  void*, int64_t, hipStream_t s) {
  return ndmps_sgemm(0, transB, m, n, k, A, k, B, ldb, C, n, s);
}
inline int gemm_T(int transB, int64_t m, int64_t n, int64_t k, const __bf16* A, const __bf16* B, int64_t ldb, __bf16* C,
                  void* tws, int64_t tws_bytes, hipStream_t s) {
  return ndmps_gemm_bf16(transB, m, n, k, A, k, B, ldb, C, n, tws, tws_bytes, s);
}

inline int gemm_T(int transB, int64_t m, int64_t n, int64_t k, const double* A, const double* B, int64_t ldb, double* C,
                  void*, int64_t, hipStream_t s) {
  return ndmps_dgemm(0, transB, m, n, k, A, k, B, ldb, C, n, s);
}

// general product with explicit leading dimensions (fp32 / fp64)
inline int gemm_any(int tA, int tB, int64_t m, int64_t n, int64_t k, const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, hipStream_t s) {
  return ndmps_sgemm(tA, tB, m, n, k, A, lda, B, ldb, C, ldc, s);
}
inline int gemm_any(int tA, int tB, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                    int64_t ldb, double* C, int64_t ldc, hipStream_t s) {
  return ndmps_dgemm(tA, tB, m, n, k, A, lda, B, ldb, C, ldc, s);
}

// products of a whole lockstep group in one launch (fp32 / fp64 storage; bf16 storage goes volume by volume)
inline bool gemm_batched_T(int batch, int transB, int64_t m, int64_t n, int64_t k, double* const* A, double* const* B,
                           int64_t ldb, double* const* C, hipStream_t s, int* rc) {
  if (batch > ndmps_gemm_batched_max()) return false;
  *rc = ndmps_dgemm_batched(batch, 0, transB, m, n, k, (const double* const*)A, k, (const double* const*)B, ldb, C, n, s);
  return true;
}
inline bool gemm_batched_T(int batch, int transB, int64_t m, int64_t n, int64_t k, float* const* A, float* const* B,
                           int64_t ldb, float* const* C, hipStream_t s, int* rc) {
  if (batch > ndmps_gemm_batched_max()) return false;
  *rc = ndmps_sgemm_batched(batch, 0, transB, m, n, k, (const float* const*)A, k, (const float* const*)B, ldb, C, n, s);
  return true;
}
inline bool gemm_batched_T(int, int, int64_t, int64_t, int64_t, __bf16* const*, __bf16* const*, int64_t, __bf16* const*,
                           hipStream_t, int*) {
  return false;
}
// fp32 only: Gram and projection of the merged run through the permutation tables
inline int gram_src(const float* vol, int64_t m, int64_t n, const SweepSource& src, double* G, void* ws, int64_t wsb,
                    hipStream_t s) {
  return ndmps_gram_indexed_f32(vol, m, n, src.row_sorted, src.col_off, src.col_perm, G, ws, wsb, s);
}
inline int gram_src(const __bf16*, int64_t, int64_t, const SweepSource&, double*, void*, int64_t, hipStream_t) {
  ndmps::set_error("the fused reshape stage is fp32 only");
  return NDMPS_EINVAL;
}
inline int gram_src(const double*, int64_t, int64_t, const SweepSource&, double*, void*, int64_t, hipStream_t) {
  ndmps::set_error("the fused reshape stage is fp32 only");
  return NDMPS_EINVAL;
}
inline int project_src(const double*, int64_t, int64_t, int64_t, const SweepSource&, const double*, double*, double*,
                       hipStream_t) {
  return NDMPS_EINVAL;
}
inline int project_src(const float* vol, int64_t m, int64_t k, int64_t n, const SweepSource& src, const float* W,
                       float* scratch, float* out, hipStream_t s) {
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid1d(n * k)), dim3(256), 0, s, W, n, k, src.col_perm, scratch);
  NDMPS_LAUNCH_CHECK();
  return ndmps_sgemm_indexed(m, k, n, vol, 0, src.row_off, src.col_off, 1, scratch, k, out, k, nullptr, nullptr, s);
}
inline int project_src(const __bf16*, int64_t, int64_t, int64_t, const SweepSource&, const __bf16*, __bf16*, __bf16*,
                       hipStream_t) {
  return NDMPS_EINVAL;
}

// status of the direct solver for one matrix (eig_tridiag.hip): 1 = Cholesky breakdown, 2 = a team gave up waiting
inline int solver_failed(int site, int volume, int status) {
  if (status == 2) {
    ndmps::set_error("site %d, volume %d: the resident tridiagonalisation gave up waiting for its workgroups (3 s; is the "
                     "GPU shared with another process?); repeat the sweep after ndmps_syevd_topk_set_team(0)", site, volume);
    return NDMPS_ETEAM;
  }
  ndmps::set_error("site %d, volume %d: eigenvector block lost rank in the orthonormalisation", site, volume);
  return NDMPS_ENOCONV;
}

// The two halves of a sweep whose ranks are decided on the device: everything such a sweep needs from the host is known
// before it starts, so it can be ENQUEUED as a whole (SweepAsync: ranks, status words and kept singular values travel to
// pinned host memory behind the last kernel, nobody waits) and READ later, once the caller has synchronised with the
// stream (sweep_collect: bonds, spectra, a solver's failure).  The caller's host thread is free in between -- the objects
// of the previous batch are built while this one runs (core/batch.py).
struct SweepAsync {
  int* h_ranks;    // pinned, 2 L batch ints: [site][volume] ranks, then [site][volume] solver status
  double* h_spec;  // pinned, L batch spec_stride doubles (may be NULL when the layout keeps no spectra)
};
int sweep_collect(int batch, int L, int64_t spec_stride, const int* host_i, const double* host_s, int64_t* h_bonds_out,
                  double* h_spectra, const int64_t* h_spec_offsets) {
  const int64_t spec_total = h_spec_offsets ? h_spec_offsets[L] : 0;
  for (int b = 0; b < batch; ++b) {
    h_bonds_out[(int64_t)b * (L + 1)] = 1;
    h_bonds_out[(int64_t)b * (L + 1) + L] = 1;
  }
  for (int i = 1; i < L; ++i)
    for (int b = 0; b < batch; ++b) {
      if (host_i[(size_t)L * batch + (size_t)i * batch + b] != 0)
        return solver_failed(i, b, host_i[(size_t)L * batch + (size_t)i * batch + b]);
      h_bonds_out[(int64_t)b * (L + 1) + i] = host_i[(size_t)i * batch + b];
      if (h_spectra && h_spec_offsets) {
        const int64_t room = h_spec_offsets[i + 1] - h_spec_offsets[i];
        double* dst = h_spectra + (int64_t)b * spec_total + h_spec_offsets[i];
        const int64_t have = std::min(room, spec_stride);
        for (int64_t t = 0; t < room; ++t)
          dst[t] = (t < have && host_s) ? host_s[((size_t)i * batch + b) * spec_stride + t] : 0.0;
      }
    }
  return NDMPS_OK;
}

// A sweep whose input is still intact (the fused one reads the volumes in place) repeats itself on the column
// launches when a resident tridiagonalisation gave up (NDMPS_ETEAM); the others hand the code to the caller, who
// owns the site-order copy the sweep has overwritten.
template <typename F>
int retry_without_team(F&& sweep) {
  int rc = sweep();
  if (rc != NDMPS_ETEAM) return rc;
  (void)ndmps_syevd_topk_note_team_fallback();
  const int was = ndmps_syevd_topk_set_team(0);
  rc = sweep();
  (void)ndmps_syevd_topk_set_team(was);
  return rc;
}

template <typename T>
int sweep_impl(int batch, T* const* h_dense, int L, const int64_t* h_dims, double cutoff, int64_t max_bond,
               T* const* h_cores, const int64_t* h_core_offsets, int64_t* h_bonds_out, double* h_spectra,
               const int64_t* h_spec_offsets, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream,
               const SweepSource* src = nullptr, const SweepAsync* async = nullptr) {
  NDMPS_REQUIRE(h_dense && h_dims && h_cores && h_core_offsets && h_bonds_out, "NULL sweep argument");
  NDMPS_REQUIRE(cutoff >= 0.0, "cutoff must be non-negative");
  SweepLayout lay;
  NDMPS_TRY(sweep_layout(L, h_dims, max_bond, batch, lay, sizeof(T) == 8 ? 8 : 4));
  if (d_ws == nullptr || ws_bytes < lay.workspace) {
    ndmps::set_error("sweep workspace too small: %lld < %lld", (long long)ws_bytes, (long long)lay.workspace);
    return NDMPS_EWORKSPACE;
  }
  for (int b = 0; b < batch; ++b) NDMPS_REQUIRE(h_dense[b] && h_cores[b], "NULL volume or core arena %d", b);
  hipStream_t s = (hipStream_t)stream;
  const int64_t sq = lay.small_max * lay.small_max;
  Arena ar(d_ws, ws_bytes);
  T* other = ar.take<T>((int64_t)batch * lay.numel);
  double* G = ar.take<double>((int64_t)batch * sq);
  double* V = ar.take<double>((int64_t)batch * sq);
  double* w = ar.take<double>((int64_t)batch * lay.small_max);
  double* sig = ar.take<double>((int64_t)batch * lay.small_max);
  const int64_t ev_ws_bytes = eig_workspace_bytes(lay.small_max, batch, max_bond);
  char* ev_ws = ar.take<char>(ev_ws_bytes);
  char* gram_ws = ar.take<char>(lay.gram_ws);
  double* A64 = ar.take<double>((int64_t)batch * lay.wide_elems);
  double* UtA = ar.take<double>((int64_t)batch * lay.wide_elems);
  NDMPS_REQUIRE(other && G && V && w && sig && ev_ws && gram_ws && A64 && UtA, "workspace carve failed");

  double* Graw = ar.take<double>((int64_t)batch * lay.merge_n * lay.merge_n);
  double* Tm = ar.take<double>((int64_t)batch * lay.merge_n * lay.merge_n);
  double* Wm[2] = {ar.take<double>((int64_t)batch * lay.merge_n * lay.merge_w),
                   ar.take<double>((int64_t)batch * lay.merge_n * lay.merge_w)};
  T* W32 = ar.take<T>((int64_t)batch * lay.merge_n * lay.merge_w);
  const int64_t tws_bytes = lay.transpose_bytes;
  char* tws = ar.take<char>(tws_bytes);
  int* d_ranks = ar.take<int>((int64_t)2 * L * batch);
  int* d_status = d_ranks ? d_ranks + (int64_t)L * batch : nullptr;
  double* d_spec = ar.take<double>((int64_t)L * batch * lay.spec_stride);
  const bool dev_rank = lay.device_rank;
  NDMPS_REQUIRE(d_ranks && (d_spec || lay.spec_stride == 0), "workspace carve failed");
  NDMPS_REQUIRE(Graw && Tm && Wm[0] && Wm[1] && W32 && tws, "workspace carve failed");

  std::vector<T*> cur(batch), nxt(batch);
  std::vector<int64_t> chi_r(batch, 1), cur_elems(batch, lay.numel), eig_n(batch), kept(batch);
  if (src) {
    // h_dense[b] is the C-order volume and stays untouched: the carried matrices ping-pong between the halves
    // of the workspace buffer (the first one is already <= half the tensor)
    NDMPS_REQUIRE(lay.merge_from < L && src->n_cols == lay.merge_n &&
                      (lay.numel / lay.merge_n) * lay.merge_w <= lay.numel / 2,
                  "the fused reshape stage needs a merged run of %lld columns (see ndmps_tt_merge_columns)",
                  (long long)src->n_cols);
  }
  for (int b = 0; b < batch; ++b) {
    cur[b] = h_dense[b];
    nxt[b] = other + (int64_t)b * lay.numel;
    h_bonds_out[(int64_t)b * (L + 1)] = 1;
    h_bonds_out[(int64_t)b * (L + 1) + L] = 1;
  }
  const int64_t spec_total = h_spec_offsets ? h_spec_offsets[L] : 0;
  std::vector<double> host_w((size_t)batch * lay.small_max);
  std::vector<int> eig_status;
  std::vector<int64_t> doubt;  // first eigenvalue in doubt per matrix (direct solver, every eigenpair wanted)

  // One batched eigen-solve for site i on the matrices G[b] (order eig_n[b], ld eig_n[b]): eigenvalues to the
  // host, rank decision per volume (kept[b]), then the kept eigenvectors in the columns of V[b].
  auto solve_site = [&](int i, bool have_a = false) -> int {
    int sweeps = 0;
    int64_t site_n = 0;
    for (int b = 0; b < batch; ++b) site_n = std::max(site_n, eig_n[b]);
    const bool topk = use_topk(site_n, max_bond);
    const int64_t k_cap = std::min<int64_t>(max_bond, site_n);
    if (dev_rank) {
      // no host round trip: eigenvalues -> rank (device) -> k_b eigenvectors, the other columns up to the cap zero
      NDMPS_TRY(ndmps_syevd_topk_values_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max, k_cap, ev_ws,
                                            ev_ws_bytes, s));
      NDMPS_TRY(ndmps_syevd_topk_vectors_auto_f64(batch, eig_n.data(), k_cap, std::max(cutoff, cutoff_floor<T>()),
                                                  d_ranks + (int64_t)i * batch, d_spec + (int64_t)i * batch * lay.spec_stride,
                                                  lay.spec_stride, d_status + (int64_t)i * batch, ev_ws, ev_ws_bytes, s));
      for (int b = 0; b < batch; ++b) kept[b] = k_cap;
      return NDMPS_OK;
    }
    bool full = !topk && use_direct_full(lay.small_max, batch, max_bond);  // every eigenpair above the cutoff, direct solver
    if (topk) {
      NDMPS_TRY(ndmps_syevd_topk_values_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max, k_cap, ev_ws,
                                            ev_ws_bytes, s));
      // the host waits for the eigenvalues anyway: a resident launch that gave up is redone on the column launches
      NDMPS_TRY(ndmps_syevd_topk_recover_f64(batch, eig_n.data(), k_cap, ev_ws, ev_ws_bytes, nullptr, s));
    } else if (full) {
      NDMPS_TRY(ndmps_syevd_topk_values_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max, lay.small_max, ev_ws,
                                            ev_ws_bytes, s));
      NDMPS_TRY(ndmps_syevd_topk_recover_f64(batch, eig_n.data(), lay.small_max, ev_ws, ev_ws_bytes, nullptr, s));
    }
    if (topk || full) {
      NDMPS_CHECK_HIP(hipMemcpyAsync(host_w.data(), w, sizeof(double) * batch * lay.small_max, hipMemcpyDeviceToHost, s));
      NDMPS_CHECK_HIP(hipStreamSynchronize(s));
      if (full) {
        // eigenvalues in doubt (direct_doubt_from): with the unfolding at hand their singular values are measured
        // as |A v| behind the solve (below); without it (merged run) the Jacobi decides (G is untouched)
        const double c = std::max(cutoff, cutoff_floor<T>());
        doubt.assign(batch, 0);
        for (int b = 0; b < batch; ++b) {
          doubt[b] = direct_doubt_from(host_w.data() + (int64_t)b * lay.small_max, eig_n[b], c);
          if (doubt[b] < eig_n[b] && !have_a) full = false;
        }
      }
    }
    if (!topk && !full) {
      NDMPS_TRY(ndmps_syevj_batched_values_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max, sweep_eig_tol(sizeof(T) == 8),
                                               ev_ws, ev_ws_bytes, &sweeps, s));
      NDMPS_CHECK_HIP(hipMemcpyAsync(host_w.data(), w, sizeof(double) * batch * lay.small_max, hipMemcpyDeviceToHost, s));
      NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    }
    for (int b = 0; b < batch; ++b) {
      const int64_t small = eig_n[b];
      std::vector<double> sv(host_w.begin() + (int64_t)b * lay.small_max,
                             host_w.begin() + (int64_t)b * lay.small_max + small);
      for (auto& x : sv) x = sqrt(std::max(x, 0.0));
      kept[b] = kept_rank(sv, cutoff, max_bond, cutoff_floor<T>());
      if (full && doubt[b] < small) kept[b] = small;  // every vector first; the rank follows from |A v| below
      if (h_spectra && h_spec_offsets) {
        // the layout reserves min(m, d_i max_bond_{i+1}) values for the bond; a merged site may be larger
        const int64_t room = h_spec_offsets[i + 1] - h_spec_offsets[i];
        memcpy(h_spectra + (int64_t)b * spec_total + h_spec_offsets[i], sv.data(),
               std::min(small, room) * sizeof(double));
      }
    }
    if (topk || full) {
      eig_status.assign(batch, 0);
      NDMPS_TRY(ndmps_syevd_topk_vectors_f64(batch, eig_n.data(), kept.data(), topk ? k_cap : lay.small_max, ev_ws, ev_ws_bytes,
                                             eig_status.data(), s));
      bool redo = false;
      for (int b = 0; b < batch; ++b)
        if (eig_status[b] != 0) {
          if (!full) return solver_failed(i, b, eig_status[b]);
          redo = true;  // the wide block lost rank (a cluster tighter than the shifts resolve): the Jacobi has no such case
        }
      if (redo) {
        NDMPS_TRY(ndmps_syevj_batched_values_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max,
                                                 sweep_eig_tol(sizeof(T) == 8), ev_ws, ev_ws_bytes, &sweeps, s));
        NDMPS_CHECK_HIP(hipMemcpyAsync(host_w.data(), w, sizeof(double) * batch * lay.small_max, hipMemcpyDeviceToHost, s));
        NDMPS_CHECK_HIP(hipStreamSynchronize(s));
        for (int b = 0; b < batch; ++b) {
          std::vector<double> sv(host_w.begin() + (int64_t)b * lay.small_max, host_w.begin() + (int64_t)b * lay.small_max + eig_n[b]);
          for (auto& x : sv) x = sqrt(std::max(x, 0.0));
          kept[b] = kept_rank(sv, cutoff, max_bond, cutoff_floor<T>());
        }
        NDMPS_TRY(ndmps_syevj_batched_vectors_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max, kept.data(), ev_ws,
                                                  ev_ws_bytes, s));
      } else if (full) {
        // ---- singular values in doubt, measured: s_j = |A v_j| (or |A^T u_j|) for the eigenvectors from doubt[b] on;
        //      G (free now) is the scratch.  The kept rank = the vectors in front + those whose s_j passes the cutoff.
        for (int b = 0; b < batch; ++b) {
          const int64_t small = eig_n[b], i0 = doubt[b], t = small - i0;
          if (t <= 0) continue;
          const int64_t n = h_dims[i] * chi_r[b], m = cur_elems[b] / n;
          const bool right = n <= m;  // G = A^T A: vectors in R^n, |A v|; else G = A A^T: |A^T u|
          const int rows = (int)(right ? m : n), cols = (int)(right ? n : m);
          const int nblk = (int)ceil_div(rows, 4);
          double* partial = G + (int64_t)b * sq;
          NDMPS_REQUIRE((int64_t)nblk * t + t <= sq, "internal: no room for the tail norms (%lld x %lld)", (long long)nblk, (long long)t);
          double* out = partial + (int64_t)nblk * t;
          hipLaunchKernelGGL(tail_norm_partial_kernel<T>, dim3((unsigned)nblk, (unsigned)ceil_div(t, 64)), dim3(256), 0, s,
                             (const T*)cur[b], right ? n : (int64_t)1, right ? (int64_t)1 : n, rows, cols,
                             (const double*)(V + (int64_t)b * sq), (int)small, (int)i0, (int)t, partial);
          hipLaunchKernelGGL(tail_norm_reduce_kernel, dim3((unsigned)ceil_div(t, 256)), dim3(256), 0, s, partial, nblk, (int)t, out);
          NDMPS_LAUNCH_CHECK();
          std::vector<double> s2((size_t)t);
          NDMPS_CHECK_HIP(hipMemcpyAsync(s2.data(), out, sizeof(double) * t, hipMemcpyDeviceToHost, s));
          NDMPS_CHECK_HIP(hipStreamSynchronize(s));
          const double s0 = sqrt(std::max(host_w[(int64_t)b * lay.small_max], 0.0));
          const double c = std::max(cutoff, cutoff_floor<T>());
          int64_t k = i0;
          for (int64_t j = 0; j < t; ++j) k += sqrt(std::max(s2[(size_t)j], 0.0)) > c * s0;
          k = std::max<int64_t>(k, 1);
          if (max_bond > 0) k = std::min(k, max_bond);
          kept[b] = std::min(k, small);
          if (h_spectra && h_spec_offsets) {
            const int64_t room = h_spec_offsets[i + 1] - h_spec_offsets[i];
            for (int64_t j = 0; j < t && i0 + j < room; ++j)
              h_spectra[(int64_t)b * spec_total + h_spec_offsets[i] + i0 + j] = sqrt(std::max(s2[(size_t)j], 0.0));
          }
        }
      }
    } else {
      NDMPS_TRY(ndmps_syevj_batched_vectors_f64(batch, G, sq, eig_n.data(), V, sq, w, lay.small_max, kept.data(),
                                                ev_ws, ev_ws_bytes, s));
    }
    return NDMPS_OK;
  };

  // every volume in the same state (always so when the ranks are decided on the device): the per-volume launches of
  // a site become one launch each
  auto uniform = [&]() {
    if (batch < 2) return false;
    for (int b = 1; b < batch; ++b)
      if (chi_r[b] != chi_r[0] || cur_elems[b] != cur_elems[0]) return false;
    return true;
  };
  auto uniform_kept = [&]() {
    for (int b = 1; b < batch; ++b)
      if (kept[b] != kept[0] || eig_n[b] != eig_n[0]) return false;
    return true;
  };

  int i_start = L - 1;
  if (lay.merge_from < L) {
    // ---- merged trailing run: sites merge_from .. L-1 from ONE Gram pass and ONE projection pass
    const int i0 = lay.merge_from;
    const int64_t n0 = lay.merge_n, ldw = lay.merge_w, m0 = lay.numel / n0;
    const int64_t stride_top = n0 * n0, stride_w = n0 * ldw;
    const int64_t raw_batched = gram_batched_need<T>(batch, m0, n0);
    if (batch > 1 && raw_batched > 0 && raw_batched <= lay.gram_ws) {
      // the whole group in one launch (long slabs: a fraction of the partial tiles, no launch gaps)
      if (src) {  // columns visited in memory order; the slab reduction stores the result in site order
        NDMPS_TRY(gram_batched_src(batch, cur.data(), m0, n0, *src, Graw, stride_top, gram_ws, lay.gram_ws, s));
      } else {
        NDMPS_TRY(gram_batched_T(batch, cur.data(), m0, n0, Graw, stride_top, gram_ws, lay.gram_ws, s));
      }
    } else {
      for (int b = 0; b < batch; ++b) {
        if (src) {
          NDMPS_TRY(gram_src(cur[b], m0, n0, *src, Graw + (int64_t)b * stride_top, gram_ws, lay.gram_ws, s));
        } else {
          NDMPS_TRY(gram_T(cur[b], m0, n0, n0, Graw + (int64_t)b * stride_top, gram_ws, lay.gram_ws, s));
        }
      }
    }
    hipLaunchKernelGGL(merge_basis_init_kernel, dim3(batch), dim3(1), 0, s, Wm[0], stride_w);
    NDMPS_LAUNCH_CHECK();
    int wcur = 0;
    int64_t n_right = 1;  // N_{i+1}
    for (int i = L - 1; i >= i0; --i) {
      const int64_t d_i = h_dims[i], n_i = n_right * d_i;
      // the top site of the run (its raw Gram IS the one computed: no block sums) of a uniform group: both
      // congruence stages as batched fp64 products on the MFMA (0.3 + 0.3 ms of scalar loops per group otherwise,
      // right behind the Gram pass on the critical path)
      bool on_mfma = false;
      if (n_i == n0 && uniform() && batch <= ndmps_gemm_batched_max() && n_i * chi_r[0] >= 4096) {
        const int64_t kr = chi_r[0], n = d_i * kr;
        std::vector<const double*> pa(batch), pb(batch);
        std::vector<double*> pc(batch);
        for (int b = 0; b < batch; ++b) {
          pa[b] = Graw + (int64_t)b * stride_top;
          pb[b] = Wm[wcur] + (int64_t)b * stride_w;
          pc[b] = Tm + (int64_t)b * stride_top;
          eig_n[b] = n;
        }
        // T[(r, blk)][q] = sum_c Graw[(r, blk)][c] W[c][q]: rows (r, blk) of n_right contiguous elements
        NDMPS_TRY(ndmps_dgemm_batched(batch, 0, 0, n_i * d_i, kr, n_right, pa.data(), n_right, pb.data(), ldw, pc.data(), kr, s));
        // G[(a, p)][j] = sum_c W[c][p] T[(a n_right + c)][j]: one product per (volume, a)
        const int per = ndmps_gemm_batched_max();
        std::vector<const double*> qa, qb;
        std::vector<double*> qc;
        for (int b = 0; b < batch; ++b)
          for (int64_t a = 0; a < d_i; ++a) {
            qa.push_back(Wm[wcur] + (int64_t)b * stride_w);
            qb.push_back(Tm + (int64_t)b * stride_top + a * n_right * n);
            qc.push_back(G + (int64_t)b * sq + a * kr * n);
          }
        for (size_t base = 0; base < qa.size(); base += per) {
          const int count = (int)std::min<size_t>(per, qa.size() - base);
          NDMPS_TRY(ndmps_dgemm_batched(count, 1, 0, kr, n, n_right, qa.data() + base, ldw, qb.data() + base, n,
                                        qc.data() + base, n, s));
        }
        on_mfma = true;
      }
      for (int base = 0; base < batch && !on_mfma; base += 64) {
        const int count = std::min(64, batch - base);
        MergeRanks rk;
        int64_t biggest = 1;
        for (int t = 0; t < count; ++t) {
          rk.k_right[t] = (int)chi_r[base + t];
          rk.k_here[t] = 0;
          eig_n[base + t] = d_i * chi_r[base + t];
          biggest = std::max(biggest, eig_n[base + t]);
        }
        const int g1 = (int)std::min<int64_t>(ceil_div(n_i * biggest, 256), 1024);
        const int g2 = (int)std::min<int64_t>(ceil_div(biggest * biggest, 256), 1024);
        hipLaunchKernelGGL(merge_stage1_kernel, dim3(g1, count), dim3(256), 0, s, Graw + base * stride_top, stride_top,
                           (int)n0, Wm[wcur] + base * stride_w, stride_w, (int)ldw, Tm + base * stride_top, stride_top,
                           (int)n_i, (int)n_right, (int)d_i, rk);
        hipLaunchKernelGGL(merge_stage2_kernel, dim3(g2, count), dim3(256), 0, s, Tm + base * stride_top, stride_top,
                           Wm[wcur] + base * stride_w, stride_w, (int)ldw, G + base * sq, sq, (int)n_right, (int)d_i, rk);
      }
      NDMPS_LAUNCH_CHECK();
      NDMPS_TRY(solve_site(i));
      for (int base = 0; base < batch; base += 64) {
        const int count = std::min(64, batch - base);
        MergeRanks rk;
        int64_t biggest = 1;
        for (int t = 0; t < count; ++t) {
          rk.k_right[t] = (int)chi_r[base + t];
          rk.k_here[t] = (int)kept[base + t];
          biggest = std::max(biggest, kept[base + t]);
        }
        const int g3 = (int)std::min<int64_t>(ceil_div(n_i * biggest, 256), 1024);
        hipLaunchKernelGGL(merge_basis_kernel<T>, dim3(g3, count), dim3(256), 0, s, Wm[wcur] + base * stride_w, stride_w,
                           (int)ldw, V + base * sq, sq, Wm[wcur ^ 1] + base * stride_w,
                           i == i0 ? W32 + base * stride_w : (T*)nullptr, stride_w, (int)n_right, (int)d_i, rk);
      }
      NDMPS_LAUNCH_CHECK();
      if (batch > 1 && uniform_kept()) {
        for (int base = 0; base < batch; base += kSmallBatch) {
          const int count = std::min(kSmallBatch, batch - base);
          BatchOps ops;
          for (int t = 0; t < count; ++t) ops.out[t] = h_cores[base + t] + h_core_offsets[i];
          hipLaunchKernelGGL(core_from_vectors_batched_kernel<T>, dim3(grid1d(kept[0] * eig_n[0]), count), dim3(256), 0, s,
                             V + (int64_t)base * sq, sq, eig_n[0], kept[0], ops);
        }
      } else {
        for (int b = 0; b < batch; ++b)
          hipLaunchKernelGGL(core_from_vectors_kernel<T>, dim3(grid1d(kept[b] * eig_n[b])), dim3(256), 0, s,
                             V + (int64_t)b * sq, eig_n[b], kept[b], h_cores[b] + h_core_offsets[i]);
      }
      for (int b = 0; b < batch; ++b) {
        chi_r[b] = kept[b];
        h_bonds_out[(int64_t)b * (L + 1) + i] = kept[b];
      }
      NDMPS_LAUNCH_CHECK();
      wcur ^= 1;
      n_right = n_i;
    }
    bool projected = false;
    if (src && uniform() && batch <= ndmps_gemm_batched_max()) {
      // carry = A_raw W for the whole group: the basis rows into memory order (one launch), then one batched product
      // that reads the volumes through the permutation tables
      const int64_t k = chi_r[0];
      float* wperm0 = reinterpret_cast<float*>(Tm);
      const int64_t wperm_stride = stride_top * 2;  // fp64 slots of T, in floats
      hipLaunchKernelGGL(gather_rows_kernel, dim3(grid1d(n0 * k), batch), dim3(256), 0, s, (const float*)W32, n0, k,
                         src->col_perm, wperm0, stride_w, wperm_stride);
      NDMPS_LAUNCH_CHECK();
      std::vector<const float*> pa(batch), pb(batch);
      std::vector<float*> pc(batch);
      for (int b = 0; b < batch; ++b) {
        pa[b] = (const float*)cur[b];
        pb[b] = wperm0 + (int64_t)b * wperm_stride;
        pc[b] = (float*)nxt[b];
      }
      // 64 gathered columns (a bond cap of 32): the stream over the rows in memory order; anything else: the tile kernel
      if (n0 == 64 && (k == 32 || k == 64) && src->row_order && src->row_sorted != src->row_off && !getenv("NDMPS_PROJ64_TILES"))
        NDMPS_TRY(ndmps_sgemm_gathered64_stream_batched(batch, m0, k, pa.data(), src->row_sorted, src->row_order, src->col_off,
                                                        pb.data(), k, pc.data(), k, s));
      else
      NDMPS_TRY(ndmps_sgemm_indexed_batched(batch, m0, k, n0, pa.data(), 0, src->row_off, src->col_off, 1, pb.data(), k,
                                            pc.data(), k, nullptr, nullptr, s));
      for (int b = 0; b < batch; ++b) {
        cur[b] = nxt[b];
        nxt[b] = nxt[b] + lay.numel / 2;
        cur_elems[b] = m0 * k;
      }
      projected = true;
    }
    for (int b = 0; b < batch && !projected; ++b) {  // carry = A_raw W (m0 x k)
      const int64_t k = chi_r[b];
      if (src) {
        // T (fp64 scratch of the congruences, free now) holds the basis with its rows in memory order
        T* wperm = reinterpret_cast<T*>(Tm + (int64_t)b * stride_top);
        NDMPS_TRY(project_src(cur[b], m0, k, n0, *src, W32 + (int64_t)b * stride_w, wperm, nxt[b], s));
        cur[b] = nxt[b];
        nxt[b] = nxt[b] + lay.numel / 2;
      } else {
        NDMPS_TRY(gemm_T(0, m0, k, n0, cur[b], W32 + (int64_t)b * stride_w, k, nxt[b], tws, tws_bytes, s));
        std::swap(cur[b], nxt[b]);
      }
      cur_elems[b] = m0 * k;
    }
    i_start = i0 - 1;
  }

  for (int i = i_start; i >= 1; --i) {
    // ---- small-side Gram matrices
    int64_t m = 0;
    // same shape in every volume (always so when the ranks are decided on the device): one Gram launch
    bool together = batch > 1;
    for (int b = 1; b < batch && together; ++b) together = chi_r[b] == chi_r[0] && cur_elems[b] == cur_elems[0];
    if (together) {
      const int64_t n = h_dims[i] * chi_r[0];
      m = cur_elems[0] / n;
      const int64_t need = n <= m ? gram_batched_need<T>(batch, m, n) : 0;
      together = need > 0 && need <= lay.gram_ws && n * n <= sq;
      if (together) {
        for (int b = 0; b < batch; ++b) eig_n[b] = n;
        NDMPS_TRY(gram_batched_T(batch, cur.data(), m, n, G, sq, gram_ws, lay.gram_ws, s));
      }
    }
    // wide unfoldings (n > m, the last sites) of a uniform group: A A^T of every volume from two launches
    const bool uni = uniform() && batch <= std::min(kSmallBatch, ndmps_gemm_batched_max());
    bool wide_together = false;
    if (!together && uni) {
      const int64_t n = h_dims[i] * chi_r[0];
      m = cur_elems[0] / n;
      if (n > m) {
        BatchOps ops;
        std::vector<const double*> pa(batch);
        std::vector<double*> pc(batch);
        for (int b = 0; b < batch; ++b) {
          ops.in[b] = cur[b];
          pa[b] = A64 + (int64_t)b * lay.wide_elems;
          pc[b] = G + (int64_t)b * sq;
          eig_n[b] = m;
        }
        hipLaunchKernelGGL(f32_to_f64_batched_kernel<T>, dim3(grid1d(m * n), batch), dim3(256), 0, s, ops, m * n, A64,
                           lay.wide_elems);
        NDMPS_LAUNCH_CHECK();
        NDMPS_TRY(ndmps_dgemm_batched(batch, 0, 1, m, m, n, pa.data(), n, pa.data(), n, pc.data(), m, s));
        wide_together = true;
      }
    }
    for (int b = 0; b < batch && !together && !wide_together; ++b) {
      const int64_t n = h_dims[i] * chi_r[b];
      m = cur_elems[b] / n;
      eig_n[b] = std::min(m, n);
      double* Gb = G + (int64_t)b * sq;
      if (n <= m) {
        const int64_t need = gram_need(cur[b], m, n);
        NDMPS_REQUIRE(need <= lay.gram_ws, "internal: Gram workspace bound violated (%lld > %lld)",
                      (long long)need, (long long)lay.gram_ws);
        NDMPS_TRY(gram_T(cur[b], m, n, n, Gb, gram_ws, lay.gram_ws, s));
      } else {
        double* Ab = A64 + (int64_t)b * lay.wide_elems;
        hipLaunchKernelGGL(f32_to_f64_kernel<T>, dim3(grid1d(m * n)), dim3(256), 0, s, cur[b], m * n, Ab);
        NDMPS_LAUNCH_CHECK();
        NDMPS_TRY(ndmps_dgemm(0, 1, m, m, n, Ab, n, Ab, n, Gb, m, s));
      }
    }
    NDMPS_TRY(solve_site(i, true));
    // ---- core and carried matrix: one launch per step for a uniform group, else volume by volume
    bool done = false;
    if (uni && uniform_kept()) {
      const int64_t n = h_dims[i] * chi_r[0], small = eig_n[0], k = kept[0];
      BatchOps cores_out, carry_out;
      std::vector<T*> pcur(batch), pcore(batch), pnxt(batch);
      for (int b = 0; b < batch; ++b) {
        pcur[b] = cur[b];
        pcore[b] = h_cores[b] + h_core_offsets[i];
        pnxt[b] = nxt[b];
        cores_out.out[b] = pcore[b];
        carry_out.out[b] = nxt[b];
      }
      if (n <= m) {
        int rc = NDMPS_OK;
        hipLaunchKernelGGL(core_from_vectors_batched_kernel<T>, dim3(grid1d(k * n), batch), dim3(256), 0, s, V, sq, n, k,
                           cores_out);
        NDMPS_LAUNCH_CHECK();
        done = gemm_batched_T(batch, 1, m, k, n, pcur.data(), pcore.data(), n, pnxt.data(), s, &rc);
        NDMPS_TRY(rc);
        if (!done)  // bf16 storage: the products go volume by volume
          for (int b = 0; b < batch; ++b) NDMPS_TRY(gemm_T(1, m, k, n, cur[b], pcore[b], n, nxt[b], tws, tws_bytes, s));
        done = true;
      } else {
        std::vector<const double*> pv(batch), pa(batch);
        std::vector<double*> pu(batch);
        for (int b = 0; b < batch; ++b) {
          pv[b] = V + (int64_t)b * sq;
          pa[b] = A64 + (int64_t)b * lay.wide_elems;
          pu[b] = UtA + (int64_t)b * lay.wide_elems;
        }
        hipLaunchKernelGGL(sqrt_clamp_batched_kernel, dim3(grid1d(small), batch), dim3(256), 0, s, w, lay.small_max, small,
                           sig);
        hipLaunchKernelGGL(scale_cols_to_f32_batched_kernel<T>, dim3(grid1d(m * k), batch), dim3(256), 0, s, V, sq, m, m, k,
                           sig, lay.small_max, 1.0, carry_out);  // carry = U_k diag(sigma_k)
        NDMPS_LAUNCH_CHECK();
        NDMPS_TRY(ndmps_dgemm_batched(batch, 1, 0, k, n, m, pv.data(), m, pa.data(), n, pu.data(), n, s));
        hipLaunchKernelGGL(scale_rows_to_f32_batched_kernel<T>, dim3(grid1d(k * n), batch), dim3(256), 0, s, UtA,
                           lay.wide_elems, k, n, sig, lay.small_max, -1.0, cores_out);  // core = diag(1/sigma_k) U_k^T A
        NDMPS_LAUNCH_CHECK();
        done = true;
      }
      for (int b = 0; b < batch; ++b) {
        std::swap(cur[b], nxt[b]);
        cur_elems[b] = m * k;
        chi_r[b] = k;
        h_bonds_out[(int64_t)b * (L + 1) + i] = k;
      }
    }
    for (int b = 0; b < batch && !done; ++b) {
      const int64_t n = h_dims[i] * chi_r[b];
      m = cur_elems[b] / n;
      const int64_t small = eig_n[b];
      const int64_t k = kept[b];
      T* core = h_cores[b] + h_core_offsets[i];
      double* Vb = V + (int64_t)b * sq;
      double* wb = w + (int64_t)b * lay.small_max;
      double* sigb = sig + (int64_t)b * lay.small_max;
      if (n <= m) {
        hipLaunchKernelGGL(core_from_vectors_kernel<T>, dim3(grid1d(k * n)), dim3(256), 0, s, Vb, n, k, core);
        NDMPS_LAUNCH_CHECK();
        NDMPS_TRY(gemm_T(1, m, k, n, cur[b], core, n, nxt[b], tws, tws_bytes, s));
      } else {
        double* Ab = A64 + (int64_t)b * lay.wide_elems;
        hipLaunchKernelGGL(sqrt_clamp_kernel, dim3(grid1d(small)), dim3(256), 0, s, wb, small, sigb);
        hipLaunchKernelGGL(scale_cols_to_f32_kernel<T>, dim3(grid1d(m * k)), dim3(256), 0, s, Vb, m, m, k, sigb, 1.0,
                           nxt[b]);  // carry = U_k diag(sigma_k)
        NDMPS_LAUNCH_CHECK();
        NDMPS_TRY(ndmps_dgemm(1, 0, k, n, m, Vb, m, Ab, n, UtA, n, s));
        hipLaunchKernelGGL(scale_rows_to_f32_kernel<T>, dim3(grid1d(k * n)), dim3(256), 0, s, UtA, k, n, sigb, -1.0,
                           core);  // core = diag(1/sigma_k) U_k^T A
        NDMPS_LAUNCH_CHECK();
      }
      std::swap(cur[b], nxt[b]);
      cur_elems[b] = m * k;
      chi_r[b] = k;
      h_bonds_out[(int64_t)b * (L + 1) + i] = k;
    }
  }
  // site 0 carries the norm: (1, d_0, chi_1)
  for (int b = 0; b < batch; ++b)
    NDMPS_CHECK_HIP(hipMemcpyAsync(h_cores[b] + h_core_offsets[0], cur[b], cur_elems[b] * sizeof(T),
                                   hipMemcpyDeviceToDevice, s));
  if (dev_rank && L > 1) {
    const size_t n_i = (size_t)2 * L * batch, n_s = (size_t)L * batch * lay.spec_stride;
    if (async) {  // enqueue only: the caller reads the pinned buffers behind its own synchronisation (sweep_collect)
      NDMPS_REQUIRE(async->h_ranks && (async->h_spec || n_s == 0), "asynchronous sweep without its host buffers");
      NDMPS_CHECK_HIP(hipMemcpyAsync(async->h_ranks, d_ranks, n_i * sizeof(int), hipMemcpyDeviceToHost, s));
      if (n_s) NDMPS_CHECK_HIP(hipMemcpyAsync(async->h_spec, d_spec, n_s * sizeof(double), hipMemcpyDeviceToHost, s));
      return NDMPS_OK;
    }
    std::vector<int> host_i(n_i);
    std::vector<double> host_s(n_s);
    NDMPS_CHECK_HIP(hipMemcpyAsync(host_i.data(), d_ranks, n_i * sizeof(int), hipMemcpyDeviceToHost, s));
    if (n_s) NDMPS_CHECK_HIP(hipMemcpyAsync(host_s.data(), d_spec, n_s * sizeof(double), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    return sweep_collect(batch, L, lay.spec_stride, host_i.data(), n_s ? host_s.data() : nullptr, h_bonds_out, h_spectra,
                         h_spec_offsets);
  }
  if (async) {
    ndmps::set_error("an asynchronous sweep needs ranks decided on the device (ndmps_tt_sweep_pads_cores) and more than one site");
    return NDMPS_EINVAL;
  }
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
}  // namespace

// 1 when the sweep for these site dims and this bond cap decides ranks on the device: cores are then written at
// the layout's offsets in PADDED shape (max_bonds[i], d_i, max_bonds[i+1]) with zeros beyond the actual bonds
// (h_bonds_out), and the caller slices them; 0: cores are compact (bonds[i], d_i, bonds[i+1]).
extern "C" int ndmps_tt_sweep_pads_cores(int L, const int64_t* h_dims, int64_t max_bond) {
  SweepLayout lay;
  if (!h_dims || sweep_layout(L, h_dims, max_bond, 1, lay) != NDMPS_OK) return 0;
  return lay.device_rank ? 1 : 0;
}

extern "C" int ndmps_tt_sweep_batched_f32(int batch, float* const* h_dense, int L, const int64_t* h_dims,
                                          double cutoff, int64_t max_bond, float* const* h_cores,
                                          const int64_t* h_core_offsets, int64_t* h_bonds_out,
                                          double* h_spectra, const int64_t* h_spec_offsets, void* d_ws,
                                          int64_t ws_bytes, ndmps_stream_t stream) {
  return sweep_impl<float>(batch, h_dense, L, h_dims, cutoff, max_bond, h_cores, h_core_offsets, h_bonds_out, h_spectra,
                           h_spec_offsets, d_ws, ws_bytes, stream);
}

// Order of the raw Gram matrix of the merged trailing run when the reshape stage can ride on it (0 otherwise):
// the caller passes ndmps_plan_split_offsets tables for that many columns to the fused sweep.
extern "C" int64_t ndmps_tt_merge_columns(int L, const int64_t* h_dims, int64_t max_bond) {
  SweepLayout lay;
  if (!h_dims || sweep_layout(L, h_dims, max_bond, 1, lay) != NDMPS_OK) return 0;
  if (lay.merge_from >= L || (lay.numel / lay.merge_n) * lay.merge_w > lay.numel / 2) return 0;
  if (lay.merge_n < 64 || lay.numel / lay.merge_n < 256 || lay.merge_n % 4 != 0) return 0;  // wide Gram path only
  return lay.merge_n;
}

// The fp32 sweep reading the C-order VOLUMES through the index permutation (core/ndmps.py:66-71 fused into the
// first Gram pass and the first projection): h_volume[b] are left untouched, no site-order tensor is formed.
// Tables: ndmps_plan_split_offsets for n_cols = ndmps_tt_merge_columns(...), columns sorted by offset
// (d_col_off ascending, in aligned runs of four consecutive offsets; d_col_perm[c] = site-order column).
extern "C" int ndmps_tt_sweep_batched_fused_f32(int batch, const float* const* h_volume, int L, const int64_t* h_dims,
                                                double cutoff, int64_t max_bond, float* const* h_cores,
                                                const int64_t* h_core_offsets, int64_t* h_bonds_out,
                                                double* h_spectra, const int64_t* h_spec_offsets,
                                                const int64_t* d_row_off, const int64_t* d_row_off_sorted,
                                                const int32_t* d_row_order, const int64_t* d_col_off,
                                                const int32_t* d_col_perm, int64_t n_cols, void* d_ws, int64_t ws_bytes,
                                                ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_row_off && d_col_off && d_col_perm && n_cols >= 1, "NULL permutation table");
  SweepSource src{d_row_off, d_row_off_sorted ? d_row_off_sorted : d_row_off, d_row_off_sorted ? d_row_order : nullptr, d_col_off,
                  d_col_perm, n_cols};
  return retry_without_team([&]() {
    return sweep_impl<float>(batch, (float* const*)h_volume, L, h_dims, cutoff, max_bond, h_cores, h_core_offsets,
                             h_bonds_out, h_spectra, h_spec_offsets, d_ws, ws_bytes, stream, &src);
  });
}

// The fused sweep in two halves (device-side ranks only: ndmps_tt_sweep_pads_cores; see SweepAsync above).
//   ..._begin   enqueues the whole sweep and the copies of ranks / status / spectra into the caller's PINNED host buffers
//               (ndmps_tt_sweep_async_ints / _doubles elements); returns without waiting.  h_bonds_scratch: batch (L + 1).
//   ndmps_tt_sweep_finish  once the caller has synchronised with the stream (an event behind _begin): bonds, spectra, or
//               the solver's error -- NDMPS_ETEAM when a resident tridiagonalisation gave up: nothing is repeated here,
//               the caller redoes the batch with ndmps_tt_sweep_batched_fused_f32 (which retries on the column launches).
// Reference: the same from_dense of core/ndmps.py:74; the reference has no counterpart of the split (NumPy is synchronous).
extern "C" int64_t ndmps_tt_sweep_async_ints(int batch, int L) { return (int64_t)2 * L * batch; }
extern "C" int64_t ndmps_tt_sweep_async_doubles(int batch, int L, const int64_t* h_dims, int64_t max_bond) {
  SweepLayout lay;
  if (!h_dims || sweep_layout(L, h_dims, max_bond, batch, lay) != NDMPS_OK) return -1;
  return (int64_t)L * batch * lay.spec_stride;
}
extern "C" int ndmps_tt_sweep_batched_fused_begin_f32(int batch, const float* const* h_volume, int L, const int64_t* h_dims,
                                                      double cutoff, int64_t max_bond, float* const* h_cores,
                                                      const int64_t* h_core_offsets, int64_t* h_bonds_scratch,
                                                      const int64_t* d_row_off, const int64_t* d_row_off_sorted,
                                                      const int32_t* d_row_order, const int64_t* d_col_off,
                                                      const int32_t* d_col_perm, int64_t n_cols, void* d_ws, int64_t ws_bytes,
                                                      int* h_pinned_ranks, double* h_pinned_spec, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_row_off && d_col_off && d_col_perm && n_cols >= 1, "NULL permutation table");
  NDMPS_REQUIRE(h_pinned_ranks != nullptr, "NULL host buffer");
  SweepSource src{d_row_off, d_row_off_sorted ? d_row_off_sorted : d_row_off, d_row_off_sorted ? d_row_order : nullptr, d_col_off,
                  d_col_perm, n_cols};
  const SweepAsync as{h_pinned_ranks, h_pinned_spec};
  return sweep_impl<float>(batch, (float* const*)h_volume, L, h_dims, cutoff, max_bond, h_cores, h_core_offsets,
                           h_bonds_scratch, nullptr, nullptr, d_ws, ws_bytes, stream, &src, &as);
}
extern "C" int ndmps_tt_sweep_finish(int batch, int L, const int64_t* h_dims, int64_t max_bond, const int* h_pinned_ranks,
                                     const double* h_pinned_spec, int64_t* h_bonds_out, double* h_spectra,
                                     const int64_t* h_spec_offsets) {
  NDMPS_REQUIRE(h_dims && h_pinned_ranks && h_bonds_out && batch >= 1 && L >= 2, "bad sweep_finish argument");
  SweepLayout lay;
  NDMPS_TRY(sweep_layout(L, h_dims, max_bond, batch, lay));
  NDMPS_REQUIRE(lay.device_rank, "this layout decides its ranks on the host: there is nothing to finish");
  return sweep_collect(batch, L, lay.spec_stride, h_pinned_ranks, lay.spec_stride ? h_pinned_spec : nullptr, h_bonds_out,
                       h_spectra, h_spec_offsets);
}

// bf16 storage: the site-order tensors, the carried matrices and the cores are bf16 in HBM; Gram matrices,
// eigen-decompositions and bases stay fp64, products accumulate in fp32 on the bf16 MFMA.  Same layout and
// workspace queries as the fp32 sweep (offsets in elements; the fp32 workspace size is an upper bound).
extern "C" int ndmps_tt_sweep_batched_bf16(int batch, void* const* h_dense, int L, const int64_t* h_dims,
                                           double cutoff, int64_t max_bond, void* const* h_cores,
                                           const int64_t* h_core_offsets, int64_t* h_bonds_out,
                                           double* h_spectra, const int64_t* h_spec_offsets, void* d_ws,
                                           int64_t ws_bytes, ndmps_stream_t stream) {
  return sweep_impl<__bf16>(batch, (__bf16* const*)h_dense, L, h_dims, cutoff, max_bond, (__bf16* const*)h_cores,
                            h_core_offsets, h_bonds_out, h_spectra, h_spec_offsets, d_ws, ws_bytes, stream);
}

// fp64 storage, the reference's own element type (core/ndmps.py:56): volume / site-order tensor, carried matrices and
// cores are fp64 in HBM, every product runs on the fp64 MFMA (ndmps_dgemm), Gram matrices and eigen-decompositions
// are fp64 as in the other storage types.  Workspace: ndmps_tt_sweep_batched_workspace_bytes_f64; layout offsets
// (ndmps_tt_layout) are in elements and shared with the other storage types.  The relative cutoff is clamped below
// at 1e-8 (singular values come from fp64 Gram matrices: sqrt(eps) s_0 is what they resolve).
extern "C" int ndmps_tt_sweep_batched_f64(int batch, double* const* h_dense, int L, const int64_t* h_dims,
                                          double cutoff, int64_t max_bond, double* const* h_cores,
                                          const int64_t* h_core_offsets, int64_t* h_bonds_out,
                                          double* h_spectra, const int64_t* h_spec_offsets, void* d_ws,
                                          int64_t ws_bytes, ndmps_stream_t stream) {
  return sweep_impl<double>(batch, h_dense, L, h_dims, cutoff, max_bond, h_cores, h_core_offsets, h_bonds_out, h_spectra,
                            h_spec_offsets, d_ws, ws_bytes, stream);
}

extern "C" int ndmps_tt_sweep_f32(float* d_dense, int L, const int64_t* h_dims, double cutoff,
                                  int64_t max_bond, float* d_cores, const int64_t* h_core_offsets,
                                  int64_t* h_bonds_out, double* h_spectra, const int64_t* h_spec_offsets,
                                  void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_dense && d_cores, "NULL sweep argument");
  float* dense[1] = {d_dense};
  float* cores[1] = {d_cores};
  return ndmps_tt_sweep_batched_f32(1, dense, L, h_dims, cutoff, max_bond, cores, h_core_offsets, h_bonds_out,
                                    h_spectra, h_spec_offsets, d_ws, ws_bytes, stream);
}

// =================================================================== bond truncation
namespace {
struct BondLayout {
  int64_t off[16];
  int64_t total;
};
}  // namespace

namespace {
// eigen workspace of compress_bond: the block Jacobi's, and the direct solver's for every eigenpair where it applies
inline int64_t bond_eig_bytes(int64_t chi) {
  const int64_t jac = ndmps_syevj_workspace_bytes(chi);
  return use_direct_full(chi, 1, 0) ? std::max(jac, ndmps_syevd_topk_workspace_bytes(chi, 1, chi)) : jac;
}
}  // namespace

extern "C" int64_t ndmps_compress_bond_workspace_bytes(int64_t chi_l, int64_t d1, int64_t chi, int64_t d2,
                                                       int64_t chi_r) {
  if (chi_l <= 0 || d1 <= 0 || chi <= 0 || d2 <= 0 || chi_r <= 0) return 0;
  const int64_t m1 = chi_l * d1, n2 = d2 * chi_r, c2 = chi * chi;
  int64_t used = 0;
  used = arena_bytes(used, 8, c2);       // G1
  used = arena_bytes(used, 8, c2);       // G2 -> destroyed
  used = arena_bytes(used, 8, c2);       // W2 / Ltilde
  used = arena_bytes(used, 8, c2);       // tmp = G1 Ltilde
  used = arena_bytes(used, 8, c2);       // H
  used = arena_bytes(used, 8, c2);       // V
  used = arena_bytes(used, 8, c2);       // P1 = Ltilde V
  used = arena_bytes(used, 8, c2);       // P2 = tmp V
  used = arena_bytes(used, 8, chi);      // w2
  used = arena_bytes(used, 8, chi);      // wh
  used = arena_bytes(used, 8, chi);      // sigma
  used = arena_bytes(used, 8, c2);       // A1 in the storage type
  used = arena_bytes(used, 8, c2);       // B2 in the storage type
  used = arena_bytes(used, 8, chi * n2); // t2 in fp64
  used = arena_bytes(used, 1, bond_eig_bytes(chi));
  used = arena_bytes(used, 1, std::max(ndmps_gram_workspace_bytes(m1, chi), ndmps_gram_f64_workspace_bytes(m1, chi)));
  return ndmps::round_up(used, 256) + 256;
}

// Truncated SVD of the two-site product P = T1 T2 through the bond, without forming P or
// Q factors: with G1 = T1^T T1, G2 = T2 T2^T = W D W^T, Lt = W D^(1/2) and H = Lt^T G1 Lt
// = V diag(s^2) V^T (the s are the singular values of P), the absorb-"both" cores are
//   T1' = T1 (Lt V_k) s_k^(-1/2),   T2' = s_k^(-3/2) (G1 Lt V_k)^T T2 .
namespace {
template <typename T>
int compress_bond_impl(const T* d_t1, const T* d_t2, int64_t chi_l, int64_t d1, int64_t chi, int64_t d2, int64_t chi_r,
                       double cutoff, int64_t max_bond, T* d_new1, T* d_new2, int64_t* h_new_chi, double* h_s, void* d_ws,
                       int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_t1 && d_t2 && d_new1 && d_new2 && h_new_chi, "NULL compress_bond argument");
  NDMPS_REQUIRE(chi_l > 0 && d1 > 0 && chi > 0 && d2 > 0 && chi_r > 0, "bad core shape");
  NDMPS_REQUIRE(cutoff >= 0.0, "cutoff must be non-negative");
  const int64_t need = ndmps_compress_bond_workspace_bytes(chi_l, d1, chi, d2, chi_r);
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("compress_bond workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t m1 = chi_l * d1, n2 = d2 * chi_r, c2 = chi * chi;
  Arena ar(d_ws, ws_bytes);
  double* G1 = ar.take<double>(c2);
  double* G2 = ar.take<double>(c2);
  double* Lt = ar.take<double>(c2);
  double* tmp = ar.take<double>(c2);
  double* H = ar.take<double>(c2);
  double* V = ar.take<double>(c2);
  double* P1 = ar.take<double>(c2);
  double* P2 = ar.take<double>(c2);
  double* w2 = ar.take<double>(chi);
  double* wh = ar.take<double>(chi);
  double* sig = ar.take<double>(chi);
  T* A1 = reinterpret_cast<T*>(ar.take<double>(c2));
  T* B2 = reinterpret_cast<T*>(ar.take<double>(c2));
  double* t2d = ar.take<double>(chi * n2);
  const int64_t ev_bytes = bond_eig_bytes(chi);
  char* ev_ws = ar.take<char>(ev_bytes);
  const int64_t gram_bytes = std::max(ndmps_gram_workspace_bytes(m1, chi), ndmps_gram_f64_workspace_bytes(m1, chi));
  char* gram_ws = ar.take<char>(gram_bytes);
  NDMPS_REQUIRE(G1 && G2 && Lt && tmp && H && V && P1 && P2 && w2 && wh && sig && A1 && B2 && t2d && ev_ws &&
                    gram_ws,
                "workspace carve failed");

  int sweeps = 0;
  NDMPS_TRY(gram_T(d_t1, m1, chi, chi, G1, gram_ws, gram_bytes, s));
  hipLaunchKernelGGL(f32_to_f64_kernel<T>, dim3(grid1d(chi * n2)), dim3(256), 0, s, d_t2, chi * n2, t2d);
  NDMPS_LAUNCH_CHECK();
  NDMPS_TRY(ndmps_dgemm(0, 1, chi, chi, n2, t2d, n2, t2d, n2, G2, chi, s));
  // Both decompositions on the direct solver where it applies (every eigenpair of G2 for its square root; of H the
  // eigenvalues, then only the kept vectors), the block Jacobi otherwise -- and for H whenever the solver's own noise
  // could move the rank (direct_rank_is_safe).
  const bool direct = use_direct_full(chi, 1, 0);
  const int64_t n1[1] = {chi};
  auto direct_values = [&](const double* M, double* vecs, double* vals) -> int {
    NDMPS_TRY(ndmps_syevd_topk_values_f64(1, M, c2, n1, vecs, c2, vals, chi, chi, ev_ws, ev_bytes, s));
    return ndmps_syevd_topk_recover_f64(1, n1, chi, ev_ws, ev_bytes, nullptr, s);
  };
  auto direct_vectors = [&](int64_t kv, bool& ok) -> int {  // ok = false: the block lost rank, the caller takes the Jacobi
    const int64_t k1[1] = {kv};
    int status = 0;
    NDMPS_TRY(ndmps_syevd_topk_vectors_f64(1, n1, k1, chi, ev_ws, ev_bytes, &status, s));
    if (status == 2) return solver_failed(-1, 0, status);
    ok = status == 0;
    return NDMPS_OK;
  };
  // Square root of G2: any Lt with Lt Lt^T = G2 serves (H = Lt^T G1 Lt has the singular values squared whatever the
  // factor, and Lt V_k, G1 Lt V_k do not depend on it).  The Cholesky factor where G2 is numerically positive definite
  // -- the cores to the right of the bond are isometries until compress() reaches them: G2 = I + rounding, a
  // chi-fold eigenvalue, the worst case of an eigen-solver and the best of a Cholesky --, else W D^(1/2) from the
  // eigen-decomposition (zero eigenvalues give zero columns).
  bool g2_chol = direct && !getenv("NDMPS_COMPRESS_EIG") && ndmps_potrf_scratch_elems(chi) <= 2 * c2;
  if (g2_chol) {
    NDMPS_CHECK_HIP(hipMemcpyAsync(Lt, G2, sizeof(double) * c2, hipMemcpyDeviceToDevice, s));
    int bad = 0;
    NDMPS_TRY(ndmps_potrf_lower_f64(Lt, chi, tmp, &bad, s));  // tmp and H (adjacent, 2 c2 doubles) are free until Lt is known
    g2_chol = bad == 0;
  }
  if (!g2_chol) {
    bool g2_direct = direct;
    if (g2_direct) {
      NDMPS_TRY(direct_values(G2, Lt, w2));
      NDMPS_TRY(direct_vectors(chi, g2_direct));
    }
    if (!g2_direct) NDMPS_TRY(ndmps_syevj_f64(G2, chi, Lt, w2, ev_ws, ev_bytes, &sweeps, s));
    hipLaunchKernelGGL(scale_cols_sqrt_kernel, dim3(grid1d(c2)), dim3(256), 0, s, Lt, chi, chi, w2);
    NDMPS_LAUNCH_CHECK();
  }
  NDMPS_TRY(ndmps_dgemm(0, 0, chi, chi, chi, G1, chi, Lt, chi, tmp, chi, s));
  NDMPS_TRY(ndmps_dgemm(1, 0, chi, chi, chi, Lt, chi, tmp, chi, H, chi, s));
  std::vector<double> sv(chi);
  bool h_direct = direct;
  if (h_direct) {
    NDMPS_TRY(direct_values(H, V, wh));  // the solver symmetrises its copy of H
    NDMPS_CHECK_HIP(hipMemcpyAsync(sv.data(), wh, chi * sizeof(double), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    h_direct = direct_rank_is_safe(sv.data(), chi, std::max(cutoff, cutoff_floor<T>()));
  }
  if (!h_direct) {
    NDMPS_TRY(ndmps_syevj_f64(H, chi, V, wh, ev_ws, ev_bytes, &sweeps, s));  // symmetrises H on entry
    NDMPS_CHECK_HIP(hipMemcpyAsync(sv.data(), wh, chi * sizeof(double), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  }
  for (auto& x : sv) x = sqrt(std::max(x, 0.0));  // eigenvalues of H are s^2
  const int64_t k = kept_rank(sv, cutoff, max_bond, cutoff_floor<T>());
  if (h_direct) {
    NDMPS_TRY(direct_vectors(k, h_direct));
    if (!h_direct) NDMPS_TRY(ndmps_syevj_f64(H, chi, V, wh, ev_ws, ev_bytes, &sweeps, s));  // same spectrum, same k
  }
  if (h_s) memcpy(h_s, sv.data(), chi * sizeof(double));
  *h_new_chi = k;

  hipLaunchKernelGGL(sqrt_clamp_kernel, dim3(grid1d(chi)), dim3(256), 0, s, wh, chi, sig);
  NDMPS_LAUNCH_CHECK();
  NDMPS_TRY(ndmps_dgemm(0, 0, chi, k, chi, Lt, chi, V, chi, P1, k, s));
  NDMPS_TRY(ndmps_dgemm(0, 0, chi, k, chi, tmp, chi, V, chi, P2, k, s));
  hipLaunchKernelGGL(scale_cols_to_f32_kernel<T>, dim3(grid1d(chi * k)), dim3(256), 0, s, P1, chi, k, k, sig, -0.5, A1);
  hipLaunchKernelGGL(scale_cols_to_f32_kernel<T>, dim3(grid1d(chi * k)), dim3(256), 0, s, P2, chi, k, k, sig, -1.5, B2);
  NDMPS_LAUNCH_CHECK();
  NDMPS_TRY(gemm_any(0, 0, m1, k, chi, d_t1, chi, A1, k, d_new1, k, s));      // (chi_l d1, k)
  NDMPS_TRY(gemm_any(1, 0, k, n2, chi, B2, k, d_t2, n2, d_new2, n2, s));      // (k, d2 chi_r)
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
}  // namespace

extern "C" int ndmps_compress_bond_f32(const float* d_t1, const float* d_t2, int64_t chi_l, int64_t d1,
                                       int64_t chi, int64_t d2, int64_t chi_r, double cutoff,
                                       int64_t max_bond, float* d_new1, float* d_new2, int64_t* h_new_chi,
                                       double* h_s, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  return compress_bond_impl<float>(d_t1, d_t2, chi_l, d1, chi, d2, chi_r, cutoff, max_bond, d_new1, d_new2, h_new_chi, h_s,
                                   d_ws, ws_bytes, stream);
}
// fp64 cores (same workspace query)
extern "C" int ndmps_compress_bond_f64(const double* d_t1, const double* d_t2, int64_t chi_l, int64_t d1,
                                       int64_t chi, int64_t d2, int64_t chi_r, double cutoff,
                                       int64_t max_bond, double* d_new1, double* d_new2, int64_t* h_new_chi,
                                       double* h_s, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  return compress_bond_impl<double>(d_t1, d_t2, chi_l, d1, chi, d2, chi_r, cutoff, max_bond, d_new1, d_new2, h_new_chi, h_s,
                                    d_ws, ws_bytes, stream);
}

// =================================================================== chain contraction
// Left->right like quimb's structured contraction (core/ndmps.py:140), with the tail pre-contracted: the
// sites j0 .. L-1 whose physical dims multiply to N_{j0} <= 4096 are first contracted among themselves,
// right to left, into R (k_{j0} x N_{j0}) -- GEMMs on matrices of at most a few MB -- so the tensor itself is
// written ONCE, by the last GEMM  Left (M_{j0} x k_{j0}) R.  The cumulative chain alone would write an
// N-element intermediate per trailing site and read it back (2 x 64 MB per site at 256^3 for multiplications
// by 64 x 64 and 8 x 8 matrices).  Same fp32 products, different association.
namespace {
constexpr int64_t kChainTailMax = 4096;

struct ChainPlan {
  int j0 = 0;            // first site of the pre-contracted tail (== L: no tail, j0 == 0 never)
  int64_t left_elems = 0, tail_elems = 0;
};

ChainPlan chain_plan(int L, const int64_t* dims, const int64_t* bonds) {
  ChainPlan p;
  p.j0 = L;
  int64_t right = 1;
  for (int i = L - 1; i >= 1; --i) {
    if (right * dims[i] > kChainTailMax) break;
    right *= dims[i];
    p.j0 = i;
  }
  int64_t rows = 1;
  for (int i = 0; i < p.j0 && i < L; ++i) {
    rows *= dims[i];
    p.left_elems = std::max(p.left_elems, rows * bonds[i + 1]);
  }
  int64_t n = 1;
  for (int i = L - 1; i >= p.j0; --i) {
    n *= dims[i];
    p.tail_elems = std::max(p.tail_elems, bonds[i] * n);
  }
  return p;
}
}  // namespace

namespace {
int64_t chain_workspace(int L, const int64_t* h_dims, const int64_t* h_bonds, int64_t elem_bytes) {
  if (L < 1 || !h_dims || !h_bonds) return 0;
  const ChainPlan p = chain_plan(L, h_dims, h_bonds);
  // + room for the transposed right operand of a bf16 product (a core or a tail matrix)
  int64_t biggest_b = p.tail_elems;
  for (int i = 0; i < L; ++i) biggest_b = std::max(biggest_b, h_bonds[i] * h_dims[i] * h_bonds[i + 1]);
  return (ndmps::round_up(p.left_elems, 64) + 2 * ndmps::round_up(p.tail_elems, 64)) * elem_bytes +
         ndmps::round_up(biggest_b * 2, 256) + 1024;
}
}  // namespace
extern "C" int64_t ndmps_chain_workspace_bytes(int L, const int64_t* h_dims, const int64_t* h_bonds) {
  return chain_workspace(L, h_dims, h_bonds, sizeof(float));
}
extern "C" int64_t ndmps_chain_workspace_bytes_f64(int L, const int64_t* h_dims, const int64_t* h_bonds) {
  return chain_workspace(L, h_dims, h_bonds, sizeof(double));
}

namespace {
struct ChainScatter {          // inverse permutation in the epilogue of the last product (fp32 only)
  const int64_t* row_off;      // [numel / n_cols] offset of tail-block r in the C-order volume
  const int64_t* col_off;      // [n_cols] offsets inside a block, ASCENDING (memory order)
  const int32_t* col_perm;     // [n_cols] site-order column of the c-th smallest offset
  int64_t n_cols;
};

__global__ void __launch_bounds__(256)
gather_cols_kernel(const float* __restrict__ in, int64_t rows, int64_t cols, const int32_t* __restrict__ perm,
                   float* __restrict__ out) {
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    out[e] = in[(e / cols) * cols + perm[e % cols]];
}

inline int final_product(int64_t rows, int64_t n_tail, int64_t k, const float* left, const float* R, float* spare,
                         float* d_out, const ChainScatter* sc, void*, int64_t, hipStream_t s) {
  if (!sc) return ndmps_sgemm(0, 0, rows, n_tail, k, left, k, R, n_tail, d_out, n_tail, s);
  // columns of R in memory order of the volume, then every element goes straight to its voxel
  hipLaunchKernelGGL(gather_cols_kernel, dim3(grid1d(k * n_tail)), dim3(256), 0, s, R, k, n_tail, sc->col_perm, spare);
  NDMPS_LAUNCH_CHECK();
  return ndmps_sgemm_indexed(rows, n_tail, k, left, k, nullptr, nullptr, 0, spare, n_tail, d_out, 0, sc->row_off,
                             sc->col_off, s);
}
inline int final_product(int64_t rows, int64_t n_tail, int64_t k, const double* left, const double* R, double*,
                         double* d_out, const ChainScatter*, void* tws, int64_t tws_bytes, hipStream_t s) {
  return gemm_T(0, rows, n_tail, k, left, R, n_tail, d_out, tws, tws_bytes, s);
}
inline int final_product(int64_t rows, int64_t n_tail, int64_t k, const __bf16* left, const __bf16* R, __bf16*,
                         __bf16* d_out, const ChainScatter*, void* tws, int64_t tws_bytes, hipStream_t s) {
  return gemm_T(0, rows, n_tail, k, left, R, n_tail, d_out, tws, tws_bytes, s);
}

template <typename T>
int chain_impl(int L, const int64_t* h_dims, const int64_t* h_bonds, const T* const* h_cores, T* d_dense, void* d_ws,
               int64_t ws_bytes, ndmps_stream_t stream, const ChainScatter* scatter = nullptr) {
  NDMPS_REQUIRE(L >= 1 && h_dims && h_bonds && h_cores && d_dense, "bad chain argument");
  NDMPS_REQUIRE(h_bonds[0] == 1 && h_bonds[L] == 1, "open boundary bonds must be 1");
  int64_t numel = 1;
  {
    // every intermediate lands in d_ws or d_dense (N = prod(dims) elements): every bond must be at most the
    // product of the site dims on either side of it, as any MPS of a dense tensor has
    int64_t left = 1;
    for (int i = 0; i < L; ++i) {
      NDMPS_REQUIRE(h_dims[i] >= 1 && h_cores[i], "dims[%d] must be positive and core %d non-NULL", i, i);
      numel *= h_dims[i];
    }
    for (int i = 0; i < L; ++i) {
      left *= h_dims[i];
      NDMPS_REQUIRE(h_bonds[i + 1] >= 1 && h_bonds[i + 1] <= left && h_bonds[i + 1] <= numel / left,
                    "bond %d = %lld exceeds min(%lld, %lld), the rank any unfolding can have", i + 1,
                    (long long)h_bonds[i + 1], (long long)left, (long long)(numel / left));
    }
  }
  const int64_t need = chain_workspace(L, h_dims, h_bonds, sizeof(T) == 8 ? 8 : 4);
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("chain workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  if (L == 1) {
    NDMPS_CHECK_HIP(hipMemcpyAsync(d_dense, h_cores[0], numel * sizeof(T), hipMemcpyDeviceToDevice, s));
    return NDMPS_OK;
  }
  const ChainPlan p = chain_plan(L, h_dims, h_bonds);
  T* ws_left = (T*)d_ws;
  T* ws_tail[2] = {ws_left + ndmps::round_up(p.left_elems, 64),
                   ws_left + ndmps::round_up(p.left_elems, 64) + ndmps::round_up(p.tail_elems, 64)};
  // scratch behind the three buffers (bf16: transposed right operands)
  char* tws = (char*)(ws_tail[1] + ndmps::round_up(p.tail_elems, 64));
  tws += (256 - ((uintptr_t)tws & 255)) & 255;
  const int64_t tws_bytes = ((char*)d_ws + ws_bytes) - tws;
  const int j0 = p.j0;
  // ---- tail, right to left: R_i (k_i x N_i) = [core_i as (k_i d_i) x k_{i+1}] R_{i+1}
  const T* R = nullptr;
  int64_t n_tail = 1;
  if (j0 < L) {
    R = h_cores[L - 1];
    n_tail = h_dims[L - 1];
    int t = 0;
    for (int i = L - 2; i >= j0; --i) {
      NDMPS_TRY(gemm_T(0, h_bonds[i] * h_dims[i], n_tail, h_bonds[i + 1], h_cores[i], R, n_tail, ws_tail[t], tws,
                       tws_bytes, s));
      R = ws_tail[t];
      t ^= 1;
      n_tail *= h_dims[i];
    }
  }
  // ---- left part, cumulative: Left_i (rows_i x k_{i+1}); the last product of the whole chain writes d_dense,
  //      the one before it must therefore land in the workspace
  const int last_left = j0 < L ? j0 - 1 : L - 1;  // index of the last cumulative GEMM (site index), 0: none
  const T* left = h_cores[0];
  int64_t rows = h_dims[0];
  for (int i = 1; i <= last_left; ++i) {
    const int64_t chi = h_bonds[i], cols = h_dims[i] * h_bonds[i + 1];
    // products remaining after this one (cumulative ones + the final Left R)
    const int remaining = (last_left - i) + (j0 < L ? 1 : 0);
    T* out = remaining % 2 == 0 ? d_dense : ws_left;
    NDMPS_TRY(gemm_T(0, rows, cols, chi, left, h_cores[i], cols, out, tws, tws_bytes, s));
    left = out;
    rows *= h_dims[i];
  }
  if (scatter)
    NDMPS_REQUIRE(j0 < L && L >= 2 && scatter->n_cols == n_tail,
                  "scatter tables are for %lld tail columns, the chain's tail has %lld", (long long)scatter->n_cols,
                  (long long)(j0 < L ? n_tail : 0));
  if (j0 < L) {
    // R sits in one tail buffer (or is the last core itself); the other one is free for its reordered copy
    T* spare = (R == ws_tail[0]) ? ws_tail[1] : ws_tail[0];
    NDMPS_TRY(final_product(rows, n_tail, h_bonds[j0], left, R, spare, d_dense, scatter, tws, tws_bytes, s));
  }
  NDMPS_REQUIRE(j0 < L || left == d_dense, "internal: chain result landed in the wrong buffer");
  return NDMPS_OK;
}
}  // namespace

extern "C" int ndmps_chain_contract_f32(int L, const int64_t* h_dims, const int64_t* h_bonds,
                                        const float* const* h_cores, float* d_dense, void* d_ws,
                                        int64_t ws_bytes, ndmps_stream_t stream) {
  return chain_impl<float>(L, h_dims, h_bonds, h_cores, d_dense, d_ws, ws_bytes, stream);
}

// number of trailing columns the chain pre-contracts (product of the dims of the tail sites), 0 if none
extern "C" int64_t ndmps_chain_tail_columns(int L, const int64_t* h_dims) {
  if (L < 2 || !h_dims) return 0;
  int64_t right = 1;
  int taken = 0;
  for (int i = L - 1; i >= 1; --i) {
    if (right * h_dims[i] > kChainTailMax) break;
    right *= h_dims[i];
    ++taken;
  }
  return taken > 0 ? right : 0;
}

// Chain contraction that writes the C-order VOLUME: the inverse index permutation (core/ndmps.py:144-148) rides
// on the last product, every element goes from the accumulator to its voxel (d_row_off / d_col_off /
// d_col_perm: ndmps_plan_split_offsets for n_cols = ndmps_chain_tail_columns, columns sorted by offset).  The
// site-order tensor is never written.
extern "C" int ndmps_chain_contract_scatter_f32(int L, const int64_t* h_dims, const int64_t* h_bonds,
                                                const float* const* h_cores, float* d_out,
                                                const int64_t* d_row_off, const int64_t* d_col_off,
                                                const int32_t* d_col_perm, int64_t n_cols, void* d_ws,
                                                int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_row_off && d_col_off && d_col_perm && n_cols >= 1, "NULL scatter table");
  ChainScatter sc{d_row_off, d_col_off, d_col_perm, n_cols};
  return chain_impl<float>(L, h_dims, h_bonds, h_cores, d_out, d_ws, ws_bytes, stream, &sc);
}

// The same for a list of MPS over the same sites (conv_to_tensors, evaluation/benchmark.py:80-100): volume b has
// bonds h_bonds[b (L + 1) ..], cores h_cores[b L ..] and goes to h_out[b]; one workspace (sized for the largest
// bonds) serves them in turn on `stream`.  The launches of all volumes are issued by this one call.
namespace {
struct PtrPairs {  // operands of a small per-volume kernel run for a whole batch (grid.y)
  const void* in[64];
  void* out[64];
};
__global__ void __launch_bounds__(256)
gather_cols_batched_kernel(PtrPairs pp, int64_t rows, int64_t cols, const int32_t* __restrict__ perm) {
  const float* in = static_cast<const float*>(pp.in[blockIdx.y]);
  float* out = static_cast<float*>(pp.out[blockIdx.y]);
  const int64_t total = rows * cols;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256)
    out[e] = in[(e / cols) * cols + perm[e % cols]];
}

// chain_impl<float> with the scatter epilogue for `count` (<= 64) MPS that share their bonds: every stage is one
// batched launch (ndmps_sgemm_batched); volume b works in its own slice of the workspace.  Same products in the
// same order as chain_impl on each volume: bit-identical results.
int chain_batched_same_bonds(int count, int L, const int64_t* h_dims, const int64_t* h_bonds, const float* const* h_cores,
                             float* const* h_out, const ChainScatter& sc, char* d_ws, int64_t ws_each, hipStream_t s) {
  const ChainPlan p = chain_plan(L, h_dims, h_bonds);
  const int j0 = p.j0;
  NDMPS_REQUIRE(L >= 2 && j0 < L, "internal: batched chain needs a pre-contracted tail");
  std::vector<const float*> A(count), B(count);
  std::vector<float*> Cc(count);
  std::vector<float*> ws_left(count), ws_tail0(count), ws_tail1(count);
  for (int b = 0; b < count; ++b) {
    ws_left[b] = (float*)(d_ws + (int64_t)b * ws_each);
    ws_tail0[b] = ws_left[b] + ndmps::round_up(p.left_elems, 64);
    ws_tail1[b] = ws_tail0[b] + ndmps::round_up(p.tail_elems, 64);
  }
  // ---- tail, right to left
  std::vector<const float*> R(count);
  for (int b = 0; b < count; ++b) R[b] = h_cores[(int64_t)b * L + L - 1];
  int64_t n_tail = h_dims[L - 1];
  int t = 0;
  for (int i = L - 2; i >= j0; --i) {
    for (int b = 0; b < count; ++b) {
      A[b] = h_cores[(int64_t)b * L + i];
      B[b] = R[b];
      Cc[b] = t == 0 ? ws_tail0[b] : ws_tail1[b];
    }
    NDMPS_TRY(ndmps_sgemm_batched(count, 0, 0, h_bonds[i] * h_dims[i], n_tail, h_bonds[i + 1], A.data(), h_bonds[i + 1],
                                  B.data(), n_tail, Cc.data(), n_tail, s));
    for (int b = 0; b < count; ++b) R[b] = Cc[b];
    t ^= 1;
    n_tail *= h_dims[i];
  }
  // ---- left part, cumulative; the product before the final one must land in the workspace
  const int last_left = j0 - 1;
  std::vector<const float*> left(count);
  for (int b = 0; b < count; ++b) left[b] = h_cores[(int64_t)b * L];
  int64_t rows = h_dims[0];
  for (int i = 1; i <= last_left; ++i) {
    const int64_t chi = h_bonds[i], cols = h_dims[i] * h_bonds[i + 1];
    const int remaining = (last_left - i) + 1;
    for (int b = 0; b < count; ++b) {
      A[b] = left[b];
      B[b] = h_cores[(int64_t)b * L + i];
      Cc[b] = remaining % 2 == 0 ? h_out[b] : ws_left[b];
    }
    NDMPS_TRY(ndmps_sgemm_batched(count, 0, 0, rows, cols, chi, A.data(), chi, B.data(), cols, Cc.data(), cols, s));
    for (int b = 0; b < count; ++b) left[b] = Cc[b];
    rows *= h_dims[i];
  }
  NDMPS_REQUIRE(sc.n_cols == n_tail, "scatter tables are for %lld tail columns, the chain's tail has %lld",
                (long long)sc.n_cols, (long long)n_tail);
  // ---- final product: columns of R in memory order, every element straight to its voxel
  PtrPairs pp;
  for (int b = 0; b < count; ++b) {
    float* spare = (R[b] == ws_tail0[b]) ? ws_tail1[b] : ws_tail0[b];
    pp.in[b] = R[b];
    pp.out[b] = spare;
    A[b] = left[b];
    B[b] = spare;
    Cc[b] = h_out[b];
  }
  const int64_t k = h_bonds[j0];
  hipLaunchKernelGGL(gather_cols_batched_kernel, dim3(grid1d(k * n_tail), count), dim3(256), 0, s, pp, k, n_tail,
                     sc.col_perm);
  NDMPS_LAUNCH_CHECK();
  return ndmps_sgemm_indexed_batched(count, rows, n_tail, k, A.data(), k, nullptr, nullptr, 0, B.data(), n_tail, Cc.data(), 0,
                                     sc.row_off, sc.col_off, s);
}
}  // namespace

// The same for a list of MPS over the same sites (conv_to_tensors, evaluation/benchmark.py:80-100): volume b has
// bonds h_bonds[b (L + 1) ..], cores h_cores[b L ..] and goes to h_out[b].  MPS that share their bonds (a lockstep
// group whose caps bind) go through the chain TOGETHER, one batched launch per stage, each in its own slice of
// d_ws (ndmps_chain_batched_workspace_bytes); otherwise the volumes are contracted in turn.  Bit-identical to
// ndmps_chain_contract_scatter_f32 on each volume either way.
extern "C" int64_t ndmps_chain_batched_workspace_bytes(int batch, int L, const int64_t* h_dims, const int64_t* h_bonds) {
  if (batch < 1 || L < 1 || !h_dims || !h_bonds) return 0;
  int64_t each = 0;
  bool same = true;
  for (int b = 0; b < batch; ++b) {
    each = std::max(each, ndmps_chain_workspace_bytes(L, h_dims, h_bonds + (int64_t)b * (L + 1)));
    for (int i = 0; i <= L; ++i) same = same && h_bonds[(int64_t)b * (L + 1) + i] == h_bonds[i];
  }
  each = ndmps::round_up(each, 256);
  return same ? each * std::min(batch, 64) : each;
}

extern "C" int ndmps_chain_contract_scatter_batched_f32(int batch, int L, const int64_t* h_dims, const int64_t* h_bonds,
                                                        const float* const* h_cores, float* const* h_out,
                                                        const int64_t* d_row_off, const int64_t* d_col_off,
                                                        const int32_t* d_col_perm, int64_t n_cols, void* d_ws,
                                                        int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(batch >= 1 && L >= 1 && h_bonds && h_cores && h_out, "bad batched chain argument");
  NDMPS_REQUIRE(d_row_off && d_col_off && d_col_perm && n_cols >= 1, "NULL scatter table");
  ChainScatter sc{d_row_off, d_col_off, d_col_perm, n_cols};
  bool same = batch > 1 && L >= 2;
  for (int b = 1; b < batch && same; ++b)
    for (int i = 0; i <= L; ++i) same = same && h_bonds[(int64_t)b * (L + 1) + i] == h_bonds[i];
  const int64_t each = ndmps::round_up(ndmps_chain_workspace_bytes(L, h_dims, h_bonds), 256);
  if (same && chain_plan(L, h_dims, h_bonds).j0 < L && d_ws && ws_bytes >= each * std::min(batch, 64)) {
    // validate once through the single-volume entry's checks (bonds, operands) on volume 0 without launching
    for (int b = 0; b < batch; ++b) {
      NDMPS_REQUIRE(h_out[b], "NULL output %d", b);
      for (int i = 0; i < L; ++i) NDMPS_REQUIRE(h_cores[(int64_t)b * L + i], "core %d of volume %d is NULL", i, b);
    }
    {
      int64_t numel = 1, left = 1;
      NDMPS_REQUIRE(h_bonds[0] == 1 && h_bonds[L] == 1, "open boundary bonds must be 1");
      for (int i = 0; i < L; ++i) {
        NDMPS_REQUIRE(h_dims[i] >= 1, "dims[%d] must be positive", i);
        numel *= h_dims[i];
      }
      for (int i = 0; i < L; ++i) {
        left *= h_dims[i];
        NDMPS_REQUIRE(h_bonds[i + 1] >= 1 && h_bonds[i + 1] <= left && h_bonds[i + 1] <= numel / left,
                      "bond %d = %lld exceeds min(%lld, %lld), the rank any unfolding can have", i + 1,
                      (long long)h_bonds[i + 1], (long long)left, (long long)(numel / left));
      }
    }
    for (int base = 0; base < batch; base += 64) {
      const int count = std::min(64, batch - base);
      NDMPS_TRY(chain_batched_same_bonds(count, L, h_dims, h_bonds, h_cores + (int64_t)base * L, h_out + base, sc,
                                         (char*)d_ws, each, (hipStream_t)stream));
    }
    return NDMPS_OK;
  }
  for (int b = 0; b < batch; ++b)
    NDMPS_TRY(chain_impl<float>(L, h_dims, h_bonds + (int64_t)b * (L + 1), h_cores + (int64_t)b * L, h_out[b], d_ws,
                                ws_bytes, stream, &sc));
  return NDMPS_OK;
}

extern "C" int ndmps_chain_contract_bf16(int L, const int64_t* h_dims, const int64_t* h_bonds,
                                         const void* const* h_cores, void* d_dense, void* d_ws,
                                         int64_t ws_bytes, ndmps_stream_t stream) {
  return chain_impl<__bf16>(L, h_dims, h_bonds, (const __bf16* const*)h_cores, (__bf16*)d_dense, d_ws, ws_bytes, stream);
}

// fp64 cores: every product on the fp64 MFMA (workspace: ndmps_chain_workspace_bytes_f64)
extern "C" int ndmps_chain_contract_f64(int L, const int64_t* h_dims, const int64_t* h_bonds,
                                        const double* const* h_cores, double* d_dense, void* d_ws,
                                        int64_t ws_bytes, ndmps_stream_t stream) {
  return chain_impl<double>(L, h_dims, h_bonds, h_cores, d_dense, d_ws, ws_bytes, stream);
}

// =================================================================== overlap
extern "C" int64_t ndmps_overlap_workspace_bytes(int L, const int64_t* h_dims, const int64_t* h_bonds_a,
                                                 const int64_t* h_bonds_b) {
  if (L < 1 || !h_dims || !h_bonds_a || !h_bonds_b) return 0;
  int64_t emax = 1, amax = 1, bmax = 1, xmax = 1;
  for (int i = 0; i < L; ++i) {
    emax = std::max(emax, h_bonds_a[i + 1] * h_bonds_b[i + 1]);
    amax = std::max(amax, h_bonds_a[i] * h_dims[i] * h_bonds_a[i + 1]);
    bmax = std::max(bmax, h_bonds_b[i] * h_dims[i] * h_bonds_b[i + 1]);
    xmax = std::max(xmax, h_bonds_b[i] * h_dims[i] * h_bonds_a[i + 1]);
  }
  int64_t used = 0;
  used = arena_bytes(used, 8, emax);
  used = arena_bytes(used, 8, emax);
  used = arena_bytes(used, 8, amax);
  used = arena_bytes(used, 8, bmax);
  used = arena_bytes(used, 8, xmax);
  return ndmps::round_up(used, 256) + 256;
}

namespace {
template <typename T>
int overlap_impl(int L, const int64_t* h_dims, const int64_t* h_bonds_a, const T* const* h_cores_a,
                 const int64_t* h_bonds_b, const T* const* h_cores_b, double* h_out, void* d_ws, int64_t ws_bytes,
                 ndmps_stream_t stream) {
  NDMPS_REQUIRE(L >= 1 && h_dims && h_bonds_a && h_bonds_b && h_cores_a && h_cores_b && h_out,
                "bad overlap argument");
  const int64_t need = ndmps_overlap_workspace_bytes(L, h_dims, h_bonds_a, h_bonds_b);
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("overlap workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  int64_t emax = 1, amax = 1, bmax = 1, xmax = 1;
  for (int i = 0; i < L; ++i) {
    emax = std::max(emax, h_bonds_a[i + 1] * h_bonds_b[i + 1]);
    amax = std::max(amax, h_bonds_a[i] * h_dims[i] * h_bonds_a[i + 1]);
    bmax = std::max(bmax, h_bonds_b[i] * h_dims[i] * h_bonds_b[i + 1]);
    xmax = std::max(xmax, h_bonds_b[i] * h_dims[i] * h_bonds_a[i + 1]);
  }
  Arena ar(d_ws, ws_bytes);
  double* E[2] = {ar.take<double>(emax), ar.take<double>(emax)};
  double* A = ar.take<double>(amax);
  double* B = ar.take<double>(bmax);
  double* X = ar.take<double>(xmax);
  NDMPS_REQUIRE(E[0] && E[1] && A && B && X, "workspace carve failed");

  hipLaunchKernelGGL(set_scalar_f64_kernel, dim3(1), dim3(1), 0, s, E[0], 1.0);  // no copy from pageable host memory
  int cur = 0;
  for (int i = 0; i < L; ++i) {
    const int64_t ca = h_bonds_a[i], ca2 = h_bonds_a[i + 1];
    const int64_t cb = h_bonds_b[i], cb2 = h_bonds_b[i + 1];
    const int64_t d = h_dims[i];
    hipLaunchKernelGGL(f32_to_f64_kernel<T>, dim3(grid1d(ca * d * ca2)), dim3(256), 0, s, h_cores_a[i], ca * d * ca2, A);
    hipLaunchKernelGGL(f32_to_f64_kernel<T>, dim3(grid1d(cb * d * cb2)), dim3(256), 0, s, h_cores_b[i], cb * d * cb2, B);
    NDMPS_LAUNCH_CHECK();
    // X (cb, d ca2) = E^T (cb, ca) A (ca, d ca2)
    NDMPS_TRY(ndmps_dgemm(1, 0, cb, d * ca2, ca, E[cur], cb, A, d * ca2, X, d * ca2, s));
    // E' (ca2, cb2) = X'^T B' with X' = (cb d, ca2), B' = (cb d, cb2)
    NDMPS_TRY(ndmps_dgemm(1, 0, ca2, cb2, cb * d, X, ca2, B, cb2, E[cur ^ 1], cb2, s));
    cur ^= 1;
  }
  NDMPS_CHECK_HIP(hipMemcpyAsync(h_out, E[cur], sizeof(double), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
}  // namespace

extern "C" int ndmps_overlap_f32(int L, const int64_t* h_dims, const int64_t* h_bonds_a,
                                 const float* const* h_cores_a, const int64_t* h_bonds_b,
                                 const float* const* h_cores_b, double* h_out, void* d_ws,
                                 int64_t ws_bytes, ndmps_stream_t stream) {
  return overlap_impl<float>(L, h_dims, h_bonds_a, h_cores_a, h_bonds_b, h_cores_b, h_out, d_ws, ws_bytes, stream);
}
// fp64 cores (same workspace query)
extern "C" int ndmps_overlap_f64(int L, const int64_t* h_dims, const int64_t* h_bonds_a,
                                 const double* const* h_cores_a, const int64_t* h_bonds_b,
                                 const double* const* h_cores_b, double* h_out, void* d_ws,
                                 int64_t ws_bytes, ndmps_stream_t stream) {
  return overlap_impl<double>(L, h_dims, h_bonds_a, h_cores_a, h_bonds_b, h_cores_b, h_out, d_ws, ws_bytes, stream);
}
