// Direct symmetric eigen-solver for the leading k eigenpairs (fp64), batched: the production path of the
// bond-capped sweep (k = chi <= 128 of n = d chi columns).
//
//   G = Q T Q^T      Householder tridiagonalisation, unblocked, the rank-2 update of step j-1 applied while
//                    step j reads the matrix (one pass over the trailing matrix per column)
//   T  -> w          all n eigenvalues by multi-section on a Sturm sequence (no divisions in the chain)
//   T  -> Z (n x k)  inverse iteration from random starts, pivoted tridiagonal LU, three rounds with a
//                    Cholesky-QR of the whole block between rounds (clusters need no special casing)
//   V = Q Z          reflectors replayed on the k columns held in registers
//
// Why not Jacobi (eig_block.hip): at n = 512 a Jacobi solve is ~14 sweeps x 31 dependent launches of
// 16-26 us and ~50 n^3 flops; this path is n - 128 dependent launches of ~5 us plus four short kernels and
// 4/3 n^3 flops, independent of the spectrum (the noise-floor clusters of the volume Gram matrices make
// Jacobi converge linearly for most of its sweeps).
//
// Launch structure of the reduction.  A column step needs the product of the whole trailing matrix with
// the new reflector, i.e. one grid-wide dependency per column.  Kernel boundaries provide it: launch j
// covers column j for every matrix of the batch, workgroup (r, b) owns 32 columns of matrix b and reads
// them over all trailing rows (lanes own columns, so by symmetry the column sums ARE the matrix-vector
// product and no cross-lane reduction is needed).  Every workgroup recomputes the O(n) vector part
// (previous step's w, this step's reflector) redundantly and bit-identically.  When the trailing matrix
// fits the LDS (<= 128 x 128) one workgroup per matrix finishes the reduction without further launches.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int kTail = 128;        // trailing order finished in LDS by one workgroup
constexpr int kTailLd = 132;      // LDS row stride of the tail matrix (ld = 4 mod 32: conflict-free row walks)
constexpr int kColsPerWg = 32;    // columns of one workgroup of the column kernel
constexpr int kMaxK = 128;        // largest number of eigenvectors (Cholesky factor lives in LDS)
constexpr int kMaxN = 4096;       // three n-vectors of the column kernel live in LDS
constexpr int kSect = 8;          // lanes per eigenvalue in the multi-section
constexpr int kInvIters = 3;

struct TrdDesc {       // one per matrix (device array), blockIdx.y selects it
  const double* G_in;  // n x n (ld n)
  double* V_out;       // n x n (ld n): eigenvector c in column c, columns 0..k-1 written
  double* w_out;       // n eigenvalues, descending
  int n;
  int k;               // eigenvectors wanted (vectors phase)
  int status;          // != 0: Cholesky breakdown in the orthonormalisation
  int pad;
};

struct TrdWork {       // per-matrix strides; everything indexed by blockIdx.y
  double* A;           // [B][n_max][lda] working copy, trailing part updated in place
  double* Vh;          // [B][n_max][lda] reflector j in row j (zeros up to j, 1 at j + 1)
  double* y;           // [B][2][lda] matrix-vector products, by parity of the column
  double* tau;         // [B][n_max]
  double* d;           // [B][n_max] diagonal of T
  double* e;           // [B][n_max] sub-diagonal of T
  double* lam;         // [B][n_max] eigenvalues of T / bound, descending
  double* bound;       // [B] Gershgorin bound of T
  double* Z;           // [B][n_max][kp] eigenvectors of T
  double* lu;          // [B][4][n_max][kp] dl, 1/d, du, du2 of the pivoted factorisations
  unsigned char* piv;  // [B][n_max][kp]
  int n_max, lda, kp;
};

template <int STEPS>
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int i = 0; i < STEPS; ++i) y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
template <int STEPS>
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int i = 0; i < STEPS; ++i) y = y * fma(-x, y, 2.0);
  return y;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// sum over the workgroup, identical in every thread; `red` holds one double per wave; NW waves
template <int NW>
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  __syncthreads();  // red may still be read from the previous reduction
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += red[w];
  return s;
}

// Householder vector of x = (alpha, rest) with sigma = |rest|^2:  H x = beta e_1,  H = I - tau v v^T, v_1 = 1,
// v_rest = rest * scale.  sigma == 0 gives H = I (LAPACK dlarfg).
__device__ __forceinline__ void householder(double alpha, double sigma, double& beta, double& tau, double& scale) {
  if (sigma == 0.0) {
    beta = alpha;
    tau = 0.0;
    scale = 0.0;
    return;
  }
  const double h2 = fma(alpha, alpha, sigma);
  const double rs = fast_rsqrt<2>(h2);
  const double nrm = h2 * rs;
  beta = alpha >= 0.0 ? -nrm : nrm;
  tau = (beta - alpha) * (alpha >= 0.0 ? -rs : rs);
  scale = fast_rcp<2>(alpha - beta);
}

// ---------------------------------------------------------------------------------------------- load
__global__ void __launch_bounds__(256) trd_load_kernel(const TrdDesc* __restrict__ desc, TrdWork w) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, lda = w.lda;
  const double* G = d.G_in;
  double* A = w.A + (int64_t)blockIdx.y * w.n_max * lda;
  const int64_t total = (int64_t)n * lda;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / lda), c = (int)(e % lda);
    A[e] = c < n ? 0.5 * (G[(int64_t)r * n + c] + G[(int64_t)c * n + r]) : 0.0;
  }
}

// ------------------------------------------------------------------------------------- column kernel
// Launch j, workgroup (r, b): columns [32 r, 32 r + 32) of matrix b, all rows > j.
//   prologue (redundant in every workgroup, O(n)): w' = tau' y' - (tau'^2 / 2)(y' . v') v' of step j - 1,
//            row j with that update applied -> d_j and the reflector v_j, tau_j, e_j
//   body   : A[i][c] -= v'[i] w'[c] + w'[i] v'[c];  y[c] += A[i][c] v_j[i]   (i > j)
// Lane layout of the body: 16 lanes x 16 bytes cover the 32 columns of one row, 4 rows per wave instruction.
__global__ void __launch_bounds__(256)
trd_column_kernel(const TrdDesc* __restrict__ desc, TrdWork w, int j) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  if (j >= n - kTail) return;              // this matrix is (or will be) finished by the tail kernel
  const int c0 = blockIdx.x * kColsPerWg;
  if (c0 >= n || c0 + kColsPerWg <= j + 1) return;  // columns <= j are finished
  const int lda = w.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  double* A = w.A + b * w.n_max * lda;
  double* Vh = w.Vh + b * w.n_max * lda;
  double* ybuf = w.y + b * 2 * lda;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* vp = lds;               // v'  (reflector of step j - 1)
  double* wp = lds + w.n_max;     // w'
  double* vj = lds + 2 * w.n_max; // reflector of this step
  __shared__ double red[4];
  __shared__ double part[4][kColsPerWg];

  // ---- prologue a: w' of step j - 1
  if (j >= 1) {
    const double* yprev = ybuf + ((j - 1) & 1) * lda;
    const double* vprev = Vh + (int64_t)(j - 1) * lda;
    const double taup = w.tau[b * w.n_max + j - 1];
    double dot = 0.0;
    for (int i = tid; i < n; i += 256) {
      const double vv = i >= j ? vprev[i] : 0.0;
      const double yv = i >= j ? yprev[i] : 0.0;
      vp[i] = vv;
      wp[i] = yv;
      dot = fma(yv, vv, dot);
    }
    dot = block_sum<4>(dot, red);
    const double al = 0.5 * taup * taup * dot;
    for (int i = tid; i < n; i += 256) wp[i] = taup * wp[i] - al * vp[i];
    __syncthreads();
  }
  // ---- prologue b: row j, updated; d_j, x = row[j+1:], sigma = |x[1:]|^2
  const double* rowj = A + (int64_t)j * lda;
  const double vpj = j >= 1 ? vp[j] : 0.0, wpj = j >= 1 ? wp[j] : 0.0;
  double sigma = 0.0;
  for (int i = tid; i < n; i += 256) {
    double a = 0.0;
    if (i >= j) {
      a = rowj[i];
      if (j >= 1) a -= vpj * wp[i] + wpj * vp[i];
    }
    vj[i] = a;
    if (i > j + 1) sigma = fma(a, a, sigma);
  }
  sigma = block_sum<4>(sigma, red);
  const double dj = vj[j], alpha = vj[j + 1];
  double beta, tau, scale;
  householder(alpha, sigma, beta, tau, scale);
  __syncthreads();  // everybody has read vj[j], vj[j + 1]
  const bool writer = (int)blockIdx.x == (j + 1) / kColsPerWg;
  for (int i = tid; i < lda; i += 256) {
    double v = 0.0;
    if (i < n) {
      v = i > j + 1 ? vj[i] * scale : (i == j + 1 ? 1.0 : 0.0);
      vj[i] = v;
    }
    if (writer) Vh[(int64_t)j * lda + i] = v;
  }
  if (writer && tid == 0) {
    w.d[b * w.n_max + j] = dj;
    w.e[b * w.n_max + j] = beta;
    w.tau[b * w.n_max + j] = tau;
  }
  __syncthreads();

  // ---- body
  const int q = lane >> 4, p = lane & 15;
  const int c = c0 + 2 * p;
  const bool col_ok = c < lda;  // lda is even: c + 1 < lda too; the padding column holds zeros
  double wc0 = 0.0, wc1 = 0.0, vc0 = 0.0, vc1 = 0.0;
  if (j >= 1 && col_ok) {
    if (c < n) { wc0 = wp[c]; vc0 = vp[c]; }
    if (c + 1 < n) { wc1 = wp[c + 1]; vc1 = vp[c + 1]; }
  }
  double acc0 = 0.0, acc1 = 0.0;
  constexpr int U = 8;
  const int first = j + 1 + 4 * wave + q;
  for (int i0 = first; i0 < n; i0 += 16 * U) {
    double2 a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 16 * u;
      a[u] = (i < n && col_ok) ? *reinterpret_cast<const double2*>(A + (int64_t)i * lda + c) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + 16 * u;
      if (i < n && col_ok) {
        const double vji = vj[i];
        if (j >= 1) {
          const double vpi = vp[i], wpi = wp[i];
          a[u].x -= fma(vpi, wc0, wpi * vc0);
          a[u].y -= fma(vpi, wc1, wpi * vc1);
          *reinterpret_cast<double2*>(A + (int64_t)i * lda + c) = a[u];
        }
        acc0 = fma(a[u].x, vji, acc0);
        acc1 = fma(a[u].y, vji, acc1);
      }
    }
  }
  // ---- epilogue: rows -> one value per column, fixed order
  acc0 += __shfl_xor(acc0, 16, 64);
  acc0 += __shfl_xor(acc0, 32, 64);
  acc1 += __shfl_xor(acc1, 16, 64);
  acc1 += __shfl_xor(acc1, 32, 64);
  if (q == 0) {
    part[wave][2 * p] = acc0;
    part[wave][2 * p + 1] = acc1;
  }
  __syncthreads();
  if (tid < kColsPerWg && c0 + tid < lda)
    ybuf[(j & 1) * lda + c0 + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// --------------------------------------------------------------------------------------- tail kernel
// One workgroup per matrix finishes columns J = max(n - 128, 0) .. n - 2 with the trailing matrix in LDS.
// 512 threads: thread (r, t) = (tid / 4, tid % 4) owns row r of the trailing matrix, columns c = t mod 4.
__global__ void __launch_bounds__(512) trd_tail_kernel(const TrdDesc* __restrict__ desc, TrdWork w) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  const int J = max(n - kTail, 0), m = n - J;
  const int lda = w.lda;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.y;
  const double* A = w.A + b * w.n_max * lda;
  double* Vh = w.Vh + b * w.n_max * lda;
  const double* ybuf = w.y + b * 2 * lda;
  double* dd = w.d + b * w.n_max;
  double* ee = w.e + b * w.n_max;
  double* tt = w.tau + b * w.n_max;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;                        // [kTail][kTailLd]
  double* vp = lds + kTail * kTailLd;     // pending update (step jj - 1): v', w'
  double* wp = vp + kTail;
  double* vj = wp + kTail;                // reflector of the current step
  double* yv = vj + kTail;                // S v_j
  __shared__ double red[8];

  // pending update of the last column launch
  double taup = 0.0;
  if (J >= 1) {
    const double* yprev = ybuf + ((J - 1) & 1) * lda;
    const double* vprev = Vh + (int64_t)(J - 1) * lda;
    taup = tt[J - 1];
    double dot = 0.0;
    if (tid < m) {
      const double vv = vprev[J + tid], y0 = yprev[J + tid];
      vp[tid] = vv;
      wp[tid] = y0;
      dot = y0 * vv;
    }
    dot = block_sum<8>(dot, red);
    const double al = 0.5 * taup * taup * dot;
    if (tid < m) wp[tid] = taup * wp[tid] - al * vp[tid];
  } else if (tid < m) {
    vp[tid] = 0.0;
    wp[tid] = 0.0;
  }
  __syncthreads();
  for (int e = tid; e < m * m; e += 512) {
    const int r = e / m, c = e % m;
    S[r * kTailLd + c] = A[(int64_t)(J + r) * lda + J + c] - (vp[r] * wp[c] + wp[r] * vp[c]);
  }
  __syncthreads();
  if (tid < m) {  // the update is applied: nothing pending
    vp[tid] = 0.0;
    wp[tid] = 0.0;
  }
  __syncthreads();

  const int r = tid >> 2, t = tid & 3;
  for (int jj = 0; jj + 1 < m; ++jj) {
    const int j = J + jj;
    // row jj with the pending update -> d_j, x, sigma
    const double vpj = vp[jj], wpj = wp[jj];
    double sigma = 0.0, a = 0.0;
    if (tid < m && tid >= jj) {
      a = S[jj * kTailLd + tid] - (vpj * wp[tid] + wpj * vp[tid]);
      if (tid > jj + 1) sigma = a * a;
    }
    if (tid < m) vj[tid] = a;
    sigma = block_sum<8>(sigma, red);
    const double dj = vj[jj], alpha = vj[jj + 1];
    double beta, tau, scale;
    householder(alpha, sigma, beta, tau, scale);
    __syncthreads();  // everybody has read vj[jj], vj[jj + 1]
    if (tid < m) vj[tid] = tid > jj + 1 ? a * scale : (tid == jj + 1 ? 1.0 : 0.0);
    if (tid == 0) {
      dd[j] = dj;
      ee[j] = beta;
      tt[j] = tau;
    }
    __syncthreads();
    for (int i = tid; i < lda; i += 512) Vh[(int64_t)j * lda + i] = (i >= J && i < n) ? vj[i - J] : 0.0;
    // one pass: apply the pending update, accumulate y = S v_j (rows and columns > jj)
    double acc = 0.0;
    if (r < m && r > jj) {
      const double vpr = vp[r], wpr = wp[r];
      double* row = S + r * kTailLd;
      for (int c = jj + 1 + t; c < m; c += 4) {
        const double s = row[c] - (vpr * wp[c] + wpr * vp[c]);
        row[c] = s;
        acc = fma(s, vj[c], acc);
      }
    }
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    if (t == 0 && r < m) yv[r] = r > jj ? acc : 0.0;
    __syncthreads();
    double dot = 0.0;
    if (tid < m) dot = yv[tid] * vj[tid];
    dot = block_sum<8>(dot, red);
    const double al = 0.5 * tau * tau * dot;
    if (tid < m) {
      wp[tid] = tau * yv[tid] - al * vj[tid];
      vp[tid] = vj[tid];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int jj = m - 1;
    dd[n - 1] = S[jj * kTailLd + jj] - 2.0 * vp[jj] * wp[jj];
    ee[n - 1] = 0.0;
    tt[n - 1] = 0.0;
  }
  for (int i = tid; i < lda; i += 512) Vh[(int64_t)(n - 1) * lda + i] = 0.0;
}

// ----------------------------------------------------------------------------------------- bisection
// # eigenvalues of the scaled T below x: sign changes of p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}, a
// zero taking the sign opposite to its predecessor.  No division in the dependent chain; the pair is
// rescaled by a power of two every fourth step (|d - x| <= 2, e^2 <= 1 after scaling by the Gershgorin
// bound, so four steps stay far inside the fp64 range).  de2[i] = (d_i, e_{i-1}^2) in LDS.
__device__ __forceinline__ int sturm_count(const double2* __restrict__ de2, int n, double x) {
  double pm = 1.0, pc = de2[0].x - x;
  bool neg = !(pc > 0.0);  // zero counts as a change from p_{-1} = 1
  int cnt = neg ? 1 : 0;
  for (int i = 1; i < n; ++i) {
    const double2 v = de2[i];
    const double pn = fma(v.x - x, pc, -v.y * pm);
    const bool neg_n = pn == 0.0 ? !neg : pn < 0.0;
    cnt += neg_n != neg;
    neg = neg_n;
    pm = pc;
    pc = pn;
    if ((i & 3) == 0) {
      const double mx = fmax(fabs(pm), fabs(pc));
      int ex = 0;
      if (mx > 0.0) (void)frexp(mx, &ex);
      pm = ldexp(pm, -ex);
      pc = ldexp(pc, -ex);
    }
  }
  return cnt;
}

// grid (ceil(n_max * kSect / 256), B): groups of kSect lanes find one eigenvalue each by multi-section
__global__ void __launch_bounds__(256) trd_bisect_kernel(const TrdDesc* __restrict__ desc, TrdWork w) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.y;
  const int first = blockIdx.x * (256 / kSect);
  if (first >= n) return;
  const double* dd = w.d + b * w.n_max;
  const double* ee = w.e + b * w.n_max;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double2* de2 = reinterpret_cast<double2*>(lds);
  __shared__ double red[4];
  // Gershgorin bound (max over the workgroup, identical in every workgroup of the matrix)
  double g = 0.0;
  for (int i = tid; i < n; i += 256)
    g = fmax(g, fabs(dd[i]) + (i > 0 ? fabs(ee[i - 1]) : 0.0) + (i + 1 < n ? fabs(ee[i]) : 0.0));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) g = fmax(g, __shfl_xor(g, off, 64));
  if ((tid & 63) == 0) red[tid >> 6] = g;
  __syncthreads();
  const double bound = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const double inv = bound > 0.0 ? 1.0 / bound : 0.0;
  for (int i = tid; i < n; i += 256) {
    const double es = i > 0 ? ee[i - 1] * inv : 0.0;
    de2[i] = make_double2(dd[i] * inv, es * es);
  }
  __syncthreads();
  const int grp = tid / kSect, s = tid % kSect;
  const int m = first + grp;          // ascending index of this group's eigenvalue
  const bool live = m < n;
  double lo = -1.001, hi = 1.001;
  // 2.002 / 9^18 = 1.3e-17: below the spacing of fp64 numbers at the bound
  for (int round = 0; round < 18; ++round) {
    const double h = (hi - lo) * (1.0 / (kSect + 1));
    const double x = lo + h * (s + 1);
    const int cnt = live ? sturm_count(de2, n, x) : 0;
    // points whose count is <= m lie at or below the eigenvalue
    const unsigned long long bal = __ballot(cnt <= m);
    const int shift = (tid & 63) - s;
    const int below = __popcll((bal >> shift) & ((1ull << kSect) - 1));
    lo = lo + h * below;
    hi = lo + h;
  }
  if (live && s == 0) {
    const double lam = 0.5 * (lo + hi);
    w.lam[b * w.n_max + (n - 1 - m)] = lam;
    d.w_out[n - 1 - m] = lam * bound;
  }
  if (blockIdx.x == 0 && tid == 0) w.bound[b] = bound;
}

// ------------------------------------------------------------------------- inverse iteration + CholQR
__device__ __forceinline__ double hash_uniform(unsigned a, unsigned b) {
  unsigned x = a * 0x9E3779B1u + b * 0x85EBCA77u + 0x165667B1u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return (double)x * (2.0 / 4294967296.0) - 1.0 + 1.1e-10;  // never exactly zero
}

// One workgroup (512 threads) per matrix.  Threads c < k factor T - lam_c I (pivoted, dgttrf order) and solve;
// the whole workgroup orthonormalises the block: S = Z^T Z (f64 MFMA from global), Cholesky in LDS,
// Z <- Z L^-T row by row.
__global__ void __launch_bounds__(512) trd_invit_kernel(TrdDesc* __restrict__ desc, TrdWork w) {
  TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, k = d.k, kp = w.kp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const double* dd = w.d + b * w.n_max;
  const double* ee = w.e + b * w.n_max;
  double* Z = w.Z + b * w.n_max * kp;
  const int64_t plane = (int64_t)w.n_max * kp;
  double* DL = w.lu + b * 4 * plane;
  double* DI = DL + plane;
  double* DU = DI + plane;
  double* DU2 = DU + plane;
  unsigned char* PV = w.piv + b * plane;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Ls = lds;  // [k16][k16 + 1] Cholesky factor
  const int k16 = (k + 15) & ~15, ldl = k16 + 1;
  const double bound = w.bound[b];
  const double inv = bound > 0.0 ? 1.0 / bound : 0.0;
  constexpr double kEps = 2.220446049250313e-16;

  // ---- factorisation, one thread per shift (scaled T: |T| <= 1)
  if (tid < k) {
    const int c = tid;
    const double mu = w.lam[b * w.n_max + c];
    double di = dd[0] * inv - mu;            // current diagonal
    double ui = n > 1 ? ee[0] * inv : 0.0;   // current super-diagonal du[i]
    for (int i = 0; i + 1 < n; ++i) {
      const double li = ee[i] * inv;                          // sub-diagonal below row i
      const double dn = dd[i + 1] * inv - mu;                 // d[i + 1] before elimination
      const double un = i + 2 < n ? ee[i + 1] * inv : 0.0;    // du[i + 1] before elimination
      const bool swap = fabs(di) < fabs(li);
      double piv = swap ? li : di;
      if (piv == 0.0) piv = kEps;
      const double rp = fast_rcp<2>(piv);
      const double fact = (swap ? di : li) * rp;
      const double du_i = swap ? dn : ui;
      const double up = swap ? ui : dn;
      const int64_t o = (int64_t)i * kp + c;
      DL[o] = fact;
      DI[o] = rp;
      DU[o] = du_i;
      DU2[o] = swap ? un : 0.0;
      PV[o] = swap ? 1 : 0;
      di = fma(-fact, du_i, up);
      ui = swap ? -fact * un : un;
    }
    if (di == 0.0) di = kEps;
    DI[(int64_t)(n - 1) * kp + c] = fast_rcp<2>(di);
  }
  // ---- random start (all columns of the padded block: the pad stays zero)
  for (int e = tid; e < n * kp; e += 512) {
    const int i = e / kp, c = e % kp;
    Z[e] = c < k ? hash_uniform((unsigned)i, (unsigned)c) : 0.0;
  }
  __syncthreads();

  for (int it = 0; it < kInvIters; ++it) {
    // ---- solve (T - mu I) x = z, normalise
    if (tid < k) {
      const int c = tid;
      double xi = Z[c];
      for (int i = 0; i + 1 < n; ++i) {
        const int64_t o = (int64_t)i * kp + c;
        const double xn = Z[o + kp];
        const bool swap = PV[o] != 0;
        const double top = swap ? xn : xi;
        const double bot = swap ? xi : xn;
        Z[o] = top;
        xi = fma(-DL[o], top, bot);
      }
      double x2 = 0.0, x1;
      x1 = xi * DI[(int64_t)(n - 1) * kp + c];
      Z[(int64_t)(n - 1) * kp + c] = x1;
      double ss = x1 * x1;
      for (int i = n - 2; i >= 0; --i) {
        const int64_t o = (int64_t)i * kp + c;
        const double x0 = (Z[o] - DU[o] * x1 - DU2[o] * x2) * DI[o];
        Z[o] = x0;
        ss = fma(x0, x0, ss);
        x2 = x1;
        x1 = x0;
      }
      const double sc = fast_rsqrt<2>(ss);
      for (int i = 0; i < n; ++i) Z[(int64_t)i * kp + c] *= sc;
    }
    __syncthreads();
    // ---- S = Z^T Z, 16 x 16 tiles (ta <= tb) on f64 MFMA, operands straight from global / L2
    const int nt = k16 / 16, ntiles = nt * (nt + 1) / 2;
    for (int tile = wave; tile < ntiles; tile += 8) {
      int ta = 0, u = tile;
      while (u >= nt - ta) {
        u -= nt - ta;
        ++ta;
      }
      const int tb = ta + u;
      const int li = lane & 15, lk = lane >> 4;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      const double* za = Z + ta * 16 + li;
      const double* zb = Z + tb * 16 + li;
      int i0 = 0;
      for (; i0 + 4 <= n; i0 += 4) {
        const int64_t o = (int64_t)(i0 + lk) * kp;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(za[o], zb[o], acc, 0, 0, 0);
      }
      if (i0 < n) {
        const bool ok = i0 + lk < n;
        const int64_t o = (int64_t)(ok ? i0 + lk : 0) * kp;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ok ? za[o] : 0.0, ok ? zb[o] : 0.0, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ra = ta * 16 + lk + 4 * r, cb = tb * 16 + li;
        Ls[ra * ldl + cb] = acc[r];
        Ls[cb * ldl + ra] = acc[r];
      }
    }
    __syncthreads();
    // ---- Cholesky S = L L^T, left-looking by columns (lower triangle of Ls), rows >= k untouched
    for (int jc = 0; jc < k; ++jc) {
      double v = 0.0;
      if (tid >= jc && tid < k) {
        v = Ls[tid * ldl + jc];
        for (int pz = 0; pz < jc; ++pz) v = fma(-Ls[tid * ldl + pz], Ls[jc * ldl + pz], v);
        Ls[tid * ldl + jc] = v;
      }
      __syncthreads();
      const double piv = Ls[jc * ldl + jc];
      __syncthreads();
      if (tid >= jc && tid < k) {
        if (piv > 0.0) {
          const double rs = fast_rsqrt<2>(piv);
          Ls[tid * ldl + jc] = tid == jc ? piv * rs : v * rs;
        } else {  // breakdown: two columns parallel to working precision
          Ls[tid * ldl + jc] = tid == jc ? 1.0 : 0.0;
          if (tid == jc) d.status = 1;
        }
      }
      __syncthreads();
    }
    // ---- Z <- Z L^-T : row i, z_c = (z_c - sum_{a<c} z_a L[c][a]) / L[c][c]; 32-column register blocks
    for (int i = tid; i < n; i += 512) {
      double* zr = Z + (int64_t)i * kp;
      for (int cb = 0; cb < k; cb += 32) {
        double zz[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) zz[u] = cb + u < k ? zr[cb + u] : 0.0;
        for (int a = 0; a < cb; ++a) {
          const double za = zr[a];
#pragma unroll
          for (int u = 0; u < 32; ++u)
            if (cb + u < k) zz[u] = fma(-za, Ls[(cb + u) * ldl + a], zz[u]);
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
          if (cb + u < k) {
            double v = zz[u];
#pragma unroll
            for (int a = 0; a < u; ++a) v = fma(-zz[a], Ls[(cb + u) * ldl + cb + a], v);
            v *= fast_rcp<2>(Ls[(cb + u) * ldl + cb + u]);
            zz[u] = v;
          }
        }
#pragma unroll
        for (int u = 0; u < 32; ++u)
          if (cb + u < k) zr[cb + u] = zz[u];
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------ back-transform
// V = H_0 H_1 ... H_{n-2} Z on k columns.  A column lives in the registers of SEG lanes (row i in lane
// i mod SEG, slot i / SEG), so v^T z is a shuffle reduction and a reflector costs no barrier and no LDS.
// grid (ceil(k / (256 / SEG)), B), 256 threads.
template <int SEG, int R>
__global__ void __launch_bounds__(256) trd_back_kernel(const TrdDesc* __restrict__ desc, TrdWork w) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, k = d.k, kp = w.kp, lda = w.lda;
  const int tid = threadIdx.x;
  const int seg = tid % SEG;
  const int c = blockIdx.x * (256 / SEG) + tid / SEG;
  if (blockIdx.x * (256 / SEG) >= k) return;
  const bool live = c < k;
  const int64_t b = blockIdx.y;
  const double* Z = w.Z + b * w.n_max * kp;
  const double* Vh = w.Vh + b * w.n_max * lda;
  const double* tt = w.tau + b * w.n_max;
  double x[R], v[R], vn[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = seg + SEG * r;
    x[r] = (live && i < n) ? Z[(int64_t)i * kp + c] : 0.0;
  }
  int j = n - 2;
  if (j >= 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = seg + SEG * r;
      vn[r] = i < n ? Vh[(int64_t)j * lda + i] : 0.0;
    }
  }
  for (; j >= 0; --j) {
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = vn[r];
    if (j >= 1) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = seg + SEG * r;
        vn[r] = i < n ? Vh[(int64_t)(j - 1) * lda + i] : 0.0;
      }
    }
    const double tau = tt[j];
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) s = fma(v[r], x[r], s);
#pragma unroll
    for (int off = SEG / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    s *= tau;
#pragma unroll
    for (int r = 0; r < R; ++r) x[r] = fma(-s, v[r], x[r]);
  }
  // sign convention of the library: the largest-magnitude component is positive
  double best = 0.0, val = 0.0;
  int bi = 0x7fffffff;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = seg + SEG * r;
    if (i < n && fabs(x[r]) > best) {
      best = fabs(x[r]);
      val = x[r];
      bi = i;
    }
  }
#pragma unroll
  for (int off = SEG / 2; off > 0; off >>= 1) {
    const double ob = __shfl_xor(best, off, 64), ov = __shfl_xor(val, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ob > best || (ob == best && oi < bi)) {
      best = ob;
      val = ov;
      bi = oi;
    }
  }
  const double sg = val < 0.0 ? -1.0 : 1.0;
  if (live) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = seg + SEG * r;
      if (i < n) d.V_out[(int64_t)i * n + c] = sg * x[r];
    }
  }
}

// Descriptors and ranks reach the device as kernel arguments (copied at launch): no pageable-memory copy
// whose source would have to outlive the call, hence no synchronisation.
constexpr int kDescChunk = 64;
struct DescChunk {
  TrdDesc v[kDescChunk];
};
struct RankChunk {
  int v[256];
};
__global__ void trd_setdesc_kernel(TrdDesc* __restrict__ desc, DescChunk chunk, int base, int count) {
  const int t = threadIdx.x;
  if (t < count) desc[base + t] = chunk.v[t];
}
__global__ void trd_setk_kernel(TrdDesc* __restrict__ desc, RankChunk chunk, int base, int count) {
  const int t = threadIdx.x;
  if (t < count) {
    desc[base + t].k = chunk.v[t];
    desc[base + t].status = 0;
  }
}

// ------------------------------------------------------------------------------------------ host side
struct TrdLayout {
  int64_t n_max, lda, kp;
  int64_t off_a, off_vh, off_y, off_tau, off_d, off_e, off_lam, off_bound, off_z, off_lu, off_piv, off_desc, total;
};

TrdLayout trd_layout(int64_t n_max, int64_t batch, int64_t k_max) {
  TrdLayout l;
  l.n_max = n_max;
  l.lda = ndmps::round_up(n_max, 2);
  l.kp = ndmps::round_up(std::max<int64_t>(k_max, 1), 16);
  int64_t used = 0;
  auto take = [&](int64_t bytes) {
    const int64_t off = ndmps::round_up(used, 256);
    used = off + bytes;
    return off;
  };
  l.off_a = take(batch * n_max * l.lda * 8);
  l.off_vh = take(batch * n_max * l.lda * 8);
  l.off_y = take(batch * 2 * l.lda * 8);
  l.off_tau = take(batch * n_max * 8);
  l.off_d = take(batch * n_max * 8);
  l.off_e = take(batch * n_max * 8);
  l.off_lam = take(batch * n_max * 8);
  l.off_bound = take(batch * 8);
  l.off_z = take(batch * n_max * l.kp * 8);
  l.off_lu = take(batch * 4 * n_max * l.kp * 8);
  l.off_piv = take(batch * n_max * l.kp);
  l.off_desc = take(batch * (int64_t)sizeof(TrdDesc));
  l.total = ndmps::round_up(used, 256);
  return l;
}

TrdWork trd_work(const TrdLayout& l, void* d_ws) {
  char* base = (char*)d_ws;
  TrdWork w;
  w.A = (double*)(base + l.off_a);
  w.Vh = (double*)(base + l.off_vh);
  w.y = (double*)(base + l.off_y);
  w.tau = (double*)(base + l.off_tau);
  w.d = (double*)(base + l.off_d);
  w.e = (double*)(base + l.off_e);
  w.lam = (double*)(base + l.off_lam);
  w.bound = (double*)(base + l.off_bound);
  w.Z = (double*)(base + l.off_z);
  w.lu = (double*)(base + l.off_lu);
  w.piv = (unsigned char*)(base + l.off_piv);
  w.n_max = (int)l.n_max;
  w.lda = (int)l.lda;
  w.kp = (int)l.kp;
  return w;
}

constexpr size_t kTailLds = ((size_t)kTail * kTailLd + 4 * kTail) * sizeof(double);

// kernels that need more than 64 KB of dynamic LDS are opted in once per device
int trd_opt_in() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64 || done[dev]) return NDMPS_OK;
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_tail_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTailLds));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_column_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * kMaxN * 8));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_invit_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kMaxK * (kMaxK + 1) * 8));
  done[dev] = true;
  return NDMPS_OK;
}

int trd_check_sizes(int batch, const int64_t* h_n, int64_t& n_max) {
  NDMPS_REQUIRE(batch >= 1 && batch <= 4096, "batch=%d outside [1, 4096]", batch);
  NDMPS_REQUIRE(h_n, "NULL size array");
  n_max = 0;
  for (int b = 0; b < batch; ++b) {
    NDMPS_REQUIRE(h_n[b] >= 1 && h_n[b] <= kMaxN, "eigen size n=%lld outside [1, %d]", (long long)h_n[b], kMaxN);
    n_max = std::max(n_max, h_n[b]);
  }
  return NDMPS_OK;
}

}  // namespace

extern "C" int64_t ndmps_syevd_topk_max_n(void) { return kMaxN; }
extern "C" int64_t ndmps_syevd_topk_max_k(void) { return kMaxK; }

extern "C" int64_t ndmps_syevd_topk_workspace_bytes(int64_t n_max, int batch, int64_t k_max) {
  if (n_max <= 0 || n_max > kMaxN || batch <= 0 || k_max <= 0 || k_max > kMaxK) return 0;
  return trd_layout(n_max, batch, std::min(k_max, n_max)).total;
}

// Phase 1: tridiagonalise and deliver all eigenvalues (descending) in d_w.  Asynchronous on `stream`.
extern "C" int ndmps_syevd_topk_values_f64(int batch, const double* d_G, int64_t stride_G, const int64_t* h_n,
                                           double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                           int64_t k_max, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  int64_t n_max = 0;
  NDMPS_TRY(trd_check_sizes(batch, h_n, n_max));
  NDMPS_REQUIRE(d_G && d_V && d_w, "NULL eigen operand");
  NDMPS_REQUIRE(k_max >= 1 && k_max <= kMaxK, "k_max=%lld outside [1, %d]", (long long)k_max, kMaxK);
  for (int b = 0; b < batch; ++b)
    NDMPS_REQUIRE(stride_G >= h_n[b] * h_n[b] && stride_V >= h_n[b] * h_n[b] && stride_w >= h_n[b],
                  "batch stride smaller than a matrix");
  const TrdLayout l = trd_layout(n_max, batch, std::min(k_max, n_max));
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevd_topk workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  NDMPS_TRY(trd_opt_in());
  hipStream_t s = (hipStream_t)stream;
  TrdWork w = trd_work(l, d_ws);
  TrdDesc* desc = (TrdDesc*)((char*)d_ws + l.off_desc);
  for (int base = 0; base < batch; base += kDescChunk) {
    DescChunk chunk;
    const int count = std::min(kDescChunk, batch - base);
    for (int t = 0; t < count; ++t) {
      const int b = base + t;
      chunk.v[t].G_in = d_G + b * stride_G;
      chunk.v[t].V_out = d_V + b * stride_V;
      chunk.v[t].w_out = d_w + b * stride_w;
      chunk.v[t].n = (int)h_n[b];
      chunk.v[t].k = 0;
      chunk.v[t].status = 0;
      chunk.v[t].pad = 0;
    }
    hipLaunchKernelGGL(trd_setdesc_kernel, dim3(1), dim3(kDescChunk), 0, s, desc, chunk, base, count);
  }
  const unsigned B = (unsigned)batch;
  const int load_grid = (int)std::min<int64_t>(ndmps::ceil_div(n_max * l.lda, 256), 512);
  hipLaunchKernelGGL(trd_load_kernel, dim3(load_grid, B), dim3(256), 0, s, desc, w);
  const int W = (int)ndmps::ceil_div(n_max, kColsPerWg);
  const size_t col_lds = (size_t)3 * n_max * sizeof(double);
  for (int j = 0; j < n_max - kTail; ++j)
    hipLaunchKernelGGL(trd_column_kernel, dim3(W, B), dim3(256), col_lds, s, desc, w, j);
  hipLaunchKernelGGL(trd_tail_kernel, dim3(1, B), dim3(512), kTailLds, s, desc, w);
  const int bis_grid = (int)ndmps::ceil_div(n_max * kSect, 256);
  hipLaunchKernelGGL(trd_bisect_kernel, dim3(bis_grid, B), dim3(256), (size_t)n_max * 16, s, desc, w);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

// Phase 2: the first h_k[b] eigenvectors of matrix b into columns 0..k-1 of its V (ld n).  Same workspace,
// untouched since phase 1.  Asynchronous on `stream`; *h_status (may be NULL) is filled only when the call
// synchronises, i.e. when h_status is given.
extern "C" int ndmps_syevd_topk_vectors_f64(int batch, const int64_t* h_n, const int64_t* h_k, int64_t k_max,
                                            void* d_ws, int64_t ws_bytes, int* h_status, ndmps_stream_t stream) {
  int64_t n_max = 0;
  NDMPS_TRY(trd_check_sizes(batch, h_n, n_max));
  NDMPS_REQUIRE(h_k, "NULL rank array");
  NDMPS_REQUIRE(k_max >= 1 && k_max <= kMaxK, "k_max=%lld outside [1, %d]", (long long)k_max, kMaxK);
  const TrdLayout l = trd_layout(n_max, batch, std::min(k_max, n_max));
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevd_topk workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  TrdWork w = trd_work(l, d_ws);
  TrdDesc* desc = (TrdDesc*)((char*)d_ws + l.off_desc);
  int kk = 1;
  for (int b = 0; b < batch; ++b) {
    NDMPS_REQUIRE(h_k[b] >= 1 && h_k[b] <= h_n[b] && h_k[b] <= std::min(k_max, n_max), "k[%d]=%lld out of range", b,
                  (long long)h_k[b]);
    kk = std::max(kk, (int)h_k[b]);
  }
  const unsigned B = (unsigned)batch;
  for (int base = 0; base < batch; base += 256) {
    RankChunk chunk;
    const int count = std::min(256, batch - base);
    for (int t = 0; t < count; ++t) chunk.v[t] = (int)h_k[base + t];
    hipLaunchKernelGGL(trd_setk_kernel, dim3(1), dim3(256), 0, s, desc, chunk, base, count);
  }
  const int k16 = (kk + 15) & ~15;
  hipLaunchKernelGGL(trd_invit_kernel, dim3(1, B), dim3(512), (size_t)k16 * (k16 + 1) * 8, s, desc, w);
  // rows per lane of the back-transform: n <= SEG * R
  const int per32 = (int)ndmps::ceil_div(n_max, 32), per64 = (int)ndmps::ceil_div(n_max, 64);
  if (per32 <= 4)
    hipLaunchKernelGGL((trd_back_kernel<32, 4>), dim3(ndmps::ceil_div(kk, 8), B), dim3(256), 0, s, desc, w);
  else if (per32 <= 8)
    hipLaunchKernelGGL((trd_back_kernel<32, 8>), dim3(ndmps::ceil_div(kk, 8), B), dim3(256), 0, s, desc, w);
  else if (per32 <= 16)
    hipLaunchKernelGGL((trd_back_kernel<32, 16>), dim3(ndmps::ceil_div(kk, 8), B), dim3(256), 0, s, desc, w);
  else if (per32 <= 32)
    hipLaunchKernelGGL((trd_back_kernel<32, 32>), dim3(ndmps::ceil_div(kk, 8), B), dim3(256), 0, s, desc, w);
  else if (per64 <= 32)
    hipLaunchKernelGGL((trd_back_kernel<64, 32>), dim3(ndmps::ceil_div(kk, 4), B), dim3(256), 0, s, desc, w);
  else
    hipLaunchKernelGGL((trd_back_kernel<64, 64>), dim3(ndmps::ceil_div(kk, 4), B), dim3(256), 0, s, desc, w);
  NDMPS_LAUNCH_CHECK();
  if (h_status) {
    std::vector<TrdDesc> host(batch);
    NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), desc, sizeof(TrdDesc) * batch, hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < batch; ++b) h_status[b] = host[b].status;
  }
  return NDMPS_OK;
}
