// Symmetric eigen-decomposition in fp64 by parallel two-sided Jacobi.
//
// Stands in for the LAPACK dgesdd that np.linalg.svd runs inside quimb's from_dense /
// tensor_compress_bond for the reference (core/ndmps.py:74, :104-106): the per-site SVD is
// taken from the small-side Gram matrix (gemm.hip) and this solver.
//
// Algorithm: cyclic Jacobi with the round-robin ("circle") ordering -- n-1 steps per
// sweep, n/2 disjoint rotations per step, all applied at once: G' = J^T G J, V' = V J.
// One kernel per step, one thread per (row pair, column pair) 2x2 block, ping-pong
// buffers so a step never reads what it writes.  The host reads one counter per sweep.
// Latency-bound (n-1 dependent launches per sweep).  This is the simple reference solver
// (ndmps_syevj_simple_f64), kept to cross-check the block solver of eig_block.hip, which is
// what ndmps_syevj_f64 and the sweep use.
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

struct JacobiCtl {
  double tol_conv;  // |a_pq| above this counts as "still rotating"
  double tol_rot;   // |a_pq| at or below this is left alone
  int rotated;
  int pad;
};

__device__ __forceinline__ void pair_of(int k, int step, int n, int& p, int& q) {
  const int m1 = n - 1;
  if (k == 0) {
    p = step % m1;
    q = n - 1;
  } else {
    p = (step + k) % m1;
    q = (step - k + m1) % m1;
  }
}

__device__ __forceinline__ void rotation(double app, double aqq, double apq, double tol_rot, double& c,
                                         double& s, double& t) {
  if (fabs(apq) <= tol_rot) {
    c = 1.0;
    s = 0.0;
    t = 0.0;
    return;
  }
  const double tau = (aqq - app) / (2.0 * apq);
  t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  c = 1.0 / sqrt(1.0 + t * t);
  s = t * c;
}

__global__ void __launch_bounds__(256)
jacobi_init_kernel(const double* __restrict__ G, int n, double* __restrict__ Gp, double* __restrict__ Vp,
                   int np) {
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / np), c = (int)(e % np);
    Gp[e] = (r < n && c < n) ? 0.5 * (G[(int64_t)r * n + c] + G[(int64_t)c * n + r]) : 0.0;
    Vp[e] = (r == c) ? 1.0 : 0.0;
  }
}

__global__ void __launch_bounds__(256) jacobi_scale_kernel(const double* __restrict__ G, int n, JacobiCtl* ctl) {
  __shared__ double red[256];
  double mx = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) mx = fmax(mx, fabs(G[(int64_t)i * n + i]));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    ctl->tol_conv = 1e-15 * red[0];
    ctl->tol_rot = 1e-19 * red[0];
    ctl->rotated = 0;
  }
}

// grid: (ceil(np/2 / 64), np/2); block 64: x -> column pair M, blockIdx.y -> row pair K
__global__ void __launch_bounds__(64)
jacobi_step_kernel(const double* __restrict__ Gin, const double* __restrict__ Vin, double* __restrict__ Gout,
                   double* __restrict__ Vout, int np, int step, JacobiCtl* __restrict__ ctl) {
  const int half = np >> 1;
  const int M = blockIdx.x * 64 + threadIdx.x;
  const int K = blockIdx.y;
  if (M >= half) return;
  int p, q, r, s;
  pair_of(K, step, np, p, q);
  pair_of(M, step, np, r, s);
  const double tol_rot = ctl->tol_rot;
  const int64_t ld = np;

  double c1, s1, t1, c2, s2, t2;
  const double app = Gin[p * ld + p], aqq = Gin[q * ld + q], apq = Gin[p * ld + q];
  const double arr = Gin[r * ld + r], ass = Gin[s * ld + s], ars = Gin[r * ld + s];
  rotation(app, aqq, apq, tol_rot, c1, s1, t1);
  rotation(arr, ass, ars, tol_rot, c2, s2, t2);

  if (K == M) {
    // diagonal block: closed form keeps it exactly diagonal / symmetric
    Gout[p * ld + p] = app - t1 * apq;
    Gout[q * ld + q] = aqq + t1 * apq;
    Gout[p * ld + q] = (s1 == 0.0) ? apq : 0.0;
    Gout[q * ld + p] = (s1 == 0.0) ? apq : 0.0;
    if (fabs(apq) > ctl->tol_conv) atomicAdd(&ctl->rotated, 1);
  } else {
    const double gpr = Gin[p * ld + r], gps = Gin[p * ld + s];
    const double gqr = Gin[q * ld + r], gqs = Gin[q * ld + s];
    // columns (r, s) by rotation 2, then rows (p, q) by rotation 1
    const double xpr = c2 * gpr - s2 * gps, xps = s2 * gpr + c2 * gps;
    const double xqr = c2 * gqr - s2 * gqs, xqs = s2 * gqr + c2 * gqs;
    Gout[p * ld + r] = c1 * xpr - s1 * xqr;
    Gout[p * ld + s] = c1 * xps - s1 * xqs;
    Gout[q * ld + r] = s1 * xpr + c1 * xqr;
    Gout[q * ld + s] = s1 * xps + c1 * xqs;
  }
  // V' = V J : rows p and q of V are just two rows; columns (r, s) rotate
  {
    const double vpr = Vin[p * ld + r], vps = Vin[p * ld + s];
    const double vqr = Vin[q * ld + r], vqs = Vin[q * ld + s];
    Vout[p * ld + r] = c2 * vpr - s2 * vps;
    Vout[p * ld + s] = s2 * vpr + c2 * vps;
    Vout[q * ld + r] = c2 * vqr - s2 * vqs;
    Vout[q * ld + s] = s2 * vqr + c2 * vqs;
  }
}

// rank of every eigenvalue (descending, ties by index) and sign of every eigenvector
__global__ void __launch_bounds__(256)
eig_rank_kernel(const double* __restrict__ G, const double* __restrict__ V, int n, int np,
                int* __restrict__ rank, double* __restrict__ sign, double* __restrict__ w_sorted) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double wi = G[(int64_t)i * np + i];
  int rk = 0;
  for (int j = 0; j < n; ++j) {
    const double wj = G[(int64_t)j * np + j];
    rk += (wj > wi) || (wj == wi && j < i);
  }
  double best = 0.0, sg = 1.0;
  for (int r = 0; r < n; ++r) {
    const double v = V[(int64_t)r * np + i];
    if (fabs(v) > best) {
      best = fabs(v);
      sg = v < 0.0 ? -1.0 : 1.0;
    }
  }
  rank[i] = rk;
  sign[i] = sg;
  w_sorted[rk] = wi;
}

__global__ void __launch_bounds__(256)
eig_gather_kernel(const double* __restrict__ V, int n, int np, const int* __restrict__ rank,
                  const double* __restrict__ sign, double* __restrict__ Vout) {
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / n), c = (int)(e % n);
    Vout[(int64_t)r * n + rank[c]] = sign[c] * V[(int64_t)r * np + c];
  }
}

struct SyevjLayout {
  int64_t np;
  int64_t off_g0, off_g1, off_v0, off_v1, off_ctl, off_rank, off_sign, total;
};

SyevjLayout syevj_layout(int64_t n) {
  SyevjLayout l;
  l.np = n + (n & 1);
  if (l.np < 2) l.np = 2;
  int64_t used = 0;
  const int64_t sq = l.np * l.np;
  auto take = [&](int64_t bytes) {
    int64_t off = ndmps::round_up(used, 256);
    used = off + bytes;
    return off;
  };
  l.off_g0 = take(sq * 8);
  l.off_g1 = take(sq * 8);
  l.off_v0 = take(sq * 8);
  l.off_v1 = take(sq * 8);
  l.off_ctl = take(sizeof(JacobiCtl));
  l.off_rank = take(n * 4);
  l.off_sign = take(n * 8);
  l.total = ndmps::round_up(used, 256);
  return l;
}

constexpr int kMaxSweeps = 40;

}  // namespace

extern "C" int64_t ndmps_syevj_simple_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return syevj_layout(n).total;
}

extern "C" int ndmps_syevj_simple_f64(double* d_G, int64_t n, double* d_V, double* d_w, void* d_ws,
                               int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_G && d_V && d_w, "NULL eigen operand");
  NDMPS_REQUIRE(n >= 1 && n <= 32768, "eigen size n=%lld outside [1, 32768]", (long long)n);
  const SyevjLayout l = syevj_layout(n);
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevj workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)d_ws;
  double* g[2] = {(double*)(base + l.off_g0), (double*)(base + l.off_g1)};
  double* v[2] = {(double*)(base + l.off_v0), (double*)(base + l.off_v1)};
  JacobiCtl* ctl = (JacobiCtl*)(base + l.off_ctl);
  int* rank = (int*)(base + l.off_rank);
  double* sign = (double*)(base + l.off_sign);
  const int np = (int)l.np;

  const int init_grid = (int)std::min<int64_t>(ndmps::ceil_div((int64_t)np * np, 256), 4096);
  hipLaunchKernelGGL(jacobi_init_kernel, dim3(init_grid), dim3(256), 0, s, d_G, (int)n, g[0], v[0], np);
  hipLaunchKernelGGL(jacobi_scale_kernel, dim3(1), dim3(256), 0, s, g[0], np, ctl);
  NDMPS_LAUNCH_CHECK();

  int cur = 0, sweeps = 0, rotated = 1;
  const dim3 grid((unsigned)ndmps::ceil_div(np / 2, 64), (unsigned)(np / 2));
  while (sweeps < kMaxSweeps) {
    NDMPS_CHECK_HIP(hipMemsetAsync(&ctl->rotated, 0, sizeof(int), s));
    for (int step = 0; step < np - 1; ++step) {
      hipLaunchKernelGGL(jacobi_step_kernel, grid, dim3(64), 0, s, g[cur], v[cur], g[cur ^ 1], v[cur ^ 1],
                         np, step, ctl);
      cur ^= 1;
    }
    NDMPS_LAUNCH_CHECK();
    ++sweeps;
    NDMPS_CHECK_HIP(hipMemcpyAsync(&rotated, &ctl->rotated, sizeof(int), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    if (rotated == 0) break;
  }
  if (h_sweeps) *h_sweeps = sweeps;
  if (rotated != 0) {
    ndmps::set_error("Jacobi did not converge in %d sweeps (n=%lld)", kMaxSweeps, (long long)n);
    return NDMPS_ENOCONV;
  }
  hipLaunchKernelGGL(eig_rank_kernel, dim3((unsigned)ndmps::ceil_div(n, 256)), dim3(256), 0, s, g[cur],
                     v[cur], (int)n, np, rank, sign, d_w);
  const int gather_grid = (int)std::min<int64_t>(ndmps::ceil_div(n * n, 256), 8192);
  hipLaunchKernelGGL(eig_gather_kernel, dim3(gather_grid), dim3(256), 0, s, v[cur], (int)n, np, rank, sign,
                     d_V);
  NDMPS_LAUNCH_CHECK();
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
