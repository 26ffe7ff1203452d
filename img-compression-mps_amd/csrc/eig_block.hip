// Block two-sided Jacobi eigen-solver (fp64), the production path of ndmps_syevj_f64.
//
// The matrix (padded to a multiple of 32) is cut into 16-wide index blocks.  A sweep visits
// every pair of blocks once (round-robin over blocks, nb-1 outer steps, nb/2 disjoint block
// pairs per step).  Per outer step, two launches:
//
//   diag  : one workgroup per block pair (I, J) keeps the 32x32 diagonal sub-matrix of
//           I u J in LDS, runs the Jacobi rotations of that visit on it (wave-parallel:
//           16 disjoint rotations per inner step, 256 threads each owning one 2x2 block) and
//           accumulates their product Q (32x32).  Outer step 0 of a sweep rotates ALL index
//           pairs inside I u J (31 inner steps, covers the intra-block pairs once per sweep);
//           later steps rotate only the 16x16 cross pairs (16 inner steps) -- every index pair
//           of the matrix is rotated exactly once per sweep, as in the scalar cyclic method.
//   apply : one workgroup per 32x32 tile: G_tile <- Q_A^T G_tile Q_B for every off-diagonal
//           tile (A, B) of block pairs, V_strip <- V_strip Q_B; f64 MFMA (16x16x4), operands
//           from LDS.  A tile is read and written by its own workgroup only, so the update
//           is in place.
//
// Sequential depth per sweep: 2 (nb-1) launches (62 for n = 512) instead of n-1 = 511 for
// the scalar-parallel method in eig.hip (kept as ndmps_syevj_simple_f64 for cross-checks).
// Before the first sweep the matrix is permuted so its diagonal is descending (faster
// convergence on the graded Gram matrices of the sweep).
#include <math.h>

#include <algorithm>

#include "common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int BS = 16;       // block size
constexpr int PS = 2 * BS;   // pair size (LDS sub-problem edge)
constexpr int LD = PS + 1;   // padded LDS row
constexpr int kMaxSweepsBlock = 40;

struct BlockCtl {
  double tol_conv;
  double tol_rot;
  int rotated;
  int pad;
};

// circle-method pairing of `count` players (even), round `step`: pair k -> (a, b)
__device__ __forceinline__ void circle_pair(int k, int step, int count, int& a, int& b) {
  const int m1 = count - 1;
  if (k == 0) {
    a = step % m1;
    b = count - 1;
  } else {
    a = (step + k) % m1;
    b = (step - k + m1) % m1;
  }
}

// 1/sqrt(x) and 1/x for x in a safe range: hardware estimate + two Newton steps (the
// compiler's IEEE expansions of fp64 sqrt/div cost several hundred dependent cycles each,
// and the rotation is the serial part of every inner step).
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * fma(-x, y, 2.0);
  y = y * fma(-x, y, 2.0);
  return y;
}

// Jacobi rotation zeroing a_pq: t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), tau = (aqq-app)/(2apq),
// evaluated as t = sgn(a b) |b| / (|a| + hypot(a, b)) with a = aqq - app, b = 2 apq scaled by a
// power of two (one rsqrt, one rcp), then c = rsqrt(1 + t^2), s = t c.
__device__ __forceinline__ void rotation64(double app, double aqq, double apq, double tol_rot, double& c,
                                           double& s, double& t) {
  if (!(fabs(apq) > tol_rot)) {
    c = 1.0;
    s = 0.0;
    t = 0.0;
    return;
  }
  const double a = aqq - app, b = 2.0 * apq;
  int e;
  (void)frexp(fmax(fabs(a), fabs(b)), &e);
  const double as = ldexp(fabs(a), -e), bs = ldexp(fabs(b), -e);  // max of the two in [0.5, 1)
  const double h2 = fma(as, as, bs * bs);
  const double h = h2 * fast_rsqrt(h2);
  const double mag = bs * fast_rcp(as + h);
  t = ((a < 0.0) != (b < 0.0)) ? -mag : mag;
  c = fast_rsqrt(fma(t, t, 1.0));
  s = t * c;
}

// ---------------------------------------------------------------------------- setup
__global__ void __launch_bounds__(256) blk_scale_kernel(const double* __restrict__ G, int n, BlockCtl* ctl) {
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

// position of every index after sorting the diagonal descending (ties by index)
__global__ void __launch_bounds__(256) blk_order_kernel(const double* __restrict__ G, int n, int* __restrict__ pos) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double di = G[(int64_t)i * n + i];
  int rk = 0;
  for (int j = 0; j < n; ++j) {
    const double dj = G[(int64_t)j * n + j];
    rk += (dj > di) || (dj == di && j < i);
  }
  pos[i] = rk;
}

// Gp = P^T sym(G) P (padded with zeros), Vp = P (so that the accumulated V is P W)
__global__ void __launch_bounds__(256)
blk_init_kernel(const double* __restrict__ G, int n, const int* __restrict__ pos, double* __restrict__ Gp,
                double* __restrict__ Vp, int np) {
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    Gp[e] = 0.0;
    Vp[e] = 0.0;
  }
}
__global__ void __launch_bounds__(256)
blk_scatter_kernel(const double* __restrict__ G, int n, const int* __restrict__ pos, double* __restrict__ Gp,
                   double* __restrict__ Vp, int np) {
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    Gp[(int64_t)pos[r] * np + pos[c]] = 0.5 * (G[(int64_t)r * n + c] + G[(int64_t)c * n + r]);
    if (r == c) Vp[(int64_t)r * np + pos[r]] = 1.0;
  }
}
__global__ void __launch_bounds__(256) blk_pad_identity_kernel(double* __restrict__ Vp, int n, int np) {
  const int i = n + blockIdx.x * 256 + threadIdx.x;
  if (i < np) Vp[(int64_t)i * np + i] = 1.0;
}

// ---------------------------------------------------------------------------- diag phase
// full != 0: all pairs inside the 32 indices (31 inner steps); else the 16x16 cross pairs.
// solve != 0 (single block pair = whole matrix): repeat full sweeps in LDS until converged.
__global__ void __launch_bounds__(256)
blk_diag_kernel(double* __restrict__ G, double* __restrict__ Qbuf, int np, int nb, int outer_step, int full,
                int solve, BlockCtl* __restrict__ ctl) {
  __shared__ double S[PS][LD];
  __shared__ double Q[PS][LD];
  __shared__ double rc[BS], rs[BS], rt[BS];
  __shared__ int cnt;

  const int tid = threadIdx.x;
  int bi, bj;
  circle_pair(blockIdx.x, outer_step, nb, bi, bj);
  if (bi > bj) {
    const int tmp = bi;
    bi = bj;
    bj = tmp;
  }
  const double tol_rot = ctl->tol_rot, tol_conv = ctl->tol_conv;
  if (tid == 0) cnt = 0;
  // load the 32x32 diagonal sub-matrix (symmetrised) and Q = I
  for (int e = tid; e < PS * PS; e += 256) {
    const int a = e / PS, b = e % PS;
    const int64_t ga = (a < BS ? bi * BS + a : bj * BS + a - BS);
    const int64_t gb = (b < BS ? bi * BS + b : bj * BS + b - BS);
    S[a][b] = 0.5 * (G[ga * np + gb] + G[gb * np + ga]);
    Q[a][b] = (a == b) ? 1.0 : 0.0;
  }
  __syncthreads();

  const int K = tid >> 4, M = tid & 15;
  const int n_inner = full ? PS - 1 : BS;
  const int max_rounds = solve ? kMaxSweepsBlock : 1;
  bool converged = false;
  for (int round = 0; round < max_rounds; ++round) {
    int before = 0;
    if (solve) {
      before = cnt;
      __syncthreads();  // nobody may bump cnt for this round before everyone has read it
    }
    for (int st = 0; st < n_inner; ++st) {
      if (tid < BS) {
        int p, q;
        if (full) circle_pair(tid, st, PS, p, q);
        else {
          p = tid;
          q = BS + ((tid + st) & (BS - 1));
        }
        const double apq = S[p][q];
        double c, s, t;
        rotation64(S[p][p], S[q][q], apq, tol_rot, c, s, t);
        rc[tid] = c;
        rs[tid] = s;
        rt[tid] = t;
        if (fabs(apq) > tol_conv) atomicAdd(&cnt, 1);
      }
      __syncthreads();
      {
        int p, q, r, s_;
        if (full) {
          circle_pair(K, st, PS, p, q);
          circle_pair(M, st, PS, r, s_);
        } else {
          p = K;
          q = BS + ((K + st) & (BS - 1));
          r = M;
          s_ = BS + ((M + st) & (BS - 1));
        }
        const double c1 = rc[K], s1 = rs[K], c2 = rc[M], s2 = rs[M];
        if (K == M) {
          const double app = S[p][p], aqq = S[q][q], apq = S[p][q], t1 = rt[K];
          S[p][p] = app - t1 * apq;
          S[q][q] = aqq + t1 * apq;
          const double off = (s1 == 0.0) ? apq : 0.0;
          S[p][q] = off;
          S[q][p] = off;
        } else {
          const double gpr = S[p][r], gps = S[p][s_], gqr = S[q][r], gqs = S[q][s_];
          const double xpr = c2 * gpr - s2 * gps, xps = s2 * gpr + c2 * gps;
          const double xqr = c2 * gqr - s2 * gqs, xqs = s2 * gqr + c2 * gqs;
          S[p][r] = c1 * xpr - s1 * xqr;
          S[p][s_] = c1 * xps - s1 * xqs;
          S[q][r] = s1 * xpr + c1 * xqr;
          S[q][s_] = s1 * xps + c1 * xqs;
        }
        // Q <- Q J: rows 2K, 2K+1 (static), columns (r, s_)
        const int r0 = 2 * K, r1 = 2 * K + 1;
        const double q0r = Q[r0][r], q0s = Q[r0][s_], q1r = Q[r1][r], q1s = Q[r1][s_];
        Q[r0][r] = c2 * q0r - s2 * q0s;
        Q[r0][s_] = s2 * q0r + c2 * q0s;
        Q[r1][r] = c2 * q1r - s2 * q1s;
        Q[r1][s_] = s2 * q1r + c2 * q1s;
      }
      __syncthreads();
    }
    if (solve && cnt == before) {  // cnt is stable here: last write was before the barrier
      converged = true;
      break;
    }
  }

  // write back the rotated diagonal tile and Q
  for (int e = tid; e < PS * PS; e += 256) {
    const int a = e / PS, b = e % PS;
    const int64_t ga = (a < BS ? bi * BS + a : bj * BS + a - BS);
    const int64_t gb = (b < BS ? bi * BS + b : bj * BS + b - BS);
    G[ga * np + gb] = S[a][b];
    Qbuf[(int64_t)blockIdx.x * PS * PS + e] = Q[a][b];
  }
  if (tid == 0) {
    if (!solve && cnt > 0) atomicAdd(&ctl->rotated, cnt);
    if (solve && !converged) atomicAdd(&ctl->rotated, 1);
  }
}

// ---------------------------------------------------------------------------- apply phase
// C(32x32) = op(A) * B in LDS, f64 MFMA; 4 waves, one 16x16 output tile each.
// TRANS_A: A given as (k, i) (i.e. C = A^T B).
template <bool TRANS_A>
__device__ __forceinline__ void lds_gemm32(const double (*A)[LD], const double (*B)[LD], double (*Cout)[LD],
                                           int wave, int lane) {
  const int i0 = (wave >> 1) * 16, j0 = (wave & 1) * 16;
  const int li = lane & 15, lk = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k0 = 0; k0 < PS; k0 += 4) {
    const double a = TRANS_A ? A[k0 + lk][i0 + li] : A[i0 + li][k0 + lk];
    const double b = B[k0 + lk][j0 + li];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Cout[i0 + lk + 4 * r][j0 + li] = acc[r];
}

// blockIdx.x < half*half : G tile (A, B); else V strip tile (R, B)
__global__ void __launch_bounds__(256)
blk_apply_kernel(double* __restrict__ G, double* __restrict__ V, const double* __restrict__ Qbuf, int np, int nb,
                 int outer_step) {
  __shared__ double T[PS][LD];
  __shared__ double QA[PS][LD];
  __shared__ double QB[PS][LD];
  __shared__ double X[PS][LD];
  const int half = nb >> 1;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int bid = blockIdx.x;
  const bool is_g = bid < half * half;
  int pa = 0, pb, strip = 0;
  if (is_g) {
    pa = bid / half;
    pb = bid % half;
    if (pa == pb) return;  // diagonal tiles are written by the diag phase
  } else {
    bid -= half * half;
    strip = bid / half;
    pb = bid % half;
  }
  int bi_b, bj_b;
  circle_pair(pb, outer_step, nb, bi_b, bj_b);
  if (bi_b > bj_b) {
    const int tmp = bi_b;
    bi_b = bj_b;
    bj_b = tmp;
  }
  int bi_a = 0, bj_a = 0;
  if (is_g) {
    circle_pair(pa, outer_step, nb, bi_a, bj_a);
    if (bi_a > bj_a) {
      const int tmp = bi_a;
      bi_a = bj_a;
      bj_a = tmp;
    }
  }
  double* M = is_g ? G : V;
  for (int e = tid; e < PS * PS; e += 256) {
    const int a = e / PS, b = e % PS;
    const int64_t gr = is_g ? (a < BS ? bi_a * BS + a : bj_a * BS + a - BS) : (int64_t)strip * PS + a;
    const int64_t gc = (b < BS ? bi_b * BS + b : bj_b * BS + b - BS);
    T[a][b] = M[gr * np + gc];
    QB[a][b] = Qbuf[(int64_t)pb * PS * PS + e];
    if (is_g) QA[a][b] = Qbuf[(int64_t)pa * PS * PS + e];
  }
  __syncthreads();
  lds_gemm32<false>(T, QB, X, wave, lane);  // X = T Q_B
  __syncthreads();
  if (is_g) {
    lds_gemm32<true>(QA, X, T, wave, lane);  // T = Q_A^T X
    __syncthreads();
  }
  for (int e = tid; e < PS * PS; e += 256) {
    const int a = e / PS, b = e % PS;
    const int64_t gr = is_g ? (a < BS ? bi_a * BS + a : bj_a * BS + a - BS) : (int64_t)strip * PS + a;
    const int64_t gc = (b < BS ? bi_b * BS + b : bj_b * BS + b - BS);
    M[gr * np + gc] = is_g ? T[a][b] : X[a][b];
  }
}

// ---------------------------------------------------------------------------- finish
__global__ void __launch_bounds__(256)
blk_rank_kernel(const double* __restrict__ G, const double* __restrict__ V, int n, int np, int* __restrict__ rank,
                double* __restrict__ sign, double* __restrict__ w_sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
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
blk_gather_kernel(const double* __restrict__ V, int n, int np, const int* __restrict__ rank,
                  const double* __restrict__ sign, double* __restrict__ Vout) {
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    Vout[(int64_t)r * n + rank[c]] = sign[c] * V[(int64_t)r * np + c];
  }
}

struct BlockLayout {
  int64_t np, nb;
  int64_t off_g, off_v, off_q, off_ctl, off_pos, off_sign, total;
};

BlockLayout block_layout(int64_t n) {
  BlockLayout l;
  l.np = std::max<int64_t>(ndmps::round_up(n, PS), PS);
  l.nb = l.np / BS;
  int64_t used = 0;
  auto take = [&](int64_t bytes) {
    const int64_t off = ndmps::round_up(used, 256);
    used = off + bytes;
    return off;
  };
  l.off_g = take(l.np * l.np * 8);
  l.off_v = take(l.np * l.np * 8);
  l.off_q = take((l.nb / 2) * PS * PS * 8);
  l.off_ctl = take(sizeof(BlockCtl));
  l.off_pos = take(l.np * 4);
  l.off_sign = take(l.np * 8);
  l.total = ndmps::round_up(used, 256);
  return l;
}

}  // namespace

extern "C" int64_t ndmps_syevj_workspace_bytes(int64_t n) {
  if (n <= 0) return 0;
  return block_layout(n).total;
}

extern "C" int ndmps_syevj_f64(double* d_G, int64_t n, double* d_V, double* d_w, void* d_ws,
                               int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_G && d_V && d_w, "NULL eigen operand");
  NDMPS_REQUIRE(n >= 1 && n <= 32768, "eigen size n=%lld outside [1, 32768]", (long long)n);
  const BlockLayout l = block_layout(n);
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevj workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)d_ws;
  double* G = (double*)(base + l.off_g);
  double* V = (double*)(base + l.off_v);
  double* Qbuf = (double*)(base + l.off_q);
  BlockCtl* ctl = (BlockCtl*)(base + l.off_ctl);
  int* pos = (int*)(base + l.off_pos);
  double* sign = (double*)(base + l.off_sign);
  const int np = (int)l.np, nb = (int)l.nb, half = nb / 2;
  const int ni = (int)n;

  const int fill_grid = (int)std::min<int64_t>(ndmps::ceil_div((int64_t)np * np, 256), 4096);
  hipLaunchKernelGGL(blk_scale_kernel, dim3(1), dim3(256), 0, s, d_G, ni, ctl);
  hipLaunchKernelGGL(blk_order_kernel, dim3((unsigned)ndmps::ceil_div(n, 256)), dim3(256), 0, s, d_G, ni, pos);
  hipLaunchKernelGGL(blk_init_kernel, dim3(fill_grid), dim3(256), 0, s, d_G, ni, pos, G, V, np);
  hipLaunchKernelGGL(blk_scatter_kernel, dim3(fill_grid), dim3(256), 0, s, d_G, ni, pos, G, V, np);
  if (np > ni)
    hipLaunchKernelGGL(blk_pad_identity_kernel, dim3((unsigned)ndmps::ceil_div(np - ni, 256)), dim3(256), 0, s, V, ni, np);
  NDMPS_LAUNCH_CHECK();

  int sweeps = 0, rotated = 1;
  if (nb == 2) {
    // whole matrix is one block pair: solve it in LDS in a single launch
    hipLaunchKernelGGL(blk_diag_kernel, dim3(1), dim3(256), 0, s, G, Qbuf, np, nb, 0, 1, 1, ctl);
    hipLaunchKernelGGL(blk_apply_kernel, dim3(half * half + (np / PS) * half), dim3(256), 0, s, G, V, Qbuf, np, nb, 0);
    NDMPS_LAUNCH_CHECK();
    sweeps = 1;
    NDMPS_CHECK_HIP(hipMemcpyAsync(&rotated, &ctl->rotated, sizeof(int), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  } else {
    const int apply_grid = half * half + (np / PS) * half;
    while (sweeps < kMaxSweepsBlock) {
      NDMPS_CHECK_HIP(hipMemsetAsync(&ctl->rotated, 0, sizeof(int), s));
      for (int step = 0; step < nb - 1; ++step) {
        hipLaunchKernelGGL(blk_diag_kernel, dim3(half), dim3(256), 0, s, G, Qbuf, np, nb, step, step == 0 ? 1 : 0, 0, ctl);
        hipLaunchKernelGGL(blk_apply_kernel, dim3(apply_grid), dim3(256), 0, s, G, V, Qbuf, np, nb, step);
      }
      NDMPS_LAUNCH_CHECK();
      ++sweeps;
      NDMPS_CHECK_HIP(hipMemcpyAsync(&rotated, &ctl->rotated, sizeof(int), hipMemcpyDeviceToHost, s));
      NDMPS_CHECK_HIP(hipStreamSynchronize(s));
      if (rotated == 0) break;
    }
  }
  if (h_sweeps) *h_sweeps = sweeps;
  if (rotated != 0) {
    ndmps::set_error("block Jacobi did not converge in %d sweeps (n=%lld)", kMaxSweepsBlock, (long long)n);
    return NDMPS_ENOCONV;
  }
  hipLaunchKernelGGL(blk_rank_kernel, dim3((unsigned)ndmps::ceil_div(n, 256)), dim3(256), 0, s, G, V, ni, np, pos,
                     sign, d_w);
  const int gather_grid = (int)std::min<int64_t>(ndmps::ceil_div(n * n, 256), 8192);
  hipLaunchKernelGGL(blk_gather_kernel, dim3(gather_grid), dim3(256), 0, s, V, ni, np, pos, sign, d_V);
  NDMPS_LAUNCH_CHECK();
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
