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
// Batched: every kernel takes a batch of matrices (blockIdx.y); matrices of different size are
// padded to the common np (padded indices never rotate) and converged ones are skipped, so B
// eigenproblems cost the sequential depth of one.
//
// Sequential depth per sweep: 2 (nb-1) launches (62 for n = 512) instead of n-1 = 511 for
// the scalar-parallel method in eig.hip (kept as ndmps_syevj_simple_f64 for cross-checks).
// Before the first sweep the matrix is permuted so its diagonal is descending (faster
// convergence on the graded Gram matrices of the sweep).
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int BS = 16;       // block size
constexpr int PS = 2 * BS;   // pair size (LDS sub-problem edge)
constexpr int LD = PS + 1;   // padded LDS row
constexpr int kMaxSweepsBlock = 40;

// one entry per matrix of the batch (device array); blockIdx.y selects it in every kernel
struct BatchDesc {
  const double* G_in;  // n x n input (ld n)
  double* V_out;       // n x n eigenvectors (columns), sorted
  double* w_out;       // n eigenvalues, descending
  int n;
  int done;            // converged: later launches skip this matrix
  int rotated;         // rotations above tol_conv in the current sweep
  int pad;
  double tol_conv;
  double tol_rot;
};

struct Work {          // common padded working set, strides per matrix
  double* G;           // [B][np][np]
  double* V;           // [B][np][np]
  double* Q;           // [B][nb/2][PS][PS]
  int* pos;            // [B][np]
  double* sign;        // [B][np]
  int np;
  int nb;
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
// Measured on gfx950: v_rcp_f64 / v_rsq_f64 estimates are good to 5e-8, one Newton step gives
// 2-4e-15, two give 2e-16.  One step is enough where only the rotation ANGLE depends on it;
// c = rsqrt(1 + t^2) fixes the normalisation c^2 + s^2 = 1 and gets two.
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
  const double h = h2 * fast_rsqrt<1>(h2);
  const double mag = bs * fast_rcp<1>(as + h);
  t = ((a < 0.0) != (b < 0.0)) ? -mag : mag;
  c = fast_rsqrt<2>(fma(t, t, 1.0));
  s = t * c;
}

// ---------------------------------------------------------------------------- setup
__global__ void __launch_bounds__(256) blk_scale_kernel(BatchDesc* __restrict__ desc) {
  BatchDesc& d = desc[blockIdx.y];
  const double* G = d.G_in;
  const int n = d.n;
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
    d.tol_conv = 1e-15 * red[0];
    d.tol_rot = 1e-19 * red[0];
    d.rotated = 0;
    d.done = 0;
  }
}

// position of every index after sorting the diagonal descending (ties by index)
__global__ void __launch_bounds__(256) blk_order_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const double* G = d.G_in;
  const int n = d.n;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double di = G[(int64_t)i * n + i];
  int rk = 0;
  for (int j = 0; j < n; ++j) {
    const double dj = G[(int64_t)j * n + j];
    rk += (dj > di) || (dj == di && j < i);
  }
  w.pos[(int64_t)blockIdx.y * w.np + i] = rk;
}

// Gp = P^T sym(G) P (padded with zeros), Vp = P on the real indices, identity on the padding
__global__ void __launch_bounds__(256) blk_init_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np;
  double* Gp = w.G + (int64_t)blockIdx.y * np * np;
  double* Vp = w.V + (int64_t)blockIdx.y * np * np;
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / np), c = (int)(e % np);
    Gp[e] = 0.0;
    Vp[e] = (r == c && r >= n) ? 1.0 : 0.0;
  }
}
__global__ void __launch_bounds__(256) blk_scatter_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const double* G = d.G_in;
  const int n = d.n, np = w.np;
  double* Gp = w.G + (int64_t)blockIdx.y * np * np;
  double* Vp = w.V + (int64_t)blockIdx.y * np * np;
  const int* pos = w.pos + (int64_t)blockIdx.y * np;
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    Gp[(int64_t)pos[r] * np + pos[c]] = 0.5 * (G[(int64_t)r * n + c] + G[(int64_t)c * n + r]);
    if (r == c) Vp[(int64_t)r * np + pos[r]] = 1.0;
  }
}

// between sweeps: a matrix whose last sweep rotated nothing above tol_conv is done
__global__ void blk_check_kernel(BatchDesc* __restrict__ desc, int batch, int* __restrict__ remaining) {
  __shared__ int left;
  if (threadIdx.x == 0) left = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < batch; b += blockDim.x) {
    if (!desc[b].done) {
      if (desc[b].rotated == 0) desc[b].done = 1;
      else atomicAdd(&left, 1);
    }
    desc[b].rotated = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) *remaining = left;
}

// ---------------------------------------------------------------------------- diag phase
// full != 0: all pairs inside the 32 indices (31 inner steps); else the 16x16 cross pairs.
// solve != 0 (single block pair = whole matrix): repeat full sweeps in LDS until converged.
//
// One barrier per inner step: every lane computes the rotation of pair (lane & 15) itself
// (it is its column rotation; the row rotation comes from lane K by a wave shuffle), and S is
// double-buffered in LDS so a step reads only what the previous step wrote.
template <bool STAMP>
__global__ void __launch_bounds__(256)
blk_diag_kernel(BatchDesc* __restrict__ desc, Work w, int outer_step, int full, int solve,
                unsigned long long* __restrict__ stamps) {
  BatchDesc& d = desc[blockIdx.y];
  if (d.done) return;
  const int np = w.np, nb = w.nb;
  double* G = w.G + (int64_t)blockIdx.y * np * np;
  double* Qbuf = w.Q + (int64_t)blockIdx.y * (nb / 2) * PS * PS;
  // STAMP builds are diagnostic only (ndmps_debug_diag_stamps): per-segment s_memtime sums of
  // wave 0 go to `stamps`, which nothing else reads.
  unsigned long long t_rot = 0, t_app = 0, t_bar = 0, t_load = 0, t_store = 0, t0 = 0, t1 = 0;
#define NDMPS_STAMP(var)                                                        \
  if (STAMP) {                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                          \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                          \
  }
  NDMPS_STAMP(t0);
  __shared__ double Sb[2][PS][LD];
  __shared__ double Q[PS][LD];
  __shared__ int cnt;

  const int tid = threadIdx.x, lane = tid & 63;
  int bi, bj;
  circle_pair(blockIdx.x, outer_step, nb, bi, bj);
  if (bi > bj) {
    const int tmp = bi;
    bi = bj;
    bj = tmp;
  }
  const double tol_rot = d.tol_rot, tol_conv = d.tol_conv;
  if (tid == 0) cnt = 0;
  // load the 32x32 diagonal sub-matrix (symmetrised) and Q = I
  for (int e = tid; e < PS * PS; e += 256) {
    const int a = e / PS, b = e % PS;
    const int64_t ga = (a < BS ? bi * BS + a : bj * BS + a - BS);
    const int64_t gb = (b < BS ? bi * BS + b : bj * BS + b - BS);
    Sb[0][a][b] = 0.5 * (G[ga * np + gb] + G[gb * np + ga]);
    Q[a][b] = (a == b) ? 1.0 : 0.0;
  }
  __syncthreads();
  NDMPS_STAMP(t1);
  t_load = t1 - t0;

  const int K = tid >> 4, M = tid & 15;
  const int n_inner = full ? PS - 1 : BS;
  const int max_rounds = solve ? kMaxSweepsBlock : 1;
  bool converged = false;
  int cur = 0;
  for (int round = 0; round < max_rounds; ++round) {
    int before = 0;
    if (solve) {
      before = cnt;
      __syncthreads();  // nobody may bump cnt for this round before everyone has read it
    }
    for (int st = 0; st < n_inner; ++st) {
      NDMPS_STAMP(t0);
      double (*S)[LD] = Sb[cur];
      double (*Sn)[LD] = Sb[cur ^ 1];
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
      // own (column) rotation: pair M
      const double arr = S[r][r], ass = S[s_][s_], ars = S[r][s_];
      double c2, s2, t2;
      rotation64(arr, ass, ars, tol_rot, c2, s2, t2);
      // row rotation of pair K lives in lane K of this wave (lane & 15 == K)
      const double c1 = __shfl(c2, K, 64), s1 = __shfl(s2, K, 64);
      if (tid < 64) {  // wave 0 counts the pairs still above the convergence threshold
        const unsigned long long m = __ballot(lane < BS && fabs(ars) > tol_conv);
        if (lane == 0) cnt += __popcll(m);
      }
      NDMPS_STAMP(t1);
      t_rot += t1 - t0;
      {
        const double gpr = S[p][r], gps = S[p][s_], gqr = S[q][r], gqs = S[q][s_];
        const double xpr = c2 * gpr - s2 * gps, xps = s2 * gpr + c2 * gps;
        const double xqr = c2 * gqr - s2 * gqs, xqs = s2 * gqr + c2 * gqs;
        double ypr = c1 * xpr - s1 * xqr, yps = c1 * xps - s1 * xqs;
        double yqr = s1 * xpr + c1 * xqr, yqs = s1 * xps + c1 * xqs;
        if (K == M && s2 != 0.0) {  // the rotated pair itself: off-diagonal annihilated exactly
          yps = 0.0;
          yqr = 0.0;
        }
        Sn[p][r] = ypr;
        Sn[p][s_] = yps;
        Sn[q][r] = yqr;
        Sn[q][s_] = yqs;
        // Q <- Q J: rows 2K, 2K+1 (static), columns (r, s_); own entries only, in place
        const int r0 = 2 * K, r1 = 2 * K + 1;
        const double q0r = Q[r0][r], q0s = Q[r0][s_], q1r = Q[r1][r], q1s = Q[r1][s_];
        Q[r0][r] = c2 * q0r - s2 * q0s;
        Q[r0][s_] = s2 * q0r + c2 * q0s;
        Q[r1][r] = c2 * q1r - s2 * q1s;
        Q[r1][s_] = s2 * q1r + c2 * q1s;
      }
      NDMPS_STAMP(t0);
      t_app += t0 - t1;
      __syncthreads();
      NDMPS_STAMP(t1);
      t_bar += t1 - t0;
      cur ^= 1;
    }
    if (solve && cnt == before) {  // cnt is stable here: last write was before the barrier
      converged = true;
      break;
    }
  }

  // write back the rotated diagonal tile and Q
  NDMPS_STAMP(t0);
  for (int e = tid; e < PS * PS; e += 256) {
    const int a = e / PS, b = e % PS;
    const int64_t ga = (a < BS ? bi * BS + a : bj * BS + a - BS);
    const int64_t gb = (b < BS ? bi * BS + b : bj * BS + b - BS);
    G[ga * np + gb] = Sb[cur][a][b];
    Qbuf[(int64_t)blockIdx.x * PS * PS + e] = Q[a][b];
  }
  if (tid == 0) {
    if (!solve && cnt > 0) atomicAdd(&d.rotated, cnt);
    if (solve && !converged) atomicAdd(&d.rotated, 1);
  }
  if (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    NDMPS_STAMP(t1);
    t_store = t1 - t0;
    if (tid == 0) {
      unsigned long long* o = stamps + 8 * blockIdx.x;
      o[0] = t_load; o[1] = t_rot; o[2] = 0; o[3] = t_app; o[4] = t_bar; o[5] = t_store;
      o[6] = (unsigned long long)n_inner; o[7] = 0;
    }
  }
#undef NDMPS_STAMP
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
blk_apply_kernel(const BatchDesc* __restrict__ desc, Work w, int outer_step) {
  if (desc[blockIdx.y].done) return;
  __shared__ double T[PS][LD];
  __shared__ double QA[PS][LD];
  __shared__ double QB[PS][LD];
  __shared__ double X[PS][LD];
  const int np = w.np, nb = w.nb;
  double* G = w.G + (int64_t)blockIdx.y * np * np;
  double* V = w.V + (int64_t)blockIdx.y * np * np;
  const double* Qbuf = w.Q + (int64_t)blockIdx.y * (nb / 2) * PS * PS;
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
__global__ void __launch_bounds__(256) blk_rank_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np;
  const double* G = w.G + (int64_t)blockIdx.y * np * np;
  const double* V = w.V + (int64_t)blockIdx.y * np * np;
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
  w.pos[(int64_t)blockIdx.y * np + i] = rk;
  w.sign[(int64_t)blockIdx.y * np + i] = sg;
  d.w_out[rk] = wi;
}

__global__ void __launch_bounds__(256) blk_gather_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np;
  const double* V = w.V + (int64_t)blockIdx.y * np * np;
  const int* rank = w.pos + (int64_t)blockIdx.y * np;
  const double* sign = w.sign + (int64_t)blockIdx.y * np;
  double* Vout = d.V_out;
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    Vout[(int64_t)r * n + rank[c]] = sign[c] * V[(int64_t)r * np + c];
  }
}

struct BlockLayout {
  int64_t np, nb;
  int64_t off_g, off_v, off_q, off_desc, off_pos, off_sign, off_flag, off_stamp, total;
};

BlockLayout block_layout(int64_t n_max, int64_t batch) {
  BlockLayout l;
  l.np = std::max<int64_t>(ndmps::round_up(n_max, PS), PS);
  l.nb = l.np / BS;
  int64_t used = 0;
  auto take = [&](int64_t bytes) {
    const int64_t off = ndmps::round_up(used, 256);
    used = off + bytes;
    return off;
  };
  l.off_g = take(batch * l.np * l.np * 8);
  l.off_v = take(batch * l.np * l.np * 8);
  l.off_q = take(batch * (l.nb / 2) * PS * PS * 8);
  l.off_desc = take(batch * (int64_t)sizeof(BatchDesc));
  l.off_pos = take(batch * l.np * 4);
  l.off_sign = take(batch * l.np * 8);
  l.off_flag = take(256);
  l.off_stamp = take((l.nb / 2) * 64);
  l.total = ndmps::round_up(used, 256);
  return l;
}

int solve_batched(int batch, std::vector<BatchDesc>& host_desc, int64_t n_max, void* d_ws, int64_t ws_bytes,
                  int* h_sweeps, hipStream_t s) {
  const BlockLayout l = block_layout(n_max, batch);
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevj workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  char* base = (char*)d_ws;
  Work w;
  w.G = (double*)(base + l.off_g);
  w.V = (double*)(base + l.off_v);
  w.Q = (double*)(base + l.off_q);
  w.pos = (int*)(base + l.off_pos);
  w.sign = (double*)(base + l.off_sign);
  w.np = (int)l.np;
  w.nb = (int)l.nb;
  BatchDesc* desc = (BatchDesc*)(base + l.off_desc);
  int* flag = (int*)(base + l.off_flag);
  const int np = w.np, nb = w.nb, half = nb / 2;
  NDMPS_CHECK_HIP(hipMemcpyAsync(desc, host_desc.data(), sizeof(BatchDesc) * batch, hipMemcpyHostToDevice, s));

  const unsigned B = (unsigned)batch;
  const int fill_grid = (int)std::min<int64_t>(ndmps::ceil_div((int64_t)np * np, 256), 1024);
  hipLaunchKernelGGL(blk_scale_kernel, dim3(1, B), dim3(256), 0, s, desc);
  hipLaunchKernelGGL(blk_order_kernel, dim3((unsigned)ndmps::ceil_div(n_max, 256), B), dim3(256), 0, s, desc, w);
  hipLaunchKernelGGL(blk_init_kernel, dim3(fill_grid, B), dim3(256), 0, s, desc, w);
  hipLaunchKernelGGL(blk_scatter_kernel, dim3(fill_grid, B), dim3(256), 0, s, desc, w);
  NDMPS_LAUNCH_CHECK();

  int sweeps = 0, remaining = 1;
  const int apply_grid = half * half + (np / PS) * half;
  if (nb == 2) {
    // every matrix is one block pair: solved in LDS by a single launch
    hipLaunchKernelGGL(blk_diag_kernel<false>, dim3(1, B), dim3(256), 0, s, desc, w, 0, 1, 1, nullptr);
    hipLaunchKernelGGL(blk_apply_kernel, dim3(apply_grid, B), dim3(256), 0, s, desc, w, 0);
    hipLaunchKernelGGL(blk_check_kernel, dim3(1), dim3(64), 0, s, desc, batch, flag);
    NDMPS_LAUNCH_CHECK();
    sweeps = 1;
    NDMPS_CHECK_HIP(hipMemcpyAsync(&remaining, flag, sizeof(int), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  } else {
    while (sweeps < kMaxSweepsBlock) {
      for (int step = 0; step < nb - 1; ++step) {
        hipLaunchKernelGGL(blk_diag_kernel<false>, dim3(half, B), dim3(256), 0, s, desc, w, step,
                           step == 0 ? 1 : 0, 0, nullptr);
        hipLaunchKernelGGL(blk_apply_kernel, dim3(apply_grid, B), dim3(256), 0, s, desc, w, step);
      }
      hipLaunchKernelGGL(blk_check_kernel, dim3(1), dim3(64), 0, s, desc, batch, flag);
      NDMPS_LAUNCH_CHECK();
      ++sweeps;
      NDMPS_CHECK_HIP(hipMemcpyAsync(&remaining, flag, sizeof(int), hipMemcpyDeviceToHost, s));
      NDMPS_CHECK_HIP(hipStreamSynchronize(s));
      if (remaining == 0) break;
    }
  }
  if (h_sweeps) *h_sweeps = sweeps;
  if (remaining != 0) {
    ndmps::set_error("block Jacobi did not converge in %d sweeps (n=%lld, %d of %d matrices left)",
                     kMaxSweepsBlock, (long long)n_max, remaining, batch);
    return NDMPS_ENOCONV;
  }
  hipLaunchKernelGGL(blk_rank_kernel, dim3((unsigned)ndmps::ceil_div(n_max, 256), B), dim3(256), 0, s, desc, w);
  const int gather_grid = (int)std::min<int64_t>(ndmps::ceil_div(n_max * n_max, 256), 2048);
  hipLaunchKernelGGL(blk_gather_kernel, dim3(gather_grid, B), dim3(256), 0, s, desc, w);
  NDMPS_LAUNCH_CHECK();
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}

}  // namespace

extern "C" int64_t ndmps_syevj_batched_workspace_bytes(int64_t n_max, int batch) {
  if (n_max <= 0 || batch <= 0) return 0;
  return block_layout(n_max, batch).total;
}

extern "C" int64_t ndmps_syevj_workspace_bytes(int64_t n) { return ndmps_syevj_batched_workspace_bytes(n, 1); }

extern "C" int ndmps_syevj_batched_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n, double* d_V,
                                       int64_t stride_V, double* d_w, int64_t stride_w, void* d_ws,
                                       int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream) {
  NDMPS_REQUIRE(batch >= 1 && batch <= 4096, "batch=%d outside [1, 4096]", batch);
  NDMPS_REQUIRE(d_G && d_V && d_w && h_n, "NULL eigen operand");
  std::vector<BatchDesc> desc(batch);
  int64_t n_max = 0;
  for (int b = 0; b < batch; ++b) {
    NDMPS_REQUIRE(h_n[b] >= 1 && h_n[b] <= 32768, "eigen size n=%lld outside [1, 32768]", (long long)h_n[b]);
    NDMPS_REQUIRE(stride_G >= h_n[b] * h_n[b] && stride_V >= h_n[b] * h_n[b] && stride_w >= h_n[b],
                  "batch stride smaller than a matrix");
    n_max = std::max(n_max, h_n[b]);
    memset(&desc[b], 0, sizeof(BatchDesc));
    desc[b].G_in = d_G + b * stride_G;
    desc[b].V_out = d_V + b * stride_V;
    desc[b].w_out = d_w + b * stride_w;
    desc[b].n = (int)h_n[b];
  }
  return solve_batched(batch, desc, n_max, d_ws, ws_bytes, h_sweeps, (hipStream_t)stream);
}

extern "C" int ndmps_syevj_f64(double* d_G, int64_t n, double* d_V, double* d_w, void* d_ws,
                               int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream) {
  NDMPS_REQUIRE(n >= 1, "eigen size n=%lld must be positive", (long long)n);
  return ndmps_syevj_batched_f64(1, d_G, n * n, &n, d_V, n * n, d_w, n, d_ws, ws_bytes, h_sweeps, stream);
}

// Diagnostic (not used by the product path): run ONE diag launch of the stamped build on a padded
// np x np matrix already on the device and return wave-0 s_memtime sums per workgroup:
// [load, rotation, 0, apply, barrier, store, inner_steps, 0] x (np/32).
extern "C" int ndmps_debug_diag_stamps(double* d_Gp, int np, void* d_ws, int64_t ws_bytes,
                                       unsigned long long* h_out, int full, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_Gp && d_ws && h_out && np >= 64 && np % PS == 0, "bad debug argument");
  const BlockLayout l = block_layout(np, 1);
  if (ws_bytes < l.total) return NDMPS_EWORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)d_ws;
  Work w;
  w.G = d_Gp;  // rotate the caller's matrix in place
  w.V = (double*)(base + l.off_v);
  w.Q = (double*)(base + l.off_q);
  w.pos = (int*)(base + l.off_pos);
  w.sign = (double*)(base + l.off_sign);
  w.np = np;
  w.nb = np / BS;
  BatchDesc hd;
  memset(&hd, 0, sizeof(hd));
  hd.G_in = d_Gp;
  hd.n = np;
  BatchDesc* desc = (BatchDesc*)(base + l.off_desc);
  unsigned long long* st = (unsigned long long*)(base + l.off_stamp);
  NDMPS_CHECK_HIP(hipMemcpyAsync(desc, &hd, sizeof(hd), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(blk_scale_kernel, dim3(1, 1), dim3(256), 0, s, desc);
  hipLaunchKernelGGL(blk_diag_kernel<true>, dim3(w.nb / 2, 1), dim3(256), 0, s, desc, w, full ? 0 : 1, full ? 1 : 0, 0, st);
  NDMPS_LAUNCH_CHECK();
  NDMPS_CHECK_HIP(hipMemcpyAsync(h_out, st, (size_t)(w.nb / 2) * 64, hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
