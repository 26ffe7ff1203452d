// Block two-sided Jacobi eigen-solver (fp64), the production path of ndmps_syevj_f64 and of the
// batched sweep.
//
// The matrix (padded to a multiple of 32) is cut into 16-wide index blocks.  A sweep visits every
// pair of blocks once (round-robin over blocks: nb-1 outer steps, nb/2 disjoint block pairs per
// step).  One launch per outer step, with two kinds of workgroups:
//
//   apply (step t) : one workgroup per 32x32 tile of the block-pair tiling of step t:
//           G_out[tile] = Q_A^T G_in[tile] Q_B (off-diagonal tiles), = the rotated diagonal tile
//           prepared earlier (diagonal tiles), V[strip] <- V[strip] Q_B; f64 MFMA (16x16x4) from
//           LDS.  G ping-pongs between two buffers, V is updated in place.
//   diag (step t+1): one workgroup per block pair (I, J) of the NEXT step builds the 32x32 diagonal
//           sub-matrix of I u J as it will be after step t -- its two 16x16 diagonal blocks come
//           from the prepared diagonal tiles of step t, its cross block from Q_A^T G_in[tile] Q_B of
//           the one tile that holds it -- and runs the Jacobi rotations of that visit on it in LDS
//           (16 disjoint rotations per inner step, 256 threads each owning one 2x2 block, one
//           barrier per inner step, S double-buffered), accumulating their product Q.  The first
//           step of a sweep rotates ALL index pairs inside I u J (31 inner steps), later steps only
//           the 16x16 cross pairs (16 inner steps): every index pair once per sweep.
//
// The two roles touch disjoint outputs, so the serial chain of a sweep is nb-1 launches whose
// length is max(apply, diag) instead of 2 (nb-1) launches of apply + diag (n-1 = 511 launches for
// the scalar-parallel method in eig.hip, kept as ndmps_syevj_simple_f64 for cross-checks).
//
// Batched: every kernel takes a batch of matrices (blockIdx.y); matrices of different size are
// padded to the common np (padded indices never rotate) and converged ones are skipped, so B
// eigenproblems cost the sequential depth of one.  Before the first sweep each matrix is permuted
// so that its diagonal is descending.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include <mutex>

#include "common.h"

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// Block size 16: 256 threads and four 32x32 LDS tiles (33 KB, 4 workgroups per CU).  A 32-wide variant (1024
// threads, 64x64 tiles) halves the outer steps per sweep but measured slower on MI355X (round 1: single
// volume 36 vs 26 ms) -- the diag role's inner step moves 4x the LDS bytes per barrier and one workgroup per
// CU starves the apply role -- and was removed.
constexpr int kMaxSweepsBlock = 40;
constexpr int BS = 16;
inline int block_size_for(int64_t) { return BS; }
inline int tiles_per_wg(int) { return 8; }  // column pairs streamed by one apply workgroup (V strips)

// one entry per matrix of the batch (device array); blockIdx.y selects it in every kernel
struct BatchDesc {
  const double* G_in;  // n x n input (ld n)
  double* V_out;       // n x n eigenvectors (columns), sorted
  double* w_out;       // n eigenvalues, descending
  int n;
  int done;            // converged: later launches skip this matrix
  int rotated[2];      // rotations above tol_conv, per sweep parity
  int final_buf;       // which G buffer holds the converged matrix
  int steps_applied;   // outer steps applied when it converged (history mode: rotations to replay)
  double tol_conv;
  double tol_rot;
  double rel_tol;      // convergence threshold relative to max |diag| (input)
};

struct Work {          // common padded working set; per-matrix strides np*np and (nb/2)*PS*PS
  double* G[2];        // ping-pong
  double* V;
  double* Q[2];        // rotations of the step being applied / being prepared
  double* D[2];        // rotated diagonal tiles, same parity scheme
  int* pos;            // [B][np] position of every index after the initial diagonal sort
  int* rank;           // [B][np] rank of the eigenvalue held by working column c
  int* invrank;        // [B][np] working column of the j-th largest eigenvalue
  double* sign;        // [B][np]
  double* hist;        // history mode: every step's rotation blocks, [B][max_steps][nb/2][PS*PS]
  int64_t hist_stride; // doubles per matrix in `hist`
  int hist_mode;       // 1: V is not updated per step; eigenvectors are replayed from `hist` afterwards
  int np;
  int nb;
  int bs;              // block size the matrices are cut into (16 or 32)
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
  // branch-free: computed for every pair (NaNs from a 0/0 pair are discarded by the selects), so
  // the compiler can overlap the chain with the independent LDS traffic around it
  const double a = aqq - app, b = 2.0 * apq;
  int e;
  (void)frexp(fmax(fabs(a), fabs(b)), &e);
  const double as = ldexp(fabs(a), -e), bs = ldexp(fabs(b), -e);  // max of the two in [0.5, 1)
  const double h2 = fma(as, as, bs * bs);
  const double h = h2 * fast_rsqrt<1>(h2);
  const double mag = bs * fast_rcp<1>(as + h);
  const double tt = ((a < 0.0) != (b < 0.0)) ? -mag : mag;
  const double cc = fast_rsqrt<2>(fma(tt, tt, 1.0));
  const bool act = fabs(apq) > tol_rot;
  t = act ? tt : 0.0;
  c = act ? cc : 1.0;
  s = act ? tt * cc : 0.0;
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
    d.tol_conv = d.rel_tol * red[0];
    d.tol_rot = 1e-19 * red[0];
    d.rotated[0] = 0;
    d.rotated[1] = 0;
    d.done = 0;
    d.final_buf = 0;
  }
}

__device__ __forceinline__ int block_sum_int(int v, int* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// offset of G(r, c) in a work matrix: every 16x16 block is contiguous (2 KB), so an MFMA operand fetch of the
// register apply path is one 512-byte run and a step streams whole blocks instead of 128-byte pieces of
// 4 KB-strided rows.
__device__ __forceinline__ int64_t g_off(int, int np, int64_t r, int64_t c) {
  return (((r >> 4) * (np >> 4) + (c >> 4)) << 8) + ((r & 15) << 4) + (c & 15);
}

// position of every index after sorting the diagonal descending (ties by index);
// one workgroup per index (grid.x = n_max), the count is a block reduction
__global__ void __launch_bounds__(256) blk_order_kernel(const BatchDesc* __restrict__ desc, Work w) {
  __shared__ int red[4];
  const BatchDesc& d = desc[blockIdx.y];
  const double* G = d.G_in;
  const int n = d.n;
  const int i = blockIdx.x;
  if (i >= n) return;
  const double di = G[(int64_t)i * n + i];
  int rk = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double dj = G[(int64_t)j * n + j];
    rk += (dj > di) || (dj == di && j < i);
  }
  rk = block_sum_int(rk, red);
  if (threadIdx.x == 0) w.pos[(int64_t)blockIdx.y * w.np + i] = rk;
}

// G[0] = P^T sym(G) P (padded with zeros), V = P on the real indices, identity on the padding
__global__ void __launch_bounds__(256) blk_init_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np;
  double* Gp = w.G[0] + (int64_t)blockIdx.y * np * np;
  double* Vp = w.V + (int64_t)blockIdx.y * np * np;
  const int64_t total = (int64_t)np * np;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / np), c = (int)(e % np);
    Gp[e] = 0.0;
    if (!w.hist_mode) Vp[e] = (r == c && r >= n) ? 1.0 : 0.0;  // history mode keeps no V
  }
}
__global__ void __launch_bounds__(256) blk_scatter_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const double* G = d.G_in;
  const int n = d.n, np = w.np;
  double* Gp = w.G[0] + (int64_t)blockIdx.y * np * np;
  double* Vp = w.V + (int64_t)blockIdx.y * np * np;
  const int* pos = w.pos + (int64_t)blockIdx.y * np;
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    Gp[g_off(w.bs, np, pos[r], pos[c])] = 0.5 * (G[(int64_t)r * n + c] + G[(int64_t)c * n + r]);
    if (r == c && !w.hist_mode) Vp[(int64_t)r * np + pos[r]] = 1.0;
  }
}

// after the launches of sweep `sweep`: a matrix whose sweep rotated nothing above tol_conv is done;
// its converged G sits in buffer `buf_now`
__global__ void blk_check_kernel(BatchDesc* __restrict__ desc, int batch, int sweep, int buf_now,
                                 int steps_now, int* __restrict__ remaining) {
  __shared__ int left;
  if (threadIdx.x == 0) left = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < batch; b += blockDim.x) {
    if (!desc[b].done) {
      if (desc[b].rotated[sweep & 1] == 0) {
        desc[b].done = 1;
        desc[b].final_buf = buf_now;
        desc[b].steps_applied = steps_now;
      } else {
        atomicAdd(&left, 1);
      }
    }
    desc[b].rotated[sweep & 1] = 0;
  }
  __syncthreads();
  if (threadIdx.x == 0) *remaining = left;
}

// ---------------------------------------------------------------------------- step kernel
// C(PS x PS) = op(A) * B in LDS, f64 MFMA; (PS/16)^2 waves, one 16x16 output tile each.
// TRANS_A: A given as (k, i) (i.e. C = A^T B).
template <bool TRANS_A>
__device__ __forceinline__ void lds_gemm(const double (*A)[2 * BS + 1], const double (*B)[2 * BS + 1],
                                         double (*Cout)[2 * BS + 1], int wave, int lane) {
  constexpr int PS = 2 * BS, TPR = PS / 16;
  const int i0 = (wave / TPR) * 16, j0 = (wave % TPR) * 16;
  const int li = lane & 15, lk = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
  for (int k0 = 0; k0 < PS; k0 += 4) {
    const double a = TRANS_A ? A[k0 + lk][i0 + li] : A[i0 + li][k0 + lk];
    const double b = B[k0 + lk][j0 + li];
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Cout[i0 + lk + 4 * r][j0 + li] = acc[r];
}

// sorted blocks (lo < hi) of pair k at outer step t
__device__ __forceinline__ void pair_blocks(int k, int t, int nb, int& lo, int& hi) {
  int a, b;
  circle_pair(k, t, nb, a, b);
  lo = min(a, b);
  hi = max(a, b);
}

// which pair of step t holds block x, and is x its low (0) or high (1) block
__device__ __forceinline__ void locate_block(int x, int t, int nb, int& k, int& pos) {
  const int m1 = nb - 1;
  int partner;
  if (x == nb - 1) {
    k = 0;
    partner = t;
  } else if (x == t) {
    k = 0;
    partner = nb - 1;
  } else {
    const int kk = (x - t + m1) % m1;  // x = (t + kk) mod m1
    if (kk <= (m1 - 1) / 2) {
      k = kk;
      partner = (t - kk + m1) % m1;
    } else {
      k = m1 - kk;  // x = (t - k) mod m1
      partner = (t + k) % m1;
    }
  }
  pos = x < partner ? 0 : 1;
}

__device__ __forceinline__ int64_t pair_index(int e, int lo, int hi) {  // e in [0, 2 BS)
  return e < BS ? (int64_t)lo * BS + e : (int64_t)hi * BS + e - BS;
}

// grid.x = n_diag (diag role, dispatched first) + n_apply (apply role); grid.y = batch.
//   t        : outer step applied by the apply role (ignored when n_apply == 0)
//   t_next   : outer step prepared by the diag role; full_next: rotate all pairs (first step of a
//              sweep); sweep_next: its sweep (selects the rotation counter)
//   first    : diag role reads the initial matrix directly (nothing to apply yet)
//   solve    : single block pair = whole matrix, iterate to convergence in LDS
//   in, out  : G ping-pong indices;  q_cur: parity of the Q / D buffers being applied
__global__ void __launch_bounds__(BS * BS, 2)
blk_step_kernel(BatchDesc* __restrict__ desc, Work w, int n_diag, int t, int t_next, int full_next,
                int sweep_next, int first, int solve, int in, int q_cur, int kTilesPerWg, int gstep) {
  constexpr int PS = 2 * BS, LD = PS + 1, NT = BS * BS;  // pair size, padded LDS row, threads
  BatchDesc& d = desc[blockIdx.y];
  if (d.done) return;
  // four PS x PS tiles of (dynamic) LDS, shared by both roles (33 KB at BS = 16, 133 KB at BS = 32):
  //   apply: T, QA, QB, X.   diag: the same four while the sub-matrix is built, then
  //   Q = QA's slot, S ping-pong = X's and QB's slots (all dead by then), T unused.
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  __shared__ int cnt;
  typedef double (*tile_t)[LD];
  tile_t T = reinterpret_cast<tile_t>(lds_raw);
  tile_t QA = reinterpret_cast<tile_t>(lds_raw + PS * LD);
  tile_t QB = reinterpret_cast<tile_t>(lds_raw + 2 * PS * LD);
  tile_t X = reinterpret_cast<tile_t>(lds_raw + 3 * PS * LD);
  tile_t Q = QA;
  tile_t S0 = X;
  tile_t S1 = QB;

  const int np = w.np, nb = w.nb, half = nb >> 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t mat = (int64_t)blockIdx.y * np * np;
  const int64_t qoff = (int64_t)blockIdx.y * half * PS * PS;
  const double* Gin = w.G[in] + mat;
  // rotations of the step being applied: ping-pong buffer, or slot `gstep` of the history
  const int64_t slot = (int64_t)half * PS * PS;
  const double* Qcur = w.hist_mode ? w.hist + (int64_t)blockIdx.y * w.hist_stride + (int64_t)max(gstep, 0) * slot
                                   : w.Q[q_cur] + qoff;
  const double* Dcur = w.D[q_cur] + qoff;

  if ((int)blockIdx.x >= n_diag) {
    // ================================================================== apply role (step t)
    // One workgroup = one row pair (or one 32-row strip of V) x kTilesPerWg column pairs: Q_A is
    // loaded once, tiles stream through LDS; 16-byte global accesses (thread = row, 2-double segment).
    double* V = w.V + mat;
    const int chunks = (half + kTilesPerWg - 1) / kTilesPerWg;
    int bid = blockIdx.x - n_diag;
    {
      // G tiles, register-only: one WAVE = one upper tile (pa < pb), no LDS and no barrier.  With
      // R = Q_A^T T Q_B, the wave computes, for each 16-column half h of pair A,
      // R^T[:, h] = Q_B^T (T^T Q_A[:, h]): the accumulator of the first product (row 4r + lk, column li)
      // IS the B operand of the second (k = 4s + lk), so nothing is transposed or staged; all operands
      // are read from global / L2 straight into MFMA layout.  Wave items per matrix: `half` copies of
      // the prepared diagonal tiles + one per upper tile.
      const int n_items = half + half * (half - 1) / 2;
      const int g_wgs = (n_items + 3) >> 2;
      if (bid < g_wgs) {
        const int wid = __builtin_amdgcn_readfirstlane(bid * 4 + wave);
        if (wid >= n_items) return;
        double* Gout = w.G[in ^ 1] + mat;
        const int li = lane & 15, lk = lane >> 4;
        if (wid < half) {  // diagonal tile: prepared (already rotated) by the diag role
          int lo, hi;
          pair_blocks(wid, t, nb, lo, hi);
          const double* dsrc = Dcur + (int64_t)wid * PS * PS;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const int e = 2 * lane + 128 * q;
            *reinterpret_cast<double2*>(Gout + g_off(BS, np, pair_index(e / PS, lo, hi), pair_index(e % PS, lo, hi))) =
                *reinterpret_cast<const double2*>(dsrc + e);
          }
          return;
        }
        int u = wid - half, pa = 0;
        while (u >= half - 1 - pa) {
          u -= half - 1 - pa;
          ++pa;
        }
        const int pb = pa + 1 + u;
        int lo_a, hi_a, lo_b, hi_b;
        pair_blocks(pa, t, nb, lo_a, hi_a);
        pair_blocks(pb, t, nb, lo_b, hi_b);
        const double* qa = Qcur + (int64_t)pa * PS * PS + lk * PS + li;
        const double* qb = Qcur + (int64_t)pb * PS * PS + lk * PS + li;
        // G is symmetric and only its block-upper half is kept (16x16 block (x, y) with x < y; the
        // diagonal blocks travel in the prepared diagonal tiles): a block with x > y is read, and
        // written, through its mirror image -- 32-byte pieces instead of 512-byte runs, same bytes.
        // That halves the write traffic of a step.
        auto block_ptr = [&](int x, int y, int64_t& sx, int64_t& sy) -> int64_t {  // &G[x rows][y cols]
          if (x < y) {
            sx = BS;
            sy = 1;
            return ((int64_t)x * nb + y) * (BS * BS);
          }
          sx = 1;
          sy = BS;
          return ((int64_t)y * nb + x) * (BS * BS);
        };
        double ta[8], tb[8], qb0[8], qb1[8];
#pragma unroll
        for (int kx = 0; kx < 2; ++kx) {  // k = 16 kx + 4 s + lk: rows of pair A
          const int x = kx ? hi_a : lo_a;
          int64_t sx0, sy0, sx1, sy1;
          const double* p0 = Gin + block_ptr(x, lo_b, sx0, sy0) + lk * sx0 + li * sy0;
          const double* p1 = Gin + block_ptr(x, hi_b, sx1, sy1) + lk * sx1 + li * sy1;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            ta[4 * kx + s] = p0[4 * s * sx0];
            tb[4 * kx + s] = p1[4 * s * sx1];
          }
        }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          qb0[s] = qb[s * 4 * PS];
          qb1[s] = qb[s * 4 * PS + 16];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          double qav[8];
#pragma unroll
          for (int s = 0; s < 8; ++s) qav[s] = qa[s * 4 * PS + 16 * h];
          f64x4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 8; ++s) {  // Y = T^T Q_A[:, h]  (rows: pair-B index)
            y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[s], qav[s], y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tb[s], qav[s], y1, 0, 0, 0);
          }
          f64x4 r0 = {0.0, 0.0, 0.0, 0.0}, r1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < 8; ++s) {  // R^T[:, h] = Q_B^T Y
            const double yb = s < 4 ? y0[s & 3] : y1[s & 3];
            r0 = __builtin_amdgcn_mfma_f64_16x16x4f64(qb0[s], yb, r0, 0, 0, 0);
            r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(qb1[s], yb, r1, 0, 0, 0);
          }
          // r_ib[r] = R[row li of block x][column 4 r + lk of block y_ib]
          const int x = h ? hi_a : lo_a;
          int64_t sx0, sy0, sx1, sy1;
          double* o0 = Gout + block_ptr(x, lo_b, sx0, sy0) + li * sx0 + lk * sy0;
          double* o1 = Gout + block_ptr(x, hi_b, sx1, sy1) + li * sx1 + lk * sy1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o0[4 * r * sy0] = r0[r];
            o1[4 * r * sy1] = r1[r];
          }
        }
        return;
      }
      bid -= g_wgs;
    }
    // V strips (in-loop eigenvector update of the non-history mode): one workgroup = one 32-row strip of V x
    // kTilesPerWg column pairs, tiles stream through LDS; 16-byte global accesses (thread = row, 2-double segment)
    const int strip = bid / chunks, chunk = bid % chunks;
    const int row = tid / (BS / 2), c0 = (tid % (BS / 2)) * 2;  // this thread's row and column offset inside a block
    const int64_t grow = (int64_t)strip * PS + row;
    // software pipeline: the operands of tile j+1 are fetched into registers while tile j is in LDS
    double2 t0, t1, q0, q1;
    auto fetch = [&](int pb_) {
      int lo_b_, hi_b_;
      pair_blocks(pb_, t, nb, lo_b_, hi_b_);
      t0 = *reinterpret_cast<const double2*>(V + grow * np + (int64_t)lo_b_ * BS + c0);
      t1 = *reinterpret_cast<const double2*>(V + grow * np + (int64_t)hi_b_ * BS + c0);
      q0 = *reinterpret_cast<const double2*>(Qcur + (int64_t)pb_ * PS * PS + row * PS + c0);
      q1 = *reinterpret_cast<const double2*>(Qcur + (int64_t)pb_ * PS * PS + row * PS + BS + c0);
    };
    const int pb_first = chunk * kTilesPerWg;
    if (pb_first < half) fetch(pb_first);
    for (int j = 0; j < kTilesPerWg; ++j) {
      const int pb = pb_first + j;
      if (pb >= half) break;
      int lo_b, hi_b;
      pair_blocks(pb, t, nb, lo_b, hi_b);
      double* o0 = V + grow * np + (int64_t)lo_b * BS + c0;
      double* o1 = V + grow * np + (int64_t)hi_b * BS + c0;
      __syncthreads();  // the previous tile's readers of T / QB / X are done
      T[row][c0] = t0.x;
      T[row][c0 + 1] = t0.y;
      T[row][BS + c0] = t1.x;
      T[row][BS + c0 + 1] = t1.y;
      QB[row][c0] = q0.x;
      QB[row][c0 + 1] = q0.y;
      QB[row][BS + c0] = q1.x;
      QB[row][BS + c0 + 1] = q1.y;
      if (j + 1 < kTilesPerWg && pb + 1 < half) fetch(pb + 1);  // in flight during the GEMM
      __syncthreads();
      lds_gemm<false>(T, QB, X, wave, lane);  // X = T Q_B
      __syncthreads();
      *reinterpret_cast<double2*>(o0) = make_double2(X[row][c0], X[row][c0 + 1]);
      *reinterpret_cast<double2*>(o1) = make_double2(X[row][BS + c0], X[row][BS + c0 + 1]);
    }
    return;
  }

  // ==================================================================== diag role (step t_next)
  // the diag workgroups are the serial chain of the launch: let their waves win issue arbitration
  // against co-resident apply waves
  __builtin_amdgcn_s_setprio(3);
  double* Qnext = w.hist_mode ? w.hist + (int64_t)blockIdx.y * w.hist_stride + (int64_t)(gstep + 1) * slot
                              : w.Q[q_cur ^ 1] + qoff;
  double* Dnext = w.D[q_cur ^ 1] + qoff;
  int lo, hi;
  pair_blocks(blockIdx.x, t_next, nb, lo, hi);
  const double tol_rot = d.tol_rot, tol_conv = d.tol_conv;
  if (tid == 0) cnt = 0;
  if (first) {
    for (int e = tid; e < PS * PS; e += NT) {
      const int a = e / PS, b = e % PS;
      const int64_t ga = pair_index(a, lo, hi), gb = pair_index(b, lo, hi);
      S0[a][b] = 0.5 * (Gin[g_off(BS, np, ga, gb)] + Gin[g_off(BS, np, gb, ga)]);
      Q[a][b] = (a == b) ? 1.0 : 0.0;
    }
    __syncthreads();
  } else {
    // the 32x32 sub-matrix of (lo, hi) as step t leaves it
    int ka, pa, kb, pb;
    locate_block(lo, t, nb, ka, pa);
    locate_block(hi, t, nb, kb, pb);  // ka != kb: a block pair meets once per sweep
    int a_lo, a_hi, b_lo, b_hi;
    pair_blocks(ka, t, nb, a_lo, a_hi);
    pair_blocks(kb, t, nb, b_lo, b_hi);
    const int r16 = tid / BS, c16 = tid % BS;  // one element of every BS x BS block per thread
    // diagonal 16x16 blocks from the prepared diagonal tiles of step t (kept in registers
    // until the LDS slots they go to are free)
    const double d_lo = Dcur[(int64_t)ka * PS * PS + (pa * BS + r16) * PS + pa * BS + c16];
    const double d_hi = Dcur[(int64_t)kb * PS * PS + (pb * BS + r16) * PS + pb * BS + c16];
    for (int e = tid; e < PS * PS; e += NT) {
      const int a = e / PS, b = e % PS;
      int64_t ga = pair_index(a, a_lo, a_hi), gb = pair_index(b, b_lo, b_hi);
      if (ga / BS > gb / BS) {  // block-upper storage: read the mirror image
        const int64_t tmp = ga;
        ga = gb;
        gb = tmp;
      }
      T[a][b] = Gin[g_off(BS, np, ga, gb)];
      QA[a][b] = Qcur[(int64_t)ka * PS * PS + e];
      QB[a][b] = Qcur[(int64_t)kb * PS * PS + e];
    }
    __syncthreads();
    lds_gemm<false>(T, QB, X, wave, lane);  // X = T Q_B
    __syncthreads();
    lds_gemm<true>(QA, X, T, wave, lane);  // T = Q_A^T X  (tile (ka, kb) after step t)
    __syncthreads();
    {
      const double v = T[pa * BS + r16][pb * BS + c16];  // cross block (lo, hi)
      S0[r16][BS + c16] = v;
      S0[BS + c16][r16] = v;
      S0[r16][c16] = d_lo;
      S0[BS + r16][BS + c16] = d_hi;
    }
    for (int e = tid; e < PS * PS; e += NT) Q[e / PS][e % PS] = (e / PS == e % PS) ? 1.0 : 0.0;
    __syncthreads();
  }

  const int K = tid / BS, M = tid % BS;
  const int n_inner = full_next ? PS - 1 : BS;
  const int max_rounds = solve ? kMaxSweepsBlock : 1;
  const int r0 = 2 * K, r1 = 2 * K + 1;  // this thread's (static) rows of Q
  bool converged = false;
  int cur = 0;
  // the Q update of an inner step touches only this thread's own entries, so it is deferred by one
  // step and overlaps the next step's rotation chain
  bool pending = false;
  double pc = 1.0, ps = 0.0;
  int pcol_r = 0, pcol_s = 0;
  auto flush_q = [&]() {
    const double q0r = Q[r0][pcol_r], q0s = Q[r0][pcol_s], q1r = Q[r1][pcol_r], q1s = Q[r1][pcol_s];
    Q[r0][pcol_r] = pc * q0r - ps * q0s;
    Q[r0][pcol_s] = ps * q0r + pc * q0s;
    Q[r1][pcol_r] = pc * q1r - ps * q1s;
    Q[r1][pcol_s] = ps * q1r + pc * q1s;
  };
  for (int round = 0; round < max_rounds; ++round) {
    int before = 0;
    if (solve) {
      before = cnt;
      __syncthreads();  // nobody may bump cnt for this round before everyone has read it
    }
    for (int st = 0; st < n_inner; ++st) {
      tile_t S = cur ? S1 : S0;
      tile_t Sn = cur ? S0 : S1;
      int p, q, r, s_;
      if (full_next) {
        circle_pair(K, st, PS, p, q);
        circle_pair(M, st, PS, r, s_);
      } else {
        p = K;
        q = BS + ((K + st) & (BS - 1));
        r = M;
        s_ = BS + ((M + st) & (BS - 1));
      }
      // own (column) rotation: pair M; the row rotation of pair K lives in lane K of this wave
      const double arr = S[r][r], ass = S[s_][s_], ars = S[r][s_];
      const double gpr = S[p][r], gps = S[p][s_], gqr = S[q][r], gqs = S[q][s_];
      if (pending) flush_q();
      double c2, s2, t2;
      rotation64(arr, ass, ars, tol_rot, c2, s2, t2);
      const double c1 = __shfl(c2, K, 64), s1 = __shfl(s2, K, 64);
      if (tid < 64) {  // wave 0 counts the pairs still above the convergence threshold
        const unsigned long long m = __ballot(lane < BS && fabs(ars) > tol_conv);
        if (lane == 0) cnt += __popcll(m);
      }
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
      pending = true;
      pc = c2;
      ps = s2;
      pcol_r = r;
      pcol_s = s_;
      __syncthreads();
      cur ^= 1;
    }
    if (solve && cnt == before) {  // cnt is stable here: last write was before the barrier
      converged = true;
      break;
    }
  }
  if (pending) flush_q();
  __syncthreads();

  for (int e = tid; e < PS * PS; e += NT) {
    Dnext[(int64_t)blockIdx.x * PS * PS + e] = (cur ? S1 : S0)[e / PS][e % PS];
    Qnext[(int64_t)blockIdx.x * PS * PS + e] = Q[e / PS][e % PS];
  }
  if (tid == 0) {
    if (!solve && cnt > 0) atomicAdd(&d.rotated[sweep_next & 1], cnt);
    if (solve && !converged) atomicAdd(&d.rotated[sweep_next & 1], 1);
  }
}

// ---------------------------------------------------------------------------- finish
// rank of every eigenvalue (descending, ties by index) and, when V was accumulated in the loop, the
// sign of its eigenvector (largest component positive; ties -> lowest row): one workgroup per
// index, block reductions
__global__ void __launch_bounds__(256) blk_rank_kernel(const BatchDesc* __restrict__ desc, Work w) {
  __shared__ int red[4];
  __shared__ double best_v[4];
  __shared__ int best_r[4];
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np;
  const double* G = w.G[d.final_buf] + (int64_t)blockIdx.y * np * np;
  const int i = blockIdx.x;
  if (i >= n) return;
  const double wi = G[g_off(w.bs, np, i, i)];
  int rk = 0;
  for (int j = threadIdx.x; j < n; j += 256) {
    const double wj = G[g_off(w.bs, np, j, j)];
    rk += (wj > wi) || (wj == wi && j < i);
  }
  rk = block_sum_int(rk, red);
  if (w.hist_mode) {
    if (threadIdx.x == 0) {
      w.rank[(int64_t)blockIdx.y * np + i] = rk;
      w.invrank[(int64_t)blockIdx.y * np + rk] = i;
      d.w_out[rk] = wi;
    }
    return;
  }
  const double* V = w.V + (int64_t)blockIdx.y * np * np;
  // arg max |V[r][i]| with the lowest r on ties (matches a sequential scan with '>')
  double bv = -1.0;
  int br = 0x7fffffff;
  for (int r = threadIdx.x; r < n; r += 256) {
    const double a = fabs(V[(int64_t)r * np + i]);
    if (a > bv) {
      bv = a;
      br = r;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(bv, off, 64);
    const int orr = __shfl_down(br, off, 64);
    if (ov > bv || (ov == bv && orr < br)) {
      bv = ov;
      br = orr;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    best_v[threadIdx.x >> 6] = bv;
    best_r[threadIdx.x >> 6] = br;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < 4; ++k)
      if (best_v[k] > bv || (best_v[k] == bv && best_r[k] < br)) {
        bv = best_v[k];
        br = best_r[k];
      }
    const double v = br < n ? V[(int64_t)br * np + i] : 1.0;
    w.rank[(int64_t)blockIdx.y * np + i] = rk;
    w.sign[(int64_t)blockIdx.y * np + i] = v < 0.0 ? -1.0 : 1.0;
    d.w_out[rk] = wi;
  }
}

// History mode: eigenvectors by replaying the rotations.  V = P Q_0 Q_1 ... Q_{T-1} (P: initial sort,
// Q_t: block-diagonal over step t's pairing), so column c of V is P Q_0 ... Q_{T-1} e_c.  One
// workgroup keeps 16 columns (np x 16 doubles) in LDS and applies Q_{T-1} ... Q_0 to them: per step
// each wave multiplies the 32-row slices of its block pairs by the stored 32x32 blocks (f64 MFMA).
// Nothing but the rotation blocks is read from memory, and only the k wanted columns are formed --
// the in-loop update streams all np x np of V through the cache hierarchy every step.
// grid (ceil(k_max / 16), batch); NW waves per workgroup (8 while the LDS allows: two waves per SIMD hide
// each other's LDS / history latency); dynamic LDS: np*17 + NW*32*33 doubles.
template <int NW>
__global__ void __launch_bounds__(64 * NW, 2)
blk_backacc_kernel(const BatchDesc* __restrict__ desc, Work w, const int* __restrict__ kcols) {
  constexpr int NT = 64 * NW;
  constexpr int BS = 16, PS = 32, LDJ = PS + 1, LDY = 17;
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np, nb = w.nb, half = nb >> 1;
  const int k = kcols[blockIdx.y];
  const int col0 = blockIdx.x * 16;
  if (col0 >= k) return;
  extern __shared__ __attribute__((aligned(16))) double lds_raw[];
  double* Y = lds_raw;                                  // [np][LDY]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double (*J)[LDJ] = reinterpret_cast<double (*)[LDJ]>(lds_raw + (int64_t)np * LDY + wave * PS * LDJ);
  const int* invrank = w.invrank + (int64_t)blockIdx.y * np;
  for (int e = tid; e < np * LDY; e += NT) Y[e] = 0.0;
  __syncthreads();
  if (tid < 16 && col0 + tid < k) Y[invrank[col0 + tid] * LDY + tid] = 1.0;
  __syncthreads();

  const double* hist = w.hist + (int64_t)blockIdx.y * w.hist_stride;
  const int64_t slot = (int64_t)half * PS * PS;
  const int li = lane & 15, lk = lane >> 4;
  const int steps = d.steps_applied, per_sweep = nb - 1;
  const int per_wave = (half + NW - 1) / NW;  // block pairs of a step handled by this wave: p = wave + NW i
  // the rotation block of the NEXT (step, pair) item is fetched into registers while the current one
  // is multiplied: the L2 / Infinity-Cache latency of the history reads leaves the critical path
  double2 nx[8];
  auto fetch = [&](int st, int i) {
    const int p = wave + NW * i;
    if (st < 0 || p >= half) return;
    const double* src = hist + (int64_t)st * slot + (int64_t)p * PS * PS;
#pragma unroll
    for (int q = 0; q < 8; ++q) nx[q] = *reinterpret_cast<const double2*>(src + 2 * lane + 128 * q);
  };
  fetch(steps - 1, 0);
  for (int st = steps - 1; st >= 0; --st) {
    const int t = st % per_sweep;
    for (int i = 0; i < per_wave; ++i) {
      const int p = wave + NW * i;
      if (p < half) {
        int lo, hi;
        pair_blocks(p, t, nb, lo, hi);
#pragma unroll
        for (int q = 0; q < 8; ++q) {  // 1024 doubles, 2 per lane and pass
          const int e = 2 * lane + 128 * q;
          J[e / PS][e % PS] = nx[q].x;
          J[e / PS][e % PS + 1] = nx[q].y;
        }
      }
      if (i + 1 < per_wave) fetch(st, i + 1);
      else fetch(st - 1, 0);
      if (p < half) {
        int lo, hi;
        pair_blocks(p, t, nb, lo, hi);
        f64x4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k0 = 0; k0 < PS; k0 += 4) {
          const int kr = k0 + lk;  // row of the pair slice this lane feeds as B[k][j]
          const double yb = Y[(kr < BS ? lo * BS + kr : hi * BS + kr - BS) * LDY + li];
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(J[li][kr], yb, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(J[16 + li][kr], yb, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // rows 0..15 of the slice live in block lo, 16..31 in block hi
          Y[(lo * BS + lk + 4 * r) * LDY + li] = acc0[r];
          Y[(hi * BS + lk + 4 * r) * LDY + li] = acc1[r];
        }
      }
    }
    __syncthreads();  // the next step regroups the rows
  }

  // sign convention (largest component positive, lowest row on ties) and the initial permutation
  __shared__ double sgn[16];
  if (tid < 16) {
    double best = -1.0, val = 1.0;
    const int* pos = w.pos + (int64_t)blockIdx.y * np;
    for (int r = 0; r < n; ++r) {  // original row order, as the in-loop path scans V
      const double v = Y[pos[r] * LDY + tid];
      if (fabs(v) > best) {
        best = fabs(v);
        val = v;
      }
    }
    sgn[tid] = val < 0.0 ? -1.0 : 1.0;
  }
  __syncthreads();
  const int* pos = w.pos + (int64_t)blockIdx.y * np;
  double* Vout = d.V_out;
  for (int e = tid; e < n * 16; e += NT) {
    const int r = e >> 4, j = e & 15;
    if (col0 + j < k) Vout[(int64_t)r * n + col0 + j] = sgn[j] * Y[pos[r] * LDY + j];
  }
}

__global__ void __launch_bounds__(256) blk_gather_kernel(const BatchDesc* __restrict__ desc, Work w) {
  const BatchDesc& d = desc[blockIdx.y];
  const int n = d.n, np = w.np;
  const double* V = w.V + (int64_t)blockIdx.y * np * np;
  const int* rank = w.rank + (int64_t)blockIdx.y * np;
  const double* sign = w.sign + (int64_t)blockIdx.y * np;
  double* Vout = d.V_out;
  const int64_t total = (int64_t)n * n;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int r = (int)(e / n), c = (int)(e % n);
    Vout[(int64_t)r * n + rank[c]] = sign[c] * V[(int64_t)r * np + c];
  }
}

// History mode is used for batches (the throughput path) of matrices that fit the LDS-resident
// replay (np <= 896 at 16 columns per workgroup) with the 16-wide blocking.
constexpr int kBackacc8MaxLds = 159 * 1024;  // eight-wave replay while Y and eight J buffers fit
constexpr int kHistMaxNp = 896;  // np * 17 + 4 * 32 * 33 doubles must fit the 160 KB of LDS
inline bool use_history(int64_t n_max, int batch) {
  const char* e = getenv("NDMPS_EIG_HISTORY");  // 0 / 1 force it off / on (experiments, tests)
  const int64_t np = std::max<int64_t>(ndmps::round_up(n_max, 32), 32);
  if (block_size_for(n_max) != 16 || np > kHistMaxNp) return false;
  if (e) return atoi(e) != 0;
  return batch >= 2;
}

struct BlockLayout {
  int64_t np, nb;
  int hist;
  int64_t hist_stride;  // doubles per matrix
  int64_t off_g[2], off_v, off_q[2], off_d[2], off_desc, off_pos, off_rank, off_invrank, off_sign, off_flag,
      off_k, off_hist, total;
};

BlockLayout block_layout(int64_t n_max, int64_t batch) {
  BlockLayout l;
  const int BS = block_size_for(n_max), PS = 2 * BS;
  l.np = std::max<int64_t>(ndmps::round_up(n_max, PS), PS);
  l.nb = l.np / BS;
  l.hist = use_history(n_max, (int)batch) ? 1 : 0;
  int64_t used = 0;
  auto take = [&](int64_t bytes) {
    const int64_t off = ndmps::round_up(used, 256);
    used = off + bytes;
    return off;
  };
  const int64_t qbytes = batch * (l.nb / 2) * PS * PS * 8;
  l.off_g[0] = take(batch * l.np * l.np * 8);
  l.off_g[1] = take(batch * l.np * l.np * 8);
  l.off_v = take(l.hist ? 256 : batch * l.np * l.np * 8);
  l.off_q[0] = take(l.hist ? 256 : qbytes);
  l.off_q[1] = take(l.hist ? 256 : qbytes);
  l.off_d[0] = take(qbytes);
  l.off_d[1] = take(qbytes);
  l.off_desc = take(batch * (int64_t)sizeof(BatchDesc));
  l.off_pos = take(batch * l.np * 4);
  l.off_rank = take(batch * l.np * 4);
  l.off_invrank = take(batch * l.np * 4);
  l.off_sign = take(batch * l.np * 8);
  l.off_flag = take(256);
  l.off_k = take(batch * 4);
  // one slot of rotation blocks per outer step of up to kMaxSweepsBlock sweeps, plus the prepared one
  l.hist_stride = l.hist ? ((int64_t)kMaxSweepsBlock * (l.nb - 1) + 1) * (l.nb / 2) * PS * PS : 0;
  l.off_hist = take(l.hist ? batch * l.hist_stride * 8 : 256);
  l.total = ndmps::round_up(used, 256);
  return l;
}

// Solver state between the two phases: values() iterates to convergence and delivers the sorted
// eigenvalues; vectors(k) delivers the first k eigenvectors of every matrix.
// Two pinned ints and two events for the pipelined convergence test of values().  They come from a process-wide
// pool (one set per concurrently running solve and device, reused afterwards): per-thread storage leaked a
// pinned allocation and two events for every worker thread a caller ever created.
struct HostFlags {
  int* flags = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int device = -1;
};
class HostFlagsPool {
 public:
  HostFlags* acquire() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    {
      std::lock_guard<std::mutex> lock(mu_);
      for (size_t i = 0; i < free_.size(); ++i)
        if (free_[i]->device == dev) {
          HostFlags* hf = free_[i];
          free_.erase(free_.begin() + i);
          return hf;
        }
    }
    int* p = nullptr;
    if (hipHostMalloc((void**)&p, 64, hipHostMallocPortable) != hipSuccess) return nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreateWithFlags(&e0, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess) {
      (void)hipHostFree(p);
      return nullptr;
    }
    HostFlags* hf = new HostFlags();
    hf->flags = p;
    hf->ev[0] = e0;
    hf->ev[1] = e1;
    hf->device = dev;
    return hf;
  }
  void release(HostFlags* hf) {
    if (!hf) return;
    std::lock_guard<std::mutex> lock(mu_);
    free_.push_back(hf);
  }

 private:
  std::mutex mu_;
  std::vector<HostFlags*> free_;
};
inline HostFlagsPool& host_flags_pool() {
  static HostFlagsPool* pool = new HostFlagsPool();  // never destroyed: no HIP calls at process exit
  return *pool;
}
struct HostFlagsLease {
  HostFlags* hf;
  HostFlagsLease() : hf(host_flags_pool().acquire()) {}
  ~HostFlagsLease() { host_flags_pool().release(hf); }
};

// kernels that need more than 64 KB of dynamic LDS are opted in once per DEVICE (function attributes are
// per device: a process that moves to a second GPU must opt in there too)
inline int blk_opt_in() {
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  if (dev < 0 || dev >= 64 || done[dev]) return NDMPS_OK;
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&blk_backacc_kernel<4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (kHistMaxNp * 17 + 4 * 32 * 33) * 8));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&blk_backacc_kernel<8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kBackacc8MaxLds));
  done[dev] = true;
  return NDMPS_OK;
}

struct BatchedJacobi {
  int batch = 0;
  int64_t n_max = 0;
  BlockLayout l;
  Work w;
  BatchDesc* desc = nullptr;
  int* flag = nullptr;
  int* d_k = nullptr;
  hipStream_t s = nullptr;
  int sweeps = 0;

  int init(int batch_, std::vector<BatchDesc>& host_desc, int64_t n_max_, void* d_ws, int64_t ws_bytes,
           hipStream_t stream) {
    batch = batch_;
    n_max = n_max_;
    s = stream;
    l = block_layout(n_max, batch);
    if (d_ws == nullptr || ws_bytes < l.total) {
      ndmps::set_error("syevj workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
      return NDMPS_EWORKSPACE;
    }
    char* base = (char*)d_ws;
    for (int i = 0; i < 2; ++i) {
      w.G[i] = (double*)(base + l.off_g[i]);
      w.Q[i] = (double*)(base + l.off_q[i]);
      w.D[i] = (double*)(base + l.off_d[i]);
    }
    w.V = (double*)(base + l.off_v);
    w.pos = (int*)(base + l.off_pos);
    w.rank = (int*)(base + l.off_rank);
    w.invrank = (int*)(base + l.off_invrank);
    w.sign = (double*)(base + l.off_sign);
    w.hist = (double*)(base + l.off_hist);
    w.hist_stride = l.hist_stride;
    w.hist_mode = l.hist;
    w.np = (int)l.np;
    w.nb = (int)l.nb;
    w.bs = block_size_for(n_max);
    desc = (BatchDesc*)(base + l.off_desc);
    flag = (int*)(base + l.off_flag);
    d_k = (int*)(base + l.off_k);
    NDMPS_CHECK_HIP(hipMemcpyAsync(desc, host_desc.data(), sizeof(BatchDesc) * batch, hipMemcpyHostToDevice, s));
    return NDMPS_OK;
  }

  int values() {
    const int np = w.np, nb = w.nb, half = nb / 2;
    const int PS = 2 * BS, kTilesPerWg = tiles_per_wg(BS);
    const size_t lds_bytes = (size_t)4 * PS * (PS + 1) * sizeof(double);
    NDMPS_TRY(blk_opt_in());
    const unsigned B = (unsigned)batch;
    // NDMPS_EIG_DEBUG_ROLE = 1 / 2: launch only the diag / apply role (timing experiments; results are wrong)
    static const int debug_role = [] {
      const int v = getenv("NDMPS_EIG_DEBUG_ROLE") ? atoi(getenv("NDMPS_EIG_DEBUG_ROLE")) : 0;
      if (v) fprintf(stderr, "libndmps_hip: NDMPS_EIG_DEBUG_ROLE=%d -- timing experiment, eigen results are WRONG\n", v);
      return v;
    }();
    auto step = [&](unsigned gx, int n_diag, int t, int t_next, int full_next, int sweep_next, int first, int solve,
                    int in, int q_cur, int gstep) {
      if (debug_role == 1 && n_diag > 0) gx = n_diag;
      if (debug_role == 2 && (int)gx > n_diag && n_diag > 0) { gx -= n_diag; n_diag = 0; }
      hipLaunchKernelGGL(blk_step_kernel, dim3(gx, B), dim3(256), lds_bytes, s, desc, w, n_diag, t, t_next,
                         full_next, sweep_next, first, solve, in, q_cur, kTilesPerWg, gstep);
    };
    const int fill_grid = (int)std::min<int64_t>(ndmps::ceil_div((int64_t)np * np, 256), 1024);
    hipLaunchKernelGGL(blk_scale_kernel, dim3(1, B), dim3(256), 0, s, desc);
    hipLaunchKernelGGL(blk_order_kernel, dim3((unsigned)n_max, B), dim3(256), 0, s, desc, w);
    hipLaunchKernelGGL(blk_init_kernel, dim3(fill_grid, B), dim3(256), 0, s, desc, w);
    hipLaunchKernelGGL(blk_scatter_kernel, dim3(fill_grid, B), dim3(256), 0, s, desc, w);
    NDMPS_LAUNCH_CHECK();

    int remaining = 1;
    sweeps = 0;
    const int chunks = (half + kTilesPerWg - 1) / kTilesPerWg;
    // history mode: the apply role has no V tiles
    const int n_apply = (half + half * (half - 1) / 2 + 3) / 4 + (w.hist_mode ? 0 : (np / PS) * chunks);
    const int steps = nb - 1;  // outer steps per sweep
    if (nb == 2) {
      // every matrix is one block pair: solved in LDS by the diag role, then applied once
      step(half, half, 0, 0, 1, 0, 1, 1, 0, 1, -1);
      step(n_apply, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
      hipLaunchKernelGGL(blk_check_kernel, dim3(1), dim3(64), 0, s, desc, batch, 0, 1, 1, flag);
      NDMPS_LAUNCH_CHECK();
      sweeps = 1;
      NDMPS_CHECK_HIP(hipMemcpyAsync(&remaining, flag, sizeof(int), hipMemcpyDeviceToHost, s));
      NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    } else {
      // prepare step 0 of sweep 0 from the initial matrix (diag role only; writes Q[0] / slot 0, D[0])
      step(half, half, 0, 0, 1, 0, 1, 0, 0, 1, -1);
      int g = 0;  // global step counter: G buffer in = g & 1, Q/D parity of the step applied = g & 1
      // The convergence flag of sweep k is read on the host only after sweep k+1 has been enqueued, so
      // the stream never drains while the host looks at it (the round trip cost ~30 us per sweep, ~75
      // sweeps per 256^3 volume).  If sweep k turns out to have converged everything, the launches of
      // sweep k+1 find every matrix marked done by sweep k's check kernel and return at once.
      HostFlagsLease lease;
      HostFlags* hf = lease.hf;
      NDMPS_REQUIRE(hf != nullptr, "pinned flag buffer / events could not be created");
      int enqueued = 0, verified = 0;
      while (true) {
        if (enqueued < kMaxSweepsBlock) {
          for (int t = 0; t < steps; ++t, ++g) {
            const int t_next = (t + 1) % steps;
            const int sweep_next = enqueued + (t == steps - 1 ? 1 : 0);
            step(half + n_apply, half, t, t_next, t_next == 0 ? 1 : 0, sweep_next, 0, 0, g & 1, g & 1, g);
          }
          const int slot = enqueued & 1;
          hipLaunchKernelGGL(blk_check_kernel, dim3(1), dim3(64), 0, s, desc, batch, enqueued, g & 1, g, flag + slot);
          NDMPS_LAUNCH_CHECK();
          NDMPS_CHECK_HIP(hipMemcpyAsync(hf->flags + slot, flag + slot, sizeof(int), hipMemcpyDeviceToHost, s));
          NDMPS_CHECK_HIP(hipEventRecord(hf->ev[slot], s));
          ++enqueued;
        }
        if (enqueued > verified + 1 || enqueued == kMaxSweepsBlock) {
          NDMPS_CHECK_HIP(hipEventSynchronize(hf->ev[verified & 1]));
          remaining = hf->flags[verified & 1];
          ++verified;
          if (remaining == 0 || verified == kMaxSweepsBlock) break;
        }
      }
      sweeps = verified;
      // the copy of the last enqueued sweep's flag may still be in flight: let it land before the pinned
      // buffer goes back to the pool (that sweep's launches were no-ops)
      NDMPS_CHECK_HIP(hipEventSynchronize(hf->ev[(enqueued - 1) & 1]));
    }
    if (remaining != 0) {
      ndmps::set_error("block Jacobi did not converge in %d sweeps (n=%lld, %d of %d matrices left)",
                       kMaxSweepsBlock, (long long)n_max, remaining, batch);
      return NDMPS_ENOCONV;
    }
    hipLaunchKernelGGL(blk_rank_kernel, dim3((unsigned)n_max, B), dim3(256), 0, s, desc, w);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }

  // first h_k[b] eigenvectors of matrix b (columns 0..k-1 of its V_out); asynchronous on the stream
  int vectors(const int64_t* h_k) {
    const unsigned B = (unsigned)batch;
    if (!w.hist_mode) {  // V was accumulated in the loop: sort and sign all of it
      const int gather_grid = (int)std::min<int64_t>(ndmps::ceil_div(n_max * n_max, 256), 2048);
      hipLaunchKernelGGL(blk_gather_kernel, dim3(gather_grid, B), dim3(256), 0, s, desc, w);
      NDMPS_LAUNCH_CHECK();
      return NDMPS_OK;
    }
    std::vector<int> k32(batch);
    int k_max = 1;
    for (int b = 0; b < batch; ++b) {
      k32[b] = (int)h_k[b];
      k_max = std::max(k_max, k32[b]);
    }
    NDMPS_CHECK_HIP(hipMemcpyAsync(d_k, k32.data(), sizeof(int) * batch, hipMemcpyHostToDevice, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));  // k32 lives on this stack frame
    const size_t lds8 = ((size_t)w.np * 17 + 8 * 32 * 33) * sizeof(double);
    const size_t lds4 = ((size_t)w.np * 17 + 4 * 32 * 33) * sizeof(double);
    const dim3 grid((unsigned)ndmps::ceil_div(k_max, 16), B);
    if (lds8 <= (size_t)kBackacc8MaxLds && w.nb / 2 > 4)
      hipLaunchKernelGGL(blk_backacc_kernel<8>, grid, dim3(512), lds8, s, desc, w, d_k);
    else
      hipLaunchKernelGGL(blk_backacc_kernel<4>, grid, dim3(256), lds4, s, desc, w, d_k);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }
};

int make_desc(int batch, double* d_G, int64_t stride_G, const int64_t* h_n, double* d_V, int64_t stride_V,
              double* d_w, int64_t stride_w, double rel_tol, std::vector<BatchDesc>& desc, int64_t& n_max) {
  NDMPS_REQUIRE(batch >= 1 && batch <= 4096, "batch=%d outside [1, 4096]", batch);
  NDMPS_REQUIRE(rel_tol >= 1e-16 && rel_tol <= 1e-6, "rel_tol=%g outside [1e-16, 1e-6]", rel_tol);
  NDMPS_REQUIRE(d_G && d_V && d_w && h_n, "NULL eigen operand");
  desc.resize(batch);
  n_max = 0;
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
    desc[b].rel_tol = rel_tol;
  }
  return NDMPS_OK;
}

}  // namespace

extern "C" int64_t ndmps_syevj_batched_workspace_bytes(int64_t n_max, int batch) {
  if (n_max <= 0 || batch <= 0) return 0;
  return block_layout(n_max, batch).total;
}

extern "C" int64_t ndmps_syevj_workspace_bytes(int64_t n) { return ndmps_syevj_batched_workspace_bytes(n, 1); }

extern "C" int ndmps_syevj_batched_tol_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n,
                                           double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                           double rel_tol, void* d_ws, int64_t ws_bytes, int* h_sweeps,
                                           ndmps_stream_t stream) {
  std::vector<BatchDesc> desc;
  int64_t n_max = 0;
  NDMPS_TRY(make_desc(batch, d_G, stride_G, h_n, d_V, stride_V, d_w, stride_w, rel_tol, desc, n_max));
  BatchedJacobi jac;
  NDMPS_TRY(jac.init(batch, desc, n_max, d_ws, ws_bytes, (hipStream_t)stream));
  NDMPS_TRY(jac.values());
  if (h_sweeps) *h_sweeps = jac.sweeps;
  NDMPS_TRY(jac.vectors(h_n));  // all eigenvectors
  NDMPS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return NDMPS_OK;
}

// Two-phase form used by the sweep: eigenvalues first (the caller derives the kept rank of every
// matrix from them), then only the first h_k[b] eigenvectors.  d_ws must stay untouched in between.
extern "C" int ndmps_syevj_batched_values_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n,
                                              double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                              double rel_tol, void* d_ws, int64_t ws_bytes, int* h_sweeps,
                                              ndmps_stream_t stream) {
  std::vector<BatchDesc> desc;
  int64_t n_max = 0;
  NDMPS_TRY(make_desc(batch, d_G, stride_G, h_n, d_V, stride_V, d_w, stride_w, rel_tol, desc, n_max));
  BatchedJacobi jac;
  NDMPS_TRY(jac.init(batch, desc, n_max, d_ws, ws_bytes, (hipStream_t)stream));
  NDMPS_TRY(jac.values());
  if (h_sweeps) *h_sweeps = jac.sweeps;
  NDMPS_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  return NDMPS_OK;
}

extern "C" int ndmps_syevj_batched_vectors_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n,
                                               double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                               const int64_t* h_k, void* d_ws, int64_t ws_bytes,
                                               ndmps_stream_t stream) {
  NDMPS_REQUIRE(h_k, "NULL rank array");
  std::vector<BatchDesc> desc;
  int64_t n_max = 0;
  NDMPS_TRY(make_desc(batch, d_G, stride_G, h_n, d_V, stride_V, d_w, stride_w, 1e-15, desc, n_max));
  for (int b = 0; b < batch; ++b) NDMPS_REQUIRE(h_k[b] >= 1 && h_k[b] <= h_n[b], "k out of range");
  BatchedJacobi jac;
  // re-attach to the workspace of the values phase without touching the device-side descriptors
  jac.batch = batch;
  jac.n_max = n_max;
  jac.s = (hipStream_t)stream;
  jac.l = block_layout(n_max, batch);
  if (d_ws == nullptr || ws_bytes < jac.l.total) {
    ndmps::set_error("syevj workspace too small");
    return NDMPS_EWORKSPACE;
  }
  char* base = (char*)d_ws;
  for (int i = 0; i < 2; ++i) {
    jac.w.G[i] = (double*)(base + jac.l.off_g[i]);
    jac.w.Q[i] = (double*)(base + jac.l.off_q[i]);
    jac.w.D[i] = (double*)(base + jac.l.off_d[i]);
  }
  jac.w.V = (double*)(base + jac.l.off_v);
  jac.w.pos = (int*)(base + jac.l.off_pos);
  jac.w.rank = (int*)(base + jac.l.off_rank);
  jac.w.invrank = (int*)(base + jac.l.off_invrank);
  jac.w.sign = (double*)(base + jac.l.off_sign);
  jac.w.hist = (double*)(base + jac.l.off_hist);
  jac.w.hist_stride = jac.l.hist_stride;
  jac.w.hist_mode = jac.l.hist;
  jac.w.np = (int)jac.l.np;
  jac.w.nb = (int)jac.l.nb;
  jac.w.bs = block_size_for(n_max);
  jac.desc = (BatchDesc*)(base + jac.l.off_desc);
  jac.d_k = (int*)(base + jac.l.off_k);
  NDMPS_TRY(jac.vectors(h_k));
  return NDMPS_OK;
}

extern "C" int ndmps_syevj_batched_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n, double* d_V,
                                       int64_t stride_V, double* d_w, int64_t stride_w, void* d_ws,
                                       int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream) {
  return ndmps_syevj_batched_tol_f64(batch, d_G, stride_G, h_n, d_V, stride_V, d_w, stride_w, 1e-15, d_ws,
                                     ws_bytes, h_sweeps, stream);
}

extern "C" int ndmps_syevj_f64(double* d_G, int64_t n, double* d_V, double* d_w, void* d_ws,
                               int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream) {
  NDMPS_REQUIRE(n >= 1, "eigen size n=%lld must be positive", (long long)n);
  return ndmps_syevj_batched_f64(1, d_G, n * n, &n, d_V, n * n, d_w, n, d_ws, ws_bytes, h_sweeps, stream);
}
