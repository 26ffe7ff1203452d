// Direct symmetric eigen-solver for the leading k eigenpairs (fp64), batched: the production path of the
// bond-capped sweep (k = chi <= 128 of n = d chi columns).
//
//   G = Q T Q^T      Householder tridiagonalisation, unblocked, the rank-2 update of step j-1 applied while
//                    step j reads the matrix (one pass over the trailing matrix per column)
//   T  -> w          the k_max largest eigenvalues by 65-section on a Sturm sequence (one wave each, no
//                    divisions in the chain)
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
// That launch-per-column structure (trd_column_kernel) is the fall-back now.  Routes of trd_reduce_and_values:
//   orders <= 2048 whose teams fit the chip   trd_team_kernel: the matrix resident in registers, one launch, one
//                                             exchange between the workgroups of a matrix per column
//   above (even, uniform orders)              panel-blocked launches (eig_panel.inc), the last 2048 / 1024 / 512
//                                             columns in the resident kernel
//   everything else, recovery                 panel-blocked launches to the end (from order 1536), column launches
// and of the eigenvectors: one workgroup per matrix (lockstep groups), or across the chip (eig_wide.inc) for more
// than 128 vectors and for one or two matrices of order >= 1024.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"
#include "lanes.h"

namespace {
// `ok ? *p : 0` with the load issued unconditionally (p must be a valid address either way): a load under a condition
// is a branch of its own with a wait behind it, and a kernel that fetches its operands that way fetches them one by one
// (back_rows_step_kernel: 176 loads, 66 waits, 25 us per launch)
__device__ __forceinline__ double load_if(const double* p, bool ok) {
  const double v = *p;
  return ok ? v : 0.0;
}
}  // namespace

namespace {

typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int kTail = 128;        // trailing order finished in LDS by one workgroup
constexpr int kTailLd = 132;      // LDS row stride of the tail matrix (ld = 4 mod 32: conflict-free row walks)
constexpr int kMaxK = 128;        // largest number of eigenvectors (Cholesky factor lives in LDS)
constexpr int kMaxN = 4096;       // three n-vectors of the column kernel live in LDS

struct TrdDesc {       // one per matrix (device array), blockIdx.y selects it
  const double* G_in;  // n x n (ld n)
  double* V_out;       // n x n (ld n): eigenvector c in column c, columns 0..k-1 written
  double* w_out;       // n eigenvalues, descending
  int n;
  int k;               // eigenvectors wanted (vectors phase)
  int status;          // != 0: Cholesky breakdown in the orthonormalisation
  int pad;
};

struct __attribute__((aligned(16))) TeamRec {  // one exchanged value and the (launch, column) it belongs to
  double v;
  unsigned long long tag;
};

struct TeamSync {
  unsigned long long count;  // arrivals of the matrix's workgroups, monotonic over the columns
  int abort;                 // a wait ran out of time: every workgroup of the team leaves
  int pad[317];              // 1280 bytes apart: the counters of a batch sit in different memory channels (the
                             // arrivals and polls of adjacent counters queued behind each other: time ~ batch)
};

struct TrdWork {       // per-matrix strides; everything indexed by blockIdx.y
  double* A;           // [B][n_max][lda] working copy, trailing part updated in place
  double* Vh;          // [B][n_max][lda] reflector j in row j (zeros up to j, 1 at j + 1)
  double* y;           // [B][2][lda] matrix-vector products, by parity of the column
  double* xc;          // [B][2][lda] team kernel: column j of the trailing matrix, by parity of j
  TeamRec* yr;         // [B][2][lda] the same two vectors as self-validating (value, tag) records
  TeamRec* xr;
  TeamSync* sync;      // [B] team kernel: arrival counter of the matrix's workgroups
  double* tau;         // [B][n_max]
  double* d;           // [B][n_max] diagonal of T
  double* e;           // [B][n_max] sub-diagonal of T
  double* lam;         // [B][n_max] eigenvalues of T / bound, descending
  double* mu;          // [B][n_max] shifts of the inverse iteration: lam with coinciding values spread apart (trd_shift_kernel)
  double* bound;       // [B] Gershgorin bound of T
  double* Z;           // [B][n_max][kp] eigenvectors of T
  double* lu;          // [B][4][n_max][kp] dl, 1/d (lowest mantissa bit: rows interchanged), du / d, du2 / d of the pivoted factorisations
  unsigned char* piv;  // [B][n_max][kp] not used any more (the interchange flag rides in 1/d); kept in the layout
  double* Tw;          // [B][t_stride] WY factors of the reflector groups (back-transformation)
  int64_t t_stride;
  double* yb;          // [B][2][kBandMax][lda] band reduction: Y = A V of a panel, by parity of the panel
  double* xb;          // [B][2][kBandMax][lda] band reduction: the next panel's columns, by parity
  double* band;        // [B][n_max][2 kBandMax] band matrix, [column][distance below the diagonal]
  double* qlog;        // [B][q_stride] reflectors of the bulge chase, [sweep][step][BW]
  int64_t q_stride;
  int tail_lower;      // the resident part left only the stored half of the trailing block (trd_sym_kernel)
  int xcd_team, xcd_count;  // resident launches as a 1-D grid placed team by team on the XCDs (team_place); 0: 2-D grid
  int xcd_pair;        // second workgroup of a CU: members in reverse order (early leavers beside late ones)
  double* yrow;        // [B][2][8][lda] half-storage team kernel: row sums per workgroup, by parity of the column
  // panel-blocked reduction (orders above 512, eig_panel.inc)
  double* pv;          // [B][n_max][kPnlNB] reflectors of the current panel, row-major
  double* pw;          // [B][n_max][kPnlNB] their companions w
  double* ypart;       // [B][pnl_blocks_max][lda] tile partials of A v, slot = the other block index
  double* spart;       // [B][pnl_tiles_max] tile partials of v^T A v
  double* ucol;        // [B][2][lda] the next column, un-normalised, by parity of the column
  double* napart;      // [B][2][pnl_blocks_max][kPnlNa] per workgroup of pnl_vec_kernel: partial norm, W^T u, V^T u
  int pnl_blocks_max, pnl_tiles_max;
  long long* stamps;   // [B][16] wall-clock (100 MHz) marks of the single-workgroup kernels' phases (tools/trd_probe.py)
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

// sum over the wave, in every lane (DPP and permlane swaps: lanes.h)
__device__ __forceinline__ double wave_sum(double v) { return ndmps_lanes::sum_adjacent<64>(v); }

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
// A launch is a chain of dependent memory round trips, not a bandwidth problem (the trailing matrix of one
// workgroup is <= 128 KB), so everything whose address is known is requested at once, up front: the
// workgroup's whole share of the trailing matrix (up to 32 tiles of 16 bytes per lane = 512 rows; more rows
// stream afterwards), the three n-vectors of the prologue and the five scalars it broadcasts.  The
// prologue's arithmetic then runs under the flight time of the matrix tiles.
struct __attribute__((aligned(16))) RowVec {  // per-row operands of the body, one 32-byte LDS record
  double vp, wp, vj, pad;
};

//   NR: n-vector elements per thread (n <= 256 NR).  CW: columns per workgroup, 32 or 8: a workgroup draws
//   ~33 GB/s from the Infinity Cache whatever it does, so few matrices are cut into many narrow column
//   blocks (a 512-row launch of one matrix: 12.6 us with 16 workgroups of 32 columns) and many matrices into
//   wide ones (the redundant prologue is paid per workgroup).
template <int NR, int CW>
__global__ void __launch_bounds__(256)
trd_column_kernel(const TrdDesc* __restrict__ desc, TrdWork w, int j) {
  constexpr int LPR = CW / 2;        // lanes per row (16 bytes each)
  constexpr int RPW = 64 / LPR;      // rows per wave instruction
  constexpr int RPI = 4 * RPW;       // rows per workgroup iteration
  constexpr int PRE = 512 / RPI;     // tiles prefetched per lane: 512 rows
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  if (j >= n - kTail) return;              // this matrix is (or will be) finished by the tail kernel
  const int c0 = blockIdx.x * CW;
  if (c0 >= n || c0 + CW <= j + 1) return;  // columns <= j are finished
  const int lda = w.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  double* A = w.A + b * w.n_max * lda;
  double* Vh = w.Vh + b * w.n_max * lda;
  double* ybuf = w.y + b * 2 * lda;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  RowVec* rv = reinterpret_cast<RowVec*>(lds);  // [n]
  __shared__ double red_a[4], red_b[4];
  __shared__ double part[4][CW];

  // ---- every load whose address is known: matrix tiles first (the longest flight)
  const int q = lane / LPR, p = lane % LPR;
  const int c = c0 + 2 * p;
  const bool col_ok = c < lda;  // lda is even: c + 1 < lda too; the padding column holds zeros
  const int first = j + 1 + RPW * wave + q;
  double2 a[PRE];
#pragma unroll
  for (int u = 0; u < PRE; ++u) {
    const int i = first + RPI * u;
    a[u] = (i < n && col_ok) ? *reinterpret_cast<const double2*>(A + (int64_t)i * lda + c) : make_double2(0.0, 0.0);
  }
  const double* rowj = A + (int64_t)j * lda;
  double yv[NR], vv[NR], rj[NR];
  double taup = 0.0, y_j = 0.0, y_j1 = 0.0, v_j1 = 0.0;
  const double r_j1 = rowj[j + 1];
  if (j >= 1) {
    const double* yprev = ybuf + ((j - 1) & 1) * lda;
    const double* vprev = Vh + (int64_t)(j - 1) * lda;
    taup = w.tau[b * w.n_max + j - 1];
    y_j = yprev[j];
    y_j1 = yprev[j + 1];
    v_j1 = vprev[j + 1];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = tid + 256 * r;
      const bool in = i >= j && i < n;
      yv[r] = in ? yprev[i] : 0.0;
      vv[r] = in ? vprev[i] : 0.0;
    }
  } else {
#pragma unroll
    for (int r = 0; r < NR; ++r) yv[r] = vv[r] = 0.0;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = tid + 256 * r;
    rj[r] = (i >= j && i < n) ? rowj[i] : 0.0;
  }

  // ---- prologue a: w' of step j - 1 (one barrier per reduction: each has its own slots)
  double dot = 0.0;
#pragma unroll
  for (int r = 0; r < NR; ++r) dot = fma(yv[r], vv[r], dot);
  dot = wave_sum(dot);
  if (lane == 0) red_a[wave] = dot;
  __syncthreads();
  dot = (red_a[0] + red_a[1]) + (red_a[2] + red_a[3]);
  const double al = 0.5 * taup * taup * dot;
  const double wpj = taup * y_j - al;  // v'[j] = 1
  // ---- prologue b: row j with the update applied; x = row[j + 1:], sigma = |x[1:]|^2
  double wq[NR];
  double sigma = 0.0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = tid + 256 * r;
    wq[r] = taup * yv[r] - al * vv[r];
    rj[r] -= wq[r] + wpj * vv[r];  // v'[j] w'[i] + w'[j] v'[i]
    if (i > j + 1 && i < n) sigma = fma(rj[r], rj[r], sigma);
  }
  sigma = wave_sum(sigma);
  if (lane == 0) red_b[wave] = sigma;
  __syncthreads();
  sigma = (red_b[0] + red_b[1]) + (red_b[2] + red_b[3]);
  const double alpha = r_j1 - ((taup * y_j1 - al * v_j1) + wpj * v_j1);
  double beta, tau, scale;
  householder(alpha, sigma, beta, tau, scale);
  const bool writer = (int)blockIdx.x == (j + 1) / CW;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = tid + 256 * r;
    if (i < n) {
      const double v = i > j + 1 ? rj[r] * scale : (i == j + 1 ? 1.0 : 0.0);
      RowVec rec;
      rec.vp = vv[r];
      rec.wp = wq[r];
      rec.vj = v;
      rec.pad = 0.0;
      rv[i] = rec;
      if (writer) Vh[(int64_t)j * lda + i] = v;
      if (writer && i == j) {
        w.d[b * w.n_max + j] = rj[r];
        w.e[b * w.n_max + j] = beta;
        w.tau[b * w.n_max + j] = tau;
      }
    } else if (writer && i < lda) {
      Vh[(int64_t)j * lda + i] = 0.0;
    }
  }
  __syncthreads();

  // ---- body
  double wc0 = 0.0, wc1 = 0.0, vc0 = 0.0, vc1 = 0.0;
  if (j >= 1 && col_ok) {
    if (c < n) { wc0 = rv[c].wp; vc0 = rv[c].vp; }
    if (c + 1 < n) { wc1 = rv[c + 1].wp; vc1 = rv[c + 1].vp; }
  }
  double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
  for (int u = 0; u < PRE; ++u) {
    const int i = first + RPI * u;
    if (i < n && col_ok) {
      const RowVec rec = rv[i];
      if (j >= 1) {
        a[u].x -= fma(rec.vp, wc0, rec.wp * vc0);
        a[u].y -= fma(rec.vp, wc1, rec.wp * vc1);
        *reinterpret_cast<double2*>(A + (int64_t)i * lda + c) = a[u];
      }
      acc0 = fma(a[u].x, rec.vj, acc0);
      acc1 = fma(a[u].y, rec.vj, acc1);
    }
  }
  // rows beyond the prefetched 512 (orders above 512 + j)
  constexpr int U = 8;
  for (int i0 = first + RPI * PRE; i0 < n; i0 += RPI * U) {
    double2 t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + RPI * u;
      t[u] = (i < n && col_ok) ? *reinterpret_cast<const double2*>(A + (int64_t)i * lda + c) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + RPI * u;
      if (i < n && col_ok) {
        const RowVec rec = rv[i];
        if (j >= 1) {
          t[u].x -= fma(rec.vp, wc0, rec.wp * vc0);
          t[u].y -= fma(rec.vp, wc1, rec.wp * vc1);
          *reinterpret_cast<double2*>(A + (int64_t)i * lda + c) = t[u];
        }
        acc0 = fma(t[u].x, rec.vj, acc0);
        acc1 = fma(t[u].y, rec.vj, acc1);
      }
    }
  }
  // ---- epilogue: rows -> one value per column, fixed order
  acc0 = ndmps_lanes::sum_strided<LPR>(acc0);
  acc1 = ndmps_lanes::sum_strided<LPR>(acc1);
  if (q == 0) {
    part[wave][2 * p] = acc0;
    part[wave][2 * p + 1] = acc1;
  }
  __syncthreads();
  if (tid < CW && c0 + tid < lda)
    ybuf[(j & 1) * lda + c0 + tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// ----------------------------------------------------------------------------------------- team kernel
// The same reduction with the trailing matrix RESIDENT IN REGISTERS: one launch for all columns 0 .. n - 129
// of every matrix (n <= 512).  Workgroup (r, b) keeps columns [32 r, 32 r + 32) of matrix b over all 512 rows
// in its registers (16 tiles of 4 doubles per lane = 128 KB per workgroup, two workgroups per CU), so the
// column launches' traffic -- one read and one write of the trailing matrix per column, the bound of
// trd_column_kernel -- disappears; what remains per column is the exchange the kernel boundary
// used to provide: every workgroup publishes its 32 entries of y_j = A v_j, the owner of column j + 1 publishes
// that column (row j + 1 by symmetry), the matrix's workgroups meet at a counter in global memory and each
// reads the two n-vectors back.  Exchange data moves with agent-scope atomic stores / loads (write-through,
// never served from a stale L2 line of another XCD); release / acquire fences order them around the counter.
//
// The workgroups of a matrix (its team, 16 at n = 512) must be resident together.  Liveness rests on three
// invariants kept by the host side (ndmps_syevd_topk_values_f64):
//   * a launch holds at most `team_slots()` workgroups (occupancy x compute units, at most two per CU): every team
//     of a launch is resident whatever order the dispatcher places workgroups in; larger batches go in several
//     launches;
//   * team launches in flight per device never hold more workgroups together than the device keeps resident: the turn
//     counts units (a launch that fits half the slots takes one of two, every other launch both) and is taken ON THE
//     DEVICE (ndmps::Turn, util.hip: a
//     one-thread kernel in front spins on a lock word, one behind gives it back), because two team kernels from
//     different streams could each hold slots the other's partial teams wait for.  Ordinary kernels sharing the
//     GPU end by themselves;
//   * every wait is bounded: after kTeamSpinTicks (3 s) the waiting workgroup raises the team's abort flag, every
//     member leaves at its next wait and status 2 reaches the host, which re-runs the reduction on the column
//     launches (trd_column_kernel) and counts the event (ndmps_syevd_topk_team_fallbacks).
//
// Arithmetic: the prologue and the per-element update are those of trd_column_kernel (same redundant,
// bit-identical O(n) part in every workgroup); only the association of the column sums differs (rows are
// dealt to lanes independently of j).
constexpr long long kTeamSpinTicks = 300000000LL;  // 3 s of the 100 MHz wall clock

__device__ __forceinline__ double team_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void team_store(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

// (value, tag) records: ONE 16-byte agent-scope store / load each, so a reader sees either the old record or the
// new one, never a mix; the tag names the launch and the column, a record is valid when it carries the expected tag
typedef unsigned int team_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rec_store(TeamRec* p, double v, unsigned long long tag) {
  const unsigned long long vb = (unsigned long long)__double_as_longlong(v);
  team_u32x4 q = {(unsigned)vb, (unsigned)(vb >> 32), (unsigned)tag, (unsigned)(tag >> 32)};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ bool rec_ok(const team_u32x4& q, unsigned long long tag, double& v) {
  v = __longlong_as_double((long long)(((unsigned long long)q[1] << 32) | q[0]));
  return (((unsigned long long)q[3] << 32) | q[2]) == tag;
}
// every record of a poll in flight, then one wait that names all the results (the compiler does not know these loads)
__device__ __forceinline__ void rec_issue(team_u32x4& q, const TeamRec* p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q) : "v"(p) : "memory");
}
// the same with a wave-uniform base (scalar registers) and a 32-bit byte offset per lane; IMM: 0 or 16 bytes on top
template <int IMM>
__device__ __forceinline__ void rec_issue_s(team_u32x4& q, unsigned off, const TeamRec* base) {
  static_assert(IMM == 0 || IMM == 16, "entry 0 or 1 behind the base");
  if constexpr (IMM == 0) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(q) : "v"(off), "s"(base) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, %2 offset:16 sc1" : "=v"(q) : "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ void rec_wait3(team_u32x4& q0, team_u32x4& q1, team_u32x4& q2) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2) : : "memory");
}
__device__ __forceinline__ void rec_wait4(team_u32x4& q0, team_u32x4& q1, team_u32x4& q2, team_u32x4& q3) {
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "memory");
}
template <int N>
__device__ __forceinline__ void rec_wait(team_u32x4 (&q)[N]) {
  static_assert(N == 4 || N == 8 || N == 16, "2, 4 or 8 rows per thread");
  rec_wait4(q[0], q[1], q[2], q[3]);  // vmcnt(0): everything has landed; the further waits only name the registers
  if constexpr (N >= 8) rec_wait4(q[4], q[5], q[6], q[7]);
  if constexpr (N >= 16) {
    rec_wait4(q[8], q[9], q[10], q[11]);
    rec_wait4(q[12], q[13], q[14], q[15]);
  }
}
__device__ __forceinline__ unsigned long long team_poll(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// TAGGED: the exchange without a meeting.  Every exchanged value travels as a (value, tag) record; a consumer polls
// the records it needs until they carry the tag of the column -- one memory round trip between a producer's store
// and a consumer's use, instead of store acknowledgement + counter update + counter poll + data load.  The records
// of a matrix are double-buffered by the parity of the column like the plain vectors: nobody can be two columns
// ahead of anybody, because publishing column j + 2 needs everybody's column j + 1.
// Workgroups go to the 8 XCDs round robin in dispatch order (workgroup L runs on XCD L % 8; each XCD has its own L2).
// A 1-D grid of team_size x count workgroups is dealt so that the members of a team share an XCD: their exchange --
// a handful of 4 KB vectors and one counter per column -- then stays in that XCD's L2.  Nothing depends on the
// placement for correctness (the exchange is agent-scope either way).  The last count % 8 teams are dealt as they come.
__device__ __forceinline__ bool team_place(const TrdWork& w, int& member, int& team) {
  if (w.xcd_count <= 0) {
    member = blockIdx.x;
    team = blockIdx.y;
    return true;
  }
  const int L = blockIdx.x, full = w.xcd_count & ~7;
  if (L < full * w.xcd_team) {
    const int t = L >> 3;
    member = t % w.xcd_team;
    team = (t / w.xcd_team) * 8 + (L & 7);
    if (w.xcd_pair && ((t >> 5) & 1)) {  // 32 CUs per XCD: workgroup t + 32 joins t's CU, with the members in reverse order
      const int blk = min(w.xcd_team, 32), pos = member % blk;  // (xcd_pair only if 32 % team == 0 or team % 32 == 0)
      member += blk - 1 - 2 * pos;
    }
  } else {
    const int r = w.xcd_count - full, l2 = L - full * w.xcd_team;
    member = l2 / r;
    team = full + l2 % r;
  }
  return true;
}

// CW: columns per workgroup, 32 (16 workgroups per order-512 matrix) or 8 (64 of them, for one to four matrices:
// the tile loop is a quarter as long, the per-column arithmetic of the slowest member shrinks with it).
// MR: the largest order an instantiation takes, 256 * NR rows (512: the matrices of the lockstep groups; 1024 and 2048,
// with 8-column blocks and tagged records only: one or a few big matrices spread over the whole chip -- an order-2048
// matrix is 128 registers per thread in 256 workgroups).  Dynamic LDS: MR row records.
template <int NR, bool TAGGED, int CW = 32>
__global__ void __launch_bounds__(256, NR <= 4 ? 2 : 1)
trd_team_kernel(TrdDesc* __restrict__ desc, TrdWork w, int b0, unsigned epoch) {
  constexpr int MR = 256 * NR;
  constexpr int LPR = CW / 4;        // lanes per row, 4 columns (32 bytes) each
  constexpr int RPW = 64 / LPR;      // rows per wave instruction
  constexpr int RPI = 4 * RPW;       // rows per tile of the workgroup
  constexpr int NT = MR / RPI;       // tiles per lane: MR rows
  int member, team;
  if (!team_place(w, member, team)) return;
  TrdDesc& d = desc[b0 + team];
  const int n = d.n;
  const int J = n - kTail;           // columns 0 .. J - 1 are reduced here, the rest by the tail kernel
  const int c0 = member * CW;
  if (J <= 0 || c0 >= n) return;
  const int nblk = (n + CW - 1) / CW;
  const int j_last = min(J - 1, c0 + CW - 2);  // the block's columns are finished once j + 1 >= c0 + CW
  const int lda = w.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = b0 + team;
  double* A = w.A + b * w.n_max * lda;
  double* Vh = w.Vh + b * w.n_max * lda;
  double* ybuf = w.y + b * 2 * lda;
  double* xc = w.xc + b * 2 * lda;
  TeamSync* sync = w.sync + b;
  TeamRec* yr = w.yr + b * 2 * lda;
  TeamRec* xr = w.xr + b * 2 * lda;
  const unsigned long long tag_base = (unsigned long long)epoch << 32;
  extern __shared__ __attribute__((aligned(32))) double team_dyn[];  // 32: row records are read and written as two b128
  RowVec* rv = reinterpret_cast<RowVec*>(team_dyn);  // [MR]
  __shared__ double red_a[4], red_b[4];
  __shared__ double part[4][CW];
  __shared__ int go;

  // the RPW = 8 lanes that share a column group are ADJACENT (q fastest): the column sums fold with lane-xor 1, 2, 4,
  // which are register moves inside a row of 16 lanes (xor 8 / 16 / 32 go through the LDS crossbar: 0.4 us per
  // column); the matrix itself is loaded and stored once per launch, its coalescing does not matter
  const int q = lane % RPW, p = lane / RPW;
  const int c = c0 + 4 * p;
  const int row0 = RPW * wave + q;
  double a[NT][4];
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int i = row0 + RPI * u;
    double2 lo = make_double2(0.0, 0.0), hi = lo;
    if (i < n) {  // lda is even; columns >= n up to lda hold zeros
      if (c < lda) lo = *reinterpret_cast<const double2*>(A + (int64_t)i * lda + c);
      if (c + 2 < lda) hi = *reinterpret_cast<const double2*>(A + (int64_t)i * lda + c + 2);
    }
    a[u][0] = lo.x;
    a[u][1] = lo.y;
    a[u][2] = hi.x;
    a[u][3] = hi.y;
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {  // rows beyond the order never change: zero operands
    const int i = tid + 256 * r;
    if (i >= n) {
      RowVec z;
      z.vp = z.wp = z.vj = z.pad = 0.0;
      rv[i] = z;
    }
  }
  double vv[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) vv[r] = 0.0;
  // byte offset of a lane's row in the record arrays, -16 for rows beyond the order (recomputed where it is used: the
  // 32-column blocks have no register to keep it in)
  auto row_off_of = [&](int r) { return tid + 256 * r < n ? 16 * (tid + 256 * r) : -16; };
  double taup = 0.0;
  unsigned long long target = 0;
  int failed = 0;
  long long t_start = wall_clock64();
  __syncthreads();

#ifdef NDMPS_TEAM_STAMPS
  long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TEAM_STAMP(k) { const long long now_ = (long long)__builtin_readcyclecounter(); st[k] += now_ - last_; last_ = now_; }
  long long last_ = (long long)__builtin_readcyclecounter();
#else
#define TEAM_STAMP(k)
#endif
  for (int j = 0; j <= j_last; ++j) {
    // ---- operands of the prologue: y of step j - 1 and column j (both published before the last meeting)
    double yv[NR], rj[NR];
    double y_j = 0.0, y_j1 = 0.0, v_j1 = 0.0, r_j1;
    if (j >= 1) {
      const double* yprev = ybuf + ((j - 1) & 1) * lda;
      const double* xcol = xc + (j & 1) * lda;
      if (TAGGED) {
        const TeamRec* yp = yr + ((j - 1) & 1) * lda;
        const TeamRec* xp = xr + (j & 1) * lda;
        const unsigned long long want = tag_base | (unsigned long long)j;
        // a lane's rows at and below j (rows above it and rows beyond the order look at entry j, valid like any other):
        // one max per row on the byte offsets; the bases are wave-uniform
        int row_off[NR];
        unsigned off[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          row_off[r] = row_off_of(r);
          off[r] = (unsigned)max(row_off[r], 16 * j);
        }
        auto give_up = [&](unsigned polls) -> bool {  // every 256th unsuccessful poll: the abort flag and the clock
          if ((polls & 255u) != 0) return false;
          if (__hip_atomic_load(&sync->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
              wall_clock64() - t_start > kTeamSpinTicks) {
            __hip_atomic_store(&sync->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            failed = 1;
            return true;
          }
          return false;
        };
        // the three scalars every thread needs are entries j and j + 1: one address for the whole wave each.  Teams of
        // 128 / 256 workgroups wait on THESE first (192 bytes per wave and poll) and fetch their rows once they carry
        // the column's tag: a poll of all rows is 4 - 17 MB across the chip, and the rows of a column are published
        // within a fraction of a microsecond of each other.  A team inside one XCD polls everything at once (one round
        // trip less; its polls stay in that XCD's L2).
        constexpr bool kScalarsFirst = NR >= 4;
        // 32-column blocks keep the counter: their tagged variant (128 registers of matrix per thread) spills under the
        // 256-register cap of two workgroups per CU, and every spilling build of it waited out its 3 s (round 4, twice)
        static_assert(!TAGGED || CW == 8, "tagged records: 8-column blocks only");
        bool ok = !kScalarsFirst;
        for (unsigned polls = 1; !ok; ++polls) {
          team_u32x4 qj, qj1, qx1;
          rec_issue_s<0>(qj, 0u, yp + j);
          rec_issue_s<16>(qj1, 0u, yp + j);
          rec_issue_s<16>(qx1, 0u, xp + j);
          rec_wait3(qj, qj1, qx1);
          const bool o0 = rec_ok(qj, want, y_j), o1 = rec_ok(qj1, want, y_j1), o2 = rec_ok(qx1, want, r_j1);
          ok = o0 && o1 && o2;
          if (!ok && give_up(polls)) break;
        }
        ok = false;
        for (unsigned polls = 1; !ok && !failed; ++polls) {
          team_u32x4 qq[2 * NR], qj, qj1, qx1;
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            rec_issue_s<0>(qq[2 * r], off[r], yp);
            rec_issue_s<0>(qq[2 * r + 1], off[r], xp);
          }
          if (!kScalarsFirst) {
            rec_issue_s<0>(qj, 0u, yp + j);
            rec_issue_s<16>(qj1, 0u, yp + j);
            rec_issue_s<16>(qx1, 0u, xp + j);
          }
          rec_wait(qq);
          ok = true;
          if (!kScalarsFirst) {
            rec_wait3(qj, qj1, qx1);
            const bool o0 = rec_ok(qj, want, y_j), o1 = rec_ok(qj1, want, y_j1), o2 = rec_ok(qx1, want, r_j1);
            ok = o0 && o1 && o2;
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const bool oy = rec_ok(qq[2 * r], want, yv[r]), ox = rec_ok(qq[2 * r + 1], want, rj[r]);
            ok = ok && oy && ox;
          }
          if (!ok && give_up(polls)) break;
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (row_off[r] < 16 * j) yv[r] = rj[r] = 0.0;  // finished rows, rows beyond the order (-16)
        }
        v_j1 = rv[j + 1].vj;
      } else {
      y_j = team_load(yprev + j);
      y_j1 = team_load(yprev + j + 1);
      r_j1 = team_load(xcol + j + 1);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = tid + 256 * r;
        const bool in = i >= j && i < n;
        yv[r] = in ? team_load(yprev + i) : 0.0;
        rj[r] = in ? team_load(xcol + i) : 0.0;
      }
      v_j1 = rv[j + 1].vj;  // still the reflector of step j - 1
      }
    } else {
      r_j1 = A[1];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = tid + 256 * r;
        yv[r] = 0.0;
        rj[r] = i < n ? A[i] : 0.0;
      }
    }
    TEAM_STAMP(0)
    // ---- prologue a: w' of step j - 1
    double dot = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) dot = fma(yv[r], vv[r], dot);
    dot = wave_sum(dot);
    if (lane == 0) red_a[wave] = dot;
    if (TAGGED) {
      if (__syncthreads_or(failed)) {  // a wait ran out of time somewhere in the team: everybody leaves
        if (tid == 0) d.status = 2;
        return;
      }
    } else {
      __syncthreads();
    }
    TEAM_STAMP(1)
    dot = (red_a[0] + red_a[1]) + (red_a[2] + red_a[3]);
    const double al = 0.5 * taup * taup * dot;
    const double wpj = taup * y_j - al;  // v'[j] = 1
    // ---- prologue b: row j with the update applied; x = row[j + 1:], sigma = |x[1:]|^2
    double wq[NR];
    double sigma = 0.0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = tid + 256 * r;
      wq[r] = taup * yv[r] - al * vv[r];
      rj[r] -= wq[r] + wpj * vv[r];
      if (i > j + 1 && i < n) sigma = fma(rj[r], rj[r], sigma);
    }
    sigma = wave_sum(sigma);
    if (lane == 0) red_b[wave] = sigma;
    __syncthreads();
    TEAM_STAMP(2)
    sigma = (red_b[0] + red_b[1]) + (red_b[2] + red_b[3]);
    const double alpha = r_j1 - ((taup * y_j1 - al * v_j1) + wpj * v_j1);
    double beta, tau, scale;
    householder(alpha, sigma, beta, tau, scale);
    const bool writer = member == (j + 1) / CW;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int i = tid + 256 * r;
      if (i < n) {
        const double v = i > j + 1 ? rj[r] * scale : (i == j + 1 ? 1.0 : 0.0);
        RowVec rec;
        rec.vp = vv[r];
        rec.wp = wq[r];
        rec.vj = v;
        rec.pad = 0.0;
        rv[i] = rec;
        vv[r] = v;
      }
    }
    if (writer) {  // uniform: one workgroup of the team per column
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        const int i = tid + 256 * r;
        if (i < lda) Vh[(int64_t)j * lda + i] = vv[r];  // zeros beyond the order
        if (i == j) {
          w.d[b * w.n_max + j] = rj[r];
          w.e[b * w.n_max + j] = beta;
          w.tau[b * w.n_max + j] = tau;
        }
      }
    }
    taup = tau;
    __syncthreads();
    TEAM_STAMP(3)

    // ---- body: pending update applied in the registers, column sums with the new reflector
    double wc[4], vc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = min(c + k, MR - 1);  // records of columns >= n are zero
      wc[k] = rv[cc].wp;
      vc[k] = rv[cc].vp;
    }
    const int u0 = (j + 1) / RPI;  // tiles made of finished rows only
    const int u_end = (n + RPI - 1) / RPI;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    // finished tiles are skipped in PAIRS (one uniform branch per pair): a basic block then holds two tiles -- eight
    // independent FMA chains, both row records in flight together.  A finished tile inside a live pair is updated like
    // the other one: its rows have v_j = 0 and are never read again (half a tile of extra work per column on average).
#pragma unroll
    for (int g = 0; g < NT / 2; ++g) {
      if (2 * g + 1 >= u0 && 2 * g < u_end) {  // u_end: tiles of rows beyond the order hold zeros
        const RowVec ra = rv[row0 + RPI * (2 * g)], rb = rv[row0 + RPI * (2 * g + 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          a[2 * g][k] = fma(-ra.wp, vc[k], fma(-ra.vp, wc[k], a[2 * g][k]));
          a[2 * g + 1][k] = fma(-rb.wp, vc[k], fma(-rb.vp, wc[k], a[2 * g + 1][k]));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fma(a[2 * g][k], ra.vj, acc[k]);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = fma(a[2 * g + 1][k], rb.vj, acc[k]);
      }
    }
    TEAM_STAMP(4)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = ndmps_lanes::sum_adjacent<RPW>(acc[k]);
    if (q == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) part[wave][4 * p + k] = acc[k];
    }
    __syncthreads();
    if (tid < CW && c0 + tid < lda) {
      const double yc = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
      if (!TAGGED || j + 1 >= J) team_store(ybuf + (j & 1) * lda + c0 + tid, yc);  // plain vector: the tail kernel reads it
      if (TAGGED) rec_store(yr + (j & 1) * lda + c0 + tid, yc, tag_base | (unsigned long long)(j + 1));
    }
    TEAM_STAMP(5)
    if (j + 1 >= J) break;  // the tail kernel continues from the stored matrix
    // ---- column j + 1 of the updated matrix, by its owner (publishing it before the column sums are folded gains
    //      nothing at orders <= 1024 and costs the 32-column kernel its last registers)
    if (member == (j + 1) / CW) {
      const int kk = j + 1 - c0;
      if (p == kk / 4) {
        double* out = xc + ((j + 1) & 1) * lda;
        // the column's sixteen values first (one unrolled copy per column slot: a run-time register index would
        // send the tiles to scratch), then one store loop (storing straight from the tiles, one loop per slot, costs
        // the 32-column kernel 190 spilled registers)
        double colv[NT];
#define NDMPS_TEAM_PICK(KS) _Pragma("unroll") for (int u = 0; u < NT; ++u) colv[u] = a[u][KS];
        switch (kk % 4) {
          case 0: NDMPS_TEAM_PICK(0) break;
          case 1: NDMPS_TEAM_PICK(1) break;
          case 2: NDMPS_TEAM_PICK(2) break;
          default: NDMPS_TEAM_PICK(3) break;
        }
#undef NDMPS_TEAM_PICK
#pragma unroll
        for (int u = 0; u < NT; ++u) {
          const int i = row0 + RPI * u;
          if (u >= u0 && i > j && i < n) {
            if (TAGGED) rec_store(xr + ((j + 1) & 1) * lda + i, colv[u], tag_base | (unsigned long long)(j + 1));
            else team_store(out + i, colv[u]);
          }
        }
      }
    }
    if (TAGGED) {
      t_start = wall_clock64();  // the next prologue waits for the records themselves
      continue;
    }
    // ---- meeting of the matrix's workgroups (a block on its last column only announces itself)
    target += (unsigned long long)(nblk - (j + 1) / CW);
    // exchange data are write-through agent-scope stores: once every storing wave has seen them acknowledged
    // (vmcnt(0); a workgroup-scope release fence does not wait for the vector memory counter on this target) they
    // are visible to the agent-scope loads of the other workgroups; no L2 write-back / invalidate, which would
    // also flush what concurrent kernels of other streams keep there (an agent-scope release cost ~3x the time)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(&sync->count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 1;
      if (j < j_last) {
        const long long t0 = wall_clock64();
        // the counter is polled through the scalar unit with the cache bypassed (0.47 us per hand-off between two
        // workgroups, 0.65 with agent-scope vector loads: tools/scratch/xcc_probe.hip); the abort flag and the
        // clock are looked at every 64th poll
        unsigned polls = 0;
        while (team_poll(&sync->count) < target) {
          if ((++polls & 63u) == 0) {
            if (__hip_atomic_load(&sync->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
              ok = 0;
              break;
            }
            if (wall_clock64() - t0 > kTeamSpinTicks) {
              __hip_atomic_store(&sync->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              d.status = 2;
              ok = 0;
              break;
            }
          }
        }
      }
      go = ok;
    }
    __syncthreads();
    TEAM_STAMP(6)
    if (!go) return;
  }
#ifdef NDMPS_TEAM_STAMPS
  if (tid == 0 && member == nblk - 1)
    for (int q = 0; q < 8; ++q) w.stamps[b * 16 + q] = st[q];
#endif

  // ---- hand-over to the tail kernel: the trailing block as the column launches would have left it (updates
  //      through J - 2 applied, step J - 1 pending in y / Vh / tau)
  if (j_last == J - 1) {
    const int u0 = J / RPI;
#pragma unroll
    for (int u = 0; u < NT; ++u) {
      const int i = row0 + RPI * u;
      if (u >= u0 && i < n) {
        if (c < lda) *reinterpret_cast<double2*>(A + (int64_t)i * lda + c) = make_double2(a[u][0], a[u][1]);
        if (c + 2 < lda) *reinterpret_cast<double2*>(A + (int64_t)i * lda + c + 2) = make_double2(a[u][2], a[u][3]);
      }
    }
  }
}

// --------------------------------------------------------------------------------------- tail kernel
// One workgroup per matrix finishes columns J = max(n - 128, 0) .. n - 2 with the trailing matrix in LDS.
// 512 threads: thread (r, t) = (tid / 4, tid % 4) owns row r of the trailing matrix, column pairs t mod 4.
// Four barriers per column; per-column operands (v', w', v_j) sit in one 32-byte record per column.
__global__ void __launch_bounds__(512) trd_tail_kernel(const TrdDesc* __restrict__ desc, TrdWork w) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  const int J = max(n - kTail, 0), m = n - J;
  const int lda = w.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const double* A = w.A + b * w.n_max * lda;
  double* Vh = w.Vh + b * w.n_max * lda;
  const double* ybuf = w.y + b * 2 * lda;
  double* dd = w.d + b * w.n_max;
  double* ee = w.e + b * w.n_max;
  double* tt = w.tau + b * w.n_max;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* S = lds;                                                 // [kTail][kTailLd]
  RowVec* cv = reinterpret_cast<RowVec*>(lds + kTail * kTailLd);   // [kTail] {v', w', v_j}
  __shared__ double red_s[8], red_d[8];

  // pending update of the last column launch
  if (J >= 1) {
    const double* yprev = ybuf + ((J - 1) & 1) * lda;
    const double* vprev = Vh + (int64_t)(J - 1) * lda;
    const double taup = tt[J - 1];
    double dot = 0.0, vv = 0.0, y0 = 0.0;
    if (tid < m) {
      vv = vprev[J + tid];
      y0 = yprev[J + tid];
      dot = y0 * vv;
    }
    dot = wave_sum(dot);
    if (lane == 0) red_s[wave] = dot;
    __syncthreads();
    dot = 0.0;
#pragma unroll
    for (int x = 0; x < 8; ++x) dot += red_s[x];
    const double al = 0.5 * taup * taup * dot;
    if (tid < m) {
      cv[tid].vp = vv;
      cv[tid].wp = taup * y0 - al * vv;
    }
  } else if (tid < m) {
    cv[tid].vp = 0.0;
    cv[tid].wp = 0.0;
  }
  __syncthreads();
  for (int e = tid; e < m * m; e += 512) {
    const int r = e / m, c = e % m;
    const int gr = J + r, gc = J + c;
    const bool mirror = w.tail_lower && (gr >> 5) < (gc >> 5);  // half storage (trd_sym_kernel): blocks above the diagonal are not stored
    const double av = mirror ? A[(int64_t)gc * lda + gr] : A[(int64_t)gr * lda + gc];
    S[r * kTailLd + c] = av - (cv[r].vp * cv[c].wp + cv[r].wp * cv[c].vp);
  }
  __syncthreads();
  if (tid < kTail) {  // the update is applied: nothing pending; columns >= m never contribute
    cv[tid].vp = 0.0;
    cv[tid].wp = 0.0;
    cv[tid].vj = 0.0;
  }
  for (int e = tid; e < kTail * kTailLd; e += 512) {
    const int r = e / kTailLd, c = e % kTailLd;
    if (r >= m || c >= m) S[e] = 0.0;
  }
  __syncthreads();

  const int r = tid >> 2, t = tid & 3;
  for (int jj = 0; jj + 1 < m; ++jj) {
    const int j = J + jj;
    // ---- A: row jj with the pending update -> x, sigma; alpha and d_j from LDS directly
    const double vpj = cv[jj].vp, wpj = cv[jj].wp;
    double a = 0.0, sigma = 0.0;
    if (tid < m && tid > jj) {
      a = S[jj * kTailLd + tid] - (vpj * cv[tid].wp + wpj * cv[tid].vp);
      if (tid > jj + 1) sigma = a * a;
    }
    sigma = wave_sum(sigma);
    if (lane == 0) red_s[wave] = sigma;
    const double alpha = S[jj * kTailLd + jj + 1] - (vpj * cv[jj + 1].wp + wpj * cv[jj + 1].vp);
    __syncthreads();
    sigma = 0.0;
#pragma unroll
    for (int x = 0; x < 8; ++x) sigma += red_s[x];
    double beta, tau, scale;
    householder(alpha, sigma, beta, tau, scale);
    // ---- B: reflector
    const double vme = tid > jj + 1 ? a * scale : (tid == jj + 1 ? 1.0 : 0.0);
    if (tid < m) cv[tid].vj = vme;
    if (tid == 0) {
      dd[j] = S[jj * kTailLd + jj] - 2.0 * vpj * wpj;
      ee[j] = beta;
      tt[j] = tau;
    }
    __syncthreads();
    for (int i = tid; i < lda; i += 512) Vh[(int64_t)j * lda + i] = (i >= J && i < n) ? cv[i - J].vj : 0.0;
    // ---- C: one pass: apply the pending update, y = S v_j (rows and columns > jj), pairs of columns
    double acc = 0.0;
    if (r < m && r > jj) {
      const RowVec me = cv[r];
      double* row = S + r * kTailLd;
      const int it0 = (jj + 1) >> 3;
#pragma unroll 4
      for (int it = it0; it < kTail / 8; ++it) {
        const int c = 2 * (t + 4 * it);
        // columns <= jj inside the first pair block are finished: their entries are never read again and
        // v_j is zero there, so they may be updated like the others
        double2 sv = *reinterpret_cast<double2*>(row + c);
        const RowVec c0v = cv[c], c1v = cv[c + 1];
        sv.x -= me.vp * c0v.wp + me.wp * c0v.vp;
        sv.y -= me.vp * c1v.wp + me.wp * c1v.vp;
        acc = fma(sv.x, c0v.vj, acc);
        acc = fma(sv.y, c1v.vj, acc);
        *reinterpret_cast<double2*>(row + c) = sv;
      }
    }
    acc = ndmps_lanes::sum_adjacent<4>(acc);
    const double vjr = r < kTail ? cv[r].vj : 0.0;
    double dot = (t == 0 && r < m && r > jj) ? acc * vjr : 0.0;
    dot = wave_sum(dot);
    if (lane == 0) red_d[wave] = dot;
    __syncthreads();
    dot = 0.0;
#pragma unroll
    for (int x = 0; x < 8; ++x) dot += red_d[x];
    // ---- D: w of this step becomes the pending update
    const double al = 0.5 * tau * tau * dot;
    if (t == 0 && r < m) {
      const double yr = r > jj ? acc : 0.0;
      cv[r].wp = tau * yr - al * vjr;
      cv[r].vp = vjr;
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int jj = m - 1;
    dd[n - 1] = S[jj * kTailLd + jj] - 2.0 * cv[jj].vp * cv[jj].wp;
    ee[n - 1] = 0.0;
    tt[n - 1] = 0.0;
  }
  for (int i = tid; i < lda; i += 512) Vh[(int64_t)(n - 1) * lda + i] = 0.0;
}

// The same tail with the trailing matrix in REGISTERS: thread (rg, cg) = (tid / 16, tid % 16) owns the rows rg + 32 a
// (a < 4) and the columns cg + 16 q (q < 8): for a fixed slot the 16 lanes of a row group read 16 consecutive column
// records (no bank conflicts), and the row sums fold over those 16 adjacent lanes by DPP.  The LDS version reads, per
// column and thread, 32 column records and its 32 matrix elements back and forth (1.5 KB); here a thread reads 4 row
// records and 8 column records and the matrix stays put.  Blocks of 32 rows / 16 columns that lie at or before the
// current column are skipped (uniform branches).  Row jj + 1 -- needed in full by the next column's prologue -- is
// exported to LDS by its owners in the pass that brings it up to date.
__global__ void __launch_bounds__(512) trd_tail_reg_kernel(const TrdDesc* __restrict__ desc, TrdWork w) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  const int J = max(n - kTail, 0), m = n - J;
  const int lda = w.lda;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const double* A = w.A + b * w.n_max * lda;
  double* Vh = w.Vh + b * w.n_max * lda;
  const double* ybuf = w.y + b * 2 * lda;
  double* dd = w.d + b * w.n_max;
  double* ee = w.e + b * w.n_max;
  double* tt = w.tau + b * w.n_max;
  __shared__ RowVec cv[kTail];      // {v', w', v_j} per row / column of the trailing matrix
  __shared__ double xrow[kTail];    // row jj of the matrix, current through step jj - 2
  __shared__ double yv[kTail];      // y = S v_j of the pass, for the prologue wave
  __shared__ double red_s[8];

  // the reflectors of the tail are zero outside rows J .. n - 1: written once, the loop stores the live part only
  for (int64_t e = tid; e < (int64_t)m * lda; e += 512) {
    const int i = (int)(e % lda);
    if (i < J || i >= n) Vh[(int64_t)J * lda + e] = 0.0;
  }
  // pending update of the last column of the resident part
  if (J >= 1) {
    const double* yprev = ybuf + ((J - 1) & 1) * lda;
    const double* vprev = Vh + (int64_t)(J - 1) * lda;
    const double taup = tt[J - 1];
    double dot = 0.0, vv = 0.0, y0 = 0.0;
    if (tid < m) {
      vv = vprev[J + tid];
      y0 = yprev[J + tid];
      dot = y0 * vv;
    }
    dot = wave_sum(dot);
    if (lane == 0) red_s[wave] = dot;
    __syncthreads();
    dot = 0.0;
#pragma unroll
    for (int x = 0; x < 8; ++x) dot += red_s[x];
    const double al = 0.5 * taup * taup * dot;
    if (tid < kTail) {
      cv[tid].vp = tid < m ? vv : 0.0;
      cv[tid].wp = tid < m ? taup * y0 - al * vv : 0.0;
    }
  } else if (tid < kTail) {
    cv[tid].vp = 0.0;
    cv[tid].wp = 0.0;
  }
  __syncthreads();
  const int rg = tid >> 4, cg = tid & 15;
  double s[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int r = rg + 32 * a, gr = J + r;
    const double vpr = cv[r].vp, wpr = cv[r].wp;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = cg + 16 * q, gc = J + c;
      double v = 0.0;
      if (r < m && c < m) {
        const bool mirror = w.tail_lower && (gr >> 5) < (gc >> 5);  // half storage: blocks above the diagonal are not stored
        const double av = mirror ? A[(int64_t)gc * lda + gr] : A[(int64_t)gr * lda + gc];
        v = av - (vpr * cv[c].wp + wpr * cv[c].vp);
      }
      s[a][q] = v;
    }
  }
  if (rg == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) xrow[cg + 16 * q] = s[0][q];
  }
  __syncthreads();
  if (tid < kTail) {  // the update is applied: nothing pending
    cv[tid].vp = 0.0;
    cv[tid].wp = 0.0;
    cv[tid].vj = 0.0;
  }
  __syncthreads();

  // Wave 0 is the PROLOGUE wave: lane l keeps the entries l and l + 64 of the pending update (v', w') in registers and
  // does, between the passes over the matrix, everything that is O(m): w of the finished step, the next row with the
  // update applied, sigma, the Householder scalars, the reflector -- two wave-wide sums by DPP, no barrier in
  // between.  Two barriers per column (the LDS version: four, with two block reductions through LDS).
  const int i0 = lane, i1 = lane + 64;
  double pv0 = 0.0, pv1 = 0.0, pw0 = 0.0, pw1 = 0.0;  // pending v', w' at i0, i1 (wave 0)
  double tau_now = 0.0;                                // tau of the step whose pass is running (wave 0)
  auto pick = [&](double lo, double hi, int idx) {    // entry idx of a vector held as (lane, lane + 64): idx is uniform
    const double src = idx < 64 ? lo : hi;
    const long long bits = __double_as_longlong(src);
    const int l = __builtin_amdgcn_readlane((int)bits, idx & 63), h = __builtin_amdgcn_readlane((int)(bits >> 32), idx & 63);
    return __longlong_as_double(((long long)h << 32) | (unsigned)l);
  };
  for (int jj = 0; jj + 1 < m; ++jj) {
    const int j = J + jj;
    if (wave == 0) {
      // ---- row jj with the pending update -> x, sigma, alpha, d_j; reflector
      const double vpj = pick(pv0, pv1, jj), wpj = pick(pw0, pw1, jj);
      double x0 = 0.0, x1 = 0.0;
      if (i0 < m) x0 = xrow[i0] - (vpj * pw0 + wpj * pv0);
      if (i1 < m) x1 = xrow[i1] - (vpj * pw1 + wpj * pv1);
      double sigma = (i0 > jj + 1 ? x0 * x0 : 0.0) + (i1 > jj + 1 ? x1 * x1 : 0.0);
      sigma = wave_sum(sigma);
      const double alpha = pick(x0, x1, jj + 1), djj = pick(x0, x1, jj);
      double beta, tau, scale;
      householder(alpha, sigma, beta, tau, scale);
      const double v0 = i0 > jj + 1 ? x0 * scale : (i0 == jj + 1 ? 1.0 : 0.0);
      const double v1 = i1 > jj + 1 ? x1 * scale : (i1 == jj + 1 ? 1.0 : 0.0);
      if (i0 < m) {
        cv[i0].vj = v0;
        Vh[(int64_t)j * lda + J + i0] = v0;
      }
      if (i1 < m) {
        cv[i1].vj = v1;
        Vh[(int64_t)j * lda + J + i1] = v1;
      }
      if (lane == 0) {
        dd[j] = djj;
        ee[j] = beta;
        tt[j] = tau;
      }
      tau_now = tau;
      pv0 = v0;  // v' of the step that is pending from now on; w' follows behind the pass
      pv1 = v1;
    }
    __syncthreads();
    // ---- one pass over the registers: pending update (step jj - 1), y = S v_j, row jj + 1 exported
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    {
      const int q0 = (jj + 1) >> 4, a0 = (jj + 1) >> 5;  // first column / row block with an index beyond jj
      double vpr[4], wpr[4];
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4) {
        vpr[a4] = cv[rg + 32 * a4].vp;
        wpr[a4] = cv[rg + 32 * a4].wp;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {  // four column records at a time: the kernel stays within 128 registers
        double vpc[4], wpc[4], vjc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const RowVec cr = cv[cg + 16 * (4 * h + q)];
          vpc[q] = cr.vp;
          wpc[q] = cr.wp;
          vjc[q] = cr.vj;
        }
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          if (a4 >= a0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if (4 * h + q >= q0) {
                s[a4][4 * h + q] = fma(-vpr[a4], wpc[q], fma(-wpr[a4], vpc[q], s[a4][4 * h + q]));
                acc[a4] = fma(s[a4][4 * h + q], vjc[q], acc[a4]);
              }
            }
          }
        }
      }
      if (rg == ((jj + 1) & 31)) {  // one unrolled copy per row slot: no run-time register index
#define NDMPS_TAIL_PUT(AX) _Pragma("unroll") for (int q = 0; q < 8; ++q) xrow[cg + 16 * q] = s[AX][q];
        switch ((jj + 1) >> 5) {
          case 0: NDMPS_TAIL_PUT(0) break;
          case 1: NDMPS_TAIL_PUT(1) break;
          case 2: NDMPS_TAIL_PUT(2) break;
          default: NDMPS_TAIL_PUT(3) break;
        }
#undef NDMPS_TAIL_PUT
      }
    }
    const double y0 = ndmps_lanes::sum_adjacent<16>(acc[0]), y1 = ndmps_lanes::sum_adjacent<16>(acc[1]);
    const double y2 = ndmps_lanes::sum_adjacent<16>(acc[2]), y3 = ndmps_lanes::sum_adjacent<16>(acc[3]);
    if (cg < 4) {  // lanes cg = 0 .. 3 of a row group publish rows rg + 32 cg
      const int rr = rg + 32 * cg;
      const double yr = (cg & 2) ? ((cg & 1) ? y3 : y2) : ((cg & 1) ? y1 : y0);
      yv[rr] = rr > jj && rr < m ? yr : 0.0;
    }
    __syncthreads();
    if (wave == 0) {
      // ---- w of this step becomes the pending update (cv.vp / cv.wp: read by everybody behind the next barrier)
      const double tau = tau_now;
      const double ya = yv[i0], yb = yv[i1];
      double dot = wave_sum(fma(ya, pv0, yb * pv1));
      const double al = 0.5 * tau * tau * dot;
      pw0 = tau * ya - al * pv0;
      pw1 = tau * yb - al * pv1;
      cv[i0].vp = pv0;
      cv[i0].wp = pw0;
      cv[i1].vp = pv1;
      cv[i1].wp = pw1;
    }
  }
  if (tid == 0) {
    const int jj = m - 1;
    dd[n - 1] = xrow[jj] - 2.0 * cv[jj].vp * cv[jj].wp;
    ee[n - 1] = 0.0;
    tt[n - 1] = 0.0;
  }
}

// ----------------------------------------------------------------------------------------- bisection
// # eigenvalues of the scaled T below x = sign changes of p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}.
// No division in the dependent chain; the pair is rescaled by a power of two every fourth step (|d - x| <= 2,
// e^2 <= 1 after scaling by the Gershgorin bound, so four steps stay far inside the fp64 range).
// An exact zero p_i needs no special case as long as e_i^2 > 0: p_{i+1} = -e_i^2 p_{i-1} then has the sign
// opposite to p_{i-1}, one change whatever sign the zero is given.  e^2 is therefore kept >= 2^-200 (a
// perturbation of 2^-100 |T| of a sub-diagonal entry; exactly decoupled blocks would otherwise let the
// sequence die at zero).  de2[i] = (d_i, e_{i-1}^2) in LDS, padded to a multiple of 16 with (4, 2^-200): a
// padding step multiplies p by 4 - x > 0 and changes no sign.  Sixteen records are fetched ahead of the
// chain that consumes them (the LDS latency is off the critical path).
constexpr double kE2Floor = 0x1p-200;
__device__ __forceinline__ int sturm_count(const double2* __restrict__ de2, int n16, double x) {
  double pm = 0.0, pc = 1.0;  // p_{-2}, p_{-1}
  int hp = 0;                 // high word (sign) of p_{-1}
  unsigned cnt = 0;
  for (int i0 = 0; i0 < n16; i0 += 16) {
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = de2[i0 + u];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const double pn = fma(v[u].x - x, pc, -v[u].y * pm);
      const int hn = __double2hiint(pn);
      cnt += (unsigned)(hn ^ hp) >> 31;
      hp = hn;
      pm = pc;
      pc = pn;
      if ((u & 3) == 3) {
        const double mx = fmax(fabs(pm), fabs(pc));
        const int ex = mx > 0.0 ? ilogb(mx) : 0;
        pm = ldexp(pm, -ex);
        pc = ldexp(pc, -ex);
      }
    }
  }
  return (int)cnt;
}

// grid (ceil(k_max / 4), B): one WAVE per wanted eigenvalue (the k_max largest), 65-section: every lane counts
// at one interior point, a ballot tells how many points lie below the eigenvalue.  Ten rounds shrink the
// Gershgorin interval by 65^10 = 1.3e18.  Eigenvalues beyond k_max are not computed: w is zero there.
__global__ void __launch_bounds__(256) trd_bisect_kernel(const TrdDesc* __restrict__ desc, TrdWork w, int k_max) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const int kk = min(k_max, n);
  if ((int)blockIdx.x * 4 >= kk) return;
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
  if (lane == 0) red[wave] = g;
  __syncthreads();
  const double bound = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  const double inv = bound > 0.0 ? 1.0 / bound : 0.0;
  const int n16 = (n + 15) & ~15;
  for (int i = tid; i < n16; i += 256) {
    const double es = (i > 0 && i < n) ? ee[i - 1] * inv : 0.0;
    de2[i] = i < n ? make_double2(dd[i] * inv, i > 0 ? fmax(es * es, kE2Floor) : 0.0) : make_double2(4.0, kE2Floor);
  }
  __syncthreads();
  const int md = blockIdx.x * 4 + wave;  // descending index of this wave's eigenvalue
  if (md < kk) {
    const int m = n - 1 - md;            // ascending index
    double lo = -1.001, hi = 1.001;
    for (int round = 0; round < 10; ++round) {
      const double h = (hi - lo) * (1.0 / 65.0);
      const int cnt = sturm_count(de2, n16, lo + h * (lane + 1));
      // points whose count is <= m lie at or below the eigenvalue
      const int below = __popcll(__ballot(cnt <= m));
      lo = lo + h * below;
      hi = lo + h;
    }
    if (lane == 0) {
      const double lam = 0.5 * (lo + hi);
      w.lam[b * w.n_max + md] = lam;
      d.w_out[md] = lam * bound;
    }
  }
  if (blockIdx.x == 0) {
    for (int i = kk + tid; i < n; i += 256) d.w_out[i] = 0.0;
    if (tid == 0) w.bound[b] = bound;
  }
}

// Shifts of the inverse iteration: the eigenvalues of T / bound, descending, with values closer than 10 eps spread
// 10 eps apart (LAPACK dstein's rule).  Several columns iterated with the SAME shift converge to the same vector when
// T has structure below the resolution of its eigenvalues -- the Gram matrix of an isometric core is the identity plus
// rounding noise, its 512 eigenvalues are a handful of distinct doubles, and 512 columns collapsed onto a few (the
// Cholesky-QR of the block then breaks down).  Shifts a few ulp outside such a cluster return the random starts
// themselves, which is what an invariant subspace without preferred directions should get.  One thread per matrix.
__global__ void trd_shift_kernel(const TrdDesc* __restrict__ desc, TrdWork w, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int k = min(desc[b].k, desc[b].n);
  const double* lam = w.lam + (int64_t)b * w.n_max;
  double* mu = w.mu + (int64_t)b * w.n_max;
  constexpr double kPertol = 10.0 * 2.220446049250313e-16;  // T is scaled to |T| <= 1
  double prev = 0.0;
  for (int c = 0; c < k; ++c) {
    const double m = c == 0 ? lam[0] : fmin(lam[c], prev - kPertol);
    mu[c] = m;
    prev = m;
  }
}

// ------------------------------------------------------------------------- inverse iteration + CholQR
__device__ __forceinline__ double hash_uniform(unsigned a, unsigned b) {
  unsigned x = a * 0x9E3779B1u + b * 0x85EBCA77u + 0x165667B1u;
  x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12; x *= 0x297A2D39u; x ^= x >> 15;
  return (double)x * (2.0 / 4294967296.0) - 1.0 + 1.1e-10;  // never exactly zero
}

// ---- 16 x 16 tile products on the f64 MFMA with both operands row-major in LDS (ld doubles per row):
//   nt: C[i][j] += sum_k X[r0 + i][k0 + k] Y[c0 + j][k0 + k]      nn: C[i][j] += sum_k X[r0 + i][k0 + k] Y[k0 + k][c0 + j]
// accumulator element r of a lane = C[(lane >> 4) + 4 r][lane & 15]; `sign` -1 subtracts
__device__ __forceinline__ f64x4 tile_nt(f64x4 acc, const double* X, int r0, const double* Y, int c0, int k0, int ld,
                                         int lane, double sign) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * X[(r0 + li) * ld + k0 + 4 * s4 + lk], Y[(c0 + li) * ld + k0 + 4 * s4 + lk],
                                               acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f64x4 tile_nn(f64x4 acc, const double* X, int r0, const double* Y, int c0, int k0, int ld,
                                         int lane, double sign) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(sign * X[(r0 + li) * ld + k0 + 4 * s4 + lk], Y[(k0 + 4 * s4 + lk) * ld + c0 + li],
                                               acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ f64x4 tile_load(const double* M, int r0, int c0, int ld, int lane) {
  const int li = lane & 15, lk = lane >> 4;
  f64x4 t;
#pragma unroll
  for (int r = 0; r < 4; ++r) t[r] = M[(r0 + lk + 4 * r) * ld + c0 + li];
  return t;
}
__device__ __forceinline__ void tile_store(double* M, int r0, int c0, int ld, int lane, f64x4 t) {
  const int li = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) M[(r0 + lk + 4 * r) * ld + c0 + li] = t[r];
}

// the value lane `l` (wave-uniform, here always a constant of an unrolled loop) holds: two v_readlane_b32 into scalar
// registers instead of the two ds_bpermute_b32 of __shfl -- the diagonal tiles below broadcast 270 doubles each, one
// after the other (45 us of a 64 x 64 factorisation were those LDS round trips)
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Blocked Cholesky S = L L^T and L^-1 for k16 <= 64 in LDS (S, then L, in the lower triangle of Ls; L^-1 in Li,
// both [k16][ld]); 16 x 16 tiles, diagonal tiles by one wave in registers (lane = row, shuffles broadcast the
// pivots), panels / trailing updates / inverse blocks on the f64 MFMA, three barriers per tile column.  Eight
// waves.  kk: true order (rows and columns >= kk are padding: treated as identity).  Returns false on breakdown.
__device__ bool chol_inv_blocked(double* Ls, double* Li, int k16, int kk, int ld, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int nt = k16 / 16;
  __shared__ int bad;
  if (tid == 0) bad = 0;
  for (int e = tid; e < k16 * ld; e += 512) Li[e] = 0.0;
  // padding: unit diagonal, zero elsewhere (the Gram of the zero pad columns is zero)
  for (int e = tid; e < k16; e += 512)
    if (e >= kk) Ls[e * ld + e] = 1.0;
  __syncthreads();
  for (int J = 0; J < nt; ++J) {
    const int j0 = 16 * J;
    if (wave == 0) {
      // ---- diagonal tile: lane i < 16 holds row i of S_JJ
      double row[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) row[c] = lane < 16 ? Ls[(j0 + lane) * ld + j0 + c] : 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double piv = readlane_f64(row[c], c);
        double rs = 1.0;
        if (piv > 0.0) rs = fast_rsqrt<2>(piv);
        else if (lane == 0) bad = 1;
        const double lc = lane == c ? piv * rs : (lane > c ? row[c] * rs : 0.0);
        row[c] = lc;
#pragma unroll
        for (int cc = c + 1; cc < 16; ++cc) {
          const double lcc = readlane_f64(lc, cc);
          row[cc] = fma(-lc, lcc, row[cc]);
        }
      }
      if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c) Ls[(j0 + lane) * ld + j0 + c] = row[c];  // lower triangle, zeros above
      }
      // ---- its inverse: lane c < 16 makes column c by forward substitution (rows of L broadcast by shuffle)
      double x[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        double sum = lane == i ? 1.0 : 0.0;
#pragma unroll
        for (int pz = 0; pz < i; ++pz) {
          const double lip = readlane_f64(row[pz], i);  // L[i][pz]
          sum = fma(-lip, x[pz], sum);
        }
        const double lii = readlane_f64(row[i], i);
        x[i] = lane <= i ? sum * fast_rcp<2>(lii) : 0.0;
      }
      if (lane < 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) Li[(j0 + i) * ld + j0 + lane] = x[i];
      }
    }
    __syncthreads();
    // ---- panel: L_IJ = S_IJ Linv_JJ^T  for tiles I > J, one wave each
    for (int I = J + 1 + wave; I < nt; I += 8) {
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      acc = tile_nt(acc, Ls, 16 * I, Li, j0, j0, ld, lane, 1.0);
      tile_store(Ls, 16 * I, j0, ld, lane, acc);  // the wave has read its whole tile (operands) before this store
    }
    __syncthreads();
    // ---- trailing update: S_IK -= L_IJ L_KJ^T  for J < K <= I
    {
      int t = 0;
      for (int I = J + 1; I < nt; ++I)
        for (int K = J + 1; K <= I; ++K, ++t)
          if (t % 8 == wave) {
            f64x4 acc = tile_load(Ls, 16 * I, 16 * K, ld, lane);
            acc = tile_nt(acc, Ls, 16 * I, Ls, 16 * K, j0, ld, lane, -1.0);
            tile_store(Ls, 16 * I, 16 * K, ld, lane, acc);
          }
    }
    __syncthreads();
  }
  // ---- off-diagonal blocks of L^-1, block row by block row: Linv_IJ = -Linv_II (sum_{J <= K < I} L_IK Linv_KJ)
  for (int I = 1; I < nt; ++I) {
    f64x4 tmp[1];
    const int J = wave;  // at most 3 block columns: one wave each
    if (J < I) {
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      for (int K = J; K < I; ++K) acc = tile_nn(acc, Ls, 16 * I, Li, 16 * J, 16 * K, ld, lane, 1.0);
      tmp[0] = acc;
      // through LDS to become an operand: the tile (I, J) of Li is still zero and unused by the other waves
      tile_store(Li, 16 * I, 16 * J, ld, lane, acc);
    }
    __syncthreads();
    if (J < I) {
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      // Linv_II (rows 16 I.., cols 16 I..) times the tile just stored at (I, J): nn with X = Li diag, Y = Li
      acc = tile_nn(acc, Li, 16 * I, Li, 16 * J, 16 * I, ld, lane, -1.0);
      tmp[0] = acc;
    }
    __syncthreads();
    if (J < I) tile_store(Li, 16 * I, 16 * J, ld, lane, tmp[0]);
    __syncthreads();
  }
  return bad == 0;
}

// rows per block of the sweeps' LDS ring: SB kb <= 2048 doubles per operand array (six arrays + two out-buffers: 128 KB)
__host__ __device__ constexpr int invit_rows_per_block(int kb) { return kb <= 16 ? 128 : kb <= 32 ? 64 : kb <= 64 ? 32 : 16; }

// One workgroup (512 threads) per matrix and column block.  Threads c < k factor T - lam_c I (pivoted, dgttrf order) and solve
// twice from a random start; no orthogonalisation in between (columns of a numerically multiple eigenvalue
// stay independent because their starts are), then the whole workgroup orthonormalises the block twice:
// S = Z^T Z (f64 MFMA from global), Cholesky in LDS, Z <- Z L^-T row by row.  Measured on the CPU prototype
// (tools/scratch/tridiag_proto.py): two solves + two Cholesky-QR passes give 2e-15 orthogonality and 3e-16
// residuals on the volume Gram matrices and 5e-15 on exact 10- / 20-fold eigenvalues; a third solve makes
// the columns of a multiple eigenvalue more parallel and costs a digit.
// The recurrences are sequential in i and run on k <= 128 lanes: their operands are fetched one chunk of
// CH steps ahead of the chain that consumes them (a global access costs more than a chunk of the chain).
__global__ void __launch_bounds__(512) trd_invit_kernel(TrdDesc* __restrict__ desc, TrdWork w, int cb_arg) {
  const int cb = cb_arg & 0xffff, dbg = cb_arg >> 16;  // dbg: ablation switches of tools/scratch (0 in the product)
  TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, k = d.k, kp = w.kp;
  const int tid = threadIdx.x;
  // column block of this workgroup: columns [cb x, cb x + cb) of the kp columns (cb = 128: one block for up to 128 wanted
  // vectors; narrower blocks deal the columns of one or a few big matrices to several CUs, invit_block_width); kb: width
  // of the block's tiles in LDS, kl: wanted columns inside it
  const int cbase = (int)blockIdx.x * cb;
  const int kb = min(kp - cbase, cb), kl = max(min(k - cbase, cb), 0);
  if (kb <= 0) return;
  const int64_t b = blockIdx.y;
  const double* dd = w.d + b * w.n_max;
  const double* ee = w.e + b * w.n_max;
  double* Z = w.Z + b * w.n_max * kp;
  const int64_t plane = (int64_t)w.n_max * kp;
  double* DL = w.lu + b * 4 * plane;
  double* DI = DL + plane;
  double* DU = DI + plane;
  double* DU2 = DU + plane;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const double bound = w.bound[b];
  const double inv = bound > 0.0 ? 1.0 / bound : 0.0;
  constexpr double kEps = 2.220446049250313e-16;
  constexpr int CH = 8;

  // scaled T, te[i] = (d_i, e_i) / bound, in the LDS the Cholesky factor will use later: only the
  // factorisation reads it, and that is over before the first S is stored
  double2* te = reinterpret_cast<double2*>(lds);
  long long* stamp = w.stamps + b * 16;
  const bool stamping = blockIdx.x == 0;
  if (tid == 0 && stamping) stamp[0] = wall_clock64();
  for (int i = tid; i < n; i += 512) te[i] = make_double2(dd[i] * inv, i + 1 < n ? ee[i] * inv : 0.0);
  __syncthreads();
  if (tid < kl) {
    const int c = cbase + tid;
    const double mu = w.mu[b * w.n_max + c];
    // ---- factorisation (scaled T: |T| <= 1) fused with the first forward sweep on the random start
    double di = te[0].x - mu;             // current diagonal
    double ui = te[0].y;                  // current super-diagonal du[i]
    double xi = hash_uniform(0u, (unsigned)c);
    for (int i0 = 0; i0 + 1 < n; i0 += CH) {
      double2 t0[CH + 1];
#pragma unroll
      for (int u = 0; u <= CH; ++u) t0[u] = te[min(i0 + u + 1, n - 1)];
      double li = te[i0].y;               // sub-diagonal below row i0 (= e_i0)
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int i = i0 + u;
        if (i + 1 < n) {
          const double dn = t0[u].x - mu;                    // d[i + 1] before elimination
          const double un = i + 2 < n ? t0[u].y : 0.0;       // du[i + 1] before elimination
          const bool swap = fabs(di) < fabs(li);
          double piv = swap ? li : di;
          if (piv == 0.0) piv = kEps;
          const double rp = fast_rcp<2>(piv);
          const double fact = (swap ? di : li) * rp;
          const double du_i = swap ? dn : ui;
          const double up = swap ? ui : dn;
          const int64_t o = (int64_t)i * kp + c;
          // stored as the sweeps use them (their helpers load and pass on, no arithmetic between a load and LDS): the
          // division of the back substitution folded into du, du2 and z; the row interchange rides in the lowest
          // mantissa bit of 1 / d (a scale factor: one ulp of it is a rounding like any other)
          const double rs = __hiloint2double(__double2hiint(rp), (__double2loint(rp) & ~1) | (swap ? 1 : 0));
          DL[o] = fact;
          DI[o] = rs;
          DU[o] = du_i * rs;
          DU2[o] = swap ? un * rs : 0.0;
          di = fma(-fact, du_i, up);
          ui = swap ? -fact * un : un;
          li = t0[u].y;                                      // e_{i+1}: sub-diagonal below row i + 1
          const double xn = hash_uniform((unsigned)(i + 1), (unsigned)c);
          const double top = swap ? xn : xi;
          const double bot = swap ? xi : xn;
          Z[o] = top * rs;
          xi = fma(-fact, top, bot);
        }
      }
    }
    if (di == 0.0) di = kEps;
    DI[(int64_t)(n - 1) * kp + c] = fast_rcp<2>(di);
    Z[(int64_t)(n - 1) * kp + c] = xi;  // last component of the fused forward sweep
    if (tid == 0 && stamping) stamp[1] = wall_clock64();
  }
  __syncthreads();  // the factors and the first forward sweep are visible to the whole workgroup

  // ---- the remaining sweeps: back substitution, forward sweep, back substitution.  Each is a recurrence over
  // the rows on k lanes.  Measured on the first version (one wave doing everything: 160 ns per row) the
  // recurrence itself was half of the time; the rest was that wave fetching and parking operands and storing its
  // results row by row.  Now the lanes that run the recurrence touch LDS only: threads 128 .. 511 stream the
  // operands of a block of SB rows from global memory into a double-buffered LDS ring two blocks ahead (three
  // values per row and lane, the division of the back substitution folded into them), and carry the results of
  // the block before from an LDS out-buffer back to Z with coalesced stores.
  // The direction is a compile-time constant of the sweep's body (two copies), and a group of CH rows that lies inside
  // the matrix runs without its range selects: with `backward` and the range tested per row the ISA of a row was 40
  // instructions and two branches around its two FMAs (100 ns per row; now the chain and its LDS traffic).
  {
    const int SB = invit_rows_per_block(kb);
    const int per_arr = SB * kb;               // doubles per operand array and block (<= 2048)
    constexpr int HN = 384, EPT = 6;           // helper threads, elements per helper and array (EPT HN >= per_arr)
    double* ring = lds;                         // [2][3][per_arr]
    double* outb = lds + 6 * per_arr;           // [2][per_arr]
    const bool solver = tid < kl;
    const bool helper = tid >= 128;
    const int ht = tid - 128;
    const int c = cbase + tid;
    double xi = solver ? Z[(int64_t)(n - 1) * kp + c] : 0.0;  // last component of the fused first forward sweep
    const int rows = n - 1;                     // both recurrences visit rows 0 .. n-2
    const int nblk = (rows + SB - 1) / SB;
    auto sweep_body = [&](auto back_tag) {
      constexpr bool backward = decltype(back_tag)::value;
      double st0[3][EPT], st1[3][EPT];            // two register stages x 3 arrays
      // row visited at position s of block blk (may be out of range: < 0 or > n-2)
      auto row_of = [&](int blk, int s2) { return backward ? (n - 2) - blk * SB - s2 : blk * SB + s2; };
      // three loads per element from an address inside the arrays whatever the element (rows beyond the matrix and
      // blocks beyond the last look at row 0: what they fetch is never used -- the chain ignores positions >= valid,
      // write_back skips them, a block >= nblk is never committed).  No branch and no arithmetic between the loads and
      // the stage registers: all eighteen are in flight until the next iteration parks them.  (With z / d formed here the
      // four loads of an element were waited for one element after the other: six memory round trips per block and
      // helper, 60 of the 146 us of the three sweeps at order 512.)
      auto fetch = [&](int blk, double (&dst)[3][EPT]) {
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          const int e = ht + HN * u;
          const int s2 = e / kb, cc = e % kb;
          const int r = min(max(row_of(blk, s2), 0), n - 2);
          const int64_t o = (int64_t)((dbg & 2) ? 0 : r) * kp + cbase + cc;
          if (backward) {  // x_r = z_r / d_r - (du_r / d_r) x_{r+1} - (du2_r / d_r) x_{r+2}
            dst[0][u] = Z[o];
            dst[1][u] = DU[o];
            dst[2][u] = DU2[o];
          } else {
            dst[0][u] = Z[o + kp];
            dst[1][u] = DL[o];
            dst[2][u] = DI[o];  // 1 / d_r, lowest mantissa bit set where rows r and r + 1 were interchanged
          }
        }
      };
      auto commit = [&](int blk, const double (&src)[3][EPT]) {
        double* buf = ring + (blk & 1) * 3 * per_arr;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          const int e = ht + HN * u;
          if (e < per_arr) {
#pragma unroll
            for (int a2 = 0; a2 < 3; ++a2) buf[a2 * per_arr + e] = src[a2][u];
          }
        }
      };
      auto write_back = [&](int blk) {  // results of block blk: LDS -> Z
        const double* ob = outb + (blk & 1) * per_arr;
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
          const int e = ht + HN * u;
          const int s2 = e / kb, cc = e % kb;
          const int r = row_of(blk, s2);
          if (e < per_arr && cc < kl && r >= 0 && r <= n - 2 && !(dbg & 4)) Z[(int64_t)r * kp + cbase + cc] = ob[e];
        }
      };
      double x1 = 0.0, x2 = 0.0;
      if (backward) {
        if (solver) {
          x1 = xi * DI[(int64_t)(n - 1) * kp + c];
          Z[(int64_t)(n - 1) * kp + c] = x1;
        }
      } else if (solver) {
        xi = Z[c];
      }
      __syncthreads();  // the previous sweep's stores are visible; the ring is free
      if (helper) {
        fetch(0, st0);
        fetch(1, st1);
        commit(0, st0);
      }
      __syncthreads();
      // The stage holding block blk + 1 (fetched two iterations ago) is parked in the other LDS buffer, then
      // reused for block blk + 2.  The stages are named, not indexed: a run-time index would put them in scratch.
      auto iteration = [&](int blk, double (&parked)[3][EPT], double (&refill)[3][EPT]) {
        if (helper) {
          if (blk + 1 < nblk) commit(blk + 1, parked);
          fetch(blk + 2, refill);
          if (blk >= 1) write_back(blk - 1);
        } else if (solver && !(dbg & 1)) {
          const double* buf = ring + (blk & 1) * 3 * per_arr + tid;
          double* ob = outb + (blk & 1) * per_arr + tid;
          const int valid = min(SB, rows - blk * SB);  // positions 0 .. valid - 1 of the block are rows of the matrix
          for (int s0 = 0; s0 < SB; s0 += CH) {  // eight rows' operands out of LDS, then eight steps of the chain
            double o0[CH], o1[CH], o2[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
              const int e = (s0 + u) * kb;
              o0[u] = buf[e];
              o1[u] = buf[per_arr + e];
              o2[u] = buf[2 * per_arr + e];
            }
            if (s0 + CH <= valid) {  // the whole group inside the matrix (wave-uniform): no selects on the chain
#pragma unroll
              for (int u = 0; u < CH; ++u) {
                if (backward) {
                  const double x0 = fma(-o1[u], x1, fma(-o2[u], x2, o0[u]));
                  ob[(s0 + u) * kb] = x0;
                  x2 = x1;
                  x1 = x0;
                } else {
                  const bool sw = (__double2loint(o2[u]) & 1) != 0;
                  const double top = sw ? o0[u] : xi, bot = sw ? xi : o0[u];
                  ob[(s0 + u) * kb] = top * o2[u];  // z_r / d_r, what the back substitution reads
                  xi = fma(-o1[u], top, bot);
                }
              }
            } else {
#pragma unroll
              for (int u = 0; u < CH; ++u) {
                const bool in = s0 + u < valid;
                if (backward) {
                  const double x0 = fma(-o1[u], x1, fma(-o2[u], x2, o0[u]));
                  ob[(s0 + u) * kb] = x0;
                  if (in) {
                    x2 = x1;
                    x1 = x0;
                  }
                } else {
                  const bool sw = (__double2loint(o2[u]) & 1) != 0;
                  const double top = sw ? o0[u] : xi, bot = sw ? xi : o0[u];
                  ob[(s0 + u) * kb] = top * o2[u];
                  if (in) xi = fma(-o1[u], top, bot);
                }
              }
            }
          }
        }
        __syncthreads();
      };
      for (int blk = 0; blk < nblk; blk += 2) {
        iteration(blk, st1, st0);
        if (blk + 1 < nblk) iteration(blk + 1, st0, st1);
      }
      if (helper) write_back(nblk - 1);
      __syncthreads();  // the sweep's results are in Z before the next one (or the orthonormalisation) reads them
    };
    sweep_body(std::true_type{});
    sweep_body(std::false_type{});
    sweep_body(std::true_type{});
  }
  // pad columns of the block stay zero
  if (kl < kb)
    for (int e = tid; e < n * (kb - kl); e += 512) Z[(int64_t)(e / (kb - kl)) * kp + cbase + kl + e % (kb - kl)] = 0.0;
  if (tid == 0 && stamping) stamp[2] = wall_clock64();
}

// Orthonormalisation of the n x k block Z (its own kernel: the recurrences above and the products below have
// very different register needs, one kernel for both spilled 119 VGPRs).  One workgroup (512 threads) per matrix.
// FAST: k <= 64 (blocked Cholesky, explicit L^-1, products on the MFMA); otherwise the column-by-column
// Cholesky and a row-by-row triangular solve (any k <= 128; only BASELINE config 5 gets there).
template <bool FAST>
__global__ void __launch_bounds__(512) trd_ortho_kernel(TrdDesc* __restrict__ desc, TrdWork w) {
  TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, k = d.k, kp = w.kp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  double* Z = w.Z + b * w.n_max * kp;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Ls = lds;  // [k16][k16 + 1] Cholesky factor
  const int k16 = (k + 15) & ~15, ldl = k16 + 1;
  long long* stamp = w.stamps + b * 16;

  for (int pass = 0; pass < 2; ++pass) {
    // ---- S = Z^T Z, 16 x 16 tiles (ta <= tb) on f64 MFMA, operands straight from global / L2,
    //      eight k-steps of operands requested before the MFMAs that consume them
    const int nt = k16 / 16, ntiles = nt * (nt + 1) / 2;
    for (int tile = wave; tile < ntiles; tile += 8) {
      int ta = 0, u0 = tile;
      while (u0 >= nt - ta) {
        u0 -= nt - ta;
        ++ta;
      }
      const int tb = ta + u0;
      const int li = lane & 15, lk = lane >> 4;
      f64x4 acc = {0.0, 0.0, 0.0, 0.0};
      const double* za = Z + ta * 16 + li;
      const double* zb = Z + tb * 16 + li;
      for (int i0 = 0; i0 < n; i0 += 32) {
        double av[8], bv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 4 * u + lk;
          const bool ok = i < n;
          const int64_t o = (int64_t)(ok ? i : 0) * kp;
          av[u] = ok ? za[o] : 0.0;
          bv[u] = ok ? zb[o] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ra = ta * 16 + lk + 4 * r, cb = tb * 16 + li;
        Ls[ra * ldl + cb] = acc[r];
        Ls[cb * ldl + ra] = acc[r];
      }
    }
    __syncthreads();
    if (tid == 0) stamp[3 + 3 * pass] = wall_clock64();
    if constexpr (FAST) {
      // ---- fast path: blocked Cholesky + explicit L^-1 in LDS, then Z <- Z L^-T as a product on the MFMA
      double* Li = lds + k16 * ldl;
      if (!chol_inv_blocked(Ls, Li, k16, k, ldl, tid) && tid == 0 && d.status != 2) d.status = 1;
      if (tid == 0) stamp[4 + 3 * pass] = wall_clock64();
      // one wave per 16-row block of Z: all its operands are read before anything is written back
      for (int rt = wave; rt * 16 < n; rt += 8) {
        const int i0 = 16 * rt, li = lane & 15, lk = lane >> 4;
        double av[16];
        const bool row_ok = i0 + li < n;
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
          const int col = 4 * s4 + lk;
          av[s4] = (row_ok && col < k16) ? Z[(int64_t)(i0 + li) * kp + col] : 0.0;
        }
        for (int ct = 0; ct < k16 / 16; ++ct) {
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          // columns a <= c only: L^-1 is lower triangular
#pragma unroll
          for (int s4 = 0; s4 < 16; ++s4)
            if (4 * s4 < 16 * (ct + 1) && 4 * s4 < k16)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], Li[(16 * ct + li) * ldl + 4 * s4 + lk], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + lk + 4 * r, cc = 16 * ct + li;
            if (i < n && cc < k) Z[(int64_t)i * kp + cc] = acc[r];
          }
        }
      }
      __syncthreads();
      if (tid == 0) stamp[5 + 3 * pass] = wall_clock64();
    } else {
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
          if (tid == jc && d.status != 2) d.status = 1;
        }
      }
      __syncthreads();
    }
    if (tid == 0) stamp[4 + 3 * pass] = wall_clock64();
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
    if (tid == 0) stamp[5 + 3 * pass] = wall_clock64();
    }
  }
}

// k > 64 (BASELINE config 5: k = 128): the same orthonormalisation by COLUMN BLOCKS of 64 -- block classical
// Gram-Schmidt with re-orthogonalisation: for every block, twice: project out the finished columns
// (P = Z_prev^T Z_blk, Z_blk -= Z_prev P, both on the f64 MFMA), then the blocked Cholesky-QR of trd_ortho_kernel<true>
// on the block.  The factors of one block (2 x 64 x 65 doubles) and P (64 x 65) fit the LDS; the whole 128 x 128
// factor and its inverse do not (264 KB), which sent k = 128 to the column-by-column Cholesky (3.1 ms per call).
__global__ void __launch_bounds__(512) trd_ortho_blocks_kernel(TrdDesc* __restrict__ desc, TrdWork w) {
  constexpr int KB = 64, LDL = KB + 1;
  TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, k = d.k, kp = w.kp;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int64_t b = blockIdx.y;
  double* Z = w.Z + b * w.n_max * kp;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* Ls = lds;             // [64][65] Cholesky factor of the block
  double* Li = lds + KB * LDL;  // [64][65] its inverse
  double* Ps = Li + KB * LDL;   // [64][65] one 64-row slice of P
  for (int c0 = 0; c0 < k; c0 += KB) {
    const int kb = min(KB, k - c0), kb16 = (kb + 15) & ~15;
    for (int pass = 0; pass < 2; ++pass) {
      // ---- project out the finished columns, 64 of them at a time
      for (int p0 = 0; p0 < c0; p0 += KB) {
        for (int tile = wave; tile < (KB / 16) * (kb16 / 16); tile += 8) {
          const int ta = tile / (kb16 / 16), tb = tile % (kb16 / 16);
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
          const double* za = Z + p0 + ta * 16 + li;
          const double* zb = Z + c0 + tb * 16 + li;
          const bool col_ok = tb * 16 + li < kb;
          for (int i0 = 0; i0 < n; i0 += 32) {
            double av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int i = i0 + 4 * u + lk;
              const bool ok = i < n;
              const int64_t o = (int64_t)(ok ? i : 0) * kp;
              av[u] = ok ? za[o] : 0.0;
              bv[u] = (ok && col_ok) ? zb[o] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) Ps[(ta * 16 + lk + 4 * r) * LDL + tb * 16 + li] = acc[r];
        }
        __syncthreads();
        for (int rt = wave; rt * 16 < n; rt += 8) {
          const int i0 = 16 * rt;
          const bool row_ok = i0 + li < n;
          double av[16];
#pragma unroll
          for (int s4 = 0; s4 < 16; ++s4) av[s4] = row_ok ? -Z[(int64_t)(i0 + li) * kp + p0 + 4 * s4 + lk] : 0.0;
          for (int ct = 0; ct < kb16 / 16; ++ct) {
            f64x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = i0 + lk + 4 * r, cc = 16 * ct + li;
              acc[r] = (i < n && cc < kb) ? Z[(int64_t)i * kp + c0 + cc] : 0.0;
            }
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], Ps[(4 * s4 + lk) * LDL + 16 * ct + li], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int i = i0 + lk + 4 * r, cc = 16 * ct + li;
              if (i < n && cc < kb) Z[(int64_t)i * kp + c0 + cc] = acc[r];
            }
          }
        }
        __syncthreads();
      }
      // ---- S = Z_blk^T Z_blk
      const int nt = kb16 / 16, ntiles = nt * (nt + 1) / 2;
      for (int tile = wave; tile < ntiles; tile += 8) {
        int ta = 0, u0 = tile;
        while (u0 >= nt - ta) {
          u0 -= nt - ta;
          ++ta;
        }
        const int tb = ta + u0;
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* za = Z + c0 + ta * 16 + li;
        const double* zb = Z + c0 + tb * 16 + li;
        const bool a_ok = ta * 16 + li < kb, b_ok = tb * 16 + li < kb;
        for (int i0 = 0; i0 < n; i0 += 32) {
          double av[8], bv[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + 4 * u + lk;
            const bool ok = i < n;
            const int64_t o = (int64_t)(ok ? i : 0) * kp;
            av[u] = (ok && a_ok) ? za[o] : 0.0;
            bv[u] = (ok && b_ok) ? zb[o] : 0.0;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ra = ta * 16 + lk + 4 * r, cb = tb * 16 + li;
          Ls[ra * LDL + cb] = acc[r];
          Ls[cb * LDL + ra] = acc[r];
        }
      }
      __syncthreads();
      if (!chol_inv_blocked(Ls, Li, kb16, kb, LDL, tid) && tid == 0 && d.status != 2) d.status = 1;
      // ---- Z_blk <- Z_blk L^-T
      for (int rt = wave; rt * 16 < n; rt += 8) {
        const int i0 = 16 * rt;
        const bool row_ok = i0 + li < n;
        double av[16];
#pragma unroll
        for (int s4 = 0; s4 < 16; ++s4) {
          const int col = 4 * s4 + lk;
          av[s4] = (row_ok && col < kb) ? Z[(int64_t)(i0 + li) * kp + c0 + col] : 0.0;
        }
        for (int ct = 0; ct < kb16 / 16; ++ct) {
          f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s4 = 0; s4 < 16; ++s4)
            if (4 * s4 < 16 * (ct + 1) && 4 * s4 < kb16)
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], Li[(16 * ct + li) * LDL + 4 * s4 + lk], acc, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = i0 + lk + 4 * r, cc = 16 * ct + li;
            if (i < n && cc < kb) Z[(int64_t)i * kp + c0 + cc] = acc[r];
          }
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------ back-transform
// V = H_0 H_1 ... H_{n-2} Z on k columns.  A column lives in the registers of SEG lanes (row i in lane
// i mod SEG, slot i / SEG), so v^T z is a shuffle reduction and a reflector costs no barrier.  Reflectors
// come through LDS in blocks of RB rows of Vh, double-buffered: the global loads of block t + 1 are in
// flight while block t is applied.  Inside a block, WYB consecutive reflectors are applied together in compact WY
// form, H_{j} ... H_{j+WYB-1} = I - V T V^T (T from trd_wy_kernel): their WYB dot products and shuffle
// reductions are independent, so the dependent chain per reflector shrinks by ~WYB (one reflector at a time:
// 0.49 us each, the chain dot -> reduce -> update).
// grid (ceil(k / (256 / SEG)), B), 256 threads; LDS 2 * RB * SEG * R doubles.
//
// Reflectors are numbered from the top, u = (n - 2) - j (u = 0 is applied first); WY group g holds u in
// [g WYB, (g + 1) WYB).  Inside a group a = WYB - 1 - (u mod WYB) ascends with j.
// trd_wy_kernel: T (upper triangular, LAPACK dlarft forward / columnwise) of every group:
//   T[a][a] = tau_a,  T[0:a, a] = -tau_a T[0:a, 0:a] (V[:, 0:a]^T v_a).     grid (groups, B), 64 threads.
template <int WYB>
__global__ void __launch_bounds__(64) trd_wy_kernel(const TrdDesc* __restrict__ desc, TrdWork w, double* __restrict__ Tw,
                                                   int64_t t_stride) {
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, lda = w.lda;
  const int jtop = n - 2;
  const int g = blockIdx.x;
  if (g * WYB > jtop) return;
  const int64_t b = blockIdx.y;
  const double* Vh = w.Vh + b * w.n_max * lda;
  const double* tt = w.tau + b * w.n_max;
  const int lane = threadIdx.x;
  __shared__ double G[WYB][WYB];
  __shared__ double T[WYB][WYB];
  // j of ascending index a: j_a = jtop - (g WYB + WYB - 1 - a); j_a < 0: identity (tau = 0, v = 0)
  for (int a = 0; a < WYB; ++a)
    for (int c = a + 1; c < WYB; ++c) {
      const int ja = jtop - (g * WYB + WYB - 1 - a), jc = jtop - (g * WYB + WYB - 1 - c);
      double s = 0.0;
      if (ja >= 0 && jc >= 0)
        for (int i = lane; i < n; i += 64) s = fma(Vh[(int64_t)ja * lda + i], Vh[(int64_t)jc * lda + i], s);
      s = wave_sum(s);
      if (lane == 0) G[a][c] = s;
    }
  __syncthreads();
  if (lane == 0) {
    for (int a = 0; a < WYB; ++a) {
      const int ja = jtop - (g * WYB + WYB - 1 - a);
      const double ta = ja >= 0 ? tt[ja] : 0.0;
      for (int i = 0; i < WYB; ++i) T[i][a] = 0.0;
      T[a][a] = ta;
      for (int i = 0; i < a; ++i) {
        double s = 0.0;
        for (int m = i; m < a; ++m) s = fma(T[i][m], G[m][a], s);
        T[i][a] = -ta * s;
      }
    }
  }
  __syncthreads();
  double* out = Tw + b * t_stride + (int64_t)g * WYB * WYB;
  if (lane < WYB * WYB) out[lane] = T[lane / WYB][lane % WYB];
}

template <int SEG, int R, int RB, int WYB>
__global__ void __launch_bounds__(256) trd_back_kernel(const TrdDesc* __restrict__ desc, TrdWork w, int k_fill,
                                                       const double* __restrict__ Tw, int64_t t_stride) {
  constexpr int NP = SEG * R;             // padded order
  constexpr int PER = RB * NP / 256;      // doubles per thread and block
  constexpr int NG = RB / WYB;            // WY groups per staged block
  static_assert(RB * NP % 256 == 0 && RB % WYB == 0, "block must divide over the workgroup / into WY groups");
  const TrdDesc& d = desc[blockIdx.y];
  const int n = d.n, k = d.k, kp = w.kp, lda = w.lda;
  const int tid = threadIdx.x;
  const int seg = tid % SEG;
  const int c = blockIdx.x * (256 / SEG) + tid / SEG;
  if ((int)blockIdx.x * (256 / SEG) >= k) {
    // columns beyond the rank, up to k_fill, are defined as zero (rank decided on the device: the caller's
    // buffers are sized for the bond cap and the kept columns are a prefix)
    if (c < min(k_fill, n))
      for (int i = seg; i < n; i += SEG) d.V_out[(int64_t)i * n + c] = 0.0;
    return;
  }
  const bool live = c < k;
  const int64_t b = blockIdx.y;
  const double* Z = w.Z + b * w.n_max * kp;
  const double* Vh = w.Vh + b * w.n_max * lda;
  const double* Tb = Tw + b * t_stride;
  extern __shared__ __attribute__((aligned(16))) double lds[];  // [2][RB][NP]
  __shared__ double tls[2][NG][WYB * WYB];
  double x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = seg + SEG * r;
    x[r] = (live && i < n) ? Z[(int64_t)i * kp + c] : 0.0;
  }
  // staged block t holds reflectors u = t RB + qq, qq = 0..RB-1 (j = jtop - u < 0: identity)
  const int jtop = n - 2;
  const int nblocks = jtop >= 0 ? (jtop + RB) / RB : 0;
  double stage[PER];
  double tstage = 0.0;
  auto fetch = [&](int t) {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u, qq = e / NP, i = e % NP;
      const int j = jtop - t * RB - qq;
      stage[u] = (j >= 0 && i < n) ? Vh[(int64_t)j * lda + i] : 0.0;
    }
    if (tid < NG * WYB * WYB) {
      const int g = t * NG + tid / (WYB * WYB);
      tstage = g * WYB <= jtop ? Tb[(int64_t)g * WYB * WYB + tid % (WYB * WYB)] : 0.0;
    }
  };
  auto commit = [&](int t) {
    double* buf = lds + (t & 1) * RB * NP;
#pragma unroll
    for (int u = 0; u < PER; ++u) buf[tid + 256 * u] = stage[u];
    if (tid < NG * WYB * WYB) tls[t & 1][tid / (WYB * WYB)][tid % (WYB * WYB)] = tstage;
  };
  if (nblocks > 0) {
    fetch(0);
    commit(0);
  }
  __syncthreads();
  for (int t = 0; t < nblocks; ++t) {
    if (t + 1 < nblocks) fetch(t + 1);
    const double* buf = lds + (t & 1) * RB * NP;
#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
      // group of WYB reflectors: staged rows gq WYB + q, q = 0..WYB-1; ascending index a = WYB - 1 - q
      double vr[WYB][R], y[WYB];
#pragma unroll
      for (int q = 0; q < WYB; ++q) {
        const double* v = buf + (gq * WYB + q) * NP + seg;
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          vr[q][r] = v[SEG * r];
          s = fma(vr[q][r], x[r], s);
        }
        y[WYB - 1 - q] = s;
      }
#pragma unroll
      for (int a = 0; a < WYB; ++a) y[a] = ndmps_lanes::sum_adjacent<SEG>(y[a]);
      const double* T = tls[t & 1][gq];
      double z[WYB];
#pragma unroll
      for (int a = 0; a < WYB; ++a) {
        double s = 0.0;
#pragma unroll
        for (int bb = a; bb < WYB; ++bb) s = fma(T[a * WYB + bb], y[bb], s);
        z[a] = s;
      }
#pragma unroll
      for (int q = 0; q < WYB; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = fma(-z[WYB - 1 - q], vr[q][r], x[r]);
    }
    if (t + 1 < nblocks) commit(t + 1);  // the other buffer: its readers finished before the last barrier
    __syncthreads();
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
  } else if (c < min(k_fill, n)) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = seg + SEG * r;
      if (i < n) d.V_out[(int64_t)i * n + c] = 0.0;
    }
  }
}

__global__ void trd_status_kernel(const TrdDesc* __restrict__ desc, int batch, int* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) out[b] = desc[b].status;
}

// Rank decision on the device (the rule of kept_rank in tt.hip: singular values s_i = sqrt(max(w_i, 0)); keep
// s_i > cutoff * s_0, at least one, at most k_cap): sets desc.k and reports rank and spectrum.
__global__ void __launch_bounds__(128)
trd_rank_kernel(TrdDesc* __restrict__ desc, int k_cap, double cutoff, int* __restrict__ ranks, double* __restrict__ spectra,
                int64_t spectra_stride) {
  TrdDesc& d = desc[blockIdx.x];
  const int kk = min(k_cap, d.n);
  const double s0 = sqrt(fmax(d.w_out[0], 0.0));
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int mine = 0;
  for (int i = threadIdx.x; i < kk; i += 128) {
    const double si = sqrt(fmax(d.w_out[i], 0.0));
    if (spectra) spectra[(int64_t)blockIdx.x * spectra_stride + i] = si;
    mine += si > cutoff * s0;
  }
  if (mine) atomicAdd(&cnt, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int k = min(max(cnt, 1), kk);
    d.k = k;
    if (d.status != 2) d.status = 0;  // 2: the team kernel gave up (sticky)
    ranks[blockIdx.x] = k;
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
__global__ void trd_setdesc_kernel(TrdDesc* __restrict__ desc, DescChunk chunk, int base, int count,
                                   TeamSync* __restrict__ sync) {
  const int t = threadIdx.x;
  if (t < count) {
    desc[base + t] = chunk.v[t];
    sync[base + t].count = 0;
    sync[base + t].abort = 0;
  }
}
__global__ void trd_setk_kernel(TrdDesc* __restrict__ desc, RankChunk chunk, int base, int count) {
  const int t = threadIdx.x;
  if (t < count) {
    desc[base + t].k = chunk.v[t];
    // 2: the team kernel gave up (sticky until ndmps_syevd_topk_recover_f64 has re-run the reduction)
    if (desc[base + t].status != 2) desc[base + t].status = 0;
  }
}
__global__ void trd_clear_status_kernel(TrdDesc* __restrict__ desc, int batch, TeamSync* __restrict__ sync) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) {
    desc[b].status = 0;
    sync[b].count = 0;
    sync[b].abort = 0;
  }
}

#include "eig_band.inc"
#include "eig_sym.inc"
#include "eig_panel.inc"
#include "eig_wide.inc"

// Test hook (ndmps_debug_inject_team_abort): what an aborted team launch leaves behind, without the 3 s wait
__global__ void trd_inject_abort_kernel(TrdDesc* __restrict__ desc, int batch) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) desc[b].status = 2;
}

// ------------------------------------------------------------------------------------------ host side
constexpr bool kTeamXcdDefault = true;   // teams placed XCD by XCD (team_place) when NDMPS_TRD_XCD is not set
constexpr bool kSymDefault = false;  // half-storage team kernel (batches beyond half the workgroup slots) when NDMPS_TRD_SYM is not set
constexpr int kWideOrthoMinOrder = 1024;  // chip-wide orthonormalisation of <= 128 vectors: from this order on ...
constexpr int kWideOrthoMaxBatch = 2;      // ... for at most this many matrices (they go one after the other)
constexpr int kBandDefault = 0;  // semi-bandwidth of the two-stage reduction when NDMPS_TRD_BAND is not set (0: off)

struct TrdLayout {
  int64_t n_max, lda, kp;
  int64_t off_a, off_vh, off_y, off_xc, off_yr, off_xr, off_sync, off_tau, off_d, off_e, off_lam, off_mu, off_bound, off_z, off_lu, off_piv, off_desc, off_desc2, off_stamps, off_tw, t_stride, off_yb, off_xb, off_band, off_qlog, q_stride, off_yrow, total;
  int64_t off_pv, off_pw, off_ypart, off_spart, off_ucol, off_napart, pnl_blocks, pnl_tiles;
  int64_t off_ws, off_wlinv, off_wgram, off_wt, wt_stride, kw, wgram_bytes, off_wpart, wpart_stride;  // more than kMaxK vectors (eig_wide.inc)
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
  l.off_xc = take(batch * 2 * l.lda * 8);
  l.off_yr = take(batch * 2 * l.lda * 16);
  l.off_xr = take(batch * 2 * l.lda * 16);
  l.off_sync = take(batch * (int64_t)sizeof(TeamSync));
  l.off_tau = take(batch * n_max * 8);
  l.off_d = take(batch * n_max * 8);
  l.off_e = take(batch * n_max * 8);
  l.off_lam = take(batch * n_max * 8);
  l.off_mu = take(batch * n_max * 8);
  l.off_bound = take(batch * 8);
  l.off_z = take(batch * n_max * l.kp * 8);
  l.off_lu = take(batch * 4 * n_max * l.kp * 8);
  l.off_piv = take(batch * n_max * l.kp);
  l.off_desc = take(batch * (int64_t)sizeof(TrdDesc));
  l.off_desc2 = take(batch * (int64_t)sizeof(TrdDesc));  // the trailing blocks the resident kernel finishes behind the panels
  l.off_stamps = take(batch * 16 * 8);
  l.t_stride = (n_max + 8) * 4;  // groups of WYB reflectors, WYB^2 doubles each, WYB <= 4
  l.off_tw = take(batch * l.t_stride * 8);
  // two-stage reduction (orders <= 512, eig_band.inc): exchange vectors, the band, the log of the bulge chase -- 2 MB per
  // order-512 matrix, reserved only when the environment (or the built-in default) selects that route.  The layout is a
  // function of its arguments and the environment alone: every call of a solve sees the same one.
  const bool band_ok = n_max <= 512 && (getenv("NDMPS_TRD_BAND") ? atoi(getenv("NDMPS_TRD_BAND")) != 0 : kBandDefault != 0);
  const bool sym_ok = n_max <= 512 && (getenv("NDMPS_TRD_SYM") ? atoi(getenv("NDMPS_TRD_SYM")) != 0 : kSymDefault);
  l.off_yb = take(band_ok ? batch * 2 * kBandMax * l.lda * 8 : 0);
  l.off_xb = take(band_ok ? batch * 2 * kBandMax * l.lda * 8 : 0);
  l.off_band = take(band_ok ? batch * n_max * 2 * kBandMax * 8 : 0);
  l.q_stride = band_ok ? n_max * (n_max + 2 * kBandMax) : 0;
  l.off_qlog = take(batch * l.q_stride * 8);
  l.off_yrow = take(sym_ok ? batch * 2 * 8 * l.lda * 8 : 0);
  // panel-blocked reduction (orders above 512, eig_panel.inc)
  const bool panel_ok = n_max > 512;
  l.pnl_blocks = ndmps::ceil_div(n_max, kPnlTB);
  l.pnl_tiles = l.pnl_blocks * (l.pnl_blocks + 1) / 2;
  l.off_pv = take(panel_ok ? batch * n_max * kPnlNB * 8 : 0);
  l.off_pw = take(panel_ok ? batch * n_max * kPnlNB * 8 : 0);
  l.off_ypart = take(panel_ok ? batch * l.pnl_blocks * l.lda * 8 : 0);
  l.off_spart = take(panel_ok ? batch * l.pnl_tiles * 8 : 0);
  l.off_ucol = take(panel_ok ? batch * 2 * l.lda * 8 : 0);
  l.off_napart = take(panel_ok ? batch * 2 * l.pnl_blocks * kPnlNa * 8 : 0);
  // more than kMaxK eigenvectors (eig_wide.inc): Gram matrix, L^-1 and the Gram kernel's partial tiles, one set for
  // the whole batch (its matrices are orthonormalised one after the other)
  // ... and one or two big matrices with up to kMaxK vectors: the single workgroup per matrix of trd_ortho_*_kernel
  // takes 1.2 ms for 128 vectors of order 2048, the chip 0.2 ms
  const bool wide_small = n_max >= kWideOrthoMinOrder && batch <= kWideOrthoMaxBatch;
  l.kw = (k_max > kMaxK || wide_small) ? ndmps::round_up(l.kp, kWB) : 0;
  l.wgram_bytes = l.kw ? ndmps_gram_f64_workspace_bytes(n_max, l.kp) : 0;
  l.off_ws = take(l.kw * l.kw * 8);
  l.off_wlinv = take(l.kw * l.kw * 8);
  l.off_wgram = take(l.wgram_bytes);
  l.wt_stride = l.kw ? ndmps::ceil_div(n_max, kBwB) * kBwB * kBwB : 0;  // T factors of the blocked back-transformation
  l.off_wt = take(batch * l.wt_stride * 8);
  // partial products of the row-dealt back-transformation (back_rows_step_kernel): [2][chunks][64][kp] per matrix
  l.wpart_stride = wide_small ? 2 * ndmps::ceil_div(n_max, kBrR) * kBwB * l.kp : 0;
  l.off_wpart = take(batch * l.wpart_stride * 8);
  l.total = ndmps::round_up(used, 256);
  return l;
}

TrdWork trd_work(const TrdLayout& l, void* d_ws) {
  char* base = (char*)d_ws;
  TrdWork w;
  w.A = (double*)(base + l.off_a);
  w.Vh = (double*)(base + l.off_vh);
  w.y = (double*)(base + l.off_y);
  w.xc = (double*)(base + l.off_xc);
  w.yr = (TeamRec*)(base + l.off_yr);
  w.xr = (TeamRec*)(base + l.off_xr);
  w.sync = (TeamSync*)(base + l.off_sync);
  w.tau = (double*)(base + l.off_tau);
  w.d = (double*)(base + l.off_d);
  w.e = (double*)(base + l.off_e);
  w.lam = (double*)(base + l.off_lam);
  w.mu = (double*)(base + l.off_mu);
  w.bound = (double*)(base + l.off_bound);
  w.Z = (double*)(base + l.off_z);
  w.lu = (double*)(base + l.off_lu);
  w.piv = (unsigned char*)(base + l.off_piv);
  w.stamps = (long long*)(base + l.off_stamps);
  w.Tw = (double*)(base + l.off_tw);
  w.t_stride = l.t_stride;
  w.yb = (double*)(base + l.off_yb);
  w.xb = (double*)(base + l.off_xb);
  w.band = (double*)(base + l.off_band);
  w.qlog = (double*)(base + l.off_qlog);
  w.q_stride = l.q_stride;
  w.yrow = (double*)(base + l.off_yrow);
  w.pv = (double*)(base + l.off_pv);
  w.pw = (double*)(base + l.off_pw);
  w.ypart = (double*)(base + l.off_ypart);
  w.spart = (double*)(base + l.off_spart);
  w.ucol = (double*)(base + l.off_ucol);
  w.napart = (double*)(base + l.off_napart);
  w.pnl_blocks_max = (int)l.pnl_blocks;
  w.pnl_tiles_max = (int)l.pnl_tiles;
  w.tail_lower = 0;
  w.xcd_team = w.xcd_count = w.xcd_pair = 0;
  w.n_max = (int)l.n_max;
  w.lda = (int)l.lda;
  w.kp = (int)l.kp;
  return w;
}

// inverse-iteration kernel: Cholesky factor [128][129], aliased by the scaled tridiagonal as (d, e) pairs
constexpr int kInvitLdsMax = kMaxK * (kMaxK + 1) * 8;  // >= kMaxN * 16
constexpr size_t kTailLds = ((size_t)kTail * kTailLd + 4 * kTail) * sizeof(double);  // matrix + RowVec[kTail]

// the resident launch may be switched off per host thread (the fallback after an abort runs the column launches)
thread_local int g_team_off = 0;
// Per host thread: the caller runs a STREAM of batches on several lanes (core/batch.py): small batches then take 32-column
// blocks (16 workgroups per order-512 matrix, half a turn) instead of the 8-column blocks that are fastest for one batch on
// an otherwise empty GPU (64 workgroups per matrix: eight matrices fill every slot and take the whole turn) as soon as
// the narrow launch would need more than half the slots -- the reductions of consecutive batches overlap: 8 volumes of
// 256^3 per batch 5.98 -> 4.8 - 5.0 ms, 8 x 512^3 (config 3) 23.2 -> 21.6 - 22.1.
thread_local int g_team_streamed = 0;
std::atomic<long long> g_team_fallbacks{0};
std::atomic<int> g_inject_abort{0};


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
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_team_kernel<8, true, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2048 * (int)sizeof(RowVec)));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_column_kernel<16, 32>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kMaxN * (int)sizeof(RowVec)));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_column_kernel<16, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kMaxN * (int)sizeof(RowVec)));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_invit_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kInvitLdsMax));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_ortho_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kInvitLdsMax));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_ortho_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, kInvitLdsMax));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&wide_chol_diag_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kWB * kWLd * (int)sizeof(double)));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_ortho_blocks_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 65 * 8));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_back2_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 32 * 8));
  NDMPS_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&trd_back2_kernel<4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 32 * 8));
  done[dev] = true;
  return NDMPS_OK;
}

// Semi-bandwidth of the two-stage reduction for a batch whose largest order is n_max, 0 = one-stage paths.  A pure
// function of its arguments and the environment: phase 1 and phase 2 of a solve must agree on it.
// The route phase 1 took on a workspace (semi-bandwidth of the two-stage reduction, 0: one-stage): phase 2 replays
// the reflector logs of THAT route, whatever the thread's switches say by then (a recovery redoes phase 1 with the
// resident launches off, i.e. one-stage; phase 2 then must not apply the bulge chase's reflectors).
std::mutex g_route_mu;
std::map<const void*, int> g_route;
void route_store(const void* ws, int bw) {
  std::lock_guard<std::mutex> lock(g_route_mu);
  if (g_route.size() > 256) g_route.clear();  // workspaces come and go; a forgotten entry falls back to band_width_for
  g_route[ws] = bw;
}
int route_load(const void* ws, int fallback) {
  std::lock_guard<std::mutex> lock(g_route_mu);
  auto it = g_route.find(ws);
  return it == g_route.end() ? fallback : it->second;
}

int band_width_for(int64_t n_max) {
  if (n_max > 512 || n_max <= kTail || getenv("NDMPS_TRD_NO_TEAM") || g_team_off) return 0;
  const char* e = getenv("NDMPS_TRD_BAND");
  const int bw = e ? atoi(e) : kBandDefault;
  return (bw == 2 || bw == 4) ? bw : 0;
}

// workgroups of trd_team_kernel the current device keeps resident at once (occupancy x compute units); rows_per_thread:
// 2 (orders <= 512), 4 (<= 1024) or 8 (<= 2048: one workgroup per CU)
int team_slots(int& slots, int rows_per_thread = 2) {
  static std::mutex mu;
  static int cached[3][64] = {};
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  NDMPS_REQUIRE(dev >= 0 && dev < 64, "device index %d outside [0, 64)", dev);
  const int cls = rows_per_thread <= 2 ? 0 : rows_per_thread <= 4 ? 1 : 2;
  std::lock_guard<std::mutex> lock(mu);
  if (cached[cls][dev] == 0) {
    int per_cu = 0, cus = 0;
    if (cls == 0) NDMPS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trd_team_kernel<2, false>, 256, 512 * sizeof(RowVec)));
    else if (cls == 1) NDMPS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trd_team_kernel<4, true, 8>, 256, 1024 * sizeof(RowVec)));
    else NDMPS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, trd_team_kernel<8, true, 8>, 256, 2048 * sizeof(RowVec)));
    NDMPS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    // the register budget allows two 256-thread workgroups per CU (one of the 8-row kernel); never count on more
    cached[cls][dev] = std::max(1, std::min(per_cu, cls == 2 ? 1 : 2)) * std::max(cus, 1);
  }
  slots = cached[cls][dev];
  return NDMPS_OK;
}

// At most one team kernel in flight per device (see trd_team_kernel); the turn is taken on the device
// (ndmps::Turn, util.hip).
template <typename F>
int team_launch(hipStream_t s, F&& launch, bool half = false) {
  // a launch that fills one workgroup slot per CU takes half the turn: two of them fit the GPU together
  ndmps::Turn turn(s, ndmps::kTurnTeam, half && !getenv("NDMPS_TEAM_FULL_TURN") ? 1u : 2u, 2u);
  NDMPS_TRY(turn.begin());
  launch();
  return turn.end();
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

// Smallest order that takes the panel-blocked reduction.  A dependent launch costs ~5 us here whatever it does: two of
// them per column (10.7 us at any order) beat the column launch -- one launch, but the trailing matrix read and
// written, 7.6 us per column at order 1024, 13.5 at 2048, 35 at 4096 -- from about order 1500 on
// (profiles/r04_f_panel_probe.txt: 9.6 vs 6.8 ms at 1024, 22.2 vs 26.2 at 2048, 63 vs 140 at 4096).  When the last
// 512 columns go to the resident kernel (`hybrid` below) the break-even is order 1024 for one matrix (6.9 vs 6.8 ms)
// and far below for a batch (8 of order 1024: 8.8 vs 12.6 ms; profiles/r04_w_panel_probe.txt).
int64_t panel_min_order(bool hybrid) {
  const char* e = getenv("NDMPS_TRD_PANEL_MIN");
  const int64_t v = e ? atoll(e) : hybrid ? 1024 : 1536;
  return std::max<int64_t>(v, 513);  // the workspace holds the panel arrays from order 513 on
}

// ---- launch sequence of the panel-blocked reduction, eager (default) or as a cached graph
struct PnlLaunch {
  int kind;  // 0 pnl_vec_kernel, 1 pnl_update_kernel, 2 pnl_symv_kernel
  unsigned grid_x;
  int j;
};
struct PnlGraph {
  int dev;
  const void* desc;  // the descriptors live in the caller's workspace: workspace address, layout and sizes name a graph
  const void* a;
  int64_t n_max;
  int batch, tail_cols;
  std::vector<int64_t> sizes;
  hipGraph_t graph;
  hipGraphExec_t exec;
  uint64_t used;
};
constexpr size_t kPnlGraphs = 8;  // least recently used beyond that is destroyed

int pnl_run(const std::vector<PnlLaunch>& seq, int batch, const int64_t* h_n, int64_t n_max, int tail_cols, TrdDesc* desc,
            const TrdWork& w, hipStream_t s) {
  const unsigned B = (unsigned)batch;
  void (*const vec)(const TrdDesc*, TrdWork, int, int, int) =
      n_max <= 1024 ? pnl_vec_kernel<4> : (n_max <= 2048 ? pnl_vec_kernel<8> : pnl_vec_kernel<16>);
  void (*const kernels[3])(const TrdDesc*, TrdWork, int, int, int) = {vec, pnl_update_kernel, pnl_symv_kernel};
  int n_uniform = (int)h_n[0];  // equal orders: the kernels take the order from their arguments
  for (int b = 1; b < batch; ++b)
    if (h_n[b] != h_n[0]) n_uniform = 0;
  // Launched one by one unless NDMPS_TRD_PANEL_GRAPH is set.  The replay of a cached graph takes the same device time
  // (the sequence is bound by the dependent launches, not by the host: 24.7 ms either way at order 2048) and only frees
  // the host thread -- but a graph holds the workspace addresses, and instantiating one (3900 nodes) costs about a
  // second: a caller whose workspace moves between calls would pay that every time.
  if (!getenv("NDMPS_TRD_PANEL_GRAPH")) {
    for (const PnlLaunch& q : seq)
      hipLaunchKernelGGL(kernels[q.kind], dim3(q.grid_x, B), dim3(256), 0, s, (const TrdDesc*)desc, w, q.j, n_uniform, tail_cols);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }
  static std::mutex mu;
  static std::vector<PnlGraph> cache;
  static uint64_t clock = 0;
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  hipGraphExec_t exec = nullptr;
  {
    std::lock_guard<std::mutex> lock(mu);
    for (PnlGraph& g : cache)
      if (g.dev == dev && g.desc == desc && g.a == w.A && g.n_max == w.n_max && g.batch == batch && g.tail_cols == tail_cols &&
          std::equal(g.sizes.begin(), g.sizes.end(), h_n)) {
        g.used = ++clock;
        exec = g.exec;
        break;
      }
    if (!exec) {
      PnlGraph g;
      g.dev = dev, g.desc = desc, g.a = w.A, g.n_max = w.n_max, g.batch = batch, g.tail_cols = tail_cols;
      g.sizes.assign(h_n, h_n + batch);
      NDMPS_CHECK_HIP(hipGraphCreate(&g.graph, 0));
      hipGraphNode_t prev = nullptr;
      for (const PnlLaunch& q : seq) {
        const TrdDesc* d_arg = desc;
        TrdWork w_arg = w;
        int j_arg = q.j, n_arg = n_uniform, t_arg = tail_cols;
        void* args[5] = {&d_arg, &w_arg, &j_arg, &n_arg, &t_arg};
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = reinterpret_cast<void*>(kernels[q.kind]);
        kp.gridDim = dim3(q.grid_x, B);
        kp.blockDim = dim3(256);
        kp.sharedMemBytes = 0;
        kp.kernelParams = args;
        hipGraphNode_t node = nullptr;
        const hipError_t e = hipGraphAddKernelNode(&node, g.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp);
        if (e != hipSuccess) {
          (void)hipGraphDestroy(g.graph);
          ndmps::set_error("hipGraphAddKernelNode failed: %s", hipGetErrorString(e));
          return NDMPS_EHIP;
        }
        prev = node;
      }
      const hipError_t e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
      if (e != hipSuccess) {
        (void)hipGraphDestroy(g.graph);
        ndmps::set_error("hipGraphInstantiate failed: %s", hipGetErrorString(e));
        return NDMPS_EHIP;
      }
      g.used = ++clock;
      if (cache.size() >= kPnlGraphs) {
        size_t oldest = 0;
        for (size_t i = 1; i < cache.size(); ++i)
          if (cache[i].used < cache[oldest].used) oldest = i;
        (void)hipDeviceSynchronize();  // rare (more than kPnlGraphs workspaces in use): its last replay may still run
        (void)hipGraphExecDestroy(cache[oldest].exec);
        (void)hipGraphDestroy(cache[oldest].graph);
        cache.erase(cache.begin() + (long)oldest);
      }
      exec = g.exec;
      cache.push_back(std::move(g));
    }
    NDMPS_CHECK_HIP(hipGraphLaunch(exec, s));  // under the lock: an entry is never destroyed between look-up and launch
  }
  return NDMPS_OK;
}

// Resident tridiagonalisation (trd_team_kernel and variants) of `batch` matrices of order <= 512 named by `desc`, all
// columns up to the last kTail in one launch (or a few, by the resident slots); takes the device-side turn.  Used for
// orders <= 512 and for the last 512 columns behind the panel-blocked reduction (a view of the workspace).
std::atomic<unsigned> g_team_epoch{1};  // names a launch in the tags of its exchange records
int trd_team_reduce(int batch, int64_t n_max, TrdDesc* desc, TrdWork& w, hipStream_t s) {
    if (n_max > 512) {
      // orders 513 .. 2048: 8-column blocks, 4 or 8 rows per thread, tagged records; 128 / 256 workgroups per matrix,
      // so a team spans XCDs whatever the placement (the exchange is agent-scope: MALL, not an XCD's L2)
      NDMPS_REQUIRE(n_max <= 2048, "resident reduction of order %lld > 2048", (long long)n_max);
      const int nr = n_max <= 1024 ? 4 : 8;
      int slots = 0;
      NDMPS_TRY(team_slots(slots, nr));
      const int team_size = (int)ndmps::ceil_div(n_max, 8);
      const int per_launch = std::max(1, slots / team_size);
      w.tail_lower = 0;
      int inject = g_inject_abort.load();
      while (inject > 0 && !g_inject_abort.compare_exchange_weak(inject, inject - 1)) {
      }
      if (inject > 0) {
        hipLaunchKernelGGL(trd_inject_abort_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, desc, batch);
        return NDMPS_OK;
      }
      return team_launch(s, [&]() {
        for (int b0 = 0; b0 < batch; b0 += per_launch) {
          const unsigned epoch = g_team_epoch.fetch_add(1);
          const dim3 grid((unsigned)team_size, (unsigned)std::min(per_launch, batch - b0));
          TrdWork wl = w;
          wl.xcd_count = 0;
          if (nr == 4) hipLaunchKernelGGL((trd_team_kernel<4, true, 8>), grid, dim3(256), 1024 * sizeof(RowVec), s, desc, wl, b0, epoch);
          else hipLaunchKernelGGL((trd_team_kernel<8, true, 8>), grid, dim3(256), 2048 * sizeof(RowVec), s, desc, wl, b0, epoch);
        }
      });
    }
    // a launch never holds more workgroups than the device keeps resident at once: no team then depends on the
    // order in which the dispatcher places workgroups (larger batches go in several launches)
    int slots = 0;
    NDMPS_TRY(team_slots(slots));
    // one to four matrices: 8-column blocks (64 workgroups per order-512 matrix, one per CU)
    // 8-column blocks while they fit the slots -- HALF the slots when the caller keeps several batches in flight (a narrow
    // launch that takes half a turn leaves room for its neighbours' as well; measured on three lanes, ms per batch of 256^3
    // volumes, narrow / wide: 1 volume 2.8 / 3.2, 2: 3.2 / 3.4, 4: 3.8 / 3.9, 8: 6.0 / 5.0).  NDMPS_TRD_TEAM_NARROW=1: A/B.
    const bool streamed = g_team_streamed && !getenv("NDMPS_TRD_TEAM_NARROW");
    const bool narrow_team = (int64_t)batch * ndmps::ceil_div(n_max, 8) <= (streamed ? slots / 2 : slots) &&
                             !getenv("NDMPS_TRD_TEAM_WIDE");
    // opt-in, batches that fill more than half the slots (17 order-512 matrices and more): half storage (eig_sym.inc),
    // 8 workgroups per order-512 matrix -- a group of 32 takes one workgroup slot per CU and shares the GPU with the
    // other group's reduction or kernels
    const char* sym_env = getenv("NDMPS_TRD_SYM");
    const bool sym = !narrow_team && n_max <= 512 && (sym_env ? atoi(sym_env) != 0 : kSymDefault) &&
                     (int64_t)batch * ndmps::ceil_div(n_max, 32) > slots / 2;
    w.tail_lower = sym ? 1 : 0;
    const int team_size = sym ? (int)ndmps::ceil_div(ndmps::ceil_div(n_max, 32), 2) : (int)ndmps::ceil_div(n_max, narrow_team ? 8 : 32);
    // NDMPS_TRD_TEAM_HALF=1 (A/B): launches of half the slots (16 order-512 matrices, one workgroup per CU) under a half turn,
    // so that the reductions of two batches in flight run side by side instead of alternating
    const bool half_launches = getenv("NDMPS_TRD_TEAM_HALF") != nullptr && !narrow_team && !sym;
    const int per_launch = std::max(1, (half_launches ? slots / 2 : slots) / team_size);
    const char* xcd_env = getenv("NDMPS_TRD_XCD");
    const bool xcd_placed = xcd_env ? atoi(xcd_env) != 0 : kTeamXcdDefault;
    int inject = g_inject_abort.load();
    while (inject > 0 && !g_inject_abort.compare_exchange_weak(inject, inject - 1)) {
    }
    if (inject > 0) {
      hipLaunchKernelGGL(trd_inject_abort_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, desc, batch);
    } else
    NDMPS_TRY(team_launch(s, [&]() {
      // 8-column blocks exchange without meetings (tagged records), 32-column blocks meet at a counter: a tagged
      // 32-column kernel was 3 % faster for 9 .. 16 order-512 matrices (1.84 vs 1.90 ms) and as fast for 32, but its 128
      // registers of matrix per thread leave no room for the records in flight (round 4: spills, removed)
      for (int b0 = 0; b0 < batch; b0 += per_launch) {
        const unsigned epoch = g_team_epoch.fetch_add(1);
        const int count = std::min(per_launch, batch - b0);
        TrdWork wl = w;
        dim3 grid((unsigned)team_size, (unsigned)count);
        if (xcd_placed) {
          wl.xcd_team = team_size;
          wl.xcd_count = count;
          // early leavers (low block-columns) beside late ones: needs whole teams on either side of every 32nd workgroup
          const bool pairable = 32 % team_size == 0 || team_size % 32 == 0;
          wl.xcd_pair = sym || !pairable ? 0 : getenv("NDMPS_TRD_PAIR") ? atoi(getenv("NDMPS_TRD_PAIR")) : 1;
          grid = dim3((unsigned)(team_size * count));
        }
        if (sym) hipLaunchKernelGGL(trd_sym_kernel, grid, dim3(256), 0, s, desc, wl, b0);
        else if (narrow_team) hipLaunchKernelGGL((trd_team_kernel<2, true, 8>), grid, dim3(256), 512 * sizeof(RowVec), s, desc, wl, b0, epoch);
        else hipLaunchKernelGGL((trd_team_kernel<2, false>), grid, dim3(256), 512 * sizeof(RowVec), s, desc, wl, b0, epoch);
      }
    }, (int64_t)std::min(per_launch, batch) * team_size <= slots / 2));  // half a turn only for what fits half the slots
  return NDMPS_OK;
}

// Tridiagonalisation of every matrix named by the descriptors (resident launch for orders <= 512 unless switched
// off, column launches otherwise), the last kTail columns in LDS, then the min(k_max, n) largest eigenvalues.
int trd_reduce_and_values(int batch, const int64_t* h_n, int64_t n_max, int64_t k_max, const TrdLayout& l, const TrdWork& w_in,
                          TrdDesc* desc, hipStream_t s) {
  TrdWork w = w_in;  // tail_lower is decided here
  const unsigned B = (unsigned)batch;
  const int load_grid = (int)std::min<int64_t>(ndmps::ceil_div(n_max * l.lda, 256), 512);
  hipLaunchKernelGGL(trd_load_kernel, dim3(load_grid, B), dim3(256), 0, s, desc, w);
  // narrow column blocks while the grid stays below ~2 workgroups per CU (see trd_column_kernel)
  const bool narrow = (int64_t)batch * ndmps::ceil_div(n_max, 8) <= 2 * ndmps::kNumCU && !getenv("NDMPS_TRD_WIDE");
  const int W = (int)ndmps::ceil_div(n_max, narrow ? 8 : 32);
  const size_t col_lds = (size_t)n_max * sizeof(RowVec);
  void (*column)(const TrdDesc*, TrdWork, int);
  if (narrow)
    column = n_max <= 512    ? trd_column_kernel<2, 8>
             : n_max <= 1024 ? trd_column_kernel<4, 8>
             : n_max <= 2048 ? trd_column_kernel<8, 8>
                             : trd_column_kernel<16, 8>;
  else
    column = n_max <= 512    ? trd_column_kernel<2, 32>
             : n_max <= 1024 ? trd_column_kernel<4, 32>
             : n_max <= 2048 ? trd_column_kernel<8, 32>
                             : trd_column_kernel<16, 32>;
  // orders <= 512: one resident launch for all columns (trd_team_kernel; 2.2 ms for 1 .. 16 matrices of order
  // 512, 2.5 ms for 32, against 2.4 / 3.8 / 5.7 ms of column launches); NDMPS_TRD_NO_TEAM=1 keeps the column
  // launches (A/B timing, tests of that path)
  const int bw = band_width_for(n_max);
  route_store(w_in.A, bw);
  const bool team = !bw && n_max <= 512 && n_max > kTail && !getenv("NDMPS_TRD_NO_TEAM") && !g_team_off;
  if (bw) {
    // two-stage reduction (eig_band.inc): dense -> band with one exchange per panel, then the bulge chase
    int slots = 0;
    NDMPS_TRY(team_slots(slots));
    const int team_size = (int)ndmps::ceil_div(n_max, 32);
    const int per_launch = std::max(1, slots / team_size);
    int inject = g_inject_abort.load();
    while (inject > 0 && !g_inject_abort.compare_exchange_weak(inject, inject - 1)) {
    }
    void* span = ndmps::span_begin(s);
    if (inject > 0) {
      hipLaunchKernelGGL(trd_inject_abort_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, desc, batch);
    } else {
      NDMPS_TRY(team_launch(s, [&]() {
        for (int b0 = 0; b0 < batch; b0 += per_launch) {
          const dim3 grid((unsigned)team_size, (unsigned)std::min(per_launch, batch - b0));
          if (bw == 2) hipLaunchKernelGGL(trd_band_kernel<2>, grid, dim3(256), 0, s, desc, w, b0);
          else hipLaunchKernelGGL(trd_band_kernel<4>, grid, dim3(256), 0, s, desc, w, b0);
        }
      }));
    }
    int64_t bytes = 0;
    for (int b = 0; b < batch; ++b) bytes += 2 * 8 * h_n[b] * h_n[b];
    ndmps::span_end(span, s, ndmps::kSpanTridiagTeam, 1, bytes);
    const size_t chase_lds = (size_t)n_max * 2 * bw * sizeof(double);
    if (bw == 2) hipLaunchKernelGGL(trd_chase_kernel<2>, dim3(1, B), dim3(1024), chase_lds, s, desc, w);
    else hipLaunchKernelGGL(trd_chase_kernel<4>, dim3(1, B), dim3(1024), chase_lds, s, desc, w);
    const int kk = (int)std::min(k_max, n_max);
    hipLaunchKernelGGL(trd_bisect_kernel, dim3(ndmps::ceil_div(kk, 4), B), dim3(256),
                       (size_t)ndmps::round_up(n_max, 16) * 16, s, desc, w, kk);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }
  // Orders 513 .. 2048 whose teams (n / 8 workgroups each) are all resident at once: the whole reduction in the
  // resident kernel (3 - 5 us per column against 11 us of the two panel launches).  NDMPS_TRD_TEAM_MAX=512 keeps it
  // to the orders of the lockstep groups (A/B timing, tests of the other routes).
  const bool team_on = !bw && !g_team_off && !getenv("NDMPS_TRD_NO_TEAM");
  const int64_t team_max = getenv("NDMPS_TRD_TEAM_MAX") ? std::max<int64_t>(atoll(getenv("NDMPS_TRD_TEAM_MAX")), 512) : 2048;
  auto team_fits = [&](int64_t order, bool& fits) -> int {
    int slots = 0;
    NDMPS_TRY(team_slots(slots, order <= 512 ? 2 : order <= 1024 ? 4 : 8));
    fits = (int64_t)batch * ndmps::ceil_div(order, 8) <= slots;
    return NDMPS_OK;
  };
  bool big_team = false;
  if (team_on && n_max > 512 && n_max <= std::min<int64_t>(team_max, 2048)) NDMPS_TRY(team_fits(n_max, big_team));
  // Hand-over of a panel-blocked reduction to the resident kernel for its last 2048 / 1024 / 512 columns.  Needs one
  // order for the whole batch (the view below is one offset into the workspace), an even one (the parity of a column
  // selects its exchange buffer on both sides), the resident launches switched on, and every team of the batch
  // resident in one launch.
  int hybrid = 0;
  if (team_on && !big_team && n_max > 512 && n_max % 2 == 0 && !getenv("NDMPS_TRD_NO_HYBRID")) {
    bool uniform = true;
    for (int b = 0; b < batch && uniform; ++b) uniform = h_n[b] == n_max;
    for (int h = 2048; h >= 512 && uniform && !hybrid; h /= 2) {
      bool fits = false;
      if (h > team_max || n_max - h < kPnlTB) continue;
      NDMPS_TRY(team_fits(h, fits));
      if (fits) hybrid = h;
    }
  }
  void* span = ndmps::span_begin(s);
  int64_t span_bytes = 0;  // algorithmic: every trailing element read once and written once per column
  if (team || big_team) {
    NDMPS_TRY(trd_team_reduce(batch, n_max, desc, w, s));
    // algorithmic traffic of the resident reduction: the matrix in, the reflectors out
    for (int b = 0; b < batch; ++b) span_bytes += 2 * 8 * h_n[b] * h_n[b];
    ndmps::span_end(span, s, ndmps::kSpanTridiagTeam, 1, span_bytes);
  } else if (n_max >= panel_min_order(hybrid != 0) && !getenv("NDMPS_TRD_NO_PANEL")) {
    // panel-blocked reduction (eig_panel.inc): two launches per column, one update per panel of kPnlNB columns
    NDMPS_CHECK_HIP(hipMemsetAsync(w.Vh, 0, (size_t)batch * n_max * l.lda * 8, s));
    NDMPS_CHECK_HIP(hipMemsetAsync(w.pv, 0, (size_t)batch * n_max * kPnlNB * 8, s));
    NDMPS_CHECK_HIP(hipMemsetAsync(w.pw, 0, (size_t)batch * n_max * kPnlNB * 8, s));
    const int tail_cols = hybrid ? hybrid : kTail;
    w.tail_lower = hybrid ? 0 : 1;
    const int J_max = (int)n_max - tail_cols;
    std::vector<char> ends((size_t)J_max + 1, 0);
    for (int b = 0; b < batch; ++b) ends[(size_t)std::max<int64_t>(h_n[b] - tail_cols, 0)] = 1;
    const int nbm = (int)l.pnl_blocks;
    auto tiles = [&](int first_col) {
      const int nblk = nbm - first_col / kPnlTB;
      return (unsigned)(nblk * (nblk + 1) / 2);
    };
    // the launch sequence: (kernel, grid.x, column)
    std::vector<PnlLaunch> seq;
    seq.reserve((size_t)2 * J_max + J_max / kPnlNB + 4);
    for (int j = 0; j <= J_max; ++j) {
      seq.push_back({0, (unsigned)(nbm - j / kPnlTB), j});
      if (j >= 1 && (j % kPnlNB == 0 || ends[(size_t)j])) seq.push_back({1, tiles(j), j});
      if (j < J_max) seq.push_back({2, tiles(j + 1), j});
      if (span)
        for (int b = 0; b < batch; ++b)
          if (j < h_n[b] - tail_cols) span_bytes += 4 * (h_n[b] - j - 1) * (h_n[b] - j - 1);  // the lower half, read once
    }
    NDMPS_TRY(pnl_run(seq, batch, h_n, n_max, tail_cols, desc, w, s));
    if (hybrid) {
      // the trailing 512 x 512 block in full storage, then the resident kernel on a VIEW of the workspace: matrix,
      // reflector rows, T's diagonals and the pending product of its last column all sit at offset J0 of the big
      // arrays, so the tail kernel finds everything where the column launches would have left it
      const int64_t J0 = n_max - hybrid;
      TrdDesc* sub = (TrdDesc*)((char*)w_in.A - l.off_a + l.off_desc2);
      hipLaunchKernelGGL(pnl_mirror_kernel, dim3(256, B), dim3(256), 0, s, w, (int)n_max, hybrid);
      hipLaunchKernelGGL(pnl_subdesc_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, (const TrdDesc*)desc, sub, batch, hybrid);
      TrdWork v = w;
      v.A = w.A + J0 * l.lda + J0;
      v.Vh = w.Vh + J0 * l.lda + J0;
      v.y = w.y + J0;
      v.tau = w.tau + J0;
      v.d = w.d + J0;
      v.e = w.e + J0;
      NDMPS_TRY(trd_team_reduce(batch, hybrid, sub, v, s));
      hipLaunchKernelGGL(pnl_substatus_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, desc, (const TrdDesc*)sub, batch);
      w.tail_lower = 0;  // the resident kernel hands the trailing block over in full storage
    } else {
      // nothing is pending when the tail kernel takes over: its update of "the last column launch" must vanish
      NDMPS_CHECK_HIP(hipMemsetAsync(w.y, 0, (size_t)batch * 2 * l.lda * 8, s));
    }
    ndmps::span_end(span, s, ndmps::kSpanTridiagPanel, (int64_t)seq.size(), span_bytes);
  } else {
    for (int j = 0; j < n_max - kTail; ++j) {
      hipLaunchKernelGGL(column, dim3(W, B), dim3(256), col_lds, s, desc, w, j);
      if (span)
        for (int b = 0; b < batch; ++b)
          if (j < h_n[b] - kTail) span_bytes += 2 * 8 * (h_n[b] - j - 1) * (h_n[b] - j - 1);
    }
    ndmps::span_end(span, s, ndmps::kSpanTridiagColumns, std::max<int64_t>(n_max - kTail, 0), span_bytes);
  }
  // small batches (nothing else on the GPU): the tail in registers, two barriers per column; lockstep groups: the tail
  // in LDS -- 70 registers per thread, it starts beside the resident kernel's last workgroups (158 would wait)
  const char* tail_env = getenv("NDMPS_TRD_TAIL");
  const bool tail_regs = tail_env ? !strcmp(tail_env, "regs") : batch < 16;
  void* tail_span = ndmps::span_begin(s);
  if (tail_regs) hipLaunchKernelGGL(trd_tail_reg_kernel, dim3(1, B), dim3(512), 0, s, desc, w);
  else hipLaunchKernelGGL(trd_tail_kernel, dim3(1, B), dim3(512), kTailLds, s, desc, w);
  const int kk = (int)std::min(k_max, n_max);
  hipLaunchKernelGGL(trd_bisect_kernel, dim3(ndmps::ceil_div(kk, 4), B), dim3(256),
                     (size_t)ndmps::round_up(n_max, 16) * 16, s, desc, w, kk);
  ndmps::span_end(tail_span, s, ndmps::kSpanTridiagTail, 2, 0);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

}  // namespace

// Cholesky factor of a symmetric positive definite matrix: d_S (n x n, ld n, the lower triangle is read) <- L (lower,
// zeros above), S = L L^T; blocked right-looking on the fp64 MFMA (the kernels of eig_wide.inc).  d_scratch:
// ndmps_potrf_scratch_elems(n) doubles (the inverses of the diagonal blocks and the breakdown flag).  Synchronises;
// *h_status = 1 if a pivot was not positive (not numerically positive definite: d_S is then undefined).  compress()
// takes it for the square root of G2 = T2 T2^T, which any factor with L L^T = G2 serves (tt.hip).
extern "C" int64_t ndmps_potrf_scratch_elems(int64_t n) {
  const int64_t kw = ndmps::round_up(std::max<int64_t>(n, 1), kWB);
  return kw * kw + 64;
}
extern "C" int ndmps_potrf_lower_f64(double* d_S, int64_t n, double* d_scratch, int* h_status, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_S && d_scratch && h_status && n >= 1 && n <= 32768, "bad Cholesky argument");
  NDMPS_TRY(trd_opt_in());
  hipStream_t s = (hipStream_t)stream;
  const int k = (int)n, kw = (int)ndmps::round_up(n, kWB), nblk = kw / kWB;
  double* linv = d_scratch;
  TrdDesc* d_flag = reinterpret_cast<TrdDesc*>(d_scratch + (int64_t)kw * kw);  // its status word is the breakdown flag
  NDMPS_CHECK_HIP(hipMemsetAsync(d_flag, 0, sizeof(TrdDesc), s));
  for (int pb = 0; pb < nblk; ++pb) {
    const int p = kWB * pb, below = nblk - pb - 1;
    hipLaunchKernelGGL(wide_chol_diag_kernel, dim3(1), dim3(512), (size_t)2 * kWB * kWLd * sizeof(double), s, d_S, k, linv, kw, k, p,
                       d_flag);
    if (below > 0) {
      hipLaunchKernelGGL(wide_chol_panel_kernel, dim3((unsigned)below), dim3(256), 0, s, d_S, k, (const double*)linv, kw, k, p);
      hipLaunchKernelGGL(wide_chol_trail_kernel, dim3((unsigned)(below * (below + 1) / 2)), dim3(256), 0, s, d_S, k, k, p);
    }
  }
  hipLaunchKernelGGL(wide_zero_upper_kernel, dim3((unsigned)std::min<int64_t>(ndmps::ceil_div(n * n, 256), 2048)), dim3(256), 0, s,
                     d_S, k);
  NDMPS_LAUNCH_CHECK();
  TrdDesc host;
  NDMPS_CHECK_HIP(hipMemcpyAsync(&host, d_flag, sizeof(TrdDesc), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  *h_status = host.status != 0 ? 1 : 0;
  return NDMPS_OK;
}

// phase marks of the inverse-iteration kernel of matrix b (16 x int64, 100 MHz): profiling aid
extern "C" int64_t ndmps_syevd_topk_stamps_offset(int64_t n_max, int batch, int64_t k_max) {
  if (n_max <= 0 || n_max > kMaxN || batch <= 0 || k_max <= 0 || k_max > kMaxN) return -1;
  return trd_layout(n_max, batch, std::min(k_max, n_max)).off_stamps;
}
extern "C" int64_t ndmps_syevd_topk_max_n(void) { return kMaxN; }
extern "C" int64_t ndmps_syevd_topk_max_k(void) { return kMaxK; }

// largest k of the two-phase calls (ndmps_syevd_topk_values_f64 / _vectors_f64): every eigenpair of any order the solver
// takes.  Above ndmps_syevd_topk_max_k() vectors the orthonormalisation runs across the chip (eig_wide.inc); the
// device-side rank decision (ndmps_syevd_topk_vectors_auto_f64) stays with at most ndmps_syevd_topk_max_k().
extern "C" int64_t ndmps_syevd_topk_max_k_wide(void) { return kMaxN; }

extern "C" int64_t ndmps_syevd_topk_workspace_bytes(int64_t n_max, int batch, int64_t k_max) {
  if (n_max <= 0 || n_max > kMaxN || batch <= 0 || k_max <= 0 || k_max > kMaxN) return 0;
  return trd_layout(n_max, batch, std::min(k_max, n_max)).total;
}

// Phase 1: tridiagonalise and deliver the min(k_max, n) largest eigenvalues (descending) in d_w, zeros behind
// them.  Asynchronous on `stream`.
extern "C" int ndmps_syevd_topk_values_f64(int batch, const double* d_G, int64_t stride_G, const int64_t* h_n,
                                           double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                           int64_t k_max, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  int64_t n_max = 0;
  NDMPS_TRY(trd_check_sizes(batch, h_n, n_max));
  NDMPS_REQUIRE(d_G && d_V && d_w, "NULL eigen operand");
  NDMPS_REQUIRE(k_max >= 1 && k_max <= kMaxN, "k_max=%lld outside [1, %d]", (long long)k_max, kMaxN);
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
    hipLaunchKernelGGL(trd_setdesc_kernel, dim3(1), dim3(kDescChunk), 0, s, desc, chunk, base, count, w.sync);
  }
  return trd_reduce_and_values(batch, h_n, n_max, k_max, l, w, desc, s);
}

namespace {
// A stream per device for work that depends only on phase 1 and is needed late in phase 2 (the T factors of the blocked
// back-transformations: 0.37 ms at order 2048 while inverse iteration and orthonormalisation keep a handful of CUs busy).
// fork: the side stream waits for what `s` holds so far; join: `s` waits for what the side stream was given since.
struct SideFork {
  hipStream_t side = nullptr;
  hipEvent_t done = nullptr;
};
int side_stream(hipStream_t& out) {
  static std::mutex mu;
  static hipStream_t streams[64] = {};
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  NDMPS_REQUIRE(dev >= 0 && dev < 64, "device index %d outside [0, 64)", dev);
  std::lock_guard<std::mutex> lock(mu);
  if (!streams[dev]) NDMPS_CHECK_HIP(hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking));
  out = streams[dev];
  return NDMPS_OK;
}
int side_fork(hipStream_t s, SideFork& f) {
  if (getenv("NDMPS_NO_SIDE_STREAM")) return NDMPS_OK;  // f.side stays null: the caller launches on `s`
  hipStream_t side = nullptr;
  NDMPS_TRY(side_stream(side));
  hipEvent_t here = nullptr;
  NDMPS_CHECK_HIP(hipEventCreateWithFlags(&here, hipEventDisableTiming));
  NDMPS_CHECK_HIP(hipEventRecord(here, s));
  NDMPS_CHECK_HIP(hipStreamWaitEvent(side, here, 0));
  NDMPS_CHECK_HIP(hipEventDestroy(here));  // released once it has completed
  f.side = side;
  return NDMPS_OK;
}
int side_join(hipStream_t s, SideFork& f) {
  if (!f.side) return NDMPS_OK;
  NDMPS_CHECK_HIP(hipEventCreateWithFlags(&f.done, hipEventDisableTiming));
  NDMPS_CHECK_HIP(hipEventRecord(f.done, f.side));
  NDMPS_CHECK_HIP(hipStreamWaitEvent(s, f.done, 0));
  NDMPS_CHECK_HIP(hipEventDestroy(f.done));
  f.side = nullptr;
  return NDMPS_OK;
}

// Width of the column blocks of the inverse iteration.  The recurrences of a column are a dependent chain over the rows
// whatever the width; what a block's ONE CU adds is the traffic of its helpers -- three operand arrays in, one out per sweep
// and column, 8.4 MB per sweep at order 2048 with 128 columns, at the 30 - 60 GB/s a single CU gets.  From order 1024 on the
// columns go to as many CUs as the launch leaves free (blocks of 16 at least); lockstep groups of order <= 512 keep one
// block per matrix (their CUs are wanted by the other group's kernels).  NDMPS_INVIT_CB=16|32|64|128 forces a width.
int invit_block_width(int batch, int64_t n_max, int kp) {
  static const int forced = [] {
    const char* e = getenv("NDMPS_INVIT_CB");
    const int v = e ? atoi(e) : 0;
    return (v == 16 || v == 32 || v == 64 || v == 128) ? v : 0;
  }();
  if (forced) return forced;
  if (n_max < 1024) return kMaxK;
  int cb = kMaxK;
  while (cb > 16 && (int64_t)batch * ndmps::ceil_div(kp, cb / 2) <= 256) cb /= 2;
  return cb;
}

// inverse iteration + back-transformation for ranks already stored in the descriptors; kk: largest rank any
// matrix may have (sizes the launches), k_fill: columns zero-filled beyond a matrix's rank
int trd_launch_vectors(int batch, int64_t n_max, int kk, int k_fill, TrdDesc* desc, const TrdWork& w, hipStream_t s,
                       const TrdLayout* wide_layout = nullptr, void* d_ws = nullptr, const int64_t* h_n = nullptr,
                       const int64_t* h_k = nullptr) {
  const unsigned B = (unsigned)batch;
  const int k16 = (kk + 15) & ~15;
  void* vec_span = ndmps::span_begin(s);
  // the T factors of a blocked back-transformation need the reflectors only: on the side stream, beside everything up to
  // the orthonormalisation
  const bool blocked_back = wide_layout && d_ws && wide_layout->kw > 0 && !getenv("NDMPS_BACK_NARROW") &&
                            (w.kp > kMaxK || (h_n && wide_layout->wpart_stride > 0 && !getenv("NDMPS_ORTHO_NARROW")));
  SideFork fork;
  if (blocked_back) {
    NDMPS_TRY(side_fork(s, fork));
    hipLaunchKernelGGL(back_wide_t_kernel, dim3((unsigned)ndmps::ceil_div(std::max<int64_t>(n_max - 1, 1), kBwB), B), dim3(256), 0,
                       fork.side ? fork.side : s, (const TrdDesc*)desc, w, (double*)((char*)d_ws + wide_layout->off_wt),
                       wide_layout->wt_stride);
  }
  hipLaunchKernelGGL(trd_shift_kernel, dim3((unsigned)ndmps::ceil_div(batch, 64)), dim3(64), 0, s, desc, w, batch);
  // inverse iteration in column blocks (invit_block_width: one block of up to 128 vectors per matrix of a lockstep group,
  // narrow blocks on their own CUs for one or a few big matrices)
  const int cb = invit_block_width(batch, n_max, w.kp);
  const int kb = std::min<int>(w.kp, cb);
  hipLaunchKernelGGL(trd_invit_kernel, dim3((unsigned)ndmps::ceil_div(w.kp, cb), B), dim3(512),
                     std::max((size_t)n_max * 16, (size_t)2 * 4 * invit_rows_per_block(kb) * kb * 8), s, desc, w,
                     cb | ((getenv("NDMPS_INVIT_DBG") ? atoi(getenv("NDMPS_INVIT_DBG")) : 0) << 16));
  const bool wide_small = w.kp <= kMaxK && wide_layout && d_ws && h_n && wide_layout->kw > 0 && !getenv("NDMPS_ORTHO_NARROW");
  if (wide_small) {
    // one or two big matrices: Cholesky-QR across the chip with the rank read on the device (eig_wide.inc)
    char* base = (char*)d_ws;
    for (int b = 0; b < batch; ++b)
      NDMPS_TRY(wide_orthonormalise_auto(desc + b, w.Z + (int64_t)b * w.n_max * w.kp, (int)h_n[b], w.kp,
                                         (double*)(base + wide_layout->off_ws), (double*)(base + wide_layout->off_wlinv),
                                         (int)wide_layout->kw, base + wide_layout->off_wgram, wide_layout->wgram_bytes, s));
  } else if (w.kp > kMaxK) {
    // more than 128 vectors may be wanted: Cholesky-QR across the chip, matrix by matrix (eig_wide.inc)
    NDMPS_REQUIRE(wide_layout && d_ws && h_n && h_k && wide_layout->kw > 0, "wide eigenvector block without its workspace");
    char* base = (char*)d_ws;
    for (int b = 0; b < batch; ++b)
      NDMPS_TRY(wide_orthonormalise(desc + b, w.Z + (int64_t)b * w.n_max * w.kp, (int)h_n[b], (int)h_k[b], w.kp,
                                    (double*)(base + wide_layout->off_ws), (double*)(base + wide_layout->off_wlinv),
                                    (int)wide_layout->kw, base + wide_layout->off_wgram, wide_layout->wgram_bytes, s));
  } else if (k16 <= 64)
    hipLaunchKernelGGL(trd_ortho_kernel<true>, dim3(1, B), dim3(512), (size_t)2 * k16 * (k16 + 1) * 8, s, desc, w);
  else if (!getenv("NDMPS_ORTHO_COLUMNS"))  // k > 64: column blocks of 64 (block Gram-Schmidt + blocked Cholesky-QR)
    hipLaunchKernelGGL(trd_ortho_blocks_kernel, dim3(1, B), dim3(512), (size_t)3 * 64 * 65 * 8, s, desc, w);
  else
    hipLaunchKernelGGL(trd_ortho_kernel<false>, dim3(1, B), dim3(512), (size_t)k16 * (k16 + 1) * 8, s, desc, w);
  if (wide_small && wide_layout->wpart_stride > 0 && !getenv("NDMPS_BACK_NARROW")) {
    // one or two big matrices, at most 128 columns: rows dealt to the chip, one launch per block of 64 reflectors
    double* Tw = (double*)((char*)d_ws + wide_layout->off_wt);
    double* part = (double*)((char*)d_ws + wide_layout->off_wpart);
    const int G = n_max >= 2 ? (int)((n_max - 2) / kBwB + 1) : 0;
    const int chunks = (int)ndmps::ceil_div(n_max, kBrR);
    const int cols = std::max(kk, k_fill);
    NDMPS_TRY(side_join(s, fork));
    hipLaunchKernelGGL(back_rows_init_kernel, dim3(256, B), dim3(256), 0, s, (const TrdDesc*)desc, w, k_fill);
    const dim3 grid((unsigned)chunks, (unsigned)ndmps::ceil_div(cols, kBrC), B);
    for (int g = G; g >= 0; --g)  // launch g: apply block g (none at first), form block g - 1 (none at last)
      hipLaunchKernelGGL(back_rows_step_kernel, grid, dim3(256), 0, s, (const TrdDesc*)desc, w, (const double*)Tw,
                         wide_layout->wt_stride, part, wide_layout->wpart_stride, chunks, g < G ? g : -1, g - 1);
    hipLaunchKernelGGL(back_rows_sign_kernel, dim3((unsigned)ndmps::ceil_div(cols, 16), B), dim3(1024), 0, s, (const TrdDesc*)desc);
    ndmps::span_end(vec_span, s, ndmps::kSpanEigenVectors, G + 8, 0);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }
  if (w.kp > kMaxK && !getenv("NDMPS_BACK_NARROW")) {
    // many columns: the reflectors in blocks of 64 on the MFMA, one workgroup per 16 columns (eig_wide.inc)
    double* Tw = (double*)((char*)d_ws + wide_layout->off_wt);
    NDMPS_TRY(side_join(s, fork));
    hipLaunchKernelGGL(back_wide_kernel, dim3((unsigned)ndmps::ceil_div(kk, kBwC), B), dim3(64 * kBwWaves), 0, s, (const TrdDesc*)desc, w,
                       (const double*)Tw, wide_layout->wt_stride);
    ndmps::span_end(vec_span, s, ndmps::kSpanEigenVectors, 4, 0);
    NDMPS_LAUNCH_CHECK();
    return NDMPS_OK;
  }
  const int bw = route_load(w.A, band_width_for(n_max));
  if (bw) {  // eigenvectors of T -> eigenvectors of the band matrix: the bulge chase's reflectors, sweeps in reverse
    const dim3 grid((unsigned)ndmps::ceil_div(std::min<int64_t>(k16, w.kp), 32), B);
    if (bw == 2) hipLaunchKernelGGL(trd_back2_kernel<2>, grid, dim3(1024), (size_t)n_max * 32 * 8, s, desc, w, kk);
    else hipLaunchKernelGGL(trd_back2_kernel<4>, grid, dim3(1024), (size_t)n_max * 32 * 8, s, desc, w, kk);
  }
  const int cols = std::max(kk, k_fill);
  // rows per lane of the back-transform: n <= SEG * R; RB reflectors of SEG * R doubles per LDS block
  const int per32 = (int)ndmps::ceil_div(n_max, 32), per64 = (int)ndmps::ceil_div(n_max, 64);
#define NDMPS_BACK(SEG, R, RB, WYB)                                                                            \
  do {                                                                                                          \
    const int groups = (int)ndmps::ceil_div(std::max<int64_t>(n_max - 1, 1), WYB);                               \
    hipLaunchKernelGGL((trd_wy_kernel<WYB>), dim3(groups, B), dim3(64), 0, s, desc, w, w.Tw, w.t_stride);         \
    hipLaunchKernelGGL((trd_back_kernel<SEG, R, RB, WYB>), dim3(ndmps::ceil_div(cols, 256 / SEG), B), dim3(256), \
                       (size_t)2 * RB * SEG * R * sizeof(double), s, desc, w, k_fill, w.Tw, w.t_stride);         \
  } while (0)
  if (per32 <= 4) NDMPS_BACK(32, 4, 8, 4);
  else if (per32 <= 8) NDMPS_BACK(32, 8, 8, 4);
  else if (per32 <= 16) NDMPS_BACK(32, 16, 8, 4);
  else if (per32 <= 32) NDMPS_BACK(32, 32, 4, 2);
  else if (per64 <= 32) NDMPS_BACK(64, 32, 2, 2);
  else NDMPS_BACK(64, 64, 1, 1);
#undef NDMPS_BACK
  ndmps::span_end(vec_span, s, ndmps::kSpanEigenVectors, 4, 0);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
}  // namespace

// Phase 2 with the rank decided ON THE DEVICE from the eigenvalues of phase 1: k_b = number of singular values
// sqrt(w_i) above cutoff * sqrt(w_0), at least 1, at most k_cap.  Columns k_b .. k_cap-1 of V are zero-filled, so
// callers size everything by k_cap and never wait for the rank.  d_ranks[b] (device) receives k_b, d_spectra (may
// be NULL) the k_cap leading singular values at stride spectra_stride, d_status[b] (may be NULL) != 0 if the
// orthonormalisation of matrix b broke down.  Fully asynchronous on `stream`.
extern "C" int ndmps_syevd_topk_vectors_auto_f64(int batch, const int64_t* h_n, int64_t k_cap, double cutoff,
                                                 int* d_ranks, double* d_spectra, int64_t spectra_stride,
                                                 int* d_status, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  int64_t n_max = 0;
  NDMPS_TRY(trd_check_sizes(batch, h_n, n_max));
  NDMPS_REQUIRE(d_ranks, "NULL rank output");
  NDMPS_REQUIRE(k_cap >= 1 && k_cap <= kMaxK && cutoff >= 0.0, "k_cap=%lld outside [1, %d] or negative cutoff",
                (long long)k_cap, kMaxK);
  const TrdLayout l = trd_layout(n_max, batch, std::min(k_cap, n_max));
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevd_topk workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  TrdWork w = trd_work(l, d_ws);
  TrdDesc* desc = (TrdDesc*)((char*)d_ws + l.off_desc);
  const int kk = (int)std::min(k_cap, n_max);
  hipLaunchKernelGGL(trd_rank_kernel, dim3(batch), dim3(128), 0, s, desc, kk, cutoff, d_ranks, d_spectra, spectra_stride);
  NDMPS_LAUNCH_CHECK();
  NDMPS_TRY(trd_launch_vectors(batch, n_max, kk, kk, desc, w, s, &l, d_ws, h_n));
  if (d_status) {
    hipLaunchKernelGGL(trd_status_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, desc, batch, d_status);
    NDMPS_LAUNCH_CHECK();
  }
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
  NDMPS_REQUIRE(k_max >= 1 && k_max <= kMaxN, "k_max=%lld outside [1, %d]", (long long)k_max, kMaxN);
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
  for (int base = 0; base < batch; base += 256) {
    RankChunk chunk;
    const int count = std::min(256, batch - base);
    for (int t = 0; t < count; ++t) chunk.v[t] = (int)h_k[base + t];
    hipLaunchKernelGGL(trd_setk_kernel, dim3(1), dim3(256), 0, s, desc, chunk, base, count);
  }
  NDMPS_TRY(trd_launch_vectors(batch, n_max, kk, 0, desc, w, s, &l, d_ws, h_n, h_k));
  if (h_status) {
    std::vector<TrdDesc> host(batch);
    NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), desc, sizeof(TrdDesc) * batch, hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    for (int b = 0; b < batch; ++b) h_status[b] = host[b].status;
  }
  return NDMPS_OK;
}

// ---- the resident tridiagonalisation's way out (status 2: a team gave up waiting for its workgroups, see
// trd_team_kernel).  The reduction works on a copy of G, so phase 1 can simply be done again on the column launches,
// whose only synchronisation is the kernel boundary.
//
// ndmps_syevd_topk_recover_f64: call after ndmps_syevd_topk_values_f64 with the same batch / sizes / workspace when the
// host is about to synchronise anyway (it reads the eigenvalues): waits for `stream`, and if any matrix carries
// status 2 re-runs phase 1 for the whole batch without the resident launch (asynchronous again), counts the event and
// sets *h_recovered.  Callers that never synchronise between the phases (..._vectors_auto_f64) see status 2 in
// d_status at the end and repeat their sequence after ndmps_syevd_topk_set_team(0) (the sweep does: tt.hip).
extern "C" int ndmps_syevd_topk_recover_f64(int batch, const int64_t* h_n, int64_t k_max, void* d_ws, int64_t ws_bytes,
                                            int* h_recovered, ndmps_stream_t stream) {
  int64_t n_max = 0;
  NDMPS_TRY(trd_check_sizes(batch, h_n, n_max));
  NDMPS_REQUIRE(k_max >= 1 && k_max <= kMaxN, "k_max=%lld outside [1, %d]", (long long)k_max, kMaxN);
  const TrdLayout l = trd_layout(n_max, batch, std::min(k_max, n_max));
  if (d_ws == nullptr || ws_bytes < l.total) {
    ndmps::set_error("syevd_topk workspace too small: %lld < %lld", (long long)ws_bytes, (long long)l.total);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  TrdWork w = trd_work(l, d_ws);
  TrdDesc* desc = (TrdDesc*)((char*)d_ws + l.off_desc);
  std::vector<TrdDesc> host(batch);
  NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), desc, sizeof(TrdDesc) * batch, hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  bool aborted = false;
  for (int b = 0; b < batch; ++b) aborted = aborted || host[b].status == 2;
  if (h_recovered) *h_recovered = aborted ? 1 : 0;
  if (!aborted) return NDMPS_OK;
  g_team_fallbacks.fetch_add(1);
  hipLaunchKernelGGL(trd_clear_status_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, desc, batch, w.sync);
  g_team_off += 1;
  const int rc = trd_reduce_and_values(batch, h_n, n_max, k_max, l, w, desc, s);
  g_team_off -= 1;
  return rc;
}
// Per host thread: 0 switches the resident launch off (column launches for every order), 1 back on; returns the
// previous setting.  NDMPS_TRD_NO_TEAM=1 in the environment does the same for the whole process.
extern "C" int ndmps_syevd_topk_set_team(int enabled) {
  const int was = g_team_off ? 0 : 1;
  g_team_off = enabled ? 0 : 1;
  return was;
}
// Per host thread: 1 tells the solver that its caller keeps several batches in flight (see g_team_streamed), 0 that a call is
// alone on the GPU (the default); returns the previous setting.  Results are independent of it up to the last bits (the two
// block widths associate their column sums differently).
extern "C" int ndmps_syevd_topk_set_streamed(int streamed) {
  const int was = g_team_streamed;
  g_team_streamed = streamed ? 1 : 0;
  return was;
}
// number of times a resident launch was given up and its work redone on the column launches (whole process)
extern "C" int64_t ndmps_syevd_topk_team_fallbacks(void) { return g_team_fallbacks.load(); }
extern "C" int ndmps_syevd_topk_team_slots(int64_t order) {
  int slots = 0;
  if (trd_opt_in() != NDMPS_OK || team_slots(slots, order <= 512 ? 2 : order <= 1024 ? 4 : 8) != NDMPS_OK) return 0;
  return slots;
}
extern "C" int ndmps_syevd_topk_note_team_fallback(void) {  // for callers that redo their own sequence
  g_team_fallbacks.fetch_add(1);
  return NDMPS_OK;
}
// Test hook: the next `launches` resident launches are replaced by what an aborted one leaves behind (status 2 in
// every descriptor, the reduction not done), without the 3 s wait.
namespace {
__global__ void lane_sums_kernel(const double* __restrict__ in, double* __restrict__ out) {
  const int l = threadIdx.x;
  const double v = in[l];
  using namespace ndmps_lanes;
  out[0 * 64 + l] = sum_adjacent<2>(v);
  out[1 * 64 + l] = sum_adjacent<4>(v);
  out[2 * 64 + l] = sum_adjacent<8>(v);
  out[3 * 64 + l] = sum_adjacent<16>(v);
  out[4 * 64 + l] = sum_adjacent<32>(v);
  out[5 * 64 + l] = sum_adjacent<64>(v);
  out[6 * 64 + l] = sum_strided<1>(v);
  out[7 * 64 + l] = sum_strided<2>(v);
  out[8 * 64 + l] = sum_strided<4>(v);
  out[9 * 64 + l] = sum_strided<8>(v);
  out[10 * 64 + l] = sum_strided<16>(v);
  out[11 * 64 + l] = sum_strided<32>(v);
}
}  // namespace

// TEST HOOK: the cross-lane sums of lanes.h on one wave: out[12][64] = sums over 2, 4, ..., 64 adjacent lanes, then over
// the lanes with equal lane % 1, 2, ..., 32, of in[64]
extern "C" int ndmps_debug_lane_sums_f64(const double* d_in, double* d_out, void* stream) {
  if (!d_in || !d_out) return NDMPS_EINVAL;
  hipLaunchKernelGGL(lane_sums_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_in, d_out);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_debug_inject_team_abort(int launches) {
  g_inject_abort.store(launches > 0 ? launches : 0);
  return NDMPS_OK;
}
