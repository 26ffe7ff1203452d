/*
 * ndmps_hip.h -- C ABI of libndmps_hip.so, the MI355X (gfx950) implementation of the
 * NDMPS encode -> truncate -> reconstruct hot path of Alandroid/img-compression-mps.
 *
 * The reference has no FFI: its hot path is Python calling quimb / NumPy / SciPy
 * (SURVEY.md 8b).  Each entry point below therefore names the reference call it
 * replaces (file:line relative to /root/reference/src/imgcompressionmps/).
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory (HBM) owned by the caller; h_* is host.
 *   - all matrices are row-major; cores are (chi_i, d_i, chi_{i+1}) row-major.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).
 *   - every function returns 0 on success, a negative NDMPS_E* code otherwise;
 *     ndmps_last_error() returns a thread-local message for the last failure.
 *   - no function allocates device memory except ndmps_plan_create(); workspaces are
 *     caller-provided and sized by the *_layout / *_workspace_bytes queries.
 *   - functions that return data-dependent bond sizes synchronise `stream` internally.
 */
#ifndef NDMPS_HIP_H
#define NDMPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NDMPS_OK 0
#define NDMPS_EINVAL (-1)   /* bad argument (maps to ValueError) */
#define NDMPS_EHIP (-2)     /* HIP runtime failure (maps to RuntimeError) */
#define NDMPS_ENOCONV (-3)  /* eigen-solver did not converge */
#define NDMPS_EWORKSPACE (-4) /* workspace too small */
#define NDMPS_ETEAM (-5)    /* a resident tridiagonalisation gave up waiting for its workgroups (GPU shared with
                             * something that holds the compute units): the call's outputs are invalid; repeat it
                             * after ndmps_syevd_topk_set_team(0) (sweeps whose input is intact do so themselves) */

typedef void* ndmps_stream_t;
typedef struct ndmps_plan ndmps_plan_t;

int ndmps_version(void);
const char* ndmps_last_error(void);
/* number of HIP devices visible, or a negative code; never initialises a context */
int ndmps_device_count(void);

/* n HIP streams of the current device on pairwise different hardware queues (as far as the runtime
 * has queues: *n_independent tells how many are; the rest share).  The runtime binds a stream to one
 * of its few queues at first use and does not report which, so the binding is measured with a bounded
 * spin kernel.  For the concurrent volume groups of the batch caller
 * (reference: evaluation/benchmark.py:58-100 loops over independent tensors one by one; here each
 * group of the list runs on its own stream). */
int ndmps_streams_create(int n, void** h_streams, int* n_independent);
int ndmps_streams_destroy(int n, void* const* h_streams);

/* Launch spans for the roofline of bench.py: when enabled, the kernels bench.py prices bracket themselves with HIP
 * events on the stream they are launched on and file the span under a slot: 1 = column launches of the direct
 * eigen-solver (amount = algorithmic bytes), 2 = the resident tridiagonalisation (one span per launch sequence of a
 * batch; bytes), 3 = the Gram launches that fill the GPU for milliseconds and take a device-side turn (a lockstep
 * group's raw Gram; amount = FLOPS m n (n + 1) per matrix), 4 = every other batched Gram launch (flops).  collect
 * sums device ms, launches and amounts of a slot's spans recorded so far (and clears them); it waits for their
 * events.  No reference counterpart. */
int ndmps_profile_enable(int on);
int ndmps_profile_collect(int slot, double* h_ms, int64_t* h_launches, int64_t* h_bytes);

/* ---------------------------------------------------------------------------------
 * Index permutation ("reshape stage").
 * Replaces: utils/core.py:6-35,129-168 (gen_encoding_map, never materialised here),
 *           core/ndmps.py:66-71 (scatter  contracted[enc] = tensor.flatten()) and
 *           core/ndmps.py:144-148 (gather recovered = dense[enc]).
 * factor_arr is the (L, ndim) int64 array of utils/core.py:79-126 (get_factorlist).
 * --------------------------------------------------------------------------------- */
int ndmps_plan_create(ndmps_plan_t** out, int ndim, const int64_t* h_shape, int L,
                      const int64_t* h_factor_arr);
/* The same permutation onto the site-order tensor with its AXES REVERSED (C-order over d_{L-1} .. d_0): what a
 * left-to-right sweep (quimb's other from_dense convention, SURVEY a4's caveat) works on when it runs as a
 * right-to-left sweep of the mirrored chain.  Every plan entry point works on it; "site order" then means the reversed
 * order (ndmps_plan_split_offsets: n_cols = a product of LEADING site dimensions). */
int ndmps_plan_create_reversed(ndmps_plan_t** out, int ndim, const int64_t* h_shape, int L,
                               const int64_t* h_factor_arr);
int ndmps_plan_destroy(ndmps_plan_t* plan);
int64_t ndmps_plan_numel(const ndmps_plan_t* plan);
/* 1 if the LDS-tiled kernels are used for this plan, 0 for the generic gather */
int ndmps_plan_is_tiled(const ndmps_plan_t* plan);
/* Host-only emulation of the kernels' index arithmetic from the plan's tables (no GPU
 * needed; used by the CPU tests).  h_out[numel]: mode 0 = source offset read per site-order
 * position (generic encode), 1 = site-order offset read per C-order position (generic
 * decode), 2 = source offset stored per site-order position by the tiled kernels (-1 when
 * the plan is not tiled).  Returns 1/0 (tiled or not) or a negative error code. */
int ndmps_plan_emulate(int ndim, const int64_t* h_shape, int L, const int64_t* h_factor_arr,
                       int mode, int64_t* h_out);
int ndmps_plan_emulate_reversed(int ndim, const int64_t* h_shape, int L, const int64_t* h_factor_arr,
                                int mode, int64_t* h_out);
/* dst (C-order over site dims d_0..d_{L-1}) <- src (C-order over shape); bit-exact */
int ndmps_encode_permute(const ndmps_plan_t* plan, const void* d_src, void* d_dst,
                         int elem_bytes, ndmps_stream_t stream);
/* out (C-order over shape) <- dense (C-order over site dims); bit-exact */
int ndmps_decode_permute(const ndmps_plan_t* plan, const void* d_dense, void* d_out,
                         int elem_bytes, ndmps_stream_t stream);
/* The same permutations for the volumes of a lockstep group in ONE launch (thirty-two per launch): h_src / h_dst are
 * HOST arrays of `count` device pointers.  Replaces the per-volume Python loop around core/ndmps.py:66-71 and :144-147
 * (evaluation/benchmark.py:80-100); bit-exact like the single-volume calls. */
int ndmps_encode_permute_many(const ndmps_plan_t* plan, int count, const void* const* h_src, void* const* h_dst,
                              int elem_bytes, ndmps_stream_t stream);
int ndmps_decode_permute_many(const ndmps_plan_t* plan, int count, const void* const* h_dense, void* const* h_out,
                              int elem_bytes, ndmps_stream_t stream);
/* force the generic gather kernels (testing / A-B timing) */
int ndmps_encode_permute_generic(const ndmps_plan_t* plan, const void* d_src, void* d_dst,
                                 int elem_bytes, ndmps_stream_t stream);
int ndmps_decode_permute_generic(const ndmps_plan_t* plan, const void* d_dense, void* d_out,
                                 int elem_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Last-axis orthonormal DCT-II / DCT-III.
 * Replaces: scipy.fftpack.dct / idct at core/ndmps.py:62-63 and :152-153.
 * d_x, d_y: (rows, n) row-major fp32; d_basis: (n, n) fp32 scratch filled by
 * ndmps_dct_basis_f32 once per n (forward basis B[j][k] = s_k cos(pi (2j+1) k / 2n)).
 * --------------------------------------------------------------------------------- */
int ndmps_dct_basis_f32(float* d_basis, int64_t n, ndmps_stream_t stream);
int ndmps_dct_last_f32(const float* d_x, float* d_y, int64_t rows, int64_t n,
                       const float* d_basis, ndmps_stream_t stream);
int ndmps_idct_last_f32(const float* d_y, float* d_x, int64_t rows, int64_t n,
                        const float* d_basis, ndmps_stream_t stream);
/* The same transforms for the volumes of a lockstep group in ONE launch (thirty-two per launch): h_x / h_y are HOST
 * arrays of `count` device pointers, every volume (rows, n) row-major and distinct from its result.  Replaces the
 * per-volume Python loop around core/ndmps.py:62-63 and :152-153 (evaluation/benchmark.py:80-100); results equal the
 * single-volume calls bit for bit.  d_basis may be NULL when n is a power of two in [64, 1024] (the FFT route). */
int ndmps_dct_last_many_f32(int count, const float* const* h_x, float* const* h_y, int64_t rows, int64_t n,
                            const float* d_basis, ndmps_stream_t stream);
int ndmps_idct_last_many_f32(int count, const float* const* h_y, float* const* h_x, int64_t rows, int64_t n,
                             const float* d_basis, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Reductions that keep NDMPS state (core/ndmps.py:60-61 norm, :75,:80-82 boundary_list).
 * h_out / d_out as named; *_f32 read fp32 and accumulate in fp64.
 * --------------------------------------------------------------------------------- */
int ndmps_sumsq_f32(const float* d_x, int64_t n, double* h_out, void* d_ws, int64_t ws_bytes,
                    ndmps_stream_t stream);
int ndmps_minmax_f32(const float* d_x, int64_t n, float* h_min, float* h_max, void* d_ws,
                     int64_t ws_bytes, ndmps_stream_t stream);
/* min/max (and optionally the fp64 sum of squares) of `count` tensors with one launch and one
 * synchronisation (boundary_list of all cores, core/ndmps.py:80-82): h_ptrs / h_lens are host
 * arrays, h_out gets (min, max) pairs, h_sumsq (may be NULL) one double per tensor */
int64_t ndmps_minmax_many_workspace_bytes(int count);
int ndmps_minmax_many_f32(int count, const float* const* h_ptrs, const int64_t* h_lens,
                          float* h_out, double* h_sumsq, void* d_ws, int64_t ws_bytes,
                          ndmps_stream_t stream);

/* The same reduction for the cores of a lockstep group sitting in one arena (row b = volume b, core i at
 * h_offsets[i] with h_lens[i] elements; at most 64 cores), in two halves so that nothing waits in between:
 * _launch enqueues the kernel (asynchronous; tensor (b, i) is entry b * n_cores + i), _collect copies the partials
 * back, synchronises and folds them into (min, max) pairs and sums of squares.  core/ndmps.py:75-76 computes these
 * eagerly; here they are issued with the sweep and read when boundary_list / norm_value are asked for. */
int64_t ndmps_minmax_partials_bytes(int count);
int ndmps_minmax_arena_launch_f32(const float* d_base, int64_t row_stride, int batch, int n_cores,
                                  const int64_t* h_offsets, const int64_t* h_lens, double* d_partial,
                                  ndmps_stream_t stream);
int ndmps_minmax_collect(int count, const double* d_partial, float* h_out, double* h_sumsq, ndmps_stream_t stream);
int ndmps_scale_f32(float* d_x, int64_t n, double factor, ndmps_stream_t stream);
/* h_x[t][0 .. n) *= h_factor[t] for `count` fp32 tensors of n elements each in one launch per thirty-two tensors: the
 * division by the norm of core/ndmps.py:60-61 for a whole lockstep group (h_x, h_factor: HOST arrays). */
int ndmps_scale_many_f32(int count, float* const* h_x, int64_t n, const double* h_factor, ndmps_stream_t stream);
int64_t ndmps_reduce_workspace_bytes(void);

/* ---------------------------------------------------------------------------------
 * Dense building blocks (exported for tests and for INTEGRATION.md users).
 * --------------------------------------------------------------------------------- */
/* C(m,n) = op(A)(m,k) op(B)(k,n), fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 accumulate */
int ndmps_sgemm(int transA, int transB, int64_t m, int64_t n, int64_t k, const float* d_A,
                int64_t lda, const float* d_B, int64_t ldb, float* d_C, int64_t ldc,
                ndmps_stream_t stream);
/* same in fp64 (v_mfma_f64_16x16x4_f64) */
int ndmps_dgemm(int transA, int transB, int64_t m, int64_t n, int64_t k, const double* d_A,
                int64_t lda, const double* d_B, int64_t ldb, double* d_C, int64_t ldc,
                ndmps_stream_t stream);

/* `batch` (<= ndmps_gemm_batched_max()) products of one shape in ONE launch: h_A / h_B / h_C are HOST arrays of
 * device pointers (the reference multiplies volume by volume in a Python loop, evaluation/benchmark.py:73-100);
 * same tiles and arithmetic as ndmps_sgemm / ndmps_dgemm / ndmps_sgemm_indexed on each triple. */
int ndmps_gemm_batched_max(void);
int ndmps_sgemm_batched(int batch, int transA, int transB, int64_t m, int64_t n, int64_t k, const float* const* h_A,
                        int64_t lda, const float* const* h_B, int64_t ldb, float* const* h_C, int64_t ldc,
                        ndmps_stream_t stream);
int ndmps_dgemm_batched(int batch, int transA, int transB, int64_t m, int64_t n, int64_t k, const double* const* h_A,
                        int64_t lda, const double* const* h_B, int64_t ldb, double* const* h_C, int64_t ldc,
                        ndmps_stream_t stream);
int ndmps_sgemm_indexed_batched(int batch, int64_t m, int64_t n, int64_t k, const float* const* h_A, int64_t lda,
                                const int64_t* d_a_row, const int64_t* d_a_col, int a_vec4, const float* const* h_B,
                                int64_t ldb, float* const* h_C, int64_t ldc, const int64_t* d_c_row,
                                const int64_t* d_c_col, ndmps_stream_t stream);
/* G(n,n) fp64 = A^T A for A (m,n) fp32 row-major; products exact, fp64 accumulation */
int64_t ndmps_gram_workspace_bytes(int64_t m, int64_t n);
int ndmps_gram_f32(const float* d_A, int64_t m, int64_t n, int64_t lda, double* d_G,
                   void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
/* bf16 storage path (BASELINE config 5; the reference fixes its dtype at core/ndmps.py:56).
 * ndmps_gemm_bf16: C (m, n) = A (m, k) op(B), op(B) = B (k, n) for transB == 0 and B^T with B (n, k) for
 * transB == 1; A, B, C bf16 row-major, fp32 accumulation on v_mfma_f32_32x32x16_bf16.  d_ws:
 * ndmps_gemm_bf16_workspace_bytes(transB, n, k) bytes (a (k, n) right operand is transposed first). */
int64_t ndmps_gemm_bf16_workspace_bytes(int transB, int64_t n, int64_t k);
int ndmps_gemm_bf16(int transB, int64_t m, int64_t n, int64_t k, const void* d_A, int64_t lda,
                    const void* d_B, int64_t ldb, void* d_C, int64_t ldc, void* d_ws, int64_t ws_bytes,
                    ndmps_stream_t stream);
/* The same for `batch` matrices of one shape in ONE launch (m >= 256 and n >= 128, or 64 <= n < 128 with at least
 * two matrices: _workspace_bytes returns 0 for shapes without a group-wide launch; what a lockstep group of
 * volumes needs at a site of the sweep, core/ndmps.py:74 per volume in the reference): h_A is a HOST array of
 * device pointers, G of matrix b goes to d_G + b * stride_G.  _indexed: element (r, c) of matrix b is
 * h_base[b][d_row_off[r] + d_col_off[c]] (see ndmps_gram_indexed_f32). */
int64_t ndmps_gram_batched_workspace_bytes(int batch, int64_t m, int64_t n);
int ndmps_gram_batched_f32(int batch, const float* const* h_A, int64_t m, int64_t n, int64_t lda, double* d_G,
                           int64_t stride_G, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
int ndmps_gram_batched_bf16(int batch, const void* const* h_A, int64_t m, int64_t n, int64_t lda, double* d_G,
                            int64_t stride_G, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
int ndmps_gram_batched_indexed_f32(int batch, const float* const* h_base, int64_t m, int64_t n,
                                   const int64_t* d_row_off, const int64_t* d_col_off,
                                   const int32_t* d_col_perm, double* d_G, int64_t stride_G, void* d_ws,
                                   int64_t ws_bytes, ndmps_stream_t stream);

/* G = A^T A for a bf16 matrix A (products exact, fp64 accumulation); workspace as ndmps_gram_f32 */
int ndmps_gram_bf16(const void* d_A, int64_t m, int64_t n, int64_t lda, double* d_G, void* d_ws,
                    int64_t ws_bytes, ndmps_stream_t stream);
int ndmps_convert_bf16_to_f32(const void* d_x, int64_t n, float* d_y, ndmps_stream_t stream);
int ndmps_convert_f32_to_bf16(const float* d_x, int64_t n, void* d_y, ndmps_stream_t stream);
/* symmetric eigen-decomposition (block two-sided Jacobi, fp64): G = V diag(w) V^T,
 * w descending, eigenvectors in the COLUMNS of V, each with its largest-magnitude
 * component positive.  d_G is destroyed.  Synchronises the stream. */
int64_t ndmps_syevj_workspace_bytes(int64_t n);
int ndmps_syevj_f64(double* d_G, int64_t n, double* d_V, double* d_w, void* d_ws,
                    int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream);
/* batch of independent eigenproblems solved in lockstep (one launch sequence for all):
 * matrix b is d_G + b*stride_G, n_b x n_b (ld n_b); outputs likewise.  Same conventions. */
int64_t ndmps_syevj_batched_workspace_bytes(int64_t n_max, int batch);
int ndmps_syevj_batched_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n, double* d_V,
                            int64_t stride_V, double* d_w, int64_t stride_w, void* d_ws,
                            int64_t ws_bytes, int* h_sweeps, ndmps_stream_t stream);
/* same with an explicit convergence threshold: a sweep with no |a_pq| above rel_tol * max|a_ii|
 * ends the iteration (ndmps_syevj_*_f64 use 1e-15; the fp32 sweep uses kSweepEigTol, tt.hip) */
int ndmps_syevj_batched_tol_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n,
                                double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                double rel_tol, void* d_ws, int64_t ws_bytes, int* h_sweeps,
                                ndmps_stream_t stream);
/* two-phase form (used by the sweep): eigenvalues first -- the caller derives the kept rank of every
 * matrix from them -- then only the first h_k[b] eigenvectors (columns 0..k-1 of V).  Same arguments
 * in both calls; d_ws must stay untouched in between.  For batches of matrices up to 896 the solver
 * does not update V per step: it records every step's rotation blocks in d_ws and replays them on the
 * k wanted columns held in LDS. */
int ndmps_syevj_batched_values_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n,
                                   double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                   double rel_tol, void* d_ws, int64_t ws_bytes, int* h_sweeps,
                                   ndmps_stream_t stream);
int ndmps_syevj_batched_vectors_f64(int batch, double* d_G, int64_t stride_G, const int64_t* h_n,
                                    double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                    const int64_t* h_k, void* d_ws, int64_t ws_bytes,
                                    ndmps_stream_t stream);

/* Leading k eigenpairs by a direct method (Householder tridiagonalisation, multi-section on the Sturm
 * sequence, inverse iteration with a Cholesky-QR of the block, reflectors replayed on the k columns): the
 * path of the bond-capped sweep, where k = chi <= 128 of n = d chi <= 4096 are wanted (replaces the thin SVD
 * quimb's from_dense calls per site, core/ndmps.py:74).  Same conventions as ndmps_syevj_*: w descending,
 * eigenvector c in column c of V (ld n), largest-magnitude component positive.  Two phases like the Jacobi
 * pair above; both are asynchronous on `stream` (vectors synchronises only when h_status is given;
 * h_status[b] != 0: the orthonormalisation of matrix b broke down).
 * The two-phase calls take any k_max <= ndmps_syevd_topk_max_k_wide() (= every eigenpair of the largest order): the
 * exact sweeps and compress() want all eigenpairs above a cutoff (core/ndmps.py:74,104-106: dgesdd through quimb).
 * Above ndmps_syevd_topk_max_k() wanted vectors the eigenvectors of T are iterated in column blocks of 128 and
 * orthonormalised across the chip (Gram matrix on the fp64 MFMA, blocked Cholesky, one-launch triangular solve:
 * csrc/eig_wide.inc), and so are the vectors of one or two matrices of order >= 1024.  Tridiagonalisation: orders up to
 * 2048 whose teams fit the chip together stay resident in registers for all columns (one launch); above that, panel by
 * panel (csrc/eig_panel.inc) with the last 2048 / 1024 / 512 columns handed to the resident kernel. */
int64_t ndmps_syevd_topk_max_n(void);
int64_t ndmps_syevd_topk_max_k(void);
int64_t ndmps_syevd_topk_max_k_wide(void);
int64_t ndmps_syevd_topk_workspace_bytes(int64_t n_max, int batch, int64_t k_max);
/* byte offset, inside the workspace, of 16 int64 wall-clock marks (100 MHz) per matrix left by the vectors
 * phase (profiling aid, tools/trd_probe.py) */
int64_t ndmps_syevd_topk_stamps_offset(int64_t n_max, int batch, int64_t k_max);
/* Cholesky factor S = L L^T of a symmetric positive definite matrix (lower triangle read, L with zeros above written in
 * place; blocked, fp64 MFMA): the square root compress() needs of G2 = T2 T2^T (core/ndmps.py:104-106; quimb takes an LQ of
 * T2 there).  d_scratch: ndmps_potrf_scratch_elems(n) doubles.  Synchronises; *h_status = 1: not positive definite. */
int64_t ndmps_potrf_scratch_elems(int64_t n);
int ndmps_potrf_lower_f64(double* d_S, int64_t n, double* d_scratch, int* h_status, ndmps_stream_t stream);
int ndmps_syevd_topk_values_f64(int batch, const double* d_G, int64_t stride_G, const int64_t* h_n,
                                double* d_V, int64_t stride_V, double* d_w, int64_t stride_w,
                                int64_t k_max, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
int ndmps_syevd_topk_vectors_f64(int batch, const int64_t* h_n, const int64_t* h_k, int64_t k_max,
                                 void* d_ws, int64_t ws_bytes, int* h_status, ndmps_stream_t stream);
/* The resident tridiagonalisation (orders 129..2048, one launch for all columns of every matrix) makes the workgroups
 * of a matrix wait for each other; every wait is bounded (3 s) and a team that gave up leaves status 2.
 * _recover_f64: call after _values_f64 with the same batch / sizes / workspace where the host synchronises anyway:
 * waits for `stream`; if any matrix carries status 2, phase 1 is done again for the batch on the per-column / panel launches
 * (asynchronous), *h_recovered (may be NULL) = 1.  _set_team(0 / 1): resident launch off / on for the calling host
 * thread, returns the previous setting.  _team_fallbacks: how often a resident launch was given up and redone
 * (process-wide); _note_team_fallback: for callers that repeat a sequence of their own.
 * ndmps_debug_inject_team_abort(n): TEST HOOK -- the next n resident launches are replaced by what an aborted one
 * leaves behind (status 2, reduction not done), without the 3 s wait.
 * _team_slots(order): workgroups of the resident kernel that takes matrices of `order` (<= 512, <= 1024, <= 2048) the
 * current device keeps resident at once (occupancy x compute units); a batch whose teams -- order / 8 workgroups per
 * matrix -- exceed it takes another route (more launches, the panel-blocked reduction); 0 on error. */
int ndmps_syevd_topk_recover_f64(int batch, const int64_t* h_n, int64_t k_max, void* d_ws, int64_t ws_bytes,
                                 int* h_recovered, ndmps_stream_t stream);
int ndmps_syevd_topk_set_team(int enabled);
/* Per host thread: 1 = the caller keeps several batches in flight on different streams (core/batch.py lanes): batches of five
 * to eight order-512 matrices then use 32-column blocks -- a quarter of the workgroup slots and half a turn, so the
 * reductions of consecutive batches overlap -- instead of the 8-column blocks that are fastest for ONE batch on an empty GPU
 * (one to four matrices keep them: their narrow launch fits half the slots).
 * Returns the previous setting.  No counterpart in the reference (LAPACK's dgesdd inside quimb, core/ndmps.py:74). */
int ndmps_syevd_topk_set_streamed(int streamed);
int64_t ndmps_syevd_topk_team_fallbacks(void);
int ndmps_syevd_topk_team_slots(int64_t order);
int ndmps_syevd_topk_note_team_fallback(void);
int ndmps_debug_inject_team_abort(int launches);
/* TEST HOOK: the cross-lane sums the tridiagonalisation kernels fold with (csrc/lanes.h: DPP moves and permlane swaps in
 * place of ds_bpermute), on one wave: d_out[12][64] = the sums of d_in[64] over 2, 4, 8, 16, 32, 64 adjacent lanes, then
 * over the lanes with equal lane % 1, 2, 4, 8, 16, 32 -- every lane holds its group's sum. */
int ndmps_debug_lane_sums_f64(const double* d_in, double* d_out, ndmps_stream_t stream);
/* Phase 2 with the rank decided ON THE DEVICE from phase 1's eigenvalues: k_b = #{i < k_cap : sqrt(w_i) > cutoff
 * sqrt(w_0)}, at least 1.  Columns k_b .. k_cap-1 of V are zero-filled, so a caller sizes everything by k_cap and
 * never waits for the rank.  d_ranks[b] (device) receives k_b, d_spectra (may be NULL) the k_cap leading singular
 * values at stride spectra_stride, d_status (may be NULL) the breakdown flags.  Fully asynchronous. */
int ndmps_syevd_topk_vectors_auto_f64(int batch, const int64_t* h_n, int64_t k_cap, double cutoff,
                                      int* d_ranks, double* d_spectra, int64_t spectra_stride,
                                      int* d_status, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * MPS sweep: replaces quimb MatrixProductState.from_dense (core/ndmps.py:74).
 * Right->left, per site: unfold (prod_{j<i} d_j) x (d_i chi_{i+1}) -> SVD (Gram + Jacobi,
 * fp64 small-side) -> keep s_k > cutoff*s_0, at most max_bond (<=0: unlimited) -> V^T is
 * site i, U S is carried left.  Cores are written fp32 at d_cores + core_offsets[i].
 *
 * ndmps_tt_layout fills (all host arrays of L+1 entries):
 *   h_max_bonds[i]   upper bound of chi_i (chi_0 = chi_L = 1)
 *   h_core_offsets[i] element offset of core i in the arena; [L] = arena elements
 *   h_spec_offsets[i] offset of bond i's singular values in h_spectra; [L] = total
 *   *h_workspace_bytes device workspace needed by ndmps_tt_sweep_f32
 * --------------------------------------------------------------------------------- */
int ndmps_tt_layout(int L, const int64_t* h_dims, int64_t max_bond, int64_t* h_max_bonds,
                    int64_t* h_core_offsets, int64_t* h_spec_offsets,
                    int64_t* h_workspace_bytes);
/* d_dense: N fp32 in site order (output of ndmps_encode_permute); it is overwritten.
 * h_bonds_out: L+1 actual bonds; h_spectra (may be NULL): singular values per bond. */
int ndmps_tt_sweep_f32(float* d_dense, int L, const int64_t* h_dims, double cutoff,
                       int64_t max_bond, float* d_cores, const int64_t* h_core_offsets,
                       int64_t* h_bonds_out, double* h_spectra, const int64_t* h_spec_offsets,
                       void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);

/* Batch of independent volumes of the same shape (the reference loops over lists of tensors,
 * evaluation/benchmark.py:73-76): h_dense[b] / h_cores[b] are per-volume device pointers,
 * h_bonds_out is batch x (L+1), h_spectra batch x h_spec_offsets[L].  The volumes advance
 * through the sites in lockstep so each site's eigenproblems are one batched solve. */
/* 1 when the sweep for these site dims and bond cap decides the ranks on the device (every site on the direct
 * top-k solver; no host round trip between sites): cores are then written at the layout's offsets in PADDED shape
 * (max_bonds[i], d_i, max_bonds[i+1]) with zeros beyond the actual bonds reported in h_bonds_out, and the caller
 * slices them.  0: cores are compact (bonds[i], d_i, bonds[i+1]). */
int ndmps_tt_sweep_pads_cores(int L, const int64_t* h_dims, int64_t max_bond);
/* The fused fp32 sweep in two halves, for sweeps that decide their ranks on the device (ndmps_tt_sweep_pads_cores = 1,
 * L > 1): everything such a sweep needs from the host is known before it starts.
 *   ndmps_tt_sweep_batched_fused_begin_f32  ENQUEUES the whole sweep (arguments as ndmps_tt_sweep_batched_fused_f32) and the
 *       copies of ranks, solver status and kept singular values into the caller's PINNED host buffers
 *       (h_pinned_ranks: ndmps_tt_sweep_async_ints(batch, L) ints; h_pinned_spec: ndmps_tt_sweep_async_doubles(...)
 *       doubles, may be a dummy when that is 0); does not wait.  h_bonds_scratch: batch (L + 1) entries.
 *   ndmps_tt_sweep_finish  after the caller has synchronised with the stream (an event behind _begin): fills h_bonds_out
 *       and h_spectra like the one-call form, or returns the solver's error -- NDMPS_ETEAM when a resident
 *       tridiagonalisation gave up; nothing is repeated here: the volumes are intact, the caller redoes the batch with
 *       ndmps_tt_sweep_batched_fused_f32, which retries on the column launches.
 * Replaces the same MatrixProductState.from_dense (core/ndmps.py:74); the reference has no counterpart of the split (NumPy
 * is synchronous).  core/batch.py begins batch k + 1 before it reads batch k: the objects of k are built under k + 1. */
int64_t ndmps_tt_sweep_async_ints(int batch, int L);
int64_t ndmps_tt_sweep_async_doubles(int batch, int L, const int64_t* h_dims, int64_t max_bond);
int ndmps_tt_sweep_batched_fused_begin_f32(int batch, const float* const* h_volume, int L, const int64_t* h_dims,
                                           double cutoff, int64_t max_bond, float* const* h_cores,
                                           const int64_t* h_core_offsets, int64_t* h_bonds_scratch,
                                           const int64_t* d_row_off, const int64_t* d_row_off_sorted,
                                           const int32_t* d_row_order, const int64_t* d_col_off,
                                           const int32_t* d_col_perm, int64_t n_cols, void* d_ws, int64_t ws_bytes,
                                           int* h_pinned_ranks, double* h_pinned_spec, ndmps_stream_t stream);
int ndmps_tt_sweep_finish(int batch, int L, const int64_t* h_dims, int64_t max_bond, const int* h_pinned_ranks,
                          const double* h_pinned_spec, int64_t* h_bonds_out, double* h_spectra,
                          const int64_t* h_spec_offsets);
int64_t ndmps_tt_sweep_batched_workspace_bytes(int batch, int L, const int64_t* h_dims,
                                               int64_t max_bond);
int ndmps_tt_sweep_batched_f32(int batch, float* const* h_dense, int L, const int64_t* h_dims,
                               double cutoff, int64_t max_bond, float* const* h_cores,
                               const int64_t* h_core_offsets, int64_t* h_bonds_out,
                               double* h_spectra, const int64_t* h_spec_offsets, void* d_ws,
                               int64_t ws_bytes, ndmps_stream_t stream);

/* The fp32 sweep with the reshape stage fused in (core/ndmps.py:66-71 + :74): the raw Gram pass and the
 * projection of the merged trailing run read the C-order volumes through the index permutation; h_volume[b] is
 * left untouched and no site-order tensor is formed.  n_cols = ndmps_tt_merge_columns(L, dims, max_bond)
 * (0: not available for this shape / bond cap); tables from ndmps_plan_split_offsets(plan, n_cols, ...) with the
 * columns sorted by offset (d_col_off ascending and in aligned runs of four consecutive offsets, d_col_perm[c] =
 * site-order column of the c-th smallest offset). */
int64_t ndmps_tt_merge_columns(int L, const int64_t* h_dims, int64_t max_bond);
int ndmps_tt_sweep_batched_fused_f32(int batch, const float* const* h_volume, int L, const int64_t* h_dims,
                                     double cutoff, int64_t max_bond, float* const* h_cores,
                                     const int64_t* h_core_offsets, int64_t* h_bonds_out,
                                     double* h_spectra, const int64_t* h_spec_offsets,
                                     const int64_t* d_row_off, const int64_t* d_row_off_sorted,
                                     const int32_t* d_row_order, const int64_t* d_col_off,
                                     const int32_t* d_col_perm, int64_t n_cols, void* d_ws, int64_t ws_bytes,
                                     ndmps_stream_t stream);
/* d_row_off_sorted (may be NULL): the entries of d_row_off in ascending order.  A Gram matrix is a sum over rows, so
 * its kernels may visit them in any order; in ascending order of their offsets they read the volume front to back
 * (32 x 262144 x 64: 1.14 -> 0.74 ms).  d_row_order (may be NULL): d_row_off_sorted[s] = d_row_off[d_row_order[s]]; with
 * it the first projection of a 64-column merged run (bond cap 32) is the stream below, its result rows scattered to
 * their places; every other projection keeps d_row_off.
 * ndmps_sgemm_gathered64_stream_batched: C[b] (m x n, n = 32 or 64, leading dimension ldc) = A[b] W[b] with A[b] the
 * (m x 64) matrix h_A[b][d_row_sorted[s] + d_col_off[c]] (rows in ascending order of their offsets, columns in
 * aligned runs of four consecutive offsets), row s of the product stored as row d_row_order[s] of C[b]. */
int ndmps_sgemm_gathered64_stream_batched(int batch, int64_t m, int64_t n, const float* const* h_A,
                                          const int64_t* d_row_sorted, const int32_t* d_row_order,
                                          const int64_t* d_col_off, const float* const* h_B, int64_t ldb,
                                          float* const* h_C, int64_t ldc, ndmps_stream_t stream);
/* (ndmps_gram_indexed_f32:)
 * G = A^T A where element (r, c) of A is d_base[d_row_off[r] + d_col_off[c]] (wide path: n >= 64, m >= 256);
 * d_col_perm (may be NULL): entry (a, b) of the product is stored at G[d_col_perm[a]][d_col_perm[b]] -- the columns
 * were visited in the memory order of the volume, the result comes out in site order */
int ndmps_gram_indexed_f32(const float* d_base, int64_t m, int64_t n, const int64_t* d_row_off,
                           const int64_t* d_col_off, const int32_t* d_col_perm, double* d_G, void* d_ws,
                           int64_t ws_bytes, ndmps_stream_t stream);

/* The same sweep on bf16 storage: site-order tensors, carried matrices and cores are bf16 in HBM, Gram
 * matrices / eigen-decompositions / bases fp64, products fp32-accumulated on the bf16 MFMA.  Layout and
 * workspace queries are those of the fp32 sweep (offsets count elements; its workspace size is an upper bound). */
int ndmps_tt_sweep_batched_bf16(int batch, void* const* h_dense, int L, const int64_t* h_dims,
                                double cutoff, int64_t max_bond, void* const* h_cores,
                                const int64_t* h_core_offsets, int64_t* h_bonds_out,
                                double* h_spectra, const int64_t* h_spec_offsets, void* d_ws,
                                int64_t ws_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Bond truncation: replaces quimb tensor_compress_bond(t1, t2, cutoff, cutoff_mode="rel")
 * (core/ndmps.py:104-106; reduced=True, absorb="both").  t1 (chi_l, d1, chi),
 * t2 (chi, d2, chi_r) fp32.  New cores are written to d_new1 / d_new2 (sized for chi),
 * *h_new_chi receives the kept bond; h_s (may be NULL) receives chi singular values.
 * --------------------------------------------------------------------------------- */
int64_t ndmps_compress_bond_workspace_bytes(int64_t chi_l, int64_t d1, int64_t chi, int64_t d2,
                                            int64_t chi_r);
int ndmps_compress_bond_f32(const float* d_t1, const float* d_t2, int64_t chi_l, int64_t d1,
                            int64_t chi, int64_t d2, int64_t chi_r, double cutoff,
                            int64_t max_bond, float* d_new1, float* d_new2, int64_t* h_new_chi,
                            double* h_s, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Chain contraction: replaces `mps ^ ...` (core/ndmps.py:140), cumulative left->right.
 * h_cores: host array of L device pointers; h_bonds: L+1 bonds.  d_dense receives the
 * (d_0..d_{L-1}) tensor, N fp32.
 * --------------------------------------------------------------------------------- */
int64_t ndmps_chain_workspace_bytes(int L, const int64_t* h_dims, const int64_t* h_bonds);
int ndmps_chain_contract_f32(int L, const int64_t* h_dims, const int64_t* h_bonds,
                             const float* const* h_cores, float* d_dense, void* d_ws,
                             int64_t ws_bytes, ndmps_stream_t stream);

/* Chain contraction that writes the C-order VOLUME: the inverse index permutation (core/ndmps.py:144-148)
 * rides on the last product of the chain, every element goes from the MFMA accumulator to its voxel; the
 * site-order tensor is never written.  n_cols = ndmps_chain_tail_columns(L, dims) (0: this chain has no
 * pre-contracted tail, use ndmps_chain_contract_f32 + ndmps_decode_permute); tables from
 * ndmps_plan_split_offsets(plan, n_cols, row_off, col_off): d_col_off = col_off sorted ascending, d_col_perm[c] =
 * the site-order column with the c-th smallest offset. */
int64_t ndmps_chain_tail_columns(int L, const int64_t* h_dims);
int ndmps_plan_split_offsets(const ndmps_plan_t* plan, int64_t n_cols, int64_t* h_row_off,
                             int64_t* h_col_off);
int ndmps_chain_contract_scatter_f32(int L, const int64_t* h_dims, const int64_t* h_bonds,
                                     const float* const* h_cores, float* d_out,
                                     const int64_t* d_row_off, const int64_t* d_col_off,
                                     const int32_t* d_col_perm, int64_t n_cols, void* d_ws,
                                     int64_t ws_bytes, ndmps_stream_t stream);
/* The same for a list of MPS over the same sites (the reference reconstructs a list with a Python loop,
 * evaluation/benchmark.py:80-100): volume b has bonds h_bonds[b (L + 1) ..], cores h_cores[b L ..] and is written
 * to h_out[b].  MPS that share their bonds go through the chain together (one batched launch per stage, each in
 * its own slice of d_ws); otherwise the volumes are contracted in turn.  d_ws >=
 * ndmps_chain_batched_workspace_bytes(batch, L, dims, bonds). */
int64_t ndmps_chain_batched_workspace_bytes(int batch, int L, const int64_t* h_dims, const int64_t* h_bonds);
int ndmps_chain_contract_scatter_batched_f32(int batch, int L, const int64_t* h_dims, const int64_t* h_bonds,
                                             const float* const* h_cores, float* const* h_out,
                                             const int64_t* d_row_off, const int64_t* d_col_off,
                                             const int32_t* d_col_perm, int64_t n_cols, void* d_ws,
                                             int64_t ws_bytes, ndmps_stream_t stream);
/* C = A B with table-driven addressing of A and / or C: element (m, k) of A at d_A[d_a_row[m] + d_a_col[k]],
 * element (m, n) of C at d_C[d_c_row[m] + d_c_col[n]] (NULL pair: dense row-major).  a_vec4: d_a_col comes in
 * aligned runs of four consecutive offsets. */
int ndmps_sgemm_indexed(int64_t m, int64_t n, int64_t k, const float* d_A, int64_t lda,
                        const int64_t* d_a_row, const int64_t* d_a_col, int a_vec4, const float* d_B,
                        int64_t ldb, float* d_C, int64_t ldc, const int64_t* d_c_row,
                        const int64_t* d_c_col, ndmps_stream_t stream);
/* bf16 cores in, bf16 tensor out, fp32 accumulation (v_mfma_f32_32x32x16_bf16); workspace:
 * ndmps_chain_workspace_bytes (the fp32 size covers the bf16 intermediates and the transposed operands) */
int ndmps_chain_contract_bf16(int L, const int64_t* h_dims, const int64_t* h_bonds,
                              const void* const* h_cores, void* d_dense, void* d_ws,
                              int64_t ws_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Overlap: replaces `mps @ mps` (core/ndmps.py:76,86; utils/metrics.py:160), real data,
 * no conjugation, fp64 transfer matrices.  Synchronises the stream.
 * --------------------------------------------------------------------------------- */
int64_t ndmps_overlap_workspace_bytes(int L, const int64_t* h_dims, const int64_t* h_bonds_a,
                                      const int64_t* h_bonds_b);
int ndmps_overlap_f32(int L, const int64_t* h_dims, const int64_t* h_bonds_a,
                      const float* const* h_cores_a, const int64_t* h_bonds_b,
                      const float* const* h_cores_b, double* h_out, void* d_ws,
                      int64_t ws_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Core quantisation: utils/filetools.py:20-39 scale_to_dtype/scale_back
 * with the reference's truncating cast.  bits = 8 or 16.
 * --------------------------------------------------------------------------------- */
int ndmps_quantize_f32(const float* d_x, int64_t n, float lo, float hi, int bits, void* d_q,
                       ndmps_stream_t stream);
int ndmps_dequantize_f32(const void* d_q, int64_t n, float lo, float hi, int bits, float* d_x,
                         ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * Quality metrics on device-resident volumes, semantics of utils/metrics.py:
 * ndmps_ssim_f32 = compute_ssim_by_dim(a, b) (metrics.py:108-129; 2-D / 3-D / 4-D, uniform window,
 * per-slice joint data_range, second argument clipped at 0), ndmps_psnr_f32 = compute_psnr(a, b)
 * (metrics.py:132-146).  fp32 inputs, fp64 arithmetic.  Both synchronise the stream.
 * --------------------------------------------------------------------------------- */
int64_t ndmps_ssim_workspace_bytes(int ndim, const int64_t* h_shape);
int ndmps_ssim_f32(const float* d_a, const float* d_b, int ndim, const int64_t* h_shape,
                   double* h_out, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
/* ssim_3d_axis(a, b, axis) (utils/metrics.py:35-65): the SSIM of every 2-D slice along `axis` (0, 1, 2) of a 3-D
 * volume, h_out[shape[axis]]; the 3-D value of ndmps_ssim_f32 is the mean over the axes of the means of these lists. */
int64_t ndmps_ssim_slices_workspace_bytes(const int64_t* h_shape, int axis);
int ndmps_ssim_slices_f32(const float* d_a, const float* d_b, const int64_t* h_shape, int axis, double* h_out,
                          void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
int64_t ndmps_psnr_workspace_bytes(void);
int ndmps_psnr_f32(const float* d_a, const float* d_b, int64_t n, double* h_out, void* d_ws,
                   int64_t ws_bytes, ndmps_stream_t stream);

/* ---------------------------------------------------------------------------------
 * fp64 storage -- the reference's own element type (core/ndmps.py:56 `tensor.astype(np.float64)`): volume, carried
 * matrices and cores are fp64 in HBM and every product runs on the fp64 MFMA, so that the reference's own
 * tolerances hold (tests/core/test_ndmps.py:35-38 round trip atol 1e-10; :41-44, :63-66 norm rel 1e-12).  Same
 * argument meaning as the _f32 twins; layout offsets (ndmps_tt_layout) are in elements and shared.  Singular values
 * come from fp64 Gram matrices: the relative cutoff of the sweep is clamped below at 1e-8 (sqrt(eps) s_0).
 *   ndmps_gram_f64                  G = A^T A, A fp64                    (core/ndmps.py:74, the SVD of from_dense)
 *   ndmps_tt_sweep_batched_f64      MatrixProductState.from_dense        (core/ndmps.py:74)
 *   ndmps_compress_bond_f64         tensor_compress_bond                 (core/ndmps.py:104-106)
 *   ndmps_chain_contract_f64        mps ^ ...                            (core/ndmps.py:140)
 *   ndmps_overlap_f64               mps @ mps                            (core/ndmps.py:76,86)
 *   ndmps_sumsq_f64 / _scale_f64    tensor /= np.linalg.norm(tensor)     (core/ndmps.py:60-61)
 *   ndmps_minmax_many_f64           boundary_list                        (core/ndmps.py:75,80-82)
 *   ndmps_dct_basis_f64 / _dct_last_f64 / _idct_last_f64  scipy dct / idct, norm="ortho" (core/ndmps.py:62-63,152-153)
 *   ndmps_quantize_f64 / _dequantize_f64  scale_to_dtype / scale_back    (utils/filetools.py:20-39)
 * --------------------------------------------------------------------------------- */
int64_t ndmps_gram_f64_workspace_bytes(int64_t m, int64_t n);
int ndmps_gram_f64(const double* d_A, int64_t m, int64_t n, int64_t lda, double* d_G, void* d_ws,
                   int64_t ws_bytes, ndmps_stream_t stream);
int64_t ndmps_tt_sweep_batched_workspace_bytes_f64(int batch, int L, const int64_t* h_dims, int64_t max_bond);
int ndmps_tt_sweep_batched_f64(int batch, double* const* h_dense, int L, const int64_t* h_dims,
                               double cutoff, int64_t max_bond, double* const* h_cores,
                               const int64_t* h_core_offsets, int64_t* h_bonds_out, double* h_spectra,
                               const int64_t* h_spec_offsets, void* d_ws, int64_t ws_bytes,
                               ndmps_stream_t stream);
int ndmps_compress_bond_f64(const double* d_t1, const double* d_t2, int64_t chi_l, int64_t d1,
                            int64_t chi, int64_t d2, int64_t chi_r, double cutoff, int64_t max_bond,
                            double* d_new1, double* d_new2, int64_t* h_new_chi, double* h_s, void* d_ws,
                            int64_t ws_bytes, ndmps_stream_t stream);
int64_t ndmps_chain_workspace_bytes_f64(int L, const int64_t* h_dims, const int64_t* h_bonds);
int ndmps_chain_contract_f64(int L, const int64_t* h_dims, const int64_t* h_bonds,
                             const double* const* h_cores, double* d_dense, void* d_ws, int64_t ws_bytes,
                             ndmps_stream_t stream);
int ndmps_overlap_f64(int L, const int64_t* h_dims, const int64_t* h_bonds_a,
                      const double* const* h_cores_a, const int64_t* h_bonds_b,
                      const double* const* h_cores_b, double* h_out, void* d_ws, int64_t ws_bytes,
                      ndmps_stream_t stream);
int ndmps_sumsq_f64(const double* d_x, int64_t n, double* h_out, void* d_ws, int64_t ws_bytes,
                    ndmps_stream_t stream);
int ndmps_scale_f64(double* d_x, int64_t n, double factor, ndmps_stream_t stream);
/* h_out: 2 * count doubles (min, max); h_sumsq may be NULL; workspace: ndmps_minmax_many_workspace_bytes */
int ndmps_minmax_many_f64(int count, const double* const* h_ptrs, const int64_t* h_lens, double* h_out,
                          double* h_sumsq, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream);
int ndmps_dct_basis_f64(double* d_basis, int64_t n, ndmps_stream_t stream);
int ndmps_dct_last_f64(const double* d_x, double* d_y, int64_t rows, int64_t n, const double* d_basis,
                       ndmps_stream_t stream);
int ndmps_idct_last_f64(const double* d_y, double* d_x, int64_t rows, int64_t n, const double* d_basis,
                        ndmps_stream_t stream);
int ndmps_quantize_f64(const double* d_x, int64_t n, double lo, double hi, int bits, void* d_q,
                       ndmps_stream_t stream);
int ndmps_dequantize_f64(const void* d_q, int64_t n, double lo, double hi, int bits, double* d_x,
                         ndmps_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NDMPS_HIP_H */
