// Streaming kernels that keep NDMPS state and the storage estimate:
//   sum of squares (norm=True divides by ||x||_2, core/ndmps.py:60-61),
//   min/max per core (boundary_list, core/ndmps.py:75, :80-82),
//   in-place scale, last-axis DCT basis, and the uint8/uint16 core quantisation of
//   utils/filetools.py:20-39.  All HBM-bound, 16-byte loads where aligned.
#include <float.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include <map>
#include <mutex>
#include <utility>

#include "common.h"

namespace {

constexpr int kRedBlocks = 1024;

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_down(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
  return v;
}

// stage 1: per-block partials (fixed grid -> deterministic), stage 2: one block folds them
template <typename T>
__global__ void __launch_bounds__(256)
sumsq_partial_kernel(const T* __restrict__ x, int64_t n, double* __restrict__ partial) {
  __shared__ double red[4];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const double v = (double)x[i];
    acc += v * v;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) sum_final_kernel(const double* __restrict__ partial, int count,
                                                        double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) acc += partial[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void __launch_bounds__(256)
minmax_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float rmin[4], rmax[4];
  float lo = FLT_MAX, hi = -FLT_MAX;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if ((threadIdx.x & 63) == 0) {
    rmin[threadIdx.x >> 6] = lo;
    rmax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
    partial[2 * blockIdx.x + 1] = fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
  }
}

__global__ void __launch_bounds__(256) minmax_final_kernel(const float* __restrict__ partial, int count,
                                                           float* __restrict__ out) {
  __shared__ float rmin[256], rmax[256];
  float lo = FLT_MAX, hi = -FLT_MAX;
  for (int i = threadIdx.x; i < count; i += 256) {
    lo = fminf(lo, partial[2 * i]);
    hi = fmaxf(hi, partial[2 * i + 1]);
  }
  rmin[threadIdx.x] = lo;
  rmax[threadIdx.x] = hi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      rmin[threadIdx.x] = fminf(rmin[threadIdx.x], rmin[threadIdx.x + s]);
      rmax[threadIdx.x] = fmaxf(rmax[threadIdx.x], rmax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = rmin[0];
    out[1] = rmax[0];
  }
}

// Pointer / length tables of a many-tensor reduction, handed over as KERNEL ARGUMENTS (32 entries per launch) like the
// rank tables of the sweep: no hipMemcpyAsync from pageable host memory (which stages through a pinned buffer and may
// wait for the stream) on the way to a launch whose result the caller reads back at once.
constexpr int kManyChunk = 32;
struct ManyChunk {
  const void* ptr[kManyChunk];
  int64_t len[kManyChunk];
};
__global__ void many_set_kernel(const void** __restrict__ d_ptrs, int64_t* __restrict__ d_lens, ManyChunk c, int base, int n) {
  const int t = threadIdx.x;
  if (t < n) {
    d_ptrs[base + t] = c.ptr[t];
    d_lens[base + t] = c.len[t];
  }
}
static void many_upload(const void** d_ptrs, int64_t* d_lens, const void* const* h_ptrs, const int64_t* h_lens, int count,
                        hipStream_t s) {
  for (int base = 0; base < count; base += kManyChunk) {
    ManyChunk c;
    const int n = std::min(kManyChunk, count - base);
    for (int t = 0; t < kManyChunk; ++t) {
      c.ptr[t] = t < n ? h_ptrs[base + t] : nullptr;
      c.len[t] = t < n ? h_lens[base + t] : 0;
    }
    hipLaunchKernelGGL(many_set_kernel, dim3(1), dim3(kManyChunk), 0, s, d_ptrs, d_lens, c, base, n);
  }
}

// many small tensors at once: blockIdx.y = tensor, blockIdx.x = slice; partial (min, max, sum of
// squares in fp64) per slice, as three doubles
__global__ void __launch_bounds__(256)
minmax_many_kernel(const float* const* __restrict__ ptrs, const int64_t* __restrict__ lens,
                   double* __restrict__ partial) {
  __shared__ float rmin[4], rmax[4];
  __shared__ double rsum[4];
  const float* x = ptrs[blockIdx.y];
  const int64_t n = lens[blockIdx.y];
  float lo = FLT_MAX, hi = -FLT_MAX;
  double ss = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
    ss += (double)v * (double)v;
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) {
    rmin[threadIdx.x >> 6] = lo;
    rmax[threadIdx.x >> 6] = hi;
    rsum[threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = (double)fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
    o[1] = (double)fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
    o[2] = (rsum[0] + rsum[1]) + (rsum[2] + rsum[3]);
  }
}

template <typename T>
__global__ void __launch_bounds__(256) scale_kernel(T* __restrict__ x, int64_t n, double factor) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    x[i] = (T)((double)x[i] * factor);
}

// fp64 tensors: (min, max, sum of squares) per slice, as three doubles (twin of minmax_many_kernel)
__global__ void __launch_bounds__(256)
minmax_many_f64_kernel(const double* const* __restrict__ ptrs, const int64_t* __restrict__ lens,
                       double* __restrict__ partial) {
  __shared__ double rmin[4], rmax[4], rsum[4];
  const double* x = ptrs[blockIdx.y];
  const int64_t n = lens[blockIdx.y];
  double lo = DBL_MAX, hi = -DBL_MAX, ss = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const double v = x[i];
    lo = fmin(lo, v);
    hi = fmax(hi, v);
    ss += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fmin(lo, __shfl_down(lo, off, 64));
    hi = fmax(hi, __shfl_down(hi, off, 64));
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) {
    rmin[threadIdx.x >> 6] = lo;
    rmax[threadIdx.x >> 6] = hi;
    rsum[threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = fmin(fmin(rmin[0], rmin[1]), fmin(rmin[2], rmin[3]));
    o[1] = fmax(fmax(rmax[0], rmax[1]), fmax(rmax[2], rmax[3]));
    o[2] = (rsum[0] + rsum[1]) + (rsum[2] + rsum[3]);
  }
}

// forward basis B[j][k] = s_k cos(pi (2j+1) k / (2n)); y = x B is the orthonormal DCT-II
template <typename T>
__global__ void __launch_bounds__(256) dct_basis_kernel(T* __restrict__ basis, int64_t n) {
  const int64_t total = n * n;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += stride) {
    const int64_t j = e / n, k = e % n;
    const int64_t num = ((2 * j + 1) * k) % (4 * n);  // angle = pi * num / (2n), period 4n
    const double c = cospi((double)num / (double)(2 * n));
    const double sk = (k == 0) ? sqrt(1.0 / (double)n) : sqrt(2.0 / (double)n);
    basis[e] = (T)(sk * c);
  }
}

template <typename Q, typename T = float>
__global__ void __launch_bounds__(256)
quantize_kernel(const T* __restrict__ x, int64_t n, double lo, double span, double qmax,
                Q* __restrict__ q) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    // (x - min) / max(x - min) * iinfo.max, then a truncating cast (filetools.py:24-26)
    const double u = ((double)x[i] - lo) / span;
    q[i] = (Q)(u * qmax);
  }
}

template <typename Q, typename T = float>
__global__ void __launch_bounds__(256)
dequantize_kernel(const Q* __restrict__ q, int64_t n, double lo, double span, double qmax,
                  T* __restrict__ x) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
#pragma clang fp contract(off)  // filetools.py:38-39 multiplies, then adds: two roundings, not one FMA
    const double unit = (double)q[i] / qmax;
    const double scaled = unit * span;
    x[i] = (T)(scaled + lo);
  }
}

// ---------------------------------------------------------------------------------------------- DCT by FFT
// Orthonormal DCT-II / DCT-III of rows whose length N is a power of two (64 .. 1024) in O(N log N): the basis product
// above costs 2 N flops per element (1.4 ms per 512^3 volume on the fp32 MFMA, the largest item of BASELINE's
// config 3), this one is bound by reading and writing the rows once.  Per row, one wave:
//   v[n] = x[2n], v[N-1-n] = x[2n+1];  z[n] = v[2n] + i v[2n+1] (n < M = N/2);  Z = FFT_M(z) (Stockham, radix 4 with
//   a closing radix-2 stage when log2 M is odd, ping-pong in LDS);  E = (Z[k] + conj Z[M-k]) / 2,
//   O = -i (Z[k] - conj Z[M-k]) / 2,  V[k] = E + W_N^k O;  t = exp(-i pi k / 2N) V[k]:  X[k] = s_k Re t,
//   X[N-k] = -s_k Im t.   The inverse runs the same steps backwards.  CPU prototype against SciPy: 1e-15 in fp64.
// Twiddles are computed once per workgroup in fp64 and rounded to fp32.
struct Cf {
  float re, im;
};
__device__ __forceinline__ Cf cmul(Cf a, Cf b) { return Cf{a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ Cf cadd(Cf a, Cf b) { return Cf{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ Cf csub(Cf a, Cf b) { return Cf{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ Cf cconj(Cf a) { return Cf{a.re, -a.im}; }
__device__ __forceinline__ Cf cmuli(Cf a) { return Cf{-a.im, a.re}; }   // i a
__device__ __forceinline__ Cf cmulni(Cf a) { return Cf{a.im, -a.re}; }  // -i a

__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// NN: the transform length as a compile-time constant (0: taken from the argument): stage count, strides and loop trip
// counts fold into the instruction stream -- the kernel is bound by the instructions a wave issues per row, not by LDS
// or HBM bandwidth (262144 rows of 512 in 378 us = 2.7 rows per us and CU against ~600 VALU instructions per row)
// The volumes of one launch (a lockstep group goes in chunks of kDctVols): `rows` counts the rows of all of them, row r
// belongs to volume r / rows_per_vol.
constexpr int kDctVols = 32;
struct DctVols {
  const float* in[kDctVols];
  float* out[kDctVols];
};
template <bool INVERSE, int NN>
__global__ void __launch_bounds__(256) dct_fft_kernel(DctVols dv, int64_t rows, int64_t rows_per_vol, int N_arg) {
  extern __shared__ __attribute__((aligned(16))) float fft_lds[];
  const int N = NN ? NN : N_arg;
  const int M = N / 2;
  Cf* twM = reinterpret_cast<Cf*>(fft_lds);  // exp(-2 pi i t / M), t < M
  Cf* wN = twM + M;                          // exp(-2 pi i k / N), k <= M
  Cf* wQ = wN + (M + 1);                     // exp(-i pi k / 2N),  k <= M
  float* wave_base = reinterpret_cast<float*>(wQ + (M + 1));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* rowbuf = wave_base + (size_t)wave * (N + 4 * M);  // N floats
  Cf* bufA = reinterpret_cast<Cf*>(rowbuf + N);
  Cf* bufB = bufA + M;
  for (int t = tid; t < M; t += 256) {
    double sn, cs;
    sincospi(-2.0 * (double)t / (double)M, &sn, &cs);
    twM[t] = Cf{(float)cs, (float)sn};
  }
  for (int k = tid; k <= M; k += 256) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)N, &sn, &cs);
    wN[k] = Cf{(float)cs, (float)sn};
    sincospi(-(double)k / (double)(2 * N), &sn, &cs);
    wQ[k] = Cf{(float)cs, (float)sn};
  }
  __syncthreads();
  const float s0 = (float)sqrt(1.0 / (double)N), s1 = (float)sqrt(2.0 / (double)N);
  // from here on every wave works on LDS buffers of its own, row after row: ordering inside the wave is all that is
  // needed between a stage's writes and the next stage's reads (LDS operations of one wave complete in order)
  const int64_t per_round = (int64_t)gridDim.x * 4;
  // the next row of the wave is fetched while the current one is transformed (N <= 1024: four float4 per lane)
  float4 nx[4] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f),
                  make_float4(0.f, 0.f, 0.f, 0.f)};
  auto fetch_row = [&](int64_t row) {
    if (row >= rows) return;  // wave-uniform
    const int64_t vol = row / rows_per_vol;
    const float4* in4 = reinterpret_cast<const float4*>(dv.in[vol] + (row - vol * rows_per_vol) * N);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = lane + 64 * t;
      if (i < N / 4) nx[t] = in4[i];
    }
  };
  fetch_row((int64_t)blockIdx.x * 4 + wave);
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += per_round) {
    constexpr bool on = true;
    Cf* a = bufA;
    Cf* b = bufB;
    if (!INVERSE) {
      // Makhoul's reordering straight from the registers: elements 4i .. 4i + 3 are v[2i] = x[4i], v[2i + 1] = x[4i + 2] and,
      // from the far end, v[N - 1 - 2i] = x[4i + 1], v[N - 2 - 2i] = x[4i + 3]; packed as z[n] = v[2n] + i v[2n + 1]
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        if (i < N / 4) {
          a[i] = Cf{nx[t].x, nx[t].z};
          a[M - 1 - i] = Cf{nx[t].w, nx[t].y};
        }
      }
    } else {
      float4* rb4 = reinterpret_cast<float4*>(rowbuf);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = lane + 64 * t;
        if (i < N / 4) rb4[i] = nx[t];
      }
    }
    fetch_row(row + per_round);
    wave_lds_sync();
    if (on) {
      if (!INVERSE) {
      } else {
        // V[k] = conj(wQ[k]) (X[k] - i X[N-k]) / s;  Z[k] = E + i O,  E = (V[k] + conj V[M-k]) / 2,
        // O = conj(W_N^k) (V[k] - conj V[M-k]) / 2 -- and Z[M-k] = conj(E) + i conj(O) (W_N^{M-k} = -conj W_N^k): one
        // pair (V[k], V[M-k]) serves both entries, k = 0 .. M / 2
#pragma unroll
        for (int k = lane; k <= M / 2; k += 64) {
          auto vk = [&](int q) {
            const float xr = rowbuf[q] * (q == 0 ? 1.0f / s0 : 1.0f / s1);
            const float xi = q == 0 ? 0.0f : rowbuf[N - q] * (1.0f / s1);
            return cmul(cconj(wQ[q]), Cf{xr, -xi});
          };
          const Cf v1 = vk(k), v2 = cconj(vk(M - k));
          const Cf e = Cf{0.5f * (v1.re + v2.re), 0.5f * (v1.im + v2.im)};
          const Cf d = Cf{0.5f * (v1.re - v2.re), 0.5f * (v1.im - v2.im)};
          const Cf o = cmul(cconj(wN[k]), d);
          a[k] = cadd(e, cmuli(o));
          if (k >= 1 && 2 * k != M) a[M - k] = cadd(cconj(e), cmuli(cconj(o)));
        }
      }
    }
    wave_lds_sync();
    const int lm = 31 - __builtin_clz(M);  // M = 2^lm
#pragma unroll
    for (int stage = 0; stage < (lm + 1) / 2; ++stage) {  // radix-4 stages, a radix-2 one at the end of an odd lm
      const int ls = 2 * stage;          // Ns = 2^ls: length of the finished sub-transforms
      const int Ns = 1 << ls;
      const int lr = (lm - ls) >= 2 ? 2 : 1, R = 1 << lr;
      if (on) {
        const int span = M >> lr;
#pragma unroll
        for (int j = lane; j < span; j += 64) {
          const int k = j & (Ns - 1);
          const int j0 = ((j >> ls) << (ls + lr)) + k;
          const int tstep = k << (lm - lr - ls);
          if (R == 4) {
            Cf v0 = a[j], v1 = a[j + span], v2 = a[j + 2 * span], v3 = a[j + 3 * span];
            Cf t1 = twM[tstep], t2 = twM[2 * tstep], t3 = twM[3 * tstep];
            if (INVERSE) {
              t1 = cconj(t1);
              t2 = cconj(t2);
              t3 = cconj(t3);
            }
            v1 = cmul(v1, t1);
            v2 = cmul(v2, t2);
            v3 = cmul(v3, t3);
            const Cf s02 = cadd(v0, v2), d02 = csub(v0, v2), s13 = cadd(v1, v3), d13 = csub(v1, v3);
            const Cf r13 = INVERSE ? cmuli(d13) : cmulni(d13);  // -i (v1 - v3) forward, +i inverse
            b[j0] = cadd(s02, s13);
            b[j0 + Ns] = cadd(d02, r13);
            b[j0 + 2 * Ns] = csub(s02, s13);
            b[j0 + 3 * Ns] = csub(d02, r13);
          } else {
            Cf v0 = a[j], v1 = a[j + span];
            Cf t1 = twM[tstep];
            if (INVERSE) t1 = cconj(t1);
            v1 = cmul(v1, t1);
            b[j0] = cadd(v0, v1);
            b[j0 + Ns] = csub(v0, v1);
          }
        }
      }
      wave_lds_sync();
      Cf* sw = a;
      a = b;
      b = sw;
    }
    if (on) {
      if (!INVERSE) {
        // v[k] = E + g, g = -i W_N^k D with E, D = (z[k] +- conj z[M-k]) / 2, and v[M-k] = conj(E - g): one pair serves
        // k and M - k (z[M] := z[0]), k = 0 .. M / 2; X[k] = Re(wQ[k] v[k]) s, X[N-k] = -Im(wQ[k] v[k]) s
#pragma unroll
        for (int k = lane; k <= M / 2; k += 64) {
          const Cf zk = a[k], zc = cconj(a[k == 0 ? 0 : M - k]);
          const Cf e = Cf{0.5f * (zk.re + zc.re), 0.5f * (zk.im + zc.im)};
          const Cf d = Cf{0.5f * (zk.re - zc.re), 0.5f * (zk.im - zc.im)};
          const Cf g = cmul(wN[k], cmulni(d));
          const Cf t1 = cmul(wQ[k], cadd(e, g));
          rowbuf[k] = t1.re * (k == 0 ? s0 : s1);
          if (k >= 1) rowbuf[N - k] = -t1.im * s1;
          if (2 * k != M) {
            const Cf t2 = cmul(wQ[M - k], cconj(csub(e, g)));
            rowbuf[M - k] = t2.re * s1;
            if (k >= 1) rowbuf[M + k] = -t2.im * s1;  // N - (M - k)
          }
        }
      }
    }
    if (INVERSE) {
      // x[4i .. 4i + 3] = (v[2i], v[N - 1 - 2i], v[2i + 1], v[N - 2 - 2i]) = (Re z[i], Im z[M-1-i], Im z[i], Re z[M-1-i]):
      // straight from the last stage's buffer to a coalesced 16-byte store
      const float inv = 1.0f / (float)M;
      float4* out4 = reinterpret_cast<float4*>(dv.out[row / rows_per_vol] + (row % rows_per_vol) * N);
#pragma unroll
      for (int i = lane; i < N / 4; i += 64) {
        const Cf zi = a[i], zj = a[M - 1 - i];
        out4[i] = make_float4(zi.re * inv, zj.im * inv, zi.im * inv, zj.re * inv);
      }
    } else {
      wave_lds_sync();
      float4* out4 = reinterpret_cast<float4*>(dv.out[row / rows_per_vol] + (row % rows_per_vol) * N);
      const float4* rb4 = reinterpret_cast<const float4*>(rowbuf);
#pragma unroll
      for (int i = lane; i < N / 4; i += 64) out4[i] = rb4[i];
    }
    wave_lds_sync();
  }
}

inline bool dct_fft_ok(int64_t rows, int64_t n, const void* a, const void* b) {  // one volume's operands
  return n >= 64 && n <= 1024 && (n & (n - 1)) == 0 && rows >= 1 && (uintptr_t)a % 16 == 0 && (uintptr_t)b % 16 == 0 &&
         !getenv("NDMPS_DCT_GEMM");
}
template <bool INVERSE, int NN>
int dct_fft_launch_n(const DctVols& dv, int count, int64_t rows_per_vol, int64_t n, hipStream_t s) {
  const int64_t rows = rows_per_vol * count;
  const int M = (int)n / 2;
  const size_t lds = (size_t)(M + 2 * (M + 1)) * sizeof(Cf) + (size_t)4 * (n + 4 * M) * sizeof(float);
  // exactly the workgroups the device keeps resident at once (every workgroup loops over rows: a partly filled second
  // round of workgroups would leave a third of the GPU idle for half the kernel)
  // asked once per (direction, instantiation, row length, device): the query costs host time on every volume otherwise
  static std::mutex mu;
  static std::map<std::pair<int, int64_t>, int> cache;
  int dev = 0, per_cu = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find({dev, n});
    if (it == cache.end()) {
      NDMPS_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dct_fft_kernel<INVERSE, NN>, 256, lds));
      cache[{dev, n}] = per_cu;
    } else {
      per_cu = it->second;
    }
  }
  const int grid = (int)std::min<int64_t>((rows + 3) / 4, (int64_t)ndmps::kNumCU * std::max(per_cu, 1));
  hipLaunchKernelGGL((dct_fft_kernel<INVERSE, NN>), dim3(grid), dim3(256), lds, s, dv, rows, rows_per_vol, (int)n);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
template <bool INVERSE>
int dct_fft_launch_vols(const DctVols& dv, int count, int64_t rows_per_vol, int64_t n, hipStream_t s) {
  switch (n) {  // the volume edges that occur: constants; anything else: the generic kernel
    case 128: return dct_fft_launch_n<INVERSE, 128>(dv, count, rows_per_vol, n, s);
    case 256: return dct_fft_launch_n<INVERSE, 256>(dv, count, rows_per_vol, n, s);
    case 512: return dct_fft_launch_n<INVERSE, 512>(dv, count, rows_per_vol, n, s);
    default: return dct_fft_launch_n<INVERSE, 0>(dv, count, rows_per_vol, n, s);
  }
}
template <bool INVERSE>
int dct_fft_launch(const float* src, float* dst, int64_t rows, int64_t n, hipStream_t s) {
  DctVols dv = {};
  dv.in[0] = src;
  dv.out[0] = dst;
  return dct_fft_launch_vols<INVERSE>(dv, 1, rows, n, s);
}
// the volumes of a lockstep group, thirty-two per launch; volumes the FFT does not take (row length, alignment) and
// every volume when the row length is not a power of two go through the basis product one by one
template <bool INVERSE>
int dct_many(int count, const float* const* h_in, float* const* h_out, int64_t rows, int64_t n, const float* d_basis,
             hipStream_t s) {
  bool fft = true;
  for (int v = 0; v < count; ++v) {
    NDMPS_REQUIRE(h_in[v] && h_out[v] && h_in[v] != h_out[v], "the DCT is out of place (volume %d)", v);
    fft = fft && dct_fft_ok(rows, n, h_in[v], h_out[v]);
  }
  if (!fft) {
    NDMPS_REQUIRE(d_basis != nullptr, "this row length needs the DCT basis");
    for (int v = 0; v < count; ++v)
      NDMPS_TRY(ndmps_sgemm(0, INVERSE ? 1 : 0, rows, n, n, h_in[v], n, d_basis, n, h_out[v], n, (ndmps_stream_t)s));
    return NDMPS_OK;
  }
  for (int v0 = 0; v0 < count; v0 += kDctVols) {
    const int nv = std::min(kDctVols, count - v0);
    DctVols dv = {};
    for (int v = 0; v < nv; ++v) {
      dv.in[v] = h_in[v0 + v];
      dv.out[v] = h_out[v0 + v];
    }
    NDMPS_TRY(dct_fft_launch_vols<INVERSE>(dv, nv, rows, n, s));
  }
  return NDMPS_OK;
}

// x[t] *= factor[t] for the tensors of a lockstep group in one launch (blockIdx.y = tensor)
constexpr int kScaleVols = 32;
struct ScaleVols {
  float* x[kScaleVols];
  double factor[kScaleVols];
};
__global__ void __launch_bounds__(256) scale_many_kernel(ScaleVols sv, int64_t n) {
  float* __restrict__ x = sv.x[blockIdx.y];
  const double f = sv.factor[blockIdx.y];
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] = (float)((double)x[i] * f);
}

int stream_grid(int64_t n) {
  return (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(n, 256), 1), (int64_t)ndmps::kNumCU * 8);
}

}  // namespace

extern "C" int64_t ndmps_reduce_workspace_bytes(void) { return kRedBlocks * 2 * sizeof(double) + 512; }

namespace {
template <typename T>
int sumsq_impl(const T* d_x, int64_t n, double* h_out, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && h_out && n >= 0, "bad sumsq argument");
  if (!d_ws || ws_bytes < ndmps_reduce_workspace_bytes()) {
    ndmps::set_error("reduce workspace too small");
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  double* partial = (double*)d_ws;
  double* result = partial + kRedBlocks;
  const int grid = (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(n, 256 * 8), 1), kRedBlocks);
  hipLaunchKernelGGL(sumsq_partial_kernel<T>, dim3(grid), dim3(256), 0, s, d_x, n, partial);
  hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(256), 0, s, partial, grid, result);
  NDMPS_LAUNCH_CHECK();
  NDMPS_CHECK_HIP(hipMemcpyAsync(h_out, result, sizeof(double), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}
}  // namespace

extern "C" int ndmps_sumsq_f32(const float* d_x, int64_t n, double* h_out, void* d_ws, int64_t ws_bytes,
                               ndmps_stream_t stream) {
  return sumsq_impl<float>(d_x, n, h_out, d_ws, ws_bytes, stream);
}
extern "C" int ndmps_sumsq_f64(const double* d_x, int64_t n, double* h_out, void* d_ws, int64_t ws_bytes,
                               ndmps_stream_t stream) {
  return sumsq_impl<double>(d_x, n, h_out, d_ws, ws_bytes, stream);
}

extern "C" int ndmps_minmax_f32(const float* d_x, int64_t n, float* h_min, float* h_max, void* d_ws,
                                int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && h_min && h_max && n > 0, "bad minmax argument");
  if (!d_ws || ws_bytes < ndmps_reduce_workspace_bytes()) {
    ndmps::set_error("reduce workspace too small");
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)d_ws;
  float* result = partial + 2 * kRedBlocks;
  const int grid = (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(n, 256 * 8), 1), kRedBlocks);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(grid), dim3(256), 0, s, d_x, n, partial);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(256), 0, s, partial, grid, result);
  NDMPS_LAUNCH_CHECK();
  float host[2];
  NDMPS_CHECK_HIP(hipMemcpyAsync(host, result, 2 * sizeof(float), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  *h_min = host[0];
  *h_max = host[1];
  return NDMPS_OK;
}

constexpr int kManySlices = 16;

extern "C" int64_t ndmps_minmax_many_workspace_bytes(int count) {
  if (count <= 0) return 0;
  return (int64_t)count * (8 + 8 + 3 * 8 * kManySlices) + 1024;
}

extern "C" int ndmps_minmax_many_f32(int count, const float* const* h_ptrs, const int64_t* h_lens,
                                     float* h_out, double* h_sumsq, void* d_ws, int64_t ws_bytes,
                                     ndmps_stream_t stream) {
  NDMPS_REQUIRE(count >= 1 && count <= 65535 && h_ptrs && h_lens && h_out, "bad minmax_many argument");
  if (!d_ws || ws_bytes < ndmps_minmax_many_workspace_bytes(count)) {
    ndmps::set_error("minmax_many workspace too small");
    return NDMPS_EWORKSPACE;
  }
  for (int i = 0; i < count; ++i) NDMPS_REQUIRE(h_ptrs[i] && h_lens[i] > 0, "tensor %d is empty or NULL", i);
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)d_ws;
  const float** d_ptrs = (const float**)base;
  int64_t* d_lens = (int64_t*)(base + ndmps::round_up((int64_t)count * 8, 256));
  double* partial = (double*)((char*)d_lens + ndmps::round_up((int64_t)count * 8, 256));
  many_upload((const void**)d_ptrs, d_lens, (const void* const*)h_ptrs, h_lens, count, s);
  hipLaunchKernelGGL(minmax_many_kernel, dim3(kManySlices, count), dim3(256), 0, s, d_ptrs, d_lens, partial);
  NDMPS_LAUNCH_CHECK();
  std::vector<double> host((size_t)count * kManySlices * 3);
  NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), partial, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < count; ++i) {
    double lo = FLT_MAX, hi = -FLT_MAX, ss = 0.0;
    for (int j = 0; j < kManySlices; ++j) {
      const double* o = &host[3 * ((size_t)i * kManySlices + j)];
      lo = std::min(lo, o[0]);
      hi = std::max(hi, o[1]);
      ss += o[2];
    }
    h_out[2 * i] = (float)lo;
    h_out[2 * i + 1] = (float)hi;
    if (h_sumsq) h_sumsq[i] = ss;
  }
  return NDMPS_OK;
}

// ---- the same reduction for the cores of a lockstep group that sit in one arena (row b = volume b, core i at
//      offsets[i], lens[i] elements), split in two so that nothing waits in between: _launch enqueues the kernel
//      (asynchronous; tensor (b, i) -> entry b * n_cores + i of the partial buffer), _collect brings the partials
//      back and folds them (synchronises).  Lets a caller issue the reductions with the sweep and read
//      boundary_list / norm (core/ndmps.py:75-76) when somebody asks for them.
namespace {
constexpr int kArenaCores = 64;
struct ArenaCores {
  int64_t offset[kArenaCores];
  int64_t len[kArenaCores];
};
__global__ void __launch_bounds__(256)
minmax_arena_kernel(const float* __restrict__ base, int64_t row_stride, int n_cores, ArenaCores ac,
                    double* __restrict__ partial) {
  __shared__ float rmin[4], rmax[4];
  __shared__ double rsum[4];
  const int t = blockIdx.y;  // tensor index = volume * n_cores + core
  const float* x = base + (int64_t)(t / n_cores) * row_stride + ac.offset[t % n_cores];
  const int64_t n = ac.len[t % n_cores];
  float lo = FLT_MAX, hi = -FLT_MAX;
  double ss = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
    ss += (double)v * (double)v;
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) {
    rmin[threadIdx.x >> 6] = lo;
    rmax[threadIdx.x >> 6] = hi;
    rsum[threadIdx.x >> 6] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = partial + 3 * ((int64_t)blockIdx.y * gridDim.x + blockIdx.x);
    o[0] = (double)fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
    o[1] = (double)fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3]));
    o[2] = (rsum[0] + rsum[1]) + (rsum[2] + rsum[3]);
  }
}
}  // namespace

extern "C" int64_t ndmps_minmax_partials_bytes(int count) {
  return count > 0 ? (int64_t)count * kManySlices * 3 * 8 : 0;
}

extern "C" int ndmps_minmax_arena_launch_f32(const float* d_base, int64_t row_stride, int batch, int n_cores,
                                             const int64_t* h_offsets, const int64_t* h_lens, double* d_partial,
                                             ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_base && d_partial && h_offsets && h_lens && batch >= 1 && n_cores >= 1 && n_cores <= kArenaCores &&
                    (int64_t)batch * n_cores <= 65535,
                "bad minmax_arena argument (batch=%d cores=%d)", batch, n_cores);
  ArenaCores ac;
  for (int i = 0; i < n_cores; ++i) {
    NDMPS_REQUIRE(h_offsets[i] >= 0 && h_lens[i] > 0 && h_offsets[i] + h_lens[i] <= row_stride, "core %d leaves its arena row", i);
    ac.offset[i] = h_offsets[i];
    ac.len[i] = h_lens[i];
  }
  hipLaunchKernelGGL(minmax_arena_kernel, dim3(kManySlices, batch * n_cores), dim3(256), 0, (hipStream_t)stream, d_base,
                     row_stride, n_cores, ac, d_partial);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_minmax_collect(int count, const double* d_partial, float* h_out, double* h_sumsq,
                                    ndmps_stream_t stream) {
  NDMPS_REQUIRE(count >= 1 && d_partial && h_out, "bad minmax_collect argument");
  hipStream_t s = (hipStream_t)stream;
  std::vector<double> host((size_t)count * kManySlices * 3);
  NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), d_partial, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < count; ++i) {
    double lo = FLT_MAX, hi = -FLT_MAX, ss = 0.0;
    for (int j = 0; j < kManySlices; ++j) {
      const double* o = &host[3 * ((size_t)i * kManySlices + j)];
      lo = std::min(lo, o[0]);
      hi = std::max(hi, o[1]);
      ss += o[2];
    }
    h_out[2 * i] = (float)lo;
    h_out[2 * i + 1] = (float)hi;
    if (h_sumsq) h_sumsq[i] = ss;
  }
  return NDMPS_OK;
}

extern "C" int ndmps_scale_f32(float* d_x, int64_t n, double factor, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && n >= 0, "bad scale argument");
  if (n == 0) return NDMPS_OK;
  hipLaunchKernelGGL(scale_kernel<float>, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, d_x, n, factor);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
extern "C" int ndmps_scale_f64(double* d_x, int64_t n, double factor, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && n >= 0, "bad scale argument");
  if (n == 0) return NDMPS_OK;
  hipLaunchKernelGGL(scale_kernel<double>, dim3(stream_grid(n)), dim3(256), 0, (hipStream_t)stream, d_x, n, factor);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_minmax_many_f64(int count, const double* const* h_ptrs, const int64_t* h_lens, double* h_out,
                                     double* h_sumsq, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(count >= 1 && count <= 65535 && h_ptrs && h_lens && h_out, "bad minmax_many argument");
  if (!d_ws || ws_bytes < ndmps_minmax_many_workspace_bytes(count)) {
    ndmps::set_error("minmax_many workspace too small");
    return NDMPS_EWORKSPACE;
  }
  for (int i = 0; i < count; ++i) NDMPS_REQUIRE(h_ptrs[i] && h_lens[i] > 0, "tensor %d is empty or NULL", i);
  hipStream_t s = (hipStream_t)stream;
  char* base = (char*)d_ws;
  const double** d_ptrs = (const double**)base;
  int64_t* d_lens = (int64_t*)(base + ndmps::round_up((int64_t)count * 8, 256));
  double* partial = (double*)((char*)d_lens + ndmps::round_up((int64_t)count * 8, 256));
  many_upload((const void**)d_ptrs, d_lens, (const void* const*)h_ptrs, h_lens, count, s);
  hipLaunchKernelGGL(minmax_many_f64_kernel, dim3(kManySlices, count), dim3(256), 0, s, d_ptrs, d_lens, partial);
  NDMPS_LAUNCH_CHECK();
  std::vector<double> host((size_t)count * kManySlices * 3);
  NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), partial, host.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  for (int i = 0; i < count; ++i) {
    double lo = DBL_MAX, hi = -DBL_MAX, ss = 0.0;
    for (int j = 0; j < kManySlices; ++j) {
      const double* o = &host[3 * ((size_t)i * kManySlices + j)];
      lo = std::min(lo, o[0]);
      hi = std::max(hi, o[1]);
      ss += o[2];
    }
    h_out[2 * i] = lo;
    h_out[2 * i + 1] = hi;
    if (h_sumsq) h_sumsq[i] = ss;
  }
  return NDMPS_OK;
}

extern "C" int ndmps_dct_basis_f32(float* d_basis, int64_t n, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_basis && n >= 1 && n <= 16384, "bad DCT length %lld", (long long)n);
  hipLaunchKernelGGL(dct_basis_kernel<float>, dim3(stream_grid(n * n)), dim3(256), 0, (hipStream_t)stream, d_basis, n);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
extern "C" int ndmps_dct_basis_f64(double* d_basis, int64_t n, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_basis && n >= 1 && n <= 16384, "bad DCT length %lld", (long long)n);
  hipLaunchKernelGGL(dct_basis_kernel<double>, dim3(stream_grid(n * n)), dim3(256), 0, (hipStream_t)stream, d_basis, n);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
extern "C" int ndmps_dct_last_f64(const double* d_x, double* d_y, int64_t rows, int64_t n, const double* d_basis,
                                  ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x != d_y, "DCT is out of place");
  return ndmps_dgemm(0, 0, rows, n, n, d_x, n, d_basis, n, d_y, n, stream);
}
extern "C" int ndmps_idct_last_f64(const double* d_y, double* d_x, int64_t rows, int64_t n, const double* d_basis,
                                   ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x != d_y, "IDCT is out of place");
  return ndmps_dgemm(0, 1, rows, n, n, d_y, n, d_basis, n, d_x, n, stream);
}

extern "C" int ndmps_dct_last_f32(const float* d_x, float* d_y, int64_t rows, int64_t n,
                                  const float* d_basis, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x != d_y, "DCT is out of place");
  // power-of-two rows: O(N log N) in LDS, bound by reading and writing the rows once; other lengths: the basis product
  if (d_x && d_y && dct_fft_ok(rows, n, d_x, d_y)) return dct_fft_launch<false>(d_x, d_y, rows, n, (hipStream_t)stream);
  return ndmps_sgemm(0, 0, rows, n, n, d_x, n, d_basis, n, d_y, n, stream);
}

extern "C" int ndmps_idct_last_f32(const float* d_y, float* d_x, int64_t rows, int64_t n,
                                   const float* d_basis, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x != d_y, "IDCT is out of place");
  if (d_x && d_y && dct_fft_ok(rows, n, d_x, d_y)) return dct_fft_launch<true>(d_y, d_x, rows, n, (hipStream_t)stream);
  return ndmps_sgemm(0, 1, rows, n, n, d_y, n, d_basis, n, d_x, n, stream);
}

// The same transforms for the volumes of a lockstep group in one launch (thirty-two per launch): h_x / h_y are HOST arrays
// of `count` device pointers, every volume (rows, n) row-major.  The reference transforms volume by volume in a Python
// loop (evaluation/benchmark.py:80-100 around core/ndmps.py:62-63 and :152-153); results equal the single-volume calls
// bit for bit.
extern "C" int ndmps_dct_last_many_f32(int count, const float* const* h_x, float* const* h_y, int64_t rows, int64_t n,
                                       const float* d_basis, ndmps_stream_t stream) {
  NDMPS_REQUIRE(count >= 1 && h_x && h_y && rows >= 1 && n >= 1, "bad dct_last_many argument");
  return dct_many<false>(count, h_x, h_y, rows, n, d_basis, (hipStream_t)stream);
}
extern "C" int ndmps_idct_last_many_f32(int count, const float* const* h_y, float* const* h_x, int64_t rows, int64_t n,
                                        const float* d_basis, ndmps_stream_t stream) {
  NDMPS_REQUIRE(count >= 1 && h_x && h_y && rows >= 1 && n >= 1, "bad idct_last_many argument");
  return dct_many<true>(count, h_y, h_x, rows, n, d_basis, (hipStream_t)stream);
}
// h_x[t][0 .. n) *= h_factor[t] for `count` fp32 tensors of n elements each, thirty-two per launch (the division by the
// norm of core/ndmps.py:60-61 for a whole lockstep group)
extern "C" int ndmps_scale_many_f32(int count, float* const* h_x, int64_t n, const double* h_factor, ndmps_stream_t stream) {
  NDMPS_REQUIRE(count >= 1 && h_x && h_factor && n >= 0, "bad scale_many argument");
  if (n == 0) return NDMPS_OK;
  for (int v0 = 0; v0 < count; v0 += kScaleVols) {
    const int nv = std::min(kScaleVols, count - v0);
    ScaleVols sv = {};
    for (int v = 0; v < nv; ++v) {
      NDMPS_REQUIRE(h_x[v0 + v] != nullptr, "tensor %d is NULL", v0 + v);
      sv.x[v] = h_x[v0 + v];
      sv.factor[v] = h_factor[v0 + v];
    }
    const int gx = (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(n, 256 * 8), 1), (int64_t)ndmps::kNumCU * 8);
    hipLaunchKernelGGL(scale_many_kernel, dim3(gx, nv), dim3(256), 0, (hipStream_t)stream, sv, n);
  }
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_quantize_f32(const float* d_x, int64_t n, float lo, float hi, int bits, void* d_q,
                                  ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && d_q && n >= 0, "bad quantize argument");
  NDMPS_REQUIRE(bits == 8 || bits == 16, "bits=%d not supported (8 or 16)", bits);
  if (n == 0) return NDMPS_OK;
  const double span = (double)hi - (double)lo;
  hipStream_t s = (hipStream_t)stream;
  if (bits == 8)
    hipLaunchKernelGGL(quantize_kernel<uint8_t>, dim3(stream_grid(n)), dim3(256), 0, s, d_x, n, (double)lo, span,
                       255.0, (uint8_t*)d_q);
  else
    hipLaunchKernelGGL(quantize_kernel<uint16_t>, dim3(stream_grid(n)), dim3(256), 0, s, d_x, n, (double)lo,
                       span, 65535.0, (uint16_t*)d_q);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_dequantize_f32(const void* d_q, int64_t n, float lo, float hi, int bits, float* d_x,
                                    ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && d_q && n >= 0, "bad dequantize argument");
  NDMPS_REQUIRE(bits == 8 || bits == 16, "bits=%d not supported (8 or 16)", bits);
  if (n == 0) return NDMPS_OK;
  const double span = (double)hi - (double)lo;
  hipStream_t s = (hipStream_t)stream;
  if (bits == 8)
    hipLaunchKernelGGL(dequantize_kernel<uint8_t>, dim3(stream_grid(n)), dim3(256), 0, s, (const uint8_t*)d_q, n,
                       (double)lo, span, 255.0, d_x);
  else
    hipLaunchKernelGGL(dequantize_kernel<uint16_t>, dim3(stream_grid(n)), dim3(256), 0, s, (const uint16_t*)d_q,
                       n, (double)lo, span, 65535.0, d_x);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_quantize_f64(const double* d_x, int64_t n, double lo, double hi, int bits, void* d_q,
                                  ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && d_q && n >= 0, "bad quantize argument");
  NDMPS_REQUIRE(bits == 8 || bits == 16, "bits=%d not supported (8 or 16)", bits);
  if (n == 0) return NDMPS_OK;
  const double span = hi - lo;
  hipStream_t s = (hipStream_t)stream;
  if (bits == 8)
    hipLaunchKernelGGL((quantize_kernel<uint8_t, double>), dim3(stream_grid(n)), dim3(256), 0, s, d_x, n, lo, span, 255.0,
                       (uint8_t*)d_q);
  else
    hipLaunchKernelGGL((quantize_kernel<uint16_t, double>), dim3(stream_grid(n)), dim3(256), 0, s, d_x, n, lo, span,
                       65535.0, (uint16_t*)d_q);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}

extern "C" int ndmps_dequantize_f64(const void* d_q, int64_t n, double lo, double hi, int bits, double* d_x,
                                    ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_x && d_q && n >= 0, "bad dequantize argument");
  NDMPS_REQUIRE(bits == 8 || bits == 16, "bits=%d not supported (8 or 16)", bits);
  if (n == 0) return NDMPS_OK;
  const double span = hi - lo;
  hipStream_t s = (hipStream_t)stream;
  if (bits == 8)
    hipLaunchKernelGGL((dequantize_kernel<uint8_t, double>), dim3(stream_grid(n)), dim3(256), 0, s, (const uint8_t*)d_q, n,
                       lo, span, 255.0, d_x);
  else
    hipLaunchKernelGGL((dequantize_kernel<uint16_t, double>), dim3(stream_grid(n)), dim3(256), 0, s, (const uint16_t*)d_q,
                       n, lo, span, 65535.0, d_x);
  NDMPS_LAUNCH_CHECK();
  return NDMPS_OK;
}
