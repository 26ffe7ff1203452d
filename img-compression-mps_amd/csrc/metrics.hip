// Quality metrics on device-resident volumes: SSIM and PSNR with the exact
// semantics of the reference's utils/metrics.py:
//   compute_ssim_2d (metrics.py:11-32)  skimage structural_similarity with an explicit data_range,
//       uniform window win = min(7, min_dim) made odd, K1 = 0.01, K2 = 0.03, sample covariance,
//       border crop (win-1)/2; data_range = joint max - joint min of the slice pair; the SECOND
//       argument is clipped at 0 first (metrics.py:23,54).
//   avg_ssim_3d (metrics.py:68-85)      mean over the three axes of the mean slice-wise SSIM.
//   avg_ssim_4d (metrics.py:88-105)     mean over last-axis frames of avg_ssim_3d.
//   compute_psnr (metrics.py:132-146)   10 log10(max(original)^2 / mse).
// fp32 inputs, fp64 arithmetic (the reference evaluates float64 arrays in float64).
//
// One workgroup = one 16 x 32 tile of window centres of one 2-D slice: the (16+win-1) x (32+win-1)
// patch of both volumes goes to LDS, box sums are separable (row sums to LDS, then column sums),
// per-workgroup partial sums are folded per slice in fixed order (deterministic).  Slices along the
// contiguous axis are read with a stride; neighbouring slices run concurrently and share lines in L2.
#include <float.h>
#include <math.h>

#include <algorithm>
#include <vector>

#include "common.h"

namespace {

constexpr int TH = 16, TW = 32, WMAX = 7;
constexpr int PH = TH + WMAX - 1, PW = TW + WMAX - 1;

struct Slicing {        // 2-D slices of a (D0, D1, D2[, T]) volume along one axis
  int64_t base_stride_t, base_stride_i;  // slice z = t * n_i + i starts at t*base_stride_t + i*base_stride_i
  int64_t n_i;                           // slices per frame
  int64_t H, W, sH, sW;                  // slice extent and element strides
};

__device__ __forceinline__ int64_t slice_base(const Slicing& s, int64_t z) {
  return (z / s.n_i) * s.base_stride_t + (z % s.n_i) * s.base_stride_i;
}

// data_range of every slice pair: max(a.max, clip(b).max) - min(a.min, clip(b).min)
__global__ void __launch_bounds__(256)
slice_range_kernel(const float* __restrict__ a, const float* __restrict__ b, Slicing s, double* __restrict__ range) {
  __shared__ float rmin[4], rmax[4];
  const int64_t base = slice_base(s, blockIdx.x);
  float lo = FLT_MAX, hi = -FLT_MAX;
  const int64_t n = s.H * s.W;
  for (int64_t e = threadIdx.x; e < n; e += 256) {
    const int64_t off = base + (e / s.W) * s.sH + (e % s.W) * s.sW;
    const float x = a[off], y = fmaxf(b[off], 0.f);
    lo = fminf(lo, fminf(x, y));
    hi = fmaxf(hi, fmaxf(x, y));
  }
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_down(lo, off, 64));
    hi = fmaxf(hi, __shfl_down(hi, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    rmin[threadIdx.x >> 6] = lo;
    rmax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0)
    range[blockIdx.x] = (double)fmaxf(fmaxf(rmax[0], rmax[1]), fmaxf(rmax[2], rmax[3])) -
                        (double)fminf(fminf(rmin[0], rmin[1]), fminf(rmin[2], rmin[3]));
}

// grid (tiles_w, tiles_h, n_slices); partial[z][tile] = sum of the SSIM map over the tile
__global__ void __launch_bounds__(256)
ssim_tile_kernel(const float* __restrict__ a, const float* __restrict__ b, Slicing s, int win,
                 const double* __restrict__ range, double* __restrict__ partial, int64_t z_off) {
  __shared__ double pa[PH][PW + 1], pb[PH][PW + 1];
  __shared__ double hs[5][PH][TW + 1];
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int64_t z = z_off + blockIdx.z;  // grid.z is limited to 65535 slices per launch
  const int64_t base = slice_base(s, z);
  const int64_t oh = s.H - win + 1, ow = s.W - win + 1;  // window centres (cropped map)
  const int64_t y0 = (int64_t)blockIdx.y * TH, x0 = (int64_t)blockIdx.x * TW;
  const int ph = (int)min<int64_t>(TH, oh - y0) + win - 1, pw = (int)min<int64_t>(TW, ow - x0) + win - 1;
  for (int e = tid; e < ph * pw; e += 256) {
    const int r = e / pw, c = e % pw;
    const int64_t off = base + (y0 + r) * s.sH + (x0 + c) * s.sW;
    pa[r][c] = (double)a[off];
    pb[r][c] = (double)fmaxf(b[off], 0.f);
  }
  __syncthreads();
  const int tw = pw - win + 1, th = ph - win + 1;  // outputs of this tile
  for (int e = tid; e < ph * tw; e += 256) {  // row sums
    const int r = e / tw, c = e % tw;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int k = 0; k < win; ++k) {
      const double x = pa[r][c + k], y = pb[r][c + k];
      sx += x;
      sy += y;
      sxx += x * x;
      syy += y * y;
      sxy += x * y;
    }
    hs[0][r][c] = sx;
    hs[1][r][c] = sy;
    hs[2][r][c] = sxx;
    hs[3][r][c] = syy;
    hs[4][r][c] = sxy;
  }
  __syncthreads();
  const double np_ = (double)(win * win), cov_norm = np_ / (np_ - 1.0);
  const double R = range[z], c1 = (0.01 * R) * (0.01 * R), c2 = (0.03 * R) * (0.03 * R);
  double acc = 0.0;
  for (int e = tid; e < th * tw; e += 256) {  // column sums + SSIM
    const int r = e / tw, c = e % tw;
    double sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
    for (int k = 0; k < win; ++k) {
      sx += hs[0][r + k][c];
      sy += hs[1][r + k][c];
      sxx += hs[2][r + k][c];
      syy += hs[3][r + k][c];
      sxy += hs[4][r + k][c];
    }
    const double ux = sx / np_, uy = sy / np_;
    const double vx = cov_norm * (sxx / np_ - ux * ux), vy = cov_norm * (syy / np_ - uy * uy);
    const double vxy = cov_norm * (sxy / np_ - ux * uy);
    acc += ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0)
    partial[(z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// mean over slices of (sum of tile partials / map size); one workgroup, fixed order
__global__ void __launch_bounds__(256)
ssim_finish_kernel(const double* __restrict__ partial, int64_t n_slices, int tiles, double map_size,
                   double* __restrict__ out) {
  __shared__ double red[256];
  double acc = 0.0;
  for (int64_t z = threadIdx.x; z < n_slices; z += 256) {
    double s = 0.0;
    for (int t = 0; t < tiles; ++t) s += partial[z * tiles + t];
    acc += s / map_size;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] / (double)n_slices;
}

// one value per slice: sum of its tile partials / map size (the per-slice list of ssim_3d_axis)
__global__ void __launch_bounds__(256)
ssim_slices_kernel(const double* __restrict__ partial, int64_t n_slices, int tiles, double map_size,
                   double* __restrict__ out) {
  const int64_t z = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (z >= n_slices) return;
  double s = 0.0;
  for (int t = 0; t < tiles; ++t) s += partial[z * tiles + t];  // fixed order, as ssim_finish_kernel
  out[z] = s / map_size;
}

__global__ void __launch_bounds__(256)
psnr_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, double* __restrict__ partial) {
  __shared__ double rs[4];
  __shared__ float rm[4];
  double acc = 0.0;
  float mx = -FLT_MAX;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const double d = (double)a[i] - (double)b[i];
    acc += d * d;
    mx = fmaxf(mx, a[i]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_down(acc, off, 64);
    mx = fmaxf(mx, __shfl_down(mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    rs[threadIdx.x >> 6] = acc;
    rm[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = (rs[0] + rs[1]) + (rs[2] + rs[3]);
    partial[2 * blockIdx.x + 1] = (double)fmaxf(fmaxf(rm[0], rm[1]), fmaxf(rm[2], rm[3]));
  }
}

constexpr int kPsnrBlocks = 1024;

struct SsimPlan {
  int n_axes;
  Slicing sl[3];
  int64_t n_slices[3];
  int win;
};

int ssim_plan(int ndim, const int64_t* shape, SsimPlan& p) {
  NDMPS_REQUIRE(ndim >= 2 && ndim <= 4, "Unsupported tensor dimension for SSIM: %d", ndim);
  for (int i = 0; i < ndim; ++i) NDMPS_REQUIRE(shape[i] >= 1, "shape[%d] must be positive", i);
  if (ndim == 2) {
    p.n_axes = 1;
    p.sl[0] = Slicing{0, 0, 1, shape[0], shape[1], shape[1], 1};
    p.n_slices[0] = 1;
    int64_t md = std::min(shape[0], shape[1]);
    p.win = (int)std::min<int64_t>(7, md);
  } else {
    const int64_t D0 = shape[0], D1 = shape[1], D2 = shape[2], T = ndim == 4 ? shape[3] : 1;
    const int64_t s2 = T, s1 = D2 * T, s0 = D1 * D2 * T;  // element strides of the three spatial axes
    p.n_axes = 3;
    p.sl[0] = Slicing{1, s0, D0, D1, D2, s1, s2};  // slices [i, :, :]
    p.sl[1] = Slicing{1, s1, D1, D0, D2, s0, s2};  // slices [:, i, :]
    p.sl[2] = Slicing{1, s2, D2, D0, D1, s0, s1};  // slices [:, :, i]
    p.n_slices[0] = D0 * T;
    p.n_slices[1] = D1 * T;
    p.n_slices[2] = D2 * T;
    // win_size = min(7, min(slice shape)) made odd -- per slice orientation it would differ only if
    // an extent is below 7; the reference evaluates it per 2-D call, so do the same per axis below
    p.win = 0;
  }
  return NDMPS_OK;
}

inline int win_for(int64_t H, int64_t W) {
  int w = (int)std::min<int64_t>(7, std::min(H, W));
  if (w % 2 == 0) --w;
  return w;
}

}  // namespace

extern "C" int64_t ndmps_ssim_workspace_bytes(int ndim, const int64_t* h_shape) {
  SsimPlan p;
  if (!h_shape || ssim_plan(ndim, h_shape, p) != NDMPS_OK) return -1;
  int64_t worst = 0;
  for (int ax = 0; ax < p.n_axes; ++ax) {
    const int w = win_for(p.sl[ax].H, p.sl[ax].W);
    if (w < 3) return -1;
    const int64_t oh = p.sl[ax].H - w + 1, ow = p.sl[ax].W - w + 1;
    const int64_t tiles = ndmps::ceil_div(oh, TH) * ndmps::ceil_div(ow, TW);
    worst = std::max(worst, p.n_slices[ax] * (tiles + 1) * 8);
  }
  return worst + 1024;
}

// h_out: the SSIM of compute_ssim_by_dim(a, b) (a = "original", b = "compressed", clipped at 0)
extern "C" int ndmps_ssim_f32(const float* d_a, const float* d_b, int ndim, const int64_t* h_shape,
                              double* h_out, void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_a && d_b && h_shape && h_out, "NULL SSIM argument");
  SsimPlan p;
  NDMPS_TRY(ssim_plan(ndim, h_shape, p));
  const int64_t need = ndmps_ssim_workspace_bytes(ndim, h_shape);
  NDMPS_REQUIRE(need >= 0, "win_size exceeds image extent (every slice needs at least 3 x 3 pixels)");
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("SSIM workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  double total = 0.0;
  for (int ax = 0; ax < p.n_axes; ++ax) {
    const Slicing& sl = p.sl[ax];
    const int w = win_for(sl.H, sl.W);
    const int64_t nz = p.n_slices[ax];
    const int64_t oh = sl.H - w + 1, ow = sl.W - w + 1;
    const unsigned gx = (unsigned)ndmps::ceil_div(ow, TW), gy = (unsigned)ndmps::ceil_div(oh, TH);
    NDMPS_REQUIRE(nz < 2147483647LL && gy < 65536, "SSIM grid too large");
    double* range = (double*)d_ws;
    double* partial = range + nz;
    double* result = partial + nz * gx * gy;
    hipLaunchKernelGGL(slice_range_kernel, dim3((unsigned)nz), dim3(256), 0, s, d_a, d_b, sl, range);
    for (int64_t z0 = 0; z0 < nz; z0 += 65535) {
      const unsigned gz = (unsigned)std::min<int64_t>(65535, nz - z0);
      hipLaunchKernelGGL(ssim_tile_kernel, dim3(gx, gy, gz), dim3(256), 0, s, d_a, d_b, sl, w, range, partial, z0);
    }
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(256), 0, s, partial, nz, (int)(gx * gy),
                       (double)(oh * ow), result);
    NDMPS_LAUNCH_CHECK();
    double axis_mean = 0.0;
    NDMPS_CHECK_HIP(hipMemcpyAsync(&axis_mean, result, sizeof(double), hipMemcpyDeviceToHost, s));
    NDMPS_CHECK_HIP(hipStreamSynchronize(s));
    total += axis_mean;
  }
  *h_out = total / p.n_axes;
  return NDMPS_OK;
}

// ssim_3d_axis(a, b, axis) (metrics.py:35-65): the SSIM of every 2-D slice along `axis` of a 3-D volume, b clipped at
// 0, each slice with its own joint data range and the window of its own extent.  h_out: shape[axis] values.  Same
// kernels as ndmps_ssim_f32 (whose 3-D value is the mean over the three axes of the means of these lists); workspace
// as ndmps_ssim_workspace_bytes(3, shape) + 8 * shape[axis].
extern "C" int64_t ndmps_ssim_slices_workspace_bytes(const int64_t* h_shape, int axis) {
  if (!h_shape || axis < 0 || axis > 2) return -1;
  const int64_t base = ndmps_ssim_workspace_bytes(3, h_shape);
  return base < 0 ? -1 : base + 8 * h_shape[axis] + 256;
}
extern "C" int ndmps_ssim_slices_f32(const float* d_a, const float* d_b, const int64_t* h_shape, int axis, double* h_out,
                                     void* d_ws, int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_a && d_b && h_shape && h_out, "NULL SSIM argument");
  NDMPS_REQUIRE(axis >= 0 && axis <= 2, "Invalid axis %d for 3D SSIM.", axis);
  SsimPlan p;
  NDMPS_TRY(ssim_plan(3, h_shape, p));
  const int64_t need = ndmps_ssim_slices_workspace_bytes(h_shape, axis);
  NDMPS_REQUIRE(need >= 0, "win_size exceeds image extent (every slice needs at least 3 x 3 pixels)");
  if (!d_ws || ws_bytes < need) {
    ndmps::set_error("SSIM workspace too small: %lld < %lld", (long long)ws_bytes, (long long)need);
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const Slicing& sl = p.sl[axis];
  const int w = win_for(sl.H, sl.W);
  const int64_t nz = p.n_slices[axis];
  const int64_t oh = sl.H - w + 1, ow = sl.W - w + 1;
  const unsigned gx = (unsigned)ndmps::ceil_div(ow, TW), gy = (unsigned)ndmps::ceil_div(oh, TH);
  NDMPS_REQUIRE(nz < 2147483647LL && gy < 65536, "SSIM grid too large");
  double* range = (double*)d_ws;
  double* partial = range + nz;
  double* values = partial + nz * gx * gy;
  hipLaunchKernelGGL(slice_range_kernel, dim3((unsigned)nz), dim3(256), 0, s, d_a, d_b, sl, range);
  for (int64_t z0 = 0; z0 < nz; z0 += 65535) {
    const unsigned gz = (unsigned)std::min<int64_t>(65535, nz - z0);
    hipLaunchKernelGGL(ssim_tile_kernel, dim3(gx, gy, gz), dim3(256), 0, s, d_a, d_b, sl, w, range, partial, z0);
  }
  hipLaunchKernelGGL(ssim_slices_kernel, dim3((unsigned)ndmps::ceil_div(nz, 256)), dim3(256), 0, s, partial, nz,
                     (int)(gx * gy), (double)(oh * ow), values);
  NDMPS_LAUNCH_CHECK();
  NDMPS_CHECK_HIP(hipMemcpyAsync(h_out, values, sizeof(double) * nz, hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  return NDMPS_OK;
}

extern "C" int64_t ndmps_psnr_workspace_bytes(void) { return kPsnrBlocks * 2 * 8 + 256; }

// 10 log10(max(a)^2 / mean((a - b)^2)); +inf when the arrays are identical (metrics.py:143-146)
extern "C" int ndmps_psnr_f32(const float* d_a, const float* d_b, int64_t n, double* h_out, void* d_ws,
                              int64_t ws_bytes, ndmps_stream_t stream) {
  NDMPS_REQUIRE(d_a && d_b && h_out && n > 0, "bad PSNR argument");
  if (!d_ws || ws_bytes < ndmps_psnr_workspace_bytes()) {
    ndmps::set_error("PSNR workspace too small");
    return NDMPS_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  double* partial = (double*)d_ws;
  const int grid = (int)std::min<int64_t>(std::max<int64_t>(ndmps::ceil_div(n, 256 * 8), 1), kPsnrBlocks);
  hipLaunchKernelGGL(psnr_partial_kernel, dim3(grid), dim3(256), 0, s, d_a, d_b, n, partial);
  NDMPS_LAUNCH_CHECK();
  std::vector<double> host(2 * (size_t)grid);
  NDMPS_CHECK_HIP(hipMemcpyAsync(host.data(), partial, host.size() * 8, hipMemcpyDeviceToHost, s));
  NDMPS_CHECK_HIP(hipStreamSynchronize(s));
  double sum = 0.0, mx = -DBL_MAX;
  for (int i = 0; i < grid; ++i) {
    sum += host[2 * i];
    mx = std::max(mx, host[2 * i + 1]);
  }
  const double mse = sum / (double)n;
  *h_out = mse == 0.0 ? INFINITY : 10.0 * log10(mx * mx / mse);
  return NDMPS_OK;
}
