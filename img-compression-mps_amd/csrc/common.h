// Shared host-side helpers for libndmps_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ndmps_hip.h"

namespace ndmps {

void set_error(const char* fmt, ...);

#define NDMPS_CHECK_HIP(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ndmps::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),        \
                       __FILE__, __LINE__);                                          \
      return NDMPS_EHIP;                                                             \
    }                                                                                \
  } while (0)

#define NDMPS_REQUIRE(cond, ...)                                                     \
  do {                                                                               \
    if (!(cond)) {                                                                   \
      ndmps::set_error(__VA_ARGS__);                                                 \
      return NDMPS_EINVAL;                                                           \
    }                                                                                \
  } while (0)

#define NDMPS_TRY(expr)                                                              \
  do {                                                                               \
    int _r = (expr);                                                                 \
    if (_r != NDMPS_OK) return _r;                                                   \
  } while (0)

#define NDMPS_LAUNCH_CHECK() NDMPS_CHECK_HIP(hipGetLastError())

static inline int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }
static inline int64_t ceil_div(int64_t x, int64_t a) { return (x + a - 1) / a; }

// bump allocator over a caller-provided device workspace (256-byte aligned pieces)
struct Arena {
  char* base;
  int64_t size;
  int64_t used;
  Arena(void* p, int64_t n) : base((char*)p), size(n), used(0) {}
  template <typename T>
  T* take(int64_t count) {
    int64_t off = round_up(used, 256);
    int64_t end = off + (int64_t)sizeof(T) * count;
    used = end;
    if (end > size || base == nullptr) return nullptr;
    return (T*)(base + off);
  }
};
static inline int64_t arena_bytes(int64_t used, int64_t elem, int64_t count) {
  return round_up(used, 256) + elem * count;
}

constexpr int kNumCU = 256;  // MI355X

// element conversions of the storage types (fp32, bf16, fp64) to and from the fp64 side of the path
__device__ __forceinline__ double to_f64(float x) { return (double)x; }
__device__ __forceinline__ double to_f64(__bf16 x) { return (double)(float)x; }
__device__ __forceinline__ double to_f64(double x) { return x; }
template <typename T>
__device__ __forceinline__ T from_f64(double x) { return (T)(float)x; }
template <>
__device__ __forceinline__ double from_f64<double>(double x) { return x; }

// launch spans for bench.py's roofline (util.hip); slot ids
constexpr int kSpanTridiagColumns = 1;
constexpr int kSpanTridiagTeam = 2;
constexpr int kSpanGram = 3;       // Gram launches that take the device-side turn (a lockstep group's raw Gram); `bytes` = flops
constexpr int kSpanGramSmall = 4;  // every other batched Gram launch; `bytes` = flops
constexpr int kSpanTridiagPanel = 5;   // panel-blocked tridiagonalisation (orders from 1536 on): two launches per column
constexpr int kSpanTridiagTail = 6;    // last 128 columns in LDS + eigenvalues by multi-section (single-workgroup kernels)
constexpr int kSpanEigenVectors = 7;   // inverse iteration, orthonormalisation, back-transformation of the kept vectors
void* span_begin(hipStream_t s);
void span_end(void* begin, hipStream_t s, int slot, int64_t launches, int64_t bytes);

// turns between streams for kernels that fill the GPU alone (util.hip): begin() enqueues the acquire, end() (or the
// destructor, on an early return) the release; the submission of a whole turn is atomic per device
constexpr int kTurnTeam = 0, kTurnGram = 1;
class Turn {
 public:
  Turn(hipStream_t s, int which, unsigned weight = 1, unsigned cap = 1) : s_(s), which_(which), weight_(weight), cap_(cap) {}
  ~Turn();
  Turn(const Turn&) = delete;
  Turn& operator=(const Turn&) = delete;
  int begin();
  int end();

 private:
  hipStream_t s_;
  int which_;
  unsigned weight_, cap_;  // units asked for / units there are
  void* dev_ = nullptr;    // per-device turn state while the turn is open
};

}  // namespace ndmps
