// Error plumbing and trivial queries of libndmps_hip.so.
#include <stdarg.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "common.h"

namespace ndmps {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ndmps

extern "C" int ndmps_version(void) { return 105; }  // round 4: direct solver for any k <= n <= 4096, potrf, per-slice SSIM, team slots; 105: sweep in two halves, group-wide permute / DCT / scaling

extern "C" const char* ndmps_last_error(void) { return ndmps::g_err; }

extern "C" int ndmps_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    ndmps::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return NDMPS_EHIP;
  }
  return n;
}

// ---------------------------------------------------------------------------------- streams
// Concurrent volume groups need streams that sit on DIFFERENT hardware queues: the runtime maps
// streams onto a handful of queues (4 by default) when a stream is first used, and two groups on
// one queue serialise (measured: 4 groups x 8 volumes 80 ms on distinct queues, 117 ms with one
// shared pair, 156 ms all on one).  The mapping is not queryable, so it is measured: a bounded spin
// kernel is launched alternately on two streams; sharing a queue doubles the wall time.
namespace ndmps {
__global__ void spin_kernel(long long ticks) {  // wall_clock64: constant 100 MHz; always terminates
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {
  }
}

static double spin_ms(hipStream_t a, hipStream_t b, int reps, long long ticks) {
  (void)hipStreamSynchronize(a);
  if (b) (void)hipStreamSynchronize(b);
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, a, ticks);
    if (b) hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, b, ticks);
  }
  (void)hipStreamSynchronize(a);
  if (b) (void)hipStreamSynchronize(b);
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}  // namespace ndmps

extern "C" int ndmps_streams_create(int n, void** h_streams, int* n_independent) {
  using namespace ndmps;
  NDMPS_REQUIRE(n >= 1 && n <= 64 && h_streams, "n=%d outside [1, 64] or NULL output", n);
  constexpr int kReps = 6;
  constexpr long long kTicks = 10000;  // 100 us
  std::vector<hipStream_t> chosen, spare;
  for (int tries = 0; tries < 6 * n && (int)chosen.size() < n; ++tries) {
    hipStream_t s = nullptr;
    NDMPS_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, s, 1LL);  // first use binds the queue
    const double alone = spin_ms(s, nullptr, kReps, kTicks);
    bool independent = true;
    for (hipStream_t c : chosen) {
      // 2.0 = same queue; ~1.3 = two queues of one dispatch pipe (seen with GPU_MAX_HW_QUEUES = 8)
      if (spin_ms(s, c, kReps, kTicks) > 1.2 * alone) {
        independent = false;
        break;
      }
    }
    (independent ? chosen : spare).push_back(s);
  }
  NDMPS_CHECK_HIP(hipGetLastError());
  const int found = (int)chosen.size();
  while ((int)chosen.size() < n && !spare.empty()) {  // fewer queues than groups: share them
    chosen.push_back(spare.back());
    spare.pop_back();
  }
  while ((int)chosen.size() < n) {
    hipStream_t s = nullptr;
    NDMPS_CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    chosen.push_back(s);
  }
  for (hipStream_t s : spare) (void)hipStreamDestroy(s);
  for (int i = 0; i < n; ++i) h_streams[i] = chosen[i];
  if (n_independent) *n_independent = found;
  return NDMPS_OK;
}

extern "C" int ndmps_streams_destroy(int n, void* const* h_streams) {
  NDMPS_REQUIRE(n >= 0 && (n == 0 || h_streams), "bad stream list");
  for (int i = 0; i < n; ++i)
    if (h_streams[i]) NDMPS_CHECK_HIP(hipStreamDestroy((hipStream_t)h_streams[i]));
  return NDMPS_OK;
}

// ---------------------------------------------------------------------------------- launch spans
// bench.py's roofline needs the average duration of the dominant kernel's launches INSIDE the timed region,
// on the stream they are launched on.  When enabled, a kernel sequence brackets itself with two HIP events
// (ndmps::span_begin / span_end) and files them under a slot; ndmps_profile_collect() sums the elapsed
// times.  Disabled (the default) nothing is recorded and the calls cost one relaxed load.
namespace ndmps {
namespace {
struct Span {
  hipEvent_t a, b;
  int slot;
  int64_t launches, bytes;
};
std::mutex g_span_mu;
std::vector<Span> g_spans;
std::vector<hipEvent_t> g_free_events;
volatile int g_profile_on = 0;

hipEvent_t take_event() {
  if (!g_free_events.empty()) {
    hipEvent_t e = g_free_events.back();
    g_free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

void* span_begin(hipStream_t s) {
  if (!g_profile_on) return nullptr;
  std::lock_guard<std::mutex> lock(g_span_mu);
  hipEvent_t e = take_event();
  if (e && hipEventRecord(e, s) != hipSuccess) e = nullptr;
  return e;
}

void span_end(void* begin, hipStream_t s, int slot, int64_t launches, int64_t bytes) {
  if (!begin) return;
  std::lock_guard<std::mutex> lock(g_span_mu);
  hipEvent_t e = take_event();
  if (!e || hipEventRecord(e, s) != hipSuccess) return;
  g_spans.push_back(Span{(hipEvent_t)begin, e, slot, launches, bytes});
}
}  // namespace ndmps

extern "C" int ndmps_profile_enable(int on) {
  ndmps::g_profile_on = on ? 1 : 0;
  return NDMPS_OK;
}

// Sums (and clears) the spans of `slot` recorded so far: total device milliseconds between their events,
// kernel launches and algorithmic bytes they covered.  Synchronises on the recorded events.
extern "C" int ndmps_profile_collect(int slot, double* h_ms, int64_t* h_launches, int64_t* h_bytes) {
  using namespace ndmps;
  NDMPS_REQUIRE(h_ms && h_launches && h_bytes, "NULL profile output");
  std::lock_guard<std::mutex> lock(g_span_mu);
  double ms = 0.0;
  int64_t launches = 0, bytes = 0;
  std::vector<Span> keep;
  for (const Span& sp : g_spans) {
    if (sp.slot != slot) {
      keep.push_back(sp);
      continue;
    }
    float t = 0.f;
    NDMPS_CHECK_HIP(hipEventSynchronize(sp.b));
    NDMPS_CHECK_HIP(hipEventElapsedTime(&t, sp.a, sp.b));
    ms += t;
    launches += sp.launches;
    bytes += sp.bytes;
    g_free_events.push_back(sp.a);
    g_free_events.push_back(sp.b);
  }
  g_spans.swap(keep);
  *h_ms = ms;
  *h_launches = launches;
  *h_bytes = bytes;
  return NDMPS_OK;
}

// ---------------------------------------------------------------------------------------------- device-side turns
// Kernels that fill the GPU alone (the register-resident tridiagonalisation, the big Gram launches) take turns
// between streams ON THE DEVICE: a one-thread kernel in front spins on a lock word until it owns it, a one-thread
// kernel behind gives it back.  The spinner holds one wave slot and a handful of registers, so the owner's kernels
// always fit beside it, and whichever stream gets there first runs first (an event chain would follow the order in
// which the HOST enqueues).
//
// Streams may share a hardware queue (the runtime has 4 by default; ndmps_streams_create hands out shared ones when
// more groups are asked for, and a caller's own stream is never measured).  Packets of one queue run in order, so a
// spinner that landed BETWEEN another stream's acquire and release in a shared queue would wait for a release that
// sits behind it.  Hence a turn [acquire, kernels, release] is SUBMITTED ATOMICALLY: one host mutex per device is held
// from turn begin to turn end, for every kind of turn (two mutexes would still allow a cycle through two locks).  In
// any queue an owner's release then has only its own acquire and its own kernels in front of it, which end by
// themselves, so every spinner is released in finite time whatever the stream-to-queue map is.
// Every wait is bounded (kTurnSpinTicks): a spinner that gave up takes its units regardless.
namespace {
constexpr long long kTurnSpinTicks = 300000000LL;  // 3 s of the 100 MHz wall clock
// The lock word counts the UNITS in use; a turn asks for `weight` units of `cap` (team launches: 2 of 2 when they fill
// both workgroup slots of every CU, 1 of 2 when they fill one -- two such launches fit the GPU together and leave a
// slot per CU to everybody else; Gram launches: 1 of 1).  A spinner that gave up after kTurnSpinTicks takes its units
// regardless, so every release subtracts what its acquire added.
__global__ void turn_acquire_kernel(unsigned* __restrict__ lock, unsigned weight, unsigned cap) {
  const long long t0 = wall_clock64();
  for (;;) {
    const unsigned c = __hip_atomic_load(lock, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c + weight <= cap && atomicCAS(lock, c, c + weight) == c) return;
    if (wall_clock64() - t0 > kTurnSpinTicks) {
      atomicAdd(lock, weight);
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}
__global__ void turn_release_kernel(unsigned* __restrict__ lock, unsigned weight) { atomicSub(lock, weight); }

struct TurnDevice {
  unsigned* base = nullptr;  // 256 bytes per device, once: the only allocation outside ndmps_plan_create
  std::mutex submit;         // held while a turn is being enqueued
};
TurnDevice g_turn[64];
std::mutex g_turn_alloc;

int turn_device(TurnDevice** out) {
  int dev = 0;
  NDMPS_CHECK_HIP(hipGetDevice(&dev));
  NDMPS_REQUIRE(dev >= 0 && dev < 64, "device index %d outside [0, 64)", dev);
  TurnDevice& td = g_turn[dev];
  {
    std::lock_guard<std::mutex> guard(g_turn_alloc);
    if (!td.base) {
      NDMPS_CHECK_HIP(hipMalloc((void**)&td.base, 256));
      NDMPS_CHECK_HIP(hipMemset(td.base, 0, 256));
    }
  }
  *out = &td;
  return NDMPS_OK;
}
}  // namespace

namespace ndmps {
Turn::~Turn() { (void)end(); }

int Turn::begin() {
  if (dev_) return NDMPS_OK;
  NDMPS_REQUIRE(which_ >= 0 && which_ < 4, "turn kind %d outside [0, 4)", which_);
  TurnDevice* td = nullptr;
  NDMPS_TRY(turn_device(&td));
  td->submit.lock();
  dev_ = td;
  hipLaunchKernelGGL(turn_acquire_kernel, dim3(1), dim3(1), 0, s_, td->base + 16 * which_, weight_, cap_);  // words 64 B apart
  if (hipGetLastError() != hipSuccess) {
    dev_ = nullptr;
    td->submit.unlock();
    set_error("turn_acquire_kernel launch failed");
    return NDMPS_EHIP;
  }
  return NDMPS_OK;
}

int Turn::end() {
  if (!dev_) return NDMPS_OK;
  TurnDevice* td = (TurnDevice*)dev_;
  dev_ = nullptr;
  hipLaunchKernelGGL(turn_release_kernel, dim3(1), dim3(1), 0, s_, td->base + 16 * which_, weight_);
  const hipError_t e = hipGetLastError();
  td->submit.unlock();
  if (e != hipSuccess) {
    set_error("turn_release_kernel launch failed: %s", hipGetErrorString(e));
    return NDMPS_EHIP;
  }
  return NDMPS_OK;
}
}  // namespace ndmps
