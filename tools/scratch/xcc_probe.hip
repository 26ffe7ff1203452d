// Which XCD does workgroup i land on, and what does a flag hand-off between two workgroups cost through the L2 of
// one XCD (sc0 loads, L2 atomics) and through the memory side (agent scope)?  Every wait is bounded.
// hipcc -O3 --offload-arch=gfx950 xcc_probe.hip -o _bin/xcc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void where(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15);
}
__device__ __forceinline__ unsigned long long ld_sc0(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_plain(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_inv(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_nt(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ unsigned long long ld_scalar(const unsigned long long* p) {
  unsigned long long v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
// workgroups a and b bounce a counter: mode 0 = L2 atomics + sc0 loads, 1 = agent scope, 2 = L2 atomics + plain loads
__global__ void pingpong(unsigned long long* flag, int a, int b, int rounds, int mode, long long* ticks, int* fail) {
  const int me = blockIdx.x;
  if ((me != a && me != b) || threadIdx.x != 0) return;
  const int side = me == a ? 0 : 1;
  const long long t0 = wall_clock64();
  for (int r = 0; r < rounds; ++r) {
    const unsigned long long want = 2ull * r + side;  // a moves on even values, b on odd ones
    const long long w0 = wall_clock64();
    for (;;) {
      unsigned long long v = mode == 1   ? __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                             : mode == 0 ? ld_sc0(flag)
                             : mode == 2 ? ld_plain(flag)
                             : mode == 3 ? ld_inv(flag)
                             : mode == 5 ? ld_nt(flag)
                             : mode == 6 ? ld_scalar(flag)
                                         : __hip_atomic_fetch_or(flag, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (v >= want) break;
      if (wall_clock64() - w0 > 20000000LL) {  // 0.2 s
        *fail = 1 + r;
        return;
      }
    }
    if (mode == 1) __hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (side == 0) *ticks = wall_clock64() - t0;
}
int main() {
  int* out;
  hipMalloc(&out, 4096 * 4);
  where<<<64, 64>>>(out);
  int h[64];
  hipMemcpy(h, out, 64 * 4, hipMemcpyDeviceToHost);
  printf("XCC id of workgroups 0..63:");
  for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
  printf("\n");
  unsigned long long* flag;
  long long* ticks;
  int* fail;
  hipMalloc(&flag, 4096);
  hipMalloc(&ticks, 8);
  hipMalloc(&fail, 4);
  const int rounds = 2000;
  for (int mode = 0; mode < 7; ++mode)
    for (int partner : {8, 1}) {  // same XCD (ids congruent mod 8) / different XCDs
      hipMemset(flag, 0, 4096);
      hipMemset(fail, 0, 4);
      hipMemset(ticks, 0, 8);
      pingpong<<<16, 64>>>(flag, 0, partner, rounds, mode, ticks, fail);
      hipDeviceSynchronize();
      long long t;
      int f;
      hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
      hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
      printf("mode %d (%s), workgroups 0 and %d (XCC %d / %d): %s, %.3f us per hand-off\n", mode,
             mode == 0 ? "L2 atomics + sc0 loads" : mode == 1 ? "agent scope" : mode == 2 ? "L2 atomics + plain loads" : mode == 3 ? "L2 atomics + buffer_inv sc0 + plain load" : mode == 4 ? "L2 atomics, polled by an atomic or" : mode == 5 ? "L2 atomics + nt loads" : "L2 atomics + scalar glc loads", partner, h[0], h[partner],
             f ? "TIMED OUT" : "ok", f ? 0.0 : t / 100.0 / (2.0 * rounds));
    }
  return 0;
}
