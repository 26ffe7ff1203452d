// Error plumbing and trivial queries of libndmps_hip.so.
#include <stdarg.h>

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

extern "C" int ndmps_version(void) { return 100; }

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
