// Library-wide plumbing of libjtsm_hip.so: error reporting, version, device probe.
#include "common.h"

namespace jtsm {

char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace jtsm

extern "C" {

const char* jtsm_last_error(void) { return jtsm::error_buffer(); }

const char* jtsm_version(void) { return "jtsm_hip 0.1 gfx950"; }

int jtsm_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    jtsm::fail(JTSM_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}

}  // extern "C"
