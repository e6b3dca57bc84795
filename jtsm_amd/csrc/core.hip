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

/* ---- launch timing (bench.py's roofline leg) ---- */
void* jtsm_event_create(void) {
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) { jtsm::fail(JTSM_ELAUNCH, "hipEventCreate failed"); return nullptr; }
  return e;
}
int jtsm_event_record(void* event, void* stream) {
  JTSM_REQUIRE(event, "event_record: null event");
  JTSM_CHECK_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), reinterpret_cast<hipStream_t>(stream)));
  return JTSM_OK;
}
int jtsm_event_elapsed_ms(void* start, void* stop, float* ms) {
  JTSM_REQUIRE(start && stop && ms, "event_elapsed_ms: null argument");
  JTSM_CHECK_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(stop)));
  JTSM_CHECK_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
  return JTSM_OK;
}
void jtsm_event_destroy(void* event) {
  if (event) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(event));
}

}  // extern "C"
