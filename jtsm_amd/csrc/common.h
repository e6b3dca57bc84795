// Shared host-side helpers of libjtsm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/jtsm_hip.h"

namespace jtsm {

constexpr int kWave = 64;  // CDNA4 wavefront

// Thread-local message returned by jtsm_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Launch-error check that does not synchronise.
#define JTSM_CHECK_LAUNCH(what)                                                        \
  do {                                                                                 \
    hipError_t e_ = hipGetLastError();                                                 \
    if (e_ != hipSuccess) return ::jtsm::fail(JTSM_ELAUNCH, "%s: %s", what, hipGetErrorString(e_)); \
  } while (0)

#define JTSM_CHECK_HIP(expr)                                                           \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) return ::jtsm::fail(JTSM_ELAUNCH, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define JTSM_REQUIRE(cond, ...)                                                        \
  do {                                                                                 \
    if (!(cond)) return ::jtsm::fail(JTSM_EINVAL, __VA_ARGS__);                        \
  } while (0)

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace jtsm
