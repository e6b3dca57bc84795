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

// ordered compaction of a per-thread flag over the 256 threads of the workgroup: returns this thread's slot (if
// flagged) and the total; two barriers.
__device__ __forceinline__ int compact256(bool flag, int* __restrict__ wave_count, int& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  __syncthreads();                       // previous readers of wave_count are done
  if (lane == 0) wave_count[wv] = __popcll(bal);
  __syncthreads();
  int off = 0;
  total = 0;
  for (int w = 0; w < 4; ++w) { if (w < wv) off += wave_count[w]; total += wave_count[w]; }
  return off + __popcll(bal & ((1ull << lane) - 1ull));
}


}  // namespace jtsm
