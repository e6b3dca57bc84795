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


// Ordered compaction of a per-thread flag over the NW wavefronts of the workgroup: this thread's slot (if flagged) and
// the total; two barriers.
template <int NW>
__device__ __forceinline__ int compact_wg(bool flag, int* __restrict__ wave_count, int& total) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(flag);
  __syncthreads();                       // previous readers of wave_count are done
  if (lane == 0) wave_count[wv] = __popcll(bal);
  __syncthreads();
  int off = 0;
  total = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) { const int n = wave_count[w]; if (w < wv) off += n; total += n; }
  return off + __popcll(bal & ((1ull << lane) - 1ull));
}

// The launch plan, from the census (one workgroup of 1024): plan[0] = the census maximum (fallback test), plan[1] = the
// number of work entries, plan[2 ...] = `fan` entries (tile * fan + k) for every tile whose census is at least
// `heavy_min`, heaviest first (four load classes relative to the maximum, index order inside a class — the busy workgroups stride this list, so
// the few tiles under a pile of proposals start first instead of ending the launch).
// `order` (optional, `total` entries): ALL tiles by descending load — tiles at or above heavy_min first (another kernel's:
// their workgroups leave at once), then the lighter ones in 64 load classes, heaviest class first, empty tiles last.  A
// one-workgroup-per-tile launch that maps blockIdx through it starts its longest tiles first instead of wherever the
// map's layout put them (the MOIPool backward's light-tile gather was one long tail: 230 us for ~70 us of work per slot).
static __global__ __launch_bounds__(1024) void tile_plan_kernel(const int* __restrict__ census, int total,
                                                                int* __restrict__ plan, int heavy_min, int fan,
                                                                int* __restrict__ order = nullptr) {
  __shared__ int wave_count[16];
  __shared__ int part[16];
  __shared__ int cls_count[66];
  const int t = threadIdx.x;
  if (order) {     // counting sort by load class (the order inside a class is the atomics'; only the schedule sees it)
    auto cls_of = [&](int v) {
      if (v >= heavy_min) return 0;
      if (v <= 0) return 65;
      return 64 - (int)min(63L, (long)v * 64 / max(heavy_min, 1));
    };
    if (t < 66) cls_count[t] = 0;
    __syncthreads();
    for (int i = t; i < total; i += 1024) atomicAdd(&cls_count[cls_of(census[i])], 1);
    __syncthreads();
    if (t == 0) {
      int acc = 0;
      for (int c = 0; c < 66; ++c) { const int v = cls_count[c]; cls_count[c] = acc; acc += v; }
    }
    __syncthreads();
    for (int i = t; i < total; i += 1024) order[atomicAdd(&cls_count[cls_of(census[i])], 1)] = i;
    __syncthreads();
  }
  int m = 0;
  for (int i = t; i < total; i += 1024) m = max(m, census[i]);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((t & 63) == 0) part[t >> 6] = m;
  __syncthreads();
  int mm = 0;
  for (int w = 0; w < 16; ++w) mm = max(mm, part[w]);
  if (t == 0) plan[0] = mm;
  int filled = 0;
  if (mm >= heavy_min) {   // (uniform)
    // load classes relative to the heaviest tile: [max / 2, ..), [max / 8, max / 2), [max / 32, max / 8), [heavy_min, max / 32)
    const long h = heavy_min, t0 = max(h, (long)mm / 2), t1 = max(h, (long)mm / 8), t2 = max(h, (long)mm / 32);
    const long lo[4] = {t0, t1, t2, h}, hi[4] = {1L << 40, t0, t1, t2};
    for (int cls = 0; cls < 4; ++cls)
      for (int base = 0; base < total; base += 1024) {
        const int i = base + t;
        const long v = i < total ? census[i] : 0;
        const bool take = v >= lo[cls] && v < hi[cls];
        int cnt;
        const int slot = compact_wg<16>(take, wave_count, cnt);
        if (take)
          for (int k = 0; k < fan; ++k) plan[2 + (filled + slot) * fan + k] = i * fan + k;
        filled += cnt;
      }
  }
  if (t == 0) plan[1] = filled * fan;
}


}  // namespace jtsm
