// Bandwidth-bound helpers around the contraction kernels (all NHWC, fp32, 16 B per lane).
#include "common.h"

namespace jtsm {
namespace {

// g = dy where y > 0 else 0      (ReLU backward; y is the layer's OUTPUT)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float4* __restrict__ dy,
                                                       const float4* __restrict__ y,
                                                       float4* __restrict__ g, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (long)gridDim.x * blockDim.x) {
    const float4 a = dy[i], b = y[i];
    float4 r;
    r.x = b.x > 0.f ? a.x : 0.f;
    r.y = b.y > 0.f ? a.y : 0.f;
    r.z = b.z > 0.f ? a.z : 0.f;
    r.w = b.w > 0.f ? a.w : 0.f;
    g[i] = r;
  }
}
__global__ __launch_bounds__(256) void relu_bwd_tail(const float* __restrict__ dy,
                                                     const float* __restrict__ y,
                                                     float* __restrict__ g, long beg, long n) {
  const long i = beg + (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// out[c] = sum over rows of g[row][c]   (bias gradient).  Grid: (C/64 column groups, row slabs);
// each block sums its slab with 4 waves striding rows, then one atomic per column.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ g,
                                                          float* __restrict__ out, long rows,
                                                          int C, long rows_per_block) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc = 0.f;
  if (c < C)
    for (long r = r0 + wv; r < r1; r += 4) acc += g[r * C + c];
  part[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && c < C) atomicAdd(out + c, part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

int jtsm_relu_backward_f32(const float* dy, const float* y, float* g, long n, void* stream) {
  JTSM_REQUIRE(n >= 0, "relu_backward: negative size");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(dy && y && g, "relu_backward: null pointer");
  hipStream_t st = as_stream(stream);
  const bool vec = (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)g) & 15) == 0;
  const long n4 = vec ? n / 4 : 0;
  if (n4) {
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)dy,
                       (const float4*)y, (float4*)g, n4);
  }
  if (n4 * 4 < n)
    hipLaunchKernelGGL(relu_bwd_tail, dim3(ceil_div(n - n4 * 4, 256)), dim3(256), 0, st, dy, y, g,
                       n4 * 4, n);
  JTSM_CHECK_LAUNCH("relu_backward");
  return JTSM_OK;
}

int jtsm_channel_sum_f32(const float* g, float* out, long rows, int C, void* stream) {
  JTSM_REQUIRE(rows >= 0 && C > 0, "channel_sum: bad sizes");
  JTSM_REQUIRE(out, "channel_sum: null out");
  hipStream_t st = as_stream(stream);
  JTSM_CHECK_HIP(hipMemsetAsync(out, 0, (size_t)C * sizeof(float), st));
  if (rows == 0) return JTSM_OK;
  JTSM_REQUIRE(g, "channel_sum: null pointer");
  const int cgroups = ceil_div(C, 64);
  long slabs = 2048 / cgroups;
  if (slabs < 1) slabs = 1;
  if (slabs > (rows + 63) / 64) slabs = (rows + 63) / 64;
  const long rpb = (rows + slabs - 1) / slabs;
  hipLaunchKernelGGL(channel_sum_kernel, dim3(cgroups, (unsigned)((rows + rpb - 1) / rpb)), dim3(256),
                     0, st, g, out, rows, C, rpb);
  JTSM_CHECK_LAUNCH("channel_sum");
  return JTSM_OK;
}

}  // extern "C"
