// fp16 tensors at the pooling boundary.
//
// The reference dispatches MOIPool on half as well (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// projects/WSL/wsl/layers/csrc/MOIPool/MOIPool_cuda.cu:400,415,484) and its Python layers feed half tensors to the
// align operators (detectron2/layers/roi_align_rotated.py:79-85 up-casts to fp32 and casts the result back).  These
// entry points take and return IEEE binary16 tensors (uint16_t bit patterns): values are widened to fp32 in a
// caller-supplied workspace, pooled by the fp32 kernels of this library, and rounded to fp16 once on the way out.
//   MOIPool: the result is EXACT (a maximum of fp16 values is one of them).  The roi corners are the reference's half
//            arithmetic: round(Half(x) * Half(scale)) with the product rounded to half (c10::Half operator*), which
//            is what the widening pass stores; the fp32 kernel then runs with scale 1.
//   ROIAlign / ROIAlignRotated: accumulate in fp32, round once (what the reference's Python up-cast path does).
#include "common.h"

namespace jtsm {
namespace {

__global__ __launch_bounds__(256) void widen_f16_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}
__global__ __launch_bounds__(256) void narrow_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = (_Float16)src[i];
}
// rois (M, cols): column 0 (batch index) widened; the others optionally pre-multiplied in half arithmetic
__global__ __launch_bounds__(256) void widen_rois_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, long n,
                                                         int cols, float scale, int premultiply) {
  const _Float16 hs = (_Float16)scale;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const _Float16 v = src[i];
    dst[i] = (premultiply && (i % cols) != 0) ? (float)(_Float16)((float)v * (float)hs) : (float)v;
  }
}

inline int blocks_for(long n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }
inline size_t up(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

size_t jtsm_pool_f16_workspace_bytes(long in_elems, long roi_elems, long out_elems, size_t extra) {
  return up((size_t)in_elems * 4) + up((size_t)roi_elems * 4) + up((size_t)out_elems * 4) + up(extra) + 256;
}

static int align_f16(int rotated, int backward, const uint16_t* a, const uint16_t* rois, uint16_t* out, int B, int C, int H,
                     int W, int M, float scale, int ph, int pw, int sr, int aligned, int layout, void* workspace,
                     size_t workspace_bytes, void* stream) {
  const int cols = rotated ? 6 : 5;
  const long map = (long)B * C * H * W, pooled = (long)M * C * ph * pw;
  const long n_in = backward ? pooled : map, n_out = backward ? map : pooled;
  JTSM_REQUIRE(workspace_bytes >= jtsm_pool_f16_workspace_bytes(n_in, (long)M * cols, n_out, 0) && workspace &&
               ((uintptr_t)workspace & 15) == 0, "pooling f16: workspace too small (jtsm_pool_f16_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* in32 = reinterpret_cast<float*>(w); w += up((size_t)n_in * 4);
  float* r32 = reinterpret_cast<float*>(w); w += up((size_t)M * cols * 4);
  float* out32 = reinterpret_cast<float*>(w);
  if (n_in) hipLaunchKernelGGL(widen_f16_kernel, dim3(blocks_for(n_in)), dim3(256), 0, st, reinterpret_cast<const _Float16*>(a), in32, n_in);
  if (M) hipLaunchKernelGGL(widen_rois_kernel, dim3(blocks_for((long)M * cols)), dim3(256), 0, st,
                            reinterpret_cast<const _Float16*>(rois), r32, (long)M * cols, cols, 1.f, 0);
  int rc;
  if (rotated)
    rc = backward ? jtsm_roi_align_rotated_backward_f32(in32, r32, out32, B, C, H, W, M, scale, ph, pw, sr, layout, stream)
                  : jtsm_roi_align_rotated_forward_f32(in32, r32, out32, B, C, H, W, M, scale, ph, pw, sr, layout, stream);
  else
    rc = backward ? jtsm_roi_align_backward_f32(in32, r32, out32, B, C, H, W, M, scale, ph, pw, sr, aligned, layout, stream)
                  : jtsm_roi_align_forward_f32(in32, r32, out32, B, C, H, W, M, scale, ph, pw, sr, aligned, layout, stream);
  if (rc) return rc;
  if (n_out) hipLaunchKernelGGL(narrow_f16_kernel, dim3(blocks_for(n_out)), dim3(256), 0, st, out32, reinterpret_cast<_Float16*>(out), n_out);
  JTSM_CHECK_LAUNCH("pooling f16");
  return JTSM_OK;
}

int jtsm_roi_align_forward_f16(const uint16_t* input, const uint16_t* rois, uint16_t* output, int B, int C, int H, int W,
                               int M, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                               int layout, void* workspace, size_t workspace_bytes, void* stream) {
  return align_f16(0, 0, input, rois, output, B, C, H, W, M, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned,
                   layout, workspace, workspace_bytes, stream);
}
int jtsm_roi_align_backward_f16(const uint16_t* grad, const uint16_t* rois, uint16_t* grad_input, int B, int C, int H, int W,
                                int M, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio, int aligned,
                                int layout, void* workspace, size_t workspace_bytes, void* stream) {
  return align_f16(0, 1, grad, rois, grad_input, B, C, H, W, M, spatial_scale, pooled_h, pooled_w, sampling_ratio, aligned,
                   layout, workspace, workspace_bytes, stream);
}
int jtsm_roi_align_rotated_forward_f16(const uint16_t* input, const uint16_t* rois, uint16_t* output, int B, int C, int H,
                                       int W, int M, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                       int layout, void* workspace, size_t workspace_bytes, void* stream) {
  return align_f16(1, 0, input, rois, output, B, C, H, W, M, spatial_scale, pooled_h, pooled_w, sampling_ratio, 0, layout,
                   workspace, workspace_bytes, stream);
}
int jtsm_roi_align_rotated_backward_f16(const uint16_t* grad, const uint16_t* rois, uint16_t* grad_input, int B, int C, int H,
                                        int W, int M, float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                        int layout, void* workspace, size_t workspace_bytes, void* stream) {
  return align_f16(1, 1, grad, rois, grad_input, B, C, H, W, M, spatial_scale, pooled_h, pooled_w, sampling_ratio, 0,
                   layout, workspace, workspace_bytes, stream);
}

size_t jtsm_moi_pool_f16_workspace_bytes(int B, int C, int H, int W, int M, int L, int pooled_h, int pooled_w) {
  return jtsm_pool_f16_workspace_bytes((long)B * C * H * W, (long)M * 5, (long)M * C * pooled_h * pooled_w,
                                       jtsm_moi_pool_workspace_bytes(B, H, W, M, L));
}

int jtsm_moi_pool_forward_f16(const uint16_t* input, const uint16_t* rois, const int32_t* oh_labels,
                              const int32_t* superpixels, uint16_t* output, int32_t* argmax, void* workspace,
                              size_t workspace_bytes, int B, int C, int H, int W, int M, int L, int Hs, int Ws,
                              float spatial_scale, int pooled_h, int pooled_w, int layout, void* stream) {
  const long map = (long)B * C * H * W, pooled = (long)M * C * pooled_h * pooled_w;
  JTSM_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 &&
               workspace_bytes >= jtsm_moi_pool_f16_workspace_bytes(B, C, H, W, M, L, pooled_h, pooled_w),
               "moi_pool f16: workspace too small (jtsm_moi_pool_f16_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* in32 = reinterpret_cast<float*>(w); w += up((size_t)map * 4);
  float* r32 = reinterpret_cast<float*>(w); w += up((size_t)M * 5 * 4);
  float* out32 = reinterpret_cast<float*>(w); w += up((size_t)pooled * 4);
  if (map) hipLaunchKernelGGL(widen_f16_kernel, dim3(blocks_for(map)), dim3(256), 0, st, reinterpret_cast<const _Float16*>(input), in32, map);
  // corners: Half(x) * Half(scale) rounded to half, as the reference's half kernel computes them; scale 1 below
  if (M) hipLaunchKernelGGL(widen_rois_kernel, dim3(blocks_for((long)M * 5)), dim3(256), 0, st,
                            reinterpret_cast<const _Float16*>(rois), r32, (long)M * 5, 5, spatial_scale, 1);
  int rc = jtsm_moi_pool_forward_f32(in32, r32, oh_labels, superpixels, out32, argmax, w, B, C, H, W, M, L, Hs, Ws, 1.0f,
                                     pooled_h, pooled_w, layout, stream);
  if (rc) return rc;
  if (pooled) hipLaunchKernelGGL(narrow_f16_kernel, dim3(blocks_for(pooled)), dim3(256), 0, st, out32, reinterpret_cast<_Float16*>(output), pooled);
  JTSM_CHECK_LAUNCH("moi_pool f16");
  return JTSM_OK;
}

int jtsm_moi_pool_backward_f16(const uint16_t* grad, const uint16_t* rois, const int32_t* argmax, uint16_t* grad_input,
                               void* workspace, size_t workspace_bytes, int B, int C, int H, int W, int M, int pooled_h,
                               int pooled_w, int layout, void* stream) {
  const long map = (long)B * C * H * W, pooled = (long)M * C * pooled_h * pooled_w;
  JTSM_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 &&
               workspace_bytes >= jtsm_pool_f16_workspace_bytes(pooled, (long)M * 5, map, 0),
               "moi_pool backward f16: workspace too small (jtsm_pool_f16_workspace_bytes)");
  hipStream_t st = as_stream(stream);
  char* w = reinterpret_cast<char*>(workspace);
  float* g32 = reinterpret_cast<float*>(w); w += up((size_t)pooled * 4);
  float* r32 = reinterpret_cast<float*>(w); w += up((size_t)M * 5 * 4);
  float* gin32 = reinterpret_cast<float*>(w);
  if (pooled) hipLaunchKernelGGL(widen_f16_kernel, dim3(blocks_for(pooled)), dim3(256), 0, st, reinterpret_cast<const _Float16*>(grad), g32, pooled);
  if (M) hipLaunchKernelGGL(widen_rois_kernel, dim3(blocks_for((long)M * 5)), dim3(256), 0, st,
                            reinterpret_cast<const _Float16*>(rois), r32, (long)M * 5, 5, 1.f, 0);
  int rc = jtsm_moi_pool_backward_f32(g32, r32, argmax, gin32, B, C, H, W, M, pooled_h, pooled_w, layout, stream);
  if (rc) return rc;
  if (map) hipLaunchKernelGGL(narrow_f16_kernel, dim3(blocks_for(map)), dim3(256), 0, st, gin32, reinterpret_cast<_Float16*>(grad_input), map);
  JTSM_CHECK_LAUNCH("moi_pool backward f16");
  return JTSM_OK;
}

}  // extern "C"
