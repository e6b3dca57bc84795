// ROIAlign / ROIAlignRotated for gfx950 — forward, backward and the sample-table dump.
//
// Replaces ROIAlign_forward/backward (detectron2/layers/csrc/ROIAlign/ROIAlign.h:7-27) and
// ROIAlignRotated_forward/backward (.../ROIAlignRotated/ROIAlignRotated.h:7-27).
//
// NHWC path (what the MI355X model runs): ONE WAVEFRONT PER (roi, bin).  The 64 lanes first
// compute the bin's bilinear samples in parallel (lane s owns sample s of the gh x gw grid:
// 4 tap positions + 4 weights), then the samples are replayed one at a time: the owner's
// taps are broadcast with wavefront shuffles and every lane gathers ITS channels of the four
// neighbours — VEC contiguous floats per lane, so a 256-channel row is one coalesced 1 KiB
// access per tap.  HBM/L2 traffic per bin: 4*S rows read (neighbouring bins hit L2), one
// row written.  The per-sample accumulation order equals the CPU reference's (iy outer, ix
// inner, four taps as one expression) and contraction is off, so outputs are bit-identical
// to the oracle, not only within tolerance.
//
// NCHW path (the reference's own layout, kept so the FFI is a true drop-in): one thread per
// output element, lanes running over (pw, ph, c) like the reference kernel.
//
// Backward: same decomposition, scatter with float atomics (global_atomic_add_f32 /
// _f64); in NHWC each wave instruction adds 256 contiguous bytes, the shape the memory-side
// atomic units run at full rate on (MI355X_MICROARCH.md "Global float atomics").
#include "common.h"
#include "roi_geometry.h"

namespace jtsm {
namespace {

#pragma clang fp contract(off)

template <typename T, int VEC> struct VecOf;
template <> struct VecOf<float, 4> { using type = float4; };
template <> struct VecOf<float, 1> { using type = float; };
template <> struct VecOf<double, 2> { using type = double2; };
template <> struct VecOf<double, 1> { using type = double; };

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T* p, T (&v)[VEC]) {
  using V = typename VecOf<T, VEC>::type;
  const V x = *reinterpret_cast<const V*>(p);
  const T* e = reinterpret_cast<const T*>(&x);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = e[i];
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T* p, const T (&v)[VEC]) {
  using V = typename VecOf<T, VEC>::type;
  V x;
  T* e = reinterpret_cast<T*>(&x);
#pragma unroll
  for (int i = 0; i < VEC; ++i) e[i] = v[i];
  *reinterpret_cast<V*>(p) = x;
}

// ---------------------------------------------------------------- NHWC forward
template <typename T, int VEC, bool ROT>
__global__ __launch_bounds__(256) void align_fwd_nhwc(const T* __restrict__ in,
                                                      const T* __restrict__ rois,
                                                      T* __restrict__ out, int C, int H, int W,
                                                      int M, T scale, int PH, int PW, int sr,
                                                      int aligned,
                                                      const int* __restrict__ roi_level, int level) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int nbins = PH * PW;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= (long)M * nbins) return;
  if (roi_level && roi_level[wave / nbins] != level) return;  // FPN: this roi lives on another level  // whole wave leaves together
  const int n = (int)(wave / nbins);
  const int bin = (int)(wave - (long)n * nbins);
  const int ph = bin / PW, pw = bin - ph * PW;

  const RoiGeom<T> g = roi_geometry<T, ROT>(rois, n, scale, PH, PW, sr, aligned != 0);
  const int S = (g.gh > 0 && g.gw > 0) ? g.gh * g.gw : 0;
  const int cells = g.gh * g.gw;
  const T count = (T)(cells > 1 ? cells : 1);
  const T* __restrict__ plane = in + (size_t)g.b * H * W * C;
  T* __restrict__ orow = out + ((size_t)n * nbins + bin) * C;

  for (int cb = 0; cb < C; cb += 64 * VEC) {
    const int c = cb + lane * VEC;
    const bool live = c < C;
    T acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = (T)0;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      Tap<T> mine;
      if (s < S) {
        const int iy = s / g.gw;
        mine = sample_tap<T, ROT>(g, H, W, ph, pw, iy, s - iy * g.gw);
      } else {
        mine.pos[0] = mine.pos[1] = mine.pos[2] = mine.pos[3] = -1;
        mine.w[0] = mine.w[1] = mine.w[2] = mine.w[3] = (T)0;
      }
      const int cnt = (S - s0) < 64 ? (S - s0) : 64;
      for (int j = 0; j < cnt; ++j) {
        const int p0 = __shfl(mine.pos[0], j);
        if (p0 < 0) continue;  // wave-uniform: the sample lies outside the map
        const int p1 = __shfl(mine.pos[1], j), p2 = __shfl(mine.pos[2], j),
                  p3 = __shfl(mine.pos[3], j);
        const T w0 = __shfl(mine.w[0], j), w1 = __shfl(mine.w[1], j),
                w2 = __shfl(mine.w[2], j), w3 = __shfl(mine.w[3], j);
        if (live) {
          T a[VEC], b[VEC], d[VEC], e[VEC];
          load_vec<T, VEC>(plane + (size_t)p0 * C + c, a);
          load_vec<T, VEC>(plane + (size_t)p1 * C + c, b);
          load_vec<T, VEC>(plane + (size_t)p2 * C + c, d);
          load_vec<T, VEC>(plane + (size_t)p3 * C + c, e);
#pragma unroll
          for (int v = 0; v < VEC; ++v) acc[v] += w0 * a[v] + w1 * b[v] + w2 * d[v] + w3 * e[v];
        }
      }
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) acc[v] /= count;
      store_vec<T, VEC>(orow + c, acc);
    }
  }
}

// ---------------------------------------------------------------- NHWC backward
template <typename T, int VEC, bool ROT>
__global__ __launch_bounds__(256) void align_bwd_nhwc(const T* __restrict__ grad,
                                                      const T* __restrict__ rois,
                                                      T* __restrict__ gin, int C, int H, int W,
                                                      int M, T scale, int PH, int PW, int sr,
                                                      int aligned,
                                                      const int* __restrict__ roi_level, int level) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63;
  const int nbins = PH * PW;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= (long)M * nbins) return;
  if (roi_level && roi_level[wave / nbins] != level) return;  // FPN: this roi lives on another level
  const int n = (int)(wave / nbins);
  const int bin = (int)(wave - (long)n * nbins);
  const int ph = bin / PW, pw = bin - ph * PW;

  const RoiGeom<T> g = roi_geometry<T, ROT>(rois, n, scale, PH, PW, sr, aligned != 0);
  const int S = (g.gh > 0 && g.gw > 0) ? g.gh * g.gw : 0;
  const T count = (T)(g.gh * g.gw);
  T* __restrict__ plane = gin + (size_t)g.b * H * W * C;
  const T* __restrict__ grow = grad + ((size_t)n * nbins + bin) * C;

  for (int cb = 0; cb < C; cb += 64 * VEC) {
    const int c = cb + lane * VEC;
    const bool live = c < C;
    T go[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) go[v] = (T)0;
    if (live) load_vec<T, VEC>(grow + c, go);
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      Tap<T> mine;
      if (s < S) {
        const int iy = s / g.gw;
        mine = sample_tap<T, ROT>(g, H, W, ph, pw, iy, s - iy * g.gw);
      } else {
        mine.pos[0] = mine.pos[1] = mine.pos[2] = mine.pos[3] = -1;
        mine.w[0] = mine.w[1] = mine.w[2] = mine.w[3] = (T)0;
      }
      const int cnt = (S - s0) < 64 ? (S - s0) : 64;
      for (int j = 0; j < cnt; ++j) {
        const int p0 = __shfl(mine.pos[0], j);
        if (p0 < 0) continue;
        int p[4] = {p0, __shfl(mine.pos[1], j), __shfl(mine.pos[2], j), __shfl(mine.pos[3], j)};
        T w[4] = {__shfl(mine.w[0], j), __shfl(mine.w[1], j), __shfl(mine.w[2], j),
                  __shfl(mine.w[3], j)};
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            T* dst = plane + (size_t)p[q] * C + c;
#pragma unroll
            for (int v = 0; v < VEC; ++v) atomicAdd(dst + v, go[v] * w[q] / count);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- NCHW (reference layout)
template <typename T, bool ROT>
__global__ __launch_bounds__(256) void align_fwd_nchw(const T* __restrict__ in,
                                                      const T* __restrict__ rois,
                                                      T* __restrict__ out, int C, int H, int W,
                                                      long total, T scale, int PH, int PW,
                                                      int sr, int aligned) {
#pragma clang fp contract(off)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / PW / PH) % C);
    const int n = (int)(idx / PW / PH / C);
    const RoiGeom<T> g = roi_geometry<T, ROT>(rois, n, scale, PH, PW, sr, aligned != 0);
    const int cells = g.gh * g.gw;
    const T count = (T)(cells > 1 ? cells : 1);
    const T* __restrict__ plane = in + ((size_t)g.b * C + c) * H * W;
    T acc = (T)0;
    for (int iy = 0; iy < g.gh; ++iy)
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap<T> t = sample_tap<T, ROT>(g, H, W, ph, pw, iy, ix);
        if (t.pos[0] < 0) continue;
        acc += t.w[0] * plane[t.pos[0]] + t.w[1] * plane[t.pos[1]] + t.w[2] * plane[t.pos[2]] +
               t.w[3] * plane[t.pos[3]];
      }
    out[idx] = acc / count;
  }
}

template <typename T, bool ROT>
__global__ __launch_bounds__(256) void align_bwd_nchw(const T* __restrict__ grad,
                                                      const T* __restrict__ rois,
                                                      T* __restrict__ gin, int C, int H, int W,
                                                      long total, T scale, int PH, int PW,
                                                      int sr, int aligned) {
#pragma clang fp contract(off)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW);
    const int ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / PW / PH) % C);
    const int n = (int)(idx / PW / PH / C);
    const RoiGeom<T> g = roi_geometry<T, ROT>(rois, n, scale, PH, PW, sr, aligned != 0);
    const T count = (T)(g.gh * g.gw);
    T* __restrict__ plane = gin + ((size_t)g.b * C + c) * H * W;
    const T go = grad[idx];
    for (int iy = 0; iy < g.gh; ++iy)
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap<T> t = sample_tap<T, ROT>(g, H, W, ph, pw, iy, ix);
        if (t.pos[0] < 0) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) atomicAdd(plane + t.pos[q], go * t.w[q] / count);
      }
  }
}

// ---------------------------------------------------------------- sample-table dump
template <typename T, bool ROT>
__global__ void sample_table_kernel(const T* __restrict__ rois, int M, int H, int W, T scale,
                                    int PH, int PW, int sr, int aligned, int* __restrict__ grid,
                                    int* __restrict__ pos, T* __restrict__ w, int cap) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)M * cap) return;
  const int m = (int)(idx / cap), s = (int)(idx - (long)m * cap);
  const RoiGeom<T> g = roi_geometry<T, ROT>(rois, m, scale, PH, PW, sr, aligned != 0);
  if (s == 0) {
    grid[2 * m] = g.gh;
    grid[2 * m + 1] = g.gw;
  }
  int* p = pos + idx * 4;
  T* ww = w + idx * 4;
  const long per = (g.gh > 0 && g.gw > 0) ? (long)g.gh * g.gw : 0;
  if (s >= per * PH * PW) {
    for (int q = 0; q < 4; ++q) { p[q] = -2; ww[q] = (T)0; }  // -2: beyond this roi's table
    return;
  }
  const int bin = (int)(s / per), r = (int)(s - bin * per);
  const int iy = r / g.gw, ix = r - iy * g.gw;
  const Tap<T> t = sample_tap<T, ROT>(g, H, W, bin / PW, bin % PW, iy, ix);
  for (int q = 0; q < 4; ++q) { p[q] = t.pos[0] < 0 ? -1 : t.pos[q]; ww[q] = t.w[q]; }
}

template <typename T> struct WideVec;
template <> struct WideVec<float> { static constexpr int value = 4; };
template <> struct WideVec<double> { static constexpr int value = 2; };

template <typename T, bool ROT>
int launch_forward(const T* in, const T* rois, T* out, int B, int C, int H, int W, int M,
                   T scale, int PH, int PW, int sr, int aligned, int layout, void* stream,
                   const int* roi_level = nullptr, int level = 0) {
  JTSM_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && M >= 0 && PH > 0 && PW > 0,
               "roi_align: negative size (B=%d C=%d H=%d W=%d M=%d PH=%d PW=%d)", B, C, H, W, M,
               PH, PW);
  JTSM_REQUIRE(layout == JTSM_NCHW || layout == JTSM_NHWC, "roi_align: unknown layout %d", layout);
  if ((long)M * C * PH * PW == 0) return JTSM_OK;  // empty output returns early (ROIAlign_cuda.cu:343)
  JTSM_REQUIRE(in && rois && out, "roi_align: null pointer");
  JTSM_REQUIRE(B > 0 && H > 0 && W > 0, "roi_align: empty feature map with %d rois", M);
  hipStream_t st = as_stream(stream);
  if (layout == JTSM_NHWC) {
    const long waves = (long)M * PH * PW;
    const int blocks = ceil_div(waves, 4);
    constexpr int V = WideVec<T>::value;
    if (C % V == 0 && ((uintptr_t)in % (V * sizeof(T))) == 0 && ((uintptr_t)out % (V * sizeof(T))) == 0)
      hipLaunchKernelGGL((align_fwd_nhwc<T, V, ROT>), dim3(blocks), dim3(256), 0, st, in, rois,
                         out, C, H, W, M, scale, PH, PW, sr, aligned, roi_level, level);
    else
      hipLaunchKernelGGL((align_fwd_nhwc<T, 1, ROT>), dim3(blocks), dim3(256), 0, st, in, rois,
                         out, C, H, W, M, scale, PH, PW, sr, aligned, roi_level, level);
  } else {
    JTSM_REQUIRE(!roi_level, "roi_align: per-level filtering needs the NHWC layout");
    const long total = (long)M * C * PH * PW;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL((align_fwd_nchw<T, ROT>), dim3(blocks), dim3(256), 0, st, in, rois, out, C,
                       H, W, total, scale, PH, PW, sr, aligned);
  }
  JTSM_CHECK_LAUNCH("roi_align forward");
  return JTSM_OK;
}

template <typename T, bool ROT>
int launch_backward(const T* grad, const T* rois, T* gin, int B, int C, int H, int W, int M,
                    T scale, int PH, int PW, int sr, int aligned, int layout, void* stream,
                    const int* roi_level = nullptr, int level = 0) {
  JTSM_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && M >= 0 && PH > 0 && PW > 0,
               "roi_align backward: negative size");
  JTSM_REQUIRE(layout == JTSM_NCHW || layout == JTSM_NHWC, "roi_align: unknown layout %d", layout);
  hipStream_t st = as_stream(stream);
  const size_t in_elems = (size_t)B * C * H * W;
  if (in_elems == 0) return JTSM_OK;
  JTSM_REQUIRE(gin, "roi_align backward: null grad_input");
  JTSM_CHECK_HIP(hipMemsetAsync(gin, 0, in_elems * sizeof(T), st));
  if ((long)M * C * PH * PW == 0) return JTSM_OK;  // empty gradient: zeros (ROIAlign_cuda.cu:402-405)
  JTSM_REQUIRE(grad && rois, "roi_align backward: null pointer");
  if (layout == JTSM_NHWC) {
    const long waves = (long)M * PH * PW;
    const int blocks = ceil_div(waves, 4);
    constexpr int V = WideVec<T>::value;
    if (C % V == 0 && ((uintptr_t)grad % (V * sizeof(T))) == 0)
      hipLaunchKernelGGL((align_bwd_nhwc<T, V, ROT>), dim3(blocks), dim3(256), 0, st, grad, rois,
                         gin, C, H, W, M, scale, PH, PW, sr, aligned, roi_level, level);
    else
      hipLaunchKernelGGL((align_bwd_nhwc<T, 1, ROT>), dim3(blocks), dim3(256), 0, st, grad, rois,
                         gin, C, H, W, M, scale, PH, PW, sr, aligned, roi_level, level);
  } else {
    JTSM_REQUIRE(!roi_level, "roi_align: per-level filtering needs the NHWC layout");
    const long total = (long)M * C * PH * PW;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL((align_bwd_nchw<T, ROT>), dim3(blocks), dim3(256), 0, st, grad, rois, gin,
                       C, H, W, total, scale, PH, PW, sr, aligned);
  }
  JTSM_CHECK_LAUNCH("roi_align backward");
  return JTSM_OK;
}

// ---- mask targets of the pseudo-GT rectangles (get_pgt_mask with the rectangle substitution of SURVEY F8;
// structures/masks.py:169-200 crop_and_resize = ROIAlign(1.0, sampling 0, aligned) of the instance bitmask).
// The bitmask of a shrunk rectangle is an analytic image (pixel centre inside the box), so nothing is rasterised:
// thread (roi, ph, pw) runs the very ROIAlign sampling arithmetic above on that image and thresholds at 0.5.
__global__ __launch_bounds__(256) void rect_mask_targets_kernel(const float* __restrict__ rois,   // (N,4) boxes
                                                                const float* __restrict__ rects,  // (N,4) matched pseudo GT
                                                                unsigned char* __restrict__ out, long total, int side,
                                                                int H, int W, float erode) {
#pragma clang fp contract(off)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % side), ph = (int)((idx / side) % side);
    const long n = idx / side / side;
    const float roi5[5] = {0.f, rois[n * 4], rois[n * 4 + 1], rois[n * 4 + 2], rois[n * 4 + 3]};
    const RoiGeom<float> g = geom_box<float>(roi5, 1.0f, side, side, 0, true);
    const float* r = rects + n * 4;
    const float rx0 = r[0] + erode, ry0 = r[1] + erode, rx1 = r[2] - erode, ry1 = r[3] - erode;
    const int cells = g.gh * g.gw;
    const float count = (float)(cells > 1 ? cells : 1);
    float acc = 0.f;
    for (int iy = 0; iy < g.gh; ++iy)
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap<float> t = sample_tap<float, false>(g, H, W, ph, pw, iy, ix);
        if (t.pos[0] < 0) continue;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int y = t.pos[k] / W, x = t.pos[k] - y * W;
          const float xs = (float)x + 0.5f, ys = (float)y + 0.5f;
          v[k] = (xs >= rx0 && xs <= rx1 && ys >= ry0 && ys <= ry1) ? 1.f : 0.f;
        }
        acc += t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3];
      }
    out[idx] = (acc / count) >= 0.5f ? 1 : 0;
  }
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

int jtsm_roi_align_forward_f32(const float* input, const float* rois, float* output, int B, int C,
                               int H, int W, int M, float spatial_scale, int pooled_h,
                               int pooled_w, int sampling_ratio, int aligned, int layout,
                               void* stream) {
  return launch_forward<float, false>(input, rois, output, B, C, H, W, M, spatial_scale, pooled_h,
                                      pooled_w, sampling_ratio, aligned, layout, stream);
}
int jtsm_roi_align_backward_f32(const float* grad, const float* rois, float* grad_input, int B,
                                int C, int H, int W, int M, float spatial_scale, int pooled_h,
                                int pooled_w, int sampling_ratio, int aligned, int layout,
                                void* stream) {
  return launch_backward<float, false>(grad, rois, grad_input, B, C, H, W, M, spatial_scale,
                                       pooled_h, pooled_w, sampling_ratio, aligned, layout, stream);
}
int jtsm_roi_align_forward_f64(const double* input, const double* rois, double* output, int B,
                               int C, int H, int W, int M, double spatial_scale, int pooled_h,
                               int pooled_w, int sampling_ratio, int aligned, int layout,
                               void* stream) {
  return launch_forward<double, false>(input, rois, output, B, C, H, W, M, spatial_scale, pooled_h,
                                       pooled_w, sampling_ratio, aligned, layout, stream);
}
int jtsm_roi_align_backward_f64(const double* grad, const double* rois, double* grad_input, int B,
                                int C, int H, int W, int M, double spatial_scale, int pooled_h,
                                int pooled_w, int sampling_ratio, int aligned, int layout,
                                void* stream) {
  return launch_backward<double, false>(grad, rois, grad_input, B, C, H, W, M, spatial_scale,
                                        pooled_h, pooled_w, sampling_ratio, aligned, layout, stream);
}

int jtsm_roi_align_rotated_forward_f32(const float* input, const float* rois, float* output,
                                       int B, int C, int H, int W, int M, float spatial_scale,
                                       int pooled_h, int pooled_w, int sampling_ratio, int layout,
                                       void* stream) {
  return launch_forward<float, true>(input, rois, output, B, C, H, W, M, spatial_scale, pooled_h,
                                     pooled_w, sampling_ratio, 1, layout, stream);
}
int jtsm_roi_align_rotated_backward_f32(const float* grad, const float* rois, float* grad_input,
                                        int B, int C, int H, int W, int M, float spatial_scale,
                                        int pooled_h, int pooled_w, int sampling_ratio, int layout,
                                        void* stream) {
  return launch_backward<float, true>(grad, rois, grad_input, B, C, H, W, M, spatial_scale,
                                      pooled_h, pooled_w, sampling_ratio, 1, layout, stream);
}
int jtsm_roi_align_rotated_forward_f64(const double* input, const double* rois, double* output,
                                       int B, int C, int H, int W, int M, double spatial_scale,
                                       int pooled_h, int pooled_w, int sampling_ratio, int layout,
                                       void* stream) {
  return launch_forward<double, true>(input, rois, output, B, C, H, W, M, spatial_scale, pooled_h,
                                      pooled_w, sampling_ratio, 1, layout, stream);
}
int jtsm_roi_align_rotated_backward_f64(const double* grad, const double* rois, double* grad_input,
                                        int B, int C, int H, int W, int M, double spatial_scale,
                                        int pooled_h, int pooled_w, int sampling_ratio, int layout,
                                        void* stream) {
  return launch_backward<double, true>(grad, rois, grad_input, B, C, H, W, M, spatial_scale,
                                       pooled_h, pooled_w, sampling_ratio, 1, layout, stream);
}

int jtsm_roi_align_forward_level_f32(const float* input, const float* rois, const int32_t* roi_level,
                                     int level, float* output, int B, int C, int H, int W, int M,
                                     float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                     int aligned, void* stream) {
  JTSM_REQUIRE(roi_level || M == 0, "roi_align level: null roi_level");
  return launch_forward<float, false>(input, rois, output, B, C, H, W, M, spatial_scale, pooled_h,
                                      pooled_w, sampling_ratio, aligned, JTSM_NHWC, stream, roi_level, level);
}
int jtsm_roi_align_backward_level_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                      int level, float* grad_input, int B, int C, int H, int W, int M,
                                      float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                      int aligned, void* stream) {
  JTSM_REQUIRE(roi_level || M == 0, "roi_align level: null roi_level");
  return launch_backward<float, false>(grad, rois, grad_input, B, C, H, W, M, spatial_scale, pooled_h,
                                       pooled_w, sampling_ratio, aligned, JTSM_NHWC, stream, roi_level, level);
}

int jtsm_roi_sample_table_f32(const float* rois, int rotated, int M, int H, int W,
                              float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                              int aligned, int* grid, int* pos, float* w, int cap, void* stream) {
  JTSM_REQUIRE(M >= 0 && cap > 0 && pooled_h > 0 && pooled_w > 0, "sample_table: bad sizes");
  if (M == 0) return JTSM_OK;
  JTSM_REQUIRE(rois && grid && pos && w, "sample_table: null pointer");
  const long total = (long)M * cap;
  const int blocks = ceil_div(total, 256);
  if (rotated)
    hipLaunchKernelGGL((sample_table_kernel<float, true>), dim3(blocks), dim3(256), 0,
                       as_stream(stream), rois, M, H, W, spatial_scale, pooled_h, pooled_w,
                       sampling_ratio, 1, grid, pos, w, cap);
  else
    hipLaunchKernelGGL((sample_table_kernel<float, false>), dim3(blocks), dim3(256), 0,
                       as_stream(stream), rois, M, H, W, spatial_scale, pooled_h, pooled_w,
                       sampling_ratio, aligned, grid, pos, w, cap);
  JTSM_CHECK_LAUNCH("roi_sample_table");
  return JTSM_OK;
}

int jtsm_rect_mask_targets_f32(const float* rois, const float* rects, uint8_t* out, int N, int side, int H, int W,
                               float erode, void* stream) {
  JTSM_REQUIRE(N >= 0 && side > 0 && H > 0 && W > 0, "rect_mask_targets: bad sizes");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(rois && rects && out, "rect_mask_targets: null pointer");
  const long total = (long)N * side * side;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(rect_mask_targets_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), rois, rects, out, total,
                     side, H, W, erode);
  JTSM_CHECK_LAUNCH("rect_mask_targets");
  return JTSM_OK;
}

}  // extern "C"
