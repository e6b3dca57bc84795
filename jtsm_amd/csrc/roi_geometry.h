// Device-side ROI sampling geometry shared by ROIAlign / ROIAlignRotated kernels and the
// sample-table dump.  Behaviour contract: SURVEY.md Appendix A.1/A.2, i.e. the arithmetic of
//   detectron2/layers/csrc/ROIAlign/ROIAlign_cpu.cpp:20-114,134-169   (axis-aligned)
//   detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:27-129,218-261 (rotated)
// The integer results (grid size, y_low/x_low/..., validity) must equal the CPU path's bit for
// bit, so every floating-point step here is individually rounded: FMA contraction is switched
// off for this header (hipcc contracts by default; the x86-64 reference build has no FMA).
#pragma once
#include <hip/hip_runtime.h>

namespace jtsm {

#pragma clang fp contract(off)

template <typename T>
struct Tap {
  int pos[4];  // flat y*W+x of the four neighbours; pos[0] == -1 marks "out of range"
  T w[4];
};

template <typename T>
struct RoiGeom {
  int b;
  T y0, x0, bh, bw;
  int gh, gw;
  T cy, cx, cs, sn;
};

template <typename T> __device__ __forceinline__ T ceil_t(T v);
template <> __device__ __forceinline__ float ceil_t<float>(float v) { return ceilf(v); }
template <> __device__ __forceinline__ double ceil_t<double>(double v) { return ceil(v); }

// Axis-aligned box (x0,y0,x1,y1).  Negative extents in aligned mode give an empty grid
// (the CUDA reference does the same; only its CPU path asserts, ROIAlign_cpu.cpp:149-152).
template <typename T>
__device__ __forceinline__ RoiGeom<T> geom_box(const T* __restrict__ roi, T scale, int PH,
                                               int PW, int sampling_ratio, bool aligned) {
#pragma clang fp contract(off)
  RoiGeom<T> g;
  g.b = (int)roi[0];
  const T off = aligned ? (T)0.5 : (T)0.0;
  const T sx = roi[1] * scale - off, sy = roi[2] * scale - off;
  const T ex = roi[3] * scale - off, ey = roi[4] * scale - off;
  T rw = ex - sx, rh = ey - sy;
  if (!aligned) {
    rw = rw < (T)1 ? (T)1 : rw;
    rh = rh < (T)1 ? (T)1 : rh;
  }
  g.y0 = sy;
  g.x0 = sx;
  g.bh = rh / (T)PH;
  g.bw = rw / (T)PW;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceil_t<T>(rh / (T)PH);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceil_t<T>(rw / (T)PW);
  g.cy = g.cx = (T)0;
  g.cs = (T)1;
  g.sn = (T)0;
  return g;
}

// Rotated box (cx,cy,w,h,deg).  The reference evaluates the angle in double and calls the
// double cos/sin, rounding the results to T (ROIAlignRotated_cpu.cpp:232-234).
template <typename T>
__device__ __forceinline__ RoiGeom<T> geom_rbox(const T* __restrict__ roi, T scale, int PH,
                                                int PW, int sampling_ratio) {
#pragma clang fp contract(off)
  RoiGeom<T> g;
  g.b = (int)roi[0];
  g.cx = roi[1] * scale - (T)0.5;
  g.cy = roi[2] * scale - (T)0.5;
  const T rw = roi[3] * scale, rh = roi[4] * scale;
  const T theta = (T)((double)roi[5] * 3.14159265358979323846 / 180.0);
  g.cs = (T)cos((double)theta);
  g.sn = (T)sin((double)theta);
  g.bh = rh / (T)PH;
  g.bw = rw / (T)PW;
  g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceil_t<T>(rh / (T)PH);
  g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceil_t<T>(rw / (T)PW);
  g.y0 = (T)(-(double)rh / 2.0);
  g.x0 = (T)(-(double)rw / 2.0);
  return g;
}

// One bilinear sample.  STRICT selects the rotated variant's "y < 0" clamp test.
template <typename T, bool STRICT>
__device__ __forceinline__ Tap<T> make_tap(int H, int W, T y, T x) {
#pragma clang fp contract(off)
  Tap<T> t;
  if (y < (T)-1.0 || y > (T)H || x < (T)-1.0 || x > (T)W) {
    t.pos[0] = t.pos[1] = t.pos[2] = t.pos[3] = -1;
    t.w[0] = t.w[1] = t.w[2] = t.w[3] = (T)0;
    return t;
  }
  if (STRICT) {
    if (y < (T)0) y = (T)0;
    if (x < (T)0) x = (T)0;
  } else {
    if (y <= (T)0) y = (T)0;
    if (x <= (T)0) x = (T)0;
  }
  int yl = (int)y, xl = (int)x, yh, xh;
  if (yl >= H - 1) { yh = yl = H - 1; y = (T)yl; } else { yh = yl + 1; }
  if (xl >= W - 1) { xh = xl = W - 1; x = (T)xl; } else { xh = xl + 1; }
  const T ly = y - (T)yl, lx = x - (T)xl;
  const T hy = (T)1 - ly, hx = (T)1 - lx;
  t.pos[0] = yl * W + xl;
  t.pos[1] = yl * W + xh;
  t.pos[2] = yh * W + xl;
  t.pos[3] = yh * W + xh;
  t.w[0] = hy * hx;
  t.w[1] = hy * lx;
  t.w[2] = ly * hx;
  t.w[3] = ly * lx;
  return t;
}

template <typename T, bool ROT>
__device__ __forceinline__ Tap<T> sample_tap(const RoiGeom<T>& g, int H, int W, int ph, int pw,
                                             int iy, int ix) {
#pragma clang fp contract(off)
  // (T)(iy + .5f): the reference forms the half-offset in float first, also for double.
  const T yy = g.y0 + (T)ph * g.bh + (T)((float)iy + .5f) * g.bh / (T)g.gh;
  const T xx = g.x0 + (T)pw * g.bw + (T)((float)ix + .5f) * g.bw / (T)g.gw;
  if (!ROT) return make_tap<T, false>(H, W, yy, xx);
  const T y = yy * g.cs - xx * g.sn + g.cy;
  const T x = yy * g.sn + xx * g.cs + g.cx;
  return make_tap<T, true>(H, W, y, x);
}

template <typename T, bool ROT>
__device__ __forceinline__ RoiGeom<T> roi_geometry(const T* __restrict__ rois, int n, T scale,
                                                   int PH, int PW, int sampling_ratio,
                                                   bool aligned) {
  return ROT ? geom_rbox<T>(rois + (size_t)n * 6, scale, PH, PW, sampling_ratio)
             : geom_box<T>(rois + (size_t)n * 5, scale, PH, PW, sampling_ratio, aligned);
}

}  // namespace jtsm
