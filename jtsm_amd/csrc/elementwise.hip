// Bandwidth-bound helpers around the contraction kernels (all NHWC, fp32, 16 B per lane).
#include <cfloat>

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
// the same, also writing the bf16 hi / lo planes of g (eight elements per thread)
typedef __bf16 ew_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void relu_bwd_split_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                             float4* __restrict__ g, ew_bf16x8* __restrict__ hi,
                                                             ew_bf16x8* __restrict__ lo, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a0 = dy[2 * i], a1 = dy[2 * i + 1], b0 = y[2 * i], b1 = y[2 * i + 1];
    const float v[8] = {b0.x > 0.f ? a0.x : 0.f, b0.y > 0.f ? a0.y : 0.f, b0.z > 0.f ? a0.z : 0.f, b0.w > 0.f ? a0.w : 0.f,
                        b1.x > 0.f ? a1.x : 0.f, b1.y > 0.f ? a1.y : 0.f, b1.z > 0.f ? a1.z : 0.f, b1.w > 0.f ? a1.w : 0.f};
    g[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
    g[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    ew_bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 hh = (__bf16)v[e];
      h[e] = hh;
      l[e] = (__bf16)(v[e] - (float)hh);
    }
    hi[i] = h;
    lo[i] = l;
  }
}
// fp16 form: ONE plane of g * 2^shift
typedef _Float16 ew_f16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void relu_bwd_split_f16_kernel(const float4* __restrict__ dy, const float4* __restrict__ y,
                                                                 float4* __restrict__ g, ew_f16x8* __restrict__ h, long n8,
                                                                 int shift) {
  const float sc = __int_as_float((127 + shift) << 23);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a0 = dy[2 * i], a1 = dy[2 * i + 1], b0 = y[2 * i], b1 = y[2 * i + 1];
    const float v[8] = {b0.x > 0.f ? a0.x : 0.f, b0.y > 0.f ? a0.y : 0.f, b0.z > 0.f ? a0.z : 0.f, b0.w > 0.f ? a0.w : 0.f,
                        b1.x > 0.f ? a1.x : 0.f, b1.y > 0.f ? a1.y : 0.f, b1.z > 0.f ? a1.z : 0.f, b1.w > 0.f ? a1.w : 0.f};
    g[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
    g[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    ew_f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(v[e] * sc);
    h[i] = o;
  }
}
// ---- one streaming pass that ends in operand planes (layers/conv.py), three uses:
//   kGateScaled  v = y > 0 ? dy * scale : 0          the backward of ReLU followed by dropout (y = the dropout's output:
//                                                    positive exactly where the unit was kept AND active), g = v stored
//   kRowScale    v = x[r][c] * row_scale[r]          planes of a row-rescaled matrix (the per-roi factor in front of the
//                                                    box head) without materialising it; nothing else stored
//   kDropout     v = keep(i) ? x * scale : 0         inverted dropout, y = v stored (may alias x); keep(i) is a
//                                                    counter-based hash of (seed, element index): nothing to remember,
//                                                    the backward reads the mask off y (kGateScaled)
// Planes: bf16 hi + lo, or (lo == null) one fp16 plane of v * 2^shift; hi == null: no planes.
enum PassMode { kGateScaled = 0, kRowScale = 1, kDropout = 2 };
struct PassArgs {
  const float* a;           // dy | x | x
  const float* b;           // y  | - | -
  float* out;               // g  | - | y
  unsigned short* hi;
  unsigned short* lo;
  const float* row_scale;
  long n8;                  // elements / 8
  int cols;                 // kRowScale: row length (a multiple of 8)
  float scale;
  float drop_p;
  unsigned long long seed;
  int shift;
};
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);     // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);            // 24 bits -> [0, 1)
}
template <int MODE>
__global__ __launch_bounds__(256) void planes_pass_kernel(const PassArgs q) {
  const float4* __restrict__ a = reinterpret_cast<const float4*>(q.a);
  const float4* __restrict__ b = reinterpret_cast<const float4*>(q.b);
  float4* __restrict__ out = reinterpret_cast<float4*>(q.out);
  const float sh = __int_as_float((127 + q.shift) << 23);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < q.n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a0 = a[2 * i], a1 = a[2 * i + 1];
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    if (MODE == kGateScaled) {
      const float4 b0 = b[2 * i], b1 = b[2 * i + 1];
      const float y[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = y[e] > 0.f ? v[e] * q.scale : 0.f;
    } else if (MODE == kRowScale) {
      const float rs = q.row_scale[(i * 8) / q.cols];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= rs;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = uniform01(q.seed, (unsigned long long)(i * 8 + e)) >= q.drop_p ? v[e] * q.scale : 0.f;
    }
    if (MODE != kRowScale) {
      out[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
      out[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (!q.hi) continue;
    if (q.lo) {
      ew_bf16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const __bf16 hh = (__bf16)v[e];
        h[e] = hh;
        l[e] = (__bf16)(v[e] - (float)hh);
      }
      reinterpret_cast<ew_bf16x8*>(q.hi)[i] = h;
      reinterpret_cast<ew_bf16x8*>(q.lo)[i] = l;
    } else {
      ew_f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (_Float16)(v[e] * sh);
      reinterpret_cast<ew_f16x8*>(q.hi)[i] = o;
    }
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


// Wide form (C % 4 == 0, 16-byte aligned rows): a lane owns four channels, a wavefront 256 consecutive channels of a
// row, the four wavefronts stride the rows of the block's slab; partial sums go to part[slab][C] and a second kernel
// adds the slabs in order — no atomics (2048 workgroups hammering the same 256 addresses were the bottleneck:
// 0.2 ms for 257 MB), and reproducible bit for bit.
__global__ __launch_bounds__(256) void channel_sum4_kernel(const float4* __restrict__ g, float4* __restrict__ part,
                                                           long rows, int C4, long rows_per_block) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + lane;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c4 < C4) {
    long r = r0 + wv;
    for (; r + 12 < r1; r += 16) {   // four rows in flight per wavefront
      const float4 a = g[r * C4 + c4], b = g[(r + 4) * C4 + c4], c = g[(r + 8) * C4 + c4], d = g[(r + 12) * C4 + c4];
      acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
      acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
    }
    for (; r < r1; r += 4) {
      const float4 a = g[r * C4 + c4];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
  }
  red[wv][lane] = acc;
  __syncthreads();
  if (wv == 0 && c4 < C4) {
    const float4 a = red[0][lane], b = red[1][lane], c = red[2][lane], d = red[3][lane];
    part[(size_t)blockIdx.y * C4 + c4] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y),
                                                     (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
  }
}
// The same from operand PLANES (layers/conv.py): value = hi + lo (split-bf16) or the fp16 plane times 2^-shift.  A lane
// owns eight channels (16 bytes per plane), slabs and fold as above.  For gradients a chain keeps as planes only.
typedef __bf16 ew_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ew_h8 __attribute__((ext_vector_type(8)));
// Up to kCsumMulti gradients of the same width in ONE launch (blockIdx.z = which): the bias gradients of a chain's
// layers (the mask tower) are summed together when the chain's backward is over, one launch + one fold instead of one
// pair per layer.
constexpr int kCsumMulti = 8;
struct CsumPlanes {
  const unsigned short* hi[kCsumMulti];
  const unsigned short* lo[kCsumMulti];
  float* out[kCsumMulti];
  long rows[kCsumMulti];
  long rows_per_block[kCsumMulti];
  int nslab[kCsumMulti];
  int n;
};
__global__ __launch_bounds__(256) void channel_sum_planes_kernel(const CsumPlanes q, float* __restrict__ part_all,
                                                                 int max_slab, int C8, float unshift, int cols8) {
  const int which = blockIdx.z;
  if ((int)blockIdx.y >= q.nslab[which]) return;
  const unsigned short* __restrict__ hi = q.hi[which];
  const unsigned short* __restrict__ lo = q.lo[which];
  float* __restrict__ part = part_all + (size_t)which * max_slab * C8 * 8;
  const long rows = q.rows[which], rows_per_block = q.rows_per_block[which];
  // cols8 <= 256 eight-channel groups per workgroup, 256 / cols8 rows per trip: every lane busy whatever the width
  // (C = 256 is 32 groups: 8 rows per trip), 32 bytes per lane per row in flight x 2 trips unrolled
  __shared__ float red[256][9];
  const int tid = threadIdx.x, rpi = 256 / cols8;
  const int cl = tid % cols8, rg = tid / cols8;
  const int c8 = blockIdx.x * cols8 + cl;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  auto add_row = [&](long r) {
    const size_t o = ((size_t)r * C8 + c8) * 8;
    if (lo) {
      const ew_bf16x8 a = *reinterpret_cast<const ew_bf16x8*>(hi + o), b = *reinterpret_cast<const ew_bf16x8*>(lo + o);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)a[j] + (float)b[j];
    } else {
      const ew_h8 a = *reinterpret_cast<const ew_h8*>(hi + o);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += (float)a[j];
    }
  };
  if (rg < rpi && c8 < C8) {
    long r = r0 + rg;
    for (; r + rpi < r1; r += 2 * rpi) { add_row(r); add_row(r + rpi); }
    for (; r < r1; r += rpi) add_row(r);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[tid][j] = acc[j];
  __syncthreads();
  if (rg == 0 && c8 < C8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
      for (int k = 0; k < rpi; ++k) t += red[k * cols8 + cl][j];
      part[(size_t)blockIdx.y * C8 * 8 + c8 * 8 + j] = t * unshift;
    }
  }
}
// Any C (not a multiple of 4, narrower than 128, few rows): `cols` <= 256 columns per workgroup, 256 / cols row groups
// striding the slab's rows, partial sums to part[slab][C] — the same fixed-order scheme, so bias gradients are
// reproducible whatever the layer's width (the atomic form above is kept for callers without a workspace).
__global__ __launch_bounds__(256) void channel_sum_slab_kernel(const float* __restrict__ g, float* __restrict__ part,
                                                               long rows, int C, long rows_per_block, int cols) {
  __shared__ float red[256];
  const int rpi = 256 / cols, tid = threadIdx.x;
  const int cl = tid % cols, rg = tid / cols;
  const int c = blockIdx.x * cols + cl;
  const long r0 = (long)blockIdx.y * rows_per_block;
  const long r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float acc = 0.f;
  if (rg < rpi && c < C)
    for (long r = r0 + rg; r < r1; r += rpi) acc += g[r * C + c];
  red[tid] = acc;
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f;
    for (int k = 0; k < rpi; ++k) t += red[k * cols + cl];
    part[(size_t)blockIdx.y * C + c] = t;
  }
}
// out[c] = sum_s part[s][c], slabs in a fixed order: a workgroup folds 16 channels, 16 threads per channel each
// taking every 16th slab, then the 16 partial sums are added in order.
__global__ __launch_bounds__(256) void channel_sum_fold_multi_kernel(const CsumPlanes q, const float* __restrict__ part_all,
                                                                     int max_slab, int C) {
  __shared__ float red[16][17];
  const int which = blockIdx.y, slabs = q.nslab[which];
  const float* __restrict__ part = part_all + (size_t)which * max_slab * C;
  float* __restrict__ out = q.out[which];
  const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f;
  if (c < C)
    for (int s = j; s < slabs; s += 16) a += part[(size_t)s * C + c];
  red[j][cl] = a;
  __syncthreads();
  if (j == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    out[c] = t;
  }
}
__global__ __launch_bounds__(256) void channel_sum_fold_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                               int C, int slabs) {
  __shared__ float red[16][17];
  const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float a = 0.f;
  if (c < C)
    for (int s = j; s < slabs; s += 16) a += part[(size_t)s * C + c];
  red[j][cl] = a;
  __syncthreads();
  if (j == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    out[c] = t;
  }
}

// ---- NHWC spatial helpers (C % 4 == 0: one float4 = 4 channels of one pixel) ---------------------
__device__ __forceinline__ float4 f4max(float4 a, float4 b) {
  return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) {
  return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// max_pool2d(kernel 3, stride 2, padding 1)  — BasicStem (resnet.py:355-359)
__global__ __launch_bounds__(256) void maxpool3s2_fwd(const float4* __restrict__ x,
                                                      float4* __restrict__ y, int N, int H, int W,
                                                      int C4, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        m = f4max(m, x[((long)(n * H + ih) * W + iw) * C4 + c]);
      }
    }
    y[i] = m;
  }
}
// backward: recompute the first maximum of each window and add the gradient there
__global__ __launch_bounds__(256) void maxpool3s2_bwd(const float* __restrict__ x,
                                                      const float* __restrict__ gy,
                                                      float* __restrict__ gx, int N, int H, int W,
                                                      int C, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m = -FLT_MAX;
    long at = -1;
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh * 2 - 1 + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow * 2 - 1 + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const long o = ((long)(n * H + ih) * W + iw) * C + c;
        const float v = x[o];
        if (v > m || at < 0) { m = v; at = o; }
      }
    }
    if (at >= 0) atomicAdd(gx + at, gy[i]);
  }
}

// out = lateral + nearest_upsample_x2(top)      (FPN top-down path, fpn.py:133-136)
typedef __bf16 ew_bf16x4 __attribute__((ext_vector_type(4)));
// bf16 hi / lo planes of four values (what jtsm_split_bf16_f32 makes of them), for a bf16x3 consumer
__device__ __forceinline__ void ew_planes4(unsigned short* hi, unsigned short* lo, long i4, const float4& v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  ew_bf16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
  reinterpret_cast<ew_bf16x4*>(hi)[i4] = h;
  reinterpret_cast<ew_bf16x4*>(lo)[i4] = l;
}

__global__ __launch_bounds__(256) void upsample2_add_fwd(const float4* __restrict__ top,
                                                         const float4* __restrict__ lat,
                                                         float4* __restrict__ out, unsigned short* __restrict__ out_hi,
                                                         unsigned short* __restrict__ out_lo, int N, int H,
                                                         int W, int C4) {
  const long total = (long)N * H * W * C4;
  const int Ht = H / 2, Wt = W / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float4 v = f4add(lat[i], top[((long)(n * Ht + h / 2) * Wt + w / 2) * C4 + c]);
    out[i] = v;
    if (out_hi) ew_planes4(out_hi, out_lo, i, v);
  }
}
// out = x0 + x1 (+ x2 (+ x3)), summed in that order (the accumulation order of `x = x + y` loops), optional bf16
// planes of the sum: the level sum of SemSegFPNHead.layers (semantic_seg.py:178-186) in one pass.
struct SumPtrs { const float4* p[4]; };
__global__ __launch_bounds__(256) void sum_tensors_kernel(SumPtrs in, int n, long total4, float4* __restrict__ out,
                                                          unsigned short* __restrict__ hi, unsigned short* __restrict__ lo) {
#pragma clang fp contract(off)
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    float4 v = in.p[0][i];
    for (int k = 1; k < n; ++k) v = f4add(v, in.p[k][i]);
    out[i] = v;
    if (hi) ew_planes4(hi, lo, i, v);
  }
}
// d_top[n,h,w] = sum of the 2x2 block of g
__global__ __launch_bounds__(256) void sum2x2_kernel(const float4* __restrict__ g,
                                                     float4* __restrict__ out, int N, int Ht, int Wt,
                                                     int C4) {
  const long total = (long)N * Ht * Wt * C4;
  const int W = Wt * 2, H = Ht * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % Wt); t /= Wt;
    const int h = (int)(t % Ht);
    const int n = (int)(t / Ht);
    const long b = ((long)(n * H + 2 * h) * W + 2 * w) * C4 + c;
    out[i] = f4add(f4add(g[b], g[b + C4]), f4add(g[b + (long)W * C4], g[b + (long)W * C4 + C4]));
  }
}
// y[n,h,w] = x[n,2h,2w]  (max_pool2d kernel 1 stride 2 = LastLevelMaxPool, fpn.py:173-185);
// scatter=1 runs it backwards: x[n,2h,2w] = y[n,h,w] into a zero-filled x.
__global__ __launch_bounds__(256) void subsample2_kernel(const float4* __restrict__ src,
                                                         float4* __restrict__ dst, int N, int H, int W,
                                                         int C4, int Ho, int Wo, int scatter) {
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % Wo); t /= Wo;
    const int h = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const long big = ((long)(n * H + 2 * h) * W + 2 * w) * C4 + c;
    if (scatter) dst[big] = src[i]; else dst[i] = src[big];
  }
}

// ---- SGD with momentum over many tensors in one launch (torch.optim.SGD semantics: dampening 0, no nesterov;
// detectron2/solver/build.py:110-195 builds exactly that).  Table rows are eight 64-bit words:
// {param, grad, momentum buffer, n, first_block, lr (float bits), weight_decay (float bits), momentum (float bits)}.
struct SgdEntry { float* p; const float* g; float* buf; long n; long first_block; long lr; long wd; long mu; };

__device__ __forceinline__ float word_f32(long w) { return __int_as_float((int)w); }

__global__ __launch_bounds__(256) void sgd_multi_kernel(const SgdEntry* __restrict__ tab, int nent, int first_step) {
  int lo = 0, hi = nent - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= (long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const SgdEntry t = tab[lo];
  const float lr = word_f32(t.lr), wd = word_f32(t.wd), mu = word_f32(t.mu);
  const long base = ((long)blockIdx.x - t.first_block) * 1024 + threadIdx.x * 4;
  if (base >= t.n) return;
  if (base + 4 <= t.n && ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.buf) & 15) == 0)) {
    float4 p = *reinterpret_cast<const float4*>(t.p + base);
    const float4 g = *reinterpret_cast<const float4*>(t.g + base);
    float4 b = first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(t.buf + base);
    const float d[4] = {g.x + wd * p.x, g.y + wd * p.y, g.z + wd * p.z, g.w + wd * p.w};
    b.x = first_step ? d[0] : mu * b.x + d[0];
    b.y = first_step ? d[1] : mu * b.y + d[1];
    b.z = first_step ? d[2] : mu * b.z + d[2];
    b.w = first_step ? d[3] : mu * b.w + d[3];
    p.x -= lr * b.x; p.y -= lr * b.y; p.z -= lr * b.z; p.w -= lr * b.w;
    *reinterpret_cast<float4*>(t.buf + base) = b;
    *reinterpret_cast<float4*>(t.p + base) = p;
  } else {
    for (long i = base; i < t.n && i < base + 4; ++i) {
      const float d = t.g[i] + wd * t.p[i];
      const float b = first_step ? d : mu * t.buf[i] + d;
      t.buf[i] = b;
      t.p[i] -= lr * b;
    }
  }
}

// ---- 2x2 max pooling of the WSL ResNet-v2 backbone (projects/WSL/wsl/modeling/backbone/resnet_wsl_v2.py:157-165,
// 413): MaxPool2d(2, stride 2), or ZeroPad2d((0,1,0,1)) + MaxPool2d(2, stride 1) in the dilated stages.  NHWC,
// float4 of channels per thread.  Window cells outside the input read the pad value 0 (stride-1 form only).
__device__ __forceinline__ float4 mp_load(const float4* __restrict__ x, int n, int h, int w, int H, int W, int C4, int c) {
  if (h >= H || w >= W) return make_float4(0.f, 0.f, 0.f, 0.f);
  return x[((size_t)(n * H + h) * W + w) * C4 + c];
}
__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ y, int N,
                                                             int H, int W, int C4, int stride, int Ho, int Wo, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const int h = oh * stride, w = ow * stride;
    const float4 a = mp_load(x, n, h, w, H, W, C4, c), b = mp_load(x, n, h, w + 1, H, W, C4, c);
    const float4 d = mp_load(x, n, h + 1, w, H, W, C4, c), e = mp_load(x, n, h + 1, w + 1, H, W, C4, c);
    float4 o;
    o.x = fmaxf(fmaxf(a.x, b.x), fmaxf(d.x, e.x));
    o.y = fmaxf(fmaxf(a.y, b.y), fmaxf(d.y, e.y));
    o.z = fmaxf(fmaxf(a.z, b.z), fmaxf(d.z, e.z));
    o.w = fmaxf(fmaxf(a.w, b.w), fmaxf(d.w, e.w));
    y[i] = o;
  }
}
// Backward as a gather: input pixel (h,w) collects from the <= 4 windows that contain it and whose FIRST maximum
// (window order (0,0),(0,1),(1,0),(1,1), as ATen's max_pool2d_with_indices) it is.
__device__ __forceinline__ int mp_first_max(float a, float b, float d, float e) {
  int k = 0; float m = a;
  if (b > m) { m = b; k = 1; }
  if (d > m) { m = d; k = 2; }
  if (e > m) { m = e; k = 3; }
  return k;
}
__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                             float* __restrict__ gx, int N, int H, int W, int C,
                                                             int stride, int Ho, int Wo, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float acc = 0.f;
    for (int dh = 0; dh < 2; ++dh)
      for (int dw = 0; dw < 2; ++dw) {
        const int th = h - dh, tw = w - dw;   // window top-left in input coordinates
        if (th < 0 || tw < 0 || th % stride || tw % stride) continue;
        const int oh = th / stride, ow = tw / stride;
        if (oh >= Ho || ow >= Wo) continue;
        float v[4];
        for (int k = 0; k < 4; ++k) {
          const int hh = th + (k >> 1), ww = tw + (k & 1);
          v[k] = (hh < H && ww < W) ? x[((size_t)(n * H + hh) * W + ww) * C + c] : 0.f;
        }
        if (mp_first_max(v[0], v[1], v[2], v[3]) == dh * 2 + dw) acc += gy[((size_t)(n * Ho + oh) * Wo + ow) * C + c];
      }
    gx[i] = acc;
  }
}


// Column sums of up to 16 small matrices in ONE launch (the per-row-tile partial sums the data-gradient epilogues leave
// for a bias gradient, conv_igemm.hip: Params::colsum): block b adds the rows of matrix b — out[b][c] = sum_r
// part[b][r][c].  Pointers and row counts by value: no table upload.
constexpr int kFoldMax = 16;
struct FoldParts {
  const float* part[kFoldMax];
  int rows[kFoldMax];
};
__global__ __launch_bounds__(1024) void colsum_fold_kernel(const FoldParts f, int width, float* __restrict__ out) {
  // thread = (16-byte column piece, row group): G = 1024 / (width / 4) row groups each add every G-th row in order with
  // eight loads in flight, then the groups are added in group order through LDS — a fixed order, reproducible
  __shared__ float4 red[1024];
  // (blockIdx.y: a block of at most 128 columns, so that a 256-wide matrix gets 2 x 32 row groups)
  const int b = blockIdx.x, t = threadIdx.x, wpr = width >> 2;                 // float4 pieces per row
  const int c0 = blockIdx.y * 32, cpr = min(32, wpr - c0), G = 1024 / cpr;     // this block's pieces, its row groups
  const int cg = t % cpr, rg = t / cpr;
  const int rows = f.rows[b];
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rg < G) {
    const float4* __restrict__ p = reinterpret_cast<const float4*>(f.part[b]) + c0 + cg;
    int r = rg;
    for (; r + 7 * G < rows; r += 8 * G) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(r + u * G) * wpr];
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; r < rows; r += G) {
      const float4 v = p[(size_t)r * wpr];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  red[t] = acc;
  __syncthreads();
  if (t < cpr) {
    float4 s = red[t];
    for (int g = 1; g < G; ++g) {
      const float4 q = red[t + g * cpr];
      s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
    }
    reinterpret_cast<float4*>(out + (size_t)b * width)[c0 + t] = s;
  }
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

int jtsm_relu_backward_split_f32(const float* dy, const float* y, float* g, uint16_t* g_hi, uint16_t* g_lo, long n,
                                 void* stream) {
  JTSM_REQUIRE(n >= 0 && n % 8 == 0, "relu_backward_split: n must be a non-negative multiple of 8");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(dy && y && g && g_hi && g_lo, "relu_backward_split: null pointer");
  JTSM_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)g | (uintptr_t)g_hi | (uintptr_t)g_lo) & 15) == 0,
               "relu_backward_split: pointers must be 16-byte aligned");
  const long n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  hipLaunchKernelGGL(relu_bwd_split_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)dy,
                     (const float4*)y, (float4*)g, reinterpret_cast<ew_bf16x8*>(g_hi), reinterpret_cast<ew_bf16x8*>(g_lo),
                     n8);
  JTSM_CHECK_LAUNCH("relu_backward_split");
  return JTSM_OK;
}

int jtsm_relu_backward_split_f16(const float* dy, const float* y, float* g, uint16_t* g_h, long n, int shift,
                                 void* stream) {
  JTSM_REQUIRE(n >= 0 && n % 8 == 0, "relu_backward_split_f16: n must be a non-negative multiple of 8");
  JTSM_REQUIRE(shift >= 0 && shift <= 24, "relu_backward_split_f16: shift must be in 0..24");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(dy && y && g && g_h, "relu_backward_split_f16: null pointer");
  JTSM_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)g | (uintptr_t)g_h) & 15) == 0,
               "relu_backward_split_f16: pointers must be 16-byte aligned");
  const long n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 8192 ? (n8 + 255) / 256 : 8192);
  hipLaunchKernelGGL(relu_bwd_split_f16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)dy,
                     (const float4*)y, (float4*)g, reinterpret_cast<ew_f16x8*>(g_h), n8, shift);
  JTSM_CHECK_LAUNCH("relu_backward_split_f16");
  return JTSM_OK;
}

static int launch_planes_pass(int mode, const PassArgs& q, const char* what, void* stream) {
  const int blocks = (int)((q.n8 + 255) / 256 < 8192 ? (q.n8 + 255) / 256 : 8192);
  hipStream_t st = as_stream(stream);
  if (mode == kGateScaled) hipLaunchKernelGGL(planes_pass_kernel<kGateScaled>, dim3(blocks), dim3(256), 0, st, q);
  else if (mode == kRowScale) hipLaunchKernelGGL(planes_pass_kernel<kRowScale>, dim3(blocks), dim3(256), 0, st, q);
  else hipLaunchKernelGGL(planes_pass_kernel<kDropout>, dim3(blocks), dim3(256), 0, st, q);
  JTSM_CHECK_LAUNCH(what);
  return JTSM_OK;
}
static bool pass_aligned(const void* a, const void* b, const void* c, const void* d, const void* e) {
  return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d | (uintptr_t)e) & 15) == 0;
}

int jtsm_relu_backward_split_scaled_f32(const float* dy, const float* y, float scale, float* g, uint16_t* g_hi,
                                        uint16_t* g_lo, long n, int shift, void* stream) {
  JTSM_REQUIRE(n >= 0 && n % 8 == 0, "relu_backward_split_scaled: n must be a non-negative multiple of 8");
  JTSM_REQUIRE(shift >= 0 && shift <= 24, "relu_backward_split_scaled: shift must be in 0..24");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(dy && y && g && (g_hi || !g_lo), "relu_backward_split_scaled: null pointer");
  JTSM_REQUIRE(pass_aligned(dy, y, g, g_hi, g_lo), "relu_backward_split_scaled: pointers must be 16-byte aligned");
  PassArgs q = {};
  q.a = dy; q.b = y; q.out = g; q.hi = g_hi; q.lo = g_lo; q.n8 = n / 8; q.scale = scale; q.shift = shift;
  return launch_planes_pass(kGateScaled, q, "relu_backward_split_scaled", stream);
}

int jtsm_split_rowscale_f32(const float* src, const float* row_scale, long rows, int cols, uint16_t* hi, uint16_t* lo,
                            int shift, void* stream) {
  JTSM_REQUIRE(rows >= 0 && cols > 0 && cols % 8 == 0, "split_rowscale: cols must be a positive multiple of 8");
  JTSM_REQUIRE(shift >= 0 && shift <= 24, "split_rowscale: shift must be in 0..24");
  if (rows == 0) return JTSM_OK;
  JTSM_REQUIRE(src && row_scale && hi, "split_rowscale: null pointer");
  JTSM_REQUIRE(pass_aligned(src, hi, lo, nullptr, nullptr), "split_rowscale: pointers must be 16-byte aligned");
  PassArgs q = {};
  q.a = src; q.row_scale = row_scale; q.hi = hi; q.lo = lo; q.n8 = rows * (long)(cols / 8); q.cols = cols; q.shift = shift;
  return launch_planes_pass(kRowScale, q, "split_rowscale", stream);
}

int jtsm_dropout_split_f32(const float* x, float* y, uint16_t* y_hi, uint16_t* y_lo, long n, float p,
                           unsigned long long seed, void* stream) {
  JTSM_REQUIRE(n >= 0 && n % 8 == 0, "dropout_split: n must be a non-negative multiple of 8");
  JTSM_REQUIRE(p >= 0.f && p < 1.f, "dropout_split: p must be in [0, 1)");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(x && y && (y_hi || !y_lo), "dropout_split: null pointer");
  JTSM_REQUIRE(pass_aligned(x, y, y_hi, y_lo, nullptr), "dropout_split: pointers must be 16-byte aligned");
  PassArgs q = {};
  q.a = x; q.out = y; q.hi = y_hi; q.lo = y_lo; q.n8 = n / 8; q.scale = 1.f / (1.f - p); q.drop_p = p; q.seed = seed;
  return launch_planes_pass(kDropout, q, "dropout_split", stream);
}

static bool csum_wide(long rows, int C) { return C % 4 == 0 && C >= 128 && rows >= 256; }
static void csum_plan(long rows, int C, long* rpb, int* nslab) {
  const int cg = csum_wide(rows, C) ? ceil_div(C / 4, 64) : ceil_div(C, C < 256 ? C : 256);
  long slabs = 1024 / cg;            // ~4 workgroups per CU in all
  if (slabs < 1) slabs = 1;
  if (slabs > (rows + 63) / 64) slabs = (rows + 63) / 64;
  if (slabs < 1) slabs = 1;
  *rpb = (rows + slabs - 1) / slabs;
  *nslab = (int)((rows + *rpb - 1) / *rpb);
}

size_t jtsm_channel_sum_workspace_bytes(long rows, int C) {
  if (rows <= 0 || C <= 0) return 0;
  long rpb; int nslab;
  csum_plan(rows, C, &rpb, &nslab);
  return (size_t)nslab * C * sizeof(float) + 16;
}

int jtsm_channel_sum_ws_f32(const float* g, float* out, long rows, int C, void* workspace, size_t workspace_bytes,
                            void* stream) {
  JTSM_REQUIRE(rows >= 0 && C > 0, "channel_sum: bad sizes");
  JTSM_REQUIRE(out, "channel_sum: null out");
  hipStream_t st = as_stream(stream);
  const bool have_ws = workspace && ((uintptr_t)workspace & 15) == 0 && rows > 0 &&
                       workspace_bytes >= jtsm_channel_sum_workspace_bytes(rows, C);
  if (!have_ws) return jtsm_channel_sum_f32(g, out, rows, C, stream);
  JTSM_REQUIRE(g, "channel_sum: null pointer");
  long rpb; int nslab;
  csum_plan(rows, C, &rpb, &nslab);
  float* part = reinterpret_cast<float*>(workspace);
  if (csum_wide(rows, C) && ((uintptr_t)g & 15) == 0) {
    hipLaunchKernelGGL(channel_sum4_kernel, dim3(ceil_div(C / 4, 64), (unsigned)nslab), dim3(256), 0, st,
                       reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(part), rows, C / 4, rpb);
  } else {
    const int cols = C < 256 ? C : 256;
    hipLaunchKernelGGL(channel_sum_slab_kernel, dim3(ceil_div(C, cols), (unsigned)nslab), dim3(256), 0, st, g, part,
                       rows, C, rpb, cols);
  }
  hipLaunchKernelGGL(channel_sum_fold_kernel, dim3(ceil_div(C, 16)), dim3(256), 0, st, part, out, C, nslab);
  JTSM_CHECK_LAUNCH("channel_sum");
  return JTSM_OK;
}

static void csum_planes_plan(long rows, int C, long* rpb, int* nslab) {
  const int cols8 = C / 8 < 256 ? C / 8 : 256;
  const int cg = ceil_div(C / 8, cols8);
  long slabs = 1024 / cg;
  if (slabs > (rows + 63) / 64) slabs = (rows + 63) / 64;
  if (slabs < 1) slabs = 1;
  *rpb = (rows + slabs - 1) / slabs;
  *nslab = (int)((rows + *rpb - 1) / *rpb);
}

int jtsm_channel_sum_planes_multi(const uint16_t* const* hi, const uint16_t* const* lo, float* const* outs,
                                  const long* rows, int count, int C, int shift, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(count >= 1 && count <= kCsumMulti && hi && outs && rows, "channel_sum_planes: 1..8 gradients per call");
  JTSM_REQUIRE(C > 0 && C % 8 == 0, "channel_sum_planes: C must be a positive multiple of 8");
  JTSM_REQUIRE(shift >= 0 && shift <= 24, "channel_sum_planes: shift must be in 0..24");
  hipStream_t st = as_stream(stream);
  CsumPlanes q = {};
  int max_slab = 0;
  for (int i = 0; i < count; ++i) {
    JTSM_REQUIRE(rows[i] >= 0 && outs[i], "channel_sum_planes: bad entry %d", i);
    if (rows[i] == 0) {
      JTSM_CHECK_HIP(hipMemsetAsync(outs[i], 0, (size_t)C * sizeof(float), st));
      continue;
    }
    JTSM_REQUIRE(hi[i] && ((uintptr_t)hi[i] & 15) == 0 && (!lo || ((uintptr_t)lo[i] & 15) == 0),
                 "channel_sum_planes: planes must be non-null and 16-byte aligned");
    const int k = q.n++;
    q.hi[k] = hi[i]; q.lo[k] = lo ? lo[i] : nullptr; q.out[k] = outs[i]; q.rows[k] = rows[i];
    csum_planes_plan(rows[i], C, &q.rows_per_block[k], &q.nslab[k]);
    if (q.nslab[k] > max_slab) max_slab = q.nslab[k];
  }
  if (q.n == 0) return JTSM_OK;
  const size_t need = (size_t)q.n * max_slab * C * sizeof(float);
  JTSM_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 && workspace_bytes >= need,
               "channel_sum_planes: workspace of %zu bytes needed (count x 1024 x C floats always suffice)", need);
  float* part = reinterpret_cast<float*>(workspace);
  const int cols8 = C / 8 < 256 ? C / 8 : 256;
  const int cg = ceil_div(C / 8, cols8);
  const float unshift = __builtin_ldexpf(1.f, -shift);
  hipLaunchKernelGGL(channel_sum_planes_kernel, dim3(cg, (unsigned)max_slab, (unsigned)q.n), dim3(256), 0, st, q, part,
                     max_slab, C / 8, unshift, cols8);
  hipLaunchKernelGGL(channel_sum_fold_multi_kernel, dim3(ceil_div(C, 16), (unsigned)q.n), dim3(256), 0, st, q, part,
                     max_slab, C);
  JTSM_CHECK_LAUNCH("channel_sum_planes");
  return JTSM_OK;
}

int jtsm_channel_sum_planes(const uint16_t* hi, const uint16_t* lo, float* out, long rows, int C, int shift,
                            void* workspace, size_t workspace_bytes, void* stream) {
  return jtsm_channel_sum_planes_multi(&hi, lo ? &lo : nullptr, &out, &rows, 1, C, shift, workspace, workspace_bytes,
                                       stream);
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

static inline int grid_for(long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

int jtsm_maxpool3x3s2_forward_f32(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "maxpool: bad sizes (C %% 4 != 0?)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(x && y, "maxpool: null pointer");
  hipLaunchKernelGGL(maxpool3s2_fwd, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const float4*)x, (float4*)y, N, H, W, C / 4, Ho, Wo);
  JTSM_CHECK_LAUNCH("maxpool forward");
  return JTSM_OK;
}

int jtsm_maxpool3x3s2_backward_f32(const float* x, const float* gy, float* gx, int N, int H, int W, int C,
                                   void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0, "maxpool backward: bad sizes");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipStream_t st = as_stream(stream);
  if ((long)N * H * W * C == 0) return JTSM_OK;
  JTSM_REQUIRE(x && gy && gx, "maxpool backward: null pointer");
  JTSM_CHECK_HIP(hipMemsetAsync(gx, 0, (size_t)N * H * W * C * sizeof(float), st));
  const long total = (long)N * Ho * Wo * C;
  hipLaunchKernelGGL(maxpool3s2_bwd, dim3(grid_for(total)), dim3(256), 0, st, x, gy, gx, N, H, W, C, Ho, Wo);
  JTSM_CHECK_LAUNCH("maxpool backward");
  return JTSM_OK;
}

int jtsm_upsample2_add_f32(const float* top, const float* lateral, float* out, uint16_t* out_hi, uint16_t* out_lo,
                           int N, int H, int W, int C, void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && H % 2 == 0 && W % 2 == 0,
               "upsample2_add: need even H, W and C %% 4 == 0");
  const long total = (long)N * H * W * (C / 4);
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(top && lateral && out, "upsample2_add: null pointer");
  JTSM_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), "upsample2_add: give both planes or neither");
  hipLaunchKernelGGL(upsample2_add_fwd, dim3(grid_for(total)), dim3(256), 0, as_stream(stream),
                     (const float4*)top, (const float4*)lateral, (float4*)out, out_hi, out_lo, N, H, W, C / 4);
  JTSM_CHECK_LAUNCH("upsample2_add");
  return JTSM_OK;
}

int jtsm_sum_tensors_f32(const float* const* inputs, int n, long numel, float* out, uint16_t* out_hi, uint16_t* out_lo,
                         void* stream) {
  JTSM_REQUIRE(n >= 1 && n <= 4 && numel >= 0 && numel % 4 == 0, "sum_tensors: 1..4 inputs, numel %% 4 == 0");
  if (numel == 0) return JTSM_OK;
  JTSM_REQUIRE(inputs && out, "sum_tensors: null pointer");
  JTSM_REQUIRE((out_hi == nullptr) == (out_lo == nullptr), "sum_tensors: give both planes or neither");
  SumPtrs sp = {};
  for (int k = 0; k < n; ++k) {
    JTSM_REQUIRE(inputs[k] && ((uintptr_t)inputs[k] & 15) == 0, "sum_tensors: input %d null or not 16-byte aligned", k);
    sp.p[k] = reinterpret_cast<const float4*>(inputs[k]);
  }
  hipLaunchKernelGGL(sum_tensors_kernel, dim3(grid_for(numel / 4)), dim3(256), 0, as_stream(stream), sp, n, numel / 4,
                     (float4*)out, out_hi, out_lo);
  JTSM_CHECK_LAUNCH("sum_tensors");
  return JTSM_OK;
}

int jtsm_sum2x2_f32(const float* g, float* out, int N, int Ht, int Wt, int C, void* stream) {
  JTSM_REQUIRE(N >= 0 && Ht > 0 && Wt > 0 && C > 0 && C % 4 == 0, "sum2x2: bad sizes");
  const long total = (long)N * Ht * Wt * (C / 4);
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(g && out, "sum2x2: null pointer");
  hipLaunchKernelGGL(sum2x2_kernel, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), (const float4*)g,
                     (float4*)out, N, Ht, Wt, C / 4);
  JTSM_CHECK_LAUNCH("sum2x2");
  return JTSM_OK;
}

int jtsm_subsample2_f32(const float* src, float* dst, int N, int H, int W, int C, int scatter, void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "subsample2: bad sizes");
  const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  hipStream_t st = as_stream(stream);
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(src && dst, "subsample2: null pointer");
  if (scatter) JTSM_CHECK_HIP(hipMemsetAsync(dst, 0, (size_t)N * H * W * C * sizeof(float), st));
  hipLaunchKernelGGL(subsample2_kernel, dim3(grid_for(total)), dim3(256), 0, st, (const float4*)src,
                     (float4*)dst, N, H, W, C / 4, Ho, Wo, scatter);
  JTSM_CHECK_LAUNCH("subsample2");
  return JTSM_OK;
}

int jtsm_sgd_momentum_multi_f32(const void* table, int entries, long blocks, int first_step, void* stream) {
  JTSM_REQUIRE(entries >= 0 && blocks >= 0 && blocks < 2147483647L, "sgd_multi: bad sizes");
  if (entries == 0 || blocks == 0) return JTSM_OK;
  JTSM_REQUIRE(table && ((uintptr_t)table & 15) == 0, "sgd_multi: table must be a 16-byte aligned device pointer");
  hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const SgdEntry*>(table), entries, first_step);
  JTSM_CHECK_LAUNCH("sgd_multi");
  return JTSM_OK;
}

static int mp2_out(int in, int stride) { return stride == 1 ? in : in / 2; }

int jtsm_maxpool2x2_forward_f32(const float* x, float* y, int N, int H, int W, int C, int stride, void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && (stride == 1 || stride == 2), "maxpool2x2: bad sizes");
  const int Ho = mp2_out(H, stride), Wo = mp2_out(W, stride);
  const long total = (long)N * Ho * Wo * (C / 4);
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(x && y, "maxpool2x2: null pointer");
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), (const float4*)x, (float4*)y, N,
                     H, W, C / 4, stride, Ho, Wo, total);
  JTSM_CHECK_LAUNCH("maxpool2x2 forward");
  return JTSM_OK;
}

int jtsm_maxpool2x2_backward_f32(const float* x, const float* gy, float* gx, int N, int H, int W, int C, int stride,
                                 void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2), "maxpool2x2 backward: bad sizes");
  const long total = (long)N * H * W * C;
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(x && gy && gx, "maxpool2x2 backward: null pointer");
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, gy, gx, N, H, W, C, stride,
                     mp2_out(H, stride), mp2_out(W, stride), total);
  JTSM_CHECK_LAUNCH("maxpool2x2 backward");
  return JTSM_OK;
}

int jtsm_colsum_fold_f32(const float* const* parts, const int* rows, int n, int width, float* out, void* stream) {
  JTSM_REQUIRE(n >= 1 && n <= jtsm::kFoldMax && width > 0 && parts && rows && out, "colsum_fold: 1..%d matrices", jtsm::kFoldMax);
  jtsm::FoldParts f = {};
  for (int b = 0; b < n; ++b) {
    JTSM_REQUIRE(parts[b] && rows[b] >= 0, "colsum_fold: matrix %d", b);
    f.part[b] = parts[b];
    f.rows[b] = rows[b];
  }
  JTSM_REQUIRE(width % 4 == 0 && width <= 4096, "colsum_fold: width %% 4 == 0 and <= 4096, got %d", width);
  hipLaunchKernelGGL(jtsm::colsum_fold_kernel, dim3(n, jtsm::ceil_div(width, 128)), dim3(1024), 0, jtsm::as_stream(stream), f,
                     width, out);
  JTSM_CHECK_LAUNCH("colsum_fold");
  return JTSM_OK;
}

}  // extern "C"
