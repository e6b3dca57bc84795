// Inference / post-processing kernels of libjtsm_hip.so: K-head score averaging + box decoding,
// per-class greedy NMS, detection selection, mask probabilities and pasting, bilinear resize + arg-max of the
// semantic logits, and the panoptic merge.  HBM-bound integer / byte work: one pass over the data per stage,
// coalesced rows, LDS only for the 64-box column tiles of the NMS bit matrix; no host synchronisation anywhere
// (data-dependent counts stay in device memory until the caller asks for them).
#include "common.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cstdint>

namespace jtsm {
namespace {

typedef unsigned long long u64;

constexpr int kMaxHeads = 8;
struct HeadPtrs { const float* p[kMaxHeads]; };

// ---------------------------------------------------------------------------------------------------------------
// OICROutputLayers.predict_probs_K / predict_boxes_K (fast_rcnn_oicr.py:712-783): mean over heads of the row
// soft-max, mean over heads of the deltas, Box2BoxTransform.apply_deltas.  One wavefront per proposal.
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ __launch_bounds__(256) void oicr_predict_kernel(HeadPtrs logits, HeadPtrs deltas, int heads, int R, int C1,
                                                           int Kb, const float* __restrict__ proposals,
                                                           float wx, float wy, float ww, float wh, float clampv,
                                                           float* __restrict__ probs, float* __restrict__ boxes) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  for (int c0 = 0; c0 < C1; c0 += 64) {   // accumulate per 64-column chunk; the row max / sum need the whole row
    const int c = c0 + lane;
    float acc = 0.f;
    for (int h = 0; h < heads; ++h) {
      const float* z = logits.p[h] + (size_t)r * C1;
      float m = -INFINITY;
      for (int j = lane; j < C1; j += 64) m = fmaxf(m, z[j]);
      m = wave_max(m);
      float s = 0.f;
      for (int j = lane; j < C1; j += 64) s += expf(z[j] - m);
      s = wave_sum(s);
      if (c < C1) acc += expf(z[c] - m) / s;
    }
    if (c < C1) probs[(size_t)r * C1 + c] = acc / (float)heads;
  }
  if (!boxes) return;
  const float* p = proposals + 4 * (size_t)r;
  const float w = p[2] - p[0], hgt = p[3] - p[1];
  const float cx = p[0] + 0.5f * w, cy = p[1] + 0.5f * hgt;
  for (int k = lane; k < Kb; k += 64) {
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < heads; ++h)
      for (int j = 0; j < 4; ++j) d[j] += deltas.p[h][(size_t)r * Kb * 4 + 4 * k + j];
    for (int j = 0; j < 4; ++j) d[j] = d[j] / (float)heads;
    const float dx = d[0] / wx, dy = d[1] / wy;
    const float dw = fminf(d[2] / ww, clampv), dh = fminf(d[3] / wh, clampv);
    const float pcx = dx * w + cx, pcy = dy * hgt + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * hgt;
    float* o = boxes + (size_t)r * Kb * 4 + 4 * k;
    o[0] = pcx - 0.5f * pw;
    o[1] = pcy - 0.5f * ph;
    o[2] = pcx + 0.5f * pw;
    o[3] = pcy + 0.5f * ph;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Greedy per-class NMS.  Elements live in a flat array (box, score, class); class == num_classes marks a dropped
// element.  Stages: 64-bit keys (class, descending score) -> radix sort -> class segment starts -> 64x64 IoU bit
// matrix restricted to W column words per row (a class never has more than max_per_class members) -> one wavefront
// per class walks its segment -> second radix sort puts the survivors first, by descending score, ties by index.
struct NmsHeader {
  int n_valid;      // elements with class < num_classes
  int max_coord;    // order-preserving int image of the largest coordinate among valid boxes
  int num_keep;
  int overflow;     // a class had more members than max_per_class allows
};

__device__ __forceinline__ unsigned ordered_desc(float f) {   // larger float -> smaller key (-0 sorts after +0)
  unsigned u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~u;
}
__device__ __forceinline__ int float_to_ordered_int(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_int_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

__global__ void nms_header_init(NmsHeader* h) {
  h->n_valid = 0;
  h->max_coord = float_to_ordered_int(-INFINITY);
  h->num_keep = 0;
  h->overflow = 0;
}

// per-row validity of fast_rcnn_inference_single_image (fast_rcnn_oicr.py:129-133): every box coordinate and every
// score (background column included) finite.  One wavefront per row.
__global__ __launch_bounds__(256) void det_row_valid_kernel(const float* __restrict__ boxes, int ldb,
                                                            const float* __restrict__ scores, int lds_, int R,
                                                            uint8_t* __restrict__ valid) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  bool ok = true;
  for (int j = lane; j < ldb; j += 64) ok = ok && isfinite(boxes[(size_t)r * ldb + j]);
  for (int j = lane; j < lds_; j += 64) ok = ok && isfinite(scores[(size_t)r * lds_ + j]);
  ok = __all(ok);
  if (lane == 0) valid[r] = ok ? 1 : 0;
}

__device__ __forceinline__ void header_accumulate(NmsHeader* h, bool ok, float cmax) {
  // wave-level reduction, then one atomic per wavefront (integer atomics: order-independent results)
  const u64 b = __ballot(ok);
  int m = ok ? float_to_ordered_int(cmax) : float_to_ordered_int(-INFINITY);
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0 && b) {
    atomicAdd(&h->n_valid, __popcll(b));
    atomicMax(&h->max_coord, m);
  }
}

// elements of fast_rcnn_inference_single_image: e = r * K + k; clipped class-k box of row r, its score, class k or
// K (dropped: invalid row or score <= thresh).
__global__ __launch_bounds__(256) void det_elements_kernel(const float* __restrict__ boxes, int Kb,
                                                           const float* __restrict__ scores, int R, int K,
                                                           const uint8_t* __restrict__ valid, float img_h, float img_w,
                                                           float thresh, float4* __restrict__ ebox,
                                                           float* __restrict__ escore, int* __restrict__ eclass,
                                                           NmsHeader* h) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long N = (long)R * K;
  bool ok = false;
  float cmax = -INFINITY;
  if (e < N) {
    const int r = (int)(e / K), k = (int)(e % K);
    const float s = scores[(size_t)r * (K + 1) + k];
    ok = valid[r] && s > thresh;
    const float* b = boxes + (size_t)r * Kb * 4 + (Kb == 1 ? 0 : 4 * k);
    float4 v;
    v.x = fminf(fmaxf(b[0], 0.f), img_w);
    v.y = fminf(fmaxf(b[1], 0.f), img_h);
    v.z = fminf(fmaxf(b[2], 0.f), img_w);
    v.w = fminf(fmaxf(b[3], 0.f), img_h);
    if (!valid[r]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    ebox[e] = v;
    escore[e] = ok ? s : -INFINITY;
    eclass[e] = ok ? k : K;
    cmax = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
  }
  header_accumulate(h, ok, cmax);
}

// elements of a plain batched_nms call
__global__ __launch_bounds__(256) void nms_elements_kernel(const float4* __restrict__ boxes,
                                                           const float* __restrict__ scores,
                                                           const int64_t* __restrict__ idxs, int n, int K,
                                                           float4* __restrict__ ebox, float* __restrict__ escore,
                                                           int* __restrict__ eclass, NmsHeader* h) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  bool ok = false;
  float cmax = -INFINITY;
  if (e < n) {
    const int64_t c = idxs[e];
    ok = c >= 0 && c < K;
    const float4 v = boxes[e];
    ebox[e] = v;
    escore[e] = scores[e];
    eclass[e] = ok ? (int)c : K;
    cmax = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
  }
  header_accumulate(h, ok, cmax);
}

__global__ __launch_bounds__(256) void nms_keys_kernel(const float* __restrict__ escore, const int* __restrict__ eclass,
                                                       int N, u64* __restrict__ key, int* __restrict__ iota) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N) return;
  key[e] = ((u64)(unsigned)eclass[e] << 32) | ordered_desc(escore[e]);
  iota[e] = e;
}

// seg[c] = first sorted position whose class is >= c, c = 0..K (seg[K] = number of valid elements)
__global__ void nms_segments_kernel(const u64* __restrict__ keys, int N, int K, int* __restrict__ seg) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > K) return;
  int lo = 0, hi = N;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int)(keys[mid] >> 32) < c) lo = mid + 1; else hi = mid;
  }
  seg[c] = lo;
}

// torchvision's nms IoU test (ops/csrc/cuda/nms_cuda.cu devIoU @0.8.1; the CPU kernel computes the same expression)
__device__ __forceinline__ bool iou_above(const float4& a, const float4& b, float thr) {
#pragma clang fp contract(off)
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float inter = width * height;
  const float sa = (a.z - a.x) * (a.w - a.y);
  const float sb = (b.z - b.x) * (b.w - b.y);
  return (inter / (sa + sb - inter)) > thr;
}

// trick: 0 plain per-class, 1 torchvision's coordinate offsets, 2 offsets iff fewer than 40000 valid elements
// (layers/nms.py:19-21)
__device__ __forceinline__ float4 nms_box(const float4* __restrict__ ebox, int e, int cls, float step) {
#pragma clang fp contract(off)
  float4 v = ebox[e];
  if (step != 0.f) {
    const float off = (float)cls * step;   // idxs.to(boxes) * (max_coordinate + 1)
    v.x = v.x + off; v.y = v.y + off; v.z = v.z + off; v.w = v.w + off;
  }
  return v;
}

__global__ __launch_bounds__(64) void nms_bits_kernel(const float4* __restrict__ ebox, const u64* __restrict__ keys,
                                                      const int* __restrict__ sidx, int N, int K, int W, float thr,
                                                      int trick, const NmsHeader* __restrict__ h,
                                                      u64* __restrict__ mask) {
#pragma clang fp contract(off)
  __shared__ float4 cbox[64];
  __shared__ int ccls[64];
  const int rb = blockIdx.x, y = blockIdx.y, t = threadIdx.x;
  const int p = rb * 64 + t, cb = rb + y;
  u64 bits = 0;
  const int col0 = cb * 64;
  // uniform early-outs: no column, rows all dropped, or the last row's class precedes the first column's class
  const int row_first_cls = rb * 64 < N ? (int)(keys[rb * 64] >> 32) : K;
  bool live = col0 < N && row_first_cls < K;
  if (live) {
    const int last = min(rb * 64 + 63, N - 1);
    live = (int)(keys[col0] >> 32) <= (int)(keys[last] >> 32);
  }
  if (live) {
    const bool use_trick = trick == 1 || (trick == 2 && h->n_valid < 40000);
    const float step = use_trick ? ordered_int_to_float(h->max_coord) + 1.0f : 0.f;
    const int q = col0 + t;
    int qc = K;
    if (q < N) {
      qc = (int)(keys[q] >> 32);
      cbox[t] = nms_box(ebox, sidx[q], qc, step);
    }
    ccls[t] = qc;
    __syncthreads();
    if (p < N) {
      const int pc = (int)(keys[p] >> 32);
      if (pc < K) {
        const float4 a = nms_box(ebox, sidx[p], pc, step);
        const int j0 = y == 0 ? t + 1 : 0;   // strictly upper triangle
        for (int j = j0; j < 64; ++j)
          if (ccls[j] == pc && iou_above(a, cbox[j], thr)) bits |= 1ull << j;
      }
    }
  }
  mask[(size_t)p * W + y] = bits;
}

__device__ __forceinline__ u64 shfl64(u64 v, int src) {
  const int lo = __shfl((int)(unsigned)(v & 0xffffffffull), src);
  const int hi = __shfl((int)(unsigned)(v >> 32), src);
  return ((u64)(unsigned)hi << 32) | (unsigned)lo;
}

// One wavefront per class walks its segment in score order.  Per 64-row block: the intra-block decisions are made
// from register-resident words (no memory latency in the serial chain), then every lane ORs the kept rows' words
// of one later column block into the LDS-resident `removed` vector (independent loads, all in flight together).
__global__ __launch_bounds__(64) void nms_scan_kernel(const u64* __restrict__ mask, int W,
                                                      const int* __restrict__ seg, uint8_t* __restrict__ kept,
                                                      NmsHeader* h) {
  extern __shared__ u64 removed[];   // 2W words
  const int c = blockIdx.x, lane = threadIdx.x;
  const int s = seg[c], e = seg[c + 1];
  if (s >= e) return;
  const int b0 = s >> 6, b1 = (e - 1) >> 6;
  if (b1 - b0 >= W) {   // more members than the caller promised
    if (lane == 0) atomicExch(&h->overflow, 1);
    return;
  }
  for (int w = lane; w < 2 * W; w += 64) removed[w] = 0;
  __syncthreads();
  int total = 0;
  for (int b = b0; b <= b1; ++b) {
    const int p = b * 64 + lane;
    const bool valid = p >= s && p < e;
    const u64 own = valid ? mask[(size_t)p * W] : 0ull;
    u64 cur = removed[b - b0];
    const u64 vbits = __ballot(valid);
    u64 keptbits = 0;
    for (int j = 0; j < 64; ++j) {
      const u64 o = shfl64(own, j);
      if (((vbits >> j) & 1) && !((cur >> j) & 1)) {
        keptbits |= 1ull << j;
        cur |= o;
      }
    }
    if (valid) kept[p] = (uint8_t)((keptbits >> lane) & 1);
    total += __popcll(keptbits);
    for (int w = 1 + lane; w < W; w += 64) {
      // every row of the block is read (independent loads, all in flight together) and masked by its kept bit:
      // cheaper than a dependent load per kept row
      const u64* col = mask + (size_t)b * 64 * W + w;
      u64 acc = 0;
#pragma unroll 16
      for (int j = 0; j < 64; ++j) acc |= col[(size_t)j * W] & (0ull - ((keptbits >> j) & 1ull));
      removed[b - b0 + w] |= acc;
    }
    __syncthreads();
  }
  if (lane == 0) atomicAdd(&h->num_keep, total);
}

__global__ __launch_bounds__(256) void nms_final_keys_kernel(const int* __restrict__ sidx,
                                                             const uint8_t* __restrict__ kept,
                                                             const float* __restrict__ escore, int N,
                                                             u64* __restrict__ key2) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int e = sidx[p];
  key2[e] = ((u64)(kept[p] ? 0u : 1u) << 32) | ordered_desc(escore[e]);
}

__global__ __launch_bounds__(256) void nms_emit_keep_kernel(const int* __restrict__ order, int n,
                                                            const NmsHeader* __restrict__ h, int64_t* __restrict__ keep,
                                                            int* __restrict__ num_keep, int* __restrict__ overflow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) keep[i] = order[i];
  if (i == 0) {
    *num_keep = h->num_keep;
    if (overflow) *overflow = h->overflow;
  }
}

__global__ __launch_bounds__(256) void det_emit_kernel(const int* __restrict__ order, const float4* __restrict__ ebox,
                                                       const float* __restrict__ escore, int K, int topk, int cap,
                                                       const NmsHeader* __restrict__ h, float4* __restrict__ out_boxes,
                                                       float* __restrict__ out_scores, int64_t* __restrict__ out_classes,
                                                       int64_t* __restrict__ out_rows, int* __restrict__ out_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int n = h->num_keep;
  if (topk >= 0 && n > topk) n = topk;
  if (n > cap) n = cap;
  if (i == 0) *out_count = n;
  if (i >= cap) return;
  if (i < n) {
    const int e = order[i];
    out_boxes[i] = ebox[e];
    out_scores[i] = escore[e];
    out_classes[i] = e % K;
    out_rows[i] = e / K;
  } else {
    out_boxes[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    out_scores[i] = 0.f;
    out_classes[i] = -1;
    out_rows[i] = -1;
  }
}

inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

struct NmsLayout {
  int N, W, nblk, K;
  size_t header, ebox, escore, eclass, key_a, key_b, iota, sidx, order, seg, kept, rowvalid, mask, cub, total;
  size_t cub_bytes;
};

size_t cub_temp_bytes(int N) {
  size_t a = 0, b = 0;
  u64* k = nullptr;
  int* v = nullptr;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, k, k, v, v, N, 0, 64, (hipStream_t) nullptr);
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, b, k, k, v, v, N, 0, 33, (hipStream_t) nullptr);
  return a > b ? a : b;
}

NmsLayout nms_layout(int N, int K, int max_per_class, int R) {
  NmsLayout l = {};
  l.N = N; l.K = K;
  l.W = ceil_div(max_per_class > 0 ? max_per_class : 1, 64) + 1;
  l.nblk = ceil_div(N > 0 ? N : 1, 64);
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += align256(bytes); return at; };
  l.header = take(sizeof(NmsHeader));
  l.ebox = take((size_t)N * 16);
  l.escore = take((size_t)N * 4);
  l.eclass = take((size_t)N * 4);
  l.key_a = take((size_t)N * 8);
  l.key_b = take((size_t)N * 8);
  l.iota = take((size_t)N * 4);
  l.sidx = take((size_t)N * 4);
  l.order = take((size_t)N * 4);
  l.seg = take((size_t)(K + 2) * 4);
  l.kept = take((size_t)l.nblk * 64);
  l.rowvalid = take((size_t)(R > 0 ? R : 1));
  l.mask = take((size_t)l.nblk * 64 * l.W * 8);
  l.cub_bytes = N > 0 ? cub_temp_bytes(N) : 0;
  l.cub = take(l.cub_bytes);
  l.total = o;
  return l;
}

int bits_for(int k) {   // bits needed to hold values 0..k
  int b = 1;
  while ((1 << b) <= k) ++b;
  return b;
}

// elements already written (ebox / escore / eclass / header); runs sort -> bits -> scan -> sort
int nms_core(char* ws, const NmsLayout& l, float thr, int trick, hipStream_t st) {
  const int N = l.N, K = l.K;
  NmsHeader* h = reinterpret_cast<NmsHeader*>(ws + l.header);
  float4* ebox = reinterpret_cast<float4*>(ws + l.ebox);
  float* escore = reinterpret_cast<float*>(ws + l.escore);
  int* eclass = reinterpret_cast<int*>(ws + l.eclass);
  u64* key_a = reinterpret_cast<u64*>(ws + l.key_a);
  u64* key_b = reinterpret_cast<u64*>(ws + l.key_b);
  int* iota = reinterpret_cast<int*>(ws + l.iota);
  int* sidx = reinterpret_cast<int*>(ws + l.sidx);
  int* order = reinterpret_cast<int*>(ws + l.order);
  int* seg = reinterpret_cast<int*>(ws + l.seg);
  uint8_t* kept = reinterpret_cast<uint8_t*>(ws + l.kept);
  u64* mask = reinterpret_cast<u64*>(ws + l.mask);
  size_t cub_bytes = l.cub_bytes;
  hipLaunchKernelGGL(nms_keys_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, escore, eclass, N, key_a, iota);
  JTSM_CHECK_LAUNCH("nms keys");
  JTSM_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(ws + l.cub, cub_bytes, key_a, key_b, iota, sidx, N, 0,
                                                    32 + bits_for(K), st));
  hipLaunchKernelGGL(nms_segments_kernel, dim3(ceil_div(K + 1, 64)), dim3(64), 0, st, key_b, N, K, seg);
  JTSM_CHECK_LAUNCH("nms segments");
  hipLaunchKernelGGL(nms_bits_kernel, dim3(l.nblk, l.W), dim3(64), 0, st, ebox, key_b, sidx, N, K, l.W, thr, trick, h,
                     mask);
  JTSM_CHECK_LAUNCH("nms bits");
  JTSM_CHECK_HIP(hipMemsetAsync(kept, 0, (size_t)l.nblk * 64, st));
  hipLaunchKernelGGL(nms_scan_kernel, dim3(K), dim3(64), (size_t)2 * l.W * sizeof(u64), st, mask, l.W, seg, kept, h);
  JTSM_CHECK_LAUNCH("nms scan");
  hipLaunchKernelGGL(nms_final_keys_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, sidx, kept, escore, N, key_a);
  JTSM_CHECK_LAUNCH("nms final keys");
  cub_bytes = l.cub_bytes;
  JTSM_CHECK_HIP(hipcub::DeviceRadixSort::SortPairs(ws + l.cub, cub_bytes, key_a, key_b, iota, order, N, 0, 33, st));
  return JTSM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// mask_rcnn_inference (projects/WSL/wsl/modeling/roi_heads/mask_head.py:106-147): sigmoid of the predicted class's
// channel of (sum over heads / heads).
__global__ __launch_bounds__(256) void mask_probs_kernel(HeadPtrs logits, int heads, const int64_t* __restrict__ classes,
                                                         int N, int C, int MM, float* __restrict__ out) {
#pragma clang fp contract(off)
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)N * MM) return;
  const int n = (int)(i / MM), m = (int)(i % MM);
  const int c = C == 1 ? 0 : (int)classes[n];
  float z = 0.f;
  for (int h = 0; h < heads; ++h) z += logits.p[h][((size_t)n * C + c) * MM + m];
  z = z / (float)heads;
  out[i] = 1.f / (1.f + expf(-z));
}

// paste_masks_in_image (detectron2/layers/mask_ops.py:17-145, the GPU branch: whole image, grid_sample bilinear /
// zeros / align_corners=False, then `>= threshold`).  One thread per 4 consecutive output pixels.
__device__ __forceinline__ float paste_tap(const float* __restrict__ m, int M, int iy, int ix) {
  return (iy >= 0 && iy < M && ix >= 0 && ix < M) ? m[iy * M + ix] : 0.f;
}

__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks,
                                                          const float* __restrict__ boxes, int M, int H, int Wd,
                                                          float threshold, uint8_t* __restrict__ out) {
#pragma clang fp contract(off)
  const int n = blockIdx.z, y = blockIdx.y;
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (x4 >= Wd) return;
  const float* m = masks + (size_t)n * M * M;
  const float x0 = boxes[4 * n], y0 = boxes[4 * n + 1], x1 = boxes[4 * n + 2], y1 = boxes[4 * n + 3];
  const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.f - 1.f;
  const float iy = ((gy + 1.f) * (float)M - 1.f) / 2.f;
  const float fy = floorf(iy);
  const int iy0 = (int)fy, iy1 = iy0 + 1;
  const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy;
  uint8_t v[4];
  for (int k = 0; k < 4; ++k) {
    const int x = x4 + k;
    const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.f - 1.f;
    const float ix = ((gx + 1.f) * (float)M - 1.f) / 2.f;
    const float fx = floorf(ix);
    float acc = 0.f;
    // NaN coordinates (degenerate box) never pass the bounds test: grid_sample yields 0 there as well
    if (fx >= -1.f && fx < (float)M && fy >= -1.f && fy < (float)M) {
      const int ix0 = (int)fx, ix1 = ix0 + 1;
      const float wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
      acc += paste_tap(m, M, iy0, ix0) * (wx0 * wy0);
      acc += paste_tap(m, M, iy0, ix1) * (wx1 * wy0);
      acc += paste_tap(m, M, iy1, ix0) * (wx0 * wy1);
      acc += paste_tap(m, M, iy1, ix1) * (wx1 * wy1);
    }
    v[k] = threshold >= 0.f ? (uint8_t)(acc >= threshold ? 1 : 0) : (uint8_t)(acc * 255.f);
  }
  uint8_t* o = out + ((size_t)n * H + y) * Wd + x4;
  if (x4 + 3 < Wd && ((size_t)o & 3) == 0) *reinterpret_cast<uchar4*>(o) = make_uchar4(v[0], v[1], v[2], v[3]);
  else for (int k = 0; k < 4 && x4 + k < Wd; ++k) o[k] = v[k];
}

// ---------------------------------------------------------------------------------------------------------------
// F.interpolate(mode="bilinear", align_corners=False) of the top-left (crop_h, crop_w) window of x, result planar
// (N, C, out_h, out_w).  src = max(scale * (dst + 0.5) - 0.5, 0) as upsample_bilinear2d does.
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, int nhwc, int C, int H, int Wd,
                                                              int crop_h, int crop_w, int OH, int OW, float sh, float sw,
                                                              float* __restrict__ y) {
#pragma clang fp contract(off)
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y, n = blockIdx.z;
  if (ox >= OW) return;
  const float ry = fmaxf(sh * ((float)oy + 0.5f) - 0.5f, 0.f), rx = fmaxf(sw * ((float)ox + 0.5f) - 0.5f, 0.f);
  const int y0 = min((int)ry, crop_h - 1), x0 = min((int)rx, crop_w - 1);
  const int yp = y0 < crop_h - 1 ? 1 : 0, xp = x0 < crop_w - 1 ? 1 : 0;
  const float ly1 = ry - (float)y0, ly0 = 1.f - ly1, lx1 = rx - (float)x0, lx0 = 1.f - lx1;
  const size_t sc = nhwc ? 1 : (size_t)H * Wd, sx = nhwc ? C : 1, sy = nhwc ? (size_t)Wd * C : Wd;
  const float* b = x + (size_t)n * C * H * Wd;
  const float* p00 = b + y0 * sy + x0 * sx;
  const float* p01 = p00 + xp * sx;
  const float* p10 = p00 + yp * sy;
  const float* p11 = p10 + xp * sx;
  float* o = y + ((size_t)n * C * OH + oy) * OW + ox;
  for (int c = 0; c < C; ++c)
    o[(size_t)c * OH * OW] = ly0 * (lx0 * p00[c * sc] + lx1 * p01[c * sc]) + ly1 * (lx0 * p10[c * sc] + lx1 * p11[c * sc]);
}

// F.interpolate(mode="nearest"): src = min(floor(dst * scale), in - 1), scale = in / out in float
// (upsample_nearest2d).  Planar (C, H, W) -> (C, OH, OW), optional horizontal flip of the SOURCE columns.
__global__ __launch_bounds__(256) void resize_nearest_kernel(const float* __restrict__ x, int C, int H, int Wd, int OH,
                                                             int OW, float sh, float sw, int flip,
                                                             float* __restrict__ y) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= OW) return;
  const int sy = min((int)floorf((float)oy * sh), H - 1);
  int sx = min((int)floorf((float)ox * sw), Wd - 1);
  if (flip) sx = Wd - 1 - sx;
  for (int c = 0; c < C; ++c) y[((size_t)c * OH + oy) * OW + ox] = x[((size_t)c * H + sy) * Wd + sx];
}

// arg-max over the channel axis of a planar (C, HW) array; the first maximum wins (torch.argmax); NaN is a maximum
__global__ __launch_bounds__(256) void argmax_channels_kernel(const float* __restrict__ x, int C, long HW,
                                                              int64_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= HW) return;
  float best = x[i];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = x[(size_t)c * HW + i];
    if (!(best != best) && (v > best || v != v)) { best = v; arg = c; }
  }
  out[i] = arg;
}

// ---------------------------------------------------------------------------------------------------------------
// combine_semantic_and_instance_outputs (detectron2/modeling/meta_arch/panoptic_fpn.py:133-218).  Instances are
// visited in descending-score order; each visit is one launch (pan_step_kernel: decide + paint the previous visit, count
// this one) whose decisions read only counters completed by earlier launches of the stream.
__device__ __forceinline__ int nonzero_bytes(unsigned v) {
  return __popc(((v & 0x7f7f7f7fu) + 0x7f7f7f7fu | v) & 0x80808080u);
}

// Pre-pass over all masks: area and the range of rows that hold any set pixel (one wavefront per row).
__global__ __launch_bounds__(256) void pan_extent_kernel(const uint8_t* __restrict__ masks, int N, int H, int W,
                                                         int* __restrict__ extent /* (N, 3): area, row_min, row_max */) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long)N * H) return;
  const int inst = (int)(row / H), y = (int)(row % H);
  const uint8_t* m = masks + (size_t)row * W;
  int cnt = 0;
  if ((W & 15) == 0 && ((size_t)masks & 15) == 0) {   // 16 pixels per lane and load
    for (int x = lane * 16; x < W; x += 64 * 16) {
      const uint4 v = *reinterpret_cast<const uint4*>(m + x);
      cnt += nonzero_bytes(v.x) + nonzero_bytes(v.y) + nonzero_bytes(v.z) + nonzero_bytes(v.w);
    }
  } else {
    for (int x = lane; x < W; x += 64) cnt += m[x] != 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0 && cnt) {
    atomicAdd(&extent[3 * inst], cnt);
    atomicMin(&extent[3 * inst + 1], y);
    atomicMax(&extent[3 * inst + 2], y);
  }
}

__global__ void pan_extent_init_kernel(int* __restrict__ extent, int N, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { extent[3 * i] = 0; extent[3 * i + 1] = H; extent[3 * i + 2] = -1; }
}

constexpr int kPanBlocks = 128;   // 512 rows per sweep; a visit only walks its mask's rows

// One launch per visit: decide + paint the PREVIOUS visit's instance (its overlap count is complete: the launch before
// this one took it), then count this visit's overlap with the painted map.  A pixel belongs to the same thread in both
// halves — row y to wavefront y mod (4 * gridDim.x), column x to lane (x / 4) mod 64 — so the count reads what its own
// thread painted a moment ago and no other thread's writes of this launch: the chain is one launch per instance
// instead of two (it is bound by launch latency, ~4 us per launch: 200 -> 101 launches per image at 100 detections).
__global__ __launch_bounds__(256) void pan_step_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ order,
                                                       const float* __restrict__ scores,
                                                       const int64_t* __restrict__ classes, int visit, int visits, int H,
                                                       int W, float conf, double overlap, const int* __restrict__ extent,
                                                       int* __restrict__ counters, int* __restrict__ next_id,
                                                       int* panoptic, int* __restrict__ seg_table /* (rows, 5) */,
                                                       float* __restrict__ seg_score) {
  const int lane = threadIdx.x & 63;
  const int my_row = blockIdx.x * 4 + (threadIdx.x >> 6), row_step = gridDim.x * 4;
  const bool vec = (W & 3) == 0 && ((size_t)masks & 3) == 0 && ((size_t)panoptic & 15) == 0;
  if (visit > 0) {                                   // ---- decide + paint visit - 1
    const int pv = visit - 1, inst = order[pv];
    const int area = extent[3 * inst], inter = counters[pv];
    const bool accept = !(scores[inst] < conf) && area > 0 && !((double)inter * 1.0 / (double)area > overlap);
    const int id = next_id[pv] + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      next_id[pv + 1] = next_id[pv] + (accept ? 1 : 0);
      if (accept) {
        int* row = seg_table + 5 * (id - 1);
        row[0] = id; row[1] = 1; row[2] = (int)classes[inst]; row[3] = inst; row[4] = area - inter;
        seg_score[id - 1] = scores[inst];
      }
    }
    if (accept) {
      const int y0 = extent[3 * inst + 1], y1 = extent[3 * inst + 2];
      const int first = y0 + ((my_row - y0) % row_step + row_step) % row_step;     // the first row >= y0 this wavefront owns
      for (int y = first; y <= y1; y += row_step) {
        const uint8_t* m = masks + ((size_t)inst * H + y) * W;
        int* p = panoptic + (size_t)y * W;
        if (vec) {
          for (int x = lane * 4; x < W; x += 256) {
            const uchar4 v = *reinterpret_cast<const uchar4*>(m + x);
            if (!(v.x | v.y | v.z | v.w)) continue;
            int4 q = *reinterpret_cast<int4*>(p + x);
            if (v.x && q.x == 0) q.x = id;
            if (v.y && q.y == 0) q.y = id;
            if (v.z && q.z == 0) q.z = id;
            if (v.w && q.w == 0) q.w = id;
            *reinterpret_cast<int4*>(p + x) = q;
          }
        } else {
          for (int x = lane; x < W; x += 64)
            if (m[x] && p[x] == 0) p[x] = id;
        }
      }
    }
  }
  if (visit >= visits) return;                        // ---- count this visit
  const int inst = order[visit];
  if (scores[inst] < conf) return;   // sorted descending: this and every later visit is past the `break`
  const int y0 = extent[3 * inst + 1], y1 = extent[3 * inst + 2];
  const int first = y0 + ((my_row - y0) % row_step + row_step) % row_step;
  int inter = 0;
  for (int y = first; y <= y1; y += row_step) {
    const uint8_t* m = masks + ((size_t)inst * H + y) * W;
    const int* p = panoptic + (size_t)y * W;
    if (vec) {
      for (int x = lane * 4; x < W; x += 256) {
        const uchar4 v = *reinterpret_cast<const uchar4*>(m + x);
        if (!(v.x | v.y | v.z | v.w)) continue;
        const int4 q = *reinterpret_cast<const int4*>(p + x);
        inter += (v.x && q.x > 0) + (v.y && q.y > 0) + (v.z && q.z > 0) + (v.w && q.w > 0);
      }
    } else {
      for (int x = lane; x < W; x += 64) inter += (m[x] != 0 && p[x] > 0);
    }
  }
  for (int o = 32; o > 0; o >>= 1) inter += __shfl_xor(inter, o);
  if (lane == 0 && inter) atomicAdd(&counters[visit], inter);
}

constexpr int kMaxSem = 256;

__global__ __launch_bounds__(256) void pan_stuff_hist_kernel(const int64_t* __restrict__ sem, const int* __restrict__ panoptic,
                                                             long HW, int S, int* __restrict__ hist /* (2, S) */) {
  __shared__ int present[kMaxSem], freec[kMaxSem];
  for (int i = threadIdx.x; i < S; i += blockDim.x) { present[i] = 0; freec[i] = 0; }
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const int64_t l = sem[i];
    if (l < 0 || l >= S) continue;
    if (!present[l]) present[l] = 1;
    if (panoptic[i] == 0) atomicAdd(&freec[l], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < S; i += blockDim.x) {
    if (present[i]) atomicOr(&hist[i], 1);
    if (freec[i]) atomicAdd(&hist[S + i], freec[i]);
  }
}

__global__ void pan_stuff_assign_kernel(const int* __restrict__ hist, int S, int area_limit, int N,
                                        const int* __restrict__ next_id, int* __restrict__ label_id,
                                        int* __restrict__ seg_table, float* __restrict__ seg_score,
                                        int* __restrict__ num_segments) {
  if (blockIdx.x || threadIdx.x) return;
  int id = next_id[N];
  label_id[0] = 0;   // 0 is the special "thing" class
  for (int l = 1; l < S; ++l) {
    label_id[l] = 0;
    if (!hist[l] || hist[S + l] < area_limit) continue;
    ++id;
    label_id[l] = id;
    int* row = seg_table + 5 * (id - 1);
    row[0] = id; row[1] = 0; row[2] = l; row[3] = -1; row[4] = hist[S + l];
    seg_score[id - 1] = 0.f;
  }
  *num_segments = id;
}

__global__ __launch_bounds__(256) void pan_stuff_paint_kernel(const int64_t* __restrict__ sem, long HW, int S,
                                                              const int* __restrict__ label_id,
                                                              int* __restrict__ panoptic) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (long)gridDim.x * blockDim.x) {
    const int64_t l = sem[i];
    if (l <= 0 || l >= S || panoptic[i] != 0) continue;
    const int id = label_id[l];
    if (id) panoptic[i] = id;
  }
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

int jtsm_oicr_predict_f32(const float* const* logits, const float* const* deltas, int heads, int R, int C1, int Kb,
                          const float* proposals, const float* weights, float scale_clamp, float* probs,
                          float* boxes, void* stream) {
  JTSM_REQUIRE(heads >= 1 && heads <= kMaxHeads, "oicr_predict: 1..%d heads, got %d", kMaxHeads, heads);
  JTSM_REQUIRE(R >= 0 && C1 >= 1, "oicr_predict: bad sizes R=%d C1=%d", R, C1);
  if (R == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && probs, "oicr_predict: null pointer");
  JTSM_REQUIRE(!boxes || (deltas && proposals && weights && Kb >= 1), "oicr_predict: boxes need deltas, proposals, weights");
  HeadPtrs lp = {}, dp = {};
  for (int h = 0; h < heads; ++h) {
    JTSM_REQUIRE(logits[h], "oicr_predict: logits[%d] is null", h);
    lp.p[h] = logits[h];
    if (boxes) {
      JTSM_REQUIRE(deltas[h], "oicr_predict: deltas[%d] is null", h);
      dp.p[h] = deltas[h];
    }
  }
  const float wx = boxes ? weights[0] : 1.f, wy = boxes ? weights[1] : 1.f, ww = boxes ? weights[2] : 1.f,
              wh = boxes ? weights[3] : 1.f;
  hipLaunchKernelGGL(oicr_predict_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, as_stream(stream), lp, dp, heads, R, C1, Kb,
                     proposals, wx, wy, ww, wh, scale_clamp, probs, boxes);
  JTSM_CHECK_LAUNCH("oicr_predict");
  return JTSM_OK;
}

size_t jtsm_batched_nms_workspace_bytes(int n, int num_classes, int max_per_class) {
  if (n <= 0 || num_classes <= 0) return 256;
  return nms_layout(n, num_classes, max_per_class, 0).total;
}

int jtsm_batched_nms_f32(const float* boxes, const float* scores, const int64_t* idxs, int n, int num_classes,
                         int max_per_class, float iou_threshold, int coordinate_trick, int64_t* keep,
                         int32_t* num_keep, int32_t* overflow, void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(n >= 0 && num_classes >= 1 && max_per_class >= 0, "batched_nms: bad sizes");
  JTSM_REQUIRE(num_keep, "batched_nms: num_keep is null");
  JTSM_REQUIRE(coordinate_trick >= 0 && coordinate_trick <= 2, "batched_nms: coordinate_trick must be 0, 1 or 2");
  hipStream_t st = as_stream(stream);
  if (n == 0) {
    JTSM_CHECK_HIP(hipMemsetAsync(num_keep, 0, sizeof(int32_t), st));
    if (overflow) JTSM_CHECK_HIP(hipMemsetAsync(overflow, 0, sizeof(int32_t), st));
    return JTSM_OK;
  }
  JTSM_REQUIRE(boxes && scores && idxs && keep, "batched_nms: null pointer");
  JTSM_REQUIRE(((size_t)boxes & 15) == 0, "batched_nms: boxes must be 16-byte aligned");
  const NmsLayout l = nms_layout(n, num_classes, max_per_class, 0);
  JTSM_REQUIRE(workspace && workspace_bytes >= l.total && ((size_t)workspace & 255) == 0,
               "batched_nms: workspace of %zu bytes (256-byte aligned) needed", l.total);
  JTSM_REQUIRE((size_t)2 * l.W * 8 <= 64 * 1024, "batched_nms: max_per_class %d too large", max_per_class);
  char* ws = reinterpret_cast<char*>(workspace);
  NmsHeader* h = reinterpret_cast<NmsHeader*>(ws + l.header);
  hipLaunchKernelGGL(nms_header_init, dim3(1), dim3(1), 0, st, h);
  hipLaunchKernelGGL(nms_elements_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st,
                     reinterpret_cast<const float4*>(boxes), scores, idxs, n, num_classes,
                     reinterpret_cast<float4*>(ws + l.ebox), reinterpret_cast<float*>(ws + l.escore),
                     reinterpret_cast<int*>(ws + l.eclass), h);
  JTSM_CHECK_LAUNCH("nms elements");
  int rc = nms_core(ws, l, iou_threshold, coordinate_trick, st);
  if (rc) return rc;
  hipLaunchKernelGGL(nms_emit_keep_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, st,
                     reinterpret_cast<const int*>(ws + l.order), n, h, keep, num_keep, overflow);
  JTSM_CHECK_LAUNCH("nms emit");
  return JTSM_OK;
}

size_t jtsm_fast_rcnn_inference_workspace_bytes(int R, int K) {
  if (R <= 0 || K <= 0) return 256;
  return nms_layout(R * K, K, R, R).total;
}

int jtsm_fast_rcnn_inference_f32(const float* boxes, const float* scores, int R, int K, int Kb, float img_h,
                                 float img_w, float score_thresh, float nms_thresh, int topk, int cap,
                                 float* out_boxes, float* out_scores, int64_t* out_classes, int64_t* out_rows,
                                 int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(R >= 0 && K >= 1 && (Kb == 1 || Kb == K), "fast_rcnn_inference: bad sizes R=%d K=%d Kb=%d", R, K, Kb);
  JTSM_REQUIRE((long)R * K < 2147483647L / 64, "fast_rcnn_inference: R*K too large");
  JTSM_REQUIRE(out_count && cap >= 0, "fast_rcnn_inference: null out_count / negative cap");
  hipStream_t st = as_stream(stream);
  if (R == 0 || cap == 0) {
    JTSM_CHECK_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), st));
    return JTSM_OK;
  }
  JTSM_REQUIRE(boxes && scores && out_boxes && out_scores && out_classes && out_rows, "fast_rcnn_inference: null pointer");
  JTSM_REQUIRE(((size_t)out_boxes & 15) == 0, "fast_rcnn_inference: out_boxes must be 16-byte aligned");
  const int N = R * K;
  const NmsLayout l = nms_layout(N, K, R, R);
  JTSM_REQUIRE(workspace && workspace_bytes >= l.total && ((size_t)workspace & 255) == 0,
               "fast_rcnn_inference: workspace of %zu bytes (256-byte aligned) needed", l.total);
  JTSM_REQUIRE((size_t)2 * l.W * 8 <= 64 * 1024, "fast_rcnn_inference: R=%d too large", R);
  char* ws = reinterpret_cast<char*>(workspace);
  NmsHeader* h = reinterpret_cast<NmsHeader*>(ws + l.header);
  uint8_t* valid = reinterpret_cast<uint8_t*>(ws + l.rowvalid);
  float4* ebox = reinterpret_cast<float4*>(ws + l.ebox);
  float* escore = reinterpret_cast<float*>(ws + l.escore);
  hipLaunchKernelGGL(nms_header_init, dim3(1), dim3(1), 0, st, h);
  hipLaunchKernelGGL(det_row_valid_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, st, boxes, Kb * 4, scores, K + 1, R, valid);
  JTSM_CHECK_LAUNCH("det row valid");
  hipLaunchKernelGGL(det_elements_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, boxes, Kb, scores, R, K, valid, img_h,
                     img_w, score_thresh, ebox, escore, reinterpret_cast<int*>(ws + l.eclass), h);
  JTSM_CHECK_LAUNCH("det elements");
  int rc = nms_core(ws, l, nms_thresh, 2, st);
  if (rc) return rc;
  hipLaunchKernelGGL(det_emit_kernel, dim3(ceil_div(cap, 256)), dim3(256), 0, st,
                     reinterpret_cast<const int*>(ws + l.order), ebox, escore, K, topk, cap, h,
                     reinterpret_cast<float4*>(out_boxes), out_scores, out_classes, out_rows, out_count);
  JTSM_CHECK_LAUNCH("det emit");
  return JTSM_OK;
}

int jtsm_mask_probs_f32(const float* const* logits, int heads, const int64_t* classes, int N, int C, int M,
                        float* out, void* stream) {
  JTSM_REQUIRE(heads >= 1 && heads <= kMaxHeads, "mask_probs: 1..%d heads", kMaxHeads);
  JTSM_REQUIRE(N >= 0 && C >= 1 && M >= 1, "mask_probs: bad sizes");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && out && (C == 1 || classes), "mask_probs: null pointer");
  HeadPtrs lp = {};
  for (int h = 0; h < heads; ++h) {
    JTSM_REQUIRE(logits[h], "mask_probs: logits[%d] is null", h);
    lp.p[h] = logits[h];
  }
  const long total = (long)N * M * M;
  hipLaunchKernelGGL(mask_probs_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, as_stream(stream), lp, heads, classes, N,
                     C, M * M, out);
  JTSM_CHECK_LAUNCH("mask_probs");
  return JTSM_OK;
}

int jtsm_paste_masks_f32(const float* masks, const float* boxes, int N, int M, int img_h, int img_w, float threshold,
                         uint8_t* out, void* stream) {
  JTSM_REQUIRE(N >= 0 && M >= 1 && img_h >= 0 && img_w >= 0, "paste_masks: bad sizes");
  if (N == 0 || img_h == 0 || img_w == 0) return JTSM_OK;
  JTSM_REQUIRE(masks && boxes && out, "paste_masks: null pointer");
  JTSM_REQUIRE(N <= 65535 && img_h <= 65535, "paste_masks: at most 65535 masks / rows per call");
  hipLaunchKernelGGL(paste_masks_kernel, dim3(ceil_div(ceil_div(img_w, 4), 256), img_h, N), dim3(256), 0,
                     as_stream(stream), masks, boxes, M, img_h, img_w, threshold, out);
  JTSM_CHECK_LAUNCH("paste_masks");
  return JTSM_OK;
}

int jtsm_resize_bilinear_f32(const float* x, int layout, int N, int C, int H, int W, int crop_h, int crop_w,
                             int out_h, int out_w, float scale_h, float scale_w, float* y, void* stream) {
  JTSM_REQUIRE(layout == JTSM_NCHW || layout == JTSM_NHWC, "resize_bilinear: bad layout");
  JTSM_REQUIRE(N >= 0 && C >= 1 && crop_h >= 1 && crop_w >= 1 && crop_h <= H && crop_w <= W && out_h >= 0 && out_w >= 0,
               "resize_bilinear: bad sizes");
  if (N == 0 || out_h == 0 || out_w == 0) return JTSM_OK;
  JTSM_REQUIRE(x && y, "resize_bilinear: null pointer");
  JTSM_REQUIRE(N <= 65535 && out_h <= 65535, "resize_bilinear: at most 65535 images / rows per call");
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3(ceil_div(out_w, 256), out_h, N), dim3(256), 0, as_stream(stream), x,
                     layout == JTSM_NHWC ? 1 : 0, C, H, W, crop_h, crop_w, out_h, out_w, scale_h, scale_w, y);
  JTSM_CHECK_LAUNCH("resize_bilinear");
  return JTSM_OK;
}

int jtsm_resize_nearest_f32(const float* x, int C, int H, int W, int out_h, int out_w, int flip_source, float* y,
                            void* stream) {
  JTSM_REQUIRE(C >= 1 && H >= 1 && W >= 1 && out_h >= 0 && out_w >= 0, "resize_nearest: bad sizes");
  if (out_h == 0 || out_w == 0) return JTSM_OK;
  JTSM_REQUIRE(x && y, "resize_nearest: null pointer");
  JTSM_REQUIRE(out_h <= 65535, "resize_nearest: at most 65535 rows");
  hipLaunchKernelGGL(resize_nearest_kernel, dim3(ceil_div(out_w, 256), out_h), dim3(256), 0, as_stream(stream), x, C, H, W,
                     out_h, out_w, (float)H / (float)out_h, (float)W / (float)out_w, flip_source ? 1 : 0, y);
  JTSM_CHECK_LAUNCH("resize_nearest");
  return JTSM_OK;
}

int jtsm_argmax_channels_f32(const float* x, int C, long HW, int64_t* out, void* stream) {
  JTSM_REQUIRE(C >= 1 && HW >= 0, "argmax_channels: bad sizes");
  if (HW == 0) return JTSM_OK;
  JTSM_REQUIRE(x && out, "argmax_channels: null pointer");
  hipLaunchKernelGGL(argmax_channels_kernel, dim3(ceil_div(HW, 256)), dim3(256), 0, as_stream(stream), x, C, HW, out);
  JTSM_CHECK_LAUNCH("argmax_channels");
  return JTSM_OK;
}

size_t jtsm_panoptic_combine_workspace_bytes(int N, int S) {
  // counters (N) + next_id (N + 1) + extent (3N) + hist (2S) + label_id (S)
  const size_t n = N > 0 ? N : 0, s = S > 0 ? S : 0;
  return align256((5 * n + 1 + 3 * s + 8) * sizeof(int));
}

int jtsm_panoptic_combine(const uint8_t* masks, const int32_t* order, const float* scores, const int64_t* classes,
                          int N, int H, int W, const int64_t* sem, int S, double overlap_threshold,
                          int stuff_area_limit, float instances_confidence_threshold, int32_t* panoptic,
                          int32_t* seg_table, float* seg_score, int32_t* num_segments, int max_visits,
                          void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(N >= 0 && H >= 1 && W >= 1 && S >= 1 && S <= kMaxSem, "panoptic_combine: bad sizes (S <= %d)", kMaxSem);
  JTSM_REQUIRE(sem && panoptic && seg_table && seg_score && num_segments, "panoptic_combine: null pointer");
  JTSM_REQUIRE(N == 0 || (masks && order && scores && classes), "panoptic_combine: null instance arrays");
  JTSM_REQUIRE((long)N * H < 2147483647L * 4, "panoptic_combine: too many mask rows");
  const size_t need = jtsm_panoptic_combine_workspace_bytes(N, S);
  JTSM_REQUIRE(workspace && workspace_bytes >= need, "panoptic_combine: workspace of %zu bytes needed", need);
  hipStream_t st = as_stream(stream);
  const long HW = (long)H * W;
  int* counters = reinterpret_cast<int*>(workspace);
  int* next_id = counters + N;
  int* extent = next_id + N + 1;
  int* hist = extent + 3 * N;
  int* label_id = hist + 2 * S;
  JTSM_CHECK_HIP(hipMemsetAsync(workspace, 0, need, st));
  JTSM_CHECK_HIP(hipMemsetAsync(panoptic, 0, (size_t)HW * sizeof(int32_t), st));
  const int blocks = (int)std::min<long>(ceil_div(HW, 256), 2048);
  if (N > 0) {
    hipLaunchKernelGGL(pan_extent_init_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, extent, N, H);
    hipLaunchKernelGGL(pan_extent_kernel, dim3(ceil_div((long)N * H, 4)), dim3(256), 0, st, masks, N, H, W, extent);
  }
  // the walk stops at the first score below the confidence threshold: a caller that knows how many instances
  // clear it passes that count and saves the launches of the rest (they would all return at once)
  const int visits = (max_visits >= 0 && max_visits < N) ? max_visits : N;
  for (int v = 0; v <= visits && visits > 0; ++v)      // launch v counts visit v and paints visit v - 1
    hipLaunchKernelGGL(pan_step_kernel, dim3(kPanBlocks), dim3(256), 0, st, masks, order, scores, classes, v, visits, H, W,
                       instances_confidence_threshold, overlap_threshold, extent, counters, next_id, panoptic,
                       seg_table, seg_score);
  JTSM_CHECK_LAUNCH("panoptic instances");
  hipLaunchKernelGGL(pan_stuff_hist_kernel, dim3(blocks), dim3(256), 0, st, sem, panoptic, HW, S, hist);
  hipLaunchKernelGGL(pan_stuff_assign_kernel, dim3(1), dim3(1), 0, st, hist, S, stuff_area_limit, visits, next_id, label_id,
                     seg_table, seg_score, num_segments);
  hipLaunchKernelGGL(pan_stuff_paint_kernel, dim3(blocks), dim3(256), 0, st, sem, HW, S, label_id, panoptic);
  JTSM_CHECK_LAUNCH("panoptic stuff");
  return JTSM_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Model input boundary: GeneralizedMCNNWSL.preprocess_image (mcnn.py:303-318) — (x - mean) / std,
// zero padding to the batch shape, channels-last — from the mapper's uint8 (3, h, w) planes in one launch.
namespace jtsm {
namespace {
constexpr int kMaxImages = 16;
struct ImageTable {
  const void* p[kMaxImages];
  int h[kMaxImages], w[kMaxImages];
};

template <typename SRC>   // uint8_t (the mappers' planes) or float (images that arrive as floating point)
__global__ __launch_bounds__(256) void preprocess_kernel(ImageTable t, int C, int Hp, int Wp, float m0, float m1, float m2,
                                                         float s0, float s1, float s2, float pad,
                                                         float* __restrict__ out) {
#pragma clang fp contract(off)
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= Wp) return;
  const int h = t.h[b], w = t.w[b];
  float* o = out + (((size_t)b * Hp + y) * Wp + x) * C;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  if (y < h && x < w) {
    const SRC* src = static_cast<const SRC*>(t.p[b]) + (size_t)y * w + x;
    for (int c = 0; c < C; ++c) o[c] = ((float)src[(size_t)c * h * w] - mean[c]) / stdv[c];
  } else {
    for (int c = 0; c < C; ++c) o[c] = pad;
  }
}
}  // namespace
}  // namespace jtsm

namespace jtsm {
namespace {
template <typename SRC>
int preprocess_images(const SRC* const* images, const int32_t* heights, const int32_t* widths, int B, int C,
                      const float* mean, const float* stdv, float pad_value, int Hp, int Wp, float* out, void* stream) {
  JTSM_REQUIRE(B >= 0 && (C == 1 || C == 3) && Hp >= 1 && Wp >= 1, "preprocess_images: bad sizes (C must be 1 or 3)");
  if (B == 0) return JTSM_OK;
  JTSM_REQUIRE(images && heights && widths && mean && stdv && out, "preprocess_images: null pointer");
  JTSM_REQUIRE(Hp <= 65535, "preprocess_images: at most 65535 rows");
  for (int b0 = 0; b0 < B; b0 += kMaxImages) {
    const int nb = std::min(kMaxImages, B - b0);
    ImageTable t = {};
    for (int i = 0; i < nb; ++i) {
      JTSM_REQUIRE(images[b0 + i] && heights[b0 + i] >= 0 && widths[b0 + i] >= 0 && heights[b0 + i] <= Hp &&
                       widths[b0 + i] <= Wp, "preprocess_images: image %d does not fit the batch shape", b0 + i);
      t.p[i] = images[b0 + i]; t.h[i] = heights[b0 + i]; t.w[i] = widths[b0 + i];
    }
    hipLaunchKernelGGL(preprocess_kernel<SRC>, dim3(ceil_div(Wp, 256), Hp, nb), dim3(256), 0, as_stream(stream), t, C,
                       Hp, Wp, mean[0], C > 1 ? mean[1] : 0.f, C > 1 ? mean[2] : 0.f, stdv[0], C > 1 ? stdv[1] : 1.f,
                       C > 1 ? stdv[2] : 1.f, pad_value, out + (size_t)b0 * Hp * Wp * C);
    JTSM_CHECK_LAUNCH("preprocess_images");
  }
  return JTSM_OK;
}
}  // namespace
}  // namespace jtsm

extern "C" int jtsm_preprocess_images_u8(const uint8_t* const* images, const int32_t* heights, const int32_t* widths,
                                         int B, int C, const float* mean, const float* stdv, float pad_value, int Hp,
                                         int Wp, float* out, void* stream) {
  return jtsm::preprocess_images<uint8_t>(images, heights, widths, B, C, mean, stdv, pad_value, Hp, Wp, out, stream);
}

extern "C" int jtsm_preprocess_images_f32(const float* const* images, const int32_t* heights, const int32_t* widths,
                                          int B, int C, const float* mean, const float* stdv, float pad_value, int Hp,
                                          int Wp, float* out, void* stream) {
  return jtsm::preprocess_images<float>(images, heights, widths, B, C, mean, stdv, pad_value, Hp, Wp, out, stream);
}
