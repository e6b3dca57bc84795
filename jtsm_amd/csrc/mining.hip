// Pseudo-ground-truth mining and proposal labelling of JTSMROIHeads, fused.
//
// Replaces, per refinement round, the chain of ~150 small PyTorch launches (and several host syncs) of
//   get_pgt_top_k            projects/WSL/wsl/modeling/roi_heads/roi_heads_jtsm.py:1167-1338 (top_k = 1)
//   label_and_sample_proposals  projects/WSL/wsl/modeling/roi_heads/roi_heads.py:264-370
//     (pairwise_iou detectron2/structures/boxes.py:345-392, Matcher([0.5],[0,1]) modeling/matcher.py:61-103)
//   predict_probs / predict_boxes  .../fast_rcnn_oicr.py:684-783 (softmax, Box2BoxTransform.apply_deltas
//     detectron2/modeling/box_regression.py:73-113) — evaluated only where they are consumed
// by three launches: row log-sum-exp, one block-wide arg-max per (image, present class), and one
// IoU-match per proposal.  Results are integers (winning rows, labels, matched indices) plus the copied
// pseudo boxes / weights, in exactly the layout the OICR loss kernel takes.
// Floating-point steps that decide integers (IoU, box decoding) are evaluated un-contracted, in the
// reference's operation order.
#include <cfloat>

#include "common.h"

namespace jtsm {
namespace {

#pragma clang fp contract(off)

// lse[r] = log(sum_c exp(z[r,c]))   (wavefront per row, any ncls)
__global__ __launch_bounds__(256) void row_lse_kernel(const float* __restrict__ z, int ld, int ncls, int R,
                                                      float* __restrict__ lse) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float* row = z + (size_t)r * ld;
  float mx = -FLT_MAX;
  for (int c = lane; c < ncls; c += 64) mx = fmaxf(mx, row[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sm = 0.f;
  for (int c = lane; c < ncls; c += 64) sm += expf(row[c] - mx);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
  if (lane == 0) lse[r] = mx + logf(sm);
}

__device__ __forceinline__ void decode_box(const float* __restrict__ p, const float* __restrict__ d, float* out) {
#pragma clang fp contract(off)
  // Box2BoxTransform(10,10,5,5).apply_deltas for one (box, class) pair
  const float w = p[2] - p[0], h = p[3] - p[1];
  const float cx = p[0] + 0.5f * w, cy = p[1] + 0.5f * h;
  const float dx = d[0] / 10.f, dy = d[1] / 10.f;
  const float kClamp = 4.135166556742356f;  // log(1000/16)
  const float dw = fminf(d[2] / 5.f, kClamp), dh = fminf(d[3] / 5.f, kClamp);
  const float pcx = dx * w + cx, pcy = dy * h + cy;
  const float pw = expf(dw) * w, ph = expf(dh) * h;
  out[0] = pcx - 0.5f * pw;
  out[1] = pcy - 0.5f * ph;
  out[2] = pcx + 0.5f * pw;
  out[3] = pcy + 0.5f * ph;
}

// One workgroup per (image, class slot): arg-max over the image's proposals of the class score
// (score = scores[r,cls], or exp(scores[r,cls] - lse[r]) when lse is given); lowest row wins ties.
__global__ __launch_bounds__(256) void mine_top1_kernel(const float* __restrict__ scores, int ld,
                                                        const float* __restrict__ lse,
                                                        const float* __restrict__ proposals,
                                                        const float* __restrict__ deltas, int ld_d,
                                                        const int* __restrict__ bag_off,
                                                        const int* __restrict__ classes,
                                                        const int* __restrict__ counts, int Gmax,
                                                        const float* __restrict__ img_probs, int nprob,
                                                        int* __restrict__ out_idx, float* __restrict__ out_box,
                                                        float* __restrict__ out_score,
                                                        float* __restrict__ out_weight) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int img = blockIdx.x / Gmax, g = blockIdx.x - img * Gmax;
  if (g >= counts[img]) return;
  const int cls = classes[img * Gmax + g];
  const int r0 = bag_off[img], r1 = bag_off[img + 1];
  float best = -FLT_MAX;
  int at = 0x7fffffff;
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    float v = scores[(size_t)r * ld + cls];
    if (lse) v = expf(v - lse[r]);
    if (v > best) { best = v; at = r; }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = at;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float v = sv[threadIdx.x + o];
      const int i = si[threadIdx.x + o];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int r = si[0] == 0x7fffffff ? r0 : si[0];
    const int o = img * Gmax + g;
    out_idx[o] = r - r0;
    out_score[o] = sv[0];
    out_weight[o] = img_probs[(size_t)img * nprob + cls];
    if (deltas) decode_box(proposals + 4 * (size_t)r, deltas + (size_t)r * ld_d + 4 * cls, out_box + 4 * o);
    else for (int k = 0; k < 4; ++k) out_box[4 * o + k] = proposals[4 * (size_t)r + k];
  }
}

// One thread per proposal: IoU against the image's pseudo boxes, first maximum wins; label = class of
// the best box if IoU >= thresh else bg_label; also copies that box / weight / score.
__global__ __launch_bounds__(256) void match_label_kernel(const float* __restrict__ proposals,
                                                          const int* __restrict__ bag_off, int B, int R,
                                                          const float* __restrict__ pgt_box,
                                                          const int* __restrict__ classes,
                                                          const int* __restrict__ counts,
                                                          const float* __restrict__ pgt_weight,
                                                          const float* __restrict__ pgt_score, int Gmax,
                                                          float thresh, int bg_label, int* __restrict__ labels,
                                                          int* __restrict__ matched, float* __restrict__ gt_boxes,
                                                          float* __restrict__ gt_weights,
                                                          float* __restrict__ gt_scores) {
#pragma clang fp contract(off)
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  int img = 0;
  while (img + 1 < B && r >= bag_off[img + 1]) ++img;
  const float* p = proposals + 4 * (size_t)r;
  const float area_p = (p[2] - p[0]) * (p[3] - p[1]);
  const int n = counts[img];
  float best = -1.f;
  int at = 0;
  for (int g = 0; g < n; ++g) {
    const float* q = pgt_box + 4 * ((size_t)img * Gmax + g);
    const float area_q = (q[2] - q[0]) * (q[3] - q[1]);
    const float w = fmaxf(fminf(q[2], p[2]) - fmaxf(q[0], p[0]), 0.f);
    const float h = fmaxf(fminf(q[3], p[3]) - fmaxf(q[1], p[1]), 0.f);
    const float inter = w * h;
    const float iou = inter > 0.f ? inter / (area_q + area_p - inter) : 0.f;
    if (iou > best) { best = iou; at = g; }
  }
  const int o = img * Gmax + at;
  if (n == 0) {
    labels[r] = bg_label;
    matched[r] = 0;
    for (int k = 0; k < 4; ++k) gt_boxes[4 * (size_t)r + k] = p[k];
    gt_weights[r] = 0.f;
    if (gt_scores) gt_scores[r] = 0.f;
    return;
  }
  labels[r] = best >= thresh ? classes[o] : bg_label;
  matched[r] = at;
  for (int k = 0; k < 4; ++k) gt_boxes[4 * (size_t)r + k] = pgt_box[4 * (size_t)o + k];
  gt_weights[r] = pgt_weight[o];
  if (gt_scores) gt_scores[r] = pgt_score[o];
}

// ---- the "10 nearest" targets of the mask branch (roi_heads_jtsm.py:840-905): for every pseudo-GT box, the top_k
// FOREGROUND proposals of its image by IoU with it (torch.topk over pairwise_iou(targets, fg proposals), k = min(#fg,
// top_k)); ordering here: IoU descending, equal IoUs by ascending proposal row.  One workgroup per (image, pseudo box);
// round j picks the smallest key (−IoU, row) greater than round j-1's.  near_rows (B, Gmax, K): proposal rows, -1 pad.
__device__ __forceinline__ float iou_tp(const float* __restrict__ q /* target */, const float* __restrict__ p) {
#pragma clang fp contract(off)
  const float area_q = (q[2] - q[0]) * (q[3] - q[1]), area_p = (p[2] - p[0]) * (p[3] - p[1]);
  const float w = fmaxf(fminf(q[2], p[2]) - fmaxf(q[0], p[0]), 0.f);
  const float h = fmaxf(fminf(q[3], p[3]) - fmaxf(q[1], p[1]), 0.f);
  const float inter = w * h;
  return inter > 0.f ? inter / (area_q + area_p - inter) : 0.f;
}

__global__ __launch_bounds__(256) void near_targets_kernel(const float* __restrict__ proposals,
                                                           const int* __restrict__ bag_off,
                                                           const int* __restrict__ labels, int bg_label,
                                                           const float* __restrict__ pgt_box,
                                                           const int* __restrict__ counts, int Gmax, int K,
                                                           int* __restrict__ near_rows) {
  __shared__ float s_iou[256];
  __shared__ int s_row[256];
  __shared__ float last_iou;
  __shared__ int last_row;
  const int img = blockIdx.x / Gmax, g = blockIdx.x % Gmax;
  int* out = near_rows + (size_t)blockIdx.x * K;
  if (g >= counts[img]) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) out[k] = -1;
    return;
  }
  const int r0 = bag_off[img], r1 = bag_off[img + 1];
  const float* q = pgt_box + 4 * ((size_t)img * Gmax + g);
  if (threadIdx.x == 0) { last_iou = 2.f; last_row = -1; }
  __syncthreads();
  for (int k = 0; k < K; ++k) {
    const float li = last_iou;
    const int lr = last_row;
    float best = -1.f;
    int at = -1;
    for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
      if (labels[r] == bg_label || labels[r] < 0) continue;           // foreground proposals only
      const float v = iou_tp(q, proposals + 4 * (size_t)r);
      const bool after = v < li || (v == li && r > lr);               // strictly after the previous pick
      if (after && (v > best || (v == best && r < at))) { best = v; at = r; }
    }
    s_iou[threadIdx.x] = best;
    s_row[threadIdx.x] = at;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if (threadIdx.x < st) {
        const float a = s_iou[threadIdx.x], b = s_iou[threadIdx.x + st];
        const int ra = s_row[threadIdx.x], rb = s_row[threadIdx.x + st];
        if (rb >= 0 && (ra < 0 || b > a || (b == a && rb < ra))) { s_iou[threadIdx.x] = b; s_row[threadIdx.x] = rb; }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      out[k] = s_row[0];
      last_iou = s_iou[0];
      last_row = s_row[0] >= 0 ? s_row[0] : 0x7fffffff;   // exhausted: nothing comes after
      if (s_row[0] < 0) last_iou = -2.f;
    }
    __syncthreads();
  }
}

// Second labelling pass against the near targets (label_and_sample_proposals(instances, near_targets),
// roi_heads_jtsm.py:893-905 -> roi_heads.py:306-340): each foreground proposal takes the mask of the near target with
// the highest IoU (first maximum in list order: pseudo box major, rank minor).  matched_near[r] = that target's
// proposal row (-1 for non-foreground proposals / images without targets).
__global__ __launch_bounds__(256) void match_near_kernel(const float* __restrict__ proposals, const int* __restrict__ bag_off,
                                                         int B, int R, const int* __restrict__ labels, int bg_label,
                                                         const int* __restrict__ near_rows, const int* __restrict__ counts,
                                                         int Gmax, int K, int* __restrict__ matched_near) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  if (labels[r] == bg_label || labels[r] < 0) { matched_near[r] = -1; return; }
  int img = 0;
  while (img + 1 < B && r >= bag_off[img + 1]) ++img;
  const float* p = proposals + 4 * (size_t)r;
  float best = -1.f;
  int at = -1;
  const int n = counts[img];
  for (int g = 0; g < n; ++g)
    for (int k = 0; k < K; ++k) {
      const int row = near_rows[((size_t)img * Gmax + g) * K + k];
      if (row < 0) break;
      const float v = iou_tp(proposals + 4 * (size_t)row, p);
      if (v > best) { best = v; at = row; }
    }
  matched_near[r] = at;
}

// ---- pseudo semantic target (get_pgt_sem_seg, roi_heads_jtsm.py:2025-2070, with the rectangle substitution of
// SURVEY F8): every pseudo box paints its rectangle shrunk by `erode` pixels with value class - class_base, in
// ascending score order (the best box ends on top); then, in list order, a class whose pixels were all painted
// over is painted once more.  Pass A is pixel-parallel (max score RANK among the covering boxes) and counts the
// pixels of every value; pass B (one workgroup per image) replays the sequential "missing class" rule on those
// counts, touching only the rectangles it repaints.  No host synchronisation: list lengths stay on the device.
__device__ __forceinline__ bool paint_inside(const float* __restrict__ b, int x, int y, float erode) {
  const float xs = (float)x + 0.5f, ys = (float)y + 0.5f;
  return xs >= b[0] + erode && xs <= b[2] - erode && ys >= b[1] + erode && ys <= b[3] - erode;
}

constexpr int kPaintValues = 64;   // painted values are 1 .. 63 (0 = untouched)

__global__ __launch_bounds__(256) void paint_top_kernel(const float* __restrict__ boxes, const int* __restrict__ classes,
                                                        const float* __restrict__ scores, const int* __restrict__ counts,
                                                        int G, int class_base, int H, int W, float erode,
                                                        long* __restrict__ out, int* __restrict__ value_counts) {
  __shared__ int hist[kPaintValues];
  __shared__ float sb[64 * 4];
  __shared__ int srank[64], sval[64];
  const int b = blockIdx.y, n = min(counts[b], 64);
  if (threadIdx.x < kPaintValues) hist[threadIdx.x] = 0;
  if (threadIdx.x < n) {
    const int j = threadIdx.x;
    const float sj = scores[b * G + j];
    int r = 0;   // position of box j in ascending (score, index) order
    for (int k = 0; k < n; ++k) {
      const float sk = scores[b * G + k];
      r += (sk < sj || (sk == sj && k < j)) ? 1 : 0;
    }
    srank[j] = r;
    sval[j] = classes[b * G + j] - class_base;
    for (int e = 0; e < 4; ++e) sb[j * 4 + e] = boxes[((size_t)b * G + j) * 4 + e];
  }
  __syncthreads();
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p < (long)H * W) {
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    int best = -1, val = 0;
    for (int j = 0; j < n; ++j)
      if (paint_inside(sb + j * 4, x, y, erode) && srank[j] > best) { best = srank[j]; val = sval[j]; }
    out[(size_t)b * H * W + p] = val;
    atomicAdd(&hist[val & (kPaintValues - 1)], 1);
  }
  __syncthreads();
  if (threadIdx.x < kPaintValues && hist[threadIdx.x]) atomicAdd(value_counts + b * kPaintValues + threadIdx.x, hist[threadIdx.x]);
}

__global__ __launch_bounds__(256) void paint_missing_kernel(const float* __restrict__ boxes, const int* __restrict__ classes,
                                                            const int* __restrict__ counts, int G, int class_base, int H,
                                                            int W, float erode, long* __restrict__ out,
                                                            const int* __restrict__ value_counts) {
  __shared__ int cnt[kPaintValues];
  const int b = blockIdx.x, n = min(counts[b], 64);
  if (threadIdx.x < kPaintValues) cnt[threadIdx.x] = value_counts[b * kPaintValues + threadIdx.x];
  __syncthreads();
  long* img = out + (size_t)b * H * W;
  for (int j = 0; j < n; ++j) {
    const int v = (classes[b * G + j] - class_base) & (kPaintValues - 1);
    const bool missing = cnt[v] == 0;   // uniform: read before anyone updates it in this round
    __syncthreads();
    if (missing) {
      const float* bx = boxes + ((size_t)b * G + j) * 4;
      const int x0 = max(0, (int)floorf(bx[0] + erode - 0.5f)), x1 = min(W - 1, (int)ceilf(bx[2] - erode - 0.5f));
      const int y0 = max(0, (int)floorf(bx[1] + erode - 0.5f)), y1 = min(H - 1, (int)ceilf(bx[3] - erode - 0.5f));
      const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
      if (bw > 0 && bh > 0)
        for (long i = threadIdx.x; i < (long)bw * bh; i += 256) {
          const int y = y0 + (int)(i / bw), x = x0 + (int)(i % bw);
          if (!paint_inside(bx, x, y, erode)) continue;
          const long old = img[(size_t)y * W + x];
          img[(size_t)y * W + x] = v;
          atomicSub(&cnt[(int)old & (kPaintValues - 1)], 1);
          atomicAdd(&cnt[v], 1);
        }
    }
    __syncthreads();
  }
}


// ---- pseudo semantic target with the reference's masks (get_pgt_sem_seg :2038-2069 with need_mask=True ->
// get_pgt_top_k :1333-1334 -> object_evidence :1928-1994, superpixel branch): target j's mask is the union of the
// superpixels its oh_labels row marks, mask_j(y, x) = oh_labels[row_j][superpixels[y][x]] != 0 (a superpixel id outside
// [0, L) belongs to no mask: the reference's `poses` loop covers range(L) only).  Same two passes as above; the
// full-image masks are never rasterised.
__global__ __launch_bounds__(256) void paint_top_evidence_kernel(
    const int* __restrict__ target_idx, const int* __restrict__ bag_offsets, const int* __restrict__ oh_labels, int L,
    const int* __restrict__ superpixels, const int* __restrict__ classes, const float* __restrict__ scores,
    const int* __restrict__ counts, int G, int class_base, int H, int W, long* __restrict__ out,
    int* __restrict__ value_counts) {
  __shared__ int hist[kPaintValues];
  __shared__ int srank[64], sval[64];
  __shared__ long srow[64];
  const int b = blockIdx.y, n = min(counts[b], 64);
  if (threadIdx.x < kPaintValues) hist[threadIdx.x] = 0;
  if (threadIdx.x < n) {
    const int j = threadIdx.x;
    const float sj = scores[b * G + j];
    int r = 0;   // position of target j in ascending (score, index) order
    for (int k = 0; k < n; ++k) {
      const float sk = scores[b * G + k];
      r += (sk < sj || (sk == sj && k < j)) ? 1 : 0;
    }
    srank[j] = r;
    sval[j] = classes[b * G + j] - class_base;
    srow[j] = ((long)bag_offsets[b] + target_idx[b * G + j]) * L;
  }
  __syncthreads();
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p < (long)H * W) {
    const int s = superpixels[(size_t)b * H * W + p];
    int best = -1, val = 0;
    if (s >= 0 && s < L)
      for (int j = 0; j < n; ++j)
        if (srank[j] > best && oh_labels[srow[j] + s] != 0) { best = srank[j]; val = sval[j]; }
    out[(size_t)b * H * W + p] = val;
    atomicAdd(&hist[val & (kPaintValues - 1)], 1);
  }
  __syncthreads();
  if (threadIdx.x < kPaintValues && hist[threadIdx.x]) atomicAdd(value_counts + b * kPaintValues + threadIdx.x, hist[threadIdx.x]);
}

__global__ __launch_bounds__(1024) void paint_missing_evidence_kernel(
    const int* __restrict__ target_idx, const int* __restrict__ bag_offsets, const int* __restrict__ oh_labels, int L,
    const int* __restrict__ superpixels, const int* __restrict__ classes, const int* __restrict__ counts, int G,
    int class_base, int H, int W, long* __restrict__ out, const int* __restrict__ value_counts) {
  __shared__ int cnt[kPaintValues];
  const int b = blockIdx.x, n = min(counts[b], 64);
  if (threadIdx.x < kPaintValues) cnt[threadIdx.x] = value_counts[b * kPaintValues + threadIdx.x];
  __syncthreads();
  long* img = out + (size_t)b * H * W;
  const int* sp = superpixels + (size_t)b * H * W;
  for (int j = 0; j < n; ++j) {
    const int v = (classes[b * G + j] - class_base) & (kPaintValues - 1);
    const bool missing = cnt[v] == 0;   // uniform: read before anyone updates it in this round
    __syncthreads();
    if (missing) {
      const int* row = oh_labels + ((long)bag_offsets[b] + target_idx[b * G + j]) * L;
      for (long i = threadIdx.x; i < (long)H * W; i += 1024) {
        const int s = sp[i];
        if (s < 0 || s >= L || row[s] == 0) continue;
        const long old = img[i];
        img[i] = v;
        atomicSub(&cnt[(int)old & (kPaintValues - 1)], 1);
        atomicAdd(&cnt[v], 1);
      }
    }
    __syncthreads();
  }
}


// ---------------------------------------------------------------------------------------------------------------
// Label preparation in three launches (round 3).  What these replace were ~80 PyTorch device launches of a few hundred
// threads each per step (arange / where / sort / remainder / index_put / sqrt / log2 / floor / clamp / cat ...), 2-5 us
// apiece on a busy device.  Per-image inputs arrive as up to kMaxImages pointers BY VALUE in the kernel arguments: no
// concatenation and no table upload.
constexpr int kMaxImages = 16;
struct ImageRows {
  const void* ptr[kMaxImages];
  int first[kMaxImages + 1];      // first[b] .. first[b + 1]: the rows of image b in the concatenation
};
__device__ __forceinline__ int image_of_row(const ImageRows& im, int B, int m) {
  int b = 0;
  while (b + 1 < B && m >= im.first[b + 1]) ++b;
  return b;
}

// detectron2 poolers.py:22-58 + 61-95 as ONE pass: rois (M,5) = [image, x0, y0, x1, y1] and the 0-based level
//   floor(canonical_level + log2(sqrt(area) / canonical_size + 1e-8)) clamped to [min_level, max_level] (NaN -> min).
// The arithmetic is PyTorch's, operation by operation (its division by a host scalar is a multiplication by the
// rounded reciprocal), so the levels are the ones `assign_boxes_to_levels` computes.
__global__ __launch_bounds__(256) void pooler_rois_levels_kernel(const ImageRows im, int B, int M, float inv_canonical,
                                                                 float canonical_level, int min_level, int max_level,
                                                                 float* __restrict__ rois, int* __restrict__ level) {
#pragma clang fp contract(off)
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int b = image_of_row(im, B, m);
  const float4 bx = reinterpret_cast<const float4*>(im.ptr[b])[m - im.first[b]];
  rois[(size_t)m * 5 + 0] = (float)b;
  rois[(size_t)m * 5 + 1] = bx.x; rois[(size_t)m * 5 + 2] = bx.y;
  rois[(size_t)m * 5 + 3] = bx.z; rois[(size_t)m * 5 + 4] = bx.w;
  const float area = (bx.z - bx.x) * (bx.w - bx.y);
  float lv = floorf(canonical_level + log2f(sqrtf(area) * inv_canonical + 1e-8f));
  if (lv != lv) lv = (float)min_level;
  lv = fminf(fmaxf(lv, (float)min_level), (float)max_level);
  level[m] = (int)lv - min_level;
}

// roi_heads_jtsm.py:607-633 glue: scale[m] = bins / (nvalid[m] + 1) * (objectness[m] + 1), nvalid = the bins of roi m
// whose MOIPool arg-max (channel 0) is not -1.  arg: (M, bins, C) int32 (channels-last).  Operation order as in the
// PyTorch expression it replaces: (bins * (1 / (nvalid + 1))) * (objectness + 1).
__global__ __launch_bounds__(256) void roi_scale_kernel(const int* __restrict__ arg, int bins, int C, const ImageRows obj,
                                                        int B, int M, float* __restrict__ out) {
#pragma clang fp contract(off)
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  int nvalid = 0;
  for (int i = 0; i < bins; ++i) nvalid += arg[((size_t)m * bins + i) * C] != -1;
  const int b = image_of_row(obj, B, m);
  const float o = reinterpret_cast<const float*>(obj.ptr[b])[m - obj.first[b]];
  out[m] = ((float)bins * (1.0f / ((float)nvalid + 1.0f))) * (o + 1.0f);
}

// which of the labels 0..255 occur in each image's semantic map (values outside are clamped, as the scatter it
// replaces clamps them).  flags (B,256) int32, zeroed by the caller.
template <typename T>
__global__ __launch_bounds__(256) void sem_label_flags_kernel(const T* __restrict__ sem, long pixels, int* __restrict__ flags) {
  __shared__ int seen[256];
  seen[threadIdx.x] = 0;
  __syncthreads();
  const T* img = sem + (size_t)blockIdx.y * pixels;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pixels; i += (long)gridDim.x * blockDim.x) {
    long v = (long)img[i];
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    seen[v] = 1;            // (racing stores of the same value)
  }
  __syncthreads();
  if (seen[threadIdx.x]) flags[blockIdx.y * 256 + threadIdx.x] = 1;
}

// one-hot row, class list (present classes ascending, then the absent ones ascending, each + offset: what
// sort(where(present, c, c + C)) % C + offset gives) and count of one image.  C <= 256.
__device__ __forceinline__ void emit_class_list(bool present, int c, int C, int offset, float* __restrict__ oh,
                                                int* __restrict__ cls, int* __restrict__ cnt, int* wave_count) {
  int npresent, nabsent;
  const bool valid = c < C;
  const int ps = compact256(valid && present, wave_count, npresent);
  const int as = compact256(valid && !present, wave_count, nabsent);
  if (valid) {
    oh[c] = present ? 1.f : 0.f;
    cls[present ? ps : npresent + as] = c + offset;
  }
  if (c == 0) *cnt = npresent;
}
__global__ __launch_bounds__(256) void image_class_lists_kernel(const ImageRows things, int C, const int* __restrict__ stuff_flags,
                                                                int S, int stuff_offset, float* __restrict__ oh_things,
                                                                int* __restrict__ things_cls, int* __restrict__ things_cnt,
                                                                float* __restrict__ oh_stuff, int* __restrict__ stuff_cls,
                                                                int* __restrict__ stuff_cnt) {
  __shared__ int seen[256];
  __shared__ int wave_count[4];
  const int b = blockIdx.x, t = threadIdx.x;
  seen[t] = 0;
  __syncthreads();
  const long* gt = reinterpret_cast<const long*>(things.ptr[b]);
  const int n = things.first[b + 1] - things.first[b];
  for (int i = t; i < n; i += 256) {
    const long c = gt[i];
    if (c >= 0 && c < C) seen[c] = 1;
  }
  __syncthreads();
  emit_class_list(seen[t] != 0, t, C, 0, oh_things + (size_t)b * C, things_cls + (size_t)b * C, things_cnt + b, wave_count);
  if (stuff_flags)   // stuff label l = 1 .. S - 1 is column l - 1 (0 = things and 255 = ignore do not count)
    emit_class_list(t + 1 < 256 && stuff_flags[b * 256 + t + 1] != 0, t, S - 1, stuff_offset, oh_stuff + (size_t)b * (S - 1),
                    stuff_cls + (size_t)b * (S - 1), stuff_cnt + b, wave_count);
}


// The mask branch trains on the foreground proposals only (roi_heads_jtsm.py:754-948): their rows, in row order, with
// what the branch reads of them — one workgroup, an ordered compaction 1024 rows at a time.  fg_* hold `total` valid
// entries (a host-side slice after the count has crossed: no kernel), counts[b] = foreground rows of image b.
__global__ __launch_bounds__(1024) void fg_compact_kernel(const int* __restrict__ labels, int bg_label, const int* __restrict__ bag_offsets,
                                                          int B, int R, const float* __restrict__ boxes,
                                                          const int* __restrict__ matched, int* __restrict__ fg_rows,
                                                          float* __restrict__ fg_boxes, long* __restrict__ fg_classes,
                                                          int* __restrict__ fg_img, int* __restrict__ fg_matched,
                                                          float* __restrict__ fg_rois, long* __restrict__ counts) {
  __shared__ int wave_count[16];
  __shared__ int per_image[64];
  const int t = threadIdx.x;
  if (t < 64) per_image[t] = 0;
  __syncthreads();
  int filled = 0;
  for (int base = 0; base < R; base += 1024) {
    const int r = base + t;
    const int lab = r < R ? labels[r] : bg_label;
    const bool fg = lab != bg_label;
    int cnt;
    const int slot = filled + compact_wg<16>(fg, wave_count, cnt);
    if (fg) {
      int b = 0;
      while (b + 1 < B && r >= bag_offsets[b + 1]) ++b;
      const float4 bx = reinterpret_cast<const float4*>(boxes)[r];
      fg_rows[slot] = r;
      reinterpret_cast<float4*>(fg_boxes)[slot] = bx;
      fg_classes[slot] = lab;
      fg_img[slot] = b;
      if (matched) fg_matched[slot] = matched[r];
      fg_rois[(size_t)slot * 5 + 0] = (float)b;
      fg_rois[(size_t)slot * 5 + 1] = bx.x; fg_rois[(size_t)slot * 5 + 2] = bx.y;
      fg_rois[(size_t)slot * 5 + 3] = bx.z; fg_rois[(size_t)slot * 5 + 4] = bx.w;
      atomicAdd(&per_image[b], 1);
    }
    filled += cnt;
  }
  __syncthreads();
  if (t < B) counts[t] = per_image[t];
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

int jtsm_row_lse_f32(const float* logits, int ld, int ncls, int R, float* lse, void* stream) {
  JTSM_REQUIRE(R >= 0 && ncls > 0 && ld >= ncls, "row_lse: bad sizes");
  if (R == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && lse, "row_lse: null pointer");
  hipLaunchKernelGGL(row_lse_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, as_stream(stream), logits, ld, ncls, R, lse);
  JTSM_CHECK_LAUNCH("row_lse");
  return JTSM_OK;
}

int jtsm_mine_top1_f32(const float* scores, int ld, const float* lse, const float* proposals, const float* deltas,
                       int ld_deltas, const int32_t* bag_offsets, const int32_t* classes, const int32_t* counts,
                       int B, int Gmax, const float* img_probs, int nprob, int32_t* out_idx, float* out_box,
                       float* out_score, float* out_weight, void* stream) {
  JTSM_REQUIRE(B >= 0 && Gmax >= 0, "mine_top1: bad sizes");
  if (B == 0 || Gmax == 0) return JTSM_OK;
  JTSM_REQUIRE(scores && proposals && bag_offsets && classes && counts && img_probs && out_idx && out_box &&
               out_score && out_weight, "mine_top1: null pointer");
  hipLaunchKernelGGL(mine_top1_kernel, dim3(B * Gmax), dim3(256), 0, as_stream(stream), scores, ld, lse, proposals,
                     deltas, ld_deltas, bag_offsets, classes, counts, Gmax, img_probs, nprob, out_idx, out_box,
                     out_score, out_weight);
  JTSM_CHECK_LAUNCH("mine_top1");
  return JTSM_OK;
}

int jtsm_match_label_f32(const float* proposals, const int32_t* bag_offsets, int B, int R, const float* pgt_box,
                         const int32_t* classes, const int32_t* counts, const float* pgt_weight,
                         const float* pgt_score, int Gmax, float iou_thresh, int bg_label, int32_t* labels,
                         int32_t* matched, float* gt_boxes, float* gt_weights, float* gt_scores, void* stream) {
  JTSM_REQUIRE(B > 0 && R >= 0 && Gmax >= 0, "match_label: bad sizes");
  if (R == 0) return JTSM_OK;
  JTSM_REQUIRE(proposals && bag_offsets && pgt_box && classes && counts && pgt_weight && labels && matched &&
               gt_boxes && gt_weights, "match_label: null pointer");
  hipLaunchKernelGGL(match_label_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, as_stream(stream), proposals,
                     bag_offsets, B, R, pgt_box, classes, counts, pgt_weight, pgt_score, Gmax, iou_thresh, bg_label,
                     labels, matched, gt_boxes, gt_weights, gt_scores);
  JTSM_CHECK_LAUNCH("match_label");
  return JTSM_OK;
}

int jtsm_near_targets_f32(const float* proposals, const int32_t* bag_offsets, int B, int R, const int32_t* labels,
                          int bg_label, const float* pgt_box, const int32_t* counts, int Gmax, int top_k,
                          int32_t* near_rows, int32_t* matched_near, void* stream) {
  JTSM_REQUIRE(B > 0 && R >= 0 && Gmax > 0 && top_k > 0 && top_k <= 64, "near_targets: bad sizes");
  JTSM_REQUIRE(proposals && bag_offsets && labels && pgt_box && counts && near_rows && matched_near,
               "near_targets: null pointer");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(near_targets_kernel, dim3(B * Gmax), dim3(256), 0, st, proposals, bag_offsets, labels, bg_label,
                     pgt_box, counts, Gmax, top_k, near_rows);
  if (R > 0)
    hipLaunchKernelGGL(match_near_kernel, dim3(ceil_div(R, 256)), dim3(256), 0, st, proposals, bag_offsets, B, R, labels,
                       bg_label, near_rows, counts, Gmax, top_k, matched_near);
  JTSM_CHECK_LAUNCH("near_targets");
  return JTSM_OK;
}

size_t jtsm_paint_sem_seg_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * kPaintValues * sizeof(int) + 16; }

int jtsm_paint_sem_seg(const float* boxes, const int32_t* classes, const float* scores, const int32_t* counts, int B,
                       int G, int class_base, int H, int W, float erode, int64_t* out, void* workspace, void* stream) {
  JTSM_REQUIRE(B >= 0 && G > 0 && G <= 64 && H > 0 && W > 0, "paint_sem_seg: bad sizes (at most 64 boxes per image)");
  if (B == 0) return JTSM_OK;
  JTSM_REQUIRE(boxes && classes && scores && counts && out && workspace, "paint_sem_seg: null pointer");
  hipStream_t st = as_stream(stream);
  int* value_counts = reinterpret_cast<int*>(workspace);
  JTSM_CHECK_HIP(hipMemsetAsync(value_counts, 0, (size_t)B * kPaintValues * sizeof(int), st));
  const long hw = (long)H * W;
  hipLaunchKernelGGL(paint_top_kernel, dim3((unsigned)((hw + 255) / 256), B), dim3(256), 0, st, boxes, classes, scores,
                     counts, G, class_base, H, W, erode, reinterpret_cast<long*>(out), value_counts);
  hipLaunchKernelGGL(paint_missing_kernel, dim3(B), dim3(256), 0, st, boxes, classes, counts, G, class_base, H, W, erode,
                     reinterpret_cast<long*>(out), value_counts);
  JTSM_CHECK_LAUNCH("paint_sem_seg");
  return JTSM_OK;
}

int jtsm_paint_sem_seg_evidence(const int32_t* target_idx, const int32_t* bag_offsets, const int32_t* oh_labels, int L,
                                const int32_t* superpixels, const int32_t* classes, const float* scores,
                                const int32_t* counts, int B, int G, int class_base, int H, int W, int64_t* out,
                                void* workspace, void* stream) {
  JTSM_REQUIRE(B >= 0 && G > 0 && G <= 64 && H > 0 && W > 0 && L > 0,
               "paint_sem_seg_evidence: bad sizes (at most 64 targets per image)");
  if (B == 0) return JTSM_OK;
  JTSM_REQUIRE(target_idx && bag_offsets && oh_labels && superpixels && classes && scores && counts && out && workspace,
               "paint_sem_seg_evidence: null pointer");
  hipStream_t st = as_stream(stream);
  int* value_counts = reinterpret_cast<int*>(workspace);
  JTSM_CHECK_HIP(hipMemsetAsync(value_counts, 0, (size_t)B * kPaintValues * sizeof(int), st));
  const long hw = (long)H * W;
  hipLaunchKernelGGL(paint_top_evidence_kernel, dim3((unsigned)((hw + 255) / 256), B), dim3(256), 0, st, target_idx,
                     bag_offsets, oh_labels, L, superpixels, classes, scores, counts, G, class_base, H, W,
                     reinterpret_cast<long*>(out), value_counts);
  hipLaunchKernelGGL(paint_missing_evidence_kernel, dim3(B), dim3(1024), 0, st, target_idx, bag_offsets, oh_labels, L,
                     superpixels, classes, counts, G, class_base, H, W, reinterpret_cast<long*>(out), value_counts);
  JTSM_CHECK_LAUNCH("paint_sem_seg_evidence");
  return JTSM_OK;
}

namespace {
int pack_rows(const void* const* ptrs, const int* counts, int B, jtsm::ImageRows& out, int& total) {
  JTSM_REQUIRE(B >= 1 && B <= jtsm::kMaxImages, "label preparation: 1..%d images per call, got %d", jtsm::kMaxImages, B);
  total = 0;
  for (int b = 0; b < jtsm::kMaxImages; ++b) { out.ptr[b] = nullptr; out.first[b] = 0; }
  for (int b = 0; b < B; ++b) {
    JTSM_REQUIRE(counts[b] >= 0 && (counts[b] == 0 || ptrs[b]), "label preparation: image %d has %d rows at %p", b, counts[b], ptrs[b]);
    out.ptr[b] = ptrs[b];
    out.first[b] = total;
    total += counts[b];
  }
  for (int b = B; b <= jtsm::kMaxImages; ++b) out.first[b] = total;
  return JTSM_OK;
}
}  // namespace

int jtsm_pooler_rois_levels_f32(const float* const* boxes, const int* counts, int B, int min_level, int max_level,
                                float canonical_box_size, float canonical_level, float* rois, int32_t* level, void* stream) {
  jtsm::ImageRows im;
  int M;
  if (int rc = pack_rows(reinterpret_cast<const void* const*>(boxes), counts, B, im, M)) return rc;
  JTSM_REQUIRE(canonical_box_size > 0.f && min_level <= max_level, "pooler levels: bad canonical size / level range");
  for (int b = 0; b < B; ++b)
    JTSM_REQUIRE((reinterpret_cast<uintptr_t>(boxes[b]) & 15) == 0, "pooler levels: image %d's boxes are not 16-byte aligned", b);
  if (M == 0) return JTSM_OK;
  hipLaunchKernelGGL(jtsm::pooler_rois_levels_kernel, dim3(jtsm::ceil_div(M, 256)), dim3(256), 0, jtsm::as_stream(stream), im, B, M,
                     1.0f / canonical_box_size, canonical_level, min_level, max_level, rois, level);
  JTSM_CHECK_LAUNCH("pooler_rois_levels");
  return JTSM_OK;
}

int jtsm_roi_scale_f32(const int32_t* argmax, int bins, int C, const float* const* objectness, const int* counts, int B,
                       float* out, void* stream) {
  jtsm::ImageRows im;
  int M;
  if (int rc = pack_rows(reinterpret_cast<const void* const*>(objectness), counts, B, im, M)) return rc;
  JTSM_REQUIRE(bins > 0 && C > 0, "roi_scale: bins %d, channels %d", bins, C);
  if (M == 0) return JTSM_OK;
  hipLaunchKernelGGL(jtsm::roi_scale_kernel, dim3(jtsm::ceil_div(M, 256)), dim3(256), 0, jtsm::as_stream(stream), argmax, bins, C, im,
                     B, M, out);
  JTSM_CHECK_LAUNCH("roi_scale");
  return JTSM_OK;
}

size_t jtsm_image_labels_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * 256 * sizeof(int); }

int jtsm_image_labels(const int64_t* const* gt_classes, const int* counts, int B, int num_classes, const void* sem_seg,
                      int sem_elem_bytes, long pixels, int num_stuff, int stuff_offset, float* oh_things,
                      int32_t* things_cls, int32_t* things_cnt, float* oh_stuff, int32_t* stuff_cls, int32_t* stuff_cnt,
                      void* workspace, void* stream) {
  jtsm::ImageRows im;
  int total;
  if (int rc = pack_rows(reinterpret_cast<const void* const*>(gt_classes), counts, B, im, total)) return rc;
  JTSM_REQUIRE(num_classes >= 1 && num_classes <= 256, "image labels: 1..256 thing classes, got %d", num_classes);
  int* flags = nullptr;
  if (sem_seg) {
    JTSM_REQUIRE(num_stuff >= 2 && num_stuff <= 256 && pixels > 0 && workspace && (sem_elem_bytes == 8 || sem_elem_bytes == 1),
                 "image labels: stuff classes %d, %ld pixels, %d-byte labels", num_stuff, pixels, sem_elem_bytes);
    flags = reinterpret_cast<int*>(workspace);
    JTSM_CHECK_HIP(hipMemsetAsync(flags, 0, jtsm_image_labels_workspace_bytes(B), jtsm::as_stream(stream)));
    const dim3 grid((unsigned)(pixels / 256 / 16 < 1 ? 1 : (pixels / 256 / 16 > 256 ? 256 : pixels / 256 / 16)), B);
    if (sem_elem_bytes == 8)
      hipLaunchKernelGGL((jtsm::sem_label_flags_kernel<long>), grid, dim3(256), 0, jtsm::as_stream(stream),
                         reinterpret_cast<const long*>(sem_seg), pixels, flags);
    else
      hipLaunchKernelGGL((jtsm::sem_label_flags_kernel<unsigned char>), grid, dim3(256), 0, jtsm::as_stream(stream),
                         reinterpret_cast<const unsigned char*>(sem_seg), pixels, flags);
    JTSM_CHECK_LAUNCH("sem_label_flags");
  }
  hipLaunchKernelGGL(jtsm::image_class_lists_kernel, dim3(B), dim3(256), 0, jtsm::as_stream(stream), im, num_classes, flags, num_stuff,
                     stuff_offset, oh_things, things_cls, things_cnt, oh_stuff, stuff_cls, stuff_cnt);
  JTSM_CHECK_LAUNCH("image_class_lists");
  return JTSM_OK;
}

int jtsm_fg_compact(const int32_t* labels, int bg_label, const int32_t* bag_offsets, int B, int R, const float* boxes,
                    const int32_t* matched, int32_t* fg_rows, float* fg_boxes, int64_t* fg_classes, int32_t* fg_img,
                    int32_t* fg_matched, float* fg_rois, int64_t* counts, void* stream) {
  JTSM_REQUIRE(B >= 1 && B <= 64 && R >= 0, "fg_compact: 1..64 images, got %d (rows %d)", B, R);
  JTSM_REQUIRE(labels && bag_offsets && boxes && fg_rows && fg_boxes && fg_classes && fg_img && fg_rois && counts &&
               (!matched || fg_matched), "fg_compact: null pointer");
  JTSM_REQUIRE((reinterpret_cast<uintptr_t>(boxes) & 15) == 0 && (reinterpret_cast<uintptr_t>(fg_boxes) & 15) == 0,
               "fg_compact: boxes must be 16-byte aligned");
  hipLaunchKernelGGL(jtsm::fg_compact_kernel, dim3(1), dim3(1024), 0, jtsm::as_stream(stream), labels, bg_label, bag_offsets, B, R,
                     boxes, matched, fg_rows, fg_boxes, reinterpret_cast<long*>(fg_classes), fg_img, fg_matched, fg_rois,
                     reinterpret_cast<long*>(counts));
  JTSM_CHECK_LAUNCH("fg_compact");
  return JTSM_OK;
}

}  // extern "C"
