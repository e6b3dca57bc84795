// JTSM's per-proposal multiple-instance-learning losses, fused forward + analytic backward.
//
// Replaces the pure-PyTorch arithmetic of
//   TSMOutputLayers.forward / predict_probs_img / binary_cross_entropy_loss
//       (projects/WSL/wsl/modeling/roi_heads/fast_rcnn_tsm.py:573-586,840-854,346-362)
//   OICROutputs.softmax_cross_entropy_loss / box_reg_loss ("smooth_l1_weighted", beta = 0)
//       (projects/WSL/wsl/modeling/roi_heads/fast_rcnn_oicr.py:243-247,282-298,350-380)
// which the reference runs as 5+ small launches per image (MIL) and a dozen gathers/reductions
// per refinement branch.  Formulas: SURVEY.md Appendix B.
//
// Data shape: logits are (R, nc) row-major slices (leading dimension `ld`, so they may be column
// ranges of one fused predictor GEMM output); proposals of one image ("bag") are contiguous rows
// [bag_off[i], bag_off[i+1]).  Layout on the machine: ONE WAVEFRONT PER ROW, the 64 lanes holding
// classes lane, lane+64, lane+128 (nc <= 192) — a row softmax is then a wavefront shuffle
// reduction, and the per-bag column softmax (over the ragged proposal axis) is a segmented
// reduction: each workgroup owns a slab of rows of one bag, keeps per-lane running (max, sum) /
// partial sums, combines its 4 waves through LDS and writes one partial per (bag, slab); partials
// are folded in slab order, so results do not depend on scheduling (no float atomics).
#include <cfloat>

#include "common.h"

namespace jtsm {
namespace {

constexpr int SLOTS = 3;      // classes per lane: nc <= 192
constexpr int WAVES = 4;      // per workgroup
constexpr float kTiny = 1e-6f;

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct SlabRange { int bag, r0, r1; };

// Workgroup -> (bag, slab): slabs are `rows_per_slab` rows; grid.x = bags * slabs_per_bag.
__device__ __forceinline__ SlabRange slab_of(const int* __restrict__ bag_off, int slabs_per_bag,
                                             int rows_per_slab) {
  SlabRange s;
  s.bag = blockIdx.x / slabs_per_bag;
  const int slab = blockIdx.x - s.bag * slabs_per_bag;
  const int b0 = bag_off[s.bag], b1 = bag_off[s.bag + 1];
  s.r0 = b0 + slab * rows_per_slab;
  s.r1 = min(b1, s.r0 + rows_per_slab);
  return s;
}

// ---- MIL pass 1: running (max, sumexp) of D over the slab's rows, per class ----------------------
__global__ __launch_bounds__(256) void mil_colstats(const float* __restrict__ D, int ld, int nc,
                                                    const int* __restrict__ bag_off,
                                                    int slabs_per_bag, int rows_per_slab,
                                                    float2* __restrict__ part) {
  __shared__ float2 red[WAVES][64 * SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const SlabRange s = slab_of(bag_off, slabs_per_bag, rows_per_slab);
  float m[SLOTS], z[SLOTS];
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) { m[k] = -FLT_MAX; z[k] = 0.f; }
  for (int r = s.r0 + wv; r < s.r1; r += WAVES) {
    const float* row = D + (size_t)r * ld;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c < nc) {
        const float v = row[c];
        const float mn = fmaxf(m[k], v);
        z[k] = z[k] * __expf(m[k] - mn) + __expf(v - mn);
        m[k] = mn;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) red[wv][lane + 64 * k] = make_float2(m[k], z[k]);
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c >= nc) continue;
      float M = -FLT_MAX, Z = 0.f;
      for (int w = 0; w < WAVES; ++w) {
        const float2 q = red[w][c];
        const float mn = fmaxf(M, q.x);
        Z = Z * __expf(M - mn) + q.y * __expf(q.x - mn);
        M = mn;
      }
      part[(size_t)blockIdx.x * nc + c] = make_float2(M, Z);
    }
  }
}

// Fold the slab partials of one bag, in slab order (deterministic).
__device__ __forceinline__ void fold_colstats(const float2* __restrict__ part, int bag,
                                              int slabs_per_bag, int nc, int lane,
                                              float (&M)[SLOTS], float (&Z)[SLOTS]) {
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) { M[k] = -FLT_MAX; Z[k] = 0.f; }
  for (int s = 0; s < slabs_per_bag; ++s) {
    const float2* p = part + ((size_t)bag * slabs_per_bag + s) * nc;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c < nc) {
        const float2 q = p[c];
        const float mn = fmaxf(M[k], q.x);
        Z[k] = Z[k] * __expf(M[k] - mn) + q.y * __expf(q.x - mn);
        M[k] = mn;
      }
    }
  }
}

// The same fold, once per bag (thread = class), its result read by every workgroup of passes 2 and 3.  Those used to
// fold the bag's slab partials themselves: 125 dependent steps for a 2000-row bag, each behind its own load — ~100 us
// in front of 15 us of work, in every workgroup of both passes.  Here the loads of 16 slabs are requested together and
// the recurrence (same expression, same slab order) runs from registers; folding the ONE folded entry through
// fold_colstats reproduces it exactly (max(-FLT_MAX, m) = m; 0 * e + z * exp(0) = z).
__global__ __launch_bounds__(64 * SLOTS) void mil_fold(const float2* __restrict__ part, int slabs_per_bag, int nc,
                                                       float2* __restrict__ folded) {
  const int c = threadIdx.x, bag = blockIdx.x;
  if (c >= nc) return;
  const float2* p = part + (size_t)bag * slabs_per_bag * nc + c;
  float M = -FLT_MAX, Z = 0.f;
  int s = 0;
  for (; s + 16 <= slabs_per_bag; s += 16) {
    float2 q[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) q[u] = p[(size_t)(s + u) * nc];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float mn = fmaxf(M, q[u].x);
      Z = Z * __expf(M - mn) + q[u].y * __expf(q[u].x - mn);
      M = mn;
    }
  }
  for (; s < slabs_per_bag; ++s) {
    const float2 q = p[(size_t)s * nc];
    const float mn = fmaxf(M, q.x);
    Z = Z * __expf(M - mn) + q.y * __expf(q.x - mn);
    M = mn;
  }
  folded[(size_t)bag * nc + c] = make_float2(M, Z);
}

// Row softmax of C and bag-softmax of D for one row; returns a[], b[] per slot.
__device__ __forceinline__ void mil_row(const float* __restrict__ Crow, const float* __restrict__ Drow,
                                        int nc, int lane, const float (&M)[SLOTS],
                                        const float (&Z)[SLOTS], float (&a)[SLOTS], float (&b)[SLOTS]) {
  float cv[SLOTS], mx = -FLT_MAX;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int c = lane + 64 * k;
    cv[k] = c < nc ? Crow[c] : -FLT_MAX;
    mx = fmaxf(mx, cv[k]);
  }
  mx = wave_max(mx);
  float sm = 0.f;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int c = lane + 64 * k;
    a[k] = c < nc ? __expf(cv[k] - mx) : 0.f;
    sm += a[k];
  }
  sm = wave_sum(sm);
  const float inv = 1.f / sm;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int c = lane + 64 * k;
    a[k] *= inv;
    b[k] = c < nc ? __expf(Drow[c] - M[k]) / Z[k] : 0.f;
  }
}

// ---- MIL pass 2: scores = softmax_c(C) * softmax_bag(D); partial sums over the slab ---------------
__global__ __launch_bounds__(256) void mil_scores(const float* __restrict__ C, const float* __restrict__ D,
                                                  int ld, int nc, const int* __restrict__ bag_off,
                                                  int slabs_per_bag, int rows_per_slab,
                                                  const float2* __restrict__ part,
                                                  float* __restrict__ scores, float* __restrict__ psum_part) {
  __shared__ float red[WAVES][64 * SLOTS];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const SlabRange s = slab_of(bag_off, slabs_per_bag, rows_per_slab);
  float M[SLOTS], Z[SLOTS], acc[SLOTS];
  fold_colstats(part, s.bag, 1, nc, lane, M, Z);   // part: the per-bag result of mil_fold
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) acc[k] = 0.f;
  for (int r = s.r0 + wv; r < s.r1; r += WAVES) {
    float a[SLOTS], b[SLOTS];
    mil_row(C + (size_t)r * ld, D + (size_t)r * ld, nc, lane, M, Z, a, b);
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c < nc) {
        const float sc = a[k] * b[k];
        scores[(size_t)r * nc + c] = sc;
        acc[k] += sc;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) red[wv][lane + 64 * k] = acc[k];
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c < nc) psum_part[(size_t)blockIdx.x * nc + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
    }
  }
}

// ---- MIL pass 3 (one workgroup): image probabilities, BCE loss, dL/dp ----------------------------
// psum[i,c] = sum of scores over the bag (unclamped, kept for the backward);
// p = clamp(psum, 1e-6, 1-1e-6); loss = BCE(p, y) mean over B*nc (or sum / B);
// gp[i,c] = dL/dpsum (0 where the clamp saturates, like torch.clamp's backward).
__global__ __launch_bounds__(256) void mil_finish(const float* __restrict__ psum_part, int nbags,
                                                  int slabs_per_bag, int nc,
                                                  const float* __restrict__ labels, int mean_loss,
                                                  float* __restrict__ psum, float* __restrict__ p_img,
                                                  float* __restrict__ gp, float* __restrict__ loss) {
  __shared__ float red[256];
  float part = 0.f;
  const float norm = mean_loss ? 1.f / ((float)nbags * nc) : 1.f / (float)nbags;
  for (int idx = threadIdx.x; idx < nbags * nc; idx += 256) {
    const int bag = idx / nc, c = idx - bag * nc;
    // the slab partials in slab order, 16 loads in flight (one load per dependent add took 44 us for 125 slabs)
    const float* pp = psum_part + (size_t)bag * slabs_per_bag * nc + c;
    float sum = 0.f;
    int s = 0;
    for (; s + 16 <= slabs_per_bag; s += 16) {
      float q[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) q[u] = pp[(size_t)(s + u) * nc];
#pragma unroll
      for (int u = 0; u < 16; ++u) sum += q[u];
    }
    for (; s < slabs_per_bag; ++s) sum += pp[(size_t)s * nc];
    const float p = fminf(fmaxf(sum, kTiny), 1.f - kTiny);
    const float y = labels[idx];
    // F.binary_cross_entropy clamps each log term at -100
    const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.f - p), -100.f);
    part += -(y * lp + (1.f - y) * lq);
    psum[idx] = sum;
    p_img[idx] = p;
    const bool pass = sum >= kTiny && sum <= 1.f - kTiny;
    gp[idx] = pass ? (p - y) / (p * (1.f - p)) * norm : 0.f;
  }
  red[threadIdx.x] = part;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss = red[0] * norm;
}

// ---- MIL backward: dC, dD (Appendix B) ------------------------------------------------------------
//   dC[r,c] = up * ( g_c s[r,c] - a[r,c] * sum_c' g_c' s[r,c'] )
//   dD[r,c] = up * g_c * ( s[r,c] - b[r,c] * psum[i,c] )
__global__ __launch_bounds__(256) void mil_backward(const float* __restrict__ C, const float* __restrict__ D,
                                                    int ld, int nc, const int* __restrict__ bag_off,
                                                    int slabs_per_bag, int rows_per_slab,
                                                    const float2* __restrict__ part,
                                                    const float* __restrict__ psum, const float* __restrict__ gp,
                                                    const float* __restrict__ upstream,
                                                    float* __restrict__ dC, float* __restrict__ dD, int ldg) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const SlabRange s = slab_of(bag_off, slabs_per_bag, rows_per_slab);
  float M[SLOTS], Z[SLOTS], g[SLOTS], ps[SLOTS];
  fold_colstats(part, s.bag, 1, nc, lane, M, Z);   // part: the per-bag result of mil_fold
  const float up = upstream ? *upstream : 1.f;
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int c = lane + 64 * k;
    g[k] = c < nc ? gp[(size_t)s.bag * nc + c] * up : 0.f;
    ps[k] = c < nc ? psum[(size_t)s.bag * nc + c] : 0.f;
  }
  for (int r = s.r0 + wv; r < s.r1; r += WAVES) {
    float a[SLOTS], b[SLOTS];
    mil_row(C + (size_t)r * ld, D + (size_t)r * ld, nc, lane, M, Z, a, b);
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) dot += g[k] * a[k] * b[k];
    dot = wave_sum(dot);
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c < nc) {
        const float sc = a[k] * b[k];
        dC[(size_t)r * ldg + c] = g[k] * sc - a[k] * dot;
        dD[(size_t)r * ldg + c] = g[k] * (sc - b[k] * ps[k]);
      }
    }
  }
}

// ---- OICR refinement losses ------------------------------------------------------------------------
// acc[0] = sum_r w_r CE_r, acc[1] = #{w_r > 1e-12}, acc[2] = sum_fg w_r |d - t|_1   (per workgroup
// partials, folded in order by oicr_finish).
__device__ __forceinline__ void box_target(const float* __restrict__ pb, const float* __restrict__ gb,
                                           float (&t)[4]) {
  const float sw = pb[2] - pb[0], sh = pb[3] - pb[1];
  const float sx = pb[0] + 0.5f * sw, sy = pb[1] + 0.5f * sh;
  const float tw = gb[2] - gb[0], th = gb[3] - gb[1];
  const float tx = gb[0] + 0.5f * tw, ty = gb[1] + 0.5f * th;
  t[0] = 10.f * (tx - sx) / sw;
  t[1] = 10.f * (ty - sy) / sh;
  t[2] = 5.f * logf(tw / sw);
  t[3] = 5.f * logf(th / sh);
}

__global__ __launch_bounds__(256) void oicr_forward(const float* __restrict__ Zl, int ldz, int ncls,
                                                    const float* __restrict__ Dl, int ldd,
                                                    const int* __restrict__ labels,
                                                    const float* __restrict__ weights,
                                                    const float* __restrict__ prop, const float* __restrict__ gt,
                                                    int R, float* __restrict__ partials) {
  __shared__ float red[WAVES][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float ce = 0.f, nv = 0.f, l1 = 0.f;
  for (int r = blockIdx.x * WAVES + wv; r < R; r += gridDim.x * WAVES) {
    const int y = labels[r];
    const float w = y == -1 ? 0.f : weights[r];
    if (w > 1e-12f) nv += 1.f;
    const float* row = Zl + (size_t)r * ldz;
    float v[SLOTS], mx = -FLT_MAX;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      v[k] = c < ncls ? row[c] : -FLT_MAX;
      mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) sm += (lane + 64 * k) < ncls ? __expf(v[k] - mx) : 0.f;
    sm = wave_sum(sm);
    if (y >= 0) ce += w * (logf(sm) + mx - row[y]);            // same value in every lane
    if (y >= 0 && y < ncls - 1 && Dl) {
      float t[4];
      box_target(prop + 4 * (size_t)r, gt + 4 * (size_t)r, t);
      const float* dr = Dl + (size_t)r * ldd + 4 * y;
      l1 += w * (fabsf(dr[0] - t[0]) + fabsf(dr[1] - t[1]) + fabsf(dr[2] - t[2]) + fabsf(dr[3] - t[3]));
    }
  }
  if (lane == 0) { red[wv][0] = ce; red[wv][1] = nv; red[wv][2] = l1; }
  __syncthreads();
  if (threadIdx.x < 3)
    partials[blockIdx.x * 3 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = loss_cls = sum w CE / V, out[1] = loss_box = sum w L1 / R, out[2] = V
__global__ void oicr_finish(const float* __restrict__ partials, int nblocks, int R, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float ce = 0.f, nv = 0.f, l1 = 0.f;
    int b = 0;
    for (; b + 16 <= nblocks; b += 16) {   // block order kept; 48 loads in flight instead of one per add
      float q[48];
#pragma unroll
      for (int i = 0; i < 48; ++i) q[i] = partials[3 * b + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) { ce += q[3 * u]; nv += q[3 * u + 1]; l1 += q[3 * u + 2]; }
    }
    for (; b < nblocks; ++b) { ce += partials[3 * b]; nv += partials[3 * b + 1]; l1 += partials[3 * b + 2]; }
    out[0] = ce / nv;   // V == 0 gives 0/0 like the reference (no guard)
    out[1] = l1 / (float)R;
    out[2] = nv;
  }
}

// dZ[r,c] = up_cls * w_r / V * (softmax(z_r)[c] - [c == y_r]);
// dDl[r, 4y+j] = up_box * w_r / R * sign(d - t), every other column of the row 0.
__global__ __launch_bounds__(256) void oicr_backward(const float* __restrict__ Zl, int ldz, int ncls,
                                                     const float* __restrict__ Dl, int ldd, int nbox,
                                                     const int* __restrict__ labels,
                                                     const float* __restrict__ weights,
                                                     const float* __restrict__ prop, const float* __restrict__ gt,
                                                     int R, const float* __restrict__ fin,
                                                     const float* __restrict__ up_cls, const float* __restrict__ up_box,
                                                     float* __restrict__ dZ, int ldgz, float* __restrict__ dDl, int ldgd) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float V = fin[2];
  const float uc = (up_cls ? *up_cls : 1.f) / V, ub = (up_box ? *up_box : 1.f) / (float)R;
  for (int r = blockIdx.x * WAVES + wv; r < R; r += gridDim.x * WAVES) {
    const int y = labels[r];
    const float w = y == -1 ? 0.f : weights[r];
    const float* row = Zl + (size_t)r * ldz;
    float v[SLOTS], mx = -FLT_MAX;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      v[k] = c < ncls ? row[c] : -FLT_MAX;
      mx = fmaxf(mx, v[k]);
    }
    mx = wave_max(mx);
    float sm = 0.f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) { v[k] = (lane + 64 * k) < ncls ? __expf(v[k] - mx) : 0.f; sm += v[k]; }
    sm = wave_sum(sm);
    const float inv = 1.f / sm;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
      const int c = lane + 64 * k;
      if (c < ncls) dZ[(size_t)r * ldgz + c] = y < 0 ? 0.f : uc * w * (v[k] * inv - (c == y ? 1.f : 0.f));
    }
    if (dDl) {
      float* drow = dDl + (size_t)r * ldgd;
      for (int c = lane; c < nbox; c += 64) drow[c] = 0.f;
      if (y >= 0 && y < ncls - 1 && lane < 4) {
        float t[4];
        box_target(prop + 4 * (size_t)r, gt + 4 * (size_t)r, t);
        const float diff = Dl[(size_t)r * ldd + 4 * y + lane] - t[lane];
        drow[4 * y + lane] = ub * w * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
      }
    }
  }
}

inline void mil_plan(int max_bag_rows, int nbags, int* slabs_per_bag, int* rows_per_slab) {
  // ~512 workgroups in total, at least 16 rows (4 per wave) each
  int want = 512 / (nbags > 0 ? nbags : 1);
  if (want < 1) want = 1;
  int rps = (max_bag_rows + want - 1) / want;
  if (rps < 16) rps = 16;
  *rows_per_slab = rps;
  *slabs_per_bag = (max_bag_rows + rps - 1) / rps;
  if (*slabs_per_bag < 1) *slabs_per_bag = 1;
}

struct MilWs { float2* part; float* psum_part; float* psum; float* gp; float2* folded; };
inline size_t align16(size_t b) { return (b + 15) & ~(size_t)15; }
inline MilWs mil_carve(void* ws, int nbags, int spb, int nc) {
  char* p = (char*)ws;
  MilWs k;
  k.part = (float2*)p; p += align16((size_t)nbags * spb * nc * sizeof(float2));
  k.psum_part = (float*)p; p += align16((size_t)nbags * spb * nc * sizeof(float));
  k.psum = (float*)p; p += align16((size_t)nbags * nc * sizeof(float));
  k.gp = (float*)p; p += align16((size_t)nbags * nc * sizeof(float));
  k.folded = (float2*)p;
  return k;
}

// ---- mask loss (mask_rcnn_loss, projects/WSL/wsl/modeling/roi_heads/mask_head.py:23-103 /
// detectron2/modeling/roi_heads/mask_head.py:31-112): mean BCE-with-logits between the ground-truth class
// channel of the (N,M,M,C) logits and a 0/1 target.  Thread per (roi, pixel); fixed-order two-stage sum.
__global__ __launch_bounds__(256) void mask_bce_fwd(const float* __restrict__ z, int ld, const long* __restrict__ cls,
                                                    const unsigned char* __restrict__ target, long npix, int pix_per_roi,
                                                    float* __restrict__ part) {
  __shared__ float red[256];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long n = i / pix_per_roi;
    const float x = z[i * ld + (cls ? (int)cls[n] : 0)];
    const float t = target[i] ? 1.f : 0.f;
    s += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void mask_bce_finish(const float* __restrict__ part, int nblocks, long npix,
                                                       float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) s += part[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)(red[0] / (double)npix);
}

// dz[i, c] = (sigmoid(z) - t) * upstream / npix on the class channel, zero elsewhere (dense, as the predictor's
// backward wants it)
__global__ __launch_bounds__(256) void mask_bce_bwd(const float* __restrict__ z, int ld, const long* __restrict__ cls,
                                                    const unsigned char* __restrict__ target, long npix, int pix_per_roi,
                                                    const float* __restrict__ upstream, float* __restrict__ dz) {
  const float k = (upstream ? *upstream : 1.f) / (float)npix;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long n = i / pix_per_roi;
    const int c = cls ? (int)cls[n] : 0;
    const float x = z[i * ld + c];
    const float g = (1.f / (1.f + expf(-x)) - (target[i] ? 1.f : 0.f)) * k;
    float* row = dz + i * ld;
    for (int j = 0; j < ld; ++j) row[j] = j == c ? g : 0.f;
  }
}

// The same result written 16 bytes per lane: blockIdx.y = roi (its class channel is block-uniform), a thread owns one
// float4 piece of one pixel's row of `ld` logits.  The thread-per-pixel form above walks a 4 * ld-byte row per thread —
// stores 320 bytes apart across a wavefront: 57 us for 74 MB on the 294-roi mask heads (1.3 TB/s).
__global__ __launch_bounds__(256) void mask_bce_bwd_wide(const float* __restrict__ z, int ld, const long* __restrict__ cls,
                                                         const unsigned char* __restrict__ target, long npix,
                                                         int pix_per_roi, const float* __restrict__ upstream,
                                                         float* __restrict__ dz) {
  const float k = (upstream ? *upstream : 1.f) / (float)npix;
  const int n = blockIdx.y;
  const int c = cls ? (int)cls[n] : 0;
  const unsigned q = (unsigned)ld >> 2, pieces = (unsigned)pix_per_roi * q;
  const long base = (long)n * pix_per_roi;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < pieces; idx += gridDim.x * 256u) {
    const unsigned px = idx / q, j = (idx - px * q) * 4u;
    const long i = base + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)c - j < 4u) {
      const float x = z[i * ld + c];
      const float g = (1.f / (1.f + expf(-x)) - (target[i] ? 1.f : 0.f)) * k;
      const unsigned e = (unsigned)c - j;
      v.x = e == 0u ? g : 0.f; v.y = e == 1u ? g : 0.f; v.z = e == 2u ? g : 0.f; v.w = e == 3u ? g : 0.f;
    }
    *reinterpret_cast<float4*>(dz + i * ld + j) = v;
  }
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

size_t jtsm_mil_workspace_bytes(int nbags, int max_bag_rows, int nc) {
  if (nbags <= 0 || nc <= 0 || max_bag_rows < 0) return 16;
  int spb, rps;
  mil_plan(max_bag_rows, nbags, &spb, &rps);
  return align16((size_t)nbags * spb * nc * sizeof(float2)) + align16((size_t)nbags * spb * nc * sizeof(float)) +
         2 * align16((size_t)nbags * nc * sizeof(float)) + align16((size_t)nbags * nc * sizeof(float2)) + 16;
}

int jtsm_mil_forward_f32(const float* cls_logits, const float* det_logits, int ld, int nc,
                         const int32_t* bag_offsets, int nbags, int max_bag_rows, const float* labels,
                         int mean_loss, float* scores, float* img_probs, float* loss, void* workspace,
                         void* stream) {
  JTSM_REQUIRE(nc > 0 && nc <= 64 * SLOTS, "mil: nc must be in (0, %d], got %d", 64 * SLOTS, nc);
  JTSM_REQUIRE(nbags > 0 && max_bag_rows >= 0 && ld >= nc, "mil: bad sizes");
  JTSM_REQUIRE(cls_logits && det_logits && bag_offsets && labels && scores && img_probs && loss && workspace,
               "mil: null pointer");
  int spb, rps;
  mil_plan(max_bag_rows, nbags, &spb, &rps);
  const MilWs k = mil_carve(workspace, nbags, spb, nc);
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(mil_colstats, dim3(nbags * spb), dim3(256), 0, st, det_logits, ld, nc, bag_offsets, spb,
                     rps, k.part);
  hipLaunchKernelGGL(mil_fold, dim3(nbags), dim3(64 * SLOTS), 0, st, k.part, spb, nc, k.folded);
  hipLaunchKernelGGL(mil_scores, dim3(nbags * spb), dim3(256), 0, st, cls_logits, det_logits, ld, nc,
                     bag_offsets, spb, rps, k.folded, scores, k.psum_part);
  hipLaunchKernelGGL(mil_finish, dim3(1), dim3(256), 0, st, k.psum_part, nbags, spb, nc, labels, mean_loss,
                     k.psum, img_probs, k.gp, loss);
  JTSM_CHECK_LAUNCH("mil forward");
  return JTSM_OK;
}

int jtsm_mil_backward_f32(const float* cls_logits, const float* det_logits, int ld, int nc,
                          const int32_t* bag_offsets, int nbags, int max_bag_rows, const float* upstream,
                          float* d_cls, float* d_det, int ld_grad, const void* workspace, void* stream) {
  JTSM_REQUIRE(nc > 0 && nc <= 64 * SLOTS && nbags > 0 && ld >= nc && ld_grad >= nc, "mil backward: bad sizes");
  JTSM_REQUIRE(cls_logits && det_logits && bag_offsets && d_cls && d_det && workspace, "mil backward: null pointer");
  int spb, rps;
  mil_plan(max_bag_rows, nbags, &spb, &rps);
  const MilWs k = mil_carve(const_cast<void*>(workspace), nbags, spb, nc);
  hipLaunchKernelGGL(mil_backward, dim3(nbags * spb), dim3(256), 0, as_stream(stream), cls_logits, det_logits,
                     ld, nc, bag_offsets, spb, rps, k.folded, k.psum, k.gp, upstream, d_cls, d_det, ld_grad);
  JTSM_CHECK_LAUNCH("mil backward");
  return JTSM_OK;
}

#define OICR_BLOCKS 256

size_t jtsm_oicr_workspace_bytes(void) { return (OICR_BLOCKS * 3 + 4) * sizeof(float); }

int jtsm_oicr_forward_f32(const float* cls_logits, int ld_cls, int num_cls, const float* box_deltas, int ld_box,
                          const int32_t* labels, const float* weights, const float* proposals,
                          const float* gt_boxes, int R, float* losses, void* workspace, void* stream) {
  JTSM_REQUIRE(num_cls > 1 && num_cls <= 64 * SLOTS && R >= 0 && ld_cls >= num_cls, "oicr: bad sizes");
  JTSM_REQUIRE(cls_logits && labels && weights && losses && workspace, "oicr: null pointer");
  JTSM_REQUIRE(!box_deltas || (proposals && gt_boxes && ld_box >= 4 * (num_cls - 1)), "oicr: box branch needs boxes");
  float* part = (float*)workspace;
  hipStream_t st = as_stream(stream);
  int blocks = (R + WAVES - 1) / WAVES;
  if (blocks > OICR_BLOCKS) blocks = OICR_BLOCKS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(oicr_forward, dim3(blocks), dim3(256), 0, st, cls_logits, ld_cls, num_cls, box_deltas,
                     ld_box, labels, weights, proposals, gt_boxes, R, part);
  hipLaunchKernelGGL(oicr_finish, dim3(1), dim3(64), 0, st, part, blocks, R, losses);
  JTSM_CHECK_LAUNCH("oicr forward");
  return JTSM_OK;
}

int jtsm_oicr_backward_f32(const float* cls_logits, int ld_cls, int num_cls, const float* box_deltas, int ld_box,
                           const int32_t* labels, const float* weights, const float* proposals,
                           const float* gt_boxes, int R, const float* losses, const float* up_cls,
                           const float* up_box, float* d_cls, int ld_dcls, float* d_box, int ld_dbox,
                           void* stream) {
  JTSM_REQUIRE(num_cls > 1 && num_cls <= 64 * SLOTS && R >= 0, "oicr backward: bad sizes");
  if (R == 0) return JTSM_OK;
  JTSM_REQUIRE(cls_logits && labels && weights && losses && d_cls, "oicr backward: null pointer");
  JTSM_REQUIRE(!d_box || (box_deltas && proposals && gt_boxes), "oicr backward: box branch needs boxes");
  int blocks = (R + WAVES - 1) / WAVES;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(oicr_backward, dim3(blocks), dim3(256), 0, as_stream(stream), cls_logits, ld_cls, num_cls,
                     box_deltas, ld_box, 4 * (num_cls - 1), labels, weights, proposals, gt_boxes, R, losses,
                     up_cls, up_box, d_cls, ld_dcls, d_box, ld_dbox);
  JTSM_CHECK_LAUNCH("oicr backward");
  return JTSM_OK;
}

#define MASK_BCE_BLOCKS 256

size_t jtsm_mask_bce_workspace_bytes(void) { return MASK_BCE_BLOCKS * sizeof(float); }

int jtsm_mask_bce_forward_f32(const float* logits, int ld, int num_classes, const int64_t* gt_classes,
                              const uint8_t* target, int N, int side, float* out, void* workspace, void* stream) {
  JTSM_REQUIRE(N >= 0 && side > 0 && num_classes > 0 && ld >= num_classes, "mask_bce: bad sizes");
  JTSM_REQUIRE(out, "mask_bce: null out");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && target && workspace && (gt_classes || num_classes == 1), "mask_bce: null pointer");
  hipStream_t st = as_stream(stream);
  const long npix = (long)N * side * side;
  const int blocks = (int)((npix + 255) / 256 < MASK_BCE_BLOCKS ? (npix + 255) / 256 : MASK_BCE_BLOCKS);
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(mask_bce_fwd, dim3(blocks), dim3(256), 0, st, logits, ld, (const long*)gt_classes, target, npix,
                     side * side, part);
  hipLaunchKernelGGL(mask_bce_finish, dim3(1), dim3(256), 0, st, part, blocks, npix, out);
  JTSM_CHECK_LAUNCH("mask_bce forward");
  return JTSM_OK;
}

int jtsm_mask_bce_backward_f32(const float* logits, int ld, int num_classes, const int64_t* gt_classes,
                               const uint8_t* target, int N, int side, const float* upstream, float* dlogits,
                               void* stream) {
  JTSM_REQUIRE(N >= 0 && side > 0 && num_classes > 0 && ld >= num_classes, "mask_bce backward: bad sizes");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && target && dlogits && (gt_classes || num_classes == 1), "mask_bce backward: null pointer");
  const long npix = (long)N * side * side;
  const long pieces = (long)side * side * (ld / 4);
  if (ld % 4 == 0 && (reinterpret_cast<uintptr_t>(dlogits) & 15) == 0 && N <= 65535 && pieces < (1L << 30)) {
    const int bx = (int)((pieces + 255) / 256 < 64 ? (pieces + 255) / 256 : 64);
    hipLaunchKernelGGL(mask_bce_bwd_wide, dim3(bx, N), dim3(256), 0, as_stream(stream), logits, ld,
                       (const long*)gt_classes, target, npix, side * side, upstream, dlogits);
    JTSM_CHECK_LAUNCH("mask_bce backward");
    return JTSM_OK;
  }
  const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  hipLaunchKernelGGL(mask_bce_bwd, dim3(blocks), dim3(256), 0, as_stream(stream), logits, ld, (const long*)gt_classes,
                     target, npix, side * side, upstream, dlogits);
  JTSM_CHECK_LAUNCH("mask_bce backward");
  return JTSM_OK;
}

}  // extern "C"
