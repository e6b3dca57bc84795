// MOIPool for gfx950 — JTSM's masked ROI max-pool (box pooler of JTSMROIHeads).
//
// Replaces MOIPool_forward/backward (projects/WSL/wsl/layers/csrc/MOIPool/MOIPool.h:7-47),
// whose only implementation is CUDA: MoIForward + MoIPoolForward2 + RoIPoolBackward
// (MOIPool_cuda.cu:138-338).  Behaviour contract: SURVEY.md Appendix A.3.
//
// The reference materialises mois[M,H,W] (int32; 524 MB at M=8000, 128x128) by letting every
// (roi, cell) thread walk the stride x stride image block under the cell, chasing
// superpixels[..] -> oh_labels[n, id] one pixel at a time, and then re-reads that mask once
// per channel.  Here the test "does some superpixel under cell (h,w) carry label 1 for roi
// n" is factored into two bit sets over the L superpixel ids:
//     cell_bits[b,h,w]  : ids occurring under the cell      (built once per image, all rois share it)
//     roi_bits[n]       : ids whose oh_labels[n,id] == 1
//     mois[n,h,w] = inside_box(n,h,w) && any(cell_bits[b,h,w] & roi_bits[n])
// which is exact (same set semantics), needs no M*H*W temporary and turns the per-pixel
// pointer chase into one coalesced row of words per cell.
//
// Pool kernel (NHWC): one wavefront per (roi, bin); lanes hold the roi's label words, test the
// bin's cells one by one (coalesced word row + wave ballot), and for every surviving cell
// gather their own channels (VEC floats per lane, 1 KiB per 256-channel row).  Scan order is
// h outer / w inner with a strict '>' so ties and argmax are bit-identical to the reference.
#include <algorithm>
#include <cfloat>

#include "common.h"

namespace jtsm {
namespace {

#pragma clang fp contract(off)

struct IBox { int b, x0, y0, x1, y1; };

// CUDA/HIP round(): half away from zero (roundf), MOIPool_cuda.cu:159-162.
__device__ __forceinline__ IBox round_box(const float* __restrict__ roi, float scale) {
#pragma clang fp contract(off)
  IBox r;
  r.b = (int)roi[0];
  r.x0 = (int)roundf(roi[1] * scale);
  r.y0 = (int)roundf(roi[2] * scale);
  r.x1 = (int)roundf(roi[3] * scale);
  r.y1 = (int)roundf(roi[4] * scale);
  return r;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// cell_bits[(b*H+h)*W+w][words]: a wavefront sweeps the image block under a cell (MOIPool_cuda.cu:175-186 bounds),
// OR-ing id bits into an LDS row, then stores the row.  `cpw` cells share a wavefront (64 / cpw lanes each): at stride 4
// a cell covers 16 pixels, and one wavefront per cell left 48 lanes idle in 131 000 wavefronts (46 us of the forward).
__device__ __forceinline__ void moi_cell_bits_block(const int* __restrict__ superpixels, unsigned* __restrict__ cell_bits,
                                                    int B, int H, int W, int Hs, int Ws, int L, int words, int cpw,
                                                    long block, unsigned* __restrict__ smem) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lpc = 64 / cpw, sub = lane / lpc, pl = lane - sub * lpc;      // lanes per cell, this lane's cell and slot
  const long cell0 = (block * 4 + wv) * cpw, ncell = (long)B * H * W;
  const long cell = cell0 + sub;
  if (cpw == 1 && words <= 64) {
    // Coarse levels (a wavefront per cell, hundreds of pixels under it, a handful of distinct ids): the row lives in
    // REGISTERS, lane = word.  64 pixels at a time; every distinct id among them is picked once (first live lane's id,
    // the lanes that hold the same id drop out by a ballot) and the lane that owns its word sets the bit — no LDS row,
    // no atomics (64 lanes OR-ing the same word serialise in the LDS atomic unit), no barrier.
    if (cell >= ncell) return;          // (wave-uniform)
    const int w = (int)(cell % W), h = (int)((cell / W) % H), b = (int)(cell / W / H);
    const float s = (float)(1.0 * H / Hs);
    const int hs = clampi((int)floorf((float)h / s), 0, Hs), he = clampi((int)ceilf((float)(h + 1) / s), 0, Hs);
    const int ws = clampi((int)floorf((float)w / s), 0, Ws), we = clampi((int)ceilf((float)(w + 1) / s), 0, Ws);
    const int bw = we - ws, npix = (he - hs) * bw;
    const int* __restrict__ spp = superpixels + (size_t)b * Hs * Ws;
    unsigned mine = 0u;
    for (int p0 = 0; p0 < npix; p0 += 256) {
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {       // four batches' loads in flight
        const int p = p0 + u * 64 + lane;
        const int q = min(p, npix - 1);
        const int v = spp[(size_t)(hs + q / bw) * Ws + ws + q % bw];
        id[u] = (p < npix && v >= 0 && v < L) ? v : -1;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        unsigned long long todo = __ballot(id[u] >= 0);
        while (todo) {
          const int v = __builtin_amdgcn_readlane(id[u], (int)__builtin_ctzll(todo));
          todo &= ~__ballot(id[u] == v);
          if (lane == (v >> 5)) mine |= 1u << (v & 31);
        }
      }
    }
    if (lane < words) cell_bits[(size_t)cell * words + lane] = mine;
    return;
  }
  unsigned* rows = smem + wv * cpw * words;
  for (int i = lane; i < cpw * words; i += 64) rows[i] = 0u;
  __syncthreads();
  if (cell < ncell) {
    unsigned* row = rows + sub * words;
    const int w = (int)(cell % W), h = (int)((cell / W) % H), b = (int)(cell / W / H);
    const float s = (float)(1.0 * H / Hs);
    int hs = (int)floorf((float)h / s), ws = (int)floorf((float)w / s);
    int he = (int)ceilf((float)(h + 1) / s), we = (int)ceilf((float)(w + 1) / s);
    hs = clampi(hs, 0, Hs); he = clampi(he, 0, Hs);
    ws = clampi(ws, 0, Ws); we = clampi(we, 0, Ws);
    const int bw = we - ws, npix = (he - hs) * bw;
    const int* __restrict__ spp = superpixels + (size_t)b * Hs * Ws;
    // (eight ids requested together, then their bits: left as one load per trip the coarse levels — 256 and 1024 pixels per
    // cell — walked 4 and 16 dependent round trips per wavefront and were the launch's long pole)
    for (int p0 = pl; p0 < npix; p0 += 8 * lpc) {
      int id[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int p = min(p0 + u * lpc, npix - 1);          // (a repeated pixel sets the same bit again)
        id[u] = spp[(size_t)(hs + p / bw) * Ws + ws + p % bw];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (id[u] >= 0 && id[u] < L) atomicOr(&row[id[u] >> 5], 1u << (id[u] & 31));
    }
  }
  __syncthreads();
  const long live = min((long)cpw, ncell - cell0) * words;   // this wavefront's cells are consecutive rows of the table
  for (long i = lane; i < live; i += 64) cell_bits[cell0 * words + i] = rows[i];
}

// Fine levels whose cells cover a few pixels (the superpixel map an exact multiple of the level: every BASELINE shape; at
// most 8 x 8 pixels per cell): ONE THREAD PER CELL.  The thread reads its ph x pw ids (neighbouring lanes read neighbouring
// 16-byte pieces of the same rows) and builds the row word by word in registers — no LDS row, no atomics, no barrier.
// (The wavefront form above spent its time in two barriers and a dependent load per 16-cell workgroup: 8192 workgroups
// for the stride-4 level.  A first attempt with the ids traded by shuffles inside a cell's lanes measured SLOWER than it:
// 68 against 59 us for the bench's four levels — 32 bpermutes per word pair.)
template <int KP>     // KP = pixels per cell side (a power of two: 1, 2, 4 or 8)
__device__ __forceinline__ void moi_cell_bits_small(const int* __restrict__ superpixels, unsigned* __restrict__ cell_bits,
                                                    int B, int H, int W, int Hs, int Ws, int L, int words, long block) {
  const long cell = block * 256 + threadIdx.x, ncell = (long)B * H * W;
  if (cell >= ncell) return;
  const int w = (int)(cell % W), h = (int)((cell / W) % H), b = (int)(cell / W / H);
  const int* __restrict__ src = superpixels + ((size_t)b * Hs + (size_t)h * KP) * Ws + (size_t)w * KP;
  int id[KP * KP];
  if (KP % 4 == 0 && (Ws & 3) == 0) {          // 16-byte pieces of the rows
#pragma unroll
    for (int r = 0; r < KP; ++r)
#pragma unroll
      for (int q = 0; q < KP / 4; ++q) {
        const int4 t = *reinterpret_cast<const int4*>(src + (size_t)r * Ws + 4 * q);
        id[KP * r + 4 * q] = t.x; id[KP * r + 4 * q + 1] = t.y; id[KP * r + 4 * q + 2] = t.z; id[KP * r + 4 * q + 3] = t.w;
      }
  } else {
#pragma unroll
    for (int p = 0; p < KP * KP; ++p) id[p] = src[(size_t)(p / KP) * Ws + p % KP];
  }
#pragma unroll
  for (int p = 0; p < KP * KP; ++p) if (id[p] >= L) id[p] = -1;
  unsigned* __restrict__ dst = cell_bits + (size_t)cell * words;
  for (int w0 = 0; w0 < words; w0 += 4) {
    unsigned v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < KP * KP; ++p) {
      const int wd = id[p] >> 5;                      // (-1 >> 5 == -1: never a word index)
      const unsigned bit = 1u << (id[p] & 31);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] |= wd == w0 + k ? bit : 0u;
    }
    if (w0 + 4 <= words && (words & 3) == 0) {
      *reinterpret_cast<uint4*>(dst + w0) = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (w0 + k < words) dst[w0 + k] = v[k];
    }
  }
}
__device__ __forceinline__ void moi_cell_bits_small_any(const int* __restrict__ superpixels, unsigned* __restrict__ cell_bits,
                                                        int B, int H, int W, int Hs, int Ws, int L, int words, long block) {
  switch (Hs / H) {
    case 1: return moi_cell_bits_small<1>(superpixels, cell_bits, B, H, W, Hs, Ws, L, words, block);
    case 2: return moi_cell_bits_small<2>(superpixels, cell_bits, B, H, W, Hs, Ws, L, words, block);
    case 4: return moi_cell_bits_small<4>(superpixels, cell_bits, B, H, W, Hs, Ws, L, words, block);
    default: return moi_cell_bits_small<8>(superpixels, cell_bits, B, H, W, Hs, Ws, L, words, block);
  }
}
// (the reference derives BOTH axes' pixel ranges from s = (float)H / Hs, floor(h / s) .. ceil((h + 1) / s),
// MOIPool_cuda.cu:175-186: those are exactly [k h, k h + k) only for a power-of-two ratio k shared by both axes)
static bool moi_bits_small_ok(int H, int W, int Hs, int Ws) {
  if (H <= 0 || W <= 0 || Hs % H || Ws % W) return false;
  const int k = Hs / H;
  return k == Ws / W && (k & (k - 1)) == 0 && k <= 8;
}

__global__ __launch_bounds__(256) void moi_cell_bits_kernel(const int* __restrict__ superpixels,
                                                            unsigned* __restrict__ cell_bits,
                                                            int B, int H, int W, int Hs, int Ws,
                                                            int L, int words, int cpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  if (cpw == 0) moi_cell_bits_small_any(superpixels, cell_bits, B, H, W, Hs, Ws, L, words, (long)blockIdx.x);
  else moi_cell_bits_block(superpixels, cell_bits, B, H, W, Hs, Ws, L, words, cpw, (long)blockIdx.x, smem);
}

// cells that share a wavefront in moi_cell_bits_kernel: 64 / (pixels under a cell, rounded up to a power of two)
static int moi_cells_per_wave(int H, int W, int Hs, int Ws) {
  const long npix = (long)ceil_div(Hs, H > 0 ? H : 1) * ceil_div(Ws, W > 0 ? W : 1);
  int cpw = 1;
  while (cpw < 16 && (long)(64 / (cpw * 2)) >= npix) cpw *= 2;
  return cpw;
}

// roi_bits[n][words]: bit id set iff oh_labels[n,id] == 1 (exactly 1, MOIPool_cuda.cu:198).  A wavefront takes whole
// rows: 64 consecutive labels per load (coalesced), one ballot = two words.  (A thread per word read its 32 labels one
// after the other, the lanes of a load 128 bytes apart: 64 lines touched per wave-instruction.)
constexpr int kRoiBitsRows = 8;      // rows per workgroup (two per wavefront)
__device__ __forceinline__ void moi_roi_bits_block(const int* __restrict__ oh_labels, unsigned* __restrict__ roi_bits,
                                                   long M, int L, int words, long block) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int r = wv; r < kRoiBitsRows; r += 4) {
    const long n = block * kRoiBitsRows + r;
    if (n >= M) return;
    const int* __restrict__ rowp = oh_labels + n * L;
    unsigned* __restrict__ dst = roi_bits + n * words;
    for (int c0 = 0; c0 < L; c0 += 64) {
      const int c = c0 + lane;
      const unsigned long long m = __ballot(c < L && rowp[c] == 1);
      const int wd = (c0 >> 5) + lane;                  // lanes 0 and 1 store the two words
      if (lane < 2 && wd < words) dst[wd] = lane ? (unsigned)(m >> 32) : (unsigned)m;
    }
  }
}

__global__ __launch_bounds__(256) void moi_roi_bits_kernel(const int* __restrict__ oh_labels,
                                                           unsigned* __restrict__ roi_bits,
                                                           long M, int L, int words) {
  moi_roi_bits_block(oh_labels, roi_bits, M, L, words, (long)blockIdx.x);
}

// Both bit tables of a multi-level call in ONE launch (five launches of 10-15 us each before: four levels' cell bits
// and the roi bits): the grid is the concatenation of the per-level cell-bit grids and the roi-bit grid.
struct MoiBitsPlan {
  unsigned* cell[8];
  int H[8], W[8], cpw[8];
  int first[9];            // first block of each level; first[n] = first roi-bits block
  int n;
  int sort_block;          // the block that orders the rois (-1: none)
  int grid;                // the ordering's coarse grid (cells per side over the image)
  float scale0;            // image -> level-0 cells
};

// rois in SPATIAL order for the pooling kernel: order[] = the rois sorted by (level, image, coarse centre row, coarse
// centre column) — a counting sort by one workgroup.  The pooling kernel gives each XCD a contiguous run of that order,
// so proposals piled on one object (and neighbours on the map) are pooled together and find each other's feature rows
// in that XCD's L2: fabric-side reads of the bench's call 1.26 GB -> 0.06 GB (rocprofv3 FETCH_SIZE).  Order inside a
// bucket is whatever the atomics give; nothing but the schedule depends on it.
constexpr int kSortBuckets = 4096;
__device__ __forceinline__ void moi_sort_block(const MoiBitsPlan& plan, const float* __restrict__ rois,
                                               const int* __restrict__ roi_level, int M, int B, int* __restrict__ order,
                                               unsigned* __restrict__ hist) {
  const int t = threadIdx.x, G = plan.grid, nb = plan.n * B * G * G;
  auto key_of = [&](int n) {
    const float* r = rois + (size_t)n * 5;
    const int l = min(max(roi_level[n], 0), plan.n - 1), b = min(max((int)r[0], 0), B - 1);
    const float cx = (r[1] + r[3]) * 0.5f * plan.scale0, cy = (r[2] + r[4]) * 0.5f * plan.scale0;
    const int gx = min(max((int)(cx * (float)G / (float)plan.W[0]), 0), G - 1);
    const int gy = min(max((int)(cy * (float)G / (float)plan.H[0]), 0), G - 1);
    return ((l * B + b) * G + gy) * G + gx;
  };
  for (int i = t; i < nb; i += 256) hist[i] = 0u;
  __syncthreads();
  // (four rois' loads in flight per trip: one workgroup walks all M rois twice, a round trip per trip otherwise)
  for (int n0 = t; n0 < M; n0 += 1024) {
    int key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) key[u] = key_of(min(n0 + u * 256, M - 1));
#pragma unroll
    for (int u = 0; u < 4; ++u) if (n0 + u * 256 < M) atomicAdd(&hist[key[u]], 1u);
  }
  __syncthreads();
  // exclusive scan of the buckets: a thread sums its run, the runs are scanned by thread 0's wavefront serially per wave
  const int per = (nb + 255) / 256;
  unsigned run = 0;
  for (int i = t * per; i < min(nb, (t + 1) * per); ++i) run += hist[i];
  __shared__ unsigned run_sum[256];
  run_sum[t] = run;
  __syncthreads();
  if (t == 0) {
    unsigned acc = 0;
    for (int i = 0; i < 256; ++i) { const unsigned v = run_sum[i]; run_sum[i] = acc; acc += v; }
  }
  __syncthreads();
  unsigned base = run_sum[t];
  for (int i = t * per; i < min(nb, (t + 1) * per); ++i) { const unsigned v = hist[i]; hist[i] = base; base += v; }
  __syncthreads();
  for (int n0 = t; n0 < M; n0 += 1024) {
    int key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) key[u] = key_of(min(n0 + u * 256, M - 1));
#pragma unroll
    for (int u = 0; u < 4; ++u) if (n0 + u * 256 < M) order[atomicAdd(&hist[key[u]], 1u)] = n0 + u * 256;
  }
}

__global__ __launch_bounds__(256) void moi_bits_all_kernel(const MoiBitsPlan plan, const int* __restrict__ superpixels,
                                                           const int* __restrict__ oh_labels, unsigned* __restrict__ roi_bits,
                                                           int B, int Hs, int Ws, int L, int words, long roi_rows,
                                                           const float* __restrict__ rois, const int* __restrict__ roi_level,
                                                           int M, int* __restrict__ order, int skip) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem[];
  const int blk = blockIdx.x;
  if (skip) {   // (timing aid, tools/sweeps: leave parts of the launch out — results are then wrong)
    if (blk >= plan.first[plan.n] && blk != plan.sort_block && (skip & 16)) return;
    for (int l = 0; l < plan.n; ++l)
      if ((skip >> l & 1) && blk >= plan.first[l] && blk < plan.first[l + 1]) return;
  }
  if (blk == plan.sort_block) {
    __shared__ unsigned sort_hist[kSortBuckets];
    moi_sort_block(plan, rois, roi_level, M, B, order, sort_hist);
    return;
  }
  if (blk >= plan.first[plan.n]) {
    moi_roi_bits_block(oh_labels, roi_bits, roi_rows, L, words, (long)(blk - plan.first[plan.n]));
    return;
  }
  int l = 0;
  while (l + 1 < plan.n && blk >= plan.first[l + 1]) ++l;
  if (plan.cpw[l] == 0)     // (one thread per cell: moi_cell_bits_small)
    moi_cell_bits_small_any(superpixels, plan.cell[l], B, plan.H[l], plan.W[l], Hs, Ws, L, words, (long)(blk - plan.first[l]));
  else
    moi_cell_bits_block(superpixels, plan.cell[l], B, plan.H[l], plan.W[l], Hs, Ws, L, words, plan.cpw[l],
                        (long)(blk - plan.first[l]), smem);
}

struct BinRange { int hs, he, ws, we; };

__device__ __forceinline__ BinRange bin_range(const IBox& r, int ph, int pw, int PH, int PW,
                                              int H, int W) {
#pragma clang fp contract(off)
  const int rw = max(r.x1 - r.x0 + 1, 1), rh = max(r.y1 - r.y0 + 1, 1);
  const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
  BinRange q;
  q.hs = clampi((int)floorf((float)ph * bh) + r.y0, 0, H);
  q.he = clampi((int)ceilf((float)(ph + 1) * bh) + r.y0, 0, H);
  q.ws = clampi((int)floorf((float)pw * bw) + r.x0, 0, W);
  q.we = clampi((int)ceilf((float)(pw + 1) * bw) + r.x0, 0, W);
  return q;
}

// any(cell_bits & roi_bits) for one cell, evaluated by the whole wave.
__device__ __forceinline__ bool cell_hit(const unsigned* __restrict__ cell_row,
                                         const unsigned* __restrict__ roi_row, unsigned mine,
                                         int words, int lane) {
  unsigned hit = (lane < words) ? (cell_row[lane] & mine) : 0u;
  for (int i = 64 + lane; i < words; i += 64) hit |= cell_row[i] & roi_row[i];
  return __ballot(hit != 0u) != 0ull;
}

template <int VEC>
__device__ __forceinline__ void moi_pool_wave(
    const float* __restrict__ in, const float* __restrict__ rois,
    const unsigned* __restrict__ cell_bits, const unsigned* __restrict__ roi_bits,
    float* __restrict__ out, int* __restrict__ argmax, int C, int H, int W, int words,
    float scale, int PH, int PW, int n, int bin, int lane) {
  const int nbins = PH * PW;
  const int ph = bin / PW, pw = bin - ph * PW;
  const IBox r = round_box(rois + (size_t)n * 5, scale);
  const BinRange q = bin_range(r, ph, pw, PH, PW, H, W);
  const unsigned* __restrict__ rrow = roi_bits + (size_t)n * words;
  const unsigned mine = lane < words ? rrow[lane] : 0u;
  const float* __restrict__ plane = in + (size_t)r.b * H * W * C;
  const unsigned* __restrict__ cplane = cell_bits + (size_t)r.b * H * W * words;
  float* __restrict__ orow = out + ((size_t)n * nbins + bin) * C;
  int* __restrict__ arow = argmax + ((size_t)n * nbins + bin) * C;

  for (int cb = 0; cb < C; cb += 64 * VEC) {
    const int c = cb + lane * VEC;
    const bool live = c < C;
    float best[VEC];
    int at[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { best[v] = -FLT_MAX; at[v] = -1; }
    // The bin's cells in (h, w) order, FOUR at a time: their mask words are requested together, then the features
    // of those that hit, so a round trip to memory serves four cells instead of one (the walk is latency-bound:
    // mask word -> ballot -> feature row -> compare).  Comparison order = cell order (first maximum wins).
#ifndef JTSM_MOI_AHEAD
#define JTSM_MOI_AHEAD 4
#endif
    constexpr int kAhead = JTSM_MOI_AHEAD;
    const int bwid = q.we - q.ws, ncell = (q.he - q.hs) * bwid;
    // (the walk keeps its own (h, w) counters: the kernel is bound by instruction issue — rocprofv3 counted 850 vector
    // and 370 scalar instructions per wavefront for ~6 cells — and two integer divisions per cell were a large part)
    int ch = q.hs, cw = q.ws;
    for (int i0 = 0; i0 < ncell; i0 += kAhead) {
      int cell[kAhead];
      unsigned hb[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int idx = i0 + u;
        const int h = ch, w = cw;
        if (++cw >= q.we) { cw = q.ws; ++ch; }
        // inside the bin and the rounded (inclusive) box?  wave-uniform
        const bool in = idx < ncell && w >= r.x0 && w <= r.x1 && h >= r.y0 && h <= r.y1;
        cell[u] = in ? h * W + w : -1;
        hb[u] = 0u;
        if (in) {
          const unsigned* __restrict__ crow = cplane + (size_t)cell[u] * words;
          if (lane < words) hb[u] = crow[lane] & mine;
          for (int i = 64 + lane; i < words; i += 64) hb[u] |= crow[i] & rrow[i];
        }
      }
      bool hit[kAhead];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) hit[u] = cell[u] >= 0 && __ballot(hb[u] != 0u) != 0ull;
      float x[kAhead][VEC];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        if (!(hit[u] && live)) continue;
        const float* p = plane + (size_t)cell[u] * C + c;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(p);
          x[u][0] = t.x; x[u][1 % VEC] = t.y; x[u][2 % VEC] = t.z; x[u][3 % VEC] = t.w;
        } else {
          x[u][0] = p[0];
        }
      }
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        if (!(hit[u] && live)) continue;
#pragma unroll
        for (int v = 0; v < VEC; ++v)
          if (x[u][v] > best[v]) { best[v] = x[u][v]; at[v] = cell[u]; }
      }
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        orow[c + v] = at[v] == -1 ? 0.f : best[v];
        arow[c + v] = at[v];
      }
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void moi_pool_fwd_nhwc(
    const float* __restrict__ in, const float* __restrict__ rois,
    const unsigned* __restrict__ cell_bits, const unsigned* __restrict__ roi_bits,
    float* __restrict__ out, int* __restrict__ argmax, int C, int H, int W, int M, int words,
    float scale, int PH, int PW, const int* __restrict__ roi_level, int level) {
  const int nbins = PH * PW;
  // (readfirstlane: the wavefront index is uniform, but the compiler cannot see that through threadIdx — with it
  // the roi / bin / address arithmetic below runs on the scalar unit instead of once per lane)
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= (long)M * nbins) return;
  if (roi_level && roi_level[wave / nbins] != level) return;  // FPN: this roi lives on another level
  const int n = (int)(wave / nbins), bin = (int)(wave - (long)n * nbins);
  moi_pool_wave<VEC>(in, rois, cell_bits, roi_bits, out, argmax, C, H, W, words, scale, PH, PW, n, bin,
                     threadIdx.x & 63);
}

// All FPN levels in one launch: each roi reads the map of its own level.
constexpr int kMaxLevels = 8;
struct MoiLevels {
  const float* in[kMaxLevels];
  float* gin[kMaxLevels];
  const unsigned* cell[kMaxLevels];
  int H[kMaxLevels], W[kMaxLevels];
  float scale[kMaxLevels];
};

template <int VEC>
__global__ __launch_bounds__(256) void moi_pool_fwd_levels(
    const MoiLevels lv, const float* __restrict__ rois, const unsigned* __restrict__ roi_bits,
    float* __restrict__ out, int* __restrict__ argmax, int C, int M, int words, int PH, int PW,
    const int* __restrict__ roi_level, int nlevels) {
  const int nbins = PH * PW;
  // Workgroups are dealt round-robin over the 8 XCDs (private L2s): give each XCD a CONTIGUOUS run of (roi, bin)
  // pairs, so that the bins of a roi — which share a third of their cells — and neighbouring rois meet in one L2
  // (dealt out pair by pair every XCD fetched every roi's cells: 1.86 GB of fabric traffic per launch for 0.6 GB of
  // distinct bytes).  Speed only: any placement computes the same result.
  const unsigned nblk = gridDim.x, per = (nblk + 7) / 8, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const unsigned q = nblk >> 3, r8 = nblk & 7;   // XCDs 0 .. r8-1 hold q + 1 workgroups, the rest q
  const unsigned blk = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
  (void)per;
  // (readfirstlane: the wavefront index is uniform, but the compiler cannot see that through threadIdx — with it
  // the roi / bin / address arithmetic below runs on the scalar unit instead of once per lane)
  const long wave = (long)blk * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= (long)M * nbins) return;
  const int n = (int)(wave / nbins), bin = (int)(wave - (long)n * nbins);
  const int l = roi_level[n];
  if ((unsigned)l >= (unsigned)nlevels) return;
  moi_pool_wave<VEC>(lv.in[l], rois, lv.cell[l], roi_bits, out, argmax, C, lv.H[l], lv.W[l], words, lv.scale[l],
                     PH, PW, n, bin, threadIdx.x & 63);
}

// ---- forward, one wavefront per (roi, bin ROW) (round 4) --------------------------------------------------------------
// The per-(roi, bin) form above is a chain of dependent round trips per wavefront (roi -> mask words -> ballot -> feature
// rows, four cells at a time) at 196 000 wavefronts per call: load latency x occupancy bounds it (DESIGN §5 dead ends
// 21-23), and every cell on a bin border is tested and fetched once per bin that contains it.  Here a wavefront owns the
// PW bins of one bin row of one roi:
//   1. hit test, lane = CELL: the rows of the bin row are cut into batches of 64 consecutive cells; a lane ANDs its
//      cell's whole bit row against the roi's words (uniform loads) with all its 16-byte loads in flight, one ballot
//      per 64 cells, and the hit cells are written — in (h, w) order, by prefix popcount — to a wave-private LDS list;
//   2. pooling: the list is walked AHEAD cells at a time; every address is known before the first load, so a round trip
//      serves AHEAD feature rows (16 B per lane, 1 KiB per row), and each row updates the accumulators of the bins whose
//      column range holds its w (wave-uniform tests; a cell is fetched once per bin ROW, not once per bin).
// Order inside a bin stays h outer / w inner with a strict '>', so values and arg-max are the reference's bit for bit
// (MOIPool_cuda.cu:244-300).  A seventh of the wavefronts, each with ~2 + hits / AHEAD round trips instead of ~2 per
// four cells of every bin.
constexpr int kRowsList = 1024;        // LDS list entries per wavefront = 16 batches of 64 cells
constexpr int kRowsBatches = kRowsList / 64;
// workgroups (of 4 wavefronts = bin rows) per XCD chunk: 28 = 16 rois x 7 bin rows (JTSM_MOI_CHUNK overrides, for sweeps)
static unsigned moi_rows_chunk() {
  static const unsigned v = [] { const char* e = getenv("JTSM_MOI_CHUNK"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : 28u; }();
  return v;
}

template <int VEC, int AHEAD, int PWT, int W4, int NW, bool PIPE>
__global__ __launch_bounds__(64 * NW) void moi_pool_fwd_rows(
    const MoiLevels lv, const float* __restrict__ rois, const unsigned* __restrict__ roi_bits,
    float* __restrict__ out, int* __restrict__ argmax, int C, int M, int words, int PH,
    const int* __restrict__ roi_level, int nlevels, int only_level, const int* __restrict__ order, unsigned chunk) {
#pragma clang fp contract(off)
  __shared__ unsigned hit_list[NW][kRowsList];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // Workgroups are dealt round-robin over the 8 XCDs (private L2s).  The rois come in spatial order (moi_sort_block), so
  // CHUNKS of `chunk` consecutive workgroups (16 rois: neighbours on the map, usually one pile of proposals) go to one
  // XCD, and the chunks are dealt round-robin: every XCD sees an even mix of levels and roi sizes (whole eighths of the
  // sorted list — all small rois on one XCD, all large ones on another — measured 393 us against 353 unsorted), while
  // the rows a pile shares are fetched by one or two L2s instead of eight.  The grid is padded to whole rounds of chunks.
  const unsigned xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const unsigned blk = ((j / chunk) * 8 + xcd) * chunk + j % chunk;
  // wavefront -> (roi, bin row, channel block of 64 * VEC): the channel blocks of a bin row are neighbours in a workgroup
  const int ncb = (C + 64 * VEC - 1) / (64 * VEC);
  const long wave = (long)blk * NW + wv;
  if (wave >= (long)M * PH * ncb) return;
  const int cbi = (int)(wave % ncb);
  const long np = wave / ncb;
  const int slot = (int)(np / PH), ph = (int)(np - (long)slot * PH);
  const int n = order ? order[slot] : slot;        // (spatial order of the rois: moi_sort_block)
  int l = 0;
  if (roi_level) {
    l = roi_level[n];
    if (only_level >= 0) {
      if (l != only_level) return;     // FPN, per-level entry point: this roi lives on another level
      l = 0;
    } else if ((unsigned)l >= (unsigned)nlevels) {
      return;
    }
  }
  const int H = lv.H[l], W = lv.W[l];
  const IBox r = round_box(rois + (size_t)n * 5, lv.scale[l]);
  const int rw = max(r.x1 - r.x0 + 1, 1), rh = max(r.y1 - r.y0 + 1, 1);
  const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PWT;
  // rows of this bin row inside the rounded (inclusive) box and the map; column ranges of its PWT bins likewise
  const int hs = max(clampi((int)floorf((float)ph * bh) + r.y0, 0, H), r.y0);
  const int he = min(clampi((int)ceilf((float)(ph + 1) * bh) + r.y0, 0, H), r.y1 + 1);
  int wsb[PWT], web[PWT];
#pragma unroll
  for (int pw = 0; pw < PWT; ++pw) {
    wsb[pw] = max(clampi((int)floorf((float)pw * bw) + r.x0, 0, W), r.x0);
    web[pw] = min(clampi((int)ceilf((float)(pw + 1) * bw) + r.x0, 0, W), r.x1 + 1);
  }
  const int cx0 = clampi(r.x0, 0, W), cx1 = clampi(r.x1 + 1, 0, W);
  const int nbr = (cx1 - cx0 + 63) >> 6;                       // batches of 64 cells per row
  const int nrows = he > hs ? he - hs : 0;
  const unsigned* __restrict__ rrow = roi_bits + (size_t)n * words;
  const float* __restrict__ plane = lv.in[l] + (size_t)r.b * H * W * C;
  const unsigned* __restrict__ cplane = lv.cell[l] + (size_t)r.b * H * W * words;
  unsigned* __restrict__ list = hit_list[wv];
  const int nbins = PH * PWT;
  const unsigned long long below = (1ull << lane) - 1ull;
  // the roi's label words, once, in scalar registers (words == 4 * W4: the host picks the instantiation)
  uint4 rq[W4];
#pragma unroll
  for (int i = 0; i < W4; ++i) {
    const uint4 t = reinterpret_cast<const uint4*>(rrow)[i];
    rq[i].x = __builtin_amdgcn_readfirstlane(t.x); rq[i].y = __builtin_amdgcn_readfirstlane(t.y);
    rq[i].z = __builtin_amdgcn_readfirstlane(t.z); rq[i].w = __builtin_amdgcn_readfirstlane(t.w);
  }

  {
    const int c = cbi * 64 * VEC + lane * VEC;
    const bool live = c < C;
    const int c_safe = live ? c : 0;     // (loads are unconditional: a lane beyond C reads channel 0 and stores nothing)
    float best[PWT][VEC];
    int at[PWT][VEC];
#pragma unroll
    for (int pw = 0; pw < PWT; ++pw)
#pragma unroll
      for (int v = 0; v < VEC; ++v) { best[pw][v] = -FLT_MAX; at[pw][v] = -1; }

    int hrow = hs, kb = 0;                       // next batch: row, 64-cell block of the row
    const int nbatch = nrows * nbr;
    for (int b0 = 0; b0 < nbatch; b0 += kRowsBatches) {
      // ---- 1. hit test of up to kRowsBatches batches
      const int nb = min(kRowsBatches, nbatch - b0);
      int count = 0;
      for (int bb = 0; bb < nb; ++bb) {
        // (no lane-dependent branch in front of the loads: behind one the compiler drains the memory counter before
        // every load — a lane beyond the row's end reads the row's last cell and drops the result)
        const int w = cx0 + (kb << 6) + lane;
        const uint4* __restrict__ crow =
            reinterpret_cast<const uint4*>(cplane + ((size_t)hrow * W + min(w, cx1 - 1)) * words);
        constexpr int CH = W4 < 8 ? W4 : 8;      // 16-byte loads in flight per lane (wider label sets: several rounds)
        unsigned acc = 0u;
#pragma unroll
        for (int c0 = 0; c0 < W4; c0 += CH) {
          uint4 cw[CH];
#pragma unroll
          for (int i = 0; i < CH; ++i) cw[i] = crow[c0 + i];
#pragma unroll
          for (int i = 0; i < CH; ++i)
            acc |= (cw[i].x & rq[c0 + i].x) | (cw[i].y & rq[c0 + i].y) | (cw[i].z & rq[c0 + i].z) | (cw[i].w & rq[c0 + i].w);
        }
        acc = w < cx1 ? acc : 0u;
        const unsigned long long m = __ballot(acc != 0u);
        if (acc != 0u) list[count + __popcll(m & below)] = ((unsigned)hrow << 16) | (unsigned)w;
        count += __popcll(m);
        if (++kb == nbr) { kb = 0; ++hrow; }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- 2. the hit cells, AHEAD feature rows per round trip; two groups alternate, one's loads in flight while the
      // other is applied.  (Entries beyond the list's end repeat its last cell: a row applied twice changes nothing under
      // a strict '>', and the loads stay free of branches.)
      auto fetch = [&](int i0, int (&cell)[AHEAD], int (&cw_)[AHEAD], float (&x)[AHEAD][VEC]) {
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
          const unsigned e = __builtin_amdgcn_readfirstlane(list[min(i0 + u, count - 1)]);
          cw_[u] = (int)(e & 0xffffu);
          cell[u] = (int)(e >> 16) * W + cw_[u];
        }
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
          const float* p = plane + (size_t)cell[u] * C + c_safe;
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4*>(p);
            x[u][0] = t.x; x[u][1 % VEC] = t.y; x[u][2 % VEC] = t.z; x[u][3 % VEC] = t.w;
          } else if (VEC == 2) {
            const float2 t = *reinterpret_cast<const float2*>(p);
            x[u][0] = t.x; x[u][1 % VEC] = t.y;
          } else {
            x[u][0] = p[0];
          }
        }
      };
      auto apply = [&](const int (&cell)[AHEAD], const int (&cw_)[AHEAD], const float (&x)[AHEAD][VEC]) {
#pragma unroll
        for (int u = 0; u < AHEAD; ++u) {
#pragma unroll
          for (int pw = 0; pw < PWT; ++pw) {
            if (cw_[u] < wsb[pw] || cw_[u] >= web[pw]) continue;      // wave-uniform
#pragma unroll
            for (int v = 0; v < VEC; ++v)
              if (x[u][v] > best[pw][v]) { best[pw][v] = x[u][v]; at[pw][v] = cell[u]; }
          }
        }
      };
      if (PIPE) {
        int ca[AHEAD], wa[AHEAD], cb_[AHEAD], wb[AHEAD];
        float xa[AHEAD][VEC], xb[AHEAD][VEC];
        if (count > 0) fetch(0, ca, wa, xa);
        for (int i0 = 0; i0 < count; i0 += 2 * AHEAD) {
          fetch(i0 + AHEAD, cb_, wb, xb);
          apply(ca, wa, xa);
          fetch(i0 + 2 * AHEAD, ca, wa, xa);
          apply(cb_, wb, xb);
        }
      } else {
        for (int i0 = 0; i0 < count; i0 += AHEAD) {
          int ca[AHEAD], wa[AHEAD];
          float xa[AHEAD][VEC];
          fetch(i0, ca, wa, xa);
          apply(ca, wa, xa);
        }
      }
      __builtin_amdgcn_wave_barrier();          // the list is rewritten by the next chunk
    }
    if (live) {
#pragma unroll
      for (int pw = 0; pw < PWT; ++pw) {
        const size_t o = ((size_t)n * nbins + ph * PWT + pw) * C + c;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          out[o + v] = at[pw][v] == -1 ? 0.f : best[pw][v];
          argmax[o + v] = at[pw][v];
        }
      }
    }
  }
}

// JTSM_MOI_FWD_ROWS: 0 restores the per-(roi, bin) kernel; 2 = four cells per round trip instead of eight (A/B);
// JTSM_MOI_SORT=0 switches the spatial ordering of the rois off.
// Measured on the bench's call (4000 rois, 4 levels, 256 channels; whole forward entry point, hipEvents): per-(roi, bin)
// 441-450 us; this kernel 370 (8 cells per round trip, 108 registers), 378 (4 cells), 480 (2 channels per lane: a
// 512-byte wave-load costs the memory pipeline what a 1 KiB one does), 390 with seven-wavefront workgroups (one roi),
// 352-390 with two groups of rows alternating (software pipelining buys nothing: the kernel runs at the ~40 GB/s per CU
// the chip delivers to every CU at once) — and 352 with the bit tables in one launch.
static int moi_fwd_rows_mode() {
  static const int v = [] { const char* e = getenv("JTSM_MOI_FWD_ROWS"); return e ? atoi(e) : 1; }();
  return v;
}
static int moi_bits_skip() {
  static const int v = [] { const char* e = getenv("JTSM_MOI_BITS_SKIP"); return e ? atoi(e) : 0; }();
  return v;
}
static bool moi_sort_on() {
  static const bool v = [] { const char* e = getenv("JTSM_MOI_SORT"); return !e || atoi(e) != 0; }();
  return v;
}

template <int AHEAD, int W4>
static void launch_fwd_rows_w(const MoiLevels& lv, const float* rois, const unsigned* roi_bits, float* out, int* argmax,
                              int C, int M, int words, int PH, const int* roi_level, int nlevels, int only_level,
                              const int* order, hipStream_t st) {
  const long waves = (long)M * PH * ceil_div(C, 256);
  const unsigned chunk = moi_rows_chunk();
  const long round = 8 * (long)chunk;                           // (workgroups beyond the last wavefront leave at once)
  hipLaunchKernelGGL((moi_pool_fwd_rows<4, AHEAD, 7, W4, 4, false>), dim3(ceil_div(ceil_div(waves, 4), round) * round), dim3(256), 0, st, lv, rois,
                     roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, chunk);
}
template <int AHEAD>
static void launch_fwd_rows_as(const MoiLevels& lv, const float* rois, const unsigned* roi_bits, float* out, int* argmax,
                               int C, int M, int words, int PH, const int* roi_level, int nlevels, int only_level,
                               const int* order, hipStream_t st) {
  switch (words) {      // (moi_fwd_rows_words_ok: 4, 8, 16, 32 or 64 words = up to 128 ... 2048 superpixel ids)
    case 4: return launch_fwd_rows_w<AHEAD, 1>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
    case 8: return launch_fwd_rows_w<AHEAD, 2>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
    case 16: return launch_fwd_rows_w<AHEAD, 4>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
    case 64: return launch_fwd_rows_w<AHEAD, 16>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
    default: return launch_fwd_rows_w<AHEAD, 8>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
  }
}
static bool moi_fwd_rows_words_ok(int words) { return words == 4 || words == 8 || words == 16 || words == 32 || words == 64; }
static void launch_fwd_rows(const MoiLevels& lv, const float* rois, const unsigned* roi_bits, float* out, int* argmax, int C,
                            int M, int words, int PH, const int* roi_level, int nlevels, int only_level, const int* order,
                            hipStream_t st) {
  if (moi_fwd_rows_mode() == 2)
    return launch_fwd_rows_as<4>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
  return launch_fwd_rows_as<8>(lv, rois, roi_bits, out, argmax, C, M, words, PH, roi_level, nlevels, only_level, order, st);
}

// grad_input[level(n)][b, argmax, c] += grad[n, bin, c]: a wavefront per (roi, bin).
__global__ __launch_bounds__(256) void moi_pool_bwd_levels(
    const MoiLevels lv, const float* __restrict__ grad, const float* __restrict__ rois,
    const int* __restrict__ argmax, int C, int M, int nbins, const int* __restrict__ roi_level, int nlevels,
    const int* __restrict__ census_max, int census_limit) {
  if (census_max && *census_max <= census_limit) return;   // the gather form took this call
  const int lane = threadIdx.x & 63;
  for (long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); wave < (long)M * nbins;
       wave += (long)gridDim.x * 4) {
    const int n = (int)(wave / nbins);
    const int l = roi_level[n];
    if ((unsigned)l >= (unsigned)nlevels) continue;
    const int b = (int)rois[(size_t)n * 5];
    float* __restrict__ g = lv.gin[l] + (size_t)b * lv.H[l] * lv.W[l] * C;
    const float* __restrict__ src = grad + (size_t)wave * C;
    const int* __restrict__ arg = argmax + (size_t)wave * C;
    // one channel per lane: neighbouring channels mostly share their winning cell, so a wave-instruction's 64
    // atomics land in a few contiguous 256-byte runs (the shape the L2 atomic units take at full rate)
    for (int c = lane; c < C; c += 64) {
      const int a = arg[c];
      if (a != -1) atomicAdd(g + (size_t)a * C + c, src[c]);
    }
  }
}

// Gather form of the backward: a workgroup owns an 8 x 8-cell tile of one level's gradient map (x 256 channels, 64 KB
// of LDS), finds the (roi, bin) pairs whose bin range touches the tile from the roi geometry alone, and adds their
// gradients into LDS — thread c owns channel c, so there are no atomics, the order is the roi / bin order (bitwise
// reproducible), and every cell of the map is written exactly once (no zero fill, no read-modify-write in L2; the
// scatter form above spends its time in L2 atomic line operations).
constexpr int kTile = 8;      // tile of the census and of the per-tile gather (cells, both axes)
constexpr int kTileY = 8;
constexpr int kQuad = 4;      // the busy-tile gather splits a tile into 2 x 2 quadrants of 4 x 4 cells
struct MoiTiles {
  int first[kMaxLevels + 1];   // first workgroup of each level
  int tiles_x[kMaxLevels], tiles_y[kMaxLevels];
};


// rois of one (level, image), in index order: lists[(l * B + b) * M ...], counts[l * B + b].  One workgroup each.
__global__ __launch_bounds__(256) void moi_roi_lists_kernel(const float* __restrict__ rois,
                                                            const int* __restrict__ roi_level, int M, int B,
                                                            int* __restrict__ lists, int* __restrict__ counts) {
  __shared__ int wave_count[4];
  const int l = blockIdx.x / B, b = blockIdx.x % B;
  int* __restrict__ mine = lists + (size_t)blockIdx.x * M;
  int filled = 0;
  for (int base = 0; base < M; base += 256) {
    const int n = base + threadIdx.x;
    const bool hit = n < M && roi_level[n] == l && (int)rois[(size_t)n * 5] == b;
    int cnt;
    const int slot = compact256(hit, wave_count, cnt);
    if (hit) mine[filled + slot] = n;
    filled += cnt;
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = filled;
}

// How many (roi, bin) pairs each tile will have to walk (estimated from the box geometry): the gather handles a tile
// in ONE workgroup, so a tile under very many overlapping rois (proposals piled on one object) would take longer than
// the whole scatter form.  The census lets the call fall back: above kCensusLimit pairs on one tile the gather only
// clears the maps and the float-atomic scatter (whose cost does not depend on where the rois are) does the work.
constexpr int kCensusLimit = 64000;   // (estimated) rows on one tile above which the scatter form (~1 ms whatever the rois) takes the call
// rows on one tile from which its four quadrants go to moi_pool_bwd_busy (overridable for sweeps)
static int moi_heavy_min() {
  static const int v = [] { const char* e = getenv("JTSM_MOI_HEAVY_MIN"); return e ? atoi(e) : 1024; }();
  return v;
}

__global__ __launch_bounds__(256) void moi_tile_census_kernel(const MoiLevels lv, const MoiTiles tl,
                                                              const float* __restrict__ rois,
                                                              const int* __restrict__ roi_level, int M, int nlevels,
                                                              int PH, int PW, int* __restrict__ census) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= M) return;
  const int l = roi_level[n];
  if ((unsigned)l >= (unsigned)nlevels) return;
  const IBox r = round_box(rois + (size_t)n * 5, lv.scale[l]);
  const int H = lv.H[l], W = lv.W[l];
  const int xa = max(r.x0, 0), xz = min(r.x1, W - 1), ya = max(r.y0, 0), yz = min(r.y1, H - 1);
  if (xa > xz || ya > yz) return;
  const float bw = (float)max(r.x1 - r.x0 + 1, 1) / (float)PW, bh = (float)max(r.y1 - r.y0 + 1, 1) / (float)PH;
  for (int ty = ya / kTileY; ty <= yz / kTileY; ++ty) {
    const int oy = min(yz, ty * kTileY + kTileY - 1) - max(ya, ty * kTileY) + 1;   // overlapped rows
    const int by = min(PH, (int)((float)oy / bh) + 2);                            // bins touching them (upper bound)
    for (int tx = xa / kTile; tx <= xz / kTile; ++tx) {
      const int ox = min(xz, tx * kTile + kTile - 1) - max(xa, tx * kTile) + 1;
      const int bx = min(PW, (int)((float)ox / bw) + 2);
      atomicAdd(&census[tl.first[l] + (r.b * tl.tiles_y[l] + ty) * tl.tiles_x[l] + tx], by * bx);
    }
  }
}

// Gather form, round 3.  What round 2's profile showed (rocprofv3 counters on the bench's proposals): the launch
// averaged 2 resident wavefronts per CU — it was one long tail.  Proposals pile up on objects, small rois put all 49
// bins of every piled roi on the same few cells, and a tile was ONE workgroup whose 256 threads (thread = channel)
// added its ~2500 rows one dependent LDS read-modify-write after the other.  Now: (1) the maps are cleared by a memset
// and only the tiles some box overlaps get a workgroup (the plan above), heaviest first, through a fixed grid that
// strides the list; (2) a tile is 4 x 4 cells and its workgroup has 16 wavefronts: thread = (channel, copy), FOUR
// copies of the tile in LDS, the row list dealt to the copies in batches of 16 rows, each copy's additions in list
// order, the copies added in a fixed order at the end — still no atomics, still reproducible bit for bit; (3) a
// batch's loads are in flight while the previous batch is applied.  (LDS float atomics — ds_add_f32, fire and forget —
// were measured for the additions: 2-3x SLOWER than the plain read-modify-write.)
constexpr int kPairCap = 1024;

// Gather form of the backward: a workgroup owns an 8 x 8-cell tile of one level's gradient map (x 256 channels, 64 KB
// of LDS), finds the (roi, bin) pairs whose bin range touches the tile from the roi geometry alone, and adds their
// gradients into LDS — thread c owns channel c, so there are no atomics, the order is the roi / bin order (bitwise
// reproducible), and every cell of the map is written exactly once (no zero fill, no read-modify-write in L2; the
// scatter form above spends its time in L2 atomic line operations).  Tiles no box overlaps (census 0: most of a call)
// write their zeros at once, 16 bytes per lane; tiles at or above `heavy_min` (estimated) rows are left to
// moi_pool_bwd_busy below.
__global__ __launch_bounds__(256) void moi_pool_bwd_tiled(
    const MoiLevels lv, const MoiTiles tl, const float* __restrict__ grad, const float* __restrict__ rois,
    const int* __restrict__ argmax, int C, int M, int PH, int PW, int B, const int* __restrict__ lists,
    const int* __restrict__ counts, int nlevels, const int* __restrict__ plan, const int* __restrict__ census,
    int heavy_min, int accumulate, const int* __restrict__ tile_order) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [64 cells][256 channels]
  __shared__ int roi_list[256];
  __shared__ IBox roi_box[256];          // their rounded boxes (rounded once per roi, not once per (roi, bin))
  __shared__ int pair_list[kPairCap];   // (roi * nbins + bin) rows whose bin range touches the tile, in order
  __shared__ int wave_count[4];
  const int t = threadIdx.x;
  // workgroups start in index order: through the plan's permutation the heaviest tiles start first (tile_plan_kernel)
  const int tile = tile_order ? tile_order[blockIdx.x] : (int)blockIdx.x;
  int l = 0;
  while (l + 1 < nlevels && tile >= tl.first[l + 1]) ++l;
  const int H = lv.H[l], W = lv.W[l];
  int rel = tile - tl.first[l];
  const int tx = rel % tl.tiles_x[l]; rel /= tl.tiles_x[l];
  const int ty = rel % tl.tiles_y[l];
  const int b = rel / tl.tiles_y[l];
  const int x0 = tx * kTile, y0 = ty * kTile, x1 = min(x0 + kTile, W) - 1, y1 = min(y0 + kTile, H) - 1;
  const int c = blockIdx.y * 256 + t;
  const float scale = lv.scale[l];
  const int nbins = PH * PW;
  float* __restrict__ out = lv.gin[l] + (size_t)b * H * W * C;
  const int load = census[tile];
  const bool fallback = plan[0] > kCensusLimit;   // too many rois on one tile somewhere: the scatter form adds into zeros
  if (load == 0 || fallback) {
    if (accumulate) return;   // the map holds another consumer's gradient: nothing to add here (the scatter form adds to it)
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = t; i < kTile * kTile * 64; i += 256) {
      const int cell = i >> 6, y = y0 + cell / kTile, x = x0 + cell % kTile;
      if (y <= y1 && x <= x1)
        *reinterpret_cast<float4*>(out + ((size_t)y * W + x) * C + blockIdx.y * 256 + (i & 63) * 4) = z;
    }
    return;
  }
  if (load >= heavy_min) return;   // moi_pool_bwd_busy writes this tile
  for (int i = t; i < kTile * kTile * 64; i += 256) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  int np = 0;   // uniform

  // a / W by multiply-high (exact for a * W < 2^32: a is a cell index of one map; the host sends larger maps to the
  // scatter form): the drain runs it once per (row, channel)
  const unsigned magicW = (unsigned)((0x100000000ull + (unsigned)W - 1) / (unsigned)W);
  const unsigned magic_bins = (unsigned)((0x100000000ull + (unsigned)nbins - 1) / (unsigned)nbins);   // k < 256 * nbins
  const unsigned magic_pw = (unsigned)((0x100000000ull + (unsigned)PW - 1) / (unsigned)PW);
  constexpr int kDepth = 16;
  float* __restrict__ mine_acc = acc + t;
  const unsigned rows_in = (unsigned)(y1 - y0 + 1), cols_in = (unsigned)(x1 - x0 + 1);
  auto fetch = [&](int j, int (&a)[kDepth], float (&g)[kDepth]) {   // rows j .. j + kDepth - 1 of the list
#pragma unroll
    for (int u = 0; u < kDepth; ++u) {
      const size_t row = (size_t)pair_list[min(j + u, np - 1)];
      a[u] = j + u < np ? argmax[row * C + c] : -1;
      g[u] = j + u < np ? grad[row * C + c] : 0.f;
    }
  };
  auto apply = [&](const int (&a)[kDepth], const float (&g)[kDepth]) {
#pragma unroll
    for (int u = 0; u < kDepth; ++u) {
      const unsigned ay = __umulhi((unsigned)a[u], magicW), ax = (unsigned)a[u] - ay * (unsigned)W;
      const unsigned ry = ay - (unsigned)y0, rx = ax - (unsigned)x0;
      if ((a[u] >= 0) & (ry < rows_in) & (rx < cols_in)) mine_acc[(int)(ry * kTile + rx) * 256] += g[u];
    }
  };
  // add the listed rows' gradients into the tile, in list order; two batches of kDepth rows alternate, the loads of one
  // in flight while the other is applied
  auto drain = [&]() {
    __syncthreads();
    int a0[kDepth], a1[kDepth];
    float g0[kDepth], g1[kDepth];
    if (np > 0) fetch(0, a0, g0);
    for (int j = 0; j < np; j += 2 * kDepth) {
      if (j + kDepth < np) fetch(j + kDepth, a1, g1);
      apply(a0, g0);
      if (j + kDepth < np) {
        if (j + 2 * kDepth < np) fetch(j + 2 * kDepth, a0, g0);
        apply(a1, g1);
      }
    }
    np = 0;
  };

  const int* __restrict__ mine = lists + (size_t)(l * B + b) * M;
  const int nmine = counts[l * B + b];
  for (int base = 0; base < nmine; base += 256) {
    // ---- rois of this level / image whose box touches the tile, in index order
    const int n = base + t < nmine ? mine[base + t] : -1;
    bool hit = false;
    IBox r = {};
    if (n >= 0) {
      r = round_box(rois + (size_t)n * 5, scale);
      hit = r.x0 <= x1 && r.x1 >= x0 && r.y0 <= y1 && r.y1 >= y0;
    }
    int nroi;
    const int slot = compact256(hit, wave_count, nroi);
    if (hit) {
      roi_list[slot] = n;
      roi_box[slot] = r;
    }
    __syncthreads();
    // ---- their bins whose cell range touches the tile, in (roi, bin) order
    const int ncombo = nroi * nbins;
    for (int k0 = 0; k0 < ncombo; k0 += 256) {
      if (np > kPairCap - 256) drain();
      const int k = k0 + t;
      bool bh = false;
      int row = 0;
      if (k < ncombo) {
        const int i = (int)__umulhi((unsigned)k, magic_bins), bin = k - i * nbins, m = roi_list[i];
        const IBox rb = roi_box[i];
        const int ph = (int)__umulhi((unsigned)bin, magic_pw);
        const BinRange q = bin_range(rb, ph, bin - ph * PW, PH, PW, H, W);
        bh = q.he > q.hs && q.we > q.ws && q.hs <= y1 && q.he > y0 && q.ws <= x1 && q.we > x0;
        row = m * nbins + bin;
      }
      int cnt;
      const int ps = compact256(bh, wave_count, cnt);
      if (bh) pair_list[np + ps] = row;
      np += cnt;
    }
    __syncthreads();   // roi_list is rewritten by the next chunk
  }
  drain();
  __syncthreads();
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x)
      {
        float* dst = out + ((size_t)y * W + x) * C + c;
        const float v = acc[((y - y0) * kTile + (x - x0)) * 256 + t];
        *dst = accumulate ? *dst + v : v;
      }
}

constexpr int kCopies = 4;
constexpr int kPairCapBusy = 2048;

__global__ __launch_bounds__(1024) void moi_pool_bwd_busy(
    const MoiLevels lv, const MoiTiles tl, const float* __restrict__ grad, const float* __restrict__ rois,
    const int* __restrict__ argmax, int C, int M, int PH, int PW, int B, const int* __restrict__ lists,
    const int* __restrict__ counts, int nlevels, const int* __restrict__ plan, int accumulate) {
#pragma clang fp contract(off)
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [kCopies][kQuad * kQuad cells][256 channels]
  __shared__ int roi_list[256];
  __shared__ IBox roi_box[256];            // their rounded boxes (rounded once per roi, not once per (roi, bin))
  __shared__ int pair_list[kPairCapBusy];  // (roi * nbins + bin) rows whose bin range touches the tile, in order
  __shared__ int wave_count[16];
  if (plan[0] > kCensusLimit) return;      // piled-up rois beyond what one workgroup should walk: the scatter form
  const int nbusy = plan[1];
  const int t = threadIdx.x, ch = t & 255, copy = t >> 8;
  constexpr int kCells = kQuad * kQuad;
  const int nbins = PH * PW;
  const unsigned magic_bins = (unsigned)((0x100000000ull + (unsigned)nbins - 1) / (unsigned)nbins);   // k < 256 * nbins
  const unsigned magic_pw = (unsigned)((0x100000000ull + (unsigned)PW - 1) / (unsigned)PW);
  float* __restrict__ mine_acc = acc + copy * kCells * 256 + ch;
  constexpr int kDepth = 16;

  for (int job = blockIdx.x; job < nbusy; job += gridDim.x) {
    const int entry = plan[2 + job], tile = entry >> 2, quad = entry & 3;   // (tile of the census grid, its quadrant)
    int l = 0;
    while (l + 1 < nlevels && tile >= tl.first[l + 1]) ++l;
    const int H = lv.H[l], W = lv.W[l];
    int rel = tile - tl.first[l];
    const int tx = rel % tl.tiles_x[l]; rel /= tl.tiles_x[l];
    const int ty = rel % tl.tiles_y[l];
    const int b = rel / tl.tiles_y[l];
    const int x0 = tx * kTile + (quad & 1) * kQuad, y0 = ty * kTileY + (quad >> 1) * kQuad;
    if (x0 >= W || y0 >= H) continue;        // (a quadrant beyond the map's edge; uniform)
    const int x1 = min(x0 + kQuad, W) - 1, y1 = min(y0 + kQuad, H) - 1;
    const float scale = lv.scale[l];
    float* __restrict__ out = lv.gin[l] + (size_t)b * H * W * C;
    // a / W by multiply-high (exact for a * W < 2^32: a is a cell index of one map; the host sends larger maps to the
    // scatter form): the drain runs it once per (row, channel)
    const unsigned magicW = (unsigned)((0x100000000ull + (unsigned)W - 1) / (unsigned)W);
    const unsigned rows_in = (unsigned)(y1 - y0 + 1), cols_in = (unsigned)(x1 - x0 + 1);

    for (int cb = 0; cb < C; cb += 256) {   // (256-channel blocks: one on the JTSM path)
      const int c = cb + ch;
      __syncthreads();                       // the previous job's / block's readers of acc and the lists are done
      for (int i = t; i < kCopies * kCells * 64; i += 1024) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      int np = 0;   // uniform

      auto fetch = [&](int j, int (&a)[kDepth], float (&g)[kDepth]) {   // rows j .. j + kDepth - 1 of the list
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
          const size_t row = (size_t)pair_list[min(j + u, np - 1)];
          a[u] = j + u < np ? argmax[row * C + c] : -1;
          g[u] = j + u < np ? grad[row * C + c] : 0.f;
        }
      };
      auto apply = [&](const int (&a)[kDepth], const float (&g)[kDepth]) {
#pragma unroll
        for (int u = 0; u < kDepth; ++u) {
          const unsigned ay = __umulhi((unsigned)a[u], magicW), ax = (unsigned)a[u] - ay * (unsigned)W;
          const unsigned ry = ay - (unsigned)y0, rx = ax - (unsigned)x0;
          if ((a[u] >= 0) & (ry < rows_in) & (rx < cols_in)) mine_acc[(int)(ry * kQuad + rx) * 256] += g[u];
        }
      };
      // copy k takes the batches k, k + 4, ... of kDepth rows; two batches alternate, one's loads in flight while the
      // other is applied
      auto drain = [&]() {
        __syncthreads();
        int a0[kDepth], a1[kDepth];
        float g0[kDepth], g1[kDepth];
        constexpr int kStep = kCopies * kDepth;
        const int j0 = copy * kDepth;
        if (j0 < np) fetch(j0, a0, g0);
        for (int j = j0; j < np; j += 2 * kStep) {
          if (j + kStep < np) fetch(j + kStep, a1, g1);
          apply(a0, g0);
          if (j + kStep < np) {
            if (j + 2 * kStep < np) fetch(j + 2 * kStep, a0, g0);
            apply(a1, g1);
          }
        }
        np = 0;
        __syncthreads();                     // pair_list is rewritten next
      };

      const int* __restrict__ mine = lists + (size_t)(l * B + b) * M;
      const int nmine = counts[l * B + b];
      for (int base = 0; base < nmine; base += 256) {
        // ---- rois of this level / image whose box touches the tile, in index order
        const int n = (t < 256 && base + t < nmine) ? mine[base + t] : -1;
        bool hit = false;
        IBox r = {};
        if (n >= 0) {
          r = round_box(rois + (size_t)n * 5, scale);
          hit = r.x0 <= x1 && r.x1 >= x0 && r.y0 <= y1 && r.y1 >= y0;
        }
        int nroi;
        const int slot = compact_wg<16>(hit, wave_count, nroi);
        if (hit) {
          roi_list[slot] = n;
          roi_box[slot] = r;
        }
        __syncthreads();
        // ---- their bins whose cell range touches the tile, in (roi, bin) order
        const int ncombo = nroi * nbins;
        for (int k0 = 0; k0 < ncombo; k0 += 1024) {
          if (np > kPairCapBusy - 1024) drain();
          const int k = k0 + t;
          bool bh = false;
          int row = 0;
          if (k < ncombo) {
            const int i = (int)__umulhi((unsigned)k, magic_bins), bin = k - i * nbins, m = roi_list[i];
            const IBox rb = roi_box[i];
            const int ph = (int)__umulhi((unsigned)bin, magic_pw);
            const BinRange q = bin_range(rb, ph, bin - ph * PW, PH, PW, H, W);
            bh = q.he > q.hs && q.we > q.ws && q.hs <= y1 && q.he > y0 && q.ws <= x1 && q.we > x0;
            row = m * nbins + bin;
          }
          int cnt;
          const int ps = compact_wg<16>(bh, wave_count, cnt);
          if (bh) pair_list[np + ps] = row;
          np += cnt;
        }
        __syncthreads();   // roi_list is rewritten by the next chunk
      }
      drain();
      // the four copies, added in a fixed order; whole 1 KiB rows
      for (int i = t; i < kCells * 256; i += 1024) {
        const int cell = i >> 8, cc = i & 255;
        const int y = y0 + cell / kQuad, x = x0 + cell % kQuad;
        if (y > y1 || x > x1) continue;
        const float* a = acc + cell * 256 + cc;
        float* dst = out + ((size_t)y * W + x) * C + cb + cc;
        const float v = (a[0] + a[kCells * 256]) + (a[2 * kCells * 256] + a[3 * kCells * 256]);
        *dst = accumulate ? *dst + v : v;
      }
    }
  }
}

// NCHW (reference layout): one thread per output element; cell test done per thread.
__global__ __launch_bounds__(256) void moi_pool_fwd_nchw(
    const float* __restrict__ in, const float* __restrict__ rois,
    const unsigned* __restrict__ cell_bits, const unsigned* __restrict__ roi_bits,
    float* __restrict__ out, int* __restrict__ argmax, int C, int H, int W, long total, int words,
    float scale, int PH, int PW) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % PW), ph = (int)((idx / PW) % PH);
    const int c = (int)((idx / PW / PH) % C), n = (int)(idx / PW / PH / C);
    const IBox r = round_box(rois + (size_t)n * 5, scale);
    const BinRange q = bin_range(r, ph, pw, PH, PW, H, W);
    const unsigned* __restrict__ rrow = roi_bits + (size_t)n * words;
    const float* __restrict__ plane = in + ((size_t)r.b * C + c) * H * W;
    const unsigned* __restrict__ cplane = cell_bits + (size_t)r.b * H * W * words;
    float best = -FLT_MAX;
    int at = -1;
    for (int h = q.hs; h < q.he; ++h)
      for (int w = q.ws; w < q.we; ++w) {
        if (!(w >= r.x0 && w <= r.x1 && h >= r.y0 && h <= r.y1)) continue;
        const int cell = h * W + w;
        const unsigned* crow = cplane + (size_t)cell * words;
        unsigned hit = 0u;
        for (int i = 0; i < words && !hit; ++i) hit = crow[i] & rrow[i];
        if (!hit) continue;
        const float x = plane[cell];
        if (x > best) { best = x; at = cell; }
      }
    out[idx] = at == -1 ? 0.f : best;
    argmax[idx] = at;
  }
}

__global__ __launch_bounds__(256) void moi_mask_kernel(const float* __restrict__ rois,
                                                       const unsigned* __restrict__ cell_bits,
                                                       const unsigned* __restrict__ roi_bits,
                                                       int* __restrict__ mois, int H, int W,
                                                       long total, int words, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int w = (int)(idx % W), h = (int)((idx / W) % H), n = (int)(idx / W / H);
  const IBox r = round_box(rois + (size_t)n * 5, scale);
  int v = 0;
  if (w >= r.x0 && w <= r.x1 && h >= r.y0 && h <= r.y1) {
    const unsigned* crow = cell_bits + ((size_t)r.b * H * W + (size_t)h * W + w) * words;
    const unsigned* rrow = roi_bits + (size_t)n * words;
    unsigned hit = 0u;
    for (int i = 0; i < words && !hit; ++i) hit = crow[i] & rrow[i];
    v = hit != 0u;
  }
  mois[idx] = v;
}

// Backward: grad_input[b, argmax] += grad (RoIPoolBackward, MOIPool_cuda.cu:296-338).
template <bool NHWC>
__global__ __launch_bounds__(256) void moi_pool_bwd(const float* __restrict__ grad,
                                                    const float* __restrict__ rois,
                                                    const int* __restrict__ argmax,
                                                    float* __restrict__ gin, int C, int H, int W,
                                                    long total, int PH, int PW,
                                                    const int* __restrict__ roi_level, int level) {
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    int c, n;
    if (NHWC) {
      c = (int)(idx % C);
      n = (int)(idx / C / PW / PH);
    } else {
      c = (int)((idx / PW / PH) % C);
      n = (int)(idx / PW / PH / C);
    }
    if (roi_level && roi_level[n] != level) continue;  // before touching argmax/grad: each level reads only its rois
    const int a = argmax[idx];
    if (a == -1) continue;
    const int b = (int)rois[(size_t)n * 5];
    float* dst = NHWC ? gin + ((size_t)b * H * W + a) * C + c : gin + ((size_t)b * C + c) * H * W + a;
    atomicAdd(dst, grad[idx]);
  }
}

inline int bit_words(int L) { return (L + 31) / 32; }

struct Workspace { unsigned* cell; unsigned* roi; };

inline Workspace carve(void* ws, int B, int H, int W, int L) {
  Workspace k;
  k.cell = reinterpret_cast<unsigned*>(ws);
  size_t cell_bytes = (size_t)B * H * W * bit_words(L) * sizeof(unsigned);
  cell_bytes = (cell_bytes + 15) & ~(size_t)15;
  k.roi = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws) + cell_bytes);
  return k;
}

int build_bits(const int* oh_labels, const int* superpixels, const Workspace& k, int B, int H,
               int W, int M, int L, int Hs, int Ws, hipStream_t st) {
  const int words = bit_words(L);
  const long cells = (long)B * H * W;
  const bool small = moi_bits_small_ok(H, W, Hs, Ws);
  const int cpw = small ? 0 : moi_cells_per_wave(H, W, Hs, Ws);
  hipLaunchKernelGGL(moi_cell_bits_kernel, dim3(small ? ceil_div(cells, 256) : ceil_div(cells, 4 * cpw)), dim3(256),
                     4 * cpw * words * sizeof(unsigned), st, superpixels, k.cell, B, H, W, Hs, Ws, L,
                     words, cpw);
  JTSM_CHECK_LAUNCH("moi_cell_bits");
  hipLaunchKernelGGL(moi_roi_bits_kernel, dim3(ceil_div((long)M, kRoiBitsRows)), dim3(256), 0, st, oh_labels,
                     k.roi, (long)M, L, words);
  JTSM_CHECK_LAUNCH("moi_roi_bits");
  return JTSM_OK;
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

size_t jtsm_moi_pool_workspace_bytes(int B, int H, int W, int M, int L) {
  if (B < 0 || H < 0 || W < 0 || M < 0 || L < 0) return 0;
  size_t cell = (size_t)B * H * W * bit_words(L) * sizeof(unsigned);
  cell = (cell + 15) & ~(size_t)15;
  return cell + (size_t)M * bit_words(L) * sizeof(unsigned) + 16;
}

static int moi_forward_impl(const float* input, const float* rois, const int32_t* oh_labels,
                            const int32_t* superpixels, float* output, int32_t* argmax,
                            void* workspace, int B, int C, int H, int W, int M, int L, int Hs,
                            int Ws, float spatial_scale, int pooled_h, int pooled_w, int layout,
                            void* stream, const int32_t* roi_level, int level) {
  JTSM_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && M >= 0 && L >= 0 && pooled_h > 0 && pooled_w > 0,
               "moi_pool: negative size");
  JTSM_REQUIRE(layout == JTSM_NCHW || layout == JTSM_NHWC, "moi_pool: unknown layout %d", layout);
  if ((long)M * C * pooled_h * pooled_w == 0) return JTSM_OK;
  JTSM_REQUIRE(input && rois && oh_labels && superpixels && output && argmax && workspace,
               "moi_pool: null pointer");
  JTSM_REQUIRE(B > 0 && H > 0 && W > 0 && Hs > 0 && Ws > 0 && L > 0,
               "moi_pool: empty feature map / superpixel map / label table");
  JTSM_REQUIRE(((uintptr_t)workspace & 15) == 0, "moi_pool: workspace must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const Workspace k = carve(workspace, B, H, W, L);
  int rc = build_bits(oh_labels, superpixels, k, B, H, W, M, L, Hs, Ws, st);
  if (rc) return rc;
  const int words = bit_words(L);
  if (layout == JTSM_NHWC) {
    const int blocks = ceil_div((long)M * pooled_h * pooled_w, 4);
    if (moi_fwd_rows_mode() != 0 && pooled_w == 7 && moi_fwd_rows_words_ok(words) && C % 4 == 0 && ((uintptr_t)input & 15) == 0 &&
        ((uintptr_t)output & 15) == 0 && ((uintptr_t)argmax & 15) == 0 && H < 65536 && W < 65536) {
      MoiLevels lv = {};
      lv.in[0] = input; lv.cell[0] = k.cell; lv.H[0] = H; lv.W[0] = W; lv.scale[0] = spatial_scale;
      launch_fwd_rows(lv, rois, k.roi, output, argmax, C, M, words, pooled_h, roi_level, 1, roi_level ? level : -1, nullptr, st);
    } else if (C % 4 == 0 && ((uintptr_t)input & 15) == 0)
      hipLaunchKernelGGL(moi_pool_fwd_nhwc<4>, dim3(blocks), dim3(256), 0, st, input, rois, k.cell,
                         k.roi, output, argmax, C, H, W, M, words, spatial_scale, pooled_h,
                         pooled_w, roi_level, level);
    else
      hipLaunchKernelGGL(moi_pool_fwd_nhwc<1>, dim3(blocks), dim3(256), 0, st, input, rois, k.cell,
                         k.roi, output, argmax, C, H, W, M, words, spatial_scale, pooled_h,
                         pooled_w, roi_level, level);
  } else {
    JTSM_REQUIRE(!roi_level, "moi_pool: per-level filtering needs the NHWC layout");
    const long total = (long)M * C * pooled_h * pooled_w;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(moi_pool_fwd_nchw, dim3(blocks), dim3(256), 0, st, input, rois, k.cell, k.roi,
                       output, argmax, C, H, W, total, words, spatial_scale, pooled_h, pooled_w);
  }
  JTSM_CHECK_LAUNCH("moi_pool forward");
  return JTSM_OK;
}

static int moi_backward_impl(const float* grad, const float* rois, const int32_t* argmax,
                             float* grad_input, int B, int C, int H, int W, int M, int pooled_h,
                             int pooled_w, int layout, void* stream, const int32_t* roi_level, int level) {
  JTSM_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && M >= 0 && pooled_h > 0 && pooled_w > 0,
               "moi_pool backward: negative size");
  JTSM_REQUIRE(layout == JTSM_NCHW || layout == JTSM_NHWC, "moi_pool: unknown layout %d", layout);
  hipStream_t st = as_stream(stream);
  const size_t in_elems = (size_t)B * C * H * W;
  if (in_elems == 0) return JTSM_OK;
  JTSM_REQUIRE(grad_input, "moi_pool backward: null grad_input");
  JTSM_CHECK_HIP(hipMemsetAsync(grad_input, 0, in_elems * sizeof(float), st));
  const long total = (long)M * C * pooled_h * pooled_w;
  if (total == 0) return JTSM_OK;
  JTSM_REQUIRE(grad && rois && argmax, "moi_pool backward: null pointer");
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (layout == JTSM_NHWC)
    hipLaunchKernelGGL(moi_pool_bwd<true>, dim3(blocks), dim3(256), 0, st, grad, rois, argmax,
                       grad_input, C, H, W, total, pooled_h, pooled_w, roi_level, level);
  else
    hipLaunchKernelGGL(moi_pool_bwd<false>, dim3(blocks), dim3(256), 0, st, grad, rois, argmax,
                       grad_input, C, H, W, total, pooled_h, pooled_w, roi_level, level);
  JTSM_CHECK_LAUNCH("moi_pool backward");
  return JTSM_OK;
}

int jtsm_moi_pool_forward_f32(const float* input, const float* rois, const int32_t* oh_labels,
                              const int32_t* superpixels, float* output, int32_t* argmax,
                              void* workspace, int B, int C, int H, int W, int M, int L, int Hs,
                              int Ws, float spatial_scale, int pooled_h, int pooled_w, int layout,
                              void* stream) {
  return moi_forward_impl(input, rois, oh_labels, superpixels, output, argmax, workspace, B, C, H, W, M, L,
                          Hs, Ws, spatial_scale, pooled_h, pooled_w, layout, stream, nullptr, 0);
}
int jtsm_moi_pool_backward_f32(const float* grad, const float* rois, const int32_t* argmax,
                               float* grad_input, int B, int C, int H, int W, int M, int pooled_h,
                               int pooled_w, int layout, void* stream) {
  return moi_backward_impl(grad, rois, argmax, grad_input, B, C, H, W, M, pooled_h, pooled_w, layout,
                           stream, nullptr, 0);
}
int jtsm_moi_pool_forward_level_f32(const float* input, const float* rois, const int32_t* roi_level,
                                    int level, const int32_t* oh_labels, const int32_t* superpixels,
                                    float* output, int32_t* argmax, void* workspace, int B, int C, int H,
                                    int W, int M, int L, int Hs, int Ws, float spatial_scale, int pooled_h,
                                    int pooled_w, void* stream) {
  JTSM_REQUIRE(roi_level || M == 0, "moi_pool level: null roi_level");
  return moi_forward_impl(input, rois, oh_labels, superpixels, output, argmax, workspace, B, C, H, W, M, L,
                          Hs, Ws, spatial_scale, pooled_h, pooled_w, JTSM_NHWC, stream, roi_level, level);
}
int jtsm_moi_pool_backward_level_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                     int level, const int32_t* argmax, float* grad_input, int B, int C,
                                     int H, int W, int M, int pooled_h, int pooled_w, void* stream) {
  JTSM_REQUIRE(roi_level || M == 0, "moi_pool level: null roi_level");
  return moi_backward_impl(grad, rois, argmax, grad_input, B, C, H, W, M, pooled_h, pooled_w, JTSM_NHWC,
                           stream, roi_level, level);
}

int jtsm_moi_mask_f32(const float* rois, const int32_t* oh_labels, const int32_t* superpixels,
                      int32_t* mois, void* workspace, int B, int H, int W, int M, int L, int Hs,
                      int Ws, float spatial_scale, void* stream) {
  JTSM_REQUIRE(B > 0 && H > 0 && W > 0 && M >= 0 && L > 0 && Hs > 0 && Ws > 0, "moi_mask: bad sizes");
  if (M == 0) return JTSM_OK;
  JTSM_REQUIRE(rois && oh_labels && superpixels && mois && workspace, "moi_mask: null pointer");
  hipStream_t st = as_stream(stream);
  const Workspace k = carve(workspace, B, H, W, L);
  int rc = build_bits(oh_labels, superpixels, k, B, H, W, M, L, Hs, Ws, st);
  if (rc) return rc;
  const long total = (long)M * H * W;
  hipLaunchKernelGGL(moi_mask_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, st, rois, k.cell,
                     k.roi, mois, H, W, total, bit_words(L), spatial_scale);
  JTSM_CHECK_LAUNCH("moi_mask");
  return JTSM_OK;
}

/* All FPN levels in one launch (what ROIPooler does for MOIPool, detectron2/modeling/poolers.py:193-250 with
 * wsl/layers/moi_pool.py): inputs[l] / grad_inputs[l] are (B,H[l],W[l],C) NHWC maps, roi_level[n] in
 * [0,nlevels) picks the map of roi n.  workspace: jtsm_moi_pool_levels_workspace_bytes. */
size_t jtsm_moi_pool_levels_workspace_bytes(int B, const int* H, const int* W, int nlevels, int M, int L) {
  if (B < 0 || M < 0 || L < 0 || nlevels < 0 || nlevels > kMaxLevels || !H || !W) return 0;
  size_t total = 16 + (((size_t)M * bit_words(L) * sizeof(unsigned) + 15) & ~(size_t)15);
  for (int l = 0; l < nlevels; ++l)
    total += (((size_t)B * H[l] * W[l] * bit_words(L) * sizeof(unsigned)) + 15) & ~(size_t)15;
  total += ((size_t)M * sizeof(int) + 15) & ~(size_t)15;      // the rois' spatial order
  return total;
}

int jtsm_moi_pool_forward_levels_f32(const float* const* inputs, const int* H, const int* W, const float* scales,
                                     int nlevels, const float* rois, const int32_t* roi_level,
                                     const int32_t* oh_labels, const int32_t* superpixels, float* output,
                                     int32_t* argmax, void* workspace, int B, int C, int M, int L, int Hs, int Ws,
                                     int pooled_h, int pooled_w, void* stream) {
  JTSM_REQUIRE(nlevels > 0 && nlevels <= kMaxLevels && inputs && H && W && scales, "moi_pool levels: bad level table");
  JTSM_REQUIRE(B >= 0 && C >= 0 && M >= 0 && L >= 0 && pooled_h > 0 && pooled_w > 0, "moi_pool levels: negative size");
  if ((long)M * C == 0) return JTSM_OK;
  JTSM_REQUIRE(rois && roi_level && oh_labels && superpixels && output && argmax && workspace,
               "moi_pool levels: null pointer");
  JTSM_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && L > 0 && C % 4 == 0, "moi_pool levels: empty map / C %% 4 != 0");
  JTSM_REQUIRE(((uintptr_t)workspace & 15) == 0, "moi_pool levels: workspace must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  const int words = bit_words(L);
  char* w = reinterpret_cast<char*>(workspace);
  unsigned* roi_bits = reinterpret_cast<unsigned*>(w);
  w += (((size_t)M * words * sizeof(unsigned)) + 15) & ~(size_t)15;
  MoiLevels lv = {};
  MoiBitsPlan bp = {};
  int nblk = 0, max_cpw = 1;
  for (int l = 0; l < nlevels; ++l) {
    JTSM_REQUIRE(inputs[l] && H[l] > 0 && W[l] > 0 && ((uintptr_t)inputs[l] & 15) == 0, "moi_pool levels: bad level %d", l);
    lv.in[l] = inputs[l]; lv.H[l] = H[l]; lv.W[l] = W[l]; lv.scale[l] = scales[l];
    unsigned* cell = reinterpret_cast<unsigned*>(w);
    lv.cell[l] = cell;
    w += (((size_t)B * H[l] * W[l] * words * sizeof(unsigned)) + 15) & ~(size_t)15;
    const long cells = (long)B * H[l] * W[l];
    const bool small = moi_bits_small_ok(H[l], W[l], Hs, Ws);
    const int cpw = small ? 0 : moi_cells_per_wave(H[l], W[l], Hs, Ws);
    bp.cell[l] = cell; bp.H[l] = H[l]; bp.W[l] = W[l]; bp.cpw[l] = cpw; bp.first[l] = nblk;
    nblk += (int)(small ? ceil_div(cells, 256) : ceil_div(cells, 4 * cpw));
    max_cpw = std::max(max_cpw, cpw);
  }
  bp.n = nlevels;
  bp.first[nlevels] = nblk;
  nblk += (int)ceil_div((long)M, kRoiBitsRows);
  bool rows_ok = moi_fwd_rows_mode() != 0 && pooled_w == 7 && moi_fwd_rows_words_ok(words) && ((uintptr_t)output & 15) == 0 &&
                 ((uintptr_t)argmax & 15) == 0;
  for (int l = 0; l < nlevels; ++l) rows_ok = rows_ok && H[l] < 65536 && W[l] < 65536;
  // the rois' spatial order (one more block of the same launch): the coarsest grid whose buckets fit the block's table
  int* order = nullptr;
  bp.sort_block = -1;
  if (rows_ok && moi_sort_on()) {
    int G = 16;
    while (G > 1 && (long)nlevels * B * G * G > kSortBuckets) G >>= 1;
    if ((long)nlevels * B * G * G <= kSortBuckets) {
      order = reinterpret_cast<int*>(w);
      bp.sort_block = nblk++;
      bp.grid = G;
      bp.scale0 = scales[0];
    }
  }
  hipLaunchKernelGGL(moi_bits_all_kernel, dim3(nblk), dim3(256), 4 * max_cpw * words * sizeof(unsigned), st, bp, superpixels,
                     oh_labels, roi_bits, B, Hs, Ws, L, words, (long)M, rois, roi_level, M, order, moi_bits_skip());
  if (rows_ok) {
    launch_fwd_rows(lv, rois, roi_bits, output, argmax, C, M, words, pooled_h, roi_level, nlevels, -1, order, st);
  } else {
    const int blocks = ceil_div((long)M * pooled_h * pooled_w, 4);
    hipLaunchKernelGGL(moi_pool_fwd_levels<4>, dim3(blocks), dim3(256), 0, st, lv, rois, roi_bits, output, argmax, C, M,
                       words, pooled_h, pooled_w, roi_level, nlevels);
  }
  JTSM_CHECK_LAUNCH("moi_pool forward levels");
  return JTSM_OK;
}

// The side stream and the fork / join events of the backward's two gathers, created once per device (JTSM_MOI_BWD_STREAMS=0:
// one stream, the gathers one after the other).
static hipStream_t moi_bwd_side_stream() {
  static const bool on = [] { const char* e = getenv("JTSM_MOI_BWD_STREAMS"); return !e || atoi(e) != 0; }();
  if (!on) return nullptr;
  static hipStream_t streams[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!streams[dev] && hipStreamCreateWithFlags(&streams[dev], hipStreamNonBlocking) != hipSuccess) streams[dev] = nullptr;
  return streams[dev];
}
static bool moi_bwd_events(hipEvent_t* fork, hipEvent_t* join) {
  static hipEvent_t ev[16][2] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  for (int k = 0; k < 2; ++k)
    if (!ev[dev][k] && hipEventCreateWithFlags(&ev[dev][k], hipEventDisableTiming) != hipSuccess) return false;
  *fork = ev[dev][0];
  *join = ev[dev][1];
  return true;
}

static long census_tiles(const int* H, const int* W, int nlevels, int B) {
  long n = 0;
  for (int l = 0; l < nlevels; ++l) n += (long)B * ceil_div(W[l], kTile) * ceil_div(H[l], kTileY);
  return n;
}

size_t jtsm_moi_pool_backward_levels_workspace_bytes(const int* H, const int* W, int nlevels, int B, int M) {
  if (nlevels <= 0 || nlevels > kMaxLevels || B <= 0 || M <= 0 || !H || !W) return 0;
  // per-(level, image) roi lists + counts, then the tile census and the launch plan (maximum, count, busy tiles)
  // (... and the light-tile gather's workgroup order: one more int per tile)
  return (((size_t)nlevels * B * ((size_t)M + 1) + 6 * (size_t)census_tiles(H, W, nlevels, B) + 8) * sizeof(int) + 15) &
         ~(size_t)15;
}

int jtsm_moi_pool_backward_levels_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                      const int32_t* argmax, float* const* grad_inputs, const int* H, const int* W,
                                      const float* scales, int nlevels, int B, int C, int M, int pooled_h, int pooled_w,
                                      int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(nlevels > 0 && nlevels <= kMaxLevels && grad_inputs && H && W, "moi_pool levels backward: bad level table");
  JTSM_REQUIRE(B >= 0 && C >= 0 && M >= 0 && pooled_h > 0 && pooled_w > 0 && C % 4 == 0,
               "moi_pool levels backward: bad sizes (C %% 4 must be 0)");
  hipStream_t st = as_stream(stream);
  MoiLevels lv = {};
  bool all = true;
  for (int l = 0; l < nlevels; ++l) {
    lv.gin[l] = grad_inputs[l]; lv.H[l] = H[l]; lv.W[l] = W[l];
    lv.scale[l] = scales ? scales[l] : 0.f;
    all = all && grad_inputs[l] != nullptr;
  }
  // gather form: needs the level scales (bin geometry), 256-channel blocks, bins that fit one wavefront
  bool small_maps = true;   // argmax -> (y, x) by multiply-high needs cell * W < 2^32
  for (int l = 0; l < nlevels; ++l) small_maps = small_maps && (unsigned long long)H[l] * W[l] * W[l] < 0xFFFFFFFFull;
  const bool tiled = scales && workspace && all && small_maps && C % 256 == 0 && M > 0 && B > 0;
  if (tiled) {
    JTSM_REQUIRE(grad && rois && roi_level && argmax, "moi_pool levels backward: null pointer");
    const size_t need = jtsm_moi_pool_backward_levels_workspace_bytes(H, W, nlevels, B, M);
    JTSM_REQUIRE(workspace_bytes >= need && ((uintptr_t)workspace & 3) == 0,
                 "moi_pool levels backward: workspace of %zu bytes needed", need);
    int* lists = reinterpret_cast<int*>(workspace);
    int* counts = lists + (size_t)nlevels * B * M;
    int* census = counts + (size_t)nlevels * B;
    const int ntile = (int)census_tiles(H, W, nlevels, B);
    int* plan = census + ntile;            // [0] census maximum, [1] number of busy tiles, [2 ...] their indices
    hipLaunchKernelGGL(moi_roi_lists_kernel, dim3(nlevels * B), dim3(256), 0, st, rois, roi_level, M, B, lists, counts);
    MoiTiles tl = {};
    int blocks = 0;
    for (int l = 0; l < nlevels; ++l) {
      tl.first[l] = blocks;
      tl.tiles_x[l] = ceil_div(W[l], kTile);
      tl.tiles_y[l] = ceil_div(H[l], kTileY);
      blocks += B * tl.tiles_x[l] * tl.tiles_y[l];
    }
    tl.first[nlevels] = blocks;
    JTSM_CHECK_HIP(hipMemsetAsync(census, 0, (size_t)ntile * sizeof(int), st));
    hipLaunchKernelGGL(moi_tile_census_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, st, lv, tl, rois, roi_level, M, nlevels,
                       pooled_h, pooled_w, census);
    // the plan: the census maximum and the heavy tiles (4 quadrant entries each), heaviest first
    int* tile_order = plan + 2 + 4 * (size_t)ntile;   // (behind the busy list's at most 4 entries per tile)
    hipLaunchKernelGGL(tile_plan_kernel, dim3(1), dim3(1024), 0, st, census, ntile, plan, moi_heavy_min(), 4, tile_order);
    // The two gathers write disjoint cells and each ends in a tail of a few long workgroups (the light form's
    // heaviest tiles; the piled quadrants): side by side on two streams the tails overlap.  Fork / join by events on a
    // side stream the library keeps per device (no host synchronisation; the caller's stream waits for the join).
    hipStream_t side = moi_bwd_side_stream();
    hipEvent_t fork = nullptr, join = nullptr;
    if (side && moi_bwd_events(&fork, &join)) {
      JTSM_CHECK_HIP(hipEventRecord(fork, st));
      JTSM_CHECK_HIP(hipStreamWaitEvent(side, fork, 0));
    } else {
      side = st;
    }
    hipLaunchKernelGGL(moi_pool_bwd_busy, dim3(256), dim3(1024), (size_t)kCopies * kQuad * kQuad * 256 * sizeof(float), side,
                       lv, tl, grad, rois, argmax, C, M, pooled_h, pooled_w, B, lists, counts, nlevels, plan, accumulate);
    hipLaunchKernelGGL(moi_pool_bwd_tiled, dim3(blocks, C / 256), dim3(256), kTile * kTile * 256 * sizeof(float), st, lv, tl,
                       grad, rois, argmax, C, M, pooled_h, pooled_w, B, lists, counts, nlevels, plan, census, moi_heavy_min(),
                       accumulate, tile_order);
    if (side != st) {
      JTSM_CHECK_HIP(hipEventRecord(join, side));
      JTSM_CHECK_HIP(hipStreamWaitEvent(st, join, 0));
    }
    // (returns at once unless the census sent the gather home)
    hipLaunchKernelGGL(moi_pool_bwd_levels, dim3(std::min(ceil_div((long)M * pooled_h * pooled_w, 4), 8192)), dim3(256), 0,
                       st, lv, grad, rois, argmax, C, M, pooled_h * pooled_w, roi_level, nlevels, plan, kCensusLimit);
    JTSM_CHECK_LAUNCH("moi_pool backward levels (tiled)");
    return JTSM_OK;
  }
  for (int l = 0; l < nlevels; ++l) {
    if (!grad_inputs[l] || accumulate) continue;   // a level whose gradient is not wanted / maps that hold a gradient already
    JTSM_CHECK_HIP(hipMemsetAsync(grad_inputs[l], 0, (size_t)B * H[l] * W[l] * C * sizeof(float), st));
  }
  if ((long)M * C == 0) return JTSM_OK;
  JTSM_REQUIRE(grad && rois && roi_level && argmax, "moi_pool levels backward: null pointer");
  for (int l = 0; l < nlevels; ++l)
    JTSM_REQUIRE(grad_inputs[l], "moi_pool levels backward: every level needs a gradient buffer");
  const int nbins = pooled_h * pooled_w;
  hipLaunchKernelGGL(moi_pool_bwd_levels, dim3(std::min(ceil_div((long)M * nbins, 4), 8192)), dim3(256), 0, st, lv, grad,
                     rois, argmax, C, M, nbins, roi_level, nlevels, (const int*)nullptr, 0);
  JTSM_CHECK_LAUNCH("moi_pool backward levels");
  return JTSM_OK;
}

}  // extern "C"
