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
#include <cstdlib>
#include <type_traits>

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
  // (readfirstlane: the wavefront index is uniform, but the compiler cannot see that through threadIdx — with it
  // the roi / bin / address arithmetic below runs on the scalar unit instead of once per lane)
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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

// The maps of the call: one level (jtsm_roi_align_backward_level_f32 / the plain entry) or all FPN levels at once
// (jtsm_roi_align_backward_levels_f32) — tiles of every level are workgroups of ONE launch, so the levels' critical
// paths (the busiest tile of each) overlap instead of adding up.
constexpr int kAlignLevels = 8;
struct AlignLevels {
  float* gin[kAlignLevels];
  int H[kAlignLevels], W[kAlignLevels];
  float scale[kAlignLevels];
  int tiles_x[kAlignLevels], tiles_y[kAlignLevels];
  int first_tile[kAlignLevels + 1];   // running sum of B * tiles_x * tiles_y
  int level_id[kAlignLevels];         // the roi_level value this entry serves
  int n;
};

// ---------------------------------------------------------------- NHWC backward
template <typename T, int VEC, bool ROT>
__global__ __launch_bounds__(256) void align_bwd_nhwc(const T* __restrict__ grad,
                                                      const T* __restrict__ rois,
                                                      T* __restrict__ gin, int C, int H, int W,
                                                      int M, T scale, int PH, int PW, int sr,
                                                      int aligned,
                                                      const int* __restrict__ roi_level, int level,
                                                      const int* __restrict__ census_max, int census_limit) {
#pragma clang fp contract(off)
  if (census_max && *census_max <= census_limit) return;   // the gather form (align_bwd_tiled) took this call
  const int lane = threadIdx.x & 63;
  const int nbins = PH * PW;
  // (readfirstlane: the wavefront index is uniform, but the compiler cannot see that through threadIdx — with it
  // the roi / bin / address arithmetic below runs on the scalar unit instead of once per lane)
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
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

// The scatter form over a level table in ONE launch (the census fallback of the gather form below: before, one launch
// per level, four ~5 us launches that return at once in the usual case): every (roi, bin) wavefront adds into the map
// of its own level.
template <int VEC>
__global__ __launch_bounds__(256) void align_bwd_nhwc_levels(const float* __restrict__ grad, const float* __restrict__ rois,
                                                             const AlignLevels lv, int C, int M, int PH, int PW, int sr,
                                                             int aligned, const int* __restrict__ roi_level,
                                                             const int* __restrict__ census_max, int census_limit) {
#pragma clang fp contract(off)
  if (census_max && *census_max <= census_limit) return;   // the gather form took this call
  const int lane = threadIdx.x & 63;
  const int nbins = PH * PW;
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (wave >= (long)M * nbins) return;
  const int n = (int)(wave / nbins);
  int l = 0;
  if (roi_level) {
    const int id = roi_level[n];
    for (l = 0; l < lv.n && lv.level_id[l] != id; ++l) {}
    if (l == lv.n) return;
  }
  const int H = lv.H[l], W = lv.W[l];
  const int bin = (int)(wave - (long)n * nbins);
  const int ph = bin / PW, pw = bin - ph * PW;
  const RoiGeom<float> g = roi_geometry<float, false>(rois, n, lv.scale[l], PH, PW, sr, aligned != 0);
  const int S = (g.gh > 0 && g.gw > 0) ? g.gh * g.gw : 0;
  const float count = (float)(g.gh * g.gw);
  float* __restrict__ plane = lv.gin[l] + (size_t)g.b * H * W * C;
  const float* __restrict__ grow = grad + ((size_t)n * nbins + bin) * C;
  for (int cb = 0; cb < C; cb += 64 * VEC) {
    const int c = cb + lane * VEC;
    const bool live = c < C;
    float go[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) go[v] = 0.f;
    if (live) load_vec<float, VEC>(grow + c, go);
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      Tap<float> mine;
      if (s < S) {
        const int iy = s / g.gw;
        mine = sample_tap<float, false>(g, H, W, ph, pw, iy, s - iy * g.gw);
      } else {
        mine.pos[0] = mine.pos[1] = mine.pos[2] = mine.pos[3] = -1;
        mine.w[0] = mine.w[1] = mine.w[2] = mine.w[3] = 0.f;
      }
      const int cnt = (S - s0) < 64 ? (S - s0) : 64;
      for (int j = 0; j < cnt; ++j) {
        const int p0 = __shfl(mine.pos[0], j);
        if (p0 < 0) continue;
        int p[4] = {p0, __shfl(mine.pos[1], j), __shfl(mine.pos[2], j), __shfl(mine.pos[3], j)};
        float w[4] = {__shfl(mine.w[0], j), __shfl(mine.w[1], j), __shfl(mine.w[2], j), __shfl(mine.w[3], j)};
        if (live) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* dst = plane + (size_t)p[q] * C + c;
#pragma unroll
            for (int v = 0; v < VEC; ++v) atomicAdd(dst + v, go[v] * w[q] / count);
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------- NHWC backward as a gather
// A workgroup owns an 8 x 8-cell tile of the gradient map x 64 channels, lists — from the roi geometry alone, in
// (roi, bin) order — the bins whose bilinear taps can reach the tile, and adds their contributions in LDS.  The bin's
// samples form a gh x gw grid and a sample's four weights are (hy | ly) x (hx | lx), so the sum over the grid
// factorises into eight row weights times eight column weights of the tile (make_tap's clamping and out-of-range
// rules, roi_geometry.h, are per axis too).  The four wavefronts take the listed bins round-robin, each into its OWN
// 64-cell x 64-channel accumulator (lane = channel), and the four are summed in a fixed order at the end: no
// atomics, reproducible bit for bit, every map cell written exactly once (no zero fill).  2-5x faster than the scatter
// form above on spread-out rois; a tile under a pile of rois is walked by one workgroup, so a census of the boxes
// (estimated bins per tile) sends such calls to the scatter form instead (as csrc/moi_pool.hip does).
constexpr int kTile = 8;
// (estimated) bins on one tile above which the scatter form takes the call; overridable for sweeps
static int census_limit() {
  static const int v = [] { const char* e = getenv("JTSM_ALIGN_CENSUS_LIMIT"); return e ? atoi(e) : 200000; }();
  return v;
}


__global__ __launch_bounds__(256) void align_census_kernel(const float* __restrict__ rois, int M, int PH, int PW, int sr,
                                                           int aligned, const int* __restrict__ roi_level,
                                                           const AlignLevels lv, int* __restrict__ census,
                                                           int4* __restrict__ reach_out, int* __restrict__ meta_out) {
#pragma clang fp contract(off)
  // reach_out[n] = (ya, yz, xa, xz): the cells roi n's samples can touch; meta_out[n] = image | level entry << 16, or
  // -1 (no level of this call / degenerate box).  The tile workgroups test these five integers instead of redoing the
  // box geometry for every roi in every tile (a fifth of the gather's vector instructions were that).
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= M) return;
  meta_out[n] = -1;
  int l = 0;
  if (roi_level) {
    const int id = roi_level[n];
    for (l = 0; l < lv.n && lv.level_id[l] != id; ++l) {}
    if (l == lv.n) return;
  }
  const int H = lv.H[l], W = lv.W[l], tiles_x = lv.tiles_x[l], tiles_y = lv.tiles_y[l];
  census += lv.first_tile[l];
  const RoiGeom<float> g = geom_box<float>(rois + (size_t)n * 5, lv.scale[l], PH, PW, sr, aligned != 0);
  if (g.gh <= 0 || g.gw <= 0) return;
  const int xa = max((int)floorf(g.x0), 0), xz = min((int)floorf(g.x0 + (float)PW * g.bw) + 1, W - 1);
  const int ya = max((int)floorf(g.y0), 0), yz = min((int)floorf(g.y0 + (float)PH * g.bh) + 1, H - 1);
  reach_out[n] = make_int4(ya, yz, xa, xz);
  meta_out[n] = g.b | (l << 16);
  if (xa > xz || ya > yz) return;
  for (int ty = ya / kTile; ty <= yz / kTile; ++ty) {
    const int oy = min(yz, ty * kTile + kTile - 1) - max(ya, ty * kTile) + 1;
    const int by = min(PH, (int)((float)oy / fmaxf(g.bh, 1e-6f)) + 2);
    for (int tx = xa / kTile; tx <= xz / kTile; ++tx) {
      const int ox = min(xz, tx * kTile + kTile - 1) - max(xa, tx * kTile) + 1;
      const int bx = min(PW, (int)((float)ox / fmaxf(g.bw, 1e-6f)) + 2);
      atomicAdd(&census[(g.b * tiles_y + ty) * tiles_x + tx], by * bx);
    }
  }
}

constexpr int kSub = 32;        // listed rois whose tile weights are staged in LDS at a time
constexpr int kMaxBins = 16;    // PH, PW <= 16 (7 and 14 on the JTSM path); wider poolers use the scatter form
constexpr int kSubsets = 4;     // wavefronts that share a channel block's work items (register copies, added at the end)

// Gather form of the backward, round 3.  A fixed grid of workgroups (4 wavefronts each) strides the plan's list of
// (tile, 64-channel block) jobs — every 8 x 8-cell tile some roi can reach, heaviest first (tile_plan_kernel); the maps
// are cleared by a memset beforehand, so tiles no roi reaches (most of a call) cost nothing but that.
// Wavefront w of a job takes the work items w, w + 4, ... (lane = channel): the rois that can reach the tile are listed
// (256 at a time); for 32 of them at a time the threads compute, one (roi, axis, bin) each, the bilinear weights of
// every bin that reaches the tile on the tile's lines into LDS (the sample loops run once per workgroup), and the
// staged rois' reaching bin ROWS become one flat list of work items.  An item's contribution g * wy[r] * wx[q] is
// applied as v[q] = sum_pw g wx_pw[q], acc[r][q] += wy_ph[r] v[q] — weights from LDS as broadcast reads, the NEXT
// item's gradients (whichever roi it belongs to) already in flight.  The four register copies are added in a fixed order
// through LDS at the end, (a0 + a1) + (a2 + a3): no atomics, reproducible bit for bit, every cell of a listed tile
// written exactly once.
//
// What round 2's profile showed (rocprofv3 counters on the bench's ~300 foreground rois, 6 piles of ~50): the launch
// averaged 3.6 resident wavefronts per CU.  It was (a) ~100 us of floor — 10 880 workgroups of 215 registers and 64 KiB
// of LDS, almost all of them there to find no roi and write zeros — and (b) a tail: a tile under a pile was ONE
// workgroup walking roi after roi, 2-3 dependent load round trips each.  Measured and NOT kept: a 16-wavefront form for
// the tiles under piles (4 x 4-cell quadrants x 256 channels, items over four wavefronts per channel block): the same
// time as this form on the piled workload — its per-job listing and staging cost what the wider walk saved.
template <int TILE, int NBLK>
__global__ __launch_bounds__(256 * NBLK) void align_bwd_gather(
    const float* __restrict__ grad, const float* __restrict__ rois, const AlignLevels lv, int C, int M, int PH, int PW,
    int sr, int aligned, const int* __restrict__ plan, int census_lim, int fan, const int4* __restrict__ reach_in,
    const int* __restrict__ meta_in, int accumulate) {
#pragma clang fp contract(off)
  __shared__ int roi_list[256];
  __shared__ float roi_row[256][5];
  constexpr int NW = 4 * NBLK, NT = 256 * NBLK, NCH = 64 * NBLK;
  __shared__ int wave_count[NW];
  __shared__ __attribute__((aligned(16))) float wts[kSub][2][kMaxBins][TILE];     // [roi][axis y|x][slot][line]
  __shared__ __attribute__((aligned(16))) float copies[2][TILE * TILE][NCH];       // 32 KiB: two slots of the final sums
  __shared__ int first_bin[kSub][2], nbin[kSub][2];
  __shared__ float inv_cnt[kSub];
  __shared__ unsigned char row_lo[kSub][kMaxBins], row_hi[kSub][kMaxBins];   // tile lines with a non-zero y weight
  __shared__ int item_base[kSub + 1];
  __shared__ unsigned short items[kSub * kMaxBins];                         // roi slot << 8 | bin-row slot
  if (plan[0] > census_lim) return;   // piled-up rois beyond what one workgroup should walk: the scatter form
  const int t = threadIdx.x, lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cblk = wv % NBLK, subset = wv / NBLK;
  const int nbins = PH * PW;
  const int njobs = plan[1];

  // cells a sample coordinate range [lo, hi] can touch: floor(lo) .. floor(hi) + 1, after the clamp to the map
  auto reach = [](float lo, float hi, int n, int& a, int& z) {
    a = max((int)floorf(lo), 0);
    z = min((int)floorf(hi) + 1, n - 1);
  };

  for (int job = blockIdx.x; job < njobs; job += gridDim.x) {
    const int entry = plan[2 + job], tile = entry / fan, group = entry - tile * fan;   // (tile, channel group)
    int lvl = 0;
    while (lvl + 1 < lv.n && tile >= lv.first_tile[lvl + 1]) ++lvl;
    const int H = lv.H[lvl], W = lv.W[lvl], tiles_x = lv.tiles_x[lvl], tiles_y = lv.tiles_y[lvl];
    const float scale = lv.scale[lvl];
    int rel = tile - lv.first_tile[lvl];
    const int tx = rel % tiles_x; rel /= tiles_x;
    const int ty = rel % tiles_y;
    const int b = rel / tiles_y;
    const int x0 = tx * TILE, y0 = ty * TILE, x1 = min(x0 + TILE, W) - 1, y1 = min(y0 + TILE, H) - 1;
    float* __restrict__ out = lv.gin[lvl] + (size_t)b * H * W * C;
    {
      const int cb0 = group * NCH;
      const int nblk = min(NBLK, (C - cb0) >> 6);   // channel blocks present: wavefronts beyond them only keep step
      const bool serving = cblk < nblk;
      const int c = cb0 + cblk * 64 + lane;
      float acc[TILE][TILE];
#pragma unroll
      for (int r = 0; r < TILE; ++r)
#pragma unroll
        for (int q = 0; q < TILE; ++q) acc[r][q] = 0.f;

      for (int base = 0; base < M; base += 256) {
        const int n = base + t;
        bool hit = false;
        if (t < 256 && n < M && meta_in[n] == (b | (lvl << 16))) {   // this level, this image, a box with samples
          const int4 rc = reach_in[n];                               // (ya, yz, xa, xz)
          hit = rc.x <= y1 && rc.y >= y0 && rc.z <= x1 && rc.w >= x0;
        }
        int nroi;
        const int slot = compact_wg<NW>(hit, wave_count, nroi);
        if (hit) {
          roi_list[slot] = n;
#pragma unroll
          for (int q = 0; q < 5; ++q) roi_row[slot][q] = rois[(size_t)n * 5 + q];
        }
        __syncthreads();
        for (int sub = 0; sub < nroi; sub += kSub) {
          const int ns = min(kSub, nroi - sub);
          // ---- stage: bins of each axis that reach the tile, and their weights on the tile's lines
          if (t < ns * 2) {   // one thread per (roi, axis): the contiguous range of reaching bins
            const int i = t >> 1, axis = t & 1;
            const RoiGeom<float> g = geom_box<float>(roi_row[sub + i], scale, PH, PW, sr, aligned != 0);
            const float origin = axis ? g.x0 : g.y0, bin = axis ? g.bw : g.bh;
            const int P = axis ? PW : PH, n_lines = axis ? W : H, t0 = axis ? x0 : y0, t1 = axis ? x1 : y1;
            int lo = max(0, (int)floorf(((float)t0 - 1.f - origin) / fmaxf(bin, 1e-6f)) - 1);
            int hi = min(P - 1, (int)floorf(((float)t1 + 1.f - origin) / fmaxf(bin, 1e-6f)) + 1);
            auto reaches = [&](int p) {
              int a, z;
              reach(origin + (float)p * bin, origin + (float)(p + 1) * bin, n_lines, a, z);
              return a <= t1 && z >= t0;
            };
            while (lo <= hi && !reaches(lo)) ++lo;
            while (hi >= lo && !reaches(hi)) --hi;
            first_bin[i][axis] = lo;
            nbin[i][axis] = max(hi - lo + 1, 0);
            if (axis == 0) inv_cnt[i] = 1.f / (float)(g.gh * g.gw);
          }
          __syncthreads();
          for (int task = t; task < ns * 2 * kMaxBins; task += NT) {   // one (roi, axis, bin slot) per thread
            const int i = task / (2 * kMaxBins), axis = (task / kMaxBins) & 1, sl = task % kMaxBins;
            if (sl >= nbin[i][axis]) continue;
            const RoiGeom<float> g = geom_box<float>(roi_row[sub + i], scale, PH, PW, sr, aligned != 0);
            const float origin = axis ? g.x0 : g.y0, bin = axis ? g.bw : g.bh;
            const int grid = axis ? g.gw : g.gh, n_lines = axis ? W : H, t0 = axis ? x0 : y0;
            const int p = first_bin[i][axis] + sl;
            float w[TILE];
#pragma unroll
            for (int r = 0; r < TILE; ++r) w[r] = 0.f;
            for (int k = 0; k < grid; ++k) {
              float y = origin + (float)p * bin + (float)((float)k + .5f) * bin / (float)grid;
              if (y < -1.0f || y > (float)n_lines) continue;
              if (y <= 0.f) y = 0.f;
              int yl = (int)y, yh;
              if (yl >= n_lines - 1) { yh = yl = n_lines - 1; y = (float)yl; } else { yh = yl + 1; }
              const float ly = y - (float)yl, hy = 1.f - ly;
#pragma unroll
              for (int r = 0; r < TILE; ++r) w[r] += (yl == t0 + r ? hy : 0.f) + (yh == t0 + r ? ly : 0.f);
            }
            int lo = TILE, hi = -1;
#pragma unroll
            for (int r = 0; r < TILE; ++r) {
              wts[i][axis][sl][r] = w[r];
              if (w[r] != 0.f) { lo = min(lo, r); hi = r; }
            }
            if (axis == 0) { row_lo[i][sl] = (unsigned char)lo; row_hi[i][sl] = (unsigned char)(hi + 1); }   // [lo, hi)
          }
          // ---- the staged rois' reaching bin rows as one item list, in (roi, row) order
          if (t < 64) {   // (first wavefront: an exclusive scan of the rois' row counts)
            const int cnt = (t < ns && nbin[t][1] > 0) ? nbin[t][0] : 0;
            int inc = cnt;
#pragma unroll
            for (int o = 1; o < kSub; o <<= 1) {
              const int up = __shfl_up(inc, o);
              if (t >= o) inc += up;
            }
            if (t < ns) item_base[t] = inc - cnt;
            if (t == ns - 1) item_base[ns] = inc;
          }
          __syncthreads();
          for (int task = t; task < ns * kMaxBins; task += NT) {
            const int i = task / kMaxBins, a = task % kMaxBins;
            if (nbin[i][1] > 0 && a < nbin[i][0]) items[item_base[i] + a] = (unsigned short)(i << 8 | a);
          }
          __syncthreads();
          // ---- walk: this wavefront takes items subset, subset + 4, ... for its channel block; the next item's
          // gradient row is requested before this one is applied
          {
            const int nitems = serving ? item_base[ns] : 0;
            float gk[kMaxBins], gnext[kMaxBins];
            auto request = [&](int j, float (&g)[kMaxBins]) {
              const int it = __builtin_amdgcn_readfirstlane((int)items[j]);
              const int i = it >> 8, a = it & 255;
              const int npw = __builtin_amdgcn_readfirstlane(nbin[i][1]);
              const int ph0 = __builtin_amdgcn_readfirstlane(first_bin[i][0]), pw0 = __builtin_amdgcn_readfirstlane(first_bin[i][1]);
              const int m = __builtin_amdgcn_readfirstlane(roi_list[sub + i]);
              const float* __restrict__ grow = grad + ((size_t)m * nbins + (size_t)(ph0 + a) * PW + pw0) * C + c;
#pragma unroll
              for (int k = 0; k < kMaxBins; ++k) g[k] = k < npw ? grow[(size_t)k * C] : 0.f;
            };
            if (subset < nitems) request(subset, gnext);
            for (int j = subset; j < nitems; j += kSubsets) {
#pragma unroll
              for (int k = 0; k < kMaxBins; ++k) gk[k] = gnext[k];
              if (j + kSubsets < nitems) request(j + kSubsets, gnext);
              const int it = __builtin_amdgcn_readfirstlane((int)items[j]);
              const int i = it >> 8, a = it & 255;
              const int npw = __builtin_amdgcn_readfirstlane(nbin[i][1]);
              const float inv = inv_cnt[i];
              float v[TILE];
#pragma unroll
              for (int q = 0; q < TILE; ++q) v[q] = 0.f;
#pragma unroll
              for (int k = 0; k < kMaxBins; ++k) {
                if (k >= npw) break;
                const float gs = gk[k] * inv;
#pragma unroll
                for (int q = 0; q < TILE; ++q) v[q] += gs * wts[i][1][k][q];
              }
              const int rlo = __builtin_amdgcn_readfirstlane((int)row_lo[i][a]), rhi = __builtin_amdgcn_readfirstlane((int)row_hi[i][a]);
#pragma unroll
              for (int r = 0; r < TILE; ++r) {
                if (r < rlo || r >= rhi) continue;      // (scalar test: lines this bin row has no weight on)
                const float wy = wts[i][0][a][r];
#pragma unroll
                for (int q = 0; q < TILE; ++q) acc[r][q] += wy * v[q];
              }
            }
          }
          __syncthreads();                  // before the next batch overwrites the staged weights
        }
        __syncthreads();                    // before the next chunk overwrites roi_list / roi_row
      }
      // the four register copies of every channel block, added in a fixed order through LDS: (a0 + a1) + (a2 + a3)
      auto put = [&](int slot) {
#pragma unroll
        for (int r = 0; r < TILE; ++r)
#pragma unroll
          for (int q = 0; q < TILE; ++q) copies[slot][r * TILE + q][cblk * 64 + lane] = acc[r][q];
      };
      auto take = [&](int slot) {
#pragma unroll
        for (int r = 0; r < TILE; ++r)
#pragma unroll
          for (int q = 0; q < TILE; ++q) acc[r][q] += copies[slot][r * TILE + q][cblk * 64 + lane];
      };
      if (subset == 1) put(0);
      if (subset == 3) put(1);
      __syncthreads();
      if (subset == 0) take(0);
      if (subset == 2) take(1);
      __syncthreads();
      if (subset == 2) put(0);
      __syncthreads();
      if (subset == 0) { take(0); put(1); }
      __syncthreads();
      for (int i = t; i < TILE * TILE * (NCH / 4); i += NT) {   // whole rows, 16 bytes per lane
        const int cell = i / (NCH / 4), cc = (i % (NCH / 4)) * 4;
        const int y = y0 + cell / TILE, x = x0 + cell % TILE;
        if (y > y1 || x > x1 || cc >= nblk * 64) continue;
        const float* src = &copies[1][cell][cc];
        float4* dst = reinterpret_cast<float4*>(out + ((size_t)y * W + x) * C + cb0 + cc);
        float4 v = make_float4(src[0], src[1], src[2], src[3]);
        if (accumulate) {   // the maps hold another consumer's gradient already (same cell, same thread: plain read-add-write)
          const float4 o = *dst;
          v = make_float4(o.x + v.x, o.y + v.y, o.z + v.z, o.w + v.w);
        }
        *dst = v;
      }
      __syncthreads();                      // the copies are rewritten by the next job
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

// Gather form with the census guard over a level table: both forms are launched, the device-side census lets one of
// them return at once.
// census (one int per tile) + launch plan + per roi: reach + meta
static size_t align_gather_head_bytes(int ntile, int C) {
  return ((size_t)ntile * (1 + C / 64) * sizeof(int) + 4 * sizeof(int) + 15) & ~(size_t)15;
}
static size_t align_gather_bytes(int ntile, int C, int M) {
  return align_gather_head_bytes(ntile, C) + (size_t)M * (sizeof(int4) + sizeof(int));
}
// stream-ordered scratch of a call whose entry point has no workspace argument (the reference-shaped ones): released on
// every path out of the function
struct AsyncScratch {
  void* p = nullptr;
  hipStream_t st = nullptr;
  ~AsyncScratch() { if (p) (void)hipFreeAsync(p, st); }
};

static int align_backward_gather(const float* grad, const float* rois, AlignLevels& lv, int B, int C, int M, int PH,
                                 int PW, int sr, int aligned, const int* roi_level, hipStream_t st, bool accumulate = false,
                                 void* workspace = nullptr, size_t workspace_bytes = 0) {
  int ntile = 0;
  for (int l = 0; l < lv.n; ++l) {
    lv.tiles_x[l] = ceil_div(lv.W[l], kTile);
    lv.tiles_y[l] = ceil_div(lv.H[l], kTile);
    lv.first_tile[l] = ntile;
    ntile += B * lv.tiles_x[l] * lv.tiles_y[l];
  }
  lv.first_tile[lv.n] = ntile;
  const size_t need = align_gather_bytes(ntile, C, M);
  AsyncScratch own;
  own.st = st;
  if (workspace) {
    JTSM_REQUIRE(workspace_bytes >= need && ((uintptr_t)workspace & 15) == 0,
                 "roi_align backward: workspace of %zu bytes (16-byte aligned) needed, got %zu", need, workspace_bytes);
  } else {
    JTSM_CHECK_HIP(hipMallocAsync(&own.p, need, st));
  }
  // every map cleared first: the gather writes only the tiles some roi can reach, the scatter form adds into zeros
  // (accumulate: the maps already hold a gradient — both forms add to it and nothing is cleared)
  for (int l = 0; l < lv.n && !accumulate; ++l)
    JTSM_CHECK_HIP(hipMemsetAsync(lv.gin[l], 0, (size_t)B * lv.H[l] * lv.W[l] * C * sizeof(float), st));
  // census, then the launch plan (maximum, number of jobs, C / 64 (tile, channel block) jobs per reachable tile), then
  // per roi: reach + meta
  int* census = reinterpret_cast<int*>(workspace ? workspace : own.p);
  const int fan = C / 64;
  const size_t head = align_gather_head_bytes(ntile, C);
  JTSM_CHECK_HIP(hipMemsetAsync(census, 0, (size_t)ntile * sizeof(int), st));
  int* plan = census + ntile;
  int4* reach = reinterpret_cast<int4*>(reinterpret_cast<char*>(census) + head);   // cell reach (16-byte aligned)
  int* meta = reinterpret_cast<int*>(reach + M);                                    // image | level entry << 16
  hipLaunchKernelGGL(align_census_kernel, dim3(ceil_div(M, 256)), dim3(256), 0, st, rois, M, PH, PW, sr, aligned,
                     roi_level, lv, census, reach, meta);
  hipLaunchKernelGGL(tile_plan_kernel, dim3(1), dim3(1024), 0, st, census, ntile, plan, 1, fan);
  hipLaunchKernelGGL((align_bwd_gather<8, 1>), dim3(std::min(ntile * fan, 512)), dim3(256), 0, st, grad, rois, lv, C, M, PH,
                     PW, sr, aligned, plan, census_limit(), fan, reach, meta, accumulate ? 1 : 0);
  constexpr int V = WideVec<float>::value;
  const int blocks = ceil_div((long)M * PH * PW, 4);
  // the scatter form, all levels in one launch (returns at once unless the census says so)
  if (C % V == 0 && ((uintptr_t)grad % (V * sizeof(float))) == 0)
    hipLaunchKernelGGL((align_bwd_nhwc_levels<V>), dim3(blocks), dim3(256), 0, st, grad, rois, lv, C, M, PH, PW, sr, aligned,
                       roi_level, plan, census_limit());
  else
    hipLaunchKernelGGL((align_bwd_nhwc_levels<1>), dim3(blocks), dim3(256), 0, st, grad, rois, lv, C, M, PH, PW, sr, aligned,
                       roi_level, plan, census_limit());
  JTSM_CHECK_LAUNCH("roi_align backward (gather + census)");
  return JTSM_OK;      // (own scratch: released by AsyncScratch, also on the error returns above)
}

template <typename T, bool ROT>
int launch_backward(const T* grad, const T* rois, T* gin, int B, int C, int H, int W, int M,
                    T scale, int PH, int PW, int sr, int aligned, int layout, void* stream,
                    const int* roi_level = nullptr, int level = 0, bool accumulate = false) {
  JTSM_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0 && M >= 0 && PH > 0 && PW > 0,
               "roi_align backward: negative size");
  JTSM_REQUIRE(layout == JTSM_NCHW || layout == JTSM_NHWC, "roi_align: unknown layout %d", layout);
  hipStream_t st = as_stream(stream);
  const size_t in_elems = (size_t)B * C * H * W;
  if (in_elems == 0) return JTSM_OK;
  JTSM_REQUIRE(gin, "roi_align backward: null grad_input");
  // (a handful of rois — the mask branch's foreground set — is cheaper as memset + scatter than as one workgroup per
  // tile of the whole map: the gather starts at 8192 bins)
  if (std::is_same<T, float>::value && !ROT && layout == JTSM_NHWC && C % 64 == 0 && grad && rois && PW <= kMaxBins && PH <= kMaxBins &&
      (long)M * PH * PW >= 8192 && (long)M * PH * PW < (1L << 30)) {
    AlignLevels lv = {};
    lv.n = 1;
    lv.gin[0] = reinterpret_cast<float*>(gin); lv.H[0] = H; lv.W[0] = W; lv.scale[0] = (float)scale;
    lv.level_id[0] = level;
    return align_backward_gather(reinterpret_cast<const float*>(grad), reinterpret_cast<const float*>(rois), lv, B, C, M,
                                 PH, PW, sr, aligned, roi_level, st, accumulate);
  }
  if (!accumulate) JTSM_CHECK_HIP(hipMemsetAsync(gin, 0, in_elems * sizeof(T), st));
  if ((long)M * C * PH * PW == 0) return JTSM_OK;  // empty gradient: zeros (ROIAlign_cuda.cu:402-405)
  JTSM_REQUIRE(grad && rois, "roi_align backward: null pointer");
  if (layout == JTSM_NHWC) {
    const long waves = (long)M * PH * PW;
    const int blocks = ceil_div(waves, 4);
    constexpr int V = WideVec<T>::value;
    if (C % V == 0 && ((uintptr_t)grad % (V * sizeof(T))) == 0)
      hipLaunchKernelGGL((align_bwd_nhwc<T, V, ROT>), dim3(blocks), dim3(256), 0, st, grad, rois,
                         gin, C, H, W, M, scale, PH, PW, sr, aligned, roi_level, level, (const int*)nullptr, 0);
    else
      hipLaunchKernelGGL((align_bwd_nhwc<T, 1, ROT>), dim3(blocks), dim3(256), 0, st, grad, rois,
                         gin, C, H, W, M, scale, PH, PW, sr, aligned, roi_level, level, (const int*)nullptr, 0);
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

// ---- the crop_and_resize family, one workgroup per 16 bins of ONE roi (blockIdx.y) ------------------------------
// A bin of a roi of h x w pixels averages ceil(h / side) x ceil(w / side) samples: 1 for a small proposal, 700+ for one
// that covers the image.  With a thread per bin the step waited for the handful of wavefronts that walk the largest
// rois sample by sample (paste_crop 0.31 ms, sp_mask 0.20 ms for ~300 rois).  Here the sampling grid is block-uniform:
// rois of more than 16 samples per bin evaluate 64 samples at a time, one per lane, and fold the 64 terms in the
// reference's sequential (iy, ix) order — fp32 addition is not associative and the >= 0.5 threshold sits behind the sum,
// so the ORDER is kept and only the evaluation is spread.  A skipped sample contributes +0.f (x + 0.f == x for the
// non-negative sums here), as do the lanes past the last sample.
constexpr int CROP_BINS = 16;
template <class F>   // term(ph, pw, iy, ix): t.w[0] * v0 + t.w[1] * v1 + t.w[2] * v2 + t.w[3] * v3, or 0.f when skipped
__device__ __forceinline__ void crop_bins(int side, int gh, int gw, unsigned char* __restrict__ out_roi,
                                          float (*buf)[64], F term) {
#pragma clang fp contract(off)
  const int cells = gh * gw, nb = side * side, bin0 = blockIdx.x * CROP_BINS;
  const float count = (float)(cells > 1 ? cells : 1);
  if (cells <= 16) {   // a thread per bin
    const int b = bin0 + (int)threadIdx.x;
    if (threadIdx.x < CROP_BINS && b < nb) {
      const int ph = b / side, pw = b - ph * side;
      float acc = 0.f;
      for (int iy = 0; iy < gh; ++iy)
        for (int ix = 0; ix < gw; ++ix) acc += term(ph, pw, iy, ix);
      out_roi[b] = (acc / count) >= 0.5f ? 1 : 0;
    }
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int j = wave; j < CROP_BINS; j += 4) {   // a wavefront per bin
    const int b = bin0 + j;
    if (b >= nb) break;   // wave-uniform
    const int ph = b / side, pw = b - ph * side;
    float acc = 0.f;
    for (int k0 = 0; k0 < cells; k0 += 64) {
      const int k = k0 + lane;
      float s = 0.f;
      if (k < cells) {
        const int iy = k / gw, ix = k - iy * gw;
        s = term(ph, pw, iy, ix);
      }
      __builtin_amdgcn_wave_barrier();   // (the previous chunk's reads are issued before this write)
      buf[wave][lane] = s;
      __builtin_amdgcn_wave_barrier();   // one wavefront's LDS operations execute in order: the reads below see it
      float4 r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) r[q] = reinterpret_cast<const float4*>(buf[wave])[q];
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc += r[q].x; acc += r[q].y; acc += r[q].z; acc += r[q].w; }
    }
    if (lane == 0) out_roi[b] = (acc / count) >= 0.5f ? 1 : 0;
  }
}

// ---- mask targets from SUPERPIXEL EVIDENCE (object_evidence, roi_heads_jtsm.py:1928-1994, the reference's own
// grabCut-free construction): the instance mask of a (near) target is the union of the superpixels its oh_labels row
// marks — an image that is never rasterised here: pixel (y, x) of target t is oh_labels[t][superpixels[y][x]] != 0.
// Thread (roi, ph, pw) runs the ROIAlign(1.0, sampling 0, aligned) arithmetic of BitMasks.crop_and_resize
// (structures/masks.py:169-200) on that image and thresholds at 0.5.
__global__ __launch_bounds__(256) void sp_mask_targets_kernel(const float* __restrict__ rois,        // (N,4) boxes
                                                              const int* __restrict__ oh_row,        // (N) label row
                                                              const int* __restrict__ img_of,        // (N) image
                                                              const int* __restrict__ oh_labels, int L,
                                                              const int* __restrict__ sp,            // (B,H,W)
                                                              unsigned char* __restrict__ out, long total, int side,
                                                              int H, int W) {
#pragma clang fp contract(off)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % side), ph = (int)((idx / side) % side);
    const long n = idx / side / side;
    const int row = oh_row[n];
    if (row < 0) { out[idx] = 0; continue; }
    const float roi5[5] = {0.f, rois[n * 4], rois[n * 4 + 1], rois[n * 4 + 2], rois[n * 4 + 3]};
    const RoiGeom<float> g = geom_box<float>(roi5, 1.0f, side, side, 0, true);
    const int* lab = oh_labels + (size_t)row * L;
    const int* s = sp + (size_t)img_of[n] * H * W;
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
          const int id = s[t.pos[k]];
          v[k] = ((unsigned)id < (unsigned)L && lab[id] != 0) ? 1.f : 0.f;
        }
        acc += t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3];
      }
    out[idx] = (acc / count) >= 0.5f ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void sp_mask_targets_roi_kernel(const float* __restrict__ rois, const int* __restrict__ oh_row,
                                                                  const int* __restrict__ img_of,
                                                                  const int* __restrict__ oh_labels, int L,
                                                                  const int* __restrict__ sp, unsigned char* __restrict__ out,
                                                                  int side, int H, int W) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float buf[4][64];
  const long n = blockIdx.y;
  unsigned char* out_roi = out + n * side * side;
  const int row = oh_row[n];
  if (row < 0) {   // block-uniform
    const int b = blockIdx.x * CROP_BINS + (int)threadIdx.x;
    if (threadIdx.x < CROP_BINS && b < side * side) out_roi[b] = 0;
    return;
  }
  const float roi5[5] = {0.f, rois[n * 4], rois[n * 4 + 1], rois[n * 4 + 2], rois[n * 4 + 3]};
  const RoiGeom<float> g = geom_box<float>(roi5, 1.0f, side, side, 0, true);
  const int* lab = oh_labels + (size_t)row * L;
  const int* s = sp + (size_t)img_of[n] * H * W;
  crop_bins(side, g.gh, g.gw, out_roi, buf, [&](int ph, int pw, int iy, int ix) -> float {
    const Tap<float> t = sample_tap<float, false>(g, H, W, ph, pw, iy, ix);
    if (t.pos[0] < 0) return 0.f;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int id = s[t.pos[k]];
      v[k] = ((unsigned)id < (unsigned)L && lab[id] != 0) ? 1.f : 0.f;
    }
    return t.w[0] * v[0] + t.w[1] * v[1] + t.w[2] * v[2] + t.w[3] * v[3];
  });
}

// ---- targets of the mask REFINERY (get_pgt_mask, roi_heads_jtsm.py:1997-2022): the previous head's class
// probability (M x M) is pasted into the image at the proposal box (paste_masks_in_image: grid_sample bilinear,
// zeros, align_corners=False, >= threshold; detectron2/layers/mask_ops.py:74-152) and that bitmask is cropped back to
// the same box at side x side (crop_and_resize).  Fused: the pasted image is evaluated per ROIAlign tap, never stored.
// (The reference additionally encodes the bitmask as polygons and rasterises them again — skipped, see DESIGN.)
__device__ __forceinline__ float pc_tap(const float* __restrict__ m, int M, int iy, int ix) {
  return (iy >= 0 && iy < M && ix >= 0 && ix < M) ? m[iy * M + ix] : 0.f;
}
__device__ __forceinline__ float pasted_bit(const float* __restrict__ m, int M, float x0, float y0, float x1, float y1,
                                            int x, int y, float threshold) {
#pragma clang fp contract(off)
  const float gy = ((float)y + 0.5f - y0) / (y1 - y0) * 2.f - 1.f;
  const float iy = ((gy + 1.f) * (float)M - 1.f) / 2.f;
  const float fy = floorf(iy);
  const float gx = ((float)x + 0.5f - x0) / (x1 - x0) * 2.f - 1.f;
  const float ix = ((gx + 1.f) * (float)M - 1.f) / 2.f;
  const float fx = floorf(ix);
  float acc = 0.f;
  if (fx >= -1.f && fx < (float)M && fy >= -1.f && fy < (float)M) {
    const int iy0 = (int)fy, iy1 = iy0 + 1, ix0 = (int)fx, ix1 = ix0 + 1;
    const float wy1 = iy - fy, wy0 = (fy + 1.f) - iy, wx1 = ix - fx, wx0 = (fx + 1.f) - ix;
    acc += pc_tap(m, M, iy0, ix0) * (wx0 * wy0);
    acc += pc_tap(m, M, iy0, ix1) * (wx1 * wy0);
    acc += pc_tap(m, M, iy1, ix0) * (wx0 * wy1);
    acc += pc_tap(m, M, iy1, ix1) * (wx1 * wy1);
  }
  return acc >= threshold ? 1.f : 0.f;
}

// pasted_bit, separated by axis: everything pasted_bit derives from a pixel's row (or column) alone — the same
// expressions, evaluated once per row / column of a sample instead of once per tap (two divisions and a floor each).
struct PasteAxis { float w0, w1; int i0; bool any; };
__device__ __forceinline__ PasteAxis paste_axis(int p, float lo, float hi, int M) {
#pragma clang fp contract(off)
  const float g = ((float)p + 0.5f - lo) / (hi - lo) * 2.f - 1.f;
  const float i = ((g + 1.f) * (float)M - 1.f) / 2.f;
  const float f = floorf(i);
  PasteAxis a;
  a.any = f >= -1.f && f < (float)M;
  a.i0 = (int)f;
  a.w1 = i - f;
  a.w0 = (f + 1.f) - i;
  return a;
}
__device__ __forceinline__ float pasted_bit_axes(const float* __restrict__ m, int M, const PasteAxis& ay,
                                                 const PasteAxis& ax, float threshold) {
#pragma clang fp contract(off)
  float acc = 0.f;
  if (ax.any && ay.any) {
    acc += pc_tap(m, M, ay.i0, ax.i0) * (ax.w0 * ay.w0);
    acc += pc_tap(m, M, ay.i0, ax.i0 + 1) * (ax.w1 * ay.w0);
    acc += pc_tap(m, M, ay.i0 + 1, ax.i0) * (ax.w0 * ay.w1);
    acc += pc_tap(m, M, ay.i0 + 1, ax.i0 + 1) * (ax.w1 * ay.w1);
  }
  return acc >= threshold ? 1.f : 0.f;
}

__global__ __launch_bounds__(256) void paste_crop_targets_kernel(const float* __restrict__ probs,   // (N, M, M)
                                                                 const float* __restrict__ rois,    // (N, 4)
                                                                 unsigned char* __restrict__ out, long total, int M,
                                                                 int side, int H, int W, float threshold) {
#pragma clang fp contract(off)
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int pw = (int)(idx % side), ph = (int)((idx / side) % side);
    const long n = idx / side / side;
    const float x0 = rois[n * 4], y0 = rois[n * 4 + 1], x1 = rois[n * 4 + 2], y1 = rois[n * 4 + 3];
    const float roi5[5] = {0.f, x0, y0, x1, y1};
    const RoiGeom<float> g = geom_box<float>(roi5, 1.0f, side, side, 0, true);
    const float* m = probs + (size_t)n * M * M;
    const int cells = g.gh * g.gw;
    const float count = (float)(cells > 1 ? cells : 1);
    // (the kernel is bound by instruction issue: per tap the first form spent an integer division to recover (y, x)
    // and two float divisions + a floor per axis; a sample's four taps share two rows and two columns)
    float acc = 0.f;
    for (int iy = 0; iy < g.gh; ++iy)
      for (int ix = 0; ix < g.gw; ++ix) {
        const Tap<float> t = sample_tap<float, false>(g, H, W, ph, pw, iy, ix);
        if (t.pos[0] < 0) continue;
        // make_tap: pos = {yl W + xl, yl W + xh, yh W + xl, yh W + xh} with yh in {yl, yl + 1}, xh in {xl, xl + 1}
        const int yl = t.pos[0] / W, xl = t.pos[0] - yl * W;
        const int xh = t.pos[1] - yl * W, yh = yl + (t.pos[2] != t.pos[0] ? 1 : 0);
        const PasteAxis ayl = paste_axis(yl, y0, y1, M), axl = paste_axis(xl, x0, x1, M);
        const PasteAxis ayh = yh == yl ? ayl : paste_axis(yh, y0, y1, M);
        const PasteAxis axh = xh == xl ? axl : paste_axis(xh, x0, x1, M);
        const float v0 = pasted_bit_axes(m, M, ayl, axl, threshold), v1 = pasted_bit_axes(m, M, ayl, axh, threshold);
        const float v2 = pasted_bit_axes(m, M, ayh, axl, threshold), v3 = pasted_bit_axes(m, M, ayh, axh, threshold);
        acc += t.w[0] * v0 + t.w[1] * v1 + t.w[2] * v2 + t.w[3] * v3;
      }
    out[idx] = (acc / count) >= 0.5f ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void paste_crop_targets_roi_kernel(const float* __restrict__ probs, const float* __restrict__ rois,
                                                                     unsigned char* __restrict__ out, int M, int side, int H,
                                                                     int W, float threshold) {
#pragma clang fp contract(off)
  __shared__ __attribute__((aligned(16))) float buf[4][64];
  const long n = blockIdx.y;
  const float x0 = rois[n * 4], y0 = rois[n * 4 + 1], x1 = rois[n * 4 + 2], y1 = rois[n * 4 + 3];
  const float roi5[5] = {0.f, x0, y0, x1, y1};
  const RoiGeom<float> g = geom_box<float>(roi5, 1.0f, side, side, 0, true);
  const float* m = probs + (size_t)n * M * M;
  crop_bins(side, g.gh, g.gw, out + n * side * side, buf, [&](int ph, int pw, int iy, int ix) -> float {
    const Tap<float> t = sample_tap<float, false>(g, H, W, ph, pw, iy, ix);
    if (t.pos[0] < 0) return 0.f;
    const int yl = t.pos[0] / W, xl = t.pos[0] - yl * W;
    const int xh = t.pos[1] - yl * W, yh = yl + (t.pos[2] != t.pos[0] ? 1 : 0);
    const PasteAxis ayl = paste_axis(yl, y0, y1, M), axl = paste_axis(xl, x0, x1, M);
    const PasteAxis ayh = yh == yl ? ayl : paste_axis(yh, y0, y1, M);
    const PasteAxis axh = xh == xl ? axl : paste_axis(xh, x0, x1, M);
    const float v0 = pasted_bit_axes(m, M, ayl, axl, threshold), v1 = pasted_bit_axes(m, M, ayl, axh, threshold);
    const float v2 = pasted_bit_axes(m, M, ayh, axl, threshold), v3 = pasted_bit_axes(m, M, ayh, axh, threshold);
    return t.w[0] * v0 + t.w[1] * v1 + t.w[2] * v2 + t.w[3] * v3;
  });
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

// Rotated boxes on one FPN level (the level loop of ROIPooler.forward runs for every pooler type,
// detectron2/modeling/poolers.py:160-165,230-249): rois are (M,6), the launch serves the rois with roi_level[m] == level.
// The backward is the scatter form (float atomics) on that level's map; accumulate != 0 adds into a map that already
// holds a gradient instead of clearing it first.
int jtsm_roi_align_rotated_forward_level_f32(const float* input, const float* rois, const int32_t* roi_level,
                                             int level, float* output, int B, int C, int H, int W, int M,
                                             float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                             void* stream) {
  JTSM_REQUIRE(roi_level || M == 0, "roi_align_rotated level: null roi_level");
  return launch_forward<float, true>(input, rois, output, B, C, H, W, M, spatial_scale, pooled_h, pooled_w,
                                     sampling_ratio, 1, JTSM_NHWC, stream, roi_level, level);
}
int jtsm_roi_align_rotated_backward_level_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                              int level, float* grad_input, int B, int C, int H, int W, int M,
                                              float spatial_scale, int pooled_h, int pooled_w, int sampling_ratio,
                                              int accumulate, void* stream) {
  JTSM_REQUIRE(roi_level || M == 0, "roi_align_rotated level: null roi_level");
  return launch_backward<float, true>(grad, rois, grad_input, B, C, H, W, M, spatial_scale, pooled_h, pooled_w,
                                      sampling_ratio, 1, JTSM_NHWC, stream, roi_level, level, accumulate != 0);
}

size_t jtsm_roi_align_backward_levels_workspace_bytes(const int* H, const int* W, int nlevels, int B, int C, int M) {
  if (!H || !W || nlevels <= 0 || B <= 0 || C <= 0 || M < 0) return 0;
  long ntile = 0;
  for (int l = 0; l < nlevels; ++l) ntile += (long)B * ceil_div(W[l], kTile) * ceil_div(H[l], kTile);
  return align_gather_bytes((int)ntile, C, M);
}

int jtsm_roi_align_backward_levels_f32(const float* grad, const float* rois, const int32_t* roi_level,
                                       float* const* grad_inputs, const int* H, const int* W, const float* scales,
                                       int nlevels, int B, int C, int M, int pooled_h, int pooled_w, int sampling_ratio,
                                       int aligned, int accumulate, void* workspace, size_t workspace_bytes,
                                       void* stream) {
  JTSM_REQUIRE(nlevels > 0 && nlevels <= kAlignLevels && grad_inputs && H && W && scales,
               "roi_align levels: bad level table");
  JTSM_REQUIRE(B >= 0 && C >= 0 && M >= 0 && pooled_h > 0 && pooled_w > 0, "roi_align levels: negative size");
  JTSM_REQUIRE(roi_level || M == 0, "roi_align levels: null roi_level");
  hipStream_t st = as_stream(stream);
  const bool gather = C % 64 == 0 && C > 0 && B > 0 && grad && rois && pooled_w <= kMaxBins && pooled_h <= kMaxBins &&
                      (long)M * pooled_h * pooled_w >= 8192 && (long)M * pooled_h * pooled_w < (1L << 30);
  if (gather) {
    AlignLevels lv = {};
    lv.n = 0;
    for (int l = 0; l < nlevels; ++l) {
      if (!grad_inputs[l]) continue;           // a level whose map needs no gradient
      JTSM_REQUIRE(H[l] > 0 && W[l] > 0, "roi_align levels: empty map");
      const int k = lv.n++;
      lv.gin[k] = grad_inputs[l]; lv.H[k] = H[l]; lv.W[k] = W[l]; lv.scale[k] = scales[l]; lv.level_id[k] = l;
    }
    if (lv.n == 0) return JTSM_OK;
    return align_backward_gather(grad, rois, lv, B, C, M, pooled_h, pooled_w, sampling_ratio, aligned, roi_level, st,
                                 accumulate != 0, workspace, workspace_bytes);
  }
  for (int l = 0; l < nlevels; ++l) {
    if (!grad_inputs[l]) continue;
    const int rc = launch_backward<float, false>(grad, rois, grad_inputs[l], B, C, H[l], W[l], M, scales[l], pooled_h,
                                                 pooled_w, sampling_ratio, aligned, JTSM_NHWC, stream, roi_level, l,
                                                 accumulate != 0);
    if (rc) return rc;
  }
  return JTSM_OK;
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

int jtsm_sp_mask_targets_f32(const float* rois, const int32_t* oh_row, const int32_t* img_of, const int32_t* oh_labels,
                             int L, const int32_t* superpixels, uint8_t* out, int N, int side, int H, int W,
                             void* stream) {
  JTSM_REQUIRE(N >= 0 && side > 0 && H > 0 && W > 0 && L > 0, "sp_mask_targets: bad sizes");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(rois && oh_row && img_of && oh_labels && superpixels && out, "sp_mask_targets: null pointer");
  const long total = (long)N * side * side;
  if (N <= 65535 && side <= 1024) {   // a workgroup per 16 bins of one roi (crop_bins)
    hipLaunchKernelGGL(sp_mask_targets_roi_kernel, dim3((side * side + CROP_BINS - 1) / CROP_BINS, N), dim3(256), 0,
                       as_stream(stream), rois, oh_row, img_of, oh_labels, L, superpixels, out, side, H, W);
    JTSM_CHECK_LAUNCH("sp_mask_targets");
    return JTSM_OK;
  }
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(sp_mask_targets_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), rois, oh_row, img_of,
                     oh_labels, L, superpixels, out, total, side, H, W);
  JTSM_CHECK_LAUNCH("sp_mask_targets");
  return JTSM_OK;
}

int jtsm_paste_crop_targets_f32(const float* probs, const float* rois, uint8_t* out, int N, int M, int side, int H,
                                int W, float threshold, void* stream) {
  JTSM_REQUIRE(N >= 0 && M > 0 && side > 0 && H > 0 && W > 0, "paste_crop_targets: bad sizes");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(probs && rois && out, "paste_crop_targets: null pointer");
  const long total = (long)N * side * side;
  if (N <= 65535 && side <= 1024) {   // a workgroup per 16 bins of one roi (crop_bins)
    hipLaunchKernelGGL(paste_crop_targets_roi_kernel, dim3((side * side + CROP_BINS - 1) / CROP_BINS, N), dim3(256), 0,
                       as_stream(stream), probs, rois, out, M, side, H, W, threshold);
    JTSM_CHECK_LAUNCH("paste_crop_targets");
    return JTSM_OK;
  }
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  hipLaunchKernelGGL(paste_crop_targets_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), probs, rois, out, total,
                     M, side, H, W, threshold);
  JTSM_CHECK_LAUNCH("paste_crop_targets");
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
