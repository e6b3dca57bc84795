// fp32 implicit-GEMM convolution / linear layers on the gfx950 matrix cores.
//
// One kernel family serves every dense contraction on the JTSM hot path (SURVEY §8 a1,a2,a5,
// a12,a13,a17,a18,a19: ResNet/FPN convolutions, DAN and predictor linears, mask / sem-seg head
// convolutions) in its three roles:
//     FWD    Y[p, co]          = sum_{tap,ci} X[pix(p,tap), ci] * W[co, tap, ci]
//     DGRAD  dX[q, ci]         = sum_{tap,co} dY[opix(q,tap), co] * (kscale[co]) W[co, tap, ci]
//     WGRAD  dW[co, tap, ci]  += sum_{p}      dY[p, co] * X[pix(p,tap), ci]        (split over p)
// Activations are NHWC, weights OHWI ([Cout][KH*KW][Cin]) — both K-contiguous for FWD — so a
// 1x1 convolution and nn.Linear ([out][in]) are the same call with H=W=1.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 products, fp32 accumulate (bit-for-bit an
// fmaf chain; MI355X peak 157 TFLOP/s = 64 FLOP/clk/SIMD).  This is what makes the 1e-4
// parity bar of BASELINE.json reachable without a reduced-precision detour.
//
// Tiling: a workgroup of 4 wavefronts owns a BM x BN tile of the output; each wavefront owns
// 64x64 of it as 2x2 MFMA tiles (64 accumulator VGPRs).  K advances in tiles of 32.  Both
// operands are staged through LDS K-major ([32][BM+pad]), which makes every MFMA fragment read
// (lane -> row, lane>>5 -> k) a conflict-free ds_read_b32.  Global loads are 16 B per lane and
// coalesced along whatever axis is contiguous in memory:
//     "K-contiguous" operands (FWD/DGRAD activations, FWD weights): 8 lanes cover 128 B of one
//         row; stored to LDS transposed, row stride 129 floats -> conflict-free ds_write_b32;
//     "row-contiguous" operands (DGRAD weights, WGRAD both): 32 lanes cover 512 B of one k-row;
//         stored with one ds_write_b128, row stride 132 floats (16-B aligned).
// The next K tile is fetched into registers while the current one is multiplied (register
// double-buffering); 4 workgroups fit a CU (33 KiB LDS, <=128 VGPRs), so MFMA work of one
// group covers the barrier/stage bubbles of the others.
//
// Epilogue (fused, per output element): *scale[n] +bias[n] +residual[m,n] relu mask — i.e.
// FrozenBatchNorm + shortcut add + ReLU of a bottleneck (detectron2/modeling/backbone/
// resnet.py:195-211, layers/batch_norm.py:45-66, layers/wrappers.py:62-83) in FWD, and
// "sum the two gradient paths, then gate by the ReLU mask" in DGRAD.  WGRAD accumulates with
// float atomics into a zero-filled dW (optionally scaled per output row).
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <type_traits>

#include "common.h"

namespace jtsm {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Role { FWD = 0, DGRAD = 1, WGRAD = 2 };

struct ConvShape {
  int Bn, H, W, Cin;   // input activation  X : (Bn, H, W, Cin)
  int Ho, Wo, Cout;    // output activation Y : (Bn, Ho, Wo, Cout)
  int KH, KW, stride, pad, dil;
};

struct Epilogue {
  const float* scale;     // [N] (WGRAD: [M]) or null
  const float* bias;      // [N] or null
  const float* residual;  // [M][ldc] or null (may alias the output)
  const float* mask;      // [M][ldc] or null: keep value where mask > 0
  int relu;
  // fp16-only activations (BASELINE configs[4], the reference's AMP step: detectron2/engine/train_loop.py:289-336): the
  // residual operand as ONE fp16 plane carrying 2^res_shift (a block input kept as its operand plane only; a gradient
  // plane), used when `residual` is null.  Wide epilogue and the finishing passes only.
  const unsigned short* residual_h;
  int res_shift;
};

struct Params {
  const float* A;   // FWD: X      DGRAD: dY     WGRAD: dY
  const float* B;   // FWD: W      DGRAD: W      WGRAD: X
  float* C;         // FWD: Y      DGRAD: dX     WGRAD: dW
  const float* kscale;  // DGRAD only: per-Cout multiplier folded into W rows, or null
  int M, N, K;
  int ldc;
  int ktiles_per_split;  // split-K: K tiles per grid.y slice
  float* slab;           // FWD/DGRAD split-K: raw partial tiles go to slab[blockIdx.y][M][ldc]
  // DGRAD of a strided 1x1 convolution runs as a dense GEMM over the OUTPUT pixels; row m of the result
  // is scattered to input pixel (b, oh*stride, ow*stride) of a zero-filled dX.
  int scatter, sc_Ho, sc_Wo, sc_H, sc_W, sc_stride;
  int wide;   // epilogue through LDS with 16-byte accesses (set by the launcher when alignment allows)
  unsigned short* out_hi;   // optional: bf16 hi / lo planes of the finished output, for a bf16x3 consumer
  unsigned short* out_lo;   // (null with out_hi set: ONE IEEE fp16 plane, for an f16 consumer)
  // fp16 path: gradient planes are stored times 2^k so that small gradients stay above fp16's subnormal range.
  int in_shift;             // the accumulator is multiplied by 2^-in_shift first (an operand plane carried 2^in_shift)
  int out_shift;            // emitted planes are multiplied by 2^out_shift
  // Split-K finishing INSIDE the contraction kernel (null: the separate splitk_finish pass).  One zero-initialised
  // counter per output tile (blockIdx.x): a workgroup publishes its slab with write-through stores and takes a ticket;
  // the last of a tile's gridDim.y slices adds all slabs in slice order and runs the epilogue.
  int* tickets;
  // FWD as the GEMM of a 2x2 / stride-2 transposed convolution: row m = input pixel (b, y, x), column n = (dy, dx, c)
  // with c < shuffle_c; the result lands at output pixel (b, 2y + dy, 2x + dx), channel c (bias indexed by c).
  int shuffle_c, shuffle_h, shuffle_w;
  // A ReLU gate read from the hi (or only) PLANE of the activation instead of its fp32 copy: keep the result where the
  // plane's 16-bit pattern is a positive number (sign clear, not zero) — bf16 and fp16 round a positive fp32 to a
  // positive value, so this is the same gate at half the bytes, and lets a chain keep activations as planes only
  // (C == null: the fp32 result is not stored at all; needs out_hi).
  const unsigned short* mask_plane;
  int scale_rows;   // FWD / DGRAD: e.scale holds one factor per output ROW (pixel / roi), not per column
  // WGRAD: bias gradient computed beside dW (conv_x3.h: x3_bias_mma): per-slice partial sums bias_slab[slice][M]
  // (the result itself when there is one slice), folded into bias_out[M] by the finishing pass.
  float* bias_slab;
  float* bias_out;
  // FWD / DGRAD, unsplit, wide epilogue: per-row-tile column sums of the FINISHED result (after scale / residual / gate),
  // colsum[row tile][N] — the bias gradient of the layer whose output gradient this launch produces, taken where that
  // gradient is written instead of by a pass over it afterwards (the caller folds the row tiles in order).
  float* colsum;
  ConvShape s;
  Epilogue e;
};

constexpr int BK = 32;
constexpr int PAD_T = 1;  // transposed-store tiles: stride BM+1
constexpr int PAD_D = 4;  // direct-store tiles:     stride BM+4

// 2^k as a float (|k| < 127), exact
__device__ __forceinline__ float pow2i(int k) { return __int_as_float((127 + k) << 23); }

// Epilogue::residual_h: four / one element(s) of the fp16 residual plane as floats
__device__ __forceinline__ float4 res_h4(const Epilogue& e, size_t o) {
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  const f16x4_t h = *reinterpret_cast<const f16x4_t*>(e.residual_h + o);
  const float a = pow2i(-e.res_shift);
  return make_float4((float)h[0] * a, (float)h[1] * a, (float)h[2] * a, (float)h[3] * a);
}
__device__ __forceinline__ float res_h1(const Epilogue& e, size_t o) {
  return (float)reinterpret_cast<const _Float16*>(e.residual_h)[o] * pow2i(-e.res_shift);
}

__device__ __forceinline__ float4 ldg4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

// ---- operand addressing ------------------------------------------------------------------------
// Every operand is fetched in 16-byte chunks.  Loads are UNCONDITIONAL (an out-of-range chunk reads the
// buffer's first 16 bytes and is zeroed by a select afterwards): a conditional load makes hipcc branch
// around it and wait vmcnt(0) per chunk, which serialises the prefetch.  Index decompositions
// (k -> tap, channel;  pixel -> b, oh, ow) are advanced incrementally from K tile to K tile instead of
// being re-divided.

struct PixelRow { int b, h0, w0; bool ok; };

__device__ __forceinline__ PixelRow fwd_pixel(const ConvShape& s, int m, int M) {
  PixelRow r;
  r.ok = m < M;
  const int mm = r.ok ? m : 0;
  const int ow = mm % s.Wo, t = mm / s.Wo;
  const int oh = t % s.Ho;
  r.b = t / s.Ho;
  r.h0 = oh * s.stride - s.pad;
  r.w0 = ow * s.stride - s.pad;
  return r;
}
__device__ __forceinline__ PixelRow dgrad_pixel(const ConvShape& s, int m, int M) {
  PixelRow r;
  r.ok = m < M;
  const int mm = r.ok ? m : 0;
  const int iw = mm % s.W, t = mm / s.W;
  const int ih = t % s.H;
  r.b = t / s.H;
  r.h0 = ih + s.pad;
  r.w0 = iw + s.pad;
  return r;
}

// k = (kh*KW + kw)*C + c for a K-contiguous operand whose innermost run has C channels.
struct TapState { int k, kh, kw, c; };
__device__ __forceinline__ TapState tap_init(int k, int C, int KW) {
  TapState t;
  t.k = k;
  const int tap = k / C;
  t.c = k - tap * C;
  t.kh = tap / KW;
  t.kw = tap - t.kh * KW;
  return t;
}
__device__ __forceinline__ void tap_advance(TapState& t, int step, int C, int KW) {
  if (C < step) {  // tiny channel counts (RGB stem): many taps per step
    t = tap_init(t.k + step, C, KW);
    return;
  }
  t.k += step;
  t.c += step;
  while (t.c >= C) {
    t.c -= C;
    if (++t.kw == KW) { t.kw = 0; ++t.kh; }
  }
}

// element offset of X[pix(row, tap)][c] for FWD, or -1
__device__ __forceinline__ int fwd_a_off(const ConvShape& s, const PixelRow& r, const TapState& t, int kend) {
  const int ih = r.h0 + t.kh * s.dil, iw = r.w0 + t.kw * s.dil;
  const bool ok = r.ok && t.k < kend && (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
  return ok ? ((r.b * s.H + ih) * s.W + iw) * s.Cin + t.c : -1;
}
// element offset of dY[opix(row, tap)][co] for DGRAD, or -1
__device__ __forceinline__ int dgrad_a_off(const ConvShape& s, const PixelRow& r, const TapState& t, int kend) {
  const int th = r.h0 - t.kh * s.dil, tw = r.w0 - t.kw * s.dil;
  int oh = th, ow = tw;
  bool ok = r.ok && t.k < kend && th >= 0 && tw >= 0;
  if (s.stride != 1) {
    oh = th / s.stride;
    ow = tw / s.stride;
    ok = ok && oh * s.stride == th && ow * s.stride == tw;
  }
  ok = ok && oh < s.Ho && ow < s.Wo;
  return ok ? ((r.b * s.Ho + oh) * s.Wo + ow) * s.Cout + t.c : -1;
}

// row k = tap*Cout + co of W viewed as [K][Cin] (DGRAD B)
struct CoState { int k, tap, co; };
__device__ __forceinline__ CoState co_init(int k, int Cout) {
  CoState c;
  c.k = k;
  c.tap = k / Cout;
  c.co = k - c.tap * Cout;
  return c;
}
__device__ __forceinline__ void co_advance(CoState& c, int step, int Cout) {
  if (Cout < step) {
    c = co_init(c.k + step, Cout);
    return;
  }
  c.k += step;
  c.co += step;
  while (c.co >= Cout) { c.co -= Cout; ++c.tap; }
}

// output pixel k -> (b, oh, ow) (WGRAD B)
struct PixState { int k, b, oh, ow; };
__device__ __forceinline__ PixState pix_init(int k, int Ho, int Wo) {
  PixState p;
  p.k = k;
  p.ow = k % Wo;
  const int t = k / Wo;
  p.oh = t % Ho;
  p.b = t / Ho;
  return p;
}
__device__ __forceinline__ void pix_advance(PixState& p, int step, int Ho, int Wo) {
  if (Wo < step) {  // narrow maps (nn.Linear: Ho = Wo = 1): many wraps per step, re-divide instead
    p = pix_init(p.k + step, Ho, Wo);
    return;
  }
  p.k += step;
  p.ow += step;
  while (p.ow >= Wo) {
    p.ow -= Wo;
    if (++p.oh == Ho) { p.oh = 0; ++p.b; }
  }
}

__device__ __forceinline__ float4 load_or_zero(const float* __restrict__ base, int off) {
  const float4 v = ldg4(base + (off < 0 ? 0 : off));
  return off < 0 ? make_float4(0.f, 0.f, 0.f, 0.f) : v;
}

// Write one wavefront's 64x64 accumulator tile with the fused epilogue (or as a raw split-K slab / an
// atomic WGRAD contribution).
template <int ROLE, int TM = 2, int TN = 2>
__device__ __forceinline__ void store_tile(const Params& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                           int lane) {
  // Epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const Epilogue& e = p.e;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float sc = (ROLE != WGRAD && e.scale) ? e.scale[n] : 1.f;
    const float bi = e.bias ? e.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (ROLE != WGRAD && gridDim.y > 1) {  // split-K: raw partial into this slice's slab
          p.slab[((size_t)blockIdx.y * p.M + m) * p.ldc + n] = v;
          continue;
        }
        v *= pow2i(-p.in_shift);
        size_t o = (size_t)m * p.ldc + n;
        if (ROLE == DGRAD && p.scatter) {
          const int ow = m % p.sc_Wo, t = m / p.sc_Wo;
          const int oh = t % p.sc_Ho, b = t / p.sc_Ho;
          o = ((size_t)(b * p.sc_H + oh * p.sc_stride) * p.sc_W + ow * p.sc_stride) * p.ldc + n;
        }
        if (ROLE == WGRAD) {
          if (e.scale) v *= e.scale[m];
          atomicAdd(p.C + o, v);
        } else {
          v = v * sc + bi;
          if (e.residual) v += e.residual[o];
          if (e.relu) v = fmaxf(v, 0.f);
          if (e.mask) v = e.mask[o] > 0.f ? v : 0.f;
          p.C[o] = v;
        }
      }
    }
  }
}

// hi/lo bf16 planes of four finished outputs (what jtsm_split_bf16_f32 would produce from them); with lo == null,
// one fp16 plane of v * 2^shift (what jtsm_split_f16_f32 would produce).
__device__ __forceinline__ void emit_planes4(unsigned short* hi, unsigned short* lo, size_t o, const float4& v,
                                             int shift = 0) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const float x[4] = {v.x, v.y, v.z, v.w};
  if (!lo) {
    typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
    const float sc = pow2i(shift);
    f16x4_t h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (_Float16)(x[e] * sc);
    *reinterpret_cast<f16x4_t*>(hi + o) = h;
    return;
  }
  bf16x4_t h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
  *reinterpret_cast<bf16x4_t*>(hi + o) = h;
  *reinterpret_cast<bf16x4_t*>(lo + o) = l;
}

// The same epilogue with 16-byte global accesses: the workgroup's BM x BN accumulator tile is parked in LDS
// (free once the K loop is over; BM*BN*4 = 64 KiB), then every thread streams float4 pieces of whole output
// rows — 512 contiguous bytes per 32 lanes — through the scale / bias / residual / ReLU / gate chain.
// Short-K layers (1x1 convolutions into wide outputs) are bound by exactly this traffic.
// Needs N % 4 == 0, ldc % 4 == 0 and 16-byte aligned C / residual / mask / slab (checked by the launcher).
// Which output row a tile row is: consecutive rows of the GEMM (the default), or a TH x TW block of pixels of one
// image (the halo kernels, whose M tile is a 2-D patch).  -1 = outside the tensor.
struct LinearRows {
  int m0, M;
  __device__ __forceinline__ int operator()(int row) const { const int m = m0 + row; return m < M ? m : -1; }
};
struct PatchRows {
  int b, y0, x0, OH, OW, TW;
  __device__ __forceinline__ int operator()(int row) const {
    const int y = y0 + row / TW, x = x0 + row % TW;
    return (y < OH && x < OW) ? (b * OH + y) * OW + x : -1;
  }
};

// ---- device-scope message passing without device-scope FENCES --------------------------------------------------
// A fence at agent scope (__threadfence) writes back / invalidates the XCD's whole L2: tried for split-K finishing in
// round 1, it doubled the step.  Here only the slab traffic itself is made coherent, instruction by instruction: slabs
// are written with sc1 stores (agent scope: written through the XCD's L2) and read back with sc1 loads (agent scope:
// served from the coherent level, not from this XCD's possibly stale L2 / L1) — the encodings the LLVM AMDGPU memory
// model prescribes for monotonic agent-scope atomics on gfx942 / gfx950 — and the ticket is an agent-scope atomic.
// Ordering: a workgroup waits for its stores' acknowledgements (s_waitcnt vmcnt(0) in every thread, then the
// workgroup barrier) before thread 0 takes the ticket; the reader's loads are issued after it has seen the ticket.
typedef float fx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_sc1_x4(float* ptr, const float4& v) {
  const fx4 t = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(t) : "memory");
}
__device__ __forceinline__ void st_sc1(float* ptr, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}
// eight 16-byte loads in flight, then wait for all of them (one asm block: the compiler cannot see that the
// destinations are written asynchronously)
__device__ __forceinline__ void ld_sc1_x4_8(const float* const (&q)[8], fx4 (&o)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
      : "memory");
}
__device__ __forceinline__ void ld_sc1_8(const float* const (&q)[8], float (&o)[8]) {
  asm volatile(
      "global_load_dword %0, %8, off sc1\n\t"
      "global_load_dword %1, %9, off sc1\n\t"
      "global_load_dword %2, %10, off sc1\n\t"
      "global_load_dword %3, %11, off sc1\n\t"
      "global_load_dword %4, %12, off sc1\n\t"
      "global_load_dword %5, %13, off sc1\n\t"
      "global_load_dword %6, %14, off sc1\n\t"
      "global_load_dword %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
      : "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5]), "v"(q[6]), "v"(q[7])
      : "memory");
}

// After a workgroup's slab is written: true in exactly one workgroup of each tile — the last of its gridDim.y slices
// (which also re-arms the counter for the next launch).  `flag` is one int of LDS.
__device__ __forceinline__ bool splitk_last_arrival(const Params& p, int* flag, int tile_id) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's slab stores are acknowledged
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(p.tickets + tile_id, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old == (int)gridDim.y - 1;
    if (last) __hip_atomic_store(p.tickets + tile_id, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = last;
  }
  __syncthreads();
  return *flag != 0;
}

// The last-arriving slice's work: thread-private float4 pieces k = 0 .. npiece-1 (piece_of(k, m, n) -> valid?), each
// the sum of all gridDim.y slabs in slice order, handed to finish(m, n, sum).  Eight 16-byte loads travel together:
// SPAD slices of 8 / SPAD pieces when there are at most eight slices, otherwise eight slices of one piece per round.
template <class PIECE, class FIN>
__device__ __forceinline__ void splitk_fold(const Params& p, int npiece, PIECE piece_of, FIN finish) {
  const int S = (int)gridDim.y;
  const size_t slice = (size_t)p.M * p.ldc;
  auto run = [&](auto spad_c) {
    constexpr int SPAD = decltype(spad_c)::value, PP = 8 / SPAD;
    for (int it0 = 0; it0 < npiece; it0 += PP) {
      float4 sum[PP];
      int pm[PP], pn[PP];
#pragma unroll
      for (int u = 0; u < PP; ++u) {
        sum[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        pm[u] = -1;
        pn[u] = 0;
        int m, n;
        if (it0 + u < npiece && piece_of(it0 + u, m, n)) { pm[u] = m; pn[u] = n; }
      }
      for (int s0 = 0; s0 < S; s0 += SPAD) {            // (more than one round only when S > 8, where PP == 1)
        const float* q[8];
        bool ok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int u = j / SPAD, sl = s0 + j % SPAD;
          ok[j] = pm[u] >= 0 && sl < S;
          q[j] = ok[j] ? p.slab + (size_t)sl * slice + (size_t)pm[u] * p.ldc + pn[u] : p.slab;
        }
        fx4 o[8];
        ld_sc1_x4_8(q, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (!ok[j]) continue;
          float4& acc4 = sum[j / SPAD];                 // ascending j = ascending slice: a fixed order
          acc4.x = __fadd_rn(acc4.x, o[j][0]); acc4.y = __fadd_rn(acc4.y, o[j][1]);
          acc4.z = __fadd_rn(acc4.z, o[j][2]); acc4.w = __fadd_rn(acc4.w, o[j][3]);
        }
      }
#pragma unroll
      for (int u = 0; u < PP; ++u)
        if (pm[u] >= 0) finish(pm[u], pn[u], sum[u]);
    }
  };
  if (S <= 1) run(std::integral_constant<int, 1>{});
  else if (S <= 2) run(std::integral_constant<int, 2>{});
  else if (S <= 4) run(std::integral_constant<int, 4>{});
  else run(std::integral_constant<int, 8>{});
}

template <int ROLE, int BM, int BN, int PASSES = 1, int TM = 2, int TN = 2, int NT = 256, class MAP = LinearRows,
          bool RESH = false /* honour Epilogue::residual_h (the fp16 instantiations) */>
__device__ __forceinline__ void store_tile_wide(const Params& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                int lane, int tid, float* tile /* [BM / PASSES][BN] in LDS */,
                                                const MAP* map = nullptr, int split = -1, int tile_id = -1) {
  // (split, tile_id): this workgroup's K slice and tile when the kernel does not take them from blockIdx.y / .x
  if (split < 0) split = blockIdx.y;
  if (tile_id < 0) tile_id = blockIdx.x;
  // PASSES > 1: the tile goes through a smaller LDS window in row bands of BM / PASSES (single-buffered kernels)
  constexpr int ROWS = BM / PASSES;
  static_assert(ROWS % (32 * TM) == 0, "a band holds whole wave tiles");
  const Epilogue& e = p.e;
  constexpr int CPR = BN / 4;                 // float4 pieces per tile row
  constexpr int PIECES = ROWS * CPR / NT;     // per thread per band
  const bool raw = gridDim.y > 1;             // split-K: raw partial into this slice's slab
  const bool has_map = map != nullptr;
  const MAP rows = has_map ? *map : MAP();    // by value: the closures below keep it in registers, not on the stack
  auto row_m = [&](int r) -> int { return has_map ? rows(r) : m0 + r; };
  // where output row m, column n lives (o) and which column the per-column epilogue operands are read at (nb)
  auto locate = [&](int m, int n, int& nb) -> size_t {
    size_t o = (size_t)m * p.ldc + n;
    nb = n;
    if (ROLE == DGRAD && p.scatter) {
      const int ow = m % p.sc_Wo, t = m / p.sc_Wo;
      const int oh = t % p.sc_Ho, b = t / p.sc_Ho;
      o = ((size_t)(b * p.sc_H + oh * p.sc_stride) * p.sc_W + ow * p.sc_stride) * p.ldc + n;
    }
    if (ROLE == FWD && p.shuffle_c) {
      const int ph = n / p.shuffle_c;
      nb = n - ph * p.shuffle_c;
      const int x = m % p.shuffle_w, t = m / p.shuffle_w;
      const int y = t % p.shuffle_h, b = t / p.shuffle_h;
      o = (((size_t)(b * 2 * p.shuffle_h + 2 * y + (ph >> 1)) * (2 * p.shuffle_w)) + 2 * x + (ph & 1)) * p.shuffle_c + nb;
    }
    return o;
  };
  // the arithmetic of the fused epilogue on one float4 piece, operands already in registers.  Every step is
  // individually rounded — no FMA contraction — so that the direct epilogue, the in-kernel split-K finishing and the
  // separate splitk_finish pass produce the same bits.
  const bool res16 = RESH && !e.residual && e.residual_h;
  const bool has_res = e.residual || res16;
  auto chain = [&](float4 v, bool has_scale, const float4& sc, const float4& bi, const float4& rr, const float4& mk,
                   const short4& mp) -> float4 {   // (mk: the fp32 gate; mp: the gate read from a 16-bit plane)
    if (p.in_shift) { const float a = pow2i(-p.in_shift); v.x *= a; v.y *= a; v.z *= a; v.w *= a; }
    if (has_scale) {
      v.x = __fmul_rn(v.x, sc.x); v.y = __fmul_rn(v.y, sc.y); v.z = __fmul_rn(v.z, sc.z); v.w = __fmul_rn(v.w, sc.w);
    }
    if (e.bias) {
      v.x = __fadd_rn(v.x, bi.x); v.y = __fadd_rn(v.y, bi.y); v.z = __fadd_rn(v.z, bi.z); v.w = __fadd_rn(v.w, bi.w);
    }
    if (has_res) {
      v.x = __fadd_rn(v.x, rr.x); v.y = __fadd_rn(v.y, rr.y); v.z = __fadd_rn(v.z, rr.z); v.w = __fadd_rn(v.w, rr.w);
    }
    if (e.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (e.mask) {
      v.x = mk.x > 0.f ? v.x : 0.f; v.y = mk.y > 0.f ? v.y : 0.f; v.z = mk.z > 0.f ? v.z : 0.f; v.w = mk.w > 0.f ? v.w : 0.f;
    }
    if (p.mask_plane) {
      v.x = mp.x > 0 ? v.x : 0.f; v.y = mp.y > 0 ? v.y : 0.f; v.z = mp.z > 0 ? v.z : 0.f; v.w = mp.w > 0 ? v.w : 0.f;
    }
    return v;
  };
  const bool row_scale = e.scale && (ROLE == WGRAD || p.scale_rows);
  // one finished float4 piece at output row m, column n, operands fetched on the spot (the in-kernel split-K fold)
  auto finish_piece = [&](int m, int n, float4 v) {
    int nb;
    const size_t o = locate(m, n, nb);
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), bi = sc, rr = sc, mk = sc;
    short4 mp = make_short4(0, 0, 0, 0);
    if (row_scale) { const float r = e.scale[m]; sc = make_float4(r, r, r, r); }
    else if (e.scale) sc = *reinterpret_cast<const float4*>(e.scale + nb);
    if (e.bias) bi = *reinterpret_cast<const float4*>(e.bias + nb);
    if (e.residual) rr = *reinterpret_cast<const float4*>(e.residual + o);
    else if (res16) rr = res_h4(e, o);
    if (e.mask) mk = *reinterpret_cast<const float4*>(e.mask + o);
    if (p.mask_plane) mp = *reinterpret_cast<const short4*>(p.mask_plane + o);
    v = chain(v, e.scale != nullptr, sc, bi, rr, mk, mp);
    if (p.C) *reinterpret_cast<float4*>(p.C + o) = v;
    if (p.out_hi) emit_planes4(p.out_hi, p.out_lo, o, v, p.out_shift);
  };
  // ---- the streaming loop.  A thread's pieces all sit in the SAME four columns (NT is a multiple of the pieces per
  // row), so the per-column operands are fetched once; the per-piece operands (residual, gate) of U pieces travel
  // together, and the NEXT group's are requested before this group's results are stored: on gfx9 stores count in
  // vmcnt like loads and the counter retires in order, so a load issued behind a store also waits for that store's
  // acknowledgement.  (The first form of this loop fetched scale, bias, residual and gate one after the other, each
  // behind an s_waitcnt vmcnt(0) that also drained the previous piece's stores: three to four memory round trips
  // per 16 bytes of output, which is what bounded the short-K layers.)
  static_assert(NT % CPR == 0, "a thread keeps its columns from piece to piece");
#ifndef JTSM_EPI_UMAX_BANDED
#define JTSM_EPI_UMAX_BANDED 2
#endif
#ifndef JTSM_EPI_UMAX_BANDED8
#define JTSM_EPI_UMAX_BANDED8 JTSM_EPI_UMAX_BANDED   // the eight-wave kernels (256 registers per wave)
#endif
  constexpr int UMAX = PASSES > 1 ? (NT == 512 ? JTSM_EPI_UMAX_BANDED8 : JTSM_EPI_UMAX_BANDED) : 4;   // (banded tiles: half the waves still hold their accumulators)
  constexpr int U = PIECES % UMAX == 0 ? UMAX : (PIECES % 2 == 0 ? 2 : 1);
  constexpr int NG = PIECES / U;
  const int col = (tid % CPR) * 4, n = n0 + col;
  const bool n_ok = n < p.N;
  int nb_col = n;
  if (ROLE == FWD && p.shuffle_c) nb_col = n - (n / p.shuffle_c) * p.shuffle_c;
  float4 sc_col = make_float4(0.f, 0.f, 0.f, 0.f), bi_col = sc_col;
  if (!raw && n_ok) {
    if (e.scale && !row_scale) sc_col = *reinterpret_cast<const float4*>(e.scale + nb_col);
    if (e.bias) bi_col = *reinterpret_cast<const float4*>(e.bias + nb_col);
  }
  const bool piped = !raw && (has_res || e.mask || p.mask_plane);
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);    // (p.colsum) this thread's four columns over its rows, in row order
  int pm[U];
  float4 prr[U], pmk[U];
  float2 pmp[U];           // (four 16-bit gate words of a plane)
  auto prefetch = [&](int pass, int g) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int row = (tid + NT * (g * U + u)) / CPR;
      int m = row_m(pass * ROWS + row);
      if (m >= p.M || !n_ok) m = -1;
      pm[u] = m;
      // a lane without a piece reads the operands' first bytes: no divergent region around the loads (the waits the
      // compiler places at such a region's edges would serialise the pieces again)
      int nb;
      const size_t o = m >= 0 ? locate(m, n, nb) : 0;
      if (e.residual) prr[u] = *reinterpret_cast<const float4*>(e.residual + o);
      else if (res16) prr[u] = res_h4(e, o);
      if (e.mask) pmk[u] = *reinterpret_cast<const float4*>(e.mask + o);
      if (p.mask_plane) pmp[u] = *reinterpret_cast<const float2*>(p.mask_plane + o);
    }
  };
  auto plane_gate = [&](const float2& w) -> short4 {
    const int a = __float_as_int(w.x), b = __float_as_int(w.y);
    return make_short4((short)(a & 0xffff), (short)(a >> 16), (short)(b & 0xffff), (short)(b >> 16));
  };
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    __syncthreads();   // every wave is done reading the last K stage / streaming the previous band
    if ((wm * TM * 32) / ROWS == pass) {
      const int r0 = wm * TM * 32 - pass * ROWS;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = r0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            tile[row * BN + (wn * TN + j) * 32 + (lane & 31)] = acc[i][j][r];
          }
    }
    if (piped && pass == 0) prefetch(0, 0);   // (behind the parking stores: the accumulators' registers are free)
    __syncthreads();
    if (!raw && !piped) {   // nothing to fetch per piece: LDS -> arithmetic -> stores, four pieces unrolled
#pragma unroll 4
      for (int it = 0; it < PIECES; ++it) {
        const int row = (tid + NT * it) / CPR;
        const int m = row_m(pass * ROWS + row);
        if (m < 0 || m >= p.M || !n_ok) continue;
        float4 sc = sc_col;
        if (row_scale) { const float r = e.scale[m]; sc = make_float4(r, r, r, r); }
        const float4 v = chain(*reinterpret_cast<const float4*>(tile + row * BN + col), e.scale != nullptr, sc, bi_col,
                               sc_col, sc_col, make_short4(0, 0, 0, 0));
        int nb;
        const size_t o = locate(m, n, nb);
        if (p.C) *reinterpret_cast<float4*>(p.C + o) = v;
        if (p.out_hi) emit_planes4(p.out_hi, p.out_lo, o, v, p.out_shift);
        csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w;
      }
      continue;
    }
    if (raw) {
#pragma unroll 4
      for (int it = 0; it < PIECES; ++it) {
        const int row = (tid + NT * it) / CPR;
        const int m = row_m(pass * ROWS + row);
        if (m < 0 || m >= p.M || !n_ok) continue;
        const float4 v = *reinterpret_cast<const float4*>(tile + row * BN + col);
        float* dst = p.slab + ((size_t)split * p.M + m) * p.ldc + n;
        if (p.tickets) st_sc1_x4(dst, v);
        else *reinterpret_cast<float4*>(dst) = v;
      }
      continue;
    }
#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      float4 out[U];
      int om[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {   // (no lane-dependent branch before the stores: see prefetch)
        om[u] = pm[u];
        const int row = (tid + NT * (g * U + u)) / CPR;
        float4 sc = sc_col;
        if (row_scale) { const float r = e.scale[om[u] < 0 ? 0 : om[u]]; sc = make_float4(r, r, r, r); }
        short4 mp = make_short4(0, 0, 0, 0);
        if (p.mask_plane) mp = plane_gate(pmp[u]);
        out[u] = chain(*reinterpret_cast<const float4*>(tile + row * BN + col), e.scale != nullptr, sc, bi_col, prr[u],
                       pmk[u], mp);
      }
      if (g + 1 < NG) prefetch(pass, g + 1);
      else if (pass + 1 < PASSES) prefetch(pass + 1, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (om[u] < 0) continue;
        int nb;
        const size_t o = locate(om[u], n, nb);
        if (p.C) *reinterpret_cast<float4*>(p.C + o) = out[u];
        if (p.out_hi) emit_planes4(p.out_hi, p.out_lo, o, out[u], p.out_shift);
        csum.x += out[u].x; csum.y += out[u].y; csum.z += out[u].z; csum.w += out[u].w;
      }
    }
  }
  if (p.colsum && !raw) {
    // the NT / CPR threads that share four columns, added in thread order: no atomics, reproducible
    __syncthreads();                                   // (the LDS window is free again)
    float4* red = reinterpret_cast<float4*>(tile);
    red[tid] = csum;
    __syncthreads();
    if (tid < CPR && n_ok) {
      float4 t = red[tid];
#pragma unroll
      for (int k = 1; k < NT / CPR; ++k) {
        const float4 q = red[tid + k * CPR];
        t.x += q.x; t.y += q.y; t.z += q.z; t.w += q.w;
      }
      *reinterpret_cast<float4*>(p.colsum + (size_t)(m0 / BM) * p.N + n) = t;
    }
  }
  if (!(raw && p.tickets)) return;
  // ---- split-K finishing by the tile's last-arriving slice: every slab in slice order, then the epilogue
  __syncthreads();                                   // (the LDS window is free again)
  if (!splitk_last_arrival(p, reinterpret_cast<int*>(tile), tile_id)) return;
  constexpr int NPIECE = BM * CPR / NT;              // float4 pieces of the whole tile per thread
  splitk_fold(p, NPIECE, [&](int k, int& m, int& n) {
    const int c = tid + NT * k;
    const int row = c / CPR, col = (c % CPR) * 4;
    m = row_m(row);
    n = n0 + col;
    return m >= 0 && m < p.M && n < p.N;
  }, finish_piece);
}

// ---- the kernel ------------------------------------------------------------------------------
// BM x BN output tile, four wavefronts, each 64x64 (2x2 MFMA tiles of 32x32).
template <int ROLE, int BM, int BN>
__global__ __launch_bounds__(256, 2) void igemm_kernel(const Params p) {
  constexpr int WN = BN / 64;
  static_assert((BM / 64) * WN == 4, "four wavefronts of 64x64");
  constexpr bool A_T = ROLE != WGRAD;  // A staged transposed (K-contiguous source)?
  constexpr bool B_T = ROLE == FWD;
  constexpr int SA = BM + (A_T ? PAD_T : PAD_D);
  constexpr int SB = BN + (B_T ? PAD_T : PAD_D);
  constexpr int A_CH = BM * BK / 4 / 256;  // 16-byte chunks per thread per K tile
  constexpr int B_CH = BN * BK / 4 / 256;
  constexpr int ACPR = BM / 4, BCPR = BN / 4;          // chunks per k-row (row-contiguous operands)
  constexpr int ARS = 256 / ACPR, BRS = 256 / BCPR;    // k-rows covered per pass

  __shared__ __attribute__((aligned(16))) float lds[BK * SA + BK * SB];
  float* As = lds;
  float* Bs = lds + BK * SA;

  const ConvShape& s = p.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: blocks b and b+8 share an XCD (and its L2), so hand each XCD a
  // contiguous run of tiles; within a run tiles sweep N first, re-using the A rows.
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntn * ntm;
  int tile = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  int kbeg = 0, kend = p.K;
  if (gridDim.y > 1) {
    kbeg = blockIdx.y * p.ktiles_per_split * BK;
    kend = min(p.K, kbeg + p.ktiles_per_split * BK);
    if (kbeg >= kend && ROLE == WGRAD) return;  // FWD/DGRAD slices must still write their (zero) slab
  }

  // ---- per-thread addressing state
  PixelRow arow[A_T ? A_CH : 1];
  TapState atap = {};                      // FWD/DGRAD A: this thread's chunk along K
  CoState bco[ROLE == DGRAD ? B_CH : 1];   // DGRAD B: one k-row per chunk
  int wg_k = kbeg;                           // WGRAD: first pixel row of the current K tile
  int wg_tap_off = 0, wg_kh = 0, wg_kw = 0; bool wg_col_ok = false;  // WGRAD B column (tap, ci)
  if (A_T) {
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int m = m0 + tid / 8 + 32 * j;
      arow[j] = ROLE == FWD ? fwd_pixel(s, m, p.M) : dgrad_pixel(s, m, p.M);
    }
    atap = tap_init(kbeg + 4 * (tid & 7), ROLE == FWD ? s.Cin : s.Cout, s.KW);
  }
  if (ROLE == DGRAD) {
#pragma unroll
    for (int j = 0; j < B_CH; ++j) bco[j] = co_init(kbeg + tid / BCPR + BRS * j, s.Cout);
  }
  if (ROLE == WGRAD) {
    const int col = n0 + 4 * (tid % BCPR);
    wg_col_ok = col < p.N;
    const int cc = wg_col_ok ? col : 0;
    const int tap = cc / s.Cin;
    wg_tap_off = cc - tap * s.Cin;
    wg_kh = tap / s.KW;
    wg_kw = tap - wg_kh * s.KW;
  }
  const int bk_fwd = kbeg + 4 * (tid & 7);   // FWD B chunk k (advances by BK)
  int fwd_k = bk_fwd;

  float4 ra[A_CH], rb[B_CH];
  auto fetch = [&]() {
    if (ROLE == FWD) {
#pragma unroll
      for (int j = 0; j < A_CH; ++j) ra[j] = load_or_zero(p.A, fwd_a_off(s, arow[j], atap, kend));
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        const int n = n0 + tid / 8 + 32 * j;
        rb[j] = load_or_zero(p.B, (n < p.N && fwd_k < kend) ? n * p.K + fwd_k : -1);
      }
      tap_advance(atap, BK, s.Cin, s.KW);
      fwd_k += BK;
    } else if (ROLE == DGRAD) {
#pragma unroll
      for (int j = 0; j < A_CH; ++j) ra[j] = load_or_zero(p.A, dgrad_a_off(s, arow[j], atap, kend));
      const int col = n0 + 4 * (tid % BCPR);
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        const bool ok = bco[j].k < kend && col < p.N;
        float4 v = load_or_zero(p.B, ok ? (bco[j].co * (s.KH * s.KW) + bco[j].tap) * s.Cin + col : -1);
        if (p.kscale) {
          const float ks = p.kscale[ok ? bco[j].co : 0];
          v.x *= ks; v.y *= ks; v.z *= ks; v.w *= ks;
        }
        rb[j] = v;
        co_advance(bco[j], BK, s.Cout);
      }
      tap_advance(atap, BK, s.Cout, s.KW);
    } else {
      // (conditional loads here: for this role hipcc's branchy form measured ~15 % faster than the
      //  select form on the short K sweeps that take this kernel — fewer live registers per chunk)
      const int acol = m0 + 4 * (tid % ACPR);
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const int k = wg_k + tid / ACPR + ARS * j;
        ra[j] = (k < kend && acol < p.M) ? ldg4(p.A + (size_t)k * s.Cout + acol) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        const int k = wg_k + tid / BCPR + BRS * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wg_col_ok && k < kend) {
          const int ow = k % s.Wo, t = k / s.Wo;
          const int oh = t % s.Ho, b = t / s.Ho;
          const int ih = oh * s.stride - s.pad + wg_kh * s.dil, iw = ow * s.stride - s.pad + wg_kw * s.dil;
          if ((unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W)
            v = ldg4(p.B + ((size_t)(b * s.H + ih) * s.W + iw) * s.Cin + wg_tap_off);
        }
        rb[j] = v;
      }
      wg_k += BK;
    }
  };
  auto stage = [&]() {
    if (A_T) {
      const int kq = 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        float* d = As + kq * SA + tid / 8 + 32 * j;
        d[0] = ra[j].x; d[SA] = ra[j].y; d[2 * SA] = ra[j].z; d[3 * SA] = ra[j].w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < A_CH; ++j)
        *reinterpret_cast<float4*>(As + (tid / ACPR + ARS * j) * SA + 4 * (tid % ACPR)) = ra[j];
    }
    if (B_T) {
      const int kq = 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        float* d = Bs + kq * SB + tid / 8 + 32 * j;
        d[0] = rb[j].x; d[SB] = rb[j].y; d[2 * SB] = rb[j].z; d[3 * SB] = rb[j].w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < B_CH; ++j)
        *reinterpret_cast<float4*>(Bs + (tid / BCPR + BRS * j) * SB + 4 * (tid % BCPR)) = rb[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch();
  const float* aw = As + (lane >> 5) * SA + wm * 64 + (lane & 31);
  const float* bw = Bs + (lane >> 5) * SB + wn * 64 + (lane & 31);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // everyone done reading the previous tile
    stage();
    __syncthreads();
    if (ROLE != WGRAD || k0 + BK < kend) fetch();  // next tile: in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = aw[kk * SA], a1 = aw[kk * SA + 32];
      const float b0 = bw[kk * SB], b1 = bw[kk * SB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  store_tile<ROLE>(p, acc, m0, n0, wm, wn, lane);
}

// ================================================================================================
// igemm_dma_kernel — the same contraction with direct-to-LDS loads.
//
// `global_load_lds_dwordx4` writes 64 lanes x 16 B = 1 KiB of LDS per instruction, lane-linearly, with
// a free per-lane GLOBAL address; nothing passes through VGPRs, so the 32 staging registers and all
// ds_write instructions of the register-staged kernel disappear, tiles are double-buffered (the DMA
// of K tile t+1 flies during the MFMAs of tile t, one barrier per tile), and out-of-range chunks are
// pointed at a 16-byte zero page instead of being branched around.
//   K-contiguous operands (FWD/DGRAD activations, FWD weights) keep their [row][32 k] shape in LDS,
//     16-byte chunk c of row r at slot c ^ ((r >> 1) & 7) — the swizzle is applied to the SOURCE address
//     (lane l of a DMA writes slot l&7 of row l>>3, so it fetches chunk (l&7) ^ swz) and again on the
//     read, which is a conflict-free ds_read_b128 of 4 consecutive k.  An MFMA 32x32x2 step takes its
//     two k from the two half-waves, so half h consumes k = 8q+4h+s in step (q,s): a permutation of
//     the tile's k order applied identically to A and B.
//   Row-contiguous operands (DGRAD weights, WGRAD both) are already k-major in memory: rows of 512 B
//     land unpadded, fragments are conflict-free ds_read_b32 (consecutive lanes, consecutive columns).
// Requires every K tile to lie inside one filter tap: kernel 1x1, or channels % 32 == 0 (everything on
// the JTSM path but the RGB stem, which keeps the register-staged kernel).
__device__ __attribute__((aligned(16))) float g_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* g, float* lds_uniform_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_uniform_base, 16, 0, 0);
}

template <int ROLE, int BM, int BN, int NBUF>
__global__ __launch_bounds__(256, NBUF == 1 ? 4 : 2) void igemm_dma_kernel(const Params p) {
  constexpr int WN = BN / 64;
  static_assert((BM / 64) * WN == 4, "four wavefronts of 64x64");
  constexpr bool A_R = ROLE != WGRAD;  // A keeps [row][k] (K-contiguous source)
  constexpr bool B_R = ROLE == FWD;
  constexpr int A_FL = BM * BK, B_FL = BN * BK;
  constexpr int A_INS = BM / 32, B_INS = BN / 32;  // DMA instructions per wavefront per K tile
  __shared__ __attribute__((aligned(16))) float lds[NBUF * (A_FL + B_FL)];

  const ConvShape& s = p.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, li = lane & 31;

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntn * ntm;
  int tile = blockIdx.x;
  {
    const int q = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  int kbeg = 0, kend = p.K;
  if (gridDim.y > 1) {
    kbeg = blockIdx.y * p.ktiles_per_split * BK;
    kend = min(p.K, kbeg + p.ktiles_per_split * BK);
    if (kbeg >= kend && ROLE == WGRAD) return;
  }
  const int ntile_k = kbeg < kend ? (kend - kbeg + BK - 1) / BK : 0;

  // ---- addressing state -----------------------------------------------------------------------
  // K-contiguous operands: per DMA instruction j this lane owns row (wave*INS + j)*8 + lane/8 and the
  // chunk (lane&7) ^ swz(row) of the K tile.  The tap (kh, kw) and first channel of a K tile are
  // wave-uniform and advance incrementally.
  const int Cdim = ROLE == FWD ? s.Cin : s.Cout;   // channels of the K-contiguous activation
  int t_kh, t_kw, t_c;                             // uniform tap state of the current K tile
  {
    const int tap = kbeg / Cdim;
    t_c = kbeg - tap * Cdim;
    t_kh = tap / s.KW;
    t_kw = tap - t_kh * s.KW;
  }
  PixelRow arow[A_R ? A_INS : 1];
  int a_chunk[A_R ? A_INS : 1];
  if (A_R) {
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
      const int r = (wave * A_INS + j) * 8 + (lane >> 3);
      arow[j] = ROLE == FWD ? fwd_pixel(s, m0 + r, p.M) : dgrad_pixel(s, m0 + r, p.M);
      a_chunk[j] = 4 * ((lane & 7) ^ ((r >> 1) & 7));
    }
  }
  int b_off[B_R ? B_INS : 1], b_chunk[B_R ? B_INS : 1];   // FWD weights: n*K, chunk
  if (B_R) {
#pragma unroll
    for (int j = 0; j < B_INS; ++j) {
      const int r = (wave * B_INS + j) * 8 + (lane >> 3);
      b_off[j] = (n0 + r) < p.N ? (n0 + r) * p.K : -1;
      b_chunk[j] = 4 * ((lane & 7) ^ ((r >> 1) & 7));
    }
  }
  // row-contiguous operands: per DMA instruction j this lane owns k-row (wave*INS + j)*RPI + lane/CPR and
  // columns 4*(lane % CPR) ..+3.
  constexpr int ACPR = BM / 4, BCPR = BN / 4, ARPI = 64 / (ACPR < 64 ? ACPR : 64), BRPI = 64 / (BCPR < 64 ? BCPR : 64);
  static_assert(ACPR <= 64 && BCPR <= 64, "tile rows wider than one DMA instruction are not laid out here");
  PixState bpix[ROLE == WGRAD ? B_INS : 1];
  int wg_ci = 0, wg_kh = 0, wg_kw = 0;
  bool wg_col_ok = false;
  if (ROLE == WGRAD) {
#pragma unroll
    for (int j = 0; j < B_INS; ++j)
      bpix[j] = pix_init(kbeg + (wave * B_INS + j) * BRPI + lane / BCPR, s.Ho, s.Wo);
    const int col = n0 + 4 * (lane % BCPR);
    wg_col_ok = col < p.N;
    const int cc = wg_col_ok ? col : 0;
    const int tap = cc / s.Cin;
    wg_ci = cc - tap * s.Cin;
    wg_kh = tap / s.KW;
    wg_kw = tap - wg_kh * s.KW;
  }

  auto issue = [&](int k0, int buf) {
    float* Ab = lds + buf * (A_FL + B_FL);
    float* Bb = Ab + A_FL;
    if (A_R) {
#pragma unroll
      for (int j = 0; j < A_INS; ++j) {
        const PixelRow& r = arow[j];
        bool ok = r.ok && (k0 + a_chunk[j]) < kend;
        int off;
        if (ROLE == FWD) {
          const int ih = r.h0 + t_kh * s.dil, iw = r.w0 + t_kw * s.dil;
          ok = ok && (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
          off = ((r.b * s.H + ih) * s.W + iw) * s.Cin + t_c + a_chunk[j];
        } else {
          const int th = r.h0 - t_kh * s.dil, tw = r.w0 - t_kw * s.dil;
          int oh = th, ow = tw;
          ok = ok && th >= 0 && tw >= 0;
          if (s.stride != 1) {
            oh = th / s.stride;
            ow = tw / s.stride;
            ok = ok && oh * s.stride == th && ow * s.stride == tw;
          }
          ok = ok && oh < s.Ho && ow < s.Wo;
          off = ((r.b * s.Ho + oh) * s.Wo + ow) * s.Cout + t_c + a_chunk[j];
        }
        dma16(ok ? p.A + off : g_zero_page, Ab + (wave * A_INS + j) * 256);
      }
    } else {  // WGRAD A: dY[pixel k][cout]
      const int col = m0 + 4 * (lane % ACPR);
#pragma unroll
      for (int j = 0; j < A_INS; ++j) {
        const int k = k0 + (wave * A_INS + j) * ARPI + lane / ACPR;
        const bool ok = k < kend && col < p.M;
        dma16(ok ? p.A + k * s.Cout + col : g_zero_page, Ab + (wave * A_INS + j) * 256);
      }
    }
    if (B_R) {
#pragma unroll
      for (int j = 0; j < B_INS; ++j) {
        const bool ok = b_off[j] >= 0 && (k0 + b_chunk[j]) < kend;
        dma16(ok ? p.B + b_off[j] + k0 + b_chunk[j] : g_zero_page, Bb + (wave * B_INS + j) * 256);
      }
    } else if (ROLE == DGRAD) {  // rows k = (tap, co) of W as [K][Cin]
      const int col = n0 + 4 * (lane % BCPR);
#pragma unroll
      for (int j = 0; j < B_INS; ++j) {
        const int kr = (wave * B_INS + j) * BRPI + lane / BCPR;
        const bool ok = (k0 + kr) < kend && col < p.N;
        const int tap = t_kh * s.KW + t_kw;
        dma16(ok ? p.B + ((t_c + kr) * (s.KH * s.KW) + tap) * s.Cin + col : g_zero_page,
              Bb + (wave * B_INS + j) * 256);
      }
    } else {  // WGRAD B: X[pix(k, tap)][ci]
#pragma unroll
      for (int j = 0; j < B_INS; ++j) {
        const PixState& q = bpix[j];
        const int ih = q.oh * s.stride - s.pad + wg_kh * s.dil, iw = q.ow * s.stride - s.pad + wg_kw * s.dil;
        const bool ok = wg_col_ok && q.k < kend && (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
        dma16(ok ? p.B + ((q.b * s.H + ih) * s.W + iw) * s.Cin + wg_ci : g_zero_page,
              Bb + (wave * B_INS + j) * 256);
        pix_advance(bpix[j], BK, s.Ho, s.Wo);
      }
    }
    if (ROLE != WGRAD) {  // next K tile: same tap or the next one (tiles never straddle taps)
      t_c += BK;
      if (t_c >= Cdim) {
        t_c -= Cdim;
        if (++t_kw == s.KW) { t_kw = 0; ++t_kh; }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read bases
  int a_row_off[2], a_swz[2], b_row_off[2], b_swz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ar = wm * 64 + t * 32 + li, br = wn * 64 + t * 32 + li;
    a_row_off[t] = A_R ? ar * BK : ar;
    a_swz[t] = (ar >> 1) & 7;
    b_row_off[t] = B_R ? br * BK : br;
    b_swz[t] = (br >> 1) & 7;
  }

  if (NBUF == 2 && ntile_k > 0) issue(kbeg, 0);
  for (int t = 0; t < ntile_k; ++t) {
    if (NBUF == 1) {  // single buffer (32-40 KiB -> 4 workgroups per CU cover each other's DMA waits)
      __syncthreads();                                // everyone done reading the previous tile
      issue(kbeg + t * BK, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMAs of tile t have landed
    __syncthreads();                                  // ... everyone's; and buffer (t+1)&1 is free
    if (NBUF == 2 && t + 1 < ntile_k) issue(kbeg + (t + 1) * BK, (t + 1) & 1);
    const float* Ab = lds + (NBUF == 2 ? (t & 1) : 0) * (A_FL + B_FL);
    const float* Bb = Ab + A_FL;
    if (ROLE == WGRAD) {
#pragma unroll
      for (int kk = 0; kk < BK; kk += 2) {
        const float a0 = Ab[(kk + half) * BM + a_row_off[0]], a1 = Ab[(kk + half) * BM + a_row_off[1]];
        const float b0 = Bb[(kk + half) * BN + b_row_off[0]], b1 = Bb[(kk + half) * BN + b_row_off[1]];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 2 * q + half;
        const float4 a0 = *reinterpret_cast<const float4*>(Ab + a_row_off[0] + ((c ^ a_swz[0]) << 2));
        const float4 a1 = *reinterpret_cast<const float4*>(Ab + a_row_off[1] + ((c ^ a_swz[1]) << 2));
        float b0[4], b1[4];
        if (B_R) {
          const float4 x0 = *reinterpret_cast<const float4*>(Bb + b_row_off[0] + ((c ^ b_swz[0]) << 2));
          const float4 x1 = *reinterpret_cast<const float4*>(Bb + b_row_off[1] + ((c ^ b_swz[1]) << 2));
          b0[0] = x0.x; b0[1] = x0.y; b0[2] = x0.z; b0[3] = x0.w;
          b1[0] = x1.x; b1[1] = x1.y; b1[2] = x1.z; b1[3] = x1.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            b0[e] = Bb[(4 * c + e) * BN + b_row_off[0]];
            b1[e] = Bb[(4 * c + e) * BN + b_row_off[1]];
          }
        }
        const float av0[4] = {a0.x, a0.y, a0.z, a0.w}, av1[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], b0[e], acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[e], b1[e], acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], b0[e], acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[e], b1[e], acc[1][1], 0, 0, 0);
        }
      }
    }
  }
  store_tile<ROLE>(p, acc, m0, n0, wm, wn, lane);
}

// Which kernel?  The direct-to-LDS kernel (64-80 KiB LDS, 2 workgroups per CU, deep prefetch) wins when a
// workgroup sweeps many K tiles; short sweeps (split-K slices, 1x1 convolutions with few channels) are
// dominated by prologue/epilogue latency and do better with the register-staged kernel's 3 workgroups
// per CU.  Measured on MI355X (tools/sweeps/bench_conv.py): crossover around 24 K tiles per workgroup.
constexpr int kDmaMinKTiles = 24;
constexpr int kWgradTargetBlocks = 1024;  // WGRAD pixel-axis split: workgroups aimed at (atomic traffic grows with it)
constexpr bool kShortSweepDma = true;   // short sweeps: single-buffered DMA kernel instead of the register-staged one

// Every K tile inside one filter tap?  (kernel 1x1, or channel count a multiple of the K tile.)
inline bool dma_eligible(int role, const Params& p) {
  const ConvShape& s = p.s;
  const int taps = s.KH * s.KW;
  if (role == WGRAD) return true;
  if (role == DGRAD && p.kscale) return false;  // the DMA path cannot scale weight rows on the fly
  const int c = role == FWD ? s.Cin : s.Cout;
  return taps == 1 || c % BK == 0;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Fold the split-K slabs in slice order (deterministic) and apply the fused epilogue.  VEC = 4 when the
// row length and leading dimension are multiples of 4 (every layer of the model), else 1.
// G > 1 (many slices of a SMALL result — the weight gradients of narrow layers over 10^5 pixels): G threads share an
// output piece, thread g adding slices g, g + G, ... and the G partial sums being added in order g = 0 .. G-1 through
// LDS; with one thread per piece those launches were a few thousand threads each walking 64-256 dependent loads.
// G > 1: G threads share an output piece (the first form of the kernel, unchanged).
template <int VEC, int G>
__global__ __launch_bounds__(256) void splitk_finish_shared(const Params p, int splits, int scale_by_row) {
  if (p.bias_out) {   // the weight gradient's bias partials: 32 lanes per output channel, a fixed reduction tree
    const int r = threadIdx.x >> 5, l = threadIdx.x & 31;
    for (int mb = blockIdx.x; mb * 8 < p.M; mb += gridDim.x) {
      const int m = mb * 8 + r;
      float t = 0.f;
      if (m < p.M)
        for (int s = l; s < splits; s += 32) t = __fadd_rn(t, p.bias_slab[(size_t)s * p.M + m]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) t = __fadd_rn(t, __shfl_xor(t, o, 32));
      if (l == 0 && m < p.M) p.bias_out[m] = t;
    }
  }
  const int nv = p.N / VEC;
  const long total = (long)p.M * nv;
  const Epilogue& e = p.e;
  const float alpha = pow2i(-p.in_shift);
  constexpr int PPB = 256 / G;                       // output pieces per workgroup and trip
  __shared__ float red[G > 1 ? 256 * VEC : 1];
  const int pl = threadIdx.x % PPB, g = threadIdx.x / PPB;
  for (long i0 = (long)blockIdx.x * PPB; i0 < total; i0 += (long)gridDim.x * PPB) {   // (trip count is block-uniform)
    const long i = i0 + pl;
    const bool live = i < total;
    const int n = live ? (int)(i % nv) * VEC : 0;
    const long m = live ? i / nv : 0;
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = 0.f;
    if (live)
      for (int s = g; s < splits; s += G) {
        const float* src = p.slab + ((size_t)s * p.M + m) * p.ldc + n;
        if (VEC == 4) {
          const float4 t = *reinterpret_cast<const float4*>(src);
          v[0] = __fadd_rn(v[0], t.x); v[1 % VEC] = __fadd_rn(v[1 % VEC], t.y);
          v[2 % VEC] = __fadd_rn(v[2 % VEC], t.z); v[3 % VEC] = __fadd_rn(v[3 % VEC], t.w);
        } else {
          v[0] += src[0];
        }
      }
    if (G > 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[(g * PPB + pl) * VEC + j] = v[j];
      __syncthreads();
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float t = 0.f;
          for (int k = 0; k < G; ++k) t = __fadd_rn(t, red[(k * PPB + pl) * VEC + j]);
          v[j] = t;
        }
      }
      __syncthreads();
      if (g != 0) continue;
    }
    if (!live) continue;
    size_t o = (size_t)m * p.ldc + n;
    if (p.scatter) {
      const int ow = (int)(m % p.sc_Wo);
      const long t = m / p.sc_Wo;
      const int oh = (int)(t % p.sc_Ho), b = (int)(t / p.sc_Ho);
      o = ((size_t)(b * p.sc_H + oh * p.sc_stride) * p.sc_W + ow * p.sc_stride) * p.ldc + n;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float x = v[j] * alpha;                                   // (a power of two: exact)
      if (e.scale) x = __fmul_rn(x, e.scale[scale_by_row ? m : n + j]);
      if (e.bias) x = __fadd_rn(x, e.bias[n + j]);
      if (e.residual) x = __fadd_rn(x, e.residual[o + j]);
      else if (e.residual_h) x = __fadd_rn(x, res_h1(e, o + j));
      if (e.relu) x = fmaxf(x, 0.f);
      if (e.mask) x = e.mask[o + j] > 0.f ? x : 0.f;
      if (p.mask_plane) x = (short)p.mask_plane[o + j] > 0 ? x : 0.f;
      v[j] = x;
    }
    if (VEC == 4) {
      const float4 out = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
      if (p.C) *reinterpret_cast<float4*>(p.C + o) = out;
      if (p.out_hi) emit_planes4(p.out_hi, p.out_lo, o, out, p.out_shift);
    } else {
      p.C[o] = v[0];
    }
  }
}

// G == 1: one thread per output piece; the epilogue's operands and up to four slices' loads travel together.
template <int VEC, int G = 1>
__device__ __forceinline__ void splitk_finish_body(const Params& p, int splits, int scale_by_row) {
  if (p.bias_out) {   // the weight gradient's bias partials: 32 lanes per output channel, a fixed reduction tree
    const int r = threadIdx.x >> 5, l = threadIdx.x & 31;
    for (int mb = blockIdx.x; mb * 8 < p.M; mb += gridDim.x) {
      const int m = mb * 8 + r;
      float t = 0.f;
      if (m < p.M)
        for (int s = l; s < splits; s += 32) t = __fadd_rn(t, p.bias_slab[(size_t)s * p.M + m]);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) t = __fadd_rn(t, __shfl_xor(t, o, 32));
      if (l == 0 && m < p.M) p.bias_out[m] = t;
    }
  }
  const int nv = p.N / VEC;
  const long total = (long)p.M * nv;
  const Epilogue& e = p.e;
  const float alpha = pow2i(-p.in_shift);
  constexpr int PPB = 256 / G;                       // output pieces per workgroup and trip
  __shared__ float red[G > 1 ? 256 * VEC : 1];
  const int pl = threadIdx.x % PPB, g = threadIdx.x / PPB;
  for (long i0 = (long)blockIdx.x * PPB; i0 < total; i0 += (long)gridDim.x * PPB) {   // (trip count is block-uniform)
    const long i = i0 + pl;
    const bool live = i < total;
    const int n = live ? (int)(i % nv) * VEC : 0;
    const long m = live ? i / nv : 0;
    float v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = 0.f;
    // the epilogue's own operands are requested first (by the thread that will apply them): they travel with the slabs
    size_t o = (size_t)m * p.ldc + n;
    if (p.scatter) {
      const int ow = (int)(m % p.sc_Wo);
      const long t = m / p.sc_Wo;
      const int oh = (int)(t % p.sc_Ho), b = (int)(t / p.sc_Ho);
      o = ((size_t)(b * p.sc_H + oh * p.sc_stride) * p.sc_W + ow * p.sc_stride) * p.ldc + n;
    }
    float e_sc[VEC], e_bi[VEC], e_rr[VEC], e_mk[VEC];
    short e_mp[VEC];
    if (G == 1 && live) {   // (G > 1 — many slices of a small result — fetches them after the fold: measured faster)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (e.scale) e_sc[j] = e.scale[scale_by_row ? m : n + j];
        if (e.bias) e_bi[j] = e.bias[n + j];
      }
      if (VEC == 4) {   // (16-byte alignment of residual / mask and 8-byte alignment of the gate plane: finish_split)
        if (e.residual || e.residual_h) {
          const float4 t = e.residual ? *reinterpret_cast<const float4*>(e.residual + o) : res_h4(e, o);
          e_rr[0] = t.x; e_rr[1 % VEC] = t.y; e_rr[2 % VEC] = t.z; e_rr[3 % VEC] = t.w;
        }
        if (e.mask) {
          const float4 t = *reinterpret_cast<const float4*>(e.mask + o);
          e_mk[0] = t.x; e_mk[1 % VEC] = t.y; e_mk[2 % VEC] = t.z; e_mk[3 % VEC] = t.w;
        }
        if (p.mask_plane) {
          const short4 t = *reinterpret_cast<const short4*>(p.mask_plane + o);
          e_mp[0] = t.x; e_mp[1 % VEC] = t.y; e_mp[2 % VEC] = t.z; e_mp[3 % VEC] = t.w;
        }
      } else {
        if (e.residual) e_rr[0] = e.residual[o];
        else if (e.residual_h) e_rr[0] = res_h1(e, o);
        if (e.mask) e_mk[0] = e.mask[o];
        if (p.mask_plane) e_mp[0] = (short)p.mask_plane[o];
      }
    }
    if (live) {
      int s = g;
      if (VEC == 4 && G == 1) {
        // UF slices' loads travel together (left as one load behind an s_waitcnt per slice, the fold was a chain of
        // memory round trips: ~3.5 TB/s of cache-resident slabs); the sums are still formed in slice order.
        const size_t slab_stride = (size_t)p.M * p.ldc;
        const float* src = p.slab + ((size_t)s * p.M + m) * p.ldc + n;
        auto fold = [&](auto uf_c) {
          constexpr int UF = decltype(uf_c)::value;
          float4 t[UF];
#pragma unroll
          for (int u = 0; u < UF; ++u) t[u] = *reinterpret_cast<const float4*>(src + (size_t)u * G * slab_stride);
#pragma unroll
          for (int u = 0; u < UF; ++u) {
            v[0] = __fadd_rn(v[0], t[u].x); v[1 % VEC] = __fadd_rn(v[1 % VEC], t[u].y);
            v[2 % VEC] = __fadd_rn(v[2 % VEC], t[u].z); v[3 % VEC] = __fadd_rn(v[3 % VEC], t[u].w);
          }
          src += (size_t)UF * G * slab_stride;
          s += UF * G;
        };
        while (s + 3 * G < splits) fold(std::integral_constant<int, 4>{});
        if (s + G < splits) fold(std::integral_constant<int, 2>{});
        if (s < splits) fold(std::integral_constant<int, 1>{});
      } else if (VEC == 4) {   // (G threads per piece: a thread adds few slices; measured slower unrolled)
        for (; s < splits; s += G) {
          const float4 t = *reinterpret_cast<const float4*>(p.slab + ((size_t)s * p.M + m) * p.ldc + n);
          v[0] = __fadd_rn(v[0], t.x); v[1 % VEC] = __fadd_rn(v[1 % VEC], t.y);
          v[2 % VEC] = __fadd_rn(v[2 % VEC], t.z); v[3 % VEC] = __fadd_rn(v[3 % VEC], t.w);
        }
      } else {
        for (; s < splits; s += G) v[0] += p.slab[((size_t)s * p.M + m) * p.ldc + n];
      }
    }
    if (G > 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) red[(g * PPB + pl) * VEC + j] = v[j];
      __syncthreads();
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float t = 0.f;
          for (int k = 0; k < G; ++k) t = __fadd_rn(t, red[(k * PPB + pl) * VEC + j]);
          v[j] = t;
        }
      }
      __syncthreads();
      if (g != 0) continue;
    }
    if (!live) continue;
    if (G > 1) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        if (e.scale) e_sc[j] = e.scale[scale_by_row ? m : n + j];
        if (e.bias) e_bi[j] = e.bias[n + j];
        if (e.residual) e_rr[j] = e.residual[o + j];
        else if (e.residual_h) e_rr[j] = res_h1(e, o + j);
        if (e.mask) e_mk[j] = e.mask[o + j];
        if (p.mask_plane) e_mp[j] = (short)p.mask_plane[o + j];
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float x = v[j] * alpha;                                   // (a power of two: exact)
      if (e.scale) x = __fmul_rn(x, e_sc[j]);
      if (e.bias) x = __fadd_rn(x, e_bi[j]);
      if (e.residual || e.residual_h) x = __fadd_rn(x, e_rr[j]);
      if (e.relu) x = fmaxf(x, 0.f);
      if (e.mask) x = e_mk[j] > 0.f ? x : 0.f;
      if (p.mask_plane) x = e_mp[j] > 0 ? x : 0.f;
      v[j] = x;
    }
    if (VEC == 4) {
      const float4 out = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
      if (p.C) *reinterpret_cast<float4*>(p.C + o) = out;
      if (p.out_hi) emit_planes4(p.out_hi, p.out_lo, o, out, p.out_shift);
    } else {
      p.C[o] = v[0];
    }
  }
}

template <int VEC, int G = 1>
__global__ __launch_bounds__(256) void splitk_finish(const Params p, int splits, int scale_by_row) {
  splitk_finish_body<VEC, G>(p, splits, scale_by_row);
}

// How many K slices for an (ntiles, ktiles) problem: aim at >= 3 workgroups per CU, keep >= 4 K tiles
// (128 k) per slice, at most 16 slices.
inline int plan_splits(int ntiles, int ktiles, int target = 768) {
  if (ntiles >= 512 || ntiles <= 0) return 1;  // ntiles == 0: an empty batch
  int s = target >= 768 ? ceil_div(target, ntiles) : target / ntiles;   // (smaller targets: never exceed them)
  if (s > ktiles / 4) s = ktiles / 4;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}

// Timing hook (bench.py's roofline leg): an event handed in through jtsm_conv_set_mid_event is recorded right
// after the NEXT contraction kernel of this thread is launched — before any split-K finishing pass — so the
// caller can time that kernel alone, as rocprofv3 reports it.
static thread_local hipEvent_t g_mid_event = nullptr;
inline void record_mid(hipStream_t st) {
  if (g_mid_event) {
    (void)hipEventRecord(g_mid_event, st);
    g_mid_event = nullptr;
  }
}

// Ticket counters of the in-kernel split-K finishing (Params::tickets): kTicketCap zero-initialised ints per
// (device, stream) that has launched a split contraction — the finishing workgroup re-arms its counter, so the buffer
// is cleared once, when it is created.  Two launches on the SAME stream run in order; different streams get their own.
constexpr int kTicketCap = 4096;
static std::atomic<int> g_splitk_fused_override{-1};   // jtsm_conv_set_splitk_fused: -1 = follow the environment
inline bool splitk_fused_enabled() {
  // default OFF: measured on MI355X (round 2, bench.py 20 steps): 44.2 ms/step fused against 30.5 ms with the separate
  // pass — write-through slab stores and bypassing loads put the ~6 GB/step of slab traffic on HBM, where the
  // write-back path keeps it in L2 / Infinity Cache until splitk_finish has read it.  Kept as a measured alternative.
  static const bool on = [] { const char* e = getenv("JTSM_SPLITK_FUSED"); return e && atoi(e) != 0; }();
  const int o = g_splitk_fused_override.load();
  return o < 0 ? on : o != 0;
}
inline int* tickets_for(hipStream_t st) {
  struct Slot { int dev; hipStream_t st; int* buf; };
  static Slot slots[64];
  static int nslots = 0;
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nslots; ++i)
    if (slots[i].dev == dev && slots[i].st == st) return slots[i].buf;
  if (nslots == 64) return nullptr;
  int* buf = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&buf), kTicketCap * sizeof(int)) != hipSuccess) return nullptr;
  if (hipMemset(buf, 0, kTicketCap * sizeof(int)) != hipSuccess) { (void)hipFree(buf); return nullptr; }
  slots[nslots++] = {dev, st, buf};
  return buf;
}
// Decide how a split launch is finished: sets p.tickets (and returns true) when the kernel itself will do it.
inline bool use_fused_finish(Params& p, int ntiles, int splits, hipStream_t st) {
  p.tickets = nullptr;
  // JTSM_SPLITK_FUSED_MAX_MB: finish inside the kernel only where all slabs together stay below this size (sweeps)
  static const long fused_max_bytes = [] { const char* e = getenv("JTSM_SPLITK_FUSED_MAX_MB"); return e ? atol(e) << 20 : 0L; }();
  const bool small = g_splitk_fused_override.load() < 0 && (long)splits * p.M * p.ldc * (long)sizeof(float) <= fused_max_bytes;
  if (splits <= 1 || !p.wide || ntiles > kTicketCap || !(splitk_fused_enabled() || small)) return false;
  // (the separate pass adds 32+ slices of a small result in a different — still fixed — order: keep those on it, so
  // that the two ways of finishing stay bit-identical wherever both exist)
  if (splits >= 32 && (long)p.M * (p.N / 4) <= (1L << 18)) return false;
  p.tickets = tickets_for(st);
  return p.tickets != nullptr;
}

inline int finish_split(const Params& p, int splits, hipStream_t st, int scale_by_row = 0) {
  scale_by_row = scale_by_row || p.scale_rows;
  const bool vec = p.N % 4 == 0 && p.ldc % 4 == 0 && aligned16(p.C) && aligned16(p.slab) &&
                   (!p.e.residual || aligned16(p.e.residual)) && (!p.e.mask || aligned16(p.e.mask)) &&
                   ((uintptr_t)p.mask_plane & 7) == 0;
  const long total = (long)p.M * (vec ? p.N / 4 : p.N);
  if (vec && splits >= 32 && total <= (1L << 18)) {   // many slices of a result below 4 MB: 16 threads per piece
    const long nb = (total + 15) / 16;
    hipLaunchKernelGGL((splitk_finish_shared<4, 16>), dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, st, p, splits,
                       scale_by_row);
    JTSM_CHECK_LAUNCH("splitk_finish");
    return JTSM_OK;
  }
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  if (vec) hipLaunchKernelGGL(splitk_finish<4>, dim3(blocks), dim3(256), 0, st, p, splits, scale_by_row);
  else hipLaunchKernelGGL(splitk_finish<1>, dim3(blocks), dim3(256), 0, st, p, splits, scale_by_row);
  JTSM_CHECK_LAUNCH("splitk_finish");
  return JTSM_OK;
}

template <int ROLE, int BM, int BN>
int launch_split(Params& p, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int ntiles = ceil_div(p.N, BN) * ceil_div(p.M, BM);
  const int ktiles = ceil_div(p.K, BK);
  int splits = plan_splits(ntiles, ktiles);
  if (splits > 1 && (size_t)splits * p.M * p.ldc * sizeof(float) > workspace_bytes) splits = 1;
  const bool dma = dma_eligible(ROLE, p) && ceil_div(ktiles, splits) >= kDmaMinKTiles;
  const bool dma1 = !dma && dma_eligible(ROLE, p) && kShortSweepDma;
  if (splits <= 1) {
    if (dma) hipLaunchKernelGGL((igemm_dma_kernel<ROLE, BM, BN, 2>), dim3(ntiles, 1), dim3(256), 0, st, p);
    else if (dma1) hipLaunchKernelGGL((igemm_dma_kernel<ROLE, BM, BN, 1>), dim3(ntiles, 1), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((igemm_kernel<ROLE, BM, BN>), dim3(ntiles, 1), dim3(256), 0, st, p);
    JTSM_CHECK_LAUNCH("igemm");
    record_mid(st);
    return JTSM_OK;
  }
  p.ktiles_per_split = ceil_div(ktiles, splits);
  splits = ceil_div(ktiles, p.ktiles_per_split);
  p.slab = reinterpret_cast<float*>(workspace);
  if (dma) hipLaunchKernelGGL((igemm_dma_kernel<ROLE, BM, BN, 2>), dim3(ntiles, splits), dim3(256), 0, st, p);
  else if (dma1) hipLaunchKernelGGL((igemm_dma_kernel<ROLE, BM, BN, 1>), dim3(ntiles, splits), dim3(256), 0, st, p);
  else hipLaunchKernelGGL((igemm_kernel<ROLE, BM, BN>), dim3(ntiles, splits), dim3(256), 0, st, p);
  JTSM_CHECK_LAUNCH("igemm split-K");
  record_mid(st);
  return finish_split(p, splits, st);
}

#include "conv_x3.h"

// The finishing pass of a GROUP of same-shape weight gradients (conv_x3.h: X3Group): blockIdx.y = member.
template <int VEC>
__global__ __launch_bounds__(256) void splitk_finish_group(const Params p_in, int splits, const X3Group G) {
  Params p = p_in;
  const int z = blockIdx.y;
  p.C = G.C[z];
  p.e.scale = G.scale[z];
  p.slab += (size_t)z * G.slab_stride;
  splitk_finish_body<VEC, 1>(p, splits, 1);
}


template <int ROLE, int BM, int BN>
int launch(const Params& p, int splits, hipStream_t st) {
  const int ntiles = ceil_div(p.N, BN) * ceil_div(p.M, BM);
  if (dma_eligible(ROLE, p) && p.ktiles_per_split >= kDmaMinKTiles)
    hipLaunchKernelGGL((igemm_dma_kernel<ROLE, BM, BN, 2>), dim3(ntiles, splits), dim3(256), 0, st, p);
  else if (dma_eligible(ROLE, p) && kShortSweepDma && ROLE != WGRAD)  // WGRAD short sweeps: register-staged is on par
    hipLaunchKernelGGL((igemm_dma_kernel<ROLE, BM, BN, 1>), dim3(ntiles, splits), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((igemm_kernel<ROLE, BM, BN>), dim3(ntiles, splits), dim3(256), 0, st, p);
  JTSM_CHECK_LAUNCH("igemm");
  record_mid(st);
  return JTSM_OK;
}

int check_shape(const jtsm_conv_shape* s) {
  JTSM_REQUIRE(s, "conv: null shape");
  JTSM_REQUIRE(s->batch >= 0 && s->in_h > 0 && s->in_w > 0 && s->in_c > 0 && s->out_c > 0,
               "conv: bad tensor sizes");
  JTSM_REQUIRE(s->kernel_h > 0 && s->kernel_w > 0 && s->stride > 0 && s->dilation > 0 && s->pad >= 0,
               "conv: bad kernel geometry");
  JTSM_REQUIRE(s->in_c % 4 == 0, "conv: in_c must be a multiple of 4 (pad RGB to 4), got %d", s->in_c);
  return JTSM_OK;
}

ConvShape to_shape(const jtsm_conv_shape* s) {
  ConvShape c;
  c.Bn = s->batch; c.H = s->in_h; c.W = s->in_w; c.Cin = s->in_c; c.Cout = s->out_c;
  c.KH = s->kernel_h; c.KW = s->kernel_w; c.stride = s->stride; c.pad = s->pad; c.dil = s->dilation;
  c.Ho = (s->in_h + 2 * s->pad - s->dilation * (s->kernel_h - 1) - 1) / s->stride + 1;
  c.Wo = (s->in_w + 2 * s->pad - s->dilation * (s->kernel_w - 1) - 1) / s->stride + 1;
  return c;
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

int jtsm_conv_out_size(const jtsm_conv_shape* s, int* out_h, int* out_w) {
  int rc = check_shape(s);
  if (rc) return rc;
  const ConvShape c = to_shape(s);
  JTSM_REQUIRE(c.Ho > 0 && c.Wo > 0, "conv: kernel larger than padded input");
  if (out_h) *out_h = c.Ho;
  if (out_w) *out_w = c.Wo;
  return JTSM_OK;
}

size_t jtsm_conv_workspace_bytes(const jtsm_conv_shape* s, int backward_data) {
  if (!s || check_shape(s)) return 0;
  const ConvShape c = to_shape(s);
  if (c.Ho <= 0 || c.Wo <= 0) return 0;
  long M, N, K;
  if (!backward_data) { M = (long)c.Bn * c.Ho * c.Wo; N = c.Cout; K = (long)c.KH * c.KW * c.Cin; }
  else if (c.KH == 1 && c.KW == 1 && c.pad == 0 && c.stride > 1) { M = (long)c.Bn * c.Ho * c.Wo; N = c.Cin; K = c.Cout; }
  else { M = (long)c.Bn * c.H * c.W; N = c.Cin; K = (long)c.KH * c.KW * c.Cout; }
  const int bm = N <= 64 ? 256 : 128, bn = N <= 64 ? 64 : 128;
  int splits = plan_splits(ceil_div(N, bn) * ceil_div(M, bm), ceil_div(K, BK));
  Params p = {};   // the bf16x3 launcher may want more slices (other tiles, other round size): cover both
  p.M = (int)M; p.N = (int)N; p.K = (int)K;
  const int sx = x3_wanted_splits(p);
  if (sx > splits) splits = sx;
  return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

// What a call of this shape will launch: *kernel = 0 register-staged igemm_kernel, 1 = igemm_dma_kernel;
// tile and K-split as chosen by the launchers (assuming the caller passes the advertised workspace).
// role: 0 forward, 1 backward-data, 2 backward-weight.
int jtsm_conv_plan(const jtsm_conv_shape* s, int role, int has_kscale, int* kernel, int* tile_m, int* tile_n,
                   int* splits) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(role >= 0 && role <= 2, "conv_plan: role must be 0, 1 or 2");
  if (role == FWD) { p.M = p.s.Bn * p.s.Ho * p.s.Wo; p.N = p.s.Cout; p.K = p.s.KH * p.s.KW * p.s.Cin; }
  else if (role == DGRAD) {
    p.M = p.s.Bn * p.s.H * p.s.W; p.N = p.s.Cin; p.K = p.s.KH * p.s.KW * p.s.Cout;
    if (p.s.KH == 1 && p.s.KW == 1 && p.s.pad == 0 && p.s.stride > 1) p.M = p.s.Bn * p.s.Ho * p.s.Wo;
    if (has_kscale) p.kscale = reinterpret_cast<const float*>(1);
  } else { p.M = p.s.Cout; p.N = p.s.KH * p.s.KW * p.s.Cin; p.K = p.s.Bn * p.s.Ho * p.s.Wo; }
  int bm = 128, bn = 128, sp = 1, kps;
  const int ktiles = ceil_div(p.K, BK);
  if (role == WGRAD) {
    const int ntiles = ceil_div(p.N, 128) * ceil_div(p.M, 128);
    sp = ceil_div(kWgradTargetBlocks, ntiles);
    if (sp > ceil_div(ktiles, 8)) sp = ceil_div(ktiles, 8);
    if (sp < 1) sp = 1;
    kps = ceil_div(ktiles, sp);
    sp = kps > 0 ? ceil_div(ktiles, kps) : 1;
  } else {
    if (p.N <= 64) { bm = 256; bn = 64; }
    sp = plan_splits(ceil_div(p.N, bn) * ceil_div(p.M, bm), ktiles);
    kps = ceil_div(ktiles, sp);
    sp = kps > 0 ? ceil_div(ktiles, kps) : 1;
  }
  if (kernel) *kernel = !dma_eligible(role, p) ? 0 : (kps >= kDmaMinKTiles ? 1 : ((kShortSweepDma && role != WGRAD) ? 2 : 0));
  if (tile_m) *tile_m = bm;
  if (tile_n) *tile_n = bn;
  if (splits) *splits = sp;
  return JTSM_OK;
}

int jtsm_conv2d_forward_f32(const float* x, const float* w, float* y, const jtsm_conv_shape* s,
                            const float* scale, const float* bias, const float* residual, int relu,
                            void* workspace, size_t workspace_bytes, void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  p.M = p.s.Bn * p.s.Ho * p.s.Wo;
  p.N = p.s.Cout;
  p.K = p.s.KH * p.s.KW * p.s.Cin;
  if (p.M == 0) return JTSM_OK;
  JTSM_REQUIRE(x && w && y, "conv forward: null pointer");
  JTSM_REQUIRE(aligned16(x) && aligned16(w), "conv forward: x and w must be 16-byte aligned");
  p.A = x; p.B = w; p.C = y; p.ldc = p.N;
  p.e.scale = scale; p.e.bias = bias; p.e.residual = residual; p.e.relu = relu;
  hipStream_t st = as_stream(stream);
  if (!workspace) workspace_bytes = 0;
  if (p.N <= 64) return launch_split<FWD, 256, 64>(p, workspace, workspace_bytes, st);
  return launch_split<FWD, 128, 128>(p, workspace, workspace_bytes, st);
}

int jtsm_conv2d_backward_data_f32(const float* dy, const float* w, float* dx,
                                  const jtsm_conv_shape* s, const float* kscale,
                                  const float* accumulate, const float* relu_mask, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(p.s.Cout % 4 == 0, "conv backward-data: out_c must be a multiple of 4, got %d", p.s.Cout);
  p.M = p.s.Bn * p.s.H * p.s.W;
  p.N = p.s.Cin;
  p.K = p.s.KH * p.s.KW * p.s.Cout;
  if (p.M == 0) return JTSM_OK;
  JTSM_REQUIRE(dy && w && dx, "conv backward-data: null pointer");
  JTSM_REQUIRE(aligned16(dy) && aligned16(w), "conv backward-data: dy and w must be 16-byte aligned");
  p.A = dy; p.B = w; p.C = dx; p.ldc = p.N; p.kscale = kscale;
  p.e.residual = accumulate; p.e.mask = relu_mask;
  hipStream_t st = as_stream(stream);
  if (!workspace) workspace_bytes = 0;
  if (p.s.KH == 1 && p.s.KW == 1 && p.s.pad == 0 && p.s.stride > 1 && (!accumulate || accumulate == dx) && !relu_mask) {
    // strided 1x1: only every stride-th input pixel receives gradient.  Zero dX, then run the dense
    // GEMM over the OUTPUT pixels (a 1x1/stride-1 problem on the (Ho,Wo) grid) and scatter its rows.
    // (accumulate == dx: dX holds a gradient already and the scattered rows are added to it in place — the epilogue
    // reads its residual at the scattered position — while the pixels in between keep what they hold.)
    if (!accumulate) JTSM_CHECK_HIP(hipMemsetAsync(dx, 0, (size_t)p.M * p.N * sizeof(float), st));
    p.scatter = 1; p.sc_Ho = p.s.Ho; p.sc_Wo = p.s.Wo; p.sc_H = p.s.H; p.sc_W = p.s.W; p.sc_stride = p.s.stride;
    p.s.H = p.s.Ho; p.s.W = p.s.Wo; p.s.stride = 1;
    p.M = p.s.Bn * p.s.Ho * p.s.Wo;
  }
  if (p.N <= 64) return launch_split<DGRAD, 256, 64>(p, workspace, workspace_bytes, st);
  return launch_split<DGRAD, 128, 128>(p, workspace, workspace_bytes, st);
}

int jtsm_conv2d_backward_weight_f32(const float* dy, const float* x, float* dw,
                                    const jtsm_conv_shape* s, const float* row_scale, int zero_dw,
                                    void* stream) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(p.s.Cout % 4 == 0, "conv backward-weight: out_c must be a multiple of 4, got %d", p.s.Cout);
  p.M = p.s.Cout;
  p.N = p.s.KH * p.s.KW * p.s.Cin;
  p.K = p.s.Bn * p.s.Ho * p.s.Wo;
  JTSM_REQUIRE(dw, "conv backward-weight: null dw");
  hipStream_t st = as_stream(stream);
  if (zero_dw) JTSM_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)p.M * p.N * sizeof(float), st));
  if (p.K == 0) return JTSM_OK;
  JTSM_REQUIRE(dy && x, "conv backward-weight: null pointer");
  JTSM_REQUIRE(aligned16(dy) && aligned16(x), "conv backward-weight: dy and x must be 16-byte aligned");
  p.A = dy; p.B = x; p.C = dw; p.ldc = p.N;
  p.e.scale = row_scale;
  // Split the pixel axis so that ~4 workgroups per CU are in flight (1024 groups), but keep at
  // least 8 K tiles (256 pixels) per split so the atomic tail stays small.
  const int ntiles = ceil_div(p.N, 128) * ceil_div(p.M, 128);
  const int ktiles = ceil_div(p.K, BK);
  int splits = ceil_div(kWgradTargetBlocks, ntiles);
  if (splits > ceil_div(ktiles, 8)) splits = ceil_div(ktiles, 8);
  if (splits < 1) splits = 1;
  p.ktiles_per_split = ceil_div(ktiles, splits);
  splits = ceil_div(ktiles, p.ktiles_per_split);
  return launch<WGRAD, 128, 128>(p, splits, st);
}

/* ---- split-bf16 ("bf16x3") path ---- */
int jtsm_split_bf16_f32(const float* src, uint16_t* hi, uint16_t* lo, long n, void* stream) {
  JTSM_REQUIRE(n >= 0, "split_bf16: negative size");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(src && hi && lo, "split_bf16: null pointer");
  JTSM_REQUIRE(aligned16(src) && aligned16(hi) && aligned16(lo), "split_bf16: pointers must be 16-byte aligned");
  const long n8 = n >> 3;
  const int blocks = (int)(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src,
                     reinterpret_cast<__bf16*>(hi), reinterpret_cast<__bf16*>(lo), n);
  JTSM_CHECK_LAUNCH("split_bf16");
  return JTSM_OK;
}

int jtsm_split_bf16_paired_f32(const float* src, uint16_t* planes, long rows, int k, void* stream) {
  JTSM_REQUIRE(rows >= 0 && k >= 0, "split_bf16_paired: negative size");
  if (rows == 0 || k == 0) return JTSM_OK;
  JTSM_REQUIRE(k % 32 == 0, "split_bf16_paired: the row length must be a multiple of 32");
  JTSM_REQUIRE(src && planes, "split_bf16_paired: null pointer");
  JTSM_REQUIRE(aligned16(src) && aligned16(planes), "split_bf16_paired: pointers must be 16-byte aligned");
  const long n = rows * k, n8 = n >> 3;
  const int blocks = (int)(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
  hipLaunchKernelGGL(split_bf16_paired_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src,
                     reinterpret_cast<__bf16*>(planes), n, k);
  JTSM_CHECK_LAUNCH("split_bf16_paired");
  return JTSM_OK;
}

int jtsm_split_bf16_transposed_f32(const float* w, const float* row_scale, uint16_t* hi, uint16_t* lo, int out_c,
                                   int taps, int in_c, void* stream) {
  JTSM_REQUIRE(out_c >= 0 && taps > 0 && in_c >= 0 && taps <= 65535, "split_bf16_transposed: bad sizes");
  if (out_c == 0 || in_c == 0) return JTSM_OK;
  JTSM_REQUIRE(w && hi && lo, "split_bf16_transposed: null pointer");
  JTSM_REQUIRE(!x3_paired(hi, lo, 2) || ((long)taps * out_c) % 32 == 0 || (long)taps * out_c * in_c == 32,
               "split_bf16_transposed: paired planes (lo == hi + 32) need taps * out_c %% 32 == 0");
  hipLaunchKernelGGL(split_bf16_transposed_kernel, dim3(ceil_div(in_c, 32), ceil_div(out_c, 32), taps), dim3(256), 0,
                     as_stream(stream), w, row_scale, reinterpret_cast<__bf16*>(hi), reinterpret_cast<__bf16*>(lo), out_c,
                     taps, in_c);
  JTSM_CHECK_LAUNCH("split_bf16_transposed");
  return JTSM_OK;
}

int jtsm_conv_bf16x3_eligible(const jtsm_conv_shape* s, int role) {
  if (!s || check_shape(s)) return 0;
  const ConvShape c = to_shape(s);
  if (c.Ho <= 0 || c.Wo <= 0) return 0;
  return x3_eligible(role, c) ? 1 : 0;
}

}  // extern "C"

// NP = 2: split-bf16 planes (hi, lo).  NP = 1: one fp16 plane (the *_lo arguments are null).
// What the transposed-convolution entry points add to a forward-role launch.
struct FwdExtras {
  const float* mask = nullptr;   // keep the result where mask > 0 (a ReLU gate), as in the data-gradient role
  const uint16_t* mask_plane = nullptr;   // the same gate read from the activation's hi / fp16 plane (Params::mask_plane)
  int in_shift = 0, out_shift = 0;
  int shuffle_c = 0, shuffle_h = 0, shuffle_w = 0;   // Params::shuffle_*
  float* colsum = nullptr;                            // Params::colsum
  const uint16_t* residual_h = nullptr;               // Epilogue::residual_h (fp16 instantiations only)
  int res_shift = 0;
};

template <int NP>
static int x3_forward(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* w_hi,
                      const uint16_t* w_lo, float* y, uint16_t* y_hi, uint16_t* y_lo,
                      const jtsm_conv_shape* s, const float* scale, const float* bias,
                      const float* residual, int relu, void* workspace, size_t workspace_bytes,
                      void* stream, const FwdExtras& ex = FwdExtras()) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(x3_eligible(FWD, p.s), "conv forward bf16x3: in_c=%d with a %dx%d kernel is not eligible "
               "(see jtsm_conv_bf16x3_eligible)", p.s.Cin, p.s.KH, p.s.KW);
  p.M = p.s.Bn * p.s.Ho * p.s.Wo;
  p.N = p.s.Cout;
  p.K = p.s.KH * p.s.KW * p.s.Cin;
  if (p.M == 0) return JTSM_OK;
  JTSM_REQUIRE(x_hi && w_hi && (y || y_hi) && (NP == 1 || (x_lo && w_lo)), "conv forward bf16x3 / f16: null pointer");
  JTSM_REQUIRE(aligned16(x_hi) && aligned16(x_lo) && aligned16(w_hi) && aligned16(w_lo),
               "conv forward bf16x3 / f16: planes must be 16-byte aligned");
  X3Planes q = {reinterpret_cast<const __bf16*>(x_hi), reinterpret_cast<const __bf16*>(x_lo),
                reinterpret_cast<const __bf16*>(w_hi), reinterpret_cast<const __bf16*>(w_lo)};
  p.C = y; p.ldc = p.N;
  p.e.scale = scale; p.e.bias = bias; p.e.residual = residual; p.e.relu = relu;
  p.e.mask = ex.mask; p.mask_plane = ex.mask_plane; p.in_shift = ex.in_shift; p.out_shift = ex.out_shift;
  JTSM_REQUIRE(!ex.mask_plane || aligned16(ex.mask_plane), "conv forward bf16x3: the gate plane must be 16-byte aligned");
  p.shuffle_c = ex.shuffle_c; p.shuffle_h = ex.shuffle_h; p.shuffle_w = ex.shuffle_w;
  p.colsum = ex.colsum;
  JTSM_REQUIRE(!ex.residual_h || (NP == 1 && !residual && aligned16(ex.residual_h) && p.N % 4 == 0 && !ex.shuffle_c),
               "conv forward f16: the fp16 residual plane needs the fp16 arithmetic, out_c %% 4 == 0 and 16-byte alignment");
  p.e.residual_h = ex.residual_h; p.e.res_shift = ex.res_shift;
  JTSM_REQUIRE(!ex.colsum || (aligned16(ex.colsum) && !ex.shuffle_c), "conv forward bf16x3: column sums need a 16-byte aligned buffer (and no pixel shuffle)");
  JTSM_REQUIRE(!ex.mask || aligned16(ex.mask), "conv forward bf16x3: the gate must be 16-byte aligned");
  if (ex.shuffle_c) {   // only the wide epilogue knows the pixel-shuffle map, and a K split's finishing pass does not
    JTSM_REQUIRE(p.N == 4 * ex.shuffle_c && ex.shuffle_c % 4 == 0 && aligned16(y) && (!bias || aligned16(bias)) &&
                 !scale && !residual && !ex.mask, "conv_transpose2x2 forward: out_c %% 4 == 0 and 16-byte aligned tensors");
    workspace = nullptr;
  }
  JTSM_REQUIRE(NP == 1 || (y_hi == nullptr) == (y_lo == nullptr), "conv forward bf16x3: give both output planes or neither");
  if (y_hi) {
    JTSM_REQUIRE(p.N % 4 == 0 && aligned16(y) && aligned16(y_hi) && aligned16(y_lo) &&
                 (!scale || aligned16(scale)) && (!bias || aligned16(bias)) && (!residual || aligned16(residual)),
                 "conv forward bf16x3: output planes need out_c %% 4 == 0 and 16-byte aligned tensors");
    p.out_hi = y_hi; p.out_lo = y_lo;
  }
  hipStream_t st = as_stream(stream);
  if (!workspace) workspace_bytes = 0;
  return launch_split_x3<FWD, NP>(p, q, workspace, workspace_bytes, st);
}

template <int NP>
static int x3_backward_data(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                            const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                            const jtsm_conv_shape* s, const float* accumulate, const float* relu_mask, int grad_shift,
                            void* workspace, size_t workspace_bytes, void* stream,
                            const uint16_t* gate_plane = nullptr, const float* row_scale = nullptr,
                            float* colsum = nullptr, const uint16_t* accumulate_h = nullptr) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(x3_eligible(DGRAD, p.s), "conv backward-data bf16x3: out_c=%d with a %dx%d kernel is not eligible",
               p.s.Cout, p.s.KH, p.s.KW);
  p.M = p.s.Bn * p.s.H * p.s.W;
  p.N = p.s.Cin;
  p.K = p.s.KH * p.s.KW * p.s.Cout;
  if (p.M == 0) return JTSM_OK;
  JTSM_REQUIRE(dy_hi && wt_hi && (dx || dx_hi) && (NP == 1 || (dy_lo && wt_lo)),
               "conv backward-data bf16x3 / f16: null pointer");
  JTSM_REQUIRE(!gate_plane || aligned16(gate_plane), "conv backward-data bf16x3: the gate plane must be 16-byte aligned");
  p.mask_plane = gate_plane;
  if (row_scale) { p.e.scale = row_scale; p.scale_rows = 1; }   // dx[m][:] *= row_scale[m] (before accumulate / gates)
  JTSM_REQUIRE(grad_shift >= 0 && grad_shift <= 24, "conv backward-data f16: grad_shift must be in 0..24");
  p.in_shift = grad_shift; p.out_shift = grad_shift;   // dy planes carry 2^shift; so do the dx planes written here
  JTSM_REQUIRE(aligned16(dy_hi) && aligned16(dy_lo) && aligned16(wt_hi) && aligned16(wt_lo),
               "conv backward-data bf16x3: planes must be 16-byte aligned");
  X3Planes q = {reinterpret_cast<const __bf16*>(dy_hi), reinterpret_cast<const __bf16*>(dy_lo),
                reinterpret_cast<const __bf16*>(wt_hi), reinterpret_cast<const __bf16*>(wt_lo)};
  p.C = dx; p.ldc = p.N;
  p.e.residual = accumulate; p.e.mask = relu_mask;
  // (accumulate_h: another gradient term as an fp16 plane carrying 2^grad_shift, added in the epilogue)
  JTSM_REQUIRE(!accumulate_h || (NP == 1 && !accumulate && aligned16(accumulate_h) && p.N % 4 == 0 &&
                                 !(p.s.KH == 1 && p.s.KW == 1 && p.s.stride > 1)),
               "conv backward-data f16: an fp16 accumulate plane needs the fp16 arithmetic, in_c %% 4 == 0, no strided 1x1");
  p.e.residual_h = accumulate_h; p.e.res_shift = grad_shift;
  JTSM_REQUIRE(NP == 1 || (dx_hi == nullptr) == (dx_lo == nullptr), "conv backward-data bf16x3: give both output planes or neither");
  hipStream_t st = as_stream(stream);
  if (!workspace) workspace_bytes = 0;
  const bool scatter = p.s.KH == 1 && p.s.KW == 1 && p.s.pad == 0 && p.s.stride > 1 && (!accumulate || accumulate == dx) &&
                       !relu_mask && !gate_plane && !row_scale && dx;   // (accumulate == dx: see jtsm_conv2d_backward_data_f32)
  JTSM_REQUIRE(!colsum || (aligned16(colsum) && !scatter), "conv backward-data: column sums need a 16-byte aligned buffer (not the strided 1x1 scatter)");
  p.colsum = colsum;
  JTSM_REQUIRE(!row_scale || (p.N % 4 == 0 && aligned16(dx)), "conv backward-data: a row scale needs in_c %% 4 == 0");
  if (dx_hi) {   // planes of the finished gradient (e.g. already gated by relu_mask) for the next layer's contractions
    JTSM_REQUIRE(!scatter, "conv backward-data bf16x3: output planes are not produced by the strided 1x1 scatter path");
    JTSM_REQUIRE(p.N % 4 == 0 && aligned16(dx) && aligned16(dx_hi) && aligned16(dx_lo),
                 "conv backward-data bf16x3: output planes need in_c %% 4 == 0 and 16-byte aligned tensors");
    p.out_hi = dx_hi; p.out_lo = dx_lo;
  }
  if (scatter) {
    if (!accumulate) JTSM_CHECK_HIP(hipMemsetAsync(dx, 0, (size_t)p.M * p.N * sizeof(float), st));
    p.scatter = 1; p.sc_Ho = p.s.Ho; p.sc_Wo = p.s.Wo; p.sc_H = p.s.H; p.sc_W = p.s.W; p.sc_stride = p.s.stride;
    p.s.H = p.s.Ho; p.s.W = p.s.Wo; p.s.stride = 1;
    p.M = p.s.Bn * p.s.Ho * p.s.Wo;
  }
  return launch_split_x3<DGRAD, NP>(p, q, workspace, workspace_bytes, st);
}

// 256x256 tiles for the weight gradients whose output is large enough to fill the chip with them.
static bool x3_wgrad_big(const Params& p) {
  if (p.M < 256 || p.N < 256) return false;
  constexpr long min_work = 2000;
  const long t256 = (long)ceil_div(p.N, 256) * ceil_div(p.M, 256);
  if (t256 < 4) return false;   // 1-2 tiles cannot fill 256 CUs even at the slice cap: take four times as many 128s
  static const int force = [] { const char* e = getenv("JTSM_WGRAD_BIG"); return e ? atoi(e) : -1; }();   // sweeps only
  if (force == 0) return false;
  return t256 * ceil_div(p.K, XBK) >= min_work;   // (tiles x stages: measured crossover, tools/sweeps/wgrad_sweep.py)
}

// The LDS-halo weight gradient: 3x3, stride 1, undilated, 32-channel input blocks, output rows of whole 32-pixel
// segments, and enough pixels to feed the split.
static bool x3_wgrad_halo(const Params& p) {
  const ConvShape& s = p.s;
  return s.KH == 3 && s.KW == 3 && s.stride == 1 && s.dil == 1 && s.Cin % 32 == 0 && s.Wo % 32 == 0 &&
         s.Cout % 8 == 0 && p.K >= 2048;
}

static int x3_wgrad_splits(const Params& p) {
  if (x3_wgrad_halo(p)) {
    const int ntiles = ceil_div(p.M, 128) * (p.s.Cin / 32), segs = p.K / 32;
    int splits = ntiles >= 512 ? 1 : 512 / ntiles;
    if (splits > segs / 8) splits = segs / 8;   // at least 8 segments per workgroup
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    return ceil_div(segs, ceil_div(segs, splits));
  }
  const bool big = x3_wgrad_big(p);
  const int t = big ? 256 : 128;
  const int ntiles = ceil_div(p.N, t) * ceil_div(p.M, t);
  const int ktiles = ceil_div(p.K, XBK);
  // Pixel-axis split (measured on MI355X, tools/sweeps/wsplit.py): the kernels are bound by memory latency x bytes in
  // flight, so two workgroups per CU (512) beat one as long as each keeps >= 8 stages; 256 x 256 tiles hold one
  // workgroup per CU.
  // one resident round: 256 workgroups of 256 x 256 (one per CU), 512 of 128 x 128 (two per CU) — rounded DOWN, so the
  // launch never spills a handful of workgroups into a second round (9 tiles x 29 slices = 261 did: half the time idle)
  int splits = max(1, (big ? 256 : 512) / ntiles);
  if (!big && ceil_div(ktiles, splits) > 64) splits = max(1, 768 / ntiles);
  const int min_stages = big ? 16 : 8;
  if (splits > ceil_div(ktiles, min_stages)) splits = ceil_div(ktiles, min_stages);
  // slab traffic: the finishing pass reads splits x dW — 64 slices at most, more (up to 256) only while all the slabs
  // together stay below 48 MB: the narrow predictors (80 x 256, 56 x 128 outputs over 10^5 pixels) were running on
  // 64-128 workgroups
  int cap = 64;
  const size_t out_bytes = (size_t)p.M * p.N * sizeof(float);
  while (cap < 256 && (size_t)cap * 2 * out_bytes <= (48u << 20)) cap *= 2;
  if (splits > cap) splits = cap;
  if (splits < 1) splits = 1;
  const int kps = ceil_div(ktiles, splits);
  return kps > 0 ? ceil_div(ktiles, kps) : 1;
}

extern "C" size_t jtsm_conv_bf16x3_wgrad_workspace_bytes(const jtsm_conv_shape* s);
extern "C" size_t jtsm_conv_bf16x3_wgrad_bias_workspace_bytes(const jtsm_conv_shape* s) {
  if (!s || check_shape(s)) return 0;
  Params p = {};
  p.s = to_shape(s);
  if (p.s.Ho <= 0 || p.s.Wo <= 0) return 0;
  p.M = p.s.Cout; p.N = p.s.KH * p.s.KW * p.s.Cin; p.K = p.s.Bn * p.s.Ho * p.s.Wo;
  if (p.K == 0) return 0;
  const int splits = x3_wgrad_splits(p);
  if (splits <= 1) return 0;
  const size_t main_bytes = ((size_t)splits * p.M * p.N * sizeof(float) + 15) & ~(size_t)15;
  return main_bytes + (size_t)splits * p.M * sizeof(float);
}

extern "C" size_t jtsm_conv_bf16x3_wgrad_workspace_bytes(const jtsm_conv_shape* s) {
  if (!s || check_shape(s)) return 0;
  Params p = {};
  p.s = to_shape(s);
  if (p.s.Ho <= 0 || p.s.Wo <= 0) return 0;
  p.M = p.s.Cout; p.N = p.s.KH * p.s.KW * p.s.Cin; p.K = p.s.Bn * p.s.Ho * p.s.Wo;
  if (p.K == 0) return 0;
  const int splits = x3_wgrad_splits(p);
  return splits > 1 ? (size_t)splits * p.M * p.N * sizeof(float) : 0;
}

template <int NP>
static int x3_backward_weight(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* x_hi,
                              const uint16_t* x_lo, float* dw, const jtsm_conv_shape* s,
                              const float* row_scale, int zero_dw, int grad_shift, void* workspace,
                              size_t workspace_bytes, void* stream, float* bias_grad = nullptr) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(x3_eligible(WGRAD, p.s), "conv backward-weight bf16x3: in_c=%d and out_c=%d must be multiples of 8",
               p.s.Cin, p.s.Cout);
  p.M = p.s.Cout;
  p.N = p.s.KH * p.s.KW * p.s.Cin;
  p.K = p.s.Bn * p.s.Ho * p.s.Wo;
  JTSM_REQUIRE(dw && aligned16(dw), "conv backward-weight bf16x3: dw must be non-null and 16-byte aligned");
  hipStream_t st = as_stream(stream);
  if (p.K == 0) {
    if (zero_dw) JTSM_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)p.M * p.N * sizeof(float), st));
    if (bias_grad) JTSM_CHECK_HIP(hipMemsetAsync(bias_grad, 0, (size_t)p.M * sizeof(float), st));
    return JTSM_OK;
  }
  JTSM_REQUIRE(dy_hi && x_hi && (NP == 1 || (dy_lo && x_lo)), "conv backward-weight bf16x3 / f16: null pointer");
  JTSM_REQUIRE(grad_shift >= 0 && grad_shift <= 24, "conv backward-weight f16: grad_shift must be in 0..24");
  p.in_shift = grad_shift;
  JTSM_REQUIRE(aligned16(dy_hi) && aligned16(dy_lo) && aligned16(x_hi) && aligned16(x_lo),
               "conv backward-weight bf16x3: planes must be 16-byte aligned");
  X3Planes q = {reinterpret_cast<const __bf16*>(dy_hi), reinterpret_cast<const __bf16*>(dy_lo),
                reinterpret_cast<const __bf16*>(x_hi), reinterpret_cast<const __bf16*>(x_lo)};
  p.C = dw; p.ldc = p.N;
  p.e.scale = row_scale;
  if (!zero_dw) p.e.residual = dw;   // accumulate: dw = dw + result (the epilogue reads dw[o] before writing it)
  // Deterministic: every pixel slice writes its partial tile to a slab, a fixed-order pass adds them.
  int splits = x3_wgrad_splits(p);
  const size_t need = (size_t)splits * p.M * p.N * sizeof(float);
  JTSM_REQUIRE(splits <= 1 || (workspace && workspace_bytes >= need && aligned16(workspace)),
               "conv backward-weight bf16x3: workspace of %zu bytes needed (jtsm_conv_bf16x3_wgrad_workspace_bytes)", need);
  const bool big = x3_wgrad_big(p);
  const int tl = big ? 256 : 128;
  const int ntiles = ceil_div(p.N, tl) * ceil_div(p.M, tl);
  p.ktiles_per_split = ceil_div(ceil_div(p.K, XBK), splits);
  p.slab = splits > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  p.wide = 1;   // N = taps * in_c is a multiple of 8, dw / slab 16-byte aligned
  const bool halo = x3_wgrad_halo(p);
  if (bias_grad) {   // db beside dW: per-slice partials behind the slabs, folded by the finishing pass
    if (splits > 1) {
      const size_t off = (need + 15) & ~(size_t)15;
      JTSM_REQUIRE(workspace_bytes >= off + (size_t)splits * p.M * sizeof(float),
                   "conv backward-weight bf16x3: workspace too small for the bias partials "
                   "(jtsm_conv_bf16x3_wgrad_bias_workspace_bytes)");
      p.bias_slab = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + off);
      p.bias_out = bias_grad;
    } else {
      p.bias_slab = bias_grad;
    }
  }
  const bool fused = !bias_grad && use_fused_finish(p, halo ? ceil_div(p.M, 128) * (p.s.Cin / 32) : ntiles, splits, st);
  if (halo) {
    const int halo_tiles = ceil_div(p.M, 128) * (p.s.Cin / 32);
    if (bias_grad) hipLaunchKernelGGL((igemm_x3_wgrad_halo_kernel<NP, true>), dim3(halo_tiles, splits), dim3(256), 0, st, p, q);
    else hipLaunchKernelGGL((igemm_x3_wgrad_halo_kernel<NP, false>), dim3(halo_tiles, splits), dim3(256), 0, st, p, q);
  } else if (big) {
    if (bias_grad) hipLaunchKernelGGL((igemm_x3_wgrad_kernel<4, 2, 2, 4, 2, NP, true>), dim3(ntiles, splits), dim3(512), 0, st, p, q);
    else hipLaunchKernelGGL((igemm_x3_wgrad_kernel<4, 2, 2, 4, 2, NP, false>), dim3(ntiles, splits), dim3(512), 0, st, p, q);
  } else {
    if (bias_grad) hipLaunchKernelGGL((igemm_x3_wgrad_kernel<2, 2, 2, 2, 2, NP, true>), dim3(ntiles, splits), dim3(256), 0, st, p, q);
    else hipLaunchKernelGGL((igemm_x3_wgrad_kernel<2, 2, 2, 2, 2, NP, false>), dim3(ntiles, splits), dim3(256), 0, st, p, q);
  }
  JTSM_CHECK_LAUNCH("igemm bf16x3 wgrad");
  record_mid(st);
  if (splits > 1 && !fused) return finish_split(p, splits, st, 1);
  return JTSM_OK;
}

// K slices of a GROUP of n same-shape weight gradients: the single launch's rule with n times the tiles — one resident
// round of workgroups over the whole group.
// 256 x 256 tiles for a GROUP: the single launch's rule with the group's tile count (a res4 1x1 layer alone has 4 such
// tiles — too few to fill the chip at any slice count; six of them have 24).
static bool x3_wgrad_group_big(const Params& p, int n) {
  static const int mode = [] { const char* e = getenv("JTSM_WGRAD_GROUP_BIG"); return e ? atoi(e) : 1; }();   // sweeps
  if (mode == 0) return x3_wgrad_big(p);
  if (p.M < 256 || p.N < 256) return false;
  const long t256 = (long)ceil_div(p.N, 256) * ceil_div(p.M, 256) * n;
  if (t256 < 8 || t256 * ceil_div(p.K, XBK) < 2000) return false;
  // ... and only where slices of >= 16 stages still give most CUs a workgroup (res5's 2048 pixels do not: measured
  // 53 us on 128 such workgroups against 44 us on 512 of the 128 x 128 tiles)
  const long slices = std::min<long>(std::max<long>(1, 256 / t256), std::max<long>(1, ceil_div(p.K, XBK) / 16));
  return t256 * slices >= 192;
}

static int x3_wgrad_group_splits(const Params& p, int n) {
  if (x3_wgrad_halo(p)) {
    const int ntiles = ceil_div(p.M, 128) * (p.s.Cin / 32) * n, segs = p.K / 32;
    int splits = ntiles >= 512 ? 1 : 512 / ntiles;
    if (splits > segs / 8) splits = segs / 8;   // at least 8 segments per workgroup
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
    return ceil_div(segs, ceil_div(segs, splits));
  }
  const bool big = x3_wgrad_group_big(p, n);
  const int t = big ? 256 : 128;
  const int ntiles = ceil_div(p.N, t) * ceil_div(p.M, t) * n;
  const int ktiles = ceil_div(p.K, XBK);
  int splits = max(1, (big ? 256 : 512) / ntiles);
  const int min_stages = big ? 16 : 8;
  if (splits > ceil_div(ktiles, min_stages)) splits = ceil_div(ktiles, min_stages);
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  const int kps = ceil_div(ktiles, splits);
  return kps > 0 ? ceil_div(ktiles, kps) : 1;
}

static int group_shape(const jtsm_conv_shape* s, Params& p) {
  int rc = check_shape(s);
  if (rc) return rc;
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(x3_eligible(WGRAD, p.s), "conv backward-weight group: in_c=%d and out_c=%d must be multiples of 8",
               p.s.Cin, p.s.Cout);
  p.M = p.s.Cout;
  p.N = p.s.KH * p.s.KW * p.s.Cin;
  p.K = p.s.Bn * p.s.Ho * p.s.Wo;
  return JTSM_OK;
}

template <int NP>
static int x3_backward_weight_group(int n, const uint16_t* const* dy_hi, const uint16_t* const* dy_lo,
                                    const uint16_t* const* x_hi, const uint16_t* const* x_lo, float* const* dw,
                                    const float* const* row_scale, const jtsm_conv_shape* s, int grad_shift,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(n >= 1 && n <= kMaxGroup && dy_hi && x_hi && dw && (NP == 1 || (dy_lo && x_lo)),
               "conv backward-weight group: 1..%d members and non-null tables", kMaxGroup);
  Params p = {};
  int rc = group_shape(s, p);
  if (rc) return rc;
  JTSM_REQUIRE(p.K > 0, "conv backward-weight group: empty batch (use the single-layer entry)");
  JTSM_REQUIRE(grad_shift >= 0 && grad_shift <= 24, "conv backward-weight group: grad_shift must be in 0..24");
  hipStream_t st = as_stream(stream);
  p.in_shift = grad_shift;
  p.ldc = p.N;
  p.wide = 1;
  X3Group G = {};
  G.n = n;
  for (int i = 0; i < n; ++i) {
    JTSM_REQUIRE(dy_hi[i] && x_hi[i] && dw[i] && (NP == 1 || (dy_lo[i] && x_lo[i])), "conv backward-weight group: null member %d", i);
    JTSM_REQUIRE(aligned16(dy_hi[i]) && aligned16(x_hi[i]) && aligned16(dw[i]) && (NP == 1 || (aligned16(dy_lo[i]) && aligned16(x_lo[i]))),
                 "conv backward-weight group: member %d is not 16-byte aligned", i);
    G.A_hi[i] = reinterpret_cast<const __bf16*>(dy_hi[i]);
    G.A_lo[i] = NP == 1 ? nullptr : reinterpret_cast<const __bf16*>(dy_lo[i]);
    G.B_hi[i] = reinterpret_cast<const __bf16*>(x_hi[i]);
    G.B_lo[i] = NP == 1 ? nullptr : reinterpret_cast<const __bf16*>(x_lo[i]);
    G.C[i] = dw[i];
    G.scale[i] = row_scale ? row_scale[i] : nullptr;
  }
  const int splits = x3_wgrad_group_splits(p, n);
  G.slab_stride = (size_t)splits * p.M * p.N;
  const size_t need = splits > 1 ? (size_t)n * G.slab_stride * sizeof(float) : 0;
  JTSM_REQUIRE(splits <= 1 || (workspace && workspace_bytes >= need && aligned16(workspace)),
               "conv backward-weight group: workspace of %zu bytes needed (jtsm_conv_bf16x3_wgrad_group_workspace_bytes)", need);
  p.ktiles_per_split = ceil_div(ceil_div(p.K, XBK), splits);
  p.slab = splits > 1 ? reinterpret_cast<float*>(workspace) : nullptr;
  p.tickets = nullptr;
  const bool big = x3_wgrad_group_big(p, n);
  const int tl = big ? 256 : 128;
  const int ntiles = ceil_div(p.N, tl) * ceil_div(p.M, tl);
  if (x3_wgrad_halo(p)) {
    const int halo_tiles = ceil_div(p.M, 128) * (p.s.Cin / 32);
    hipLaunchKernelGGL((igemm_x3_wgrad_halo_group_kernel<NP>), dim3(halo_tiles, splits, n), dim3(256), 0, st, p, G);
  } else if (big) {
    hipLaunchKernelGGL((igemm_x3_wgrad_group_kernel<4, 2, 2, 4, 2, NP>), dim3(ntiles, splits, n), dim3(512), 0, st, p, G);
  } else {
    hipLaunchKernelGGL((igemm_x3_wgrad_group_kernel<2, 2, 2, 2, 2, NP>), dim3(ntiles, splits, n), dim3(256), 0, st, p, G);
  }
  JTSM_CHECK_LAUNCH("igemm bf16x3 wgrad group");
  record_mid(st);
  if (splits > 1) {
    const long total = (long)p.M * (p.N / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(splitk_finish_group<4>, dim3(blocks, n), dim3(256), 0, st, p, splits, G);
    JTSM_CHECK_LAUNCH("splitk_finish group");
  }
  return JTSM_OK;
}

extern "C" {

size_t jtsm_conv_bf16x3_wgrad_group_workspace_bytes(const jtsm_conv_shape* s, int n) {
  if (!s || n < 1 || n > kMaxGroup) return 0;
  Params p = {};
  if (group_shape(s, p) || p.K == 0) return 0;
  const int splits = x3_wgrad_group_splits(p, n);
  return splits > 1 ? (size_t)n * splits * p.M * p.N * sizeof(float) : 0;
}

int jtsm_conv_bf16x3_wgrad_group_plan(const jtsm_conv_shape* s, int n, int* tile, int* splits) {
  JTSM_REQUIRE(s && tile && splits && n >= 1 && n <= kMaxGroup, "wgrad group plan: bad arguments");
  Params p = {};
  int rc = group_shape(s, p);
  if (rc) return rc;
  *tile = x3_wgrad_halo(p) ? 0 : (x3_wgrad_group_big(p, n) ? 256 : 128);
  *splits = p.K > 0 ? x3_wgrad_group_splits(p, n) : 1;
  return JTSM_OK;
}

int jtsm_conv2d_backward_weight_group_bf16x3(int n, const uint16_t* const* dy_hi, const uint16_t* const* dy_lo,
                                             const uint16_t* const* x_hi, const uint16_t* const* x_lo,
                                             float* const* dw, const float* const* row_scale,
                                             const jtsm_conv_shape* s, void* workspace, size_t workspace_bytes,
                                             void* stream) {
  return x3_backward_weight_group<2>(n, dy_hi, dy_lo, x_hi, x_lo, dw, row_scale, s, 0, workspace, workspace_bytes, stream);
}

int jtsm_conv2d_backward_weight_group_f16(int n, const uint16_t* const* dy_h, const uint16_t* const* x_h,
                                          float* const* dw, const float* const* row_scale, const jtsm_conv_shape* s,
                                          int grad_shift, void* workspace, size_t workspace_bytes, void* stream) {
  return x3_backward_weight_group<1>(n, dy_h, nullptr, x_h, nullptr, dw, row_scale, s, grad_shift, workspace,
                                     workspace_bytes, stream);
}

int jtsm_conv2d_forward_bf16x3(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* w_hi,
                               const uint16_t* w_lo, float* y, uint16_t* y_hi, uint16_t* y_lo,
                               const jtsm_conv_shape* s, const float* scale, const float* bias,
                               const float* residual, int relu, void* workspace, size_t workspace_bytes,
                               void* stream) {
  return x3_forward<2>(x_hi, x_lo, w_hi, w_lo, y, y_hi, y_lo, s, scale, bias, residual, relu, workspace,
                       workspace_bytes, stream);
}

int jtsm_conv2d_backward_data_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                                     const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                     const jtsm_conv_shape* s, const float* accumulate, const float* relu_mask,
                                     void* workspace, size_t workspace_bytes, void* stream) {
  return x3_backward_data<2>(dy_hi, dy_lo, wt_hi, wt_lo, dx, dx_hi, dx_lo, s, accumulate, relu_mask, 0, workspace,
                             workspace_bytes, stream);
}

int jtsm_conv2d_backward_weight_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* x_hi,
                                       const uint16_t* x_lo, float* dw, const jtsm_conv_shape* s,
                                       const float* row_scale, int zero_dw, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return x3_backward_weight<2>(dy_hi, dy_lo, x_hi, x_lo, dw, s, row_scale, zero_dw, 0, workspace, workspace_bytes,
                               stream);
}

/* ---- fp16 path (BASELINE configs[4]): ONE IEEE fp16 plane per operand, fp32 accumulate, fp32 results ---- */
int jtsm_conv2d_forward_f16(const uint16_t* x_h, const uint16_t* w_h, float* y, uint16_t* y_h,
                            const jtsm_conv_shape* s, const float* scale, const float* bias,
                            const float* residual, int relu, void* workspace, size_t workspace_bytes,
                            void* stream) {
  return x3_forward<1>(x_h, nullptr, w_h, nullptr, y, y_h, nullptr, s, scale, bias, residual, relu, workspace,
                       workspace_bytes, stream);
}

/* ---- fp16-only activations: the residual / accumulate operand as an fp16 plane (include/jtsm_hip.h) ---- */
int jtsm_conv2d_forward_res16_f16(const uint16_t* x_h, const uint16_t* w_h, float* y, uint16_t* y_h,
                                  const jtsm_conv_shape* s, const float* scale, const float* bias,
                                  const uint16_t* residual_h, int relu, void* workspace, size_t workspace_bytes,
                                  void* stream) {
  FwdExtras ex;
  ex.residual_h = residual_h;
  return x3_forward<1>(x_h, nullptr, w_h, nullptr, y, y_h, nullptr, s, scale, bias, nullptr, relu, workspace,
                       workspace_bytes, stream, ex);
}
int jtsm_conv2d_backward_data_acc16_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                        const jtsm_conv_shape* s, const float* row_scale, const uint16_t* accumulate_h,
                                        const uint16_t* gate_plane, int grad_shift, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return x3_backward_data<1>(dy_h, nullptr, wt_h, nullptr, dx, dx_h, nullptr, s, nullptr, nullptr, grad_shift,
                             workspace, workspace_bytes, stream, gate_plane, row_scale, nullptr, accumulate_h);
}

// ---- ConvTranspose2d(kernel 2, stride 2, padding 0): the mask heads' upsampler -----------------------------------
// Weight = the parameter's channels_last memory, [in][dy][dx][out].  Forward is ONE GEMM of (B*H*W) x (4*out) whose
// epilogue writes each row's four 'out'-wide column groups to the four output pixels (no pixel-shuffle copy);
// its B operand [4*out][in] is the transposing split of that memory seen as a 1x1 weight (out_c = in, in_c = 4*out).
// Backward-data is the FORWARD role of the 2x2 / stride-2 convolution whose OHWI weight is the same memory.
static jtsm_conv_shape ct_gemm_shape(int batch, int h, int w, int in_c, int out_c) {
  jtsm_conv_shape g = {};
  g.batch = batch; g.in_h = h; g.in_w = w; g.in_c = in_c; g.out_c = 4 * out_c;
  g.kernel_h = g.kernel_w = 1; g.stride = 1; g.pad = 0; g.dilation = 1;
  return g;
}
static jtsm_conv_shape ct_conv_shape(int batch, int h, int w, int in_c, int out_c) {
  jtsm_conv_shape g = {};   // the convolution that the transposed convolution is the data gradient of
  g.batch = batch; g.in_h = 2 * h; g.in_w = 2 * w; g.in_c = out_c; g.out_c = in_c;
  g.kernel_h = g.kernel_w = 2; g.stride = 2; g.pad = 0; g.dilation = 1;
  return g;
}
#define JTSM_CT_DIMS_OK(what)                                                                                    \
  JTSM_REQUIRE(batch >= 0 && h > 0 && w > 0 && in_c > 0 && out_c > 0 && in_c % 32 == 0 && out_c % 32 == 0 &&      \
               (long)batch * h * w * 4 * (long)(in_c > out_c ? in_c : out_c) < (1L << 31),                       \
               what ": needs in_c %% 32 == 0, out_c %% 32 == 0 and fewer than 2^31 elements per tensor")

int jtsm_conv_transpose2x2_forward_bf16x3(const uint16_t* x_hi, const uint16_t* x_lo, const uint16_t* wt_hi,
                                          const uint16_t* wt_lo, float* y, uint16_t* y_hi, uint16_t* y_lo, int batch,
                                          int h, int w, int in_c, int out_c, const float* bias, int relu, void* stream) {
  JTSM_CT_DIMS_OK("conv_transpose2x2 forward");
  const jtsm_conv_shape g = ct_gemm_shape(batch, h, w, in_c, out_c);
  FwdExtras ex; ex.shuffle_c = out_c; ex.shuffle_h = h; ex.shuffle_w = w;
  return x3_forward<2>(x_hi, x_lo, wt_hi, wt_lo, y, y_hi, y_lo, &g, nullptr, bias, nullptr, relu, nullptr, 0, stream, ex);
}

int jtsm_conv_transpose2x2_forward_f16(const uint16_t* x_h, const uint16_t* wt_h, float* y, uint16_t* y_h, int batch,
                                       int h, int w, int in_c, int out_c, const float* bias, int relu, void* stream) {
  JTSM_CT_DIMS_OK("conv_transpose2x2 forward");
  const jtsm_conv_shape g = ct_gemm_shape(batch, h, w, in_c, out_c);
  FwdExtras ex; ex.shuffle_c = out_c; ex.shuffle_h = h; ex.shuffle_w = w;
  return x3_forward<1>(x_h, nullptr, wt_h, nullptr, y, y_h, nullptr, &g, nullptr, bias, nullptr, relu, nullptr, 0, stream,
                       ex);
}

size_t jtsm_conv_transpose2x2_workspace_bytes(int batch, int h, int w, int in_c, int out_c) {
  if (batch <= 0 || h <= 0 || w <= 0 || in_c <= 0 || out_c <= 0) return 0;
  const jtsm_conv_shape g = ct_conv_shape(batch, h, w, in_c, out_c);
  return jtsm_conv_workspace_bytes(&g, 0);
}

int jtsm_conv_transpose2x2_backward_data_bf16x3(const uint16_t* g_hi, const uint16_t* g_lo, const uint16_t* w_hi,
                                                const uint16_t* w_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                                int batch, int h, int w, int in_c, int out_c, const float* relu_mask,
                                                const uint16_t* gate_plane, void* workspace, size_t workspace_bytes,
                                                void* stream) {
  JTSM_CT_DIMS_OK("conv_transpose2x2 backward-data");
  const jtsm_conv_shape g = ct_conv_shape(batch, h, w, in_c, out_c);
  FwdExtras ex; ex.mask = relu_mask; ex.mask_plane = gate_plane;
  return x3_forward<2>(g_hi, g_lo, w_hi, w_lo, dx, dx_hi, dx_lo, &g, nullptr, nullptr, nullptr, 0, workspace,
                       workspace_bytes, stream, ex);
}

int jtsm_conv_transpose2x2_backward_data_f16(const uint16_t* g_h, const uint16_t* w_h, float* dx, uint16_t* dx_h,
                                             int batch, int h, int w, int in_c, int out_c, const float* relu_mask,
                                             const uint16_t* gate_plane, int grad_shift, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  JTSM_CT_DIMS_OK("conv_transpose2x2 backward-data");
  JTSM_REQUIRE(grad_shift >= 0 && grad_shift <= 24, "conv_transpose2x2 backward-data f16: grad_shift must be in 0..24");
  const jtsm_conv_shape g = ct_conv_shape(batch, h, w, in_c, out_c);
  FwdExtras ex; ex.mask = relu_mask; ex.mask_plane = gate_plane; ex.in_shift = grad_shift; ex.out_shift = grad_shift;
  return x3_forward<1>(g_h, nullptr, w_h, nullptr, dx, dx_h, nullptr, &g, nullptr, nullptr, nullptr, 0, workspace,
                       workspace_bytes, stream, ex);
}

// The data gradient with its ReLU gate read from the gated activation's hi / fp16 PLANE (Params::mask_plane); dx may
// be null when only the planes of the gated gradient are wanted (a chain that keeps activations as planes).
int jtsm_conv2d_backward_data_ex_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                                        const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                        const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                        const float* relu_mask, const uint16_t* gate_plane, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return x3_backward_data<2>(dy_hi, dy_lo, wt_hi, wt_lo, dx, dx_hi, dx_lo, s, accumulate, relu_mask, 0, workspace,
                             workspace_bytes, stream, gate_plane, row_scale);
}

int jtsm_conv2d_backward_data_ex_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                     const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                     const float* relu_mask, const uint16_t* gate_plane, int grad_shift, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  return x3_backward_data<1>(dy_h, nullptr, wt_h, nullptr, dx, dx_h, nullptr, s, accumulate, relu_mask, grad_shift,
                             workspace, workspace_bytes, stream, gate_plane, row_scale);
}

// The same data gradients, also leaving the column sums of the finished (gated) result per row tile: the bias gradient
// of the layer below, taken where its output gradient is written.  colsum: jtsm_conv_bf16x3_colsum_rows(s, role) x in_c
// floats; the caller adds the rows up (jtsm_channel_sum_f32) in order.
int jtsm_conv_bf16x3_colsum_rows(const jtsm_conv_shape* s, int role) {
  if (check_shape(s) || (role != FWD && role != DGRAD)) return 0;
  Params p = {};
  p.s = to_shape(s);
  if (p.s.Ho <= 0 || p.s.Wo <= 0 || !x3_eligible(role, p.s)) return 0;
  if (role == FWD) { p.M = p.s.Bn * p.s.Ho * p.s.Wo; p.N = p.s.Cout; p.K = p.s.KH * p.s.KW * p.s.Cin; }
  else {
    if (p.s.KH == 1 && p.s.KW == 1 && p.s.pad == 0 && p.s.stride > 1) return 0;   // (the scatter form)
    p.M = p.s.Bn * p.s.H * p.s.W; p.N = p.s.Cin; p.K = p.s.KH * p.s.KW * p.s.Cout;
  }
  if (p.M <= 0 || p.N % 4) return 0;
  const int c = x3_tile_choice(p);
  return ceil_div(p.M, c == 0 ? 128 : (c == 3 ? 64 : 256));
}
int jtsm_conv2d_backward_data_colsum_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* wt_hi,
                                            const uint16_t* wt_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                            const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                            const float* relu_mask, const uint16_t* gate_plane, float* colsum,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(colsum, "conv backward-data colsum: null colsum");
  return x3_backward_data<2>(dy_hi, dy_lo, wt_hi, wt_lo, dx, dx_hi, dx_lo, s, accumulate, relu_mask, 0, workspace,
                             workspace_bytes, stream, gate_plane, row_scale, colsum);
}
int jtsm_conv2d_backward_data_colsum_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                         const jtsm_conv_shape* s, const float* row_scale, const float* accumulate,
                                         const float* relu_mask, const uint16_t* gate_plane, int grad_shift, float* colsum,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(colsum, "conv backward-data colsum: null colsum");
  return x3_backward_data<1>(dy_h, nullptr, wt_h, nullptr, dx, dx_h, nullptr, s, accumulate, relu_mask, grad_shift,
                             workspace, workspace_bytes, stream, gate_plane, row_scale, colsum);
}
int jtsm_conv_transpose2x2_backward_data_colsum_bf16x3(const uint16_t* g_hi, const uint16_t* g_lo, const uint16_t* w_hi,
                                                       const uint16_t* w_lo, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                                       int batch, int h, int w, int in_c, int out_c,
                                                       const float* relu_mask, const uint16_t* gate_plane, float* colsum,
                                                       void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_CT_DIMS_OK("conv_transpose2x2 backward-data");
  JTSM_REQUIRE(colsum, "conv_transpose2x2 backward-data colsum: null colsum");
  const jtsm_conv_shape g = ct_conv_shape(batch, h, w, in_c, out_c);
  FwdExtras ex; ex.mask = relu_mask; ex.mask_plane = gate_plane; ex.colsum = colsum;
  return x3_forward<2>(g_hi, g_lo, w_hi, w_lo, dx, dx_hi, dx_lo, &g, nullptr, nullptr, nullptr, 0, workspace,
                       workspace_bytes, stream, ex);
}
int jtsm_conv_transpose2x2_backward_data_colsum_f16(const uint16_t* g_h, const uint16_t* w_h, float* dx, uint16_t* dx_h,
                                                    int batch, int h, int w, int in_c, int out_c, const float* relu_mask,
                                                    const uint16_t* gate_plane, int grad_shift, float* colsum,
                                                    void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_CT_DIMS_OK("conv_transpose2x2 backward-data");
  JTSM_REQUIRE(colsum && grad_shift >= 0 && grad_shift <= 24, "conv_transpose2x2 backward-data colsum f16: null colsum / bad grad_shift");
  const jtsm_conv_shape g = ct_conv_shape(batch, h, w, in_c, out_c);
  FwdExtras ex; ex.mask = relu_mask; ex.mask_plane = gate_plane; ex.in_shift = grad_shift; ex.out_shift = grad_shift;
  ex.colsum = colsum;
  return x3_forward<1>(g_h, nullptr, w_h, nullptr, dx, dx_h, nullptr, &g, nullptr, nullptr, nullptr, 0, workspace,
                       workspace_bytes, stream, ex);
}

// dW and db in one contraction (conv_x3.h: x3_bias_mma): db[out_c] = sum over pixels of dy, from the same planes.
int jtsm_conv2d_backward_weight_bias_bf16x3(const uint16_t* dy_hi, const uint16_t* dy_lo, const uint16_t* x_hi,
                                            const uint16_t* x_lo, float* dw, float* db, const jtsm_conv_shape* s,
                                            const float* row_scale, int zero_dw, void* workspace,
                                            size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(db, "conv backward-weight+bias: null db");
  return x3_backward_weight<2>(dy_hi, dy_lo, x_hi, x_lo, dw, s, row_scale, zero_dw, 0, workspace, workspace_bytes,
                               stream, db);
}

int jtsm_conv2d_backward_weight_bias_f16(const uint16_t* dy_h, const uint16_t* x_h, float* dw, float* db,
                                         const jtsm_conv_shape* s, const float* row_scale, int zero_dw, int grad_shift,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  JTSM_REQUIRE(db, "conv backward-weight+bias: null db");
  return x3_backward_weight<1>(dy_h, nullptr, x_h, nullptr, dw, s, row_scale, zero_dw, grad_shift, workspace,
                               workspace_bytes, stream, db);
}

int jtsm_conv2d_backward_data_f16(const uint16_t* dy_h, const uint16_t* wt_h, float* dx, uint16_t* dx_h,
                                  const jtsm_conv_shape* s, const float* accumulate, const float* relu_mask,
                                  int grad_shift, void* workspace, size_t workspace_bytes, void* stream) {
  return x3_backward_data<1>(dy_h, nullptr, wt_h, nullptr, dx, dx_h, nullptr, s, accumulate, relu_mask, grad_shift,
                             workspace, workspace_bytes, stream);
}

int jtsm_conv2d_backward_weight_f16(const uint16_t* dy_h, const uint16_t* x_h, float* dw, const jtsm_conv_shape* s,
                                    const float* row_scale, int zero_dw, int grad_shift, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return x3_backward_weight<1>(dy_h, nullptr, x_h, nullptr, dw, s, row_scale, zero_dw, grad_shift, workspace,
                               workspace_bytes, stream);
}

int jtsm_split_f16_f32(const float* src, uint16_t* h, long n, int shift, void* stream) {
  JTSM_REQUIRE(n >= 0 && shift >= 0 && shift <= 24, "split_f16: bad size / shift");
  if (n == 0) return JTSM_OK;
  JTSM_REQUIRE(src && h && aligned16(src) && aligned16(h), "split_f16: pointers must be non-null and 16-byte aligned");
  const long n8 = n >> 3;
  const int blocks = (int)(n8 / 256 + 1 < 8192 ? n8 / 256 + 1 : 8192);
  hipLaunchKernelGGL(split_f16_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), src,
                     reinterpret_cast<_Float16*>(h), n, shift);
  JTSM_CHECK_LAUNCH("split_f16");
  return JTSM_OK;
}

int jtsm_split_f16_transposed_f32(const float* w, const float* row_scale, uint16_t* h, int out_c, int taps, int in_c,
                                  void* stream) {
  JTSM_REQUIRE(out_c >= 0 && taps > 0 && in_c >= 0 && taps <= 65535, "split_f16_transposed: bad sizes");
  if (out_c == 0 || in_c == 0) return JTSM_OK;
  JTSM_REQUIRE(w && h, "split_f16_transposed: null pointer");
  hipLaunchKernelGGL(split_bf16_transposed_kernel, dim3(ceil_div(in_c, 32), ceil_div(out_c, 32), taps), dim3(256), 0,
                     as_stream(stream), w, row_scale, reinterpret_cast<__bf16*>(h), (__bf16*)nullptr, out_c, taps, in_c);
  JTSM_CHECK_LAUNCH("split_f16_transposed");
  return JTSM_OK;
}

int jtsm_split_bf16_multi_f32(const void* table, int entries, long blocks, int transposed, void* stream) {
  JTSM_REQUIRE(entries >= 0 && blocks >= 0 && blocks < 2147483647L, "split_bf16_multi: bad sizes");
  if (entries == 0 || blocks == 0) return JTSM_OK;
  JTSM_REQUIRE(table && aligned16(table), "split_bf16_multi: table must be a 16-byte aligned device pointer");
  const SplitEntry* tab = reinterpret_cast<const SplitEntry*>(table);
  if (transposed)
    hipLaunchKernelGGL(split_bf16_transposed_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tab, entries);
  else
    hipLaunchKernelGGL(split_bf16_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), tab, entries);
  JTSM_CHECK_LAUNCH("split_bf16_multi");
  return JTSM_OK;
}

/* What a bf16x3 call of this shape launches (assuming the advertised workspace): the template arguments of
 * igemm_x3_kernel<role, WM, WN, TM, TN, NBUF> / igemm_x3_wgrad_kernel<WM, WN, TM, TN, NBUF> and the K slices. */
int jtsm_conv_bf16x3_plan(const jtsm_conv_shape* s, int role, int* wm, int* wn, int* tm, int* tn, int* nbuf,
                          int* splits) {
  int rc = check_shape(s);
  if (rc) return rc;
  Params p = {};
  p.s = to_shape(s);
  JTSM_REQUIRE(p.s.Ho > 0 && p.s.Wo > 0, "conv: kernel larger than padded input");
  JTSM_REQUIRE(role >= 0 && role <= 2 && x3_eligible(role, p.s), "conv_bf16x3_plan: shape not eligible in this role");
  int cfg[4] = {2, 2, 2, 2}, nb = 2, sp = 1;
  if (role == WGRAD) {
    p.M = p.s.Cout; p.N = p.s.KH * p.s.KW * p.s.Cin; p.K = p.s.Bn * p.s.Ho * p.s.Wo;
    if (x3_wgrad_halo(p)) nb = 0;   // igemm_x3_wgrad_halo_kernel
    else if (x3_wgrad_big(p)) { cfg[0] = 4; cfg[1] = 2; cfg[2] = 2; cfg[3] = 4; }
    sp = p.K > 0 ? x3_wgrad_splits(p) : 1;
  } else {
    if (role == FWD) { p.M = p.s.Bn * p.s.Ho * p.s.Wo; p.N = p.s.Cout; p.K = p.s.KH * p.s.KW * p.s.Cin; }
    else {
      p.M = p.s.Bn * p.s.H * p.s.W; p.N = p.s.Cin; p.K = p.s.KH * p.s.KW * p.s.Cout;
      if (p.s.KH == 1 && p.s.KW == 1 && p.s.pad == 0 && p.s.stride > 1) p.M = p.s.Bn * p.s.Ho * p.s.Wo;
    }
    const int c = x3_tile_choice(p);
    if (c == 1) { cfg[0] = 4; cfg[1] = 1; }
    if (c == 2) { cfg[0] = 4; cfg[1] = 2; cfg[2] = 2; cfg[3] = 4; }
    if (c == 3 && !x3_halo_ok(role, p)) { cfg[2] = 1; cfg[3] = 1; }   // 64 x 64 (the halo kernels have their own shapes)
    if (x3_halo_ok(role, p)) {   // reported as NBUF = 0: igemm_x3_halo_kernel<role, TH, WM, WN, TN, HP16>
      const bool big = c == 2;
      const int OH = role == FWD ? p.s.Ho : p.s.H, OW = role == FWD ? p.s.Wo : p.s.W;
      const int ntiles = ceil_div(p.N, big ? 256 : 128) * p.s.Bn * ceil_div(OH, big ? 16 : 8) * ceil_div(OW, 16);
      const int Cb = (role == FWD ? p.s.Cin : p.s.Cout) / XBK, round_blocks = big ? 256 : 512;
      sp = ntiles >= round_blocks ? 1 : round_blocks / ntiles;
      if (sp > Cb) sp = Cb;
      if (sp > 16) sp = 16;
      if (sp < 1) sp = 1;
      sp = ceil_div(Cb, ceil_div(Cb, sp));
      nb = 0;
    } else {
      sp = x3_wanted_splits(p);
      const int ktiles = ceil_div(p.K, XBK);
      const int kps = ceil_div(ktiles, sp);
      sp = kps > 0 ? ceil_div(ktiles, kps) : 1;
      if (c != 2 && ceil_div(ktiles, sp) <= x3_nbuf1_stages(role)) nb = 1;
      else if (c == 3 && x3_ring_enabled() && ceil_div(ktiles, sp) >= 4) nb = 4;   // the four-stage ring
    }
  }
  if (wm) *wm = cfg[0];
  if (wn) *wn = cfg[1];
  if (tm) *tm = cfg[2];
  if (tn) *tn = cfg[3];
  if (nbuf) *nbuf = nb;
  if (splits) *splits = sp;
  return JTSM_OK;
}

void jtsm_conv_set_mid_event(void* event) { g_mid_event = reinterpret_cast<hipEvent_t>(event); }

void jtsm_conv_set_splitk_fused(int mode) { g_splitk_fused_override.store(mode < 0 ? -1 : (mode ? 1 : 0)); }

}  // extern "C"
