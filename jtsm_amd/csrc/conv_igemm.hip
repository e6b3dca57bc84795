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
  ConvShape s;
  Epilogue e;
};

constexpr int BK = 32;
constexpr int PAD_T = 1;  // transposed-store tiles: stride BM+1
constexpr int PAD_D = 4;  // direct-store tiles:     stride BM+4

__device__ __forceinline__ float4 ldg4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

// ---- operand address generators --------------------------------------------------------------
// Every operand is fetched in 16-byte chunks; `row` runs over the tile's non-K axis (m or n),
// `k` is the absolute K index of the chunk's first element (K-contiguous operands), or the
// chunk's k-row with `col` the first of 4 contiguous non-K elements (row-contiguous operands).

// FWD A: X gathered by output pixel and tap.  k = tap*Cin + ci.
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
__device__ __forceinline__ const float* fwd_a_ptr(const Params& p, const PixelRow& r, int k, int kend) {
  const ConvShape& s = p.s;
  if (!r.ok || k >= kend) return nullptr;
  const int tap = k / s.Cin, ci = k - tap * s.Cin;
  const int kh = tap / s.KW, kw = tap - kh * s.KW;
  const int ih = r.h0 + kh * s.dil, iw = r.w0 + kw * s.dil;
  if ((unsigned)ih >= (unsigned)s.H || (unsigned)iw >= (unsigned)s.W) return nullptr;
  return p.A + ((size_t)(r.b * s.H + ih) * s.W + iw) * s.Cin + ci;
}

// DGRAD A: dY gathered by INPUT pixel and tap.  k = tap*Cout + co.
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
__device__ __forceinline__ const float* dgrad_a_ptr(const Params& p, const PixelRow& r, int k, int kend) {
  const ConvShape& s = p.s;
  if (!r.ok || k >= kend) return nullptr;
  const int tap = k / s.Cout, co = k - tap * s.Cout;
  const int kh = tap / s.KW, kw = tap - kh * s.KW;
  const int th = r.h0 - kh * s.dil, tw = r.w0 - kw * s.dil;
  if (th < 0 || tw < 0) return nullptr;
  int oh = th, ow = tw;
  if (s.stride != 1) {
    oh = th / s.stride;
    ow = tw / s.stride;
    if (oh * s.stride != th || ow * s.stride != tw) return nullptr;
  }
  if (oh >= s.Ho || ow >= s.Wo) return nullptr;
  return p.A + ((size_t)(r.b * s.Ho + oh) * s.Wo + ow) * s.Cout + co;
}

// FWD B: W[n][k], K-contiguous.
__device__ __forceinline__ const float* fwd_b_ptr(const Params& p, int n, int k, int kend) {
  if (n >= p.N || k >= kend) return nullptr;
  return p.B + (size_t)n * p.K + k;
}

// DGRAD B: row k = (tap, co) of W viewed as [K][Cin]; col = ci (contiguous).
__device__ __forceinline__ const float* dgrad_b_ptr(const Params& p, int k, int col, float* ks, int kend) {
  const ConvShape& s = p.s;
  if (k >= kend || col >= p.N) return nullptr;
  const int tap = k / s.Cout, co = k - tap * s.Cout;
  *ks = p.kscale ? p.kscale[co] : 1.f;
  return p.B + ((size_t)co * (s.KH * s.KW) + tap) * s.Cin + col;
}

// WGRAD A: dY[pixel k][co], co contiguous.
__device__ __forceinline__ const float* wgrad_a_ptr(const Params& p, int k, int col, int kend) {
  if (k >= kend || col >= p.M) return nullptr;
  return p.A + (size_t)k * p.s.Cout + col;
}
// WGRAD B: X[pix(k, tap)][ci]; col = tap*Cin + ci.
__device__ __forceinline__ const float* wgrad_b_ptr(const Params& p, int k, int col, int kend) {
  const ConvShape& s = p.s;
  if (k >= kend || col >= p.N) return nullptr;
  const int tap = col / s.Cin, ci = col - tap * s.Cin;
  const int kh = tap / s.KW, kw = tap - kh * s.KW;
  const int ow = k % s.Wo, t = k / s.Wo;
  const int oh = t % s.Ho, b = t / s.Ho;
  const int ih = oh * s.stride - s.pad + kh * s.dil, iw = ow * s.stride - s.pad + kw * s.dil;
  if ((unsigned)ih >= (unsigned)s.H || (unsigned)iw >= (unsigned)s.W) return nullptr;
  return p.B + ((size_t)(b * s.H + ih) * s.W + iw) * s.Cin + ci;
}

// ---- the kernel ------------------------------------------------------------------------------
// BM x BN output tile, WM x WN wavefronts (WM*WN == 4), each wavefront 64x64.
template <int ROLE, int BM, int BN>
__global__ __launch_bounds__(256, 4) void igemm_kernel(const Params p) {
  constexpr int WN = BN / 64;
  static_assert((BM / 64) * WN == 4, "four wavefronts of 64x64");
  constexpr bool A_T = ROLE != WGRAD;  // A staged transposed (K-contiguous source)?
  constexpr bool B_T = ROLE == FWD;
  constexpr int SA = BM + (A_T ? PAD_T : PAD_D);
  constexpr int SB = BN + (B_T ? PAD_T : PAD_D);
  constexpr int A_CH = BM * BK / 4 / 256;  // 16-byte chunks per thread per K tile
  constexpr int B_CH = BN * BK / 4 / 256;

  __shared__ __attribute__((aligned(16))) float lds[BK * SA + BK * SB];
  float* As = lds;
  float* Bs = lds + BK * SA;

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

  // Per-thread chunk coordinates.
  //   transposed tiles: chunk j -> row = tid/8 + 32*j, kq = tid%8 (k offset 4*kq)
  //   direct tiles:     chunk j -> krow = tid/(BX/4) + (1024/BX)*j, col = 4*(tid%(BX/4))
  PixelRow arow[A_T ? A_CH : 1];
  if (A_T) {
#pragma unroll
    for (int j = 0; j < A_CH; ++j) {
      const int m = m0 + tid / 8 + 32 * j;
      arow[j] = ROLE == FWD ? fwd_pixel(p.s, m, p.M) : dgrad_pixel(p.s, m, p.M);
    }
  }

  float4 ra[A_CH], rb[B_CH];
  auto fetch = [&](int k0) {
    if (A_T) {
      const int k = k0 + 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const float* q = ROLE == FWD ? fwd_a_ptr(p, arow[j], k, kend) : dgrad_a_ptr(p, arow[j], k, kend);
        ra[j] = q ? ldg4(q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      constexpr int CPR = BM / 4;  // chunks per k-row
      const int col = m0 + 4 * (tid % CPR);
#pragma unroll
      for (int j = 0; j < A_CH; ++j) {
        const int k = k0 + tid / CPR + (256 / CPR) * j;
        const float* q = wgrad_a_ptr(p, k, col, kend);
        ra[j] = q ? ldg4(q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (B_T) {
      const int k = k0 + 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        const float* q = fwd_b_ptr(p, n0 + tid / 8 + 32 * j, k, kend);
        rb[j] = q ? ldg4(q) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
      constexpr int CPR = BN / 4;
      const int col = n0 + 4 * (tid % CPR);
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        const int k = k0 + tid / CPR + (256 / CPR) * j;
        float ks = 1.f;
        const float* q = ROLE == DGRAD ? dgrad_b_ptr(p, k, col, &ks, kend) : wgrad_b_ptr(p, k, col, kend);
        float4 v = q ? ldg4(q) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (ROLE == DGRAD) { v.x *= ks; v.y *= ks; v.z *= ks; v.w *= ks; }
        rb[j] = v;
      }
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
      constexpr int CPR = BM / 4;
#pragma unroll
      for (int j = 0; j < A_CH; ++j)
        *reinterpret_cast<float4*>(As + (tid / CPR + (256 / CPR) * j) * SA + 4 * (tid % CPR)) = ra[j];
    }
    if (B_T) {
      const int kq = 4 * (tid & 7);
#pragma unroll
      for (int j = 0; j < B_CH; ++j) {
        float* d = Bs + kq * SB + tid / 8 + 32 * j;
        d[0] = rb[j].x; d[SB] = rb[j].y; d[2 * SB] = rb[j].z; d[3 * SB] = rb[j].w;
      }
    } else {
      constexpr int CPR = BN / 4;
#pragma unroll
      for (int j = 0; j < B_CH; ++j)
        *reinterpret_cast<float4*>(Bs + (tid / CPR + (256 / CPR) * j) * SB + 4 * (tid % CPR)) = rb[j];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  fetch(kbeg);
  const float* aw = As + (lane >> 5) * SA + wm * 64 + (lane & 31);
  const float* bw = Bs + (lane >> 5) * SB + wn * 64 + (lane & 31);
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // everyone done reading the previous tile
    stage();
    __syncthreads();
    if (k0 + BK < kend) fetch(k0 + BK);  // in flight during the MFMAs below
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

  // Epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  const Epilogue& e = p.e;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + (lane & 31);
    if (n >= p.N) continue;
    const float sc = (ROLE != WGRAD && e.scale) ? e.scale[n] : 1.f;
    const float bi = e.bias ? e.bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= p.M) continue;
        float v = acc[i][j][r];
        if (ROLE != WGRAD && gridDim.y > 1) {  // split-K: raw partial into this slice's slab
          p.slab[((size_t)blockIdx.y * p.M + m) * p.ldc + n] = v;
          continue;
        }
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

// Fold the split-K slabs in slice order (deterministic) and apply the fused epilogue.
__global__ __launch_bounds__(256) void splitk_finish(const Params p, int splits) {
  const long total = (long)p.M * p.N;
  const Epilogue& e = p.e;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % p.N);
    const long m = i / p.N;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += p.slab[((size_t)s * p.M + m) * p.ldc + n];
    size_t o = (size_t)m * p.ldc + n;
    if (p.scatter) {
      const int ow = (int)(m % p.sc_Wo);
      const long t = m / p.sc_Wo;
      const int oh = (int)(t % p.sc_Ho), b = (int)(t / p.sc_Ho);
      o = ((size_t)(b * p.sc_H + oh * p.sc_stride) * p.sc_W + ow * p.sc_stride) * p.ldc + n;
    }
    v = v * (e.scale ? e.scale[n] : 1.f) + (e.bias ? e.bias[n] : 0.f);
    if (e.residual) v += e.residual[o];
    if (e.relu) v = fmaxf(v, 0.f);
    if (e.mask) v = e.mask[o] > 0.f ? v : 0.f;
    p.C[o] = v;
  }
}

// How many K slices for an (ntiles, ktiles) problem: aim at >= 3 workgroups per CU, keep >= 4 K tiles
// (128 k) per slice, at most 16 slices.
inline int plan_splits(int ntiles, int ktiles) {
  if (ntiles >= 512) return 1;
  int s = ceil_div(768, ntiles);
  if (s > ktiles / 4) s = ktiles / 4;
  if (s > 16) s = 16;
  return s < 1 ? 1 : s;
}

template <int ROLE, int BM, int BN>
int launch_split(Params& p, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int ntiles = ceil_div(p.N, BN) * ceil_div(p.M, BM);
  const int ktiles = ceil_div(p.K, BK);
  int splits = plan_splits(ntiles, ktiles);
  if (splits > 1 && (size_t)splits * p.M * p.ldc * sizeof(float) > workspace_bytes) splits = 1;
  if (splits <= 1) {
    hipLaunchKernelGGL((igemm_kernel<ROLE, BM, BN>), dim3(ntiles, 1), dim3(256), 0, st, p);
    JTSM_CHECK_LAUNCH("igemm");
    return JTSM_OK;
  }
  p.ktiles_per_split = ceil_div(ktiles, splits);
  splits = ceil_div(ktiles, p.ktiles_per_split);
  p.slab = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL((igemm_kernel<ROLE, BM, BN>), dim3(ntiles, splits), dim3(256), 0, st, p);
  JTSM_CHECK_LAUNCH("igemm split-K");
  const long total = (long)p.M * p.N;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(splitk_finish, dim3(blocks), dim3(256), 0, st, p, splits);
  JTSM_CHECK_LAUNCH("splitk_finish");
  return JTSM_OK;
}

template <int ROLE, int BM, int BN>
int launch(const Params& p, int splits, hipStream_t st) {
  const int ntiles = ceil_div(p.N, BN) * ceil_div(p.M, BM);
  hipLaunchKernelGGL((igemm_kernel<ROLE, BM, BN>), dim3(ntiles, splits), dim3(256), 0, st, p);
  JTSM_CHECK_LAUNCH("igemm");
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

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

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
  const int splits = plan_splits(ceil_div(N, bn) * ceil_div(M, bm), ceil_div(K, BK));
  return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
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
  if (p.s.KH == 1 && p.s.KW == 1 && p.s.pad == 0 && p.s.stride > 1 && !accumulate && !relu_mask) {
    // strided 1x1: only every stride-th input pixel receives gradient.  Zero dX, then run the dense
    // GEMM over the OUTPUT pixels (a 1x1/stride-1 problem on the (Ho,Wo) grid) and scatter its rows.
    JTSM_CHECK_HIP(hipMemsetAsync(dx, 0, (size_t)p.M * p.N * sizeof(float), st));
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
  int splits = ceil_div(1024, ntiles);
  if (splits > ceil_div(ktiles, 8)) splits = ceil_div(ktiles, 8);
  if (splits < 1) splits = 1;
  p.ktiles_per_split = ceil_div(ktiles, splits);
  splits = ceil_div(ktiles, p.ktiles_per_split);
  return launch<WGRAD, 128, 128>(p, splits, st);
}

}  // extern "C"
