// NHWC GroupNorm(+ReLU) and bilinear x2 up-sampling — the non-GEMM layers of SemSegFPNHead
// (detectron2/modeling/meta_arch/semantic_seg.py:126-150: Conv2d(norm=GroupNorm(32, C), activation=relu)
// followed by nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)).  The reference runs
// them as ATen ops on NCHW tensors; here they stay channels-last (one float4 = 4 channels of a pixel),
// ReLU is folded into the normalisation pass, and every reduction is two-stage with a fixed fold order
// (no float atomics -> deterministic).
#include "common.h"

namespace jtsm {
namespace {

constexpr int GN_SLAB_ROWS = 256;   // pixels per workgroup in the statistics passes (>= 2 workgroups per CU on the 256x256 map)

#ifdef JTSM_DIAG_LD4   // diagnostic builds (tools/sweeps/build_alt_src.sh ... -DJTSM_DIAG_LD4='"sc0 sc1"'): every 16-byte
                       // read of this file as an asm load with the given cache bits, waited for at once
typedef float diag_fx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
  diag_fx4 t;
  asm volatile("global_load_dwordx4 %0, %1, off " JTSM_DIAG_LD4 "\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(p) : "memory");
  return make_float4(t[0], t[1], t[2], t[3]);
}
#else
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
#endif
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---- forward statistics: part[n][slab][chunk] = (sum, sumsq) over the slab's pixels of the chunk's 4
// channels.  Requires 256 % (C/4) == 0 so that a thread always visits the same chunk.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, float2* __restrict__ part,
                                                       int HW, int C, int slabs) {
  __shared__ float2 red[256];
  const int C4 = C / 4, n = blockIdx.y, slab = blockIdx.x;
  const int p0 = slab * GN_SLAB_ROWS, p1 = min(HW, p0 + GN_SLAB_ROWS);
  const float* base = x + (size_t)n * HW * C;
  float s = 0.f, q = 0.f;
  for (long it = (long)p0 * C4 + threadIdx.x; it < (long)p1 * C4; it += 256) {
    const float4 v = ld4(base + it * 4);
    s += (v.x + v.y) + (v.z + v.w);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  red[threadIdx.x] = make_float2(s, q);
  __syncthreads();
  if (threadIdx.x < C4) {
    float2 a = make_float2(0.f, 0.f);
    for (int t = threadIdx.x; t < 256; t += C4) { a.x += red[t].x; a.y += red[t].y; }
    part[((size_t)n * slabs + slab) * C4 + threadIdx.x] = a;
  }
}

// mean / rstd per (n, group) from the partials, folded in slab order in double.
__global__ void gn_fold_kernel(const float2* __restrict__ part, float* __restrict__ mean,
                               float* __restrict__ rstd, int HW, int C, int G, int slabs, float eps) {
  // blockDim = 64 * S: slice j of group g folds slabs j, j+S, ... in double; the slices are added in slice order
  __shared__ double rs[16][64], rq[16][64];
  const int n = blockIdx.x, g = threadIdx.x & 63, j = threadIdx.x >> 6, S = blockDim.x >> 6;
  const int C4 = C / 4, cpg4 = C4 / G;
  double s = 0.0, q = 0.0;
  if (g < G)
    for (int sl = j; sl < slabs; sl += S)
      for (int k = 0; k < cpg4; ++k) {
        const float2 v = part[((size_t)n * slabs + sl) * C4 + g * cpg4 + k];
        s += v.x; q += v.y;
      }
  rs[j][g] = s; rq[j][g] = q;
  __syncthreads();
  if (j != 0 || g >= G) return;
  s = 0.0; q = 0.0;
  for (int jj = 0; jj < S; ++jj) { s += rs[jj][g]; q += rq[jj][g]; }
  const double m = (double)HW * (C / G);
  const double mu = s / m;
  double var = q / m - mu * mu;
  if (var < 0.0) var = 0.0;
  mean[n * G + g] = (float)mu;
  rstd[n * G + g] = (float)(1.0 / sqrt(var + (double)eps));
}

// bf16 hi / lo planes of four values (what jtsm_split_bf16_f32 would make of them), for a bf16x3 consumer
typedef __bf16 ss_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void put_planes4(unsigned short* hi, unsigned short* lo, long i4, const float4& v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  ss_bf16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
  reinterpret_cast<ss_bf16x4*>(hi)[i4] = h;
  reinterpret_cast<ss_bf16x4*>(lo)[i4] = l;
}

// y = relu?((x - mean) * rstd * gamma + beta)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ rstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float* __restrict__ y,
                                                       long HW, int C, int G, int relu, long total4) {
  const int C4 = C / 4, cpg4 = C4 / G;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long n = i / C4 / HW;
    const int g = c4 / cpg4;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    const float4 v = ld4(x + i * 4), ga = ld4(gamma + c4 * 4), be = ld4(beta + c4 * 4);
    float4 o;
    o.x = (v.x - mu) * rs * ga.x + be.x;
    o.y = (v.y - mu) * rs * ga.y + be.y;
    o.z = (v.z - mu) * rs * ga.z + be.z;
    o.w = (v.w - mu) * rs * ga.w + be.w;
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4(y + i * 4, o);
  }
}

// ---- backward statistics: per (n, slab, channel): A = sum dyr, B = sum dyr * xhat, where
// dyr = dy gated by the ReLU (z = xhat*gamma+beta > 0).
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           float4* __restrict__ partA, float4* __restrict__ partB,
                                                           int HW, int C, int G, int slabs, int relu) {
  __shared__ float4 ra[256], rb[256];
  const int C4 = C / 4, cpg4 = C4 / G, n = blockIdx.y, slab = blockIdx.x;
  const int p0 = slab * GN_SLAB_ROWS, p1 = min(HW, p0 + GN_SLAB_ROWS);
  const size_t base = (size_t)n * HW * C;
  const int c4 = threadIdx.x % C4;
  const int g = c4 / cpg4;
  const float mu = mean[n * G + g], rs = rstd[n * G + g];
  const float4 ga = ld4(gamma + c4 * 4), be = ld4(beta + c4 * 4);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  for (long it = (long)p0 * C4 + threadIdx.x; it < (long)p1 * C4; it += 256) {
    const float4 v = ld4(x + base + it * 4);
    float4 d = ld4(dy + base + it * 4);
    float4 h;
    h.x = (v.x - mu) * rs; h.y = (v.y - mu) * rs; h.z = (v.z - mu) * rs; h.w = (v.w - mu) * rs;
    if (relu) {
      if (!(h.x * ga.x + be.x > 0.f)) d.x = 0.f;
      if (!(h.y * ga.y + be.y > 0.f)) d.y = 0.f;
      if (!(h.z * ga.z + be.z > 0.f)) d.z = 0.f;
      if (!(h.w * ga.w + be.w > 0.f)) d.w = 0.f;
    }
    a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
    b.x += d.x * h.x; b.y += d.y * h.y; b.z += d.z * h.z; b.w += d.w * h.w;
  }
  ra[threadIdx.x] = a; rb[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x < C4) {
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    for (int t = threadIdx.x; t < 256; t += C4) {
      sa.x += ra[t].x; sa.y += ra[t].y; sa.z += ra[t].z; sa.w += ra[t].w;
      sb.x += rb[t].x; sb.y += rb[t].y; sb.z += rb[t].z; sb.w += rb[t].w;
    }
    partA[((size_t)n * slabs + slab) * C4 + threadIdx.x] = sa;
    partB[((size_t)n * slabs + slab) * C4 + threadIdx.x] = sb;
  }
}

// Fold: per (n, c): A, B totals; per (n, g): S1 = sum gamma*A, S2 = sum gamma*B; dgamma/dbeta per channel.
// One workgroup of (channel, slab-slice) threads.
__global__ void gn_bwd_fold_kernel(const float* __restrict__ partA, const float* __restrict__ partB,
                                   const float* __restrict__ gamma, float* __restrict__ s1,
                                   float* __restrict__ s2, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                   int N, int C, int G, int slabs, int Cpad) {
  // blockDim = Cpad * S: slice j of channel c folds slabs j, j+S, ...; the S slices are then added in slice order
  extern __shared__ float sh[];  // [S][2][Cpad] slice sums, then [2][C] gamma-weighted totals
  const int S = blockDim.x / Cpad;
  const int c = threadIdx.x % Cpad, j = threadIdx.x / Cpad, cpg = C / G;
  float* red = sh;
  float* gw = sh + (size_t)S * 2 * Cpad;
  float dg = 0.f, db = 0.f;
  for (int n = 0; n < N; ++n) {
    float a = 0.f, b = 0.f;
    if (c < C)
      for (int sl = j; sl < slabs; sl += S) {
        a += partA[((size_t)n * slabs + sl) * C + c];
        b += partB[((size_t)n * slabs + sl) * C + c];
      }
    __syncthreads();
    red[(j * 2 + 0) * Cpad + c] = a;
    red[(j * 2 + 1) * Cpad + c] = b;
    __syncthreads();
    if (j == 0 && c < C) {
      a = 0.f; b = 0.f;
      for (int jj = 0; jj < S; ++jj) { a += red[(jj * 2 + 0) * Cpad + c]; b += red[(jj * 2 + 1) * Cpad + c]; }
      dg += b; db += a;
      gw[c] = gamma[c] * a; gw[C + c] = gamma[c] * b;
    }
    __syncthreads();
    if (j == 0 && c < G) {
      float x1 = 0.f, x2 = 0.f;
      for (int k = 0; k < cpg; ++k) { x1 += gw[c * cpg + k]; x2 += gw[C + c * cpg + k]; }
      s1[n * G + c] = x1; s2[n * G + c] = x2;
    }
  }
  if (j == 0 && c < C) { dgamma[c] = dg; dbeta[c] = db; }
}

// dx = rstd * (gamma * dyr - (S1 + xhat * S2) / m)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ s1,
                                                           const float* __restrict__ s2, float* __restrict__ dx,
                                                           unsigned short* __restrict__ dx_hi,
                                                           unsigned short* __restrict__ dx_lo,
                                                           long HW, int C, int G, int relu, long total4) {
  const int C4 = C / 4, cpg4 = C4 / G;
  const float inv_m = 1.f / ((float)HW * (float)(C / G));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const long n = i / C4 / HW;
    const int g = c4 / cpg4;
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    const float a = s1[n * G + g] * inv_m, b = s2[n * G + g] * inv_m;
    const float4 v = ld4(x + i * 4), ga = ld4(gamma + c4 * 4), be = ld4(beta + c4 * 4);
    float4 d = ld4(dy + i * 4), h, o;
    h.x = (v.x - mu) * rs; h.y = (v.y - mu) * rs; h.z = (v.z - mu) * rs; h.w = (v.w - mu) * rs;
    if (relu) {
      if (!(h.x * ga.x + be.x > 0.f)) d.x = 0.f;
      if (!(h.y * ga.y + be.y > 0.f)) d.y = 0.f;
      if (!(h.z * ga.z + be.z > 0.f)) d.z = 0.f;
      if (!(h.w * ga.w + be.w > 0.f)) d.w = 0.f;
    }
    o.x = rs * (ga.x * d.x - a - h.x * b);
    o.y = rs * (ga.y * d.y - a - h.y * b);
    o.z = rs * (ga.z * d.z - a - h.z * b);
    o.w = rs * (ga.w * d.w - a - h.w * b);
    st4(dx + i * 4, o);
    if (dx_hi) put_planes4(dx_hi, dx_lo, i, o);
  }
}

// ---- bilinear x2 (align_corners = False): src coordinate of dst o is o/2 - 0.25 clamped at 0 -----------
struct Lerp { int i0, i1; float l; };
__device__ __forceinline__ Lerp lerp_of(int o, int n_src) {
  float s = ((float)o + 0.5f) * 0.5f - 0.5f;
  if (s < 0.f) s = 0.f;
  Lerp r;
  r.i0 = (int)s;
  r.i1 = r.i0 < n_src - 1 ? r.i0 + 1 : r.i0;
  r.l = s - (float)r.i0;
  return r;
}

#ifdef JTSM_DIAG_UP2   // diagnostic build: the kernel checks itself (second look at its operands past the caches,
                       // second evaluation one product at a time) and books what it finds
__device__ unsigned int g_up2_diag[8 + 64 * 40];
typedef float diag2_fx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_system(const float* p) {
  diag2_fx4 t;
  asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(t) : "v"(p) : "memory");
  return make_float4(t[0], t[1], t[2], t[3]);
}
__device__ __forceinline__ float one_at_a_time(float w00, float a, float w01, float b, float w10, float c, float w11, float d) {
  float r = __fmul_rn(w00, a);
  asm volatile("" : "+v"(r));
  float t = __fmul_rn(w01, b);
  asm volatile("" : "+v"(t));
  r = __fadd_rn(r, t);
  asm volatile("" : "+v"(r));
  t = __fmul_rn(w10, c);
  asm volatile("" : "+v"(t));
  r = __fadd_rn(r, t);
  asm volatile("" : "+v"(r));
  t = __fmul_rn(w11, d);
  asm volatile("" : "+v"(t));
  return __fadd_rn(r, t);
}
#endif

__global__ __launch_bounds__(256) void up2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      unsigned short* __restrict__ y_hi,
                                                      unsigned short* __restrict__ y_lo, int N,
                                                      int H, int W, int C4, long total4) {
  const int Ho = 2 * H, Wo = 2 * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int ow = (int)(t % Wo); t /= Wo;
    const int oh = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const Lerp a = lerp_of(oh, H), b = lerp_of(ow, W);
    const float* base = x + (size_t)n * H * W * C4 * 4;
    float4 v00 = ld4(base + ((size_t)(a.i0 * W + b.i0) * C4 + c) * 4);
    float4 v01 = ld4(base + ((size_t)(a.i0 * W + b.i1) * C4 + c) * 4);
    float4 v10 = ld4(base + ((size_t)(a.i1 * W + b.i0) * C4 + c) * 4);
    float4 v11 = ld4(base + ((size_t)(a.i1 * W + b.i1) * C4 + c) * 4);
#ifdef JTSM_DIAG_UP2_LOADWAIT   // diagnostic build: all four loads complete, then idle cycles, before the first packed multiply reads them
    {
      typedef float lw_fx4 __attribute__((ext_vector_type(4)));
      lw_fx4 t0 = {v00.x, v00.y, v00.z, v00.w}, t1 = {v01.x, v01.y, v01.z, v01.w}, t2 = {v10.x, v10.y, v10.z, v10.w}, t3 = {v11.x, v11.y, v11.z, v11.w};
      asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
      v00 = make_float4(t0[0], t0[1], t0[2], t0[3]);
      v01 = make_float4(t1[0], t1[1], t1[2], t1[3]);
      v10 = make_float4(t2[0], t2[1], t2[2], t2[3]);
      v11 = make_float4(t3[0], t3[1], t3[2], t3[3]);
    }
#endif
    const float w00 = (1.f - a.l) * (1.f - b.l), w01 = (1.f - a.l) * b.l, w10 = a.l * (1.f - b.l), w11 = a.l * b.l;
    float4 o;
    o.x = w00 * v00.x + w01 * v01.x + w10 * v10.x + w11 * v11.x;
    o.y = w00 * v00.y + w01 * v01.y + w10 * v10.y + w11 * v11.y;
    o.z = w00 * v00.z + w01 * v01.z + w10 * v10.z + w11 * v11.z;
    o.w = w00 * v00.w + w01 * v01.w + w10 * v10.w + w11 * v11.w;
#ifdef JTSM_DIAG_UP2_NOPS   // diagnostic build: idle cycles between the last packed multiply-add and the store that reads it
    {
      typedef float nops_fx4 __attribute__((ext_vector_type(4)));
      nops_fx4 t = {o.x, o.y, o.z, o.w};
#ifdef JTSM_DIAG_UP2_SLEEP
      asm volatile("s_sleep " JTSM_DIAG_UP2_SLEEP : "+v"(t));      // (64 cycles per unit)
#else
      asm volatile("s_nop 7\n\ts_nop 7" : "+v"(t));
#endif
      o = make_float4(t[0], t[1], t[2], t[3]);
    }
#endif
#ifdef JTSM_DIAG_UP2_PRE   // diagnostic build: a second evaluation (one product at a time, same operand registers) in FRONT of the store
    {
      const float p0 = one_at_a_time(w00, v00.x, w01, v01.x, w10, v10.x, w11, v11.x);
      const float p1 = one_at_a_time(w00, v00.y, w01, v01.y, w10, v10.y, w11, v11.y);
      const float p2 = one_at_a_time(w00, v00.z, w01, v01.z, w10, v10.z, w11, v11.z);
      const float p3 = one_at_a_time(w00, v00.w, w01, v01.w, w10, v10.w, w11, v11.w);
      const bool pre_differs = fabsf(p0 - o.x) > 1e-5f * (1.f + fabsf(p0)) || fabsf(p1 - o.y) > 1e-5f * (1.f + fabsf(p1)) ||
                               fabsf(p2 - o.z) > 1e-5f * (1.f + fabsf(p2)) || fabsf(p3 - o.w) > 1e-5f * (1.f + fabsf(p3));
      if (pre_differs) atomicAdd(&g_up2_diag[6], 1u);
    }
#endif
#ifdef JTSM_DIAG_UP2_PLAIN_ADDRESS   // diagnostic build: the store's address in a register of its own (no immediate offset folded in)
    {
      float* yp = y + i * 4;
      asm volatile("" : "+v"(yp));
      st4(yp, o);
    }
#else
    st4(y + i * 4, o);
#endif
    if (y_hi) put_planes4(y_hi, y_lo, i, o);
#ifdef JTSM_DIAG_UP2
    {
      const float4 r00 = ld4_system(base + ((size_t)(a.i0 * W + b.i0) * C4 + c) * 4);
      const float4 r01 = ld4_system(base + ((size_t)(a.i0 * W + b.i1) * C4 + c) * 4);
      const float4 r10 = ld4_system(base + ((size_t)(a.i1 * W + b.i0) * C4 + c) * 4);
      const float4 r11 = ld4_system(base + ((size_t)(a.i1 * W + b.i1) * C4 + c) * 4);
      const float fv[16] = {v00.x, v00.y, v00.z, v00.w, v01.x, v01.y, v01.z, v01.w, v10.x, v10.y, v10.z, v10.w, v11.x, v11.y, v11.z, v11.w};
      const float fr[16] = {r00.x, r00.y, r00.z, r00.w, r01.x, r01.y, r01.z, r01.w, r10.x, r10.y, r10.z, r10.w, r11.x, r11.y, r11.z, r11.w};
      const float fo[4] = {o.x, o.y, o.z, o.w};
      float f2[4];
      bool operands_differ = false, result_differs = false;
#pragma unroll
      for (int k = 0; k < 16; ++k) operands_differ |= __float_as_uint(fv[k]) != __float_as_uint(fr[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        f2[k] = one_at_a_time(w00, fr[k], w01, fr[4 + k], w10, fr[8 + k], w11, fr[12 + k]);
        result_differs |= fabsf(f2[k] - fo[k]) > 1e-5f * (1.f + fabsf(f2[k]));
      }
#ifdef JTSM_DIAG_UP2_READBACK
      // read the stored piece back past the caches: did the store put `o` there?
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float4 rb = ld4_system(y + i * 4);
      const bool stored_differs = __float_as_uint(rb.x) != __float_as_uint(o.x) || __float_as_uint(rb.y) != __float_as_uint(o.y) ||
                                  __float_as_uint(rb.z) != __float_as_uint(o.z) || __float_as_uint(rb.w) != __float_as_uint(o.w);
      if (stored_differs) atomicAdd(&g_up2_diag[5], 1u);
#endif
      atomicAdd(&g_up2_diag[0], 1u);
      if (operands_differ) atomicAdd(&g_up2_diag[1], 1u);
      if (result_differs) atomicAdd(&g_up2_diag[2], 1u);
      if (result_differs && !operands_differ) atomicAdd(&g_up2_diag[3], 1u);
      if (operands_differ || result_differs) {
        const unsigned slot = atomicAdd(&g_up2_diag[4], 1u);
        if (slot < 64) {
          unsigned* d = g_up2_diag + 8 + slot * 40;
          d[0] = (unsigned)(i & 0xffffffff); d[1] = threadIdx.x; d[2] = blockIdx.x; d[3] = (operands_differ ? 1u : 0u) | (result_differs ? 2u : 0u);
#pragma unroll
          for (int k = 0; k < 16; ++k) { d[4 + k] = __float_as_uint(fv[k]); d[20 + k] = __float_as_uint(fr[k]); }
#pragma unroll
          for (int k = 0; k < 4; ++k) d[36 + k] = __float_as_uint(fo[k]);
        }
      }
    }
#endif
  }
}

// backward as a gather: source pixel (h,w) collects from the <= 4x4 destination pixels that sample it.
__device__ __forceinline__ float weight_on(int o, int n_src, int i) {
  const Lerp r = lerp_of(o, n_src);
  return (r.i0 == i ? 1.f - r.l : 0.f) + (r.i1 == i ? r.l : 0.f);
}
__global__ __launch_bounds__(256) void up2_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx, int N,
                                                      int H, int W, int C4, long total4) {
  const int Ho = 2 * H, Wo = 2 * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4);
    long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    const float* base = gy + (size_t)n * Ho * Wo * C4 * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oh = max(2 * h - 1, 0); oh <= min(2 * h + 2, Ho - 1); ++oh) {
      const float wy = weight_on(oh, H, h);
      if (wy == 0.f) continue;
      for (int ow = max(2 * w - 1, 0); ow <= min(2 * w + 2, Wo - 1); ++ow) {
        const float wx = weight_on(ow, W, w);
        if (wx == 0.f) continue;
        const float4 g = ld4(base + ((size_t)(oh * Wo + ow) * C4 + c) * 4);
        const float ww = wy * wx;
        acc.x += ww * g.x; acc.y += ww * g.y; acc.z += ww * g.z; acc.w += ww * g.w;
      }
    }
    st4(gx + i * 4, acc);
  }
}


// ---- fused bilinear xS up-sampling + softmax cross-entropy (semantic_seg.py:179-188) ------------------
// The reference materialises the (N, 54, H, W) up-sampled logits (226 MB per 1024^2 image), then
// log_softmax, then nll_loss — ~0.7 GB of HBM traffic per image forward+backward.  Here the full-resolution
// logits never exist: every output pixel's logits are interpolated on the fly from the stride-S map
// (wavefront per pixel, lanes = classes, C <= 64), reduced to a log-sum-exp with wavefront shuffles, and
// only lse (4 B per pixel) is kept for the backward, which is a GATHER over the <= (2S)^2 output pixels
// that touch each source pixel (no atomics, deterministic).
struct LerpS { int i0, i1; float l0, l1; };
__device__ __forceinline__ LerpS lerp_scale(int o, int n_src, float inv_scale) {
  float s = ((float)o + 0.5f) * inv_scale - 0.5f;
  if (s < 0.f) s = 0.f;
  LerpS r;
  r.i0 = (int)s;
  r.i1 = r.i0 < n_src - 1 ? r.i0 + 1 : r.i0;
  r.l1 = s - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float interp_logit(const float* __restrict__ zn, int Ws, int ldc, const LerpS& a,
                                              const LerpS& b, int c) {
  const float v00 = zn[((size_t)a.i0 * Ws + b.i0) * ldc + c], v01 = zn[((size_t)a.i0 * Ws + b.i1) * ldc + c];
  const float v10 = zn[((size_t)a.i1 * Ws + b.i0) * ldc + c], v11 = zn[((size_t)a.i1 * Ws + b.i1) * ldc + c];
  return a.l0 * (b.l0 * v00 + b.l1 * v01) + a.l1 * (b.l0 * v10 + b.l1 * v11);   // ATen's association
}

// Forward: one THREAD per output pixel (neighbouring pixels share their four taps, so the tap loads of a
// wave collapse onto a few cache lines); the <= 64 interpolated logits stay in registers.
__global__ __launch_bounds__(256) void ce_up_fwd_kernel(const float* __restrict__ z, int ldc, int C,
                                                        const long* __restrict__ target, float* __restrict__ lse,
                                                        float* __restrict__ part, int N, int Hs, int Ws, int S,
                                                        long ignore) {
  __shared__ float red[4][2];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int H = Hs * S, W = Ws * S;
  const long npix = (long)N * H * W;
  const float inv = 1.f / (float)S;
  const int nch = ldc >> 2;   // float4 chunks per pixel row (ldc % 4 == 0, <= 16)
  float loss = 0.f, cnt = 0.f;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
    const long t = target[p];
    const int ow = (int)(p % W);
    const long q = p / W;
    const int oh = (int)(q % H), n = (int)(q / H);
    const LerpS a = lerp_scale(oh, Hs, inv), b = lerp_scale(ow, Ws, inv);
    const float* zn = z + (size_t)n * Hs * Ws * ldc;
    const float4* p00 = reinterpret_cast<const float4*>(zn + ((size_t)a.i0 * Ws + b.i0) * ldc);
    const float4* p01 = reinterpret_cast<const float4*>(zn + ((size_t)a.i0 * Ws + b.i1) * ldc);
    const float4* p10 = reinterpret_cast<const float4*>(zn + ((size_t)a.i1 * Ws + b.i0) * ldc);
    const float4* p11 = reinterpret_cast<const float4*>(zn + ((size_t)a.i1 * Ws + b.i1) * ldc);
    float v[64];
    float mx = -3.0e38f, vt = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (j < nch) {
        const float4 v00 = p00[j], v01 = p01[j], v10 = p10[j], v11 = p11[j];
        const float x00[4] = {v00.x, v00.y, v00.z, v00.w}, x01[4] = {v01.x, v01.y, v01.z, v01.w};
        const float x10[4] = {v10.x, v10.y, v10.z, v10.w}, x11[4] = {v11.x, v11.y, v11.z, v11.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 4 * j + e;
          float x = a.l0 * (b.l0 * x00[e] + b.l1 * x01[e]) + a.l1 * (b.l0 * x10[e] + b.l1 * x11[e]);   // ATen's association
          if (c >= C) x = -3.0e38f;
          v[c] = x;
          mx = fmaxf(mx, x);
          vt = (long)c == t ? x : vt;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * j + e] = -3.0e38f;
      }
    }
    float sm = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      if (j < nch) {
#pragma unroll
        for (int e = 0; e < 4; ++e) sm += __expf(v[4 * j + e] - mx);
      }
    const float l = mx + __logf(sm);
    lse[p] = l;
    if (t != ignore) {
      loss += l - vt;
      cnt += 1.f;
    }
  }
  loss = wsum(loss);
  cnt = wsum(cnt);
  if (lane == 0) { red[wv][0] = loss; red[wv][1] = cnt; }
  __syncthreads();
  if (threadIdx.x < 2)
    part[blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0] = mean loss over non-ignored pixels, out[1] = their count (fixed-order tree, deterministic)
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ part, int nblocks,
                                                        float* __restrict__ out) {
  __shared__ double sl[256], sc[256];
  double l = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) { l += part[2 * b]; c += part[2 * b + 1]; }
  sl[threadIdx.x] = l; sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = (float)(sl[0] / sc[0]);
    out[1] = (float)sc[0];
  }
}

// l0 * a + l1 * b with the contraction spelled out: the two backward forms below must round alike
__device__ __forceinline__ float mix2(float l0, float a, float l1, float b) { return __fmaf_rn(l0, a, l1 * b); }

// dz[n,h,w,c] = up/count * sum over output pixels o touching (h,w) of weight(o -> (h,w)) * (softmax_o[c] - [c == t_o])
// A gather (no atomics, deterministic): 16 lanes per source pixel, one float4 of channels per lane.  For one
// output column the three source rows h-1, h, h+1 are interpolated along x once and reused by all of that
// column's output rows.
__global__ __launch_bounds__(256) void ce_up_bwd_kernel(const float* __restrict__ z, int ldc, int C,
                                                        const long* __restrict__ target,
                                                        const float* __restrict__ lse, const float* __restrict__ fin,
                                                        const float* __restrict__ upstream, float* __restrict__ dz,
                                                        int N, int Hs, int Ws, int S, long ignore) {
  const int j = threadIdx.x & 15;
  const int H = Hs * S, W = Ws * S;
  const long nsrc = (long)N * Hs * Ws;
  const float inv = 1.f / (float)S;
  const float k = (upstream ? *upstream : 1.f) / fin[1];
  const bool live = 4 * j < ldc;
  const int c0 = 4 * j;
  for (long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; p < nsrc; p += ((long)gridDim.x * blockDim.x) >> 4) {
    if (!live) continue;
    const int w = (int)(p % Ws);
    const long q = p / Ws;
    const int h = (int)(q % Hs), n = (int)(q / Hs);
    const float* zn = z + (size_t)n * Hs * Ws * ldc + c0;
    const int hm = max(h - 1, 0), hp = min(h + 1, Hs - 1);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // rows / columns of the output that can put weight on (h, w): exact for even S, one spare each side for odd S
    const int oh0 = max(S * h - (S + 1) / 2, 0), oh1 = min(S * h + S + (S + 1) / 2, H);
    const int ow0 = max(S * w - (S + 1) / 2, 0), ow1 = min(S * w + S + (S + 1) / 2, W);
    for (int ow = ow0; ow < ow1; ++ow) {
      const LerpS b = lerp_scale(ow, Ws, inv);
      const float wx = (b.i0 == w ? b.l0 : 0.f) + (b.i1 == w ? b.l1 : 0.f);
      if (wx == 0.f) continue;
      float r[3][4];   // x-interpolated logits of source rows h-1, h, h+1 at this output column
      const int rows[3] = {hm, h, hp};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float4 u0 = *reinterpret_cast<const float4*>(zn + ((size_t)rows[i] * Ws + b.i0) * ldc);
        const float4 u1 = *reinterpret_cast<const float4*>(zn + ((size_t)rows[i] * Ws + b.i1) * ldc);
        r[i][0] = mix2(b.l0, u0.x, b.l1, u1.x); r[i][1] = mix2(b.l0, u0.y, b.l1, u1.y);
        r[i][2] = mix2(b.l0, u0.z, b.l1, u1.z); r[i][3] = mix2(b.l0, u0.w, b.l1, u1.w);
      }
      for (int oh = oh0; oh < oh1; ++oh) {
        const LerpS a = lerp_scale(oh, Hs, inv);
        const float wy = (a.i0 == h ? a.l0 : 0.f) + (a.i1 == h ? a.l1 : 0.f);
        if (wy == 0.f) continue;
        const long o = ((long)n * H + oh) * W + ow;
        const long t = target[o];
        if (t == ignore) continue;
        const float l = lse[o], wgt = wy * wx;
        const bool lo0 = a.i0 < h, hi0 = a.i0 > h, lo1 = a.i1 < h, hi1 = a.i1 > h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t0 = lo0 ? r[0][e] : (hi0 ? r[2][e] : r[1][e]);
          const float t1 = lo1 ? r[0][e] : (hi1 ? r[2][e] : r[1][e]);
          const float v = mix2(a.l0, t0, a.l1, t1);
          const float pr = c0 + e < C ? __expf(v - l) : 0.f;
          acc[e] = __fmaf_rn(wgt, pr - ((long)(c0 + e) == t ? 1.f : 0.f), acc[e]);
        }
      }
    }
    *reinterpret_cast<float4*>(dz + (size_t)p * ldc + c0) =
        make_float4(c0 + 0 < C ? acc[0] * k : 0.f, c0 + 1 < C ? acc[1] * k : 0.f, c0 + 2 < C ? acc[2] * k : 0.f,
                    c0 + 3 < C ? acc[3] * k : 0.f);
  }
}

inline int grid_for(long total) { return (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192); }

// The same gather through LDS, for S = 4 (the JTSM path: logits at stride 4).  In the form above every output pixel's
// soft-max is evaluated by each of the (up to) four source pixels it feeds — 4 x the exponentials, and the kernel is
// bound by exactly those (0.29 ms per step at 3 % of the HBM roof).  Here a workgroup owns a 4 x 4 tile of source
// pixels: the 6 x 6 source logits it interpolates from are staged once, each of the <= 20 x 20 output pixels that
// touch the tile gets its (soft-max - one-hot) row computed ONCE into LDS (1.56 x instead of 4 x), and 16 x ldc / 4
// threads gather their source pixel's sum from there — in the order of the form above (output column outer, row
// inner, the same products), so the result is the same bits.
constexpr int CE_TS = 4, CE_S = 4, CE_OT = CE_TS * CE_S + 4, CE_ZT = CE_TS + 2;
struct CeTile { int n, h0, w0, oh_lo, ow_lo, nrow, ncol, zr0, zc0, zr1, zc1; };

// A fixed grid (one workgroup per CU: 111 KB of LDS) strides the tile list, and the next tile's logits, labels and
// log-sum-exps are in flight (one float4 or two, one label per thread) while the current tile is worked on — with one
// workgroup per CU nothing else would hide those loads.
__global__ __launch_bounds__(512) void ce_up_bwd_tiled_kernel(const float* __restrict__ z, int ldc, int C,
                                                              const long* __restrict__ target,
                                                              const float* __restrict__ lse, const float* __restrict__ fin,
                                                              const float* __restrict__ upstream, float* __restrict__ dz,
                                                              int N, int Hs, int Ws, long ignore, int tiles) {
  __shared__ __attribute__((aligned(16))) float zs[CE_ZT * CE_ZT * 64];     // source logits of the tile + 1 ring
  __shared__ __attribute__((aligned(16))) float gs[CE_OT * CE_OT * 64];     // (soft-max - one-hot) of the outputs
  __shared__ int tgs[CE_OT * CE_OT];                                        // their labels (-1: ignored) ...
  __shared__ float ls[CE_OT * CE_OT];                                       // ... and log-sum-exp
  constexpr int S = CE_S;
  const int t = threadIdx.x, nq = ldc >> 2;
  const int tiles_x = (Ws + CE_TS - 1) / CE_TS, tiles_y = (Hs + CE_TS - 1) / CE_TS;
  const int H = Hs * S, W = Ws * S;
  const float inv = 1.f / (float)S;
  const float k = (upstream ? *upstream : 1.f) / fin[1];
  // a / d by multiply-high, exact for a * d < 2^32 (a < 2^15 here)
  const unsigned magic_nq = (unsigned)((0x100000000ull + (unsigned)nq - 1) / (unsigned)nq);
  const int nz = CE_ZT * CE_ZT * nq;   // float4s of the staged logits: <= 576, two per thread at most

  auto geom = [&](int tile) {
    CeTile g;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    g.n = tile / tiles_y;
    g.h0 = (tile - g.n * tiles_y) * CE_TS;
    g.w0 = tx * CE_TS;
    // outputs that can put weight on the tile (the form above: S*h - (S+1)/2 .. S*h + S + (S+1)/2)
    g.oh_lo = max(S * g.h0 - S / 2, 0);
    g.ow_lo = max(S * g.w0 - S / 2, 0);
    g.nrow = min(S * (g.h0 + CE_TS) + S / 2, H) - g.oh_lo;
    g.ncol = min(S * (g.w0 + CE_TS) + S / 2, W) - g.ow_lo;
    g.zr0 = max(g.h0 - 1, 0);
    g.zc0 = max(g.w0 - 1, 0);
    g.zr1 = min(g.h0 + CE_TS, Hs - 1);
    g.zc1 = min(g.w0 + CE_TS, Ws - 1);
    return g;
  };
  auto fetch = [&](const CeTile& g, float4 (&pz)[2], long& ptg, float& pl) {
    const float* zn = z + (size_t)g.n * Hs * Ws * ldc;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + 512 * u;
      const int pix = (int)__umulhi((unsigned)i, magic_nq), q = i - pix * nq;
      const int sy = g.zr0 + pix / CE_ZT, sx = g.zc0 + pix % CE_ZT;
      if (i < nz && sy <= g.zr1 && sx <= g.zc1)
        pz[u] = *reinterpret_cast<const float4*>(zn + ((size_t)sy * Ws + sx) * ldc + 4 * q);
    }
    if (t < g.nrow * g.ncol) {   // (<= 400 of the 512)
      const int oy = t / g.ncol, ox = t - oy * g.ncol;
      const long og = ((long)g.n * H + g.oh_lo + oy) * W + g.ow_lo + ox;
      ptg = target[og];
      pl = lse[og];
    }
  };

  int tile = blockIdx.x;
  if (tile >= tiles) return;
  CeTile cur = geom(tile);
  float4 pz[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
  long ptg = ignore;
  float pl = 0.f;
  fetch(cur, pz, ptg, pl);
  for (; tile < tiles; tile += gridDim.x) {
    // (zs / tgs / ls were last read before the previous tile's middle barrier)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = t + 512 * u;
      if (i < nz) *reinterpret_cast<float4*>(zs + 4 * i) = pz[u];    // (pix * ldc + 4 q = 4 i; rows outside the map: stale, never read)
    }
    if (t < cur.nrow * cur.ncol) {
      tgs[t] = ptg == ignore ? -1 : (int)ptg;
      ls[t] = pl;
    }
    __syncthreads();
    const CeTile g = cur;
    if (tile + (int)gridDim.x < tiles) {
      cur = geom(tile + gridDim.x);
      fetch(cur, pz, ptg, pl);
    }
    const unsigned magic_ncol = (unsigned)((0x100000000ull + (unsigned)g.ncol - 1) / (unsigned)g.ncol);
    const int nitem = g.nrow * g.ncol * nq;
#pragma unroll 2
    for (int it = t; it < nitem; it += 512) {
      const int o = (int)__umulhi((unsigned)it, magic_nq), q = it - o * nq;
      const int oy = (int)__umulhi((unsigned)o, magic_ncol), ox = o - oy * g.ncol;
      const int oh = g.oh_lo + oy, ow = g.ow_lo + ox;
      const int tg = tgs[o];
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
      if (tg >= 0) {
        const float l = ls[o];
        const LerpS a = lerp_scale(oh, Hs, inv), b = lerp_scale(ow, Ws, inv);
        const float4 u00 = *reinterpret_cast<const float4*>(zs + ((a.i0 - g.zr0) * CE_ZT + (b.i0 - g.zc0)) * ldc + 4 * q);
        const float4 u01 = *reinterpret_cast<const float4*>(zs + ((a.i0 - g.zr0) * CE_ZT + (b.i1 - g.zc0)) * ldc + 4 * q);
        const float4 u10 = *reinterpret_cast<const float4*>(zs + ((a.i1 - g.zr0) * CE_ZT + (b.i0 - g.zc0)) * ldc + 4 * q);
        const float4 u11 = *reinterpret_cast<const float4*>(zs + ((a.i1 - g.zr0) * CE_ZT + (b.i1 - g.zc0)) * ldc + 4 * q);
        const float r0[4] = {mix2(b.l0, u00.x, b.l1, u01.x), mix2(b.l0, u00.y, b.l1, u01.y),
                             mix2(b.l0, u00.z, b.l1, u01.z), mix2(b.l0, u00.w, b.l1, u01.w)};
        const float r1[4] = {mix2(b.l0, u10.x, b.l1, u11.x), mix2(b.l0, u10.y, b.l1, u11.y),
                             mix2(b.l0, u10.z, b.l1, u11.z), mix2(b.l0, u10.w, b.l1, u11.w)};
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = mix2(a.l0, r0[j], a.l1, r1[j]);
          const float pr = 4 * q + j < C ? __expf(v - l) : 0.f;
          e[j] = pr - (4 * q + j == tg ? 1.f : 0.f);
        }
        d = make_float4(e[0], e[1], e[2], e[3]);
      }
      *reinterpret_cast<float4*>(gs + 4 * it) = d;    // (o * ldc + 4 q = 4 it)
    }
    __syncthreads();
    if (t < CE_TS * CE_TS * nq) {
      const int sp = (int)__umulhi((unsigned)t, magic_nq), q = t - sp * nq;
      const int h = g.h0 + sp / CE_TS, w = g.w0 + sp % CE_TS;
      if (h < Hs && w < Ws) {
        // the 8 x 8 output pixels around (h, w): weights first (zero where the pixel does not interpolate from (h, w)
        // or lies outside the map — its LDS row is then some other finite row, times zero), then 64 independent reads
        constexpr int R = 2 * CE_S;
        float wy[R], wx[R];
        int ry[R], rx[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
          const int oh = S * h - S / 2 + i, ow = S * w - S / 2 + i;
          const LerpS a = lerp_scale(oh, Hs, inv), bb = lerp_scale(ow, Ws, inv);
          const bool yin = oh >= 0 && oh < H, xin = ow >= 0 && ow < W;
          wy[i] = yin ? (a.i0 == h ? a.l0 : 0.f) + (a.i1 == h ? a.l1 : 0.f) : 0.f;
          wx[i] = xin ? (bb.i0 == w ? bb.l0 : 0.f) + (bb.i1 == w ? bb.l1 : 0.f) : 0.f;
          ry[i] = (min(max(oh - g.oh_lo, 0), g.nrow - 1)) * g.ncol;
          rx[i] = min(max(ow - g.ow_lo, 0), g.ncol - 1);
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* gq = gs + 4 * q;
#pragma unroll
        for (int ix = 0; ix < R; ++ix) {
#pragma unroll
          for (int iy = 0; iy < R; ++iy) {
            const float wgt = wy[iy] * wx[ix];
            const float4 d = *reinterpret_cast<const float4*>(gq + (ry[iy] + rx[ix]) * ldc);
            acc[0] = __fmaf_rn(wgt, d.x, acc[0]); acc[1] = __fmaf_rn(wgt, d.y, acc[1]);
            acc[2] = __fmaf_rn(wgt, d.z, acc[2]); acc[3] = __fmaf_rn(wgt, d.w, acc[3]);
          }
        }
        const int c0 = 4 * q;
        *reinterpret_cast<float4*>(dz + (((size_t)g.n * Hs + h) * Ws + w) * ldc + c0) =
            make_float4(c0 + 0 < C ? acc[0] * k : 0.f, c0 + 1 < C ? acc[1] * k : 0.f, c0 + 2 < C ? acc[2] * k : 0.f,
                        c0 + 3 < C ? acc[3] * k : 0.f);
      }
    }
    // (gs is next written after the next tile's first barrier)
  }
}
inline size_t a16(size_t b) { return (b + 15) & ~(size_t)15; }
inline int gn_slabs(long HW) { return (int)((HW + GN_SLAB_ROWS - 1) / GN_SLAB_ROWS); }

int gn_check(int N, long HW, int C, int G) {
  JTSM_REQUIRE(N >= 0 && HW > 0 && C > 0 && G > 0 && C % G == 0 && (C / G) % 4 == 0,
               "group_norm: channels per group must be a multiple of 4 (C=%d G=%d)", C, G);
  JTSM_REQUIRE(256 % (C / 4) == 0 && C <= 1024, "group_norm: C/4 must divide 256 (C=%d)", C);
  return JTSM_OK;
}

}  // namespace
}  // namespace jtsm

using namespace jtsm;

extern "C" {

size_t jtsm_group_norm_workspace_bytes(int N, long HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 16;
  // forward needs N*slabs*(C/4) float2; backward 2 * N*slabs*C floats + 2*N*G (<= C) floats
  return a16((size_t)N * gn_slabs(HW) * C * sizeof(float) * 2) + a16((size_t)2 * N * C * sizeof(float)) + 16;
}

int jtsm_group_norm_forward_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                float* rstd, void* workspace, int N, long HW, int C, int G, float eps, int relu,
                                void* stream) {
  int rc = gn_check(N, HW, C, G);
  if (rc) return rc;
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(x && gamma && beta && y && mean && rstd && workspace, "group_norm: null pointer");
  hipStream_t st = as_stream(stream);
  const int slabs = gn_slabs(HW);
  float2* part = reinterpret_cast<float2*>(workspace);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(slabs, N), dim3(256), 0, st, x, part, (int)HW, C, slabs);
  JTSM_REQUIRE(G <= 64, "group_norm: at most 64 groups (G=%d)", G);
  hipLaunchKernelGGL(gn_fold_kernel, dim3(N), dim3(64 * 16), 0, st, part, mean, rstd, (int)HW, C, G, slabs, eps);
  const long total4 = (long)N * HW * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, st, x, mean, rstd, gamma, beta, y, HW,
                     C, G, relu, total4);
  JTSM_CHECK_LAUNCH("group_norm forward");
  return JTSM_OK;
}

int jtsm_group_norm_backward_f32(const float* x, const float* dy, const float* gamma, const float* beta,
                                 const float* mean, const float* rstd, float* dx, uint16_t* dx_hi, uint16_t* dx_lo,
                                 float* dgamma, float* dbeta, void* workspace, int N, long HW, int C, int G, int relu,
                                 void* stream) {
  int rc = gn_check(N, HW, C, G);
  if (rc) return rc;
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(x && dy && gamma && beta && mean && rstd && dx && dgamma && dbeta && workspace,
               "group_norm backward: null pointer");
  JTSM_REQUIRE((dx_hi == nullptr) == (dx_lo == nullptr), "group_norm backward: give both output planes or neither");
  hipStream_t st = as_stream(stream);
  const int slabs = gn_slabs(HW);
  char* w = reinterpret_cast<char*>(workspace);
  float* partA = reinterpret_cast<float*>(w);
  float* partB = partA + (size_t)N * slabs * C;
  float* s1 = reinterpret_cast<float*>(w + a16((size_t)N * slabs * C * sizeof(float) * 2));
  float* s2 = s1 + (size_t)N * C;
  hipLaunchKernelGGL(gn_bwd_stats_kernel, dim3(slabs, N), dim3(256), 0, st, x, dy, mean, rstd, gamma, beta,
                     reinterpret_cast<float4*>(partA), reinterpret_cast<float4*>(partB), (int)HW, C, G, slabs, relu);
  const int cpad = ((C + 63) / 64) * 64, slices = 1024 / cpad;
  hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(1), dim3(cpad * slices), ((size_t)slices * 2 * cpad + 2 * C) * sizeof(float),
                     st, partA, partB, gamma, s1, s2, dgamma, dbeta, N, C, G, slabs, cpad);
  const long total4 = (long)N * HW * (C / 4);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, st, x, dy, mean, rstd, gamma, beta,
                     s1, s2, dx, dx_hi, dx_lo, HW, C, G, relu, total4);
  JTSM_CHECK_LAUNCH("group_norm backward");
  return JTSM_OK;
}

int jtsm_upsample_bilinear2x_forward_f32(const float* x, float* y, uint16_t* y_hi, uint16_t* y_lo, int N, int H, int W,
                                         int C, void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x: bad sizes");
  const long total4 = (long)N * 2 * H * 2 * W * (C / 4);
  if (total4 == 0) return JTSM_OK;
  JTSM_REQUIRE(x && y, "upsample2x: null pointer");
  JTSM_REQUIRE((y_hi == nullptr) == (y_lo == nullptr), "upsample2x forward: give both output planes or neither");
  hipLaunchKernelGGL(up2_fwd_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(stream), x, y, y_hi, y_lo, N, H, W,
                     C / 4, total4);
  JTSM_CHECK_LAUNCH("upsample2x forward");
  return JTSM_OK;
}

int jtsm_upsample_bilinear2x_backward_f32(const float* gy, float* gx, int N, int H, int W, int C, void* stream) {
  JTSM_REQUIRE(N >= 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x backward: bad sizes");
  const long total4 = (long)N * H * W * (C / 4);
  if (total4 == 0) return JTSM_OK;
  JTSM_REQUIRE(gy && gx, "upsample2x backward: null pointer");
  hipLaunchKernelGGL(up2_bwd_kernel, dim3(grid_for(total4)), dim3(256), 0, as_stream(stream), gy, gx, N, H, W,
                     C / 4, total4);
  JTSM_CHECK_LAUNCH("upsample2x backward");
  return JTSM_OK;
}

#define CE_BLOCKS 2048

size_t jtsm_semseg_ce_workspace_bytes(int N, int Hs, int Ws, int S) {
  if (N <= 0 || Hs <= 0 || Ws <= 0 || S <= 0) return 16;
  return a16((size_t)N * Hs * S * Ws * S * sizeof(float)) + a16((size_t)CE_BLOCKS * 2 * sizeof(float)) + 16;
}

int jtsm_semseg_ce_forward_f32(const float* logits, int ld, int C, const int64_t* target, float* out,
                               void* workspace, int N, int Hs, int Ws, int S, long ignore_index, void* stream) {
  JTSM_REQUIRE(N >= 0 && Hs > 0 && Ws > 0 && S > 0 && C > 0 && C <= 64 && ld >= C && ld <= 64 && ld % 4 == 0,
               "semseg_ce: need C <= ld <= 64 and ld %% 4 == 0 (C=%d ld=%d)", C, ld);
  JTSM_REQUIRE(out, "semseg_ce: null out");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && target && workspace, "semseg_ce: null pointer");
  JTSM_REQUIRE(((uintptr_t)logits & 15) == 0, "semseg_ce: logits must be 16-byte aligned");
  hipStream_t st = as_stream(stream);
  float* lse = reinterpret_cast<float*>(workspace);
  float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) +
                                         a16((size_t)N * Hs * S * Ws * S * sizeof(float)));
  const long npix = (long)N * Hs * S * Ws * S;
  int blocks = (int)((npix + 255) / 256 < CE_BLOCKS ? (npix + 255) / 256 : CE_BLOCKS);
  hipLaunchKernelGGL(ce_up_fwd_kernel, dim3(blocks), dim3(256), 0, st, logits, ld, C, (const long*)target, lse, part,
                     N, Hs, Ws, S, ignore_index);
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, st, part, blocks, out);
  JTSM_CHECK_LAUNCH("semseg_ce forward");
  return JTSM_OK;
}

int jtsm_semseg_ce_backward_f32(const float* logits, int ld, int C, const int64_t* target, const float* fwd_out,
                                const float* upstream, float* dlogits, const void* workspace, int N, int Hs, int Ws,
                                int S, long ignore_index, void* stream) {
  JTSM_REQUIRE(N >= 0 && Hs > 0 && Ws > 0 && S > 0 && C > 0 && C <= 64 && ld >= C && ld <= 64 && ld % 4 == 0,
               "semseg_ce backward: need C <= ld <= 64 and ld %% 4 == 0");
  if (N == 0) return JTSM_OK;
  JTSM_REQUIRE(logits && target && fwd_out && dlogits && workspace, "semseg_ce backward: null pointer");
  JTSM_REQUIRE(((uintptr_t)logits & 15) == 0 && ((uintptr_t)dlogits & 15) == 0,
               "semseg_ce backward: logits and dlogits must be 16-byte aligned");
  const float* lse = reinterpret_cast<const float*>(workspace);
  const long nsrc = (long)N * Hs * Ws;
  const char* form = getenv("JTSM_CE_BWD_TILED");   // (read per call: the parity test runs both forms in one process)
  const bool tiled_ok = !form || atoi(form) != 0;
  const long tiles = (long)N * ((Hs + CE_TS - 1) / CE_TS) * ((Ws + CE_TS - 1) / CE_TS);
  if (S == CE_S && tiled_ok && tiles < (1L << 31)) {
    hipLaunchKernelGGL(ce_up_bwd_tiled_kernel, dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(512), 0,
                       as_stream(stream), logits, ld, C, (const long*)target, lse, fwd_out, upstream, dlogits, N, Hs, Ws,
                       ignore_index, (int)tiles);
    JTSM_CHECK_LAUNCH("semseg_ce backward (tiled)");
    return JTSM_OK;
  }
  int blocks = (int)((nsrc + 15) / 16 < 16384 ? (nsrc + 15) / 16 : 16384);
  hipLaunchKernelGGL(ce_up_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), logits, ld, C,
                     (const long*)target, lse, fwd_out, upstream, dlogits, N, Hs, Ws, S, ignore_index);
  JTSM_CHECK_LAUNCH("semseg_ce backward");
  return JTSM_OK;
}

#ifdef JTSM_DIAG_UP2
// diagnostic build only: copy the up-sampling kernel's self-check counters to the host and clear them
int jtsm_diag_up2_read(unsigned int* host, int words) {
  if (words > 8 + 64 * 40) words = 8 + 64 * 40;
  JTSM_CHECK_HIP(hipDeviceSynchronize());
  JTSM_CHECK_HIP(hipMemcpyFromSymbol(host, HIP_SYMBOL(jtsm::g_up2_diag), (size_t)words * 4, 0, hipMemcpyDeviceToHost));
  static unsigned int zeros[8 + 64 * 40];
  JTSM_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(jtsm::g_up2_diag), zeros, sizeof(zeros), 0, hipMemcpyHostToDevice));
  return JTSM_OK;
}
#endif
}  // extern "C"
