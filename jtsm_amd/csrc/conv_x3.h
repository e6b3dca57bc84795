// Split-bf16 ("bf16x3") contraction path — included by conv_igemm.hip inside namespace jtsm::{anonymous}.
//
// fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950, and an fp32-MFMA-dense loop is what pulls the
// chip's clock down.  Here every fp32 operand x is split once, in a streaming pre-pass, into two bf16 planes
//     hi = bf16(x),   lo = bf16(x - hi)            (x = hi + lo up to 2^-17 |x|)
// and the contraction accumulates  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  in fp32 on v_mfma_f32_32x32x16_bf16:
// three matrix instructions per sixteen k instead of eight fp32 ones per sixteen k (5.3x fewer matrix
// cycles), with a relative error per product of ~2^-16 — two orders inside the 1e-4 parity bar.
// The dropped term a_lo*b_lo is below 2^-16 |a b|.
//
// Both operands are K-contiguous: FWD  A = X planes (NHWC), B = W planes ([Cout][tap][Cin]);
//                                 DGRAD A = dY planes,      B = W^T planes ([Cin][tap][Cout], made by the
//                                 transposing split).  A stage holds 32 k: rows of 64 B per plane.
// LDS image per plane: [rows][64 B]; 16-byte chunk c of row r sits at slot c ^ ((r >> 2) & 3) — applied
// to the SOURCE address of the direct-to-LDS load (lane l of a load owns row l>>2, slot l&3) and again on
// the ds_read_b128, which is then conflict-free for the b128 lane groups of MI355X_MICROARCH.md §LDS.
// A 32x32x16 step takes k = 8h .. 8h+7 from half-wave h, i.e. chunk 2s+h in step s: no k permutation.

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int XBK = 32;   // k per stage (and the unit of split-K bookkeeping)

// NP = planes per operand.  2: split-bf16 (hi + lo, three bf16 products per fp32 product).  1: ONE IEEE fp16 plane
// (BASELINE configs[4]: fp16 operands, fp32 accumulate) — a single v_mfma_f32_32x32x16_f16 per sixteen k, half the
// operand bytes; the kernels are otherwise the same, X3Planes' lo pointers are unused.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NP>
__device__ __forceinline__ void x3_mma(f32x16& acc, const bf16x8& ah, const bf16x8& al, const bf16x8& bh,
                                       const bf16x8& bl) {
  if (NP == 2) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc, 0, 0, 0);
  }
}
// Bias gradient inside the weight-gradient contraction: db[co] = sum_p dY[p][co] is dY^T times a column of ones, so the
// workgroups of the FIRST column tile run one extra matrix instruction per dY fragment (two with hi + lo planes)
// against an all-ones B fragment; every column of the 32x32 result holds the row sums.  The gradient is in LDS anyway:
// no separate pass over it (round 2 start: 24 channel_sum launches, 0.8 ms per step).
template <int NP>
__device__ __forceinline__ void x3_bias_mma(f32x16& acc, const bf16x8& ah, const bf16x8& al) {
  if (NP == 2) {
    const __bf16 one = (__bf16)1.0f;
    const bf16x8 ones = {one, one, one, one, one, one, one, one};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ones, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, ones, acc, 0, 0, 0);
  } else {
    const _Float16 one = (_Float16)1.0f;
    const f16x8 ones = {one, one, one, one, one, one, one, one};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), ones, acc, 0, 0, 0);
  }
}
// rows of a 32x32 accumulator tile held by the lanes of column 0 -> out[m_base + row] (scaled by 2^-in_shift)
__device__ __forceinline__ void x3_bias_store(const Params& p, const f32x16& acc, int m_base, int split, int lane) {
  if ((lane & 31) != 0) return;
  const float a = pow2i(-p.in_shift);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int m = m_base + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (m < p.M) p.bias_slab[(size_t)split * p.M + m] = acc[r] * a;
  }
}
constexpr int x3_max(int a, int b) { return a > b ? a : b; }
// epilogue bands through an LDS window of `kbytes` (the K-loop buffers) for a BM x BN fp32 tile
constexpr int x3_passes(int kbytes, int BM, int BN, int TM) {
  return kbytes >= BM * BN * 4 ? 1 : (kbytes >= BM * BN * 2 ? 2 : (((BM / 4) % (32 * TM) == 0 && kbytes >= BM * BN) ? 4 : 2));
}

// (tile, K slice) of this workgroup in a tiles x slices grid, chosen so that the workgroups sharing an XCD (those
// with equal dispatch index % 8, MI355X_MICROARCH.md "Workgroup dispatch") hold a CONTIGUOUS run of the slice-major
// work list: the tiles of one K slice read the same operand rows, and now find them in the XCD's own L2 instead of
// eight L2s each fetching them from the fabric.  A speed choice only: any placement gives the same result.
__device__ __forceinline__ void xcd_slice_major(int ntiles, int& tile, int& split) {
  const int total = gridDim.x * gridDim.y;
  const int id = blockIdx.y * gridDim.x + blockIdx.x;
  const int xcd = id % 8, idx = id / 8;
  const int qq = total / 8, r = total % 8;
  const int w = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  split = w / ntiles;
  tile = w - split * ntiles;
}

// The same for a GROUP launch (blockIdx.z = member in dispatch order): the work list is member-major, slice-major, and
// an XCD's run is cut from the whole list — dispatch index % 8 names the XCD only when the z planes are counted in
// (a member's tiles x slices is rarely a multiple of 8, so per-member runs sat on shifting XCDs: 842 MB of fabric
// traffic per launch for 391 MB of operands on the 256-tile group kernel, round 4).
__device__ __forceinline__ void xcd_group_slice_major(int ntiles, int& member, int& tile, int& split) {
  const int per = gridDim.x * gridDim.y, total = per * gridDim.z;
  const int id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  const int xcd = id % 8, idx = id / 8;
  const int qq = total / 8, r = total % 8;
  const int w = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
#ifdef JTSM_GROUP_XCD_PER_MEMBER   // (A/B: round 3's placement — runs cut per member, XCD taken from the member-local index)
  { member = blockIdx.z; const int lid = blockIdx.y * gridDim.x + blockIdx.x, lx = lid % 8, li = lid / 8, lq = per / 8, lr = per % 8;
    const int lw = (lx < lr ? lx * (lq + 1) : lr * (lq + 1) + (lx - lr) * lq) + li; split = lw / ntiles; tile = lw - split * ntiles; return; }
#endif
  member = w / per;
  const int rem = w - member * per;
  split = rem / ntiles;
  tile = rem - split * ntiles;
}

struct X3Planes {
  const __bf16* A_hi; const __bf16* A_lo;
  const __bf16* B_hi; const __bf16* B_lo;
};

// Same-shape weight gradients in ONE launch (blockIdx.z = member): a weight gradient is not on the backward's critical
// path, and res3 / res4 / res5 repeat three layer shapes 3-6 times — launched one by one each of those small GEMMs
// (res4 1x1: M = 256, N = 1024, K = 8192 pixels) is cut into 32 pixel slices to fill the chip, and its slabs (32 x the
// result) are written and folded again; a group of six needs 5-6 slices for the same number of workgroups.
constexpr int kMaxGroup = 8;
struct X3Group {
  int n;
  const __bf16* A_hi[kMaxGroup]; const __bf16* A_lo[kMaxGroup];   // dY planes
  const __bf16* B_hi[kMaxGroup]; const __bf16* B_lo[kMaxGroup];   // X planes
  float* C[kMaxGroup];                                            // dW
  const float* scale[kMaxGroup];                                  // per-row factor (the FrozenBN scale), or null
  size_t slab_stride;                                             // floats between two members' slab sets
};

// PAIRED planes (weights).  A K stage takes 32 k = 64 bytes per row and plane.  With the planes in two arrays that is
// HALF a 128-byte line from each: the line's other half belongs to the next stage, by which time the 64 KiB that passed
// through the CU's 32 KiB L1 have evicted it — every line crosses L2 -> L1 twice, and that fill path (64 B/clk per CU),
// not the matrix pipes, bounded the forward / data-gradient kernels (measured by fetching the lo chunk from the other
// half of the hi chunk's line, wrong values: 12-25 % shorter launches).  A paired operand stores, per row, blocks of
// [32 k of hi][32 k of lo]: one stage = one whole line per row.  Element (row, k) of an operand with K % 32 == 0:
//     hi at row * 2K + (k / 32) * 64 + k % 32,   lo 32 elements behind it.
// The pair of pointers says which layout it is: lo == hi + 32 elements <=> paired (with separate planes lo is at least
// one whole plane away; a 32-element plane is the same bytes in both layouts).
__host__ __device__ __forceinline__ bool x3_paired(const void* hi, const void* lo, int np) {
  return np == 2 && lo && reinterpret_cast<const char*>(lo) - reinterpret_cast<const char*>(hi) == 64;
}
__host__ __device__ __forceinline__ size_t x3_paired_index(size_t row, size_t k, size_t K) {
  return row * 2 * K + (k >> 5) * 64 + (k & 31);
}

// Address of the zero page that out-of-range rows are fetched from, made opaque once per kernel: left to itself the
// compiler re-derives it through the GOT (s_getpc + s_load_dwordx2 + s_waitcnt lgkmcnt(0)) in front of EVERY
// direct-to-LDS load of the K loop — a scalar-memory round trip per piece that also drains the outstanding ds_reads.
__device__ __forceinline__ const void* zero_page_address() {
  const void* z = g_zero_page;
  asm volatile("" : "+s"(z));
  return z;
}

__device__ __forceinline__ void dma16b(const void* g, void* lds_uniform_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_uniform_base, 16, 0, 0);
}

// Tile shape: WM x WN wavefronts, each TM x TN MFMA tiles of 32x32 -> a workgroup of 64*WM*WN threads owns
// BM x BN = 32*WM*TM x 32*WN*TN outputs.  The loop is bound by how many operand bytes a CU can keep in
// flight from L2 (measured: the same kernel without its MFMAs takes 67-90 % of the full time, and LDS limits
// the bytes in flight), so the large layers use 256x256 (half the operand bytes per FLOP of 128x128).
#ifndef JTSM_X3_NBUF1_WGS
// Workgroups per CU the single-buffered four-wave kernels are compiled for.  3 (170 registers): the pipelined epilogue
// (store_tile_wide) keeps its operands in registers; at 4 (128 registers) it spills into scratch and these
// output-bound layers lose more than the fourth workgroup gives (measured, tools/sweeps/ab_libs.sh: 64 -> 256 channels
// at 256 x 256: 79 us at 3, 137 us at 4 with spills, 121 us with the unpipelined epilogue at 4).
#define JTSM_X3_NBUF1_WGS 3
#endif
template <int ROLE, int WM, int WN, int TM, int TN, int NBUF, int NP = 2>
__global__ __launch_bounds__(64 * WM * WN, (NBUF == 1 && WM * WN == 4) ? JTSM_X3_NBUF1_WGS : (WM * WN == 4 ? 2 : 2))
void igemm_x3_kernel(const Params p, const X3Planes q) {
  static_assert(ROLE == FWD || ROLE == DGRAD, "bf16x3: forward and data-gradient roles");
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  constexpr int A_PL = BM * 64, B_PL = BN * 64;          // bytes per plane per stage (64-byte rows)
  constexpr int STAGE = NP * (A_PL + B_PL);
  static_assert(BM % (16 * NW) == 0 && BN % (16 * NW) == 0, "16-row loads must divide evenly among the wavefronts");
  constexpr int A_INS = BM / 16 / NW, B_INS = BN / 16 / NW;   // loads per wavefront per plane per stage
  constexpr int PASSES = x3_passes(NBUF * STAGE, BM, BN, TM);
  constexpr int LDSB = x3_max(NBUF * STAGE, BM * BN * 4 / PASSES);
  __shared__ __attribute__((aligned(16))) char lds[LDSB];

  const ConvShape& s = p.s;
  const void* const zero_page = zero_page_address();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, li = lane & 31;

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntn * ntm;
  int tile = blockIdx.x;
  {
    const int qq = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  int kbeg = 0, kend = p.K;
  if (gridDim.y > 1) {
    kbeg = blockIdx.y * p.ktiles_per_split * XBK;
    kend = min(p.K, kbeg + p.ktiles_per_split * XBK);
  }
  const int ntile_k = kbeg < kend ? (kend - kbeg + XBK - 1) / XBK : 0;

  // ---- addressing state: the tap and first channel of a stage are wave-uniform and advance incrementally.
  // K ORDER of a k x k layer: channel block outer, taps inner (any order sums to the same product).  With the taps
  // outermost a workgroup came back to its activation rows only after a whole pass over the channels — 256 KiB per
  // workgroup, ~8 MB per XCD, twice its L2: measured 641 MB of fabric traffic per launch for 134 MB of operands on
  // the mask heads' 3x3 layers.  Now the nine taps of one 32-channel block follow each other: 32 KiB per workgroup.
  // (Eligibility guarantees whole stages per tap when there is more than one tap.)
  const int Cdim = ROLE == FWD ? s.Cin : s.Cout;
  const int taps = s.KH * s.KW;
  const bool perm = taps > 1;
  const int klim = perm ? p.K : kend;       // permuted stages are always whole: only the slice's stage range bounds them
  int t_kh, t_kw, t_c;
  if (perm) {
    const int sidx = kbeg / XBK, cb = sidx / taps, tap = sidx - cb * taps;
    t_c = cb * XBK;
    t_kh = tap / s.KW;
    t_kw = tap - t_kh * s.KW;
  } else {
    t_c = kbeg;
    t_kh = t_kw = 0;
  }
  // Per-row state: element offset of (row, tap (0,0), this lane's chunk) and one validity bit per kh and per
  // kw (zero padding, rows past M) — so a stage costs one add, two shifts and a select per load instead of
  // re-deriving ih / iw.  A strided data gradient (rare: stride inside a k > 1 kernel) keeps the general path.
  const bool linear = ROLE == FWD || s.stride == 1;
  PixelRow arow[A_INS];
  int a_chunk[A_INS], a_base[A_INS];
  unsigned a_mh[A_INS], a_mw[A_INS];
#pragma unroll
  for (int j = 0; j < A_INS; ++j) {
    const int r = (wave * A_INS + j) * 16 + (lane >> 2);
    arow[j] = ROLE == FWD ? fwd_pixel(s, m0 + r, p.M) : dgrad_pixel(s, m0 + r, p.M);
    a_chunk[j] = 8 * ((lane & 3) ^ ((r >> 2) & 3));
    unsigned mh = 0, mw = 0;
    if (ROLE == FWD) {
      for (int kh = 0; kh < s.KH; ++kh) mh |= ((unsigned)(arow[j].h0 + kh * s.dil) < (unsigned)s.H ? 1u : 0u) << kh;
      for (int kw = 0; kw < s.KW; ++kw) mw |= ((unsigned)(arow[j].w0 + kw * s.dil) < (unsigned)s.W ? 1u : 0u) << kw;
      a_base[j] = ((arow[j].b * s.H + arow[j].h0) * s.W + arow[j].w0) * s.Cin + a_chunk[j];
    } else {
      for (int kh = 0; kh < s.KH; ++kh) mh |= ((unsigned)(arow[j].h0 - kh * s.dil) < (unsigned)s.Ho ? 1u : 0u) << kh;
      for (int kw = 0; kw < s.KW; ++kw) mw |= ((unsigned)(arow[j].w0 - kw * s.dil) < (unsigned)s.Wo ? 1u : 0u) << kw;
      a_base[j] = ((arow[j].b * s.Ho + arow[j].h0) * s.Wo + arow[j].w0) * s.Cout + a_chunk[j];
    }
    a_mh[j] = arow[j].ok ? mh : 0u;
    a_mw[j] = mw;
  }
  const int b_mul = x3_paired(q.B_hi, q.B_lo, NP) ? 2 : 1;
  int b_off[B_INS], b_chunk[B_INS];
#pragma unroll
  for (int j = 0; j < B_INS; ++j) {
    const int r = (wave * B_INS + j) * 16 + (lane >> 2);
    b_off[j] = (n0 + r) < p.N ? (n0 + r) * p.K * b_mul : -1;
    b_chunk[j] = 8 * ((lane & 3) ^ ((r >> 2) & 3));
  }
#ifdef JTSM_TIMING_PAIRED_A   // TIMING-ONLY build (wrong values): the activations' lo chunk comes from the other half of the
                              // hi chunk's 128-byte line — the request pattern a paired [hi 64 B | lo 64 B] activation
                              // layout would have (tools/sweeps/x3_paired_pmc.sh -> profiles/r04_x3_paired_activations.json)
  const long lo_delta_a = 32, lo_delta_b = q.B_lo - q.B_hi;
#else
  const long lo_delta_a = q.A_lo - q.A_hi, lo_delta_b = q.B_lo - q.B_hi;   // the lo plane sits at a fixed distance
#endif

  // One direct-to-LDS load ("piece") of the next stage: pieces 0 .. 2*A_INS-1 are the A rows (hi, lo
  // alternating), the rest the B rows.  The K loop issues them one at a time BETWEEN its MFMA groups.
  constexpr int PIECES = NP * (A_INS + B_INS);
  auto issue_piece = [&](int idx, int buf) {
    char* St = lds + buf * STAGE;
    const int k0 = (t_kh * s.KW + t_kw) * Cdim + t_c;   // first k of the stage the state points at (wave-uniform)
    if (idx < NP * A_INS) {
      const int j = idx / NP, lo = idx % NP;
      bool ok;
      int off;
      if (linear) {
        const int tap_delta = ROLE == FWD ? ((t_kh * s.dil) * s.W + t_kw * s.dil) * s.Cin + t_c
                                          : -((t_kh * s.dil) * s.Wo + t_kw * s.dil) * s.Cout + t_c;
        ok = ((a_mh[j] >> t_kh) & (a_mw[j] >> t_kw) & 1u) != 0u && (k0 + a_chunk[j]) < klim;
        off = a_base[j] + tap_delta;
      } else {
        const PixelRow& r = arow[j];
        const int th = r.h0 - t_kh * s.dil, tw = r.w0 - t_kw * s.dil;
        const int oh = th / s.stride, ow = tw / s.stride;
        ok = r.ok && (k0 + a_chunk[j]) < klim && th >= 0 && tw >= 0 && oh * s.stride == th && ow * s.stride == tw &&
             oh < s.Ho && ow < s.Wo;
        off = ((r.b * s.Ho + oh) * s.Wo + ow) * s.Cout + t_c + a_chunk[j];
      }
      const __bf16* src = q.A_hi + off + (lo ? lo_delta_a : 0);
      dma16b(ok ? (const void*)src : zero_page, St + lo * A_PL + (wave * A_INS + j) * 1024);
    } else {
      const int j = (idx - NP * A_INS) / NP, lo = (idx - NP * A_INS) % NP;
      const bool ok = b_off[j] >= 0 && (k0 + b_chunk[j]) < klim;
      const __bf16* src = q.B_hi + (b_off[j] + k0 * b_mul + b_chunk[j]) + (lo ? lo_delta_b : 0);
      dma16b(ok ? (const void*)src : zero_page, St + NP * A_PL + lo * B_PL + (wave * B_INS + j) * 1024);
    }
  };
  auto advance_tap = [&]() {   // next stage: the next tap of this channel block, then the next block
    if (perm) {
      if (++t_kw == s.KW) {
        t_kw = 0;
        if (++t_kh == s.KH) { t_kh = 0; t_c += XBK; }
      }
    } else {
      t_c += XBK;
    }
  };
  auto issue = [&](int buf) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) issue_piece(i, buf);
    advance_tap();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_off[TM], a_swz[TM], bb_off[TN], b_swz[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int ar = (wm * TM + t) * 32 + li;
    a_off[t] = ar * 64;
    a_swz[t] = (ar >> 2) & 3;
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int br = (wn * TN + t) * 32 + li;
    bb_off[t] = NP * A_PL + br * 64;
    b_swz[t] = (br >> 2) & 3;
  }

  constexpr int SLOTS = 2 * TM * TN;                        // MFMA groups per stage (2 k-steps x TM x TN tiles)
  constexpr int PER_SLOT = (PIECES + SLOTS - 1) / SLOTS;    // loads placed behind each group
  if (NBUF == 2 && ntile_k > 0) issue(0);
  if (NBUF > 2) {
    // Ring of NBUF stages (the 64 x 64 tiles of the long-K 1x1 layers): a stage's six MFMA per wavefront (~0.1 us)
    // cannot hide a load's latency (~1 us) with ONE stage in flight, so NBUF - 1 stages are kept in flight and a
    // stage's loads are all issued right behind the barrier that frees its buffer.
#pragma unroll
    for (int i = 0; i < NBUF - 1; ++i)
      if (i < ntile_k) issue(i);
  }
  for (int t = 0; t < ntile_k; ++t) {
    if (NBUF == 1) {
      __syncthreads();
      issue(0);
    }
    if (NBUF > 2) {
      // loads return in order: stage t has landed once at most the loads of the `ahead` later stages are outstanding
      static_assert(NBUF <= 4 && 2 * PIECES <= 63, "ring: the vmcnt immediates below cover NBUF <= 4");
      const int ahead = min(NBUF - 2, ntile_k - 1 - t);   // wave-uniform
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PIECES) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // A bare barrier: __syncthreads() carries a workgroup-scope fence, for which the compiler waits on EVERY
      // outstanding load (s_waitcnt vmcnt(0)) — the ring's later stages included.  What the barrier must order here
      // is covered without it: each wavefront has waited for its own stage-t loads (above) before it arrives, and
      // its LDS reads of iteration t - 1 were consumed by that iteration's MFMAs.
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // (buffer (t - 1) % NBUF was read during iteration t - 1: every wavefront is past those reads at this barrier)
    if (NBUF > 2 && t + NBUF - 1 < ntile_k) issue((t + NBUF - 1) % NBUF);
    const bool more = NBUF == 2 && t + 1 < ntile_k;   // wave-uniform
    const int bnext = (t + 1) & 1;
    const char* St = lds + (NBUF == 2 ? (t & 1) : (NBUF > 2 ? t % NBUF : 0)) * STAGE;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int c = 2 * st + half;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int u = 0; u < TM; ++u) {
        const int ao = a_off[u] + ((c ^ a_swz[u]) << 4);
        ah[u] = *reinterpret_cast<const bf16x8*>(St + ao);
        al[u] = NP == 2 ? *reinterpret_cast<const bf16x8*>(St + A_PL + ao) : ah[u];
      }
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int bo = bb_off[u] + ((c ^ b_swz[u]) << 4);
        bh[u] = *reinterpret_cast<const bf16x8*>(St + bo);
        bl[u] = NP == 2 ? *reinterpret_cast<const bf16x8*>(St + B_PL + bo) : bh[u];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          x3_mma<NP>(acc[i][j], ah[i], al[i], bh[j], bl[j]);
          if (NBUF == 2) {
            const int slot = (st * TM + i) * TN + j;
            if (more) {
#pragma unroll
              for (int e = 0; e < PER_SLOT; ++e)
                if (slot * PER_SLOT + e < PIECES) issue_piece(slot * PER_SLOT + e, bnext);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the load behind this MFMA group
          }
        }
    }
    if (more) advance_tap();
  }
  static_assert(LDSB >= BM * BN * 4 / PASSES && (BM / PASSES) % (32 * TM) == 0, "epilogue band does not fit / split a wave tile");
  if (p.wide)
    store_tile_wide<ROLE, BM, BN, PASSES, TM, TN, NT, LinearRows, NP == 1>(p, acc, m0, n0, wm, wn, lane, tid,
                                                                            reinterpret_cast<float*>(lds));
  else
    store_tile<ROLE, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// ---- k x k convolutions with an LDS-resident input halo -----------------------------------------------
// The generic kernel above fetches the A operand once per filter tap: a 3x3 layer moves its activations nine
// times from L2 to LDS.  Here the M tile is a TH x TW patch of output pixels of one image; for each block of 32
// channels the (TH + (KH-1)d) x (TW + (KW-1)d) input halo is loaded ONCE and the taps read it at shifted row
// offsets, so only the weights are streamed per tap: ~42 % fewer operand bytes for 3x3 at 128 x 128.
// Stride 1 only.  FWD: source X, tap (kh,kw) reads halo row + kh*d.  DGRAD: source dY, taps mirrored.
// K order: channel block outer, taps inner (any order sums to the same product; split-K cuts channel blocks).
// Two shapes: 8 x 16 pixels x 128 channels (4 wavefronts, 80 KiB, two workgroups per CU) and 16 x 16 pixels x 256
// channels (8 wavefronts, 148 KiB) for the layers that fill the chip with 256-wide tiles.
template <int ROLE, int TH, int WM, int WN, int TN, int HP16 /* 16-row groups of the halo image */, int NP = 2>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_x3_halo_kernel(const Params p, const X3Planes q) {
  static_assert(ROLE == FWD || ROLE == DGRAD, "halo kernel: forward and data-gradient roles");
  constexpr int TW = 16, TM = 2, NW = WM * WN, NT = 64 * NW;
  constexpr int BM = TH * TW, BN = 32 * WN * TN;
  static_assert(BM == 32 * WM * TM, "patch rows = wavefront rows");
  constexpr int A_PL = HP16 * 1024;          // bytes per plane of one halo buffer
  constexpr int B_PL = BN * 64;              // bytes per plane of one weight stage
  constexpr int A_BUF = NP * A_PL, B_BUF = NP * B_PL;
  constexpr int A_J = (HP16 + NW - 1) / NW;  // halo loads per wavefront per plane
  constexpr int B_J = BN / 16 / NW;          // weight loads per wavefront per plane
  constexpr int PASSES = x3_passes(2 * A_BUF + 2 * B_BUF, BM, BN, TM);
  constexpr int LDSB = x3_max(2 * A_BUF + 2 * B_BUF, BM * BN * 4 / PASSES);
  __shared__ __attribute__((aligned(16))) char lds[LDSB];
  char* const Abase = lds;
  char* const Bbase = lds + 2 * A_BUF;

  const ConvShape& s = p.s;
  const void* const zero_page = zero_page_address();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, li = lane & 31;
  const int d = s.dil, taps = s.KH * s.KW;
  const int HPW = TW + (s.KW - 1) * d, HP = (TH + (s.KH - 1) * d) * HPW;
  // source / output geometry of this role
  const int SH = ROLE == FWD ? s.H : s.Ho, SW = ROLE == FWD ? s.W : s.Wo;     // source image
  const int OH = ROLE == FWD ? s.Ho : s.H, OW = ROLE == FWD ? s.Wo : s.W;     // output image
  const int Cdim = ROLE == FWD ? s.Cin : s.Cout;

  const int ntn = (p.N + BN - 1) / BN;
  const int tH = (OH + TH - 1) / TH, tW = (OW + TW - 1) / TW;
  const int ntiles = ntn * s.Bn * tH * tW;
  int tile = blockIdx.x;
  {
    const int qq = ntiles / 8, r = ntiles % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx;
  }
  const int n0 = (tile % ntn) * BN;
  int mt = tile / ntn;
  const int tx = mt % tW; mt /= tW;
  const int ty = mt % tH;
  const int b = mt / tH;
  const int oy0 = ty * TH, ox0 = tx * TW;
  // halo origin in the source image
  const int hy0 = ROLE == FWD ? oy0 - s.pad : oy0 + s.pad - (s.KH - 1) * d;
  const int hx0 = ROLE == FWD ? ox0 - s.pad : ox0 + s.pad - (s.KW - 1) * d;

  const int Cb = Cdim / XBK;
  int cb0 = 0, cb1 = Cb;
  if (gridDim.y > 1) {
    cb0 = blockIdx.y * p.ktiles_per_split;
    cb1 = min(Cb, cb0 + p.ktiles_per_split);
  }
  const int nstage = cb0 < cb1 ? (cb1 - cb0) * taps : 0;

  // ---- halo loads: load T = wave + 4j covers halo rows 16T .. 16T+15; this lane owns row 16T + lane/4
  int a_base[A_J];
  bool a_ok[A_J];
#pragma unroll
  for (int j = 0; j < A_J; ++j) {
    const int T = wave + NW * j;
    const int hp = 16 * T + (lane >> 2);
    const int hy = hp / HPW, hx = hp - hy * HPW;
    const int y = hy0 + hy, x = hx0 + hx;
    a_ok[j] = T < HP16 && hp < HP && (unsigned)y < (unsigned)SH && (unsigned)x < (unsigned)SW;
    a_base[j] = ((b * SH + y) * SW + x) * Cdim + 8 * ((lane & 3) ^ ((hp >> 2) & 3));
  }
  const int b_mul = x3_paired(q.B_hi, q.B_lo, NP) ? 2 : 1;
  int b_off[B_J], b_chunk[B_J];
#pragma unroll
  for (int j = 0; j < B_J; ++j) {
    const int r = (wave * B_J + j) * 16 + (lane >> 2);
    b_off[j] = (n0 + r) < p.N ? (n0 + r) * p.K * b_mul : -1;
    b_chunk[j] = 8 * ((lane & 3) ^ ((r >> 2) & 3));
  }
  const long lo_delta_a = q.A_lo - q.A_hi, lo_delta_b = q.B_lo - q.B_hi;

  constexpr int A_PIECES = NP * A_J, B_PIECES = NP * B_J;
  auto issue_a_piece = [&](int idx, int cb, int buf) {
    const int j = idx / NP, lo = idx % NP;
    const int T = wave + NW * j;
    if (T >= HP16) return;   // wave-uniform
    const __bf16* src = q.A_hi + (a_base[j] + cb * XBK) + (lo ? lo_delta_a : 0);
    dma16b(a_ok[j] ? (const void*)src : zero_page, Abase + buf * A_BUF + lo * A_PL + T * 1024);
  };
  auto issue_b_piece = [&](int idx, int cb, int tap, int buf) {
    const int j = idx / NP, lo = idx % NP;
    const __bf16* src = q.B_hi + (b_off[j] + (tap * Cdim + cb * XBK) * b_mul + b_chunk[j]) + (lo ? lo_delta_b : 0);
    dma16b(b_off[j] >= 0 ? (const void*)src : zero_page,
           Bbase + buf * B_BUF + lo * B_PL + (wave * B_J + j) * 1024);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment rows: A row = local output pixel -> halo pixel of tap (0,0); B as in the generic kernel
  int hp0[TM], bb_off[TN], b_swz[TN];
#pragma unroll
  for (int t = 0; t < TM; ++t) {
    const int ml = (wm * TM + t) * 32 + li;
    hp0[t] = (ml / TW) * HPW + (ml % TW);
  }
#pragma unroll
  for (int t = 0; t < TN; ++t) {
    const int br = (wn * TN + t) * 32 + li;
    bb_off[t] = br * 64;
    b_swz[t] = (br >> 2) & 3;
  }

  if (nstage > 0) {
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) issue_a_piece(i, cb0, 0);
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) issue_b_piece(i, cb0, 0, 0);
  }
  int cb = cb0, tap = 0, kh = 0, kw = 0;
  for (int st = 0; st < nstage; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // the stage after this one
    int ntap = tap + 1, ncb = cb;
    if (ntap == taps) { ntap = 0; ++ncb; }
    const bool more = st + 1 < nstage;          // wave-uniform
    const bool new_halo = more && ntap == 0;
    const int abuf = (cb - cb0) & 1, bbuf = st & 1;
    const char* Ab = Abase + abuf * A_BUF;
    const char* Bb = Bbase + bbuf * B_BUF;
    const int dy = (ROLE == FWD ? kh : s.KH - 1 - kh) * d, dx = (ROLE == FWD ? kw : s.KW - 1 - kw) * d;
    int a_row[TM], a_swz[TM];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int hp = hp0[t] + dy * HPW + dx;
      a_row[t] = hp * 64;
      a_swz[t] = (hp >> 2) & 3;
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int c = 2 * ks + half;
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int u = 0; u < TM; ++u) {
        const int ao = a_row[u] + ((c ^ a_swz[u]) << 4);
        ah[u] = *reinterpret_cast<const bf16x8*>(Ab + ao);
        al[u] = NP == 2 ? *reinterpret_cast<const bf16x8*>(Ab + A_PL + ao) : ah[u];
      }
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        const int bo = bb_off[u] + ((c ^ b_swz[u]) << 4);
        bh[u] = *reinterpret_cast<const bf16x8*>(Bb + bo);
        bl[u] = NP == 2 ? *reinterpret_cast<const bf16x8*>(Bb + B_PL + bo) : bh[u];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          x3_mma<NP>(acc[i][j], ah[i], al[i], bh[j], bl[j]);
          // MFMA groups 0 .. B_PIECES-1 are each followed by one weight load of the next stage, the remaining
          // groups by the next channel block's halo loads (when the next stage starts one)
          constexpr int SLOTS = 2 * TM * TN, A_SLOTS = SLOTS - B_PIECES, A_PER = (A_PIECES + A_SLOTS - 1) / A_SLOTS;
          const int slot = (ks * TM + i) * TN + j;
          if (more) {
            if (slot < B_PIECES) issue_b_piece(slot, ncb, ntap, bbuf ^ 1);
            else if (new_halo) {
#pragma unroll
              for (int e = 0; e < A_PER; ++e) {
                const int idx = (slot - B_PIECES) * A_PER + e;
                if (idx < A_PIECES) issue_a_piece(idx, ncb, abuf ^ 1);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    tap = ntap; cb = ncb;
    if (++kw == s.KW) { kw = 0; if (++kh == s.KH) kh = 0; }
  }
  const PatchRows rows = {b, oy0, ox0, OH, OW, TW};
  static_assert(LDSB >= BM * BN * 4 / PASSES && (BM / PASSES) % (32 * TM) == 0, "epilogue band does not fit");
  if (p.wide)
    store_tile_wide<ROLE, BM, BN, PASSES, TM, TN, NT, PatchRows>(p, acc, 0, n0, wm, wn, lane, tid,
                                                                 reinterpret_cast<float*>(lds), &rows);
  else {
    // (narrow fallback: scalar stores through the same row map)
    const Epilogue& e = p.e;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
      if (n >= p.N) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = rows((wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
          if (m < 0) continue;
          float v = acc[i][j][r];
          if (gridDim.y > 1) { p.slab[((size_t)blockIdx.y * p.M + m) * p.ldc + n] = v; continue; }
          const size_t o = (size_t)m * p.ldc + n;
          v = v * pow2i(-p.in_shift) * (e.scale ? e.scale[n] : 1.f) + (e.bias ? e.bias[n] : 0.f);
          if (e.residual) v += e.residual[o];
          if (e.relu) v = fmaxf(v, 0.f);
          if (e.mask) v = e.mask[o] > 0.f ? v : 0.f;
          p.C[o] = v;
        }
    }
  }
}

// ---- weight gradient ------------------------------------------------------------------------------
// dW[co][tap][ci] += sum_p dY[p][co] * X[pix(p,tap)][ci]: K runs over pixels, and both operands are
// CHANNEL-contiguous in memory, i.e. k-major.  A stage holds 32 pixels x 128 channels per plane as plain
// 256-byte rows; 16-byte chunk ch of row r sits at slot ch ^ (((r&3)<<2) | ((r>>2)&3)) (source-side
// swizzle again), and the MFMA operands (8 consecutive k of one channel per lane) come out of two
// ds_read_b64_tr_b16 each — the hardware transpose, conflict-free on this image (bank rule of
// cdna_hip_programming.md §2: a 32-lane half touches 16 distinct chunks x 2 halves of a chunk).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 lds_tr8(const char* lo_rows, const char* hi_rows) {
  const bf16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)lo_rows);
  const bf16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)hi_rows);
  return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// WM x WN wavefronts of TM x TN MFMA tiles, as in igemm_x3_kernel; operands wider than 128 channels are kept
// as several 128-channel images (each with the swizzle above).
template <int WM, int WN, int TM, int TN, int NBUF, int NP = 2, bool BIAS = false>
__device__ __forceinline__ void x3_wgrad_body(const Params& p, const X3Planes& q, int tile_given = -1, int split_given = 0) {
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
  static_assert(BM % 128 == 0 && BN % 128 == 0 && 8 % NW == 0, "128-channel images; 8 row groups shared by the wavefronts");
  constexpr int ASUB = BM / 128, BSUB = BN / 128;
  constexpr int PL = XBK * 256;                 // bytes per image per plane per stage: 32 pixel rows x 128 channels
  constexpr int STAGE = NP * (ASUB + BSUB) * PL; // [dY image 0 hi, lo, image 1 hi, lo ...][X images likewise]
  constexpr int INS = 8 / NW;                   // 4-row load groups per wavefront per image
  constexpr int PASSES = x3_passes(NBUF * STAGE, BM, BN, TM);
  constexpr int LDSB = x3_max(NBUF * STAGE, BM * BN * 4 / PASSES);
  __shared__ __attribute__((aligned(16))) char lds[LDSB];

  const ConvShape& s = p.s;
  const void* const zero_page = zero_page_address();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;

  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntn * ntm;
  // all tiles of a pixel slice on one XCD: they read the same dY / X rows (the mask heads' 3x3 layers: 9 tiles x 28
  // slices measured 679 MB of fabric traffic per launch for 152 MB of operands with the slices dealt over the XCDs)
  int tile = tile_given, split = split_given;
  if (tile_given < 0) xcd_slice_major(ntiles, tile, split);
  const int m0 = (tile / ntn) * BM, n0 = (tile % ntn) * BN;

  int kbeg = 0, kend = p.K;
  if (gridDim.y > 1) {
    kbeg = split * p.ktiles_per_split * XBK;
    kend = min(p.K, kbeg + p.ktiles_per_split * XBK);
    if (kbeg >= kend && !p.wide) return;   // (slab path: an empty slice still writes its zero slab)
  }
  const int ntile_k = kbeg < kend ? (kend - kbeg + XBK - 1) / XBK : 0;

  // ---- load addressing: row group T = INS*wave + j fills rows 4T .. 4T+3 of every image; this lane owns row
  // r = 4T + lane/16 and the chunk whose slot is lane%16
  int row_k[INS], a_col[INS][ASUB], b_ci[INS][BSUB], b_kh[INS][BSUB], b_kw[INS][BSUB];
  bool a_ok[INS][ASUB], b_ok[INS][BSUB];
  PixState bpix[INS];
#pragma unroll
  for (int j = 0; j < INS; ++j) {
    const int r = 4 * (wave * INS + j) + (lane >> 4);
    const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
    row_k[j] = r;
#pragma unroll
    for (int u = 0; u < ASUB; ++u) {
      a_col[j][u] = m0 + u * 128 + 8 * ch;
      a_ok[j][u] = a_col[j][u] < p.M;
    }
#pragma unroll
    for (int u = 0; u < BSUB; ++u) {
      const int col = n0 + u * 128 + 8 * ch;
      b_ok[j][u] = col < p.N;
      const int cc = b_ok[j][u] ? col : 0;
      const int tap = cc / s.Cin;
      b_ci[j][u] = cc - tap * s.Cin;
      b_kh[j][u] = tap / s.KW;
      b_kw[j][u] = tap - b_kh[j][u] * s.KW;
    }
    bpix[j] = pix_init(kbeg + r, s.Ho, s.Wo);
  }
  const long lo_delta_a = q.A_lo - q.A_hi, lo_delta_b = q.B_lo - q.B_hi;

  // piece idx -> (row group j, operand, image u, plane): INS * 2 * (ASUB + BSUB) pieces per wavefront per stage
  constexpr int PPJ = NP * (ASUB + BSUB);
  constexpr int PIECES = INS * PPJ;
  auto issue_piece = [&](int idx, int k0, int buf) {
    char* St = lds + buf * STAGE;
    const int j = idx / PPJ, w = idx % PPJ;
    const int dst = (wave * INS + j) * 1024;
    if (w < NP * ASUB) {
      const int u = w / NP, lo = w % NP;
      const int k = k0 + row_k[j];
      const bool ok = a_ok[j][u] && k < kend;
      const __bf16* src = q.A_hi + ((long)k * s.Cout + a_col[j][u]) + (lo ? lo_delta_a : 0);
      dma16b(ok ? (const void*)src : zero_page, St + (u * NP + lo) * PL + dst);
    } else {
      const int u = (w - NP * ASUB) / NP, lo = (w - NP * ASUB) % NP;
      const PixState& px = bpix[j];
      const int ih = px.oh * s.stride - s.pad + b_kh[j][u] * s.dil, iw = px.ow * s.stride - s.pad + b_kw[j][u] * s.dil;
      const bool ok = b_ok[j][u] && px.k < kend && (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
      const __bf16* src = q.B_hi + (((long)(px.b * s.H + ih) * s.W + iw) * s.Cin + b_ci[j][u]) + (lo ? lo_delta_b : 0);
      dma16b(ok ? (const void*)src : zero_page, St + (NP * ASUB + u * NP + lo) * PL + dst);
    }
  };
  auto advance_pix = [&]() {
#pragma unroll
    for (int j = 0; j < INS; ++j) pix_advance(bpix[j], XBK, s.Ho, s.Wo);
  };
  auto issue = [&](int k0, int buf) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) issue_piece(i, k0, buf);
    advance_pix();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // (BIAS instantiations only: layers without a bias — every FrozenBN convolution of the backbone — keep the kernel
  // without the extra accumulators)
  const bool do_bias = BIAS && p.bias_slab != nullptr && n0 == 0 && wn == 0;   // wave-uniform
  f32x16 bacc[BIAS ? TM : 1];
#pragma unroll
  for (int i = 0; i < (BIAS ? TM : 1); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) bacc[i][r] = 0.f;

  // ---- transposed fragment reads: lane = 32h + 16g + 4ql + pl supplies, for read r2 of step st, the address
  // of row 16st + 8h + 4r2 + ql, channels 32*tile + 16g + 4pl .. +3; it receives channel 16g + lane%16.
  const int h = lane >> 5, g = (lane >> 4) & 1, ql = (lane >> 2) & 3, pl = lane & 3;
  int a_rd[TM][2], b_rd[TN][2];   // [tile][r2]: byte offset inside a stage (hi plane) for step 0
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int row = 8 * h + 4 * r2 + ql;
    const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int cb = (wm * TM + t) * 32;          // first channel of this tile inside the BM tile
      const int ch = (cb % 128) / 8 + 2 * g + (pl >> 1);
      a_rd[t][r2] = (cb / 128) * NP * PL + 256 * row + 16 * (ch ^ swz) + 8 * (pl & 1);
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) {
      const int cb = (wn * TN + t) * 32;
      const int ch = (cb % 128) / 8 + 2 * g + (pl >> 1);
      b_rd[t][r2] = NP * (ASUB + cb / 128) * PL + 256 * row + 16 * (ch ^ swz) + 8 * (pl & 1);
    }
  }

  constexpr int SLOTS = 2 * TM * TN;
  constexpr int PER_SLOT = (PIECES + SLOTS - 1) / SLOTS;
  if (NBUF == 2 && ntile_k > 0) issue(kbeg, 0);
  for (int t = 0; t < ntile_k; ++t) {
    if (NBUF == 1) {
      __syncthreads();
      issue(kbeg + t * XBK, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = NBUF == 2 && t + 1 < ntile_k;
    const int knext = kbeg + (t + 1) * XBK, bnext = (t + 1) & 1;
    const char* St = lds + (NBUF == 2 ? (t & 1) : 0) * STAGE;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const char* Ss = St + st * 16 * 256;   // rows 16st ..; the swizzle does not depend on st
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int u = 0; u < TM; ++u) {
        ah[u] = lds_tr8(Ss + a_rd[u][0], Ss + a_rd[u][1]);
        al[u] = NP == 2 ? lds_tr8(Ss + PL + a_rd[u][0], Ss + PL + a_rd[u][1]) : ah[u];
      }
#pragma unroll
      for (int u = 0; u < TN; ++u) {
        bh[u] = lds_tr8(Ss + b_rd[u][0], Ss + b_rd[u][1]);
        bl[u] = NP == 2 ? lds_tr8(Ss + PL + b_rd[u][0], Ss + PL + b_rd[u][1]) : bh[u];
      }
      if (BIAS && do_bias) {
#pragma unroll
        for (int u = 0; u < (BIAS ? TM : 1); ++u) x3_bias_mma<NP>(bacc[u], ah[u], al[u]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          x3_mma<NP>(acc[i][j], ah[i], al[i], bh[j], bl[j]);
          if (NBUF == 2) {
            const int slot = (st * TM + i) * TN + j;
            if (more) {
#pragma unroll
              for (int e = 0; e < PER_SLOT; ++e)
                if (slot * PER_SLOT + e < PIECES) issue_piece(slot * PER_SLOT + e, knext, bnext);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    }
    if (more) advance_pix();
  }
  static_assert(LDSB >= BM * BN * 4 / PASSES && (BM / PASSES) % (32 * TM) == 0, "epilogue band does not fit / split a wave tile");
  if (BIAS && do_bias) {
#pragma unroll
    for (int u = 0; u < (BIAS ? TM : 1); ++u)
      x3_bias_store(p, bacc[u], m0 + (wm * TM + u) * 32, gridDim.y > 1 ? split : 0, lane);
  }
  if (p.wide)
    store_tile_wide<WGRAD, BM, BN, PASSES, TM, TN, NT>(p, acc, m0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(lds),
                                                       (const LinearRows*)nullptr, split, tile);
  else
    store_tile<WGRAD, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

template <int WM, int WN, int TM, int TN, int NBUF, int NP = 2, bool BIAS = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_x3_wgrad_kernel(const Params p, const X3Planes q) {
  x3_wgrad_body<WM, WN, TM, TN, NBUF, NP, BIAS>(p, q);
}

// The member's operands, result, row factors and slab set in place of the launch's (everything else — the shape, the
// K slicing — is shared by the group).
__device__ __forceinline__ void x3_group_member(const Params& p_in, const X3Group& G, Params& p, X3Planes& q, int z) {
  p = p_in;
  q.A_hi = G.A_hi[z]; q.A_lo = G.A_lo[z]; q.B_hi = G.B_hi[z]; q.B_lo = G.B_lo[z];
  p.C = G.C[z];
  p.e.scale = G.scale[z];
  if (p.slab) p.slab += (size_t)z * G.slab_stride;
}

template <int WM, int WN, int TM, int TN, int NBUF, int NP = 2>
__global__ __launch_bounds__(64 * WM * WN, 2) void igemm_x3_wgrad_group_kernel(const Params p_in, const X3Group G) {
  Params p;
  X3Planes q;
  int z, tile, split;
  xcd_group_slice_major((int)gridDim.x, z, tile, split);
  x3_group_member(p_in, G, p, q, z);
  x3_wgrad_body<WM, WN, TM, TN, NBUF, NP, false>(p, q, tile, split);
}

// ---- 3x3 weight gradient with an LDS-resident input halo --------------------------------------------------
// dW[co][kh][kw][ci] = sum_p dY[p][co] * X[pix(p)+(kh,kw)][ci].  The generic kernel gives every (tap, ci) column
// group its own workgroup, so X is fetched nine times.  Here a workgroup owns 128 output channels x ONE block of 32
// input channels x ALL nine taps (nine 32x32 accumulator tiles per wavefront): a K stage is a segment of 32
// consecutive pixels of one image row, for which dY (32 px x 128 co) and the 3 x 34-pixel X halo (32 ci) are
// loaded once and the taps read the halo at shifted pixel rows — 7.5 loads per wavefront per 54 MFMAs instead of
// 8 per 24.  Needs stride 1, no dilation, in_c % 32 == 0 and an output width that is a multiple of 32 (segments
// never wrap); everything else keeps the generic kernel.
// Halo image: [102 pixel rows][64 B] per plane, UNswizzled — a transposed read of 4 consecutive pixel rows x 32
// channels is one contiguous 256-byte run (all 64 banks) wherever it starts.
template <int NP = 2, bool BIAS = false>
__device__ __forceinline__ void x3_wgrad_halo_body(const Params& p, const X3Planes& q, int tile_given = -1, int split_given = 0) {
  constexpr int SEG = 32, HPW = SEG + 2, HP = 3 * HPW;          // 102 halo pixels
  constexpr int A_PL = SEG * 256;                                 // dY stage plane: 32 px x 128 co
  constexpr int B_G = (HP + 15) / 16, B_PL = B_G * 1024;          // halo plane: 7 groups of 16 pixel rows
  constexpr int STAGE = NP * (A_PL + B_PL);                       // 30 KiB (two planes)
  constexpr int TAPS = 9;
  __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

  const ConvShape& s = p.s;
  const void* const zero_page = zero_page_address();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncb = s.Cin / 32;
  // all tiles of a pixel slice on one XCD (as the generic weight gradient): they read the same dY / X rows.  Dealt out
  // in dispatch order, the 16 tiles of a slice sat on all eight XCDs and every L2 fetched every slice — rocprofv3
  // counted 464 MB of fabric traffic per launch for 65 MB of operands (7.1x) on these kernels.
  int tile = tile_given, split = split_given;
  if (tile_given < 0) xcd_slice_major((int)gridDim.x, tile, split);
  const int m0 = (tile / ncb) * 128, ci0 = (tile % ncb) * 32;

  int kbeg = 0, kend = p.K;   // K = output pixels, in stages of one 32-pixel segment
  if (gridDim.y > 1) {
    kbeg = split * p.ktiles_per_split * SEG;
    kend = min(p.K, kbeg + p.ktiles_per_split * SEG);
  }
  const int nstage = kbeg < kend ? (kend - kbeg) / SEG : 0;

  // segment position (wave-uniform), advanced one segment per stage
  int sx = kbeg % s.Wo, t0 = kbeg / s.Wo;
  int sy = t0 % s.Ho, sb = t0 / s.Ho;

  // ---- dY loads: as the generic weight gradient (rows of 128 channels, swizzled), 2 row groups per wavefront
  int a_col[2], a_rowk[2];
  bool a_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 4 * (wave * 2 + j) + (lane >> 4);
    const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
    a_rowk[j] = r;
    a_col[j] = m0 + 8 * ch;
    a_ok[j] = a_col[j] < p.M;
  }
  // ---- halo loads: group T = wave + 4j covers halo pixels 16T .. 16T+15; this lane owns pixel 16T + lane/4
  int b_ry[2], b_rx[2], b_c[2];
  bool b_in[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int T = wave + 4 * j;
    const int hp = 16 * T + (lane >> 2);
    b_in[j] = T < B_G && hp < HP;
    b_ry[j] = hp / HPW - s.pad;
    b_rx[j] = hp % HPW - s.pad;
    b_c[j] = ci0 + 8 * (lane & 3);
  }
  const long lo_delta_a = q.A_lo - q.A_hi, lo_delta_b = q.B_lo - q.B_hi;

  auto issue = [&](int k0, int buf) {
    char* St = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long aoff = (long)(k0 + a_rowk[j]) * s.Cout + a_col[j];
      const int dst = (wave * 2 + j) * 1024;
      dma16b(a_ok[j] ? (const void*)(q.A_hi + aoff) : zero_page, St + dst);
      if (NP == 2) dma16b(a_ok[j] ? (const void*)(q.A_hi + aoff + lo_delta_a) : zero_page, St + A_PL + dst);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int T = wave + 4 * j;
      if (T >= B_G) continue;   // wave-uniform
      const int y = sy * s.stride + b_ry[j], x = sx * s.stride + b_rx[j];
      const bool ok = b_in[j] && (unsigned)y < (unsigned)s.H && (unsigned)x < (unsigned)s.W;
      const long boff = ((long)(sb * s.H + y) * s.W + x) * s.Cin + b_c[j];
      dma16b(ok ? (const void*)(q.B_hi + boff) : zero_page, St + NP * A_PL + T * 1024);
      if (NP == 2) dma16b(ok ? (const void*)(q.B_hi + boff + lo_delta_b) : zero_page, St + NP * A_PL + B_PL + T * 1024);
    }
    sx += SEG;   // next segment (the output width is a multiple of 32)
    if (sx >= s.Wo) { sx = 0; if (++sy == s.Ho) { sy = 0; ++sb; } }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool do_bias = BIAS && p.bias_slab != nullptr && ci0 == 0;   // block-uniform: the first input-channel block's workgroups
  f32x16 bacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) bacc[r] = 0.f;

  // transposed fragment reads (lane = 32h + 16g + 4ql + pl supplies row 8h + 4r2 + ql, receives column 16g + lane%16)
  const int h = lane >> 5, g = (lane >> 4) & 1, ql = (lane >> 2) & 3, pl = lane & 3;
  int a_rd[2];   // dY: this wavefront's 32 output channels start at channel 32*wave of the 128-channel image
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int row = 8 * h + 4 * r2 + ql;
    const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
    const int ch = (wave * 32) / 8 + 2 * g + (pl >> 1);
    a_rd[r2] = 256 * row + 16 * (ch ^ swz) + 8 * (pl & 1);
  }
  const int b_rd = (8 * h + ql) * 64 + g * 32 + pl * 8;   // halo: row (8h + ql) [+ 4 r2 + 16 step + tap shift], 32 ci

  if (nstage > 0) issue(kbeg, 0);
  for (int st = 0; st < nstage; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nstage) issue(kbeg + (st + 1) * SEG, (st + 1) & 1);
    const char* St = lds + (st & 1) * STAGE;
    const char* Bh = St + NP * A_PL;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const char* As = St + ks * 16 * 256;
      const bf16x8 ah = lds_tr8(As + a_rd[0], As + a_rd[1]);
      const bf16x8 al = NP == 2 ? lds_tr8(As + A_PL + a_rd[0], As + A_PL + a_rd[1]) : ah;
      if (BIAS && do_bias) x3_bias_mma<NP>(bacc, ah, al);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int kh = t / 3, kw = t % 3;
        const char* Bs = Bh + ((kh * HPW + kw + 16 * ks) * 64) + b_rd;
        const bf16x8 bh = lds_tr8(Bs, Bs + 4 * 64);
        const bf16x8 bl = NP == 2 ? lds_tr8(Bs + B_PL, Bs + B_PL + 4 * 64) : bh;
        x3_mma<NP>(acc[t], ah, al, bh, bl);
      }
    }
  }

  if (BIAS && do_bias) x3_bias_store(p, bacc, m0 + wave * 32, gridDim.y > 1 ? split : 0, lane);
  // epilogue: tile t of this wavefront is dW[m0 + 32*wave + row][tap t][ci0 + col]
  const Epilogue& e = p.e;
  const int col = ci0 + (lane & 31);
#pragma unroll
  for (int t = 0; t < TAPS; ++t) {
    const int n = t * s.Cin + col;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= p.M) continue;
      float v = acc[t][r];
      if (gridDim.y > 1) {
        float* dst = p.slab + ((size_t)split * p.M + m) * p.ldc + n;
        if (p.tickets) st_sc1(dst, v); else *dst = v;
        continue;
      }
      const size_t o = (size_t)m * p.ldc + n;
      v *= pow2i(-p.in_shift);
      if (e.scale) v *= e.scale[m];
      if (e.residual) v += e.residual[o];
      p.C[o] = v;
    }
  }
  if (!(gridDim.y > 1 && p.tickets)) return;
  // split-K finishing by the tile's last-arriving slice (conv_igemm.hip: splitk_fold): 128 rows x nine 32-column runs
  __syncthreads();
  if (!splitk_last_arrival(p, reinterpret_cast<int*>(lds), tile)) return;
  constexpr int NPIECE = 128 * TAPS * 8 / 256;          // float4 pieces per thread
  splitk_fold(p, NPIECE, [&](int k, int& m, int& n) {
    const int c = tid + 256 * k;                        // (row, tap, 4-column group): the group is fastest
    const int grp = c & 7, tp = (c >> 3) % TAPS, row = c / (8 * TAPS);
    m = m0 + row;
    n = tp * s.Cin + ci0 + 4 * grp;
    return m < p.M;
  }, [&](int m, int n, float4 v) {
    const size_t o = (size_t)m * p.ldc + n;
    const float a = pow2i(-p.in_shift);                 // exact
    v.x *= a; v.y *= a; v.z *= a; v.w *= a;
    if (e.scale) {
      const float sc = e.scale[m];
      v.x = __fmul_rn(v.x, sc); v.y = __fmul_rn(v.y, sc); v.z = __fmul_rn(v.z, sc); v.w = __fmul_rn(v.w, sc);
    }
    if (e.residual) {
      const float4 rr = *reinterpret_cast<const float4*>(e.residual + o);
      v.x = __fadd_rn(v.x, rr.x); v.y = __fadd_rn(v.y, rr.y); v.z = __fadd_rn(v.z, rr.z); v.w = __fadd_rn(v.w, rr.w);
    }
    *reinterpret_cast<float4*>(p.C + o) = v;
  });
}

template <int NP = 2, bool BIAS = false>
__global__ __launch_bounds__(256, 2) void igemm_x3_wgrad_halo_kernel(const Params p, const X3Planes q) {
  x3_wgrad_halo_body<NP, BIAS>(p, q);
}

template <int NP = 2>
__global__ __launch_bounds__(256, 2) void igemm_x3_wgrad_halo_group_kernel(const Params p_in, const X3Group G) {
  Params p;
  X3Planes q;
  int z, tile, split;
  xcd_group_slice_major((int)gridDim.x, z, tile, split);
  x3_group_member(p_in, G, p, q, z);
  x3_wgrad_halo_body<NP, false>(p, q, tile, split);
}

// ---- the splitting pre-passes ---------------------------------------------------------------------
__device__ __forceinline__ void split1(float v, __bf16& h, __bf16& l) {
  h = (__bf16)v;
  l = (__bf16)(v - (float)h);
}

// hi[i], lo[i] <- src[i]; eight elements per thread (n8 = n / 8), scalar tail by the last threads.
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ hi,
                                                         __bf16* __restrict__ lo, long n) {
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 hh, ll;
      split1(v[e], hh, ll);
      h[e] = hh; l[e] = ll;
    }
    reinterpret_cast<bf16x8*>(hi)[i] = h;
    reinterpret_cast<bf16x8*>(lo)[i] = l;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = (n8 << 3) + threadIdx.x;
    split1(src[i], hi[i], lo[i]);
  }
}

// The same into PAIRED planes (x3_paired_index): src is [rows][K], K % 32 == 0, n = rows * K.
__global__ __launch_bounds__(256) void split_bf16_paired_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                long n, int K) {
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 hh, ll;
      split1(v[e], hh, ll);
      h[e] = hh; l[e] = ll;
    }
    const long row = (i << 3) / K, k = (i << 3) - row * K;
    __bf16* o = dst + x3_paired_index(row, k, K);
    *reinterpret_cast<bf16x8*>(o) = h;
    *reinterpret_cast<bf16x8*>(o + 32) = l;
  }
}

// fp16 path: ONE plane h[i] = fp16(src[i] * 2^shift) (shift: the gradient-plane exponent offset, 0 for activations
// and weights).
__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ h, long n,
                                                        int shift) {
  const long n8 = n >> 3;
  const float sc = pow2i(shift);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const float4 a = reinterpret_cast<const float4*>(src)[2 * i], b = reinterpret_cast<const float4*>(src)[2 * i + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    f16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (_Float16)(v[e] * sc);
    reinterpret_cast<f16x8*>(h)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = (n8 << 3) + threadIdx.x;
    h[i] = (_Float16)(src[i] * sc);
  }
}

// W [Cout][taps][Cin] -> planes of W^T [Cin][taps][Cout] (the K-contiguous B operand of the data gradient).
// row_scale (optional, [Cout]) multiplies row co first: the FrozenBN scale folded into the data gradient.
__global__ __launch_bounds__(256) void split_bf16_transposed_kernel(const float* __restrict__ w,
                                                                    const float* __restrict__ row_scale,
                                                                    __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                                    int Cout, int taps, int Cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * taps + tap) * Cin + ci] * (row_scale ? row_scale[co] : 1.f) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < Cin && co < Cout) {
      const size_t o = x3_paired(hi, lo, 2) ? x3_paired_index(ci, (size_t)tap * Cout + co, (size_t)taps * Cout)
                                            : ((size_t)ci * taps + tap) * Cout + co;
      if (lo) split1(tile[tx][r], hi[o], lo[o]);
      else reinterpret_cast<_Float16*>(hi)[o] = (_Float16)tile[tx][r];
    }
  }
}

// ---- many weights in one launch ---------------------------------------------------------------------
// A training step re-splits every trainable weight once (straight for the forward, transposed + BN-scaled
// for the data gradient): ~150 tiny launches.  These two kernels walk a device table instead.
struct SplitEntry {        // 8 x 8 bytes, built by the host as int64 words
  const float* src;        // weight [Cout][taps][Cin]
  __bf16* hi;
  __bf16* lo;              // null: write ONE fp16 plane into `hi` (the fp16 path)
  const float* row_scale;  // transposed form only (nullable)
  long first_block;        // first workgroup of this entry
  long n;                  // straight: elements.  transposed: Cout
  long taps;               // transposed only
  long cin;                // transposed only
};

__device__ __forceinline__ int find_entry(const SplitEntry* __restrict__ tab, int n, long block) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].first_block <= block) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// straight: each workgroup splits 2048 consecutive elements of its entry
__global__ __launch_bounds__(256) void split_bf16_multi_kernel(const SplitEntry* __restrict__ tab, int nent) {
  const int e = find_entry(tab, nent, blockIdx.x);
  const SplitEntry t = tab[e];
  const long base = ((long)blockIdx.x - t.first_block) * 2048 + threadIdx.x * 8;
  if (base + 8 <= t.n) {
    const float4 a = *reinterpret_cast<const float4*>(t.src + base), b = *reinterpret_cast<const float4*>(t.src + base + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    if (!t.lo) {   // fp16 form: one plane
      f16x8 o;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = (_Float16)v[k];
      *reinterpret_cast<f16x8*>(t.hi + base) = o;
      return;
    }
    bf16x8 h, l;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      __bf16 hh, ll;
      split1(v[k], hh, ll);
      h[k] = hh; l[k] = ll;
    }
    if (x3_paired(t.hi, t.lo, 2)) {   // paired planes: the record's `cin` word is the row length K (K % 32 == 0)
      const long row = base / t.cin, k = base - row * t.cin;
      __bf16* o = t.hi + x3_paired_index(row, k, t.cin);
      *reinterpret_cast<bf16x8*>(o) = h;
      *reinterpret_cast<bf16x8*>(o + 32) = l;
      return;
    }
    *reinterpret_cast<bf16x8*>(t.hi + base) = h;
    *reinterpret_cast<bf16x8*>(t.lo + base) = l;
  } else {
    for (long i = base; i < t.n; ++i) {
      if (t.lo) split1(t.src[i], t.hi[i], t.lo[i]);
      else reinterpret_cast<_Float16*>(t.hi)[i] = (_Float16)t.src[i];
    }
  }
}

// transposed: each workgroup handles one 32x32 (ci, co) tile of one tap of its entry
__global__ __launch_bounds__(256) void split_bf16_transposed_multi_kernel(const SplitEntry* __restrict__ tab, int nent) {
  __shared__ float tile[32][33];
  const int e = find_entry(tab, nent, blockIdx.x);
  const SplitEntry t = tab[e];
  const int Cout = (int)t.n, taps = (int)t.taps, Cin = (int)t.cin;
  const int tx_n = (Cin + 31) / 32, ty_n = (Cout + 31) / 32;
  long r = (long)blockIdx.x - t.first_block;
  const int bx = (int)(r % tx_n); r /= tx_n;
  const int by = (int)(r % ty_n);
  const int tap = (int)(r / ty_n);
  const int ci0 = bx * 32, co0 = by * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int rr = ty; rr < 32; rr += 8) {
    const int co = co0 + rr, ci = ci0 + tx;
    tile[rr][tx] = (co < Cout && ci < Cin) ? t.src[((size_t)co * taps + tap) * Cin + ci] * (t.row_scale ? t.row_scale[co] : 1.f) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int rr = ty; rr < 32; rr += 8) {
    const int ci = ci0 + rr, co = co0 + tx;
    if (ci < Cin && co < Cout) {
      const size_t o = x3_paired(t.hi, t.lo, 2) ? x3_paired_index(ci, (size_t)tap * Cout + co, (size_t)taps * Cout)
                                                : ((size_t)ci * taps + tap) * Cout + co;
      if (t.lo) split1(tile[tx][rr], t.hi[o], t.lo[o]);
      else reinterpret_cast<_Float16*>(t.hi)[o] = (_Float16)tile[tx][rr];
    }
  }
}

// Every stage inside one filter tap, and 16-byte chunks never straddling the end of K.
inline bool x3_eligible(int role, const ConvShape& s) {
  if (role == WGRAD) return s.Cin % 8 == 0 && s.Cout % 8 == 0;   // 16-byte chunks of 8 channels inside one tap
  const int taps = s.KH * s.KW;
  const int c = role == FWD ? s.Cin : s.Cout;
  return c % XBK == 0 || (taps == 1 && c % 8 == 0);
}

// Longest K sweep (in stages) the single-buffered four-wave kernels take (see launch_x3_cfg; env overrides for sweeps).
inline int x3_nbuf1_stages(int role) {
  static const int fwd = [] { const char* e = getenv("JTSM_X3_NBUF1_STAGES_FWD"); return e ? atoi(e) : 2; }();
  static const int dgrad = [] { const char* e = getenv("JTSM_X3_NBUF1_STAGES_DGRAD"); return e ? atoi(e) : 2; }();
  return role == FWD ? fwd : dgrad;
}

inline bool x3_ring_enabled() {
  static const bool on = [] { const char* e = getenv("JTSM_X3_RING"); return !e || atoi(e) != 0; }();
  return on;
}

// Launch one tile configuration, with its split-K plan.
template <int ROLE, int WM, int WN, int TM, int TN, int NP>
int launch_x3_cfg(Params& p, const X3Planes& q, void* workspace, size_t workspace_bytes, int round_blocks,
                  hipStream_t st) {
  constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, NT = 64 * WM * WN;
  const int ntiles = ceil_div(p.N, BN) * ceil_div(p.M, BM);
  const int ktiles = ceil_div(p.K, XBK);
  // aim at one full round of resident workgroups (measured: tools/sweeps/x3_sweep.py)
  int splits = plan_splits(ntiles, ktiles, round_blocks);
  if (splits > 1 && (size_t)splits * p.M * p.ldc * sizeof(float) > workspace_bytes) splits = 1;
  if (p.colsum) splits = 1;   // (column sums are taken where the finished result is written: one K slice)
  if (splits > 1) {
    p.ktiles_per_split = ceil_div(ktiles, splits);
    splits = ceil_div(ktiles, p.ktiles_per_split);
    p.slab = reinterpret_cast<float*>(workspace);
  }
  p.wide = p.N % 4 == 0 && p.ldc % 4 == 0 && aligned16(p.C) && (splits <= 1 || aligned16(p.slab)) &&
           (!p.e.residual || aligned16(p.e.residual)) && (!p.e.mask || aligned16(p.e.mask)) &&
           (!p.e.scale || aligned16(p.e.scale)) && (!p.e.bias || aligned16(p.e.bias));
  JTSM_REQUIRE(!p.out_hi || p.wide, "conv bf16x3: output planes requested but the tensors are not 16-byte aligned");
  JTSM_REQUIRE((!p.mask_plane && !p.scale_rows && p.C && !p.colsum && !p.e.residual_h) || p.wide,
               "conv bf16x3: a gate plane, a row scale, column sums, an fp16 residual plane or a planes-only result need N %% 4 == 0 and 16-byte aligned tensors");
  const dim3 grid(ntiles, splits > 1 ? splits : 1);
  const bool fused = use_fused_finish(p, ntiles, splits, st);
  if (NT == 256 && ceil_div(ktiles, splits > 1 ? splits : 1) <= x3_nbuf1_stages(ROLE))
    // A sweep of one or two stages is bound by its output / residual traffic, not by the matrix pipes: the
    // single-buffered instantiation (32-40 KiB of LDS, three workgroups per CU) keeps more of it in flight.
    // (Up to round 2 this was <= 4 stages at four workgroups per CU; with the pipelined epilogue the four-stage
    // layers run faster double-buffered — 128 -> 512 channels at 128 x 128: 56 us against 65 us — sweeps:
    // JTSM_X3_NBUF1_STAGES_FWD / _DGRAD.)
    hipLaunchKernelGGL((igemm_x3_kernel<ROLE, WM, WN, TM, TN, 1, NP>), grid, dim3(NT), 0, st, p, q);
  else if (BM == 64 && BN == 64 && x3_ring_enabled() && ceil_div(ktiles, splits > 1 ? splits : 1) >= 4)
    // 64 x 64 tiles exist for the long-K 1x1 layers of res4 / res5 (16-64 stages per workgroup): a four-stage ring
    // (64 KiB of LDS, still two workgroups per CU) keeps three stages of loads in flight.  JTSM_X3_RING=0: the
    // double-buffered instantiation (sweeps).
    hipLaunchKernelGGL((igemm_x3_kernel<ROLE, WM, WN, TM, TN, (BM == 64 && BN == 64) ? 4 : 2, NP>), grid, dim3(NT), 0, st,
                       p, q);
  else
    hipLaunchKernelGGL((igemm_x3_kernel<ROLE, WM, WN, TM, TN, 2, NP>), grid, dim3(NT), 0, st, p, q);
  JTSM_CHECK_LAUNCH("igemm bf16x3");
  record_mid(st);
  if (splits > 1 && !fused) return finish_split(p, splits, st);
  return JTSM_OK;
}

// Tile choice.  0: 128x128 (4 waves), 1: 256x64 (4 waves, narrow outputs), 2: 256x256 (8 waves, large layers).
inline int x3_tile_choice(const Params& p) {
  if (p.N <= 64) return 1;
  // 256x256 pays once its (fewer, larger) workgroups still fill the chip and K is deep enough to amortise them
  const long t256 = (long)ceil_div(p.N, 256) * ceil_div(p.M, 256);
  const int ktiles = ceil_div(p.K, XBK);
  // tiles x stages (measured crossover, tools/sweeps/x3_sweep.py + layer timings; JTSM_X3_MIN_WORK overrides for sweeps)
  static const long min_work = [] { const char* e = getenv("JTSM_X3_MIN_WORK"); return e ? atol(e) : 9216L; }();
  // (long sweeps pay earlier: 196 tiles x 32 stages — the 2x2 / stride-2 role of the mask heads' transposed convolution —
  // 133 -> 116 us on 256-tiles, while 8-stage layers of the same tiles x stages lose 3 %: two thirds of the bar from 16 stages)
  const long bar = ktiles >= 16 ? min_work * 2 / 3 : min_work;
  if (p.N >= 192 && t256 * ktiles >= bar && (p.N % 256 == 0 || p.N % 256 > 128)) return 2;
  // 64 x 64 tiles for the small layers (res4 / res5 1x1 convolutions on 64 x 64 and 32 x 32 maps): at 128 x 128 they
  // are <= 128 tiles and need 4-8 K slices to fill the chip — each slice writes a slab and a finishing launch folds
  // them; four times as many small tiles fill it with 1-2 slices.
  static const bool tile64 = [] { const char* e = getenv("JTSM_X3_TILE64"); return !e || atoi(e) != 0; }();
  const long t128 = (long)ceil_div(p.N, 128) * ceil_div(p.M, 128);
  // (with the four-stage ring the 64-tiles take layers of up to 256 128-tiles — res5 conv3 / shortcut, 512 -> 2048 on 32 x 32:
  // 1024 small tiles unsplit instead of 256 x 2 slices + a finishing launch.  Same-box sweep, ms per step: 128: 25.14,
  // 256: 24.99, 512: 25.17; together with the 512 work-list target below: 24.88.)
  static const long tile64_max = [] { const char* e = getenv("JTSM_X3_TILE64_MAX_T128"); return e ? atol(e) : 256L; }();   // (sweeps)
  if (tile64 && t128 <= tile64_max && ktiles >= 8 && p.N >= 64 && p.M >= 64) return 3;
  return 0;
}

// Work-list length the 64 x 64 tiles' K slicing aims at (two workgroups per CU are resident: 512 per round).
inline int x3_tile64_target() {
  // (1024 — two rounds — up to the ring: long K sweeps were latency bound then and more, shorter slices paid; with three
  // stages in flight one round of longer slices wins: half the slabs and finishing work.  Sweep: 25.14 -> 24.97 ms per step.)
  static const int v = [] { const char* e = getenv("JTSM_X3_TILE64_TARGET"); return e ? atoi(e) : 512; }();   // (sweeps)
  return v;
}

// K slices the launcher will want for this problem (before the workspace clamp).
inline int x3_wanted_splits(const Params& p) {
  const int c = x3_tile_choice(p);
  const int bm = c == 0 ? 128 : (c == 3 ? 64 : 256), bn = c == 0 ? 128 : (c == 1 || c == 3 ? 64 : 256);
  return plan_splits(ceil_div(p.N, bn) * ceil_div(p.M, bm), ceil_div(p.K, XBK), c == 2 ? 256 : (c == 3 ? x3_tile64_target() : 512));
}

// The halo kernel serves k x k (k > 1), stride-1, undilated layers whose contracted channels come in blocks of 32
// and whose (8+k-1) x (16+k-1) halo fits twelve 16-row groups (3x3).
inline bool x3_halo_ok(int role, const Params& p) {
  const ConvShape& s = p.s;
  const int c = role == FWD ? s.Cin : s.Cout;
  const int OH = role == FWD ? s.Ho : s.H, OW = role == FWD ? s.Wo : s.W;
  // maps below 32 x 32 waste too much of the 8 x 16 patches; layers large enough for 256 x 256 tiles stay there
  // (measured equal or better, tools/sweeps/x3_sweep.py).
  // Measured alternative, default off (JTSM_X3_HALO_SMALL=1) — pooled roi maps (the mask heads' 14 x 14): ONE 16 x 16
  // patch of the 256-channel kernel holds a whole image, its 16 x 16 input (the map and its zero border) IS the halo, so
  // the activations travel to LDS once per 32 channels instead of once per tap and the launch has one workgroup per
  // roi.  But a quarter of the patch rows are computed for nothing, and this kernel is the one closest to the matrix
  // pipes' limit (55 % busy): 172 / 176 us against 157 / 169 us (forward / data gradient, 255 rois) on the generic
  // 256 x 256 tiling.
  static const bool small_maps = [] { const char* e = getenv("JTSM_X3_HALO_SMALL"); return e && atoi(e) != 0; }();
  const bool whole_image = small_maps && OH <= 16 && OW <= 16 && OH * OW >= 160 && x3_tile_choice(p) == 2;
  return !p.scatter && s.KH == 3 && s.KW == 3 && s.stride == 1 && s.dil == 1 && c % XBK == 0 && p.N > 64 &&
         ((OH >= 32 && OW >= 32) || whole_image);
}

template <int ROLE, bool BIG, int NP>
int launch_x3_halo(Params& p, const X3Planes& q, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const ConvShape& s = p.s;
  constexpr int TH = BIG ? 16 : 8, BN = BIG ? 256 : 128;
  const int OH = ROLE == FWD ? s.Ho : s.H, OW = ROLE == FWD ? s.Wo : s.W;
  const int ntiles = ceil_div(p.N, BN) * s.Bn * ceil_div(OH, TH) * ceil_div(OW, 16);
  const int Cb = (ROLE == FWD ? s.Cin : s.Cout) / XBK;
  // split-K cuts whole channel blocks (nine stages each): aim at one round of resident workgroups
  const int round_blocks = BIG ? 256 : 512;
  int splits = ntiles >= round_blocks ? 1 : round_blocks / ntiles;
  if (splits > Cb) splits = Cb;
  if (splits > 16) splits = 16;
  if (splits < 1) splits = 1;
  if (splits > 1 && (size_t)splits * p.M * p.ldc * sizeof(float) > workspace_bytes) splits = 1;
  if (splits > 1) {
    p.ktiles_per_split = ceil_div(Cb, splits);
    splits = ceil_div(Cb, p.ktiles_per_split);
    p.slab = reinterpret_cast<float*>(workspace);
  }
  p.wide = p.N % 4 == 0 && p.ldc % 4 == 0 && aligned16(p.C) && (splits <= 1 || aligned16(p.slab)) &&
           (!p.e.residual || aligned16(p.e.residual)) && (!p.e.mask || aligned16(p.e.mask)) &&
           (!p.e.scale || aligned16(p.e.scale)) && (!p.e.bias || aligned16(p.e.bias));
  JTSM_REQUIRE(!p.out_hi || p.wide, "conv bf16x3: output planes requested but the tensors are not 16-byte aligned");
  JTSM_REQUIRE((!p.mask_plane && !p.scale_rows && p.C) || p.wide,
               "conv bf16x3: a gate plane, a row scale or a planes-only result needs N %% 4 == 0 and 16-byte aligned tensors");
  const dim3 grid(ntiles, splits > 1 ? splits : 1);
  const bool fused = use_fused_finish(p, ntiles, splits, st);
  if (BIG) hipLaunchKernelGGL((igemm_x3_halo_kernel<ROLE, 16, 4, 2, 4, 21, NP>), grid, dim3(512), 0, st, p, q);
  else hipLaunchKernelGGL((igemm_x3_halo_kernel<ROLE, 8, 2, 2, 2, 12, NP>), grid, dim3(256), 0, st, p, q);
  JTSM_CHECK_LAUNCH("igemm bf16x3 halo");
  record_mid(st);
  if (splits > 1 && !fused) return finish_split(p, splits, st);
  return JTSM_OK;
}

template <int ROLE, int NP = 2>
int launch_split_x3(Params& p, const X3Planes& q, void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (!p.colsum && !p.e.residual_h && x3_halo_ok(ROLE, p)) {   // (column sums: the generic kernel's row tiles; so is the fp16 residual plane)
    if (x3_tile_choice(p) == 2) return launch_x3_halo<ROLE, true, NP>(p, q, workspace, workspace_bytes, st);
    return launch_x3_halo<ROLE, false, NP>(p, q, workspace, workspace_bytes, st);
  }
  switch (x3_tile_choice(p)) {
    case 1: return launch_x3_cfg<ROLE, 4, 1, 2, 2, NP>(p, q, workspace, workspace_bytes, 512, st);
    case 2: return launch_x3_cfg<ROLE, 4, 2, 2, 4, NP>(p, q, workspace, workspace_bytes, 256, st);
    case 3: return launch_x3_cfg<ROLE, 2, 2, 1, 1, NP>(p, q, workspace, workspace_bytes, x3_tile64_target(), st);
    default: return launch_x3_cfg<ROLE, 2, 2, 2, 2, NP>(p, q, workspace, workspace_bytes, 512, st);
  }
}
