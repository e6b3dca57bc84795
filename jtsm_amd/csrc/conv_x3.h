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

struct X3Planes {
  const __bf16* A_hi; const __bf16* A_lo;
  const __bf16* B_hi; const __bf16* B_lo;
};

__device__ __forceinline__ void dma16b(const void* g, void* lds_uniform_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_uniform_base, 16, 0, 0);
}

// SK = k per stage: 32 (64-byte rows per plane, 16 rows per load) or 64 (128-byte rows, 8 rows per load).
template <int ROLE, int BM, int BN, int NBUF, int SK = 32>
__global__ __launch_bounds__(256, (NBUF == 1 && SK == 32) ? 4 : (NBUF * SK <= 64 ? 2 : 1)) void igemm_x3_kernel(const Params p, const X3Planes q) {
  constexpr int RB = SK * 2;            // bytes per row per plane per stage
  constexpr int RPI = 1024 / RB;        // rows per direct-to-LDS load
  constexpr int CPW = RB / 16;          // 16-byte chunks per row
  constexpr int SWS = SK == 32 ? 2 : 1, SWM = CPW - 1;   // slot = chunk ^ ((row >> SWS) & SWM)
  static_assert(ROLE == FWD || ROLE == DGRAD, "bf16x3: forward and data-gradient roles");
  constexpr int WN = BN / 64;
  static_assert((BM / 64) * WN == 4, "four wavefronts of 64x64");
  constexpr int A_PL = BM * RB, B_PL = BN * RB;          // bytes per plane per stage
  constexpr int STAGE = 2 * A_PL + 2 * B_PL;
  constexpr int A_INS = BM / RPI / 4, B_INS = BN / RPI / 4;   // loads per wavefront per plane per stage
  __shared__ __attribute__((aligned(16))) char lds[NBUF * STAGE];

  const ConvShape& s = p.s;
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
    kbeg = blockIdx.y * p.ktiles_per_split * XBK;   // (split bookkeeping stays in units of 32 k)
    kend = min(p.K, kbeg + p.ktiles_per_split * XBK);
  }
  const int ntile_k = kbeg < kend ? (kend - kbeg + SK - 1) / SK : 0;

  // ---- addressing state: the tap and first channel of a stage are wave-uniform and advance incrementally
  const int Cdim = ROLE == FWD ? s.Cin : s.Cout;
  int t_kh, t_kw, t_c;
  {
    const int tap = kbeg / Cdim;
    t_c = kbeg - tap * Cdim;
    t_kh = tap / s.KW;
    t_kw = tap - t_kh * s.KW;
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
    const int r = (wave * A_INS + j) * RPI + lane / CPW;
    arow[j] = ROLE == FWD ? fwd_pixel(s, m0 + r, p.M) : dgrad_pixel(s, m0 + r, p.M);
    a_chunk[j] = 8 * ((lane % CPW) ^ ((r >> SWS) & SWM));
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
  int b_off[B_INS], b_chunk[B_INS];
#pragma unroll
  for (int j = 0; j < B_INS; ++j) {
    const int r = (wave * B_INS + j) * RPI + lane / CPW;
    b_off[j] = (n0 + r) < p.N ? (n0 + r) * p.K : -1;
    b_chunk[j] = 8 * ((lane % CPW) ^ ((r >> SWS) & SWM));
  }
  const long lo_delta_a = q.A_lo - q.A_hi, lo_delta_b = q.B_lo - q.B_hi;   // the lo plane sits at a fixed distance

  // One direct-to-LDS load ("piece") of the next stage: pieces 0 .. 2*A_INS-1 are the A rows (hi, lo
  // alternating), the rest the B rows.  The K loop issues them one at a time BETWEEN its MFMA groups: a
  // load costs the issuing wave 60-180 cycles (MI355X_MICROARCH.md, cycle constants), which eight loads
  // in a row would take out of the matrix pipe's time, while one load per three MFMAs hides in their shadow.
  constexpr int PIECES = 2 * A_INS + 2 * B_INS;
  auto issue_piece = [&](int idx, int k0, int buf) {
    char* St = lds + buf * STAGE;
    if (idx < 2 * A_INS) {
      const int j = idx >> 1, lo = idx & 1;
      bool ok;
      int off;
      if (linear) {
        const int tap_delta = ROLE == FWD ? ((t_kh * s.dil) * s.W + t_kw * s.dil) * s.Cin + t_c
                                          : -((t_kh * s.dil) * s.Wo + t_kw * s.dil) * s.Cout + t_c;
        ok = ((a_mh[j] >> t_kh) & (a_mw[j] >> t_kw) & 1u) != 0u && (k0 + a_chunk[j]) < kend;
        off = a_base[j] + tap_delta;
      } else {
        const PixelRow& r = arow[j];
        const int th = r.h0 - t_kh * s.dil, tw = r.w0 - t_kw * s.dil;
        const int oh = th / s.stride, ow = tw / s.stride;
        ok = r.ok && (k0 + a_chunk[j]) < kend && th >= 0 && tw >= 0 && oh * s.stride == th && ow * s.stride == tw &&
             oh < s.Ho && ow < s.Wo;
        off = ((r.b * s.Ho + oh) * s.Wo + ow) * s.Cout + t_c + a_chunk[j];
      }
      const __bf16* src = q.A_hi + off + (lo ? lo_delta_a : 0);
      dma16b(ok ? (const void*)src : (const void*)g_zero_page, St + lo * A_PL + (wave * A_INS + j) * 1024);
    } else {
      const int j = (idx - 2 * A_INS) >> 1, lo = (idx - 2 * A_INS) & 1;
      const bool ok = b_off[j] >= 0 && (k0 + b_chunk[j]) < kend;
      const __bf16* src = q.B_hi + (b_off[j] + k0 + b_chunk[j]) + (lo ? lo_delta_b : 0);
      dma16b(ok ? (const void*)src : (const void*)g_zero_page, St + 2 * A_PL + lo * B_PL + (wave * B_INS + j) * 1024);
    }
  };
  auto advance_tap = [&]() {   // next stage: same tap or the next one (stages never straddle taps)
    t_c += SK;
    if (t_c >= Cdim) {
      t_c -= Cdim;
      if (++t_kw == s.KW) { t_kw = 0; ++t_kh; }
    }
  };
  auto issue = [&](int k0, int buf) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) issue_piece(i, k0, buf);
    advance_tap();
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int a_off[2], a_swz[2], bb_off[2], b_swz[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int ar = wm * 64 + t * 32 + li, br = wn * 64 + t * 32 + li;
    a_off[t] = ar * RB;
    a_swz[t] = (ar >> SWS) & SWM;
    bb_off[t] = 2 * A_PL + br * RB;
    b_swz[t] = (br >> SWS) & SWM;
  }

  constexpr int KSTEPS = SK / 16;
  constexpr int SLOTS = 4 * KSTEPS;                         // MFMA groups per stage (k-steps x 2 x 2 tiles)
  constexpr int PER_SLOT = (PIECES + SLOTS - 1) / SLOTS;    // loads placed behind each group
  if (NBUF == 2 && ntile_k > 0) issue(kbeg, 0);
  for (int t = 0; t < ntile_k; ++t) {
    if (NBUF == 1) {
      __syncthreads();
      issue(kbeg + t * SK, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool more = NBUF == 2 && t + 1 < ntile_k;   // wave-uniform
    const int knext = kbeg + (t + 1) * SK, bnext = (t + 1) & 1;
    const char* St = lds + (NBUF == 2 ? (t & 1) : 0) * STAGE;
#pragma unroll
    for (int st = 0; st < KSTEPS; ++st) {
      const int c = 2 * st + half;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int ao = a_off[u] + ((c ^ a_swz[u]) << 4);
        ah[u] = *reinterpret_cast<const bf16x8*>(St + ao);
        al[u] = *reinterpret_cast<const bf16x8*>(St + A_PL + ao);
        const int bo = bb_off[u] + ((c ^ b_swz[u]) << 4);
        bh[u] = *reinterpret_cast<const bf16x8*>(St + bo);
        bl[u] = *reinterpret_cast<const bf16x8*>(St + B_PL + bo);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          if (NBUF == 2) {
            const int slot = st * 4 + i * 2 + j;
            if (more) {
#pragma unroll
              for (int e = 0; e < PER_SLOT; ++e)
                if (slot * PER_SLOT + e < PIECES) issue_piece(slot * PER_SLOT + e, knext, bnext);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the load behind this MFMA group
          }
        }
    }
    if (more) advance_tap();
  }
  constexpr int PASSES = NBUF * STAGE >= BM * BN * 4 ? 1 : 2;
  static_assert(NBUF * STAGE >= BM * BN * 4 / PASSES, "epilogue window does not fit the stage buffers");
  if (p.wide)
    store_tile_wide<ROLE, BM, BN, PASSES>(p, acc, m0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(lds));
  else
    store_tile<ROLE>(p, acc, m0, n0, wm, wn, lane);
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

template <int NBUF>
__global__ __launch_bounds__(256, 2) void igemm_x3_wgrad_kernel(const Params p, const X3Planes q) {
  constexpr int BM = 128, BN = 128;
  constexpr int PL = XBK * 256;            // bytes per plane per stage: 32 pixel rows x 128 channels
  constexpr int STAGE = 4 * PL;            // dY hi, dY lo, X hi, X lo
  constexpr int INS = 2;                   // 4-row loads per wavefront per plane per stage
  __shared__ __attribute__((aligned(16))) char lds[NBUF * STAGE];

  const ConvShape& s = p.s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

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
    if (kbeg >= kend && !p.wide) return;   // (slab path: an empty slice still writes its zero slab)
  }
  const int ntile_k = kbeg < kend ? (kend - kbeg + XBK - 1) / XBK : 0;

  // ---- load addressing: load j of this wave fills rows 4T .. 4T+3 (T = 2*wave + j); this lane owns row
  // r = 4T + lane/16 and the chunk whose slot is lane%16
  int a_col[INS], b_ci[INS], b_kh[INS], b_kw[INS], row_k[INS];
  bool a_ok[INS], b_ok[INS];
  PixState bpix[INS];
#pragma unroll
  for (int j = 0; j < INS; ++j) {
    const int r = 4 * (wave * INS + j) + (lane >> 4);
    const int ch = (lane & 15) ^ (((r & 3) << 2) | ((r >> 2) & 3));
    row_k[j] = r;
    a_col[j] = m0 + 8 * ch;
    a_ok[j] = a_col[j] < p.M;
    const int col = n0 + 8 * ch;
    b_ok[j] = col < p.N;
    const int cc = b_ok[j] ? col : 0;
    const int tap = cc / s.Cin;
    b_ci[j] = cc - tap * s.Cin;
    b_kh[j] = tap / s.KW;
    b_kw[j] = tap - b_kh[j] * s.KW;
    bpix[j] = pix_init(kbeg + r, s.Ho, s.Wo);
  }

  auto issue = [&](int k0, int buf) {
    char* St = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < INS; ++j) {
      const int dst = (wave * INS + j) * 1024;
      const int k = k0 + row_k[j];
      const bool oka = a_ok[j] && k < kend;
      const long aoff = (long)k * s.Cout + a_col[j];
      dma16b(oka ? (const void*)(q.A_hi + aoff) : (const void*)g_zero_page, St + dst);
      dma16b(oka ? (const void*)(q.A_lo + aoff) : (const void*)g_zero_page, St + PL + dst);
      const PixState& px = bpix[j];
      const int ih = px.oh * s.stride - s.pad + b_kh[j] * s.dil, iw = px.ow * s.stride - s.pad + b_kw[j] * s.dil;
      const bool okb = b_ok[j] && px.k < kend && (unsigned)ih < (unsigned)s.H && (unsigned)iw < (unsigned)s.W;
      const long boff = ((long)(px.b * s.H + ih) * s.W + iw) * s.Cin + b_ci[j];
      dma16b(okb ? (const void*)(q.B_hi + boff) : (const void*)g_zero_page, St + 2 * PL + dst);
      dma16b(okb ? (const void*)(q.B_lo + boff) : (const void*)g_zero_page, St + 3 * PL + dst);
      pix_advance(bpix[j], XBK, s.Ho, s.Wo);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- transposed fragment reads: lane = 32h + 16g + 4ql + pl supplies, for read r2 of step st, the address
  // of row 16st + 8h + 4r2 + ql, channels 32*tile + 16g + 4pl .. +3; it receives channel 16g + lane%16.
  const int h = lane >> 5, g = (lane >> 4) & 1, ql = (lane >> 2) & 3, pl = lane & 3;
  int a_rd[2][2], b_rd[2][2];   // [tile][r2], byte offset inside a plane for step 0
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r2 = 0; r2 < 2; ++r2) {
      const int row = 8 * h + 4 * r2 + ql;
      const int swz = ((row & 3) << 2) | ((row >> 2) & 3);
      const int cha = (wm * 64 + t * 32) / 8 + 2 * g + (pl >> 1);
      const int chb = (wn * 64 + t * 32) / 8 + 2 * g + (pl >> 1);
      a_rd[t][r2] = 256 * row + 16 * (cha ^ swz) + 8 * (pl & 1);
      b_rd[t][r2] = 256 * row + 16 * (chb ^ swz) + 8 * (pl & 1);
    }

  if (NBUF == 2 && ntile_k > 0) issue(kbeg, 0);
  for (int t = 0; t < ntile_k; ++t) {
    if (NBUF == 1) {
      __syncthreads();
      issue(kbeg + t * XBK, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (NBUF == 2 && t + 1 < ntile_k) issue(kbeg + (t + 1) * XBK, (t + 1) & 1);
    const char* St = lds + (NBUF == 2 ? (t & 1) : 0) * STAGE;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const char* Ss = St + st * 16 * 256;   // rows 16st ..; the swizzle does not depend on st
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ah[u] = lds_tr8(Ss + a_rd[u][0], Ss + a_rd[u][1]);
        al[u] = lds_tr8(Ss + PL + a_rd[u][0], Ss + PL + a_rd[u][1]);
        bh[u] = lds_tr8(Ss + 2 * PL + b_rd[u][0], Ss + 2 * PL + b_rd[u][1]);
        bl[u] = lds_tr8(Ss + 3 * PL + b_rd[u][0], Ss + 3 * PL + b_rd[u][1]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  if (p.wide)
    store_tile_wide<WGRAD, BM, BN>(p, acc, m0, n0, wm, wn, lane, tid, reinterpret_cast<float*>(lds));
  else
    store_tile<WGRAD>(p, acc, m0, n0, wm, wn, lane);
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
      const size_t o = ((size_t)ci * taps + tap) * Cout + co;
      split1(tile[tx][r], hi[o], lo[o]);
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

template <int ROLE, int BM, int BN>
int launch_split_x3(Params& p, const X3Planes& q, void* workspace, size_t workspace_bytes, hipStream_t st) {
  const int ntiles = ceil_div(p.N, BN) * ceil_div(p.M, BM);
  const int ktiles = ceil_div(p.K, XBK);
  // Two 64-KiB workgroups fit a CU: aim at one full round of 512 (measured: scratch/x3_sweep.py).
  int splits = plan_splits(ntiles, ktiles, 512);
  if (splits > 1 && (size_t)splits * p.M * p.ldc * sizeof(float) > workspace_bytes) splits = 1;
  if (splits > 1) {
    p.ktiles_per_split = ceil_div(ktiles, splits);
    splits = ceil_div(ktiles, p.ktiles_per_split);
    p.slab = reinterpret_cast<float*>(workspace);
  }
  p.wide = p.N % 4 == 0 && p.ldc % 4 == 0 && aligned16(p.C) && (splits <= 1 || aligned16(p.slab)) &&
           (!p.e.residual || aligned16(p.e.residual)) && (!p.e.mask || aligned16(p.e.mask)) &&
           (!p.e.scale || aligned16(p.e.scale)) && (!p.e.bias || aligned16(p.e.bias));
  JTSM_REQUIRE(!p.out_hi || p.wide, "conv bf16x3: output planes requested but the tensors are not 16-byte aligned");
  // A sweep of <= 4 stages is bound by its output / residual traffic, not by the matrix pipes: the
  // single-buffered instantiation (32-40 KiB of LDS, four workgroups per CU) keeps more of it in flight.
  // (SK = 64 — 128-byte rows, one workgroup per CU — measured 15-25 % slower on every large layer; not dispatched)
  if (ceil_div(ktiles, splits > 1 ? splits : 1) <= 4)
    hipLaunchKernelGGL((igemm_x3_kernel<ROLE, BM, BN, 1>), dim3(ntiles, splits > 1 ? splits : 1), dim3(256), 0, st, p, q);
  else
    hipLaunchKernelGGL((igemm_x3_kernel<ROLE, BM, BN, 2>), dim3(ntiles, splits > 1 ? splits : 1), dim3(256), 0, st, p, q);
  JTSM_CHECK_LAUNCH("igemm bf16x3");
  if (splits > 1) return finish_split(p, splits, st);
  return JTSM_OK;
}
