// Per-CU operand delivery rate, L2 -> CU, in the access pattern of the 256 x 256-tile contraction (one workgroup of 8
// wavefronts per CU; a stage = 512 rows x 64 B x 2 planes = 64 KiB; rows shared between workgroups as in a GEMM):
//   mode 0: global_load_lds_dwordx4 (direct to LDS), DEPTH stages in flight (LDS ring)
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, DEPTH register sets in flight
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS write)
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/l2_to_lds tools/sweeps/micro/l2_to_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16b(const void* g, void* lds) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// operand: planes hi, lo of [rows][K] bf16 (K * 2 bytes per row).  Workgroup (tm, tn): A rows tm*256.., B rows tn*256..
template <int MODE, int DEPTH>
__global__ __launch_bounds__(512, 2) void deliver(const char* __restrict__ A, const char* __restrict__ B, long plane_bytes,
                                                  int K, int stages, int ntn, float* __restrict__ sink, int paired, int rot) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tile = blockIdx.x;
  { const int nt = gridDim.x, qq = nt / 8, r = nt % 8, xcd = tile % 8, idx = tile / 8;
    tile = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + idx; }
  const int tm = tile / ntn, tn = tile % ntn;
  // wave w owns A rows 32w .. 32w+31 (2 pieces of 16 rows) and the same of B, both planes: 8 pieces per stage
  const char* src[8];
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 16 + (lane >> 2);
    const long offA = (long)(tm * 256 + r) * K * 2 * (paired ? 2 : 1) + (lane & 3) * 16;
    const long offB = (long)(tn * 256 + r) * K * 2 * (paired ? 2 : 1) + (lane & 3) * 16;
    // paired: a row's stage is ONE 128-byte line, [32 k of hi | 32 k of lo]
    src[2 * j] = A + offA; src[2 * j + 1] = A + (paired ? 64 : plane_bytes) + offA;
    src[4 + 2 * j] = B + offB; src[4 + 2 * j + 1] = B + (paired ? 64 : plane_bytes) + offB;
  }
  const long step = paired ? 128 : 64;
  // rot: workgroup w walks K from stage (rot * w) % stages on, wrapping — so that the chip does not ask every L2 channel
  // for the same k offset of (power-of-two-pitched) rows at the same time
  const int t_rot = rot ? (int)(((long)blockIdx.x * rot) % stages) : 0;
#define ST(t) (((t) + t_rot) % stages)
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    constexpr int STAGE = 65536;
    auto issue = [&](int t) {
      char* St = lds + (t % DEPTH) * STAGE;
#pragma unroll
      for (int i = 0; i < 8; ++i) dma16b(src[i] + (long)ST(t) * step, St + (wave * 8 + i) * 1024);
    };
    for (int t = 0; t < DEPTH - 1 && t < stages; ++t) issue(t);
    for (int t = 0; t < stages; ++t) {
      const int ahead = min(DEPTH - 2, stages - 1 - t);
      if (ahead >= 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (t + DEPTH - 1 < stages) issue(t + DEPTH - 1);
      acc += *reinterpret_cast<const f4*>(lds + (t % DEPTH) * STAGE + threadIdx.x * 16);
    }
  } else {
    f4 r[DEPTH][8];
    auto issue = [&](int t, int set) {
#pragma unroll
      for (int i = 0; i < 8; ++i) r[set][i] = *reinterpret_cast<const f4*>(src[i] + (long)ST(t) * step);
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) if (d < stages) issue(d, d);
    for (int t0 = 0; t0 < stages; t0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int t = t0 + d;
        if (t < stages) {
          if (t + DEPTH - 1 < stages) issue(t + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
          if (MODE == 1) {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<f4*>(lds + (t & 1) * 65536 + (wave * 8 + i) * 1024 + lane * 16) = r[d][i];
            __builtin_amdgcn_s_barrier();
            acc += *reinterpret_cast<const f4*>(lds + (t & 1) * 65536 + threadIdx.x * 16);
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += r[d][i];
          }
        }
      }
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int MODE, int DEPTH>
void run(const char* A, const char* B, long plane_bytes, int K, int ntm, int ntn, float* sink, const char* what, int paired = 0) {
  extern int g_rot;
  const int rot = g_rot;
  const int stages = K / 32;
  const int ldsb = MODE == 0 ? DEPTH * 65536 : (MODE == 1 ? 131072 : 0);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(deliver<MODE, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((deliver<MODE, DEPTH>), dim3(ntm * ntn), dim3(512), ldsb, 0, A, B, plane_bytes, K, stages, ntn, sink, paired, rot);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  const double bytes = (double)ntm * ntn * stages * 65536.0;
  const int wgs = ntm * ntn < 256 ? ntm * ntn : 256;     // CUs at work
  printf("%-44s %8.1f us  %6.2f TB/s  %5.1f GB/s per CU  %5.1f B/clk/CU at 2.4 GHz\n", what, best * 1e3, bytes / best / 1e9,
         bytes / best / 1e6 / wgs, bytes / best / 1e6 / wgs / 2.4);
}

int g_rot = 0;
int main(int argc, char** argv) {
  g_rot = argc > 4 ? atoi(argv[4]) : 0;
  // K (elements per row: the row pitch is 2 K bytes — a power of two by default, as the fully connected layers' operands
  // have; try 4160 for a pitch that is not) and the tile grid (ntm x ntn workgroups: fewer than 256 = a part of the chip)
  const int K = argc > 1 ? atoi(argv[1]) : 4096, ntm = argc > 2 ? atoi(argv[2]) : 16, ntn = argc > 3 ? atoi(argv[3]) : 16;
  printf("K %d (row pitch %d B), %d x %d workgroups, K order rotated by %d stages per workgroup\n", K, 2 * K, ntm, ntn, g_rot);
  const long plane_bytes = (long)(ntm > ntn ? ntm : ntn) * 256 * K * 2;
  char *A, *B; float* sink;
  CHECK(hipMalloc(&A, 2 * plane_bytes)); CHECK(hipMalloc(&B, 2 * plane_bytes)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(A, 1, 2 * plane_bytes)); CHECK(hipMemset(B, 2, 2 * plane_bytes));
  run<0, 2>(A, B, plane_bytes, K, ntm, ntn, sink, "LDS-DMA dwordx4, 2 stages (1 in flight)");
  run<1, 2>(A, B, plane_bytes, K, ntm, ntn, sink, "load dwordx4 -> VGPR -> ds_write, 2 sets");
  run<1, 3>(A, B, plane_bytes, K, ntm, ntn, sink, "load dwordx4 -> VGPR -> ds_write, 3 sets");
  run<1, 4>(A, B, plane_bytes, K, ntm, ntn, sink, "load dwordx4 -> VGPR -> ds_write, 4 sets");
  run<2, 2>(A, B, plane_bytes, K, ntm, ntn, sink, "load dwordx4 -> VGPR only, 2 sets");
  run<2, 4>(A, B, plane_bytes, K, ntm, ntn, sink, "load dwordx4 -> VGPR only, 4 sets");
  run<0, 2>(A, B, plane_bytes, K, ntm, ntn, sink, "PAIRED LDS-DMA dwordx4, 2 stages", 1);
  run<1, 2>(A, B, plane_bytes, K, ntm, ntn, sink, "PAIRED load dwordx4 -> VGPR -> ds_write, 2 sets", 1);
  run<1, 4>(A, B, plane_bytes, K, ntm, ntn, sink, "PAIRED load dwordx4 -> VGPR -> ds_write, 4 sets", 1);
  run<2, 4>(A, B, plane_bytes, K, ntm, ntn, sink, "PAIRED load dwordx4 -> VGPR only, 4 sets", 1);
  return 0;
}
