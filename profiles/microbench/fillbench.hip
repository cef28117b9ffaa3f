// Micro-benchmark: which path should the WEIGHT operand of the dominant conv (3x3 256 -> 256, M = 37 636; conv_igemm2_kernel<256,5,3>)
// take into the CU?  Per 64-deep K stage a workgroup (8 waves, one per CU, 255 of them) needs 160 rows x 128 B of pixels (its own:
// 20 KB, LDS-DMA) and 256 rows x 128 B of weights (the SAME 32 KB for every workgroup, L2-resident).  Round 2 measured the product
// kernel's loads-only floor at ~31 us of its 48 (both operands through `global_load_lds`, ~70 GB/s per CU).  Question:
// do fragment-ordered weights loaded straight into VGPRs (`global_load_dwordx4`, 1 KB contiguous per wave-instruction) ADD bandwidth to
// the LDS-DMA path or do they share its limit -- and what does loading them twice (the two pixel-halves of a 2 x 4 wave grid) cost?
//   mode 0  A (pixels) by LDS-DMA + B (weights) by LDS-DMA            = the product kernel's stage (52 KB)
//   mode 1  A by LDS-DMA only                                         (20 KB)
//   mode 2  B by LDS-DMA only                                         (32 KB)
//   mode 3  A by LDS-DMA + B to VGPRs, every wave its own 4 KB        (no duplication: 1 x 8 wave grid)            (52 KB)
//   mode 4  A by LDS-DMA + B to VGPRs, wave pairs load the same 8 KB  (2 x 4 wave grid: 64 KB of weight requests)  (84 KB)
//   mode 5  B to VGPRs only, 4 KB per wave                            (32 KB)
//   mode 6  B to VGPRs only, 8 KB per wave, pairs duplicate           (64 KB)
// MF = 1 adds the stage's 40 v_mfma_f32_16x16x32_bf16 per wave (operands: whatever is in registers) and, for the LDS-resident operands,
// the stage's ds_read_b128 fragment reads (18 / 20 / 10 per wave by mode), i.e. the whole stage except the epilogue.
// Stage structure as in the product kernel: 3-slot ring, counted vmcnt, ONE raw barrier per stage.
// Second table (layouts, loads only): the pixel operand as the conv really reads it (AL = 2: [M][256 ch] map of 19 MB, 512-B pixel
// pitch, stage = (tap, 64-channel chunk), tap-shifted rows: re-read 9 x out of L2) instead of a once-read stream (AL = 1), and the
// weight tile as the product layout has it (BL = 2: [256 couts][2304 x 2 B] K-contiguous rows, a stage = 256 lines at 4 608-B pitch)
// instead of stage-major contiguous (BL = 1).
// build: hipcc --offload-arch=gfx950 -O3 -o fillbench fillbench.hip ; ./fillbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
#define C(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C(1); C(2); C(3); C(4); C(5); C(6); C(7); C(8); C(9); C(10); C(11); C(12); C(13); C(14); C(15); C(16);
#undef C
}

struct Args { const char* x; const char* w; unsigned* sink; int nstage; };

constexpr int NT = 512, A_BYTES = 160 * 128, B_BYTES = 256 * 128, STAGE = A_BYTES + B_BYTES, NSLOT = 3;

__device__ __forceinline__ u32x4 gload(const char* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
  return r;
}

template <int MODE, int MF, int AL = 1, int BL = 1>
__global__ __launch_bounds__(NT, 2) void fill_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr bool A_DMA = MODE == 0 || MODE == 1 || MODE == 3 || MODE == 4;
  constexpr bool B_DMA = MODE == 0 || MODE == 2;
  constexpr int B_REG = (MODE == 3 || MODE == 5) ? 4 : (MODE == 4 || MODE == 6) ? 8 : 0;     // dwordx4 loads per wave and stage
  constexpr int A_IT = A_DMA ? 3 : 0, B_IT = B_DMA ? 4 : 0;
  const bool a_tail = wave < 4;                                           // 20 pieces over 8 waves: 3 for waves 0-3, 2 for 4-7
  // pixel operand: this workgroup's own 160 rows of a [M][2304 B] map, one 128-B column block per stage (tap/channel walk collapsed
  // into consecutive column blocks: same bytes, same 128-B-per-row access shape as the product kernel's stage)
  const char* xb = a.x + (size_t)blockIdx.x * 160 * 4608;
  const char* wb = a.w;                                                   // [36 stages][256 rows][128 B] = the packed weights, stage-major
  int ld = 0;
  u32x4 wreg[2][8];
  auto issue_dma = [&](int slot) {
    char* sb = smem + slot * STAGE;
    if (A_DMA) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (i == 2 && !a_tail) break;
        const int row = i * 64 + (tid >> 3);
        const char* src;
        if constexpr (AL == 1) {
          src = xb + (size_t)row * 4608 + (ld % 36) * 128 + (tid & 7) * 16;
        } else {
          // conv-faithful: pixel m = blockIdx*148 + row (+ tap shift of a dilation-2 3x3 on a 97-wide map), 512-B pixels, chunk kc
          const int st = ld % 36, tap = st >> 2, kc = st & 3;
          const int dy = (tap / 3 - 1) * 2, dx = (tap % 3 - 1) * 2;
          int m = (int)blockIdx.x * 148 + row + dy * 97 + dx;
          m = m < 0 ? 0 : (m > 37635 ? 37635 : m);
          src = a.x + (size_t)m * 512 + kc * 128 + (tid & 7) * 16;
        }
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
    if (B_DMA) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const char* src = BL == 1 ? wb + (size_t)(ld % 36) * B_BYTES + (i * NT + tid) * 16                                  // stage-major
                                  : wb + (size_t)(i * 64 + (tid >> 3)) * 4608 + (ld % 36) * 128 + (tid & 7) * 16;           // K-contiguous rows
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sb + A_BYTES + (i * NT + wave * 64) * 16), 16, 0, 0);
      }
    }
  };
  auto issue_reg = [&](int set, int stage) {      // `set` is a compile-time constant at every (inlined) call site
    if constexpr (B_REG > 0) {
      // fragment-ordered weights: 1 KB contiguous per wave-instruction.  B_REG = 4: wave w owns KB [4w, 4w+4) of the stage's 32;
      // B_REG = 8: waves w and w+4 both load KB [8 (w&3), 8 (w&3) + 8)
      const int kb0 = B_REG == 4 ? wave * 4 : (wave & 3) * 8;
#pragma unroll
      for (int j = 0; j < B_REG; ++j) wreg[set][j] = gload(wb + (size_t)(stage % 36) * B_BYTES + (kb0 + j) * 1024 + lane * 16);
    }
  };
  f32x4 acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fx[20], fw[8];
#pragma unroll
  for (int i = 0; i < 20; ++i) fx[i] = (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
  for (int i = 0; i < 8; ++i) fw[i] = (bf16x8){1, 2, 3, 4, 5, 6, 7, 8};

  const int nk = a.nstage;
  // prologue: registers for stage 0, DMA for stages 0 and 1
  issue_reg(0, 0);
  issue_dma(0); ld = 1;
  issue_dma(1); ld = 2;
  int buf = 0;
  auto step = [&](int kt, auto PAR) {
    constexpr int P = decltype(PAR)::value;              // parity of kt: selects the register set at compile time (no scratch)
    // need: DMA(kt) landed (+ registers of stage kt); may stay in flight: DMA(kt+1)
    const bool tail = (kt + 1 >= nk);
    if (tail) wait_vmcnt<0>();
    else if (A_DMA && !a_tail) wait_vmcnt<A_IT - 1 + B_IT>();
    else wait_vmcnt<A_IT + B_IT>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // issue order per stage: register loads of stage kt+1 FIRST, then the DMA pieces of stage kt+2 (the counted wait above then leaves
    // exactly one stage of DMA pieces in flight and covers the register loads issued before them)
    if (kt + 1 < nk) issue_reg(1 - P, kt + 1);
    if (kt + 2 < nk) { issue_dma(buf >= 1 ? buf - 1 : NSLOT - 1); ++ld; }
    if constexpr (MF) {
      const char* px = smem + buf * STAGE + (lane & 15) * 128 + ((lane >> 4) << 4);
      constexpr int NX = (MODE == 3 || MODE == 5) ? 20 : 10;         // 1 x 8 wave grid reads all 160 rows, 2 x 4 reads 80
      constexpr int NW = B_DMA ? 8 : 0;
#pragma unroll
      for (int i = 0; i < NX; ++i) fx[i] = *(const bf16x8*)(px + (i % 10) * 2048 + (i / 10) * 64);
#pragma unroll
      for (int i = 0; i < NW; ++i) fw[i] = *(const bf16x8*)(px + A_BYTES + i * 2048);
      if constexpr (B_REG > 0) {
#pragma unroll
        for (int i = 0; i < B_REG; ++i) fw[i] = __builtin_bit_cast(bf16x8, wreg[P][i]);
      }
      if constexpr (NX == 20) {          // 1 x 8 wave grid: 160 pixels x 32 couts per wave = 10 x 2 accumulators, 2 K-halves
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 10; ++i)
              acc[j * 10 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s * 2 + j], fx[s * 10 + i], acc[j * 10 + i], 0, 0, 0);
      } else {                           // 2 x 4 wave grid: 80 pixels x 64 couts per wave = 5 x 4 accumulators
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 5; ++i)
              acc[j * 5 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s * 4 + j], fx[s * 5 + i], acc[j * 5 + i], 0, 0, 0);
      }
    } else if constexpr (B_REG > 0) {
#pragma unroll
      for (int j = 0; j < B_REG; ++j) asm volatile("" ::"v"(wreg[P][j]));
    }
    buf = (buf + 1 == NSLOT) ? 0 : buf + 1;
  };
  for (int kt = 0; kt < nk; kt += 2) {
    step(kt, std::integral_constant<int, 0>{});
    if (kt + 1 < nk) step(kt + 1, std::integral_constant<int, 1>{});
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
  if (a.sink && s == 123.456f) a.sink[blockIdx.x] = 1;
}

template <int MODE, int MF, int AL = 1, int BL = 1>
static float run(const Args& a, int grid, int reps) {
  hipFuncSetAttribute((const void*)fill_kernel<MODE, MF, AL, BL>, hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * STAGE);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < reps + 2; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<MODE, MF, AL, BL>), dim3(grid), dim3(NT), NSLOT * STAGE, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  const int grid = 255, nstage = 36 * 8;                    // 8 convs' worth of stages per launch: amortises launch + prologue
  const size_t xbytes = (size_t)grid * 160 * 4608, wbytes = (size_t)36 * B_BYTES;
  char *x, *w; unsigned* sink;
  hipMalloc(&x, xbytes); hipMalloc(&w, wbytes); hipMalloc(&sink, 4096);
  std::vector<unsigned short> h(xbytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(x, h.data(), xbytes, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), wbytes, hipMemcpyHostToDevice);
  Args a{x, w, sink, nstage};
  const char* names[7] = {"A dma + B dma (product stage)", "A dma only", "B dma only", "A dma + B vgpr 4KB/wave", "A dma + B vgpr 8KB/wave (dup)",
                          "B vgpr 4KB/wave only", "B vgpr 8KB/wave only (dup)"};
  const double kb[7] = {52, 20, 32, 52, 84, 32, 64};
  float t[2][7];
#define RUN(M) t[0][M] = run<M, 0>(a, grid, 9); t[1][M] = run<M, 1>(a, grid, 9);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  printf("%-34s %10s %12s %12s | %10s %12s\n", "mode", "loads us", "us/36 stg", "GB/s per CU", "+MFMA us", "us/36 stg");
  for (int m = 0; m < 7; ++m) {
    const double per36 = t[0][m] / 8.0, per36m = t[1][m] / 8.0;
    printf("%-34s %10.1f %12.2f %12.1f | %10.1f %12.2f\n", names[m], t[0][m], per36, kb[m] * 1024 * 36 / (per36 * 1e-6) / 1e9, t[1][m], per36m);
  }
  printf("MFMA floor: 36 stages x 1280 cycles per SIMD = 19.2 us at 2.4 GHz\n");
  printf("\nlayouts, loads only (us per 36 stages):\n");
  printf("  A only:  once-read stream %.2f   conv-faithful (19 MB map, 9 taps from L2) %.2f\n", run<1, 0, 1, 1>(a, grid, 9) / 8, run<1, 0, 2, 1>(a, grid, 9) / 8);
  printf("  B only:  stage-major contiguous %.2f   K-contiguous rows (4608-B pitch) %.2f\n", run<2, 0, 1, 1>(a, grid, 9) / 8, run<2, 0, 1, 2>(a, grid, 9) / 8);
  printf("  A conv-faithful + B stage-major %.2f   A conv-faithful + B K-contiguous rows (= product) %.2f\n", run<0, 0, 2, 1>(a, grid, 9) / 8,
         run<0, 0, 2, 2>(a, grid, 9) / 8);
  printf("  A conv-faithful + B vgpr 4KB/wave %.2f   + B vgpr 8KB/wave dup %.2f\n", run<3, 0, 2, 1>(a, grid, 9) / 8, run<4, 0, 2, 1>(a, grid, 9) / 8);
  printf("  with MFMA + fragment reads (un-staggered waves): product %.2f   B stage-major %.2f   B vgpr dup %.2f   B vgpr 4KB %.2f\n",
         run<0, 1, 2, 2>(a, grid, 9) / 8, run<0, 1, 2, 1>(a, grid, 9) / 8, run<4, 1, 2, 1>(a, grid, 9) / 8, run<3, 1, 2, 1>(a, grid, 9) / 8);
  return 0;
}
