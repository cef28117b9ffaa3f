// Micro-benchmark: what does ONE 1-KB LDS-DMA piece (global_load_lds_dwordx4) cost a SIMD that is busy with MFMAs?
// Round 3 measured on the product conv kernel (3x3 256 -> 256, M = 37 636, us): compute only 35, loads only 32, both 50 -- the loads are
// neither hidden nor is their rate the bound.  Here: 255 workgroups x 8 waves (2 per SIMD), per iteration ("K stage") every wave issues
// 40 v_mfma_f32_16x16x32_bf16 (640 cycles; 1 280 per SIMD) and P pieces into a 3-slot ring with the product kernel's counted vmcnt and
// ONE raw barrier per stage; R fragment reads (ds_read_b128) per wave and stage on top.  Address patterns of a piece (64 lanes x 16 B):
//   contig : 1 KB contiguous                                     (fragment-ordered / stage-major operand)
//   rows512: 8 rows x 128 B at 512-B pitch                       (pixel operand, Cin = 256)
//   rows4k6: 8 rows x 128 B at 4 608-B pitch                     (K-contiguous weight rows of a 3x3 256 layer)
// Source is L2-resident (2 MB window per pattern, shared by all workgroups -> hot in every XCD's L2): this isolates ISSUE cost from
// fill bandwidth.  Output: us per 288 stages and cycles per stage (at the clock implied by the MFMA-only run).
// build: hipcc --offload-arch=gfx950 -O3 -o piecebench piecebench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int N> __device__ __forceinline__ void wait_vmcnt() {
#define C(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C(1); C(2); C(3); C(4); C(5); C(6); C(7); C(8); C(9); C(10); C(11); C(12); C(13); C(14); C(15); C(16);
#undef C
}

struct Args { const char* src; unsigned* sink; int nstage; };
constexpr int NT = 512, SLOT = 16 * 1024 * 3;      // up to 16 pieces x 8 waves... sized per P below

// P pieces per wave and stage, PAT address pattern, R ds_read_b128 per wave and stage, STAG = 1: waves 4-7 multiply first and issue afterwards
template <int P, int PAT, int R, int STAG>
__global__ __launch_bounds__(NT, 2) void piece_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE = (P > 0 ? P : 1) * 8 * 1024;          // bytes per stage: P pieces x 8 waves x 1 KB
  int ld = 0;
  auto issue = [&](int slot) {
    char* sb = smem + slot * STAGE;
#pragma unroll
    for (int i = 0; i < P; ++i) {
      const char* s;
      const int pat = PAT <= 2 ? PAT : (i < 2 ? 1 : (PAT == 3 ? 2 : 0));     // PAT 3: 2 pixel pieces + K-contiguous weight rows (product); 4: + contiguous weights
      if (pat == 0) {                      // 1 KB contiguous; 16 stages x 128 KB = 2 MB window
        s = a.src + (size_t)(((ld & 15) * 128 + wave * 16 + i) * 1024 + lane * 16);
      } else if (pat == 1) {               // 8 rows x 128 B at 512-B pitch; 2 MB window
        const int r = (((ld & 3) * 128 + wave * 16 + i) * 8 + (lane >> 3));
        s = a.src + (size_t)r * 512 + (size_t)((ld >> 2) & 3) * 128 + (size_t)(lane & 7) * 16;
      } else {                             // 8 rows x 128 B at 4 608-B pitch: the [256][2304 x 2 B] weight matrix of a 3x3 256 layer
        const int r = ((wave * 8 + i) & 31) * 8 + (lane >> 3);
        s = a.src + (size_t)(4u << 20) + (size_t)r * 4608 + (size_t)(ld % 36) * 128 + (size_t)(lane & 7) * 16;
      }
      __builtin_amdgcn_global_load_lds(GPTR(s), LPTR(sb + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
    ++ld;
  };
  f32x4 acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fx[10], fw[8];
#pragma unroll
  for (int i = 0; i < 10; ++i) fx[i] = (bf16x8){1, 2, 3, 4, 5, 6, 7, (short)lane};
#pragma unroll
  for (int i = 0; i < 8; ++i) fw[i] = (bf16x8){8, 7, 6, 5, 4, 3, 2, (short)(lane + i)};
  auto mma = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 5; ++i)
          acc[j * 5 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s * 4 + j], fx[s * 5 + i], acc[j * 5 + i], 0, 0, 0);
  };
  auto reads = [&](int slot) {
    if constexpr (R > 0) {
      // the product kernel's conflict-free fragment addressing: row = 16 i + (lane & 15), 16-B chunk (4 s + lane / 16) ^ ((row >> 1) & 7)
      const int sw = (lane >> 1) & 7, kq = lane >> 4;
      const char* px = smem + (P > 0 ? slot * STAGE : 0) + (lane & 15) * 128;
      constexpr int NROWBLK = (STAGE / 2048) > 0 ? (STAGE / 2048) : 1;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int s = i & 1, blk = (i >> 1) % NROWBLK;
        const bf16x8 v = *(const bf16x8*)(px + blk * 2048 + (((4 * s + kq) ^ sw) << 4));
        if (i < 10) fx[i] = v; else fw[i - 10] = v;
      }
    }
  };
  const int nk = a.nstage;
  if (P > 0) { issue(0); issue(1); }
  int buf = 0;
  const bool late = STAG && wave >= 4;
  for (int kt = 0; kt < nk; ++kt) {
    if (P > 0) { if (kt + 1 < nk) wait_vmcnt<P>(); else wait_vmcnt<0>(); }
    if (late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (!late) {
      reads(buf);
      if (P > 0 && kt + 2 < nk) issue(buf >= 1 ? buf - 1 : 2);
      mma();
    } else {
      mma();
      if (P > 0 && kt + 2 < nk) issue(buf >= 1 ? buf - 1 : 2);
      reads(buf);
    }
    buf = (buf + 1 == 3) ? 0 : buf + 1;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
  if (a.sink && s == 123.456f) a.sink[blockIdx.x] = 1;
}

template <int P, int PAT, int R, int STAG>
static float run(const Args& a) {
  const int lds = 3 * (P > 0 ? P : 1) * 8 * 1024 > 65536 ? 3 * P * 8 * 1024 : 65536;
  hipFuncSetAttribute((const void*)piece_kernel<P, PAT, R, STAG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 9; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((piece_kernel<P, PAT, R, STAG>), dim3(255), dim3(NT), lds, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}


// Role split: waves 0-3 (one per SIMD) only multiply -- MF v_mfma per stage (80 = the whole stage's matrix work of a SIMD) and R fragment
// reads -- waves 4-7 only issue LDS-DMA pieces, P each per stage (13 = all 52 pieces of the product stage).  Does a piece issued by ANOTHER
// wave of the SIMD cost the multiplying wave anything?
template <int P, int PAT, int R, int MF>
__global__ __launch_bounds__(NT, 2) void role_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int STAGE = (P > 0 ? P : 1) * 4 * 1024;
  const int nk = a.nstage;
  if (wave >= 4) {
    const int lw = wave - 4;
    int ld = 0;
    auto issue = [&](int slot) {
      char* sb = smem + slot * STAGE;
#pragma unroll
      for (int i = 0; i < P; ++i) {
        const char* s;
        const int pat = PAT <= 2 ? PAT : (i < 5 ? 1 : 2);      // PAT 3: the product stage: 5 pixel pieces + 8 weight pieces per loader wave
        if (pat == 0) s = a.src + (size_t)(((ld & 15) * 64 + lw * 16 + i) * 1024 + lane * 16);
        else if (pat == 1) { const int r = (((ld & 3) * 64 + lw * 16 + i) * 8 + (lane >> 3)); s = a.src + (size_t)r * 512 + (size_t)((ld >> 2) & 3) * 128 + (size_t)(lane & 7) * 16; }
        else { const int r = ((lw * 8 + i) & 31) * 8 + (lane >> 3); s = a.src + (size_t)(4u << 20) + (size_t)r * 4608 + (size_t)(ld % 36) * 128 + (size_t)(lane & 7) * 16; }
        __builtin_amdgcn_global_load_lds(GPTR(s), LPTR(sb + (i * 256 + lw * 64) * 16), 16, 0, 0);
      }
      ++ld;
    };
    if (P > 0) { issue(0); issue(1); }
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (P > 0) { if (kt + 1 < nk) wait_vmcnt<(P > 16 ? 16 : P)>(); else wait_vmcnt<0>(); }
      __builtin_amdgcn_s_barrier();
      if (P > 0 && kt + 2 < nk) issue(buf >= 1 ? buf - 1 : 2);
      buf = (buf + 1 == 3) ? 0 : buf + 1;
    }
    return;
  }
  f32x4 acc[20];
#pragma unroll
  for (int i = 0; i < 20; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fx[10], fw[8];
#pragma unroll
  for (int i = 0; i < 10; ++i) fx[i] = (bf16x8){1, 2, 3, 4, 5, 6, 7, (short)lane};
#pragma unroll
  for (int i = 0; i < 8; ++i) fw[i] = (bf16x8){8, 7, 6, 5, 4, 3, 2, (short)(lane + i)};
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (R > 0) {
      const int sw = (lane >> 1) & 7, kq = lane >> 4;
      const char* px = smem + (P > 0 ? buf * STAGE : 0) + (lane & 15) * 128;
      constexpr int NROWBLK = (STAGE / 2048) > 0 ? (STAGE / 2048) : 1;
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const int s = i & 1, blk = (i >> 1) % NROWBLK;
        const bf16x8 v = *(const bf16x8*)(px + blk * 2048 + (((4 * s + kq) ^ sw) << 4));
        if (i < 10) fx[i] = v; else fw[(i - 10) & 7] = v;
      }
    }
#pragma unroll
    for (int rep = 0; rep < MF / 40; ++rep)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 5; ++i)
            acc[j * 5 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[s * 4 + j], fx[s * 5 + i], acc[j * 5 + i], 0, 0, 0);
    buf = (buf + 1 == 3) ? 0 : buf + 1;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 20; ++i) s += acc[i][0] + acc[i][3];
  if (a.sink && s == 123.456f) a.sink[blockIdx.x] = 1;
}

template <int P, int PAT, int R, int MF>
static float run_role(const Args& a) {
  const int lds = 3 * (P > 0 ? P : 1) * 4 * 1024 > 65536 ? 3 * P * 4 * 1024 : 65536;
  hipFuncSetAttribute((const void*)role_kernel<P, PAT, R, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ts;
  for (int r = 0; r < 9; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((role_kernel<P, PAT, R, MF>), dim3(255), dim3(NT), lds, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 2) ts.push_back(ms * 1e3f);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main() {
  char* src; unsigned* sink;
  const size_t bytes = 64u << 20;
  hipMalloc(&src, bytes); hipMalloc(&sink, 4096);
  std::vector<unsigned short> h(bytes / 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
  hipMemcpy(src, h.data(), bytes, hipMemcpyHostToDevice);
  Args a{src, sink, 288};
  const float base = run<0, 0, 0, 1>(a);
  printf("MFMA only (40 per wave and stage, 2 waves per SIMD), us per 288 stages: early/late halves %.1f   all waves alike %.1f  (1 280 pipe cycles per stage -> %.2f GHz if pipe-bound)\n",
         base, run<0, 0, 0, 0>(a), 288.0 * 1280.0 / base / 1e3);
  printf("+ fragment reads only (conflict-free): R=10 %.1f   R=18 %.1f   | all waves alike: R=18 %.1f\n", run<0, 0, 10, 1>(a), run<0, 0, 18, 1>(a), run<0, 0, 18, 0>(a));
#define ROW(P) printf("P=%d pieces:  contig %.1f  rows512 %.1f  rows4k6 %.1f | +10 reads: contig %.1f rows512 %.1f rows4k6 %.1f | +18 reads: contig %.1f rows512 %.1f rows4k6 %.1f | alike +18: contig %.1f\n", P, \
                      run<P, 0, 0, 1>(a), run<P, 1, 0, 1>(a), run<P, 2, 0, 1>(a), run<P, 0, 10, 1>(a), run<P, 1, 10, 1>(a), run<P, 2, 10, 1>(a), \
                      run<P, 0, 18, 1>(a), run<P, 1, 18, 1>(a), run<P, 2, 18, 1>(a), run<P, 0, 18, 0>(a));
  ROW(2) ROW(4) ROW(6)
  printf("product mix, 6 pieces (2 pixel rows512 + 4 weight pieces) + 18 reads:  weights K-contiguous rows %.1f   weights contiguous (stage-major) %.1f\n",
         run<6, 3, 18, 1>(a), run<6, 4, 18, 1>(a));
  printf("(us per 288 stages)\n");
  // warm the clocks, then the role-split runs
  for (int i = 0; i < 3; ++i) run_role<0, 0, 0, 80>(a);
  printf("\nrole split (waves 0-3: 80 MFMA per stage = 1 280 pipe cycles; waves 4-7: P LDS-DMA pieces each), us per 288 stages:\n");
  printf("  MFMA only %.1f   + 36 fragment reads %.1f\n", run_role<0, 0, 0, 80>(a), run_role<0, 0, 36, 80>(a));
  printf("  P=4:  contig %.1f  rows512 %.1f  rows4k6 %.1f\n", run_role<4, 0, 0, 80>(a), run_role<4, 1, 0, 80>(a), run_role<4, 2, 0, 80>(a));
  printf("  P=8:  contig %.1f  rows512 %.1f  rows4k6 %.1f\n", run_role<8, 0, 0, 80>(a), run_role<8, 1, 0, 80>(a), run_role<8, 2, 0, 80>(a));
  printf("  P=13: contig %.1f  rows512 %.1f  rows4k6 %.1f  product mix %.1f\n", run_role<13, 0, 0, 80>(a), run_role<13, 1, 0, 80>(a), run_role<13, 2, 0, 80>(a), run_role<13, 3, 0, 80>(a));
  printf("  P=13 + 36 reads: contig %.1f  product mix %.1f     P=13 pieces, NO MFMA (loaders alone): contig %.1f  product mix %.1f\n",
         run_role<13, 0, 36, 80>(a), run_role<13, 3, 36, 80>(a), run_role<13, 0, 0, 0>(a), run_role<13, 3, 0, 0>(a));

  return 0;
}
