// Micro-benchmark: L2/HBM -> LDS stream rates for the two candidate staging shapes of the short-K 1x1 conv (256 -> 1024, M = 37 636).
//   V0: current streaming kernel's shape (per K-stage: 128 rows x 128 B of x at 512-B pitch + the same of W; 3-slot ring)
//   V1: W tile resident (64 KB, loaded once), x streamed as whole rows: RS rows x 512 B contiguous per stage, D slots
//   V2: V1 without sharing (every workgroup reads its own slice of x once: pure HBM -> LDS)
// build: hipcc --offload-arch=gfx950 -O3 -o ldsbench ldsbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int N> __device__ __forceinline__ void wait_vmcnt() {
#define C(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  C(1); C(2); C(3); C(4); C(5); C(6); C(7); C(8); C(9); C(10); C(11); C(12); C(13); C(14); C(15); C(16); C(20); C(24);
#undef C
}
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

constexpr int NC = 512;
struct Args { const char* x; const char* w; int M, ntm, ntn; unsigned* sink; };

__global__ __launch_bounds__(NC) void v0_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int G = gridDim.x, nwg = a.ntm * a.ntn;
  const int my_n = (nwg - (int)blockIdx.x + G - 1) / G;
  const int tile0 = xcd_remap(blockIdx.x, nwg), tstep = G >> 3;
  const int nt = tile0 % a.ntn;
  const int S = my_n * 4;
  int ii = 0, ikc = 0;
  auto issue = [&](int slot) {
    const int mt = (tile0 + ii * tstep) / a.ntn;
    char* sb = smem + slot * 32768;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int m = mt * 128 + q * 64 + (tid >> 3); if (m >= a.M) m = a.M - 1;
      __builtin_amdgcn_global_load_lds(GPTR(a.x + (size_t)m * 512 + ikc * 128 + (tid & 7) * 16), LPTR(sb + (q * NC + wave * 64) * 16), 16, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int n = nt * 128 + q * 64 + (tid >> 3);
      __builtin_amdgcn_global_load_lds(GPTR(a.w + (size_t)n * 512 + ikc * 128 + (tid & 7) * 16), LPTR(sb + 16384 + (q * NC + wave * 64) * 16), 16, 0, 0);
    }
    if (++ikc == 4) { ikc = 0; ++ii; }
  };
  if (S > 0) issue(0);
  if (S > 1) issue(1);
  for (int g = 0; g < S; ++g) {
    if (g + 1 < S) wait_vmcnt<4>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (g + 2 < S) issue((g + 2) % 3);
  }
  if (tid == 0 && a.sink) a.sink[blockIdx.x] = *(unsigned*)smem;
}

// W resident + whole-row x stages.  SHARE: tile mapping of the streaming kernel (8 workgroups of an XCD read the same rows).
template <int RS, int D, bool SHARE, bool LOADW>
__global__ __launch_bounds__(NC) void v1_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int SB = RS * 512, PT = SB / 16 / NC;       // stage bytes, pieces per thread
  const int tid = threadIdx.x, wave = tid >> 6;
  const int G = gridDim.x;
  char* ring = smem + 65536;
  int my_n, tile0, tstep, ntn;
  if (SHARE) { const int nwg = a.ntm * a.ntn; my_n = (nwg - (int)blockIdx.x + G - 1) / G; tile0 = xcd_remap(blockIdx.x, nwg); tstep = G >> 3; ntn = a.ntn; }
  else { my_n = (a.ntm - (int)blockIdx.x + G - 1) / G; tile0 = blockIdx.x; tstep = G; ntn = 1; }
  const int nt = tile0 % ntn;
  constexpr int SPT = 128 / RS;                          // stages per 128-row tile
  const int S = my_n * SPT;
  if (LOADW) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      __builtin_amdgcn_global_load_lds(GPTR(a.w + (size_t)nt * 65536 + (q * NC + tid) * 16), LPTR(smem + (q * NC + wave * 64) * 16), 16, 0, 0);
  }
  int ii = 0, is = 0;
  auto issue = [&](int slot) {
    const int mt = (tile0 + ii * tstep) / ntn;
    const size_t row0 = (size_t)mt * 128 + is * RS;
    char* sb = ring + slot * SB;
#pragma unroll
    for (int q = 0; q < PT; ++q) {
      size_t off = row0 * 512 + (size_t)(q * NC + tid) * 16;
      if (off >= (size_t)a.M * 512) off = 0;
      __builtin_amdgcn_global_load_lds(GPTR(a.x + off), LPTR(sb + (q * NC + wave * 64) * 16), 16, 0, 0);
    }
    if (++is == SPT) { is = 0; ++ii; }
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) if (s < S) issue(s);
  int slot_c = 0, slot_i = D - 1;
  for (int g = 0; g < S; ++g) {
    if (g + D - 2 < S) wait_vmcnt<PT * (D - 2)>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (g + D - 1 < S) issue(slot_i);
    if (++slot_i == D) slot_i = 0;
    if (++slot_c == D) slot_c = 0;
  }
  if (tid == 0 && a.sink) a.sink[blockIdx.x] = *(unsigned*)smem;
}

template <typename F> float time_it(F&& launch, int nsets) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> r;
  for (int rnd = 0; rnd < 5; ++rnd) {
    for (int s = 0; s < nsets; ++s) launch(s);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 4; ++rep) for (int s = 0; s < nsets; ++s) launch(s);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); r.push_back(ms * 1e3f / (4 * nsets));
  }
  std::sort(r.begin(), r.end());
  return r[r.size() / 2];
}

int main() {
  const int M = 37636, ntm = (M + 127) / 128, ntn = 8, NS = 6;
  std::vector<char*> xs(NS); char* w;
  for (auto& p : xs) { hipMalloc(&p, (size_t)(M + 256) * 512); hipMemset(p, 1, (size_t)(M + 256) * 512); }
  hipMalloc(&w, 1024 * 512); hipMemset(w, 1, 1024 * 512);
  auto args = [&](int s) { Args a; a.x = xs[s]; a.w = w; a.M = M; a.ntm = ntm; a.ntn = ntn; a.sink = nullptr; return a; };
  const double xb = (double)M * 512;
#define ATTR(k, bytes) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, bytes)
  { ATTR(v0_kernel, 98304);
    float us = time_it([&](int s) { v0_kernel<<<256, NC, 98304>>>(args(s)); }, NS);
    printf("V0 strided x+W, 3x32KB ring          : %7.1f us  L2->LDS %.2f TB/s\n", us, ntm * ntn * 131072.0 / us / 1e6); }
#define RUN1(RS, D, SHARE, LOADW, label) { auto k = v1_kernel<RS, D, SHARE, LOADW>; const int lds = 65536 + RS * 512 * D; ATTR(k, lds); \
    float us = time_it([&](int s) { k<<<256, NC, lds>>>(args(s)); }, NS); \
    const double bytes = (SHARE ? ntn : 1) * xb; \
    printf("%-38s: %7.1f us  ->LDS %.2f TB/s  (%d KB LDS)\n", label, us, bytes / us / 1e6, lds / 1024); }
  RUN1(32, 3, true, true, "V1 W-resident, x rows 16KB x3 shared");
  RUN1(32, 4, true, true, "V1 W-resident, x rows 16KB x4 shared");
  RUN1(32, 6, true, true, "V1 W-resident, x rows 16KB x6 shared");
  RUN1(64, 3, true, true, "V1 W-resident, x rows 32KB x3 shared");
  RUN1(16, 6, true, true, "V1 W-resident, x rows 8KB x6 shared");
  RUN1(16, 10, true, true, "V1 W-resident, x rows 8KB x10 shared");
  RUN1(32, 4, false, false, "V2 x once (HBM), 16KB x4");
  RUN1(32, 6, false, false, "V2 x once (HBM), 16KB x6");
  hipError_t e = hipDeviceSynchronize();
  printf("status %s\n", hipGetErrorString(e));
  return 0;
}
