// Issue cost of staging 1 KB pieces beside MFMAs (2 waves per SIMD, 8 waves per CU, L2-resident source):
//   A: global_load_lds_dwordx4 (LDS-DMA)     B: global_load_dwordx4 -> registers -> ds_write_b128 one iteration later
// per iteration of 32 MFMAs (16x16x32 bf16) per wave: P pieces.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int P, int KIND>
__global__ __launch_bounds__(512) void mix(const char* src, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  f32x4 acc[4] = {};
  const int wave = threadIdx.x >> 6;
  const char* g = src + (size_t)blockIdx.x * 65536 + threadIdx.x * 16;
  uint4 r[P > 0 ? P : 1];
  for (int i = 0; i < iters; ++i) {
    if (KIND == 1 && i > 0) {
#pragma unroll
      for (int p = 0; p < P; ++p) *(uint4*)(smem + (p * 512 + threadIdx.x) * 16) = r[p];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const char* s = g + ((i * P + p) & 7) * 8192;
      if (KIND == 0) __builtin_amdgcn_global_load_lds(GPTR(s), LPTR(smem + (p * 512 + wave * 64) * 16), 16, 0, 0);
      else r[p] = *(const uint4*)s;
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
    if (KIND == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  float t = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + *(float*)(smem + threadIdx.x * 4);
  if (KIND == 1) t += __uint_as_float(r[0].x);
  out[blockIdx.x * 512 + threadIdx.x] = t;
}
template <int P, int KIND> float run(const char* src, float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)mix<P, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  mix<P, KIND><<<256, 512, 65536>>>(src, out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); mix<P, KIND><<<256, 512, 65536>>>(src, out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  char* src; float* out; hipMalloc(&src, 256 * 65536 + 65536); hipMemset(src, 0, 256 * 65536 + 65536); hipMalloc(&out, 256 * 512 * 4);
  const int it = 2000;
  float base = run<0, 0>(src, out, it);
  printf("MFMA only: %.1f us (%.0f cycles@2.1GHz per iteration per wave pair)\n", base, base * 2100 / it);
  printf("LDS-DMA        pieces/iter/wave: 2 -> %.1f   4 -> %.1f   6 -> %.1f us\n", run<2, 0>(src, out, it), run<4, 0>(src, out, it), run<6, 0>(src, out, it));
  printf("load+ds_write  pieces/iter/wave: 2 -> %.1f   4 -> %.1f   6 -> %.1f us\n", run<2, 1>(src, out, it), run<4, 1>(src, out, it), run<6, 1>(src, out, it));
  return 0;
}
