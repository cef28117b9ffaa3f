// Same-wave interleave: N independent VALU ops after each MFMA (2 waves per SIMD, 8 waves per CU): does the VALU hide in the MFMA shadow?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NV, int KIND>   // KIND 0: v_fma_f32, 1: v_pk_fma_f32, 2: v_lshl_add_u32, 3: v_cvt_pk_bf16_f32
__global__ __launch_bounds__(512) void mix(float* out, int iters) {
  bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, 1};
  f32x4 acc[4] = {};
  float s[8]; for (int e = 0; e < 8; ++e) s[e] = threadIdx.x + e;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 sp[4]; for (int e = 0; e < 4; ++e) sp[e] = (f32x2){(float)threadIdx.x, (float)e};
  unsigned u[8]; for (int e = 0; e < 8; ++e) u[e] = threadIdx.x + e;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          const int e = (q * NV + n) & 7;
          if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[e]) : "v"(1.0001f));
          if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(sp[e & 3]) : "v"((f32x2){1.0001f, 0.999f}));
          if (KIND == 2) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[e]) : "v"(0x1234567u));
          if (KIND == 3) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u[e]) : "v"(s[e]), "v"(s[(e + 1) & 7]));
        }
      }
  }
  float t = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
  for (int e = 0; e < 8; ++e) t += s[e] + (float)u[e];
  for (int e = 0; e < 4; ++e) t += sp[e][0] + sp[e][1];
  out[blockIdx.x * 512 + threadIdx.x] = t;
}
template <int NV, int KIND> float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix<NV, KIND><<<256, 512>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); mix<NV, KIND><<<256, 512>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  const int it = 2000;
  const char* names[4] = {"v_fma_f32", "v_pk_fma_f32", "v_lshl_add_u32", "v_cvt_pk_bf16_f32"};
  printf("MFMA only: %.1f us\n", run<0, 0>(out, it));
#define ROW(K) printf("%-18s per MFMA: 1 -> %.1f  2 -> %.1f  3 -> %.1f  4 -> %.1f  6 -> %.1f us\n", names[K], run<1, K>(out, it), run<2, K>(out, it), run<3, K>(out, it), run<4, K>(out, it), run<6, K>(out, it));
  ROW(0); ROW(1); ROW(2); ROW(3);
  return 0;
}
