// Do VALU work of one wave and MFMA work of other waves on the SAME SIMD overlap?  12 waves per workgroup (3 per SIMD): waves 0-7 run a
// dependent-chain MFMA loop (4 accumulators), waves 8-11 a packed-f32 VALU loop.  Times: MFMA only, VALU only, both.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
template <int MODE, int MF, int VK = 0>   // VK: 0 = v_pk_fma_f32, 1 = v_fma_f32, 2 = integer and/shift/add, 3 = ds_read_b128 stream; MODE bit 0: MFMA waves active, bit 1: VALU waves active;  MF: 0 = 16x16x32, 1 = 32x32x16
__global__ __launch_bounds__(768) void mix(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  if (wave < 8) {
    if (!(MODE & 1)) return;
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    if (MF == 0) {
      f32x4 acc[4] = {};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
      }
      out[blockIdx.x * 768 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    } else {
      typedef __attribute__((ext_vector_type(16))) float f32x16;
      f32x16 acc[2] = {};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int q = 0; q < 2; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
      }
      out[blockIdx.x * 768 + threadIdx.x] = acc[0][0] + acc[1][1];
    }
  } else {
    if (!(MODE & 2)) return;
    if (VK == 0) {
    f32x2 s[8];
    for (int e = 0; e < 8; ++e) s[e] = (f32x2){(float)threadIdx.x, (float)e};
    const f32x2 c = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)            // 128 independent-ish packed FMAs per iteration (8 chains)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = s[e] * c + c;
    }
    float t = 0; for (int e = 0; e < 8; ++e) t += s[e][0] + s[e][1];
    out[blockIdx.x * 768 + threadIdx.x] = t;
    } else if (VK == 1) {
    float s[8];
    for (int e = 0; e < 8; ++e) s[e] = (float)threadIdx.x + e;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[e]) : "v"(1.0001f));
    }
    float t = 0; for (int e = 0; e < 8; ++e) t += s[e];
    out[blockIdx.x * 768 + threadIdx.x] = t;
    } else if (VK == 2) {
    unsigned s[8];
    for (int e = 0; e < 8; ++e) s[e] = threadIdx.x + e;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int e = 0; e < 8; ++e) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(s[e]) : "v"(0x1234567u));
    }
    unsigned t = 0; for (int e = 0; e < 8; ++e) t += s[e];
    out[blockIdx.x * 768 + threadIdx.x] = (float)t;
    } else {
    __shared__ float4 buf[1024];
    buf[threadIdx.x] = make_float4(1, 2, 3, 4);
    float4 acc4 = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int r = 0; r < 128; ++r) { float4 v = buf[(threadIdx.x + r * 17) & 1023]; acc4.x += v.x; }
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc4.x;
    }
  }
}
template <int MODE, int MF, int VK = 0> float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  mix<MODE, MF, VK><<<256, 768>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); mix<MODE, MF, VK><<<256, 768>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f;
}
int main() {
  float* out; hipMalloc(&out, 256 * 768 * 4);
  const int it = 2000;
  printf("16x16x32: 32 MFMA/iter/wave, 2 MFMA waves per SIMD; VALU wave: 128 v_pk_fma_f32 per iter\n");
  float a = run<1, 0>(out, it), b = run<2, 0>(out, it), c = run<3, 0>(out, it);
  printf("  MFMA only %.1f us (%.1f cycles@2.4GHz per MFMA per SIMD)   VALU only %.1f us (%.2f cyc/instr)   both %.1f us\n", a, a * 2400 / (it * 64.0), b, b * 2400 / (it * 128.0), c);
  a = run<1, 1>(out, it); c = run<3, 1>(out, it);
  printf("32x32x16: 16 MFMA/iter/wave: MFMA only %.1f us (%.1f cycles per MFMA per SIMD)  both %.1f us\n", a, a * 2400 / (it * 32.0), c);
  { float b1 = run<2, 0, 1>(out, it), c1 = run<3, 0, 1>(out, it); printf("v_fma_f32      : VALU only %.1f us (%.2f cyc/instr)  both with 16x16x32 %.1f us\n", b1, b1 * 2400 / (it * 128.0), c1); }
  { float b1 = run<2, 0, 2>(out, it), c1 = run<3, 0, 2>(out, it); printf("v_lshl_add_u32 : VALU only %.1f us (%.2f cyc/instr)  both with 16x16x32 %.1f us\n", b1, b1 * 2400 / (it * 128.0), c1); }
  { float b1 = run<2, 0, 3>(out, it), c1 = run<3, 0, 3>(out, it); printf("ds_read_b128+add: only %.1f us  both with 16x16x32 %.1f us\n", b1, c1); }
  return 0;
}
