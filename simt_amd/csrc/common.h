// Shared device helpers for the SimT gfx950 kernels.
// Everything here is written for CDNA4 (wave64, MFMA, 160 KB LDS); there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/simt_hip.h"

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4v;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_t, b);
}

// two floats -> one dword of two bf16 (lo = a, hi = b): ONE v_cvt_pk_bf16_f32 (the scalar casts combined with shifts cost five)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 8 consecutive elements <-> 8 floats (16 B for bf16, 32 B for f32)
__device__ __forceinline__ void load8(const float* p, float* v) {
  float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
  uint4 r = *(const uint4*)p;
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
  v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}
// 8 elements as loaded (16 / 32 bytes), unpacked later: batches of loads can be in flight without holding 8 floats each
template <typename T> struct Raw8;
template <> struct Raw8<float> { float4 a, b; };
template <> struct Raw8<bf16_t> { uint4 r; };
__device__ __forceinline__ void raw8_load(const float* p, Raw8<float>& q) { q.a = *(const float4*)p; q.b = *(const float4*)(p + 4); }
__device__ __forceinline__ void raw8_load(const bf16_t* p, Raw8<bf16_t>& q) { q.r = *(const uint4*)p; }
__device__ __forceinline__ void raw8_unpack(const Raw8<float>& q, float* v) {
  v[0] = q.a.x; v[1] = q.a.y; v[2] = q.a.z; v[3] = q.a.w; v[4] = q.b.x; v[5] = q.b.y; v[6] = q.b.z; v[7] = q.b.w;
}
__device__ __forceinline__ void raw8_unpack(const Raw8<bf16_t>& q, float* v) {
  v[0] = __uint_as_float(q.r.x << 16); v[1] = __uint_as_float(q.r.x & 0xffff0000u);
  v[2] = __uint_as_float(q.r.y << 16); v[3] = __uint_as_float(q.r.y & 0xffff0000u);
  v[4] = __uint_as_float(q.r.z << 16); v[5] = __uint_as_float(q.r.z & 0xffff0000u);
  v[6] = __uint_as_float(q.r.w << 16); v[7] = __uint_as_float(q.r.w & 0xffff0000u);
}
// Large outputs are written with NON-TEMPORAL stores (global_store ... nt).  A kernel's ordinary stores stay dirty in its XCD's L2 until the
// release at the END of the kernel writes them back (MI355X_MICROARCH.md: + bytes / ~6 TB/s on the kernel boundary) -- for a conv that leaves
// 19 MB per launch that is ~3 us on a 45-us kernel, and the consumer (any XCD) reads from the fabric either way.  Streamed out as the epilogue
// produces them, the rows cost the boundary less: the step -0.13 ms with the conv family's rows non-temporal (round 4, same-box A/B).  NOT as
// inline asm: a volatile-asm store measured another 0.5 ms faster and was WRONG -- the compiler does not wait for the LDS read that feeds an asm
// operand (stale dwords in single rows; tests/test_gpu_bn_fused.py caught it), and what it gained was exactly that missing wait.
// Other kernels (rows, BatchNorm passes, weight-gradient slabs) measured no gain or a loss with non-temporal outputs: per translation unit.
#ifndef SIMT_NT_STORES
#define SIMT_NT_STORES 0          // per translation unit: a file that wants them defines SIMT_NT_STORES 1 before including this header
#endif
// (`static`: the bodies differ per translation unit through SIMT_NT_STORES -- internal linkage keeps that from being one-definition-rule
// roulette should a build ever stop inlining them or link relocatable device code)
static __device__ __forceinline__ void st_out16(void* p, const uint4& v) {
  typedef unsigned simt_u32x4 __attribute__((ext_vector_type(4)));
  const simt_u32x4 w = {v.x, v.y, v.z, v.w};
#if SIMT_NT_STORES
  __builtin_nontemporal_store(w, (simt_u32x4*)p);
#else
  *(simt_u32x4*)p = w;
#endif
}
static __device__ __forceinline__ void st_out8(bf16_t* p, const float* v) {
  uint4 r;
  r.x = pack_bf16x2(v[0], v[1]); r.y = pack_bf16x2(v[2], v[3]); r.z = pack_bf16x2(v[4], v[5]); r.w = pack_bf16x2(v[6], v[7]);
  st_out16(p, r);
}
static __device__ __forceinline__ void st_out8(float* p, const float* v) {
  st_out16(p, make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
  st_out16(p + 4, make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])));
}

static __device__ __forceinline__ void store8(float* p, const float* v) { st_out8(p, v); }
static __device__ __forceinline__ void store8(bf16_t* p, const float* v) { st_out8(p, v); }
static __device__ __forceinline__ void st_out16f(float* p, const f32x4& v) {      // an accumulator quad (fp32 results: the tap-expanded head GEMMs, weight-gradient slabs)
#if SIMT_NT_STORES
  __builtin_nontemporal_store(v, (f32x4*)p);
#else
  *(f32x4*)p = v;
#endif
}

// Blocks b and b+8 share an XCD (observed round-robin dispatch). Remap so that each XCD works on a
// contiguous chunk of the tile list -> neighbouring tiles share operand panels in one L2. Bijective
// for any nwg. Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// floor(m / d) for 0 <= m < 2^24 via a float reciprocal + one correction step each way.
__device__ __forceinline__ void fast_divmod(int m, int d, float rcp, int& q, int& r) {
  q = (int)((float)m * rcp);
  r = m - q * d;
  if (r < 0) { q--; r += d; }
  if (r >= d) { q++; r -= d; }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: a process that drives several devices (or sets a
// larger size later) needs it once per (device, size).  `cache` is one zero-initialised static per call site; the racy update is benign
// (the attribute call is idempotent).
#define SIMT_MAX_DEVICES 16
struct SimtLdsAttrCache { size_t set[SIMT_MAX_DEVICES]; };
static inline bool simt_lds_attr_needed(SimtLdsAttrCache* c, size_t lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIMT_MAX_DEVICES) return true;
  if (lds <= c->set[dev]) return false;
  c->set[dev] = lds;
  return true;
}

const void* simt_zero_page(void);  // 4 KB of device zeros (conv_igemm.hip)

#define SIMT_CHECK(cond)                                                        \
  do {                                                                          \
    if (!(cond)) {                                                              \
      simt_set_error(__FILE__, __LINE__, #cond);                                \
      return SIMT_ERR_INVALID;                                                  \
    }                                                                           \
  } while (0)
#define SIMT_LAUNCH_CHECK()                                                     \
  do {                                                                          \
    hipError_t e__ = hipGetLastError();                                         \
    if (e__ != hipSuccess) {                                                    \
      simt_set_error(__FILE__, __LINE__, hipGetErrorString(e__));               \
      return SIMT_ERR_LAUNCH;                                                   \
    }                                                                           \
  } while (0)
void simt_set_error(const char* file, int line, const char* msg);
