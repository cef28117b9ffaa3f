// Train-mode BatchNorm (frozen affine), ReLU, residual add, stem max-pool, im2col and weight packing.
// All HBM-bound: NHWC rows are walked with 16-B (bf16) / 32-B (f32) vector accesses, 8 channels per lane.
//
// Replaces, on the reference path:
//   nn.BatchNorm2d in train mode with requires_grad=False affine  (model/deeplab_multi.py:63-76,129-131,158-160)
//   nn.ReLU(inplace) + residual add                                (model/deeplab_multi.py:81-101)
//   nn.MaxPool2d(3, 2, 1, ceil_mode=True)                          (model/deeplab_multi.py:133)
// and their autograd backward passes.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// bn_finalize: per-channel batch statistics from the conv epilogue partials.
//   part [nblk][2][C] (sum, sum of squares)  ->  mean, rstd, scale = gamma*rstd, shift = beta - mean*scale
//   running_mean/var updated like PyTorch (momentum 0.1, unbiased var), if running != null.
// ---------------------------------------------------------------------------------------------
// One block = 8 channels x 32 row lanes: lane r sums partial rows r, r+32, ... , then the 32 lanes are combined in fixed
// order through LDS -> bitwise reproducible.  These kernels sit on the critical path between a conv and the pass that
// consumes its statistics, so they are shaped for latency: C/8 blocks, at most nblk/32 dependent loads per thread.
template <int NJ>
__device__ __forceinline__ void part_colsum8(const float* part, int nblk, int C, int c0, double* out /*[NJ]*/, double (*red)[8][NJ]) {
  const int cl = threadIdx.x & 7, r = threadIdx.x >> 3;
  const int c = c0 + cl;
  double s[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) s[j] = 0.0;
  if (c < C) {
    int b = r;
    for (; b + 96 < nblk; b += 128) {
      float v[4][NJ];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) v[u][j] = part[((long)(b + 32 * u) * NJ + j) * C + c];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < NJ; ++j) s[j] += (double)v[u][j];
    }
    for (; b < nblk; b += 32)
#pragma unroll
      for (int j = 0; j < NJ; ++j) s[j] += (double)part[((long)b * NJ + j) * C + c];
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j) red[r][cl][j] = s[j];
  __syncthreads();
  if (r == 0) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      double t = 0.0;
      for (int q = 0; q < 32; ++q) t += red[q][cl][j];
      out[j] = t;
    }
  }
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* part, int nblk, int C, long count, const float* gamma,
                                                         const float* beta, float* running_mean, float* running_var,
                                                         float momentum, float eps, float* mean_out, float* rstd_out,
                                                         float* scale_out, float* shift_out) {
  __shared__ double red[32][8][2];
  double s[2];
  part_colsum8<2>(part, nblk, C, blockIdx.x * 8, s, red);
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  if ((threadIdx.x >> 3) != 0 || c >= C) return;
  double mean = s[0] / (double)count;
  double var = s[1] / (double)count - mean * mean;
  if (var < 0.0) var = 0.0;
  float rstd = (float)(1.0 / sqrt(var + (double)eps));
  float g = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  float sc = g * rstd;
  mean_out[c] = (float)mean;
  rstd_out[c] = rstd;
  scale_out[c] = sc;
  shift_out[c] = bt - (float)mean * sc;
  if (running_mean) {
    double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

extern "C" int simt_bn_finalize(const float* part, int nblk, int C, long count, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* mean,
                                float* rstd, float* scale, float* shift, simt_stream_t stream) {
  SIMT_CHECK(part && mean && rstd && scale && shift && C > 0 && nblk > 0);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, part, nblk, C, count,
                     gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale, shift);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---------------------------------------------------------------------------------------------
// bn_apply:  z = act( y*scale + shift  [+ res]  [+ y2*scale2 + shift2] )
// ---------------------------------------------------------------------------------------------
// bits (optional): one byte per 8 channels, bit e = (z[e] > 0): the ReLU mask the backward needs, 16x smaller than z
template <typename T>
__global__ void bn_apply_kernel(const T* y, const float* scale, const float* shift, const T* res, const T* y2,
                                const float* scale2, const float* shift2, T* z, unsigned char* bits, long nvec, int C, int relu) {
  const int vpr = C >> 3;
  // the grid stride (gridDim * 256) is a multiple of vpr (vpr divides 256, checked by the launcher): a thread's 8 channels never
  // change, so the per-channel constants are loaded once instead of on every iteration (they were 2-5x the bytes of the data)
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i0 % vpr) << 3;
  float sc[8], sh[8], sc2[8], sh2[8];
  load8(scale + c, sc);
  load8(shift + c, sh);
  if (y2) { load8(scale2 + c, sc2); load8(shift2 + c, sh2); }
  for (long i = i0; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    load8(y + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
    if (res) {
      float r[8];
      load8(res + i * 8, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (y2) {
      float r[8];
      load8(y2 + i * 8, r);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += r[e] * sc2[e] + sh2[e];
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
    }
    store8(z + i * 8, v);
    if (bits) {
      unsigned b = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) b |= (v[e] > 0.f ? 1u : 0u) << e;
      bits[i] = (unsigned char)b;
    }
  }
}

static inline int ew_grid(long nvec) {
  long g = (nvec + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int simt_bn_apply_bits(const void* y, const float* scale, const float* shift, const void* res, const void* y2,
                                  const float* scale2, const float* shift2, void* z, unsigned char* bits, long M, int C,
                                  int relu, int dtype, simt_stream_t stream) {
  SIMT_CHECK(y && scale && shift && z && C % 8 == 0 && 256 % (C / 8) == 0);
  long nvec = M * (C / 8);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y,
                       scale, shift, (const bf16_t*)res, (const bf16_t*)y2, scale2, shift2, (bf16_t*)z, bits, nvec, C, relu);
  else
    hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)y,
                       scale, shift, (const float*)res, (const float*)y2, scale2, shift2, (float*)z, bits, nvec, C, relu);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
extern "C" int simt_bn_apply(const void* y, const float* scale, const float* shift, const void* res, const void* y2,
                             const float* scale2, const float* shift2, void* z, long M, int C, int relu, int dtype,
                             simt_stream_t stream) {
  return simt_bn_apply_bits(y, scale, shift, res, y2, scale2, shift2, z, nullptr, M, C, relu, dtype, stream);
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward (batch statistics, frozen affine).
//   g   = dz * mask         mask: z>0 (mask_mode 1), y*scale+shift>0 (mask_mode 2), 1 (mask_mode 0),
//                           bit (c&7) of byte z[(m*C+c)/8] written by simt_bn_apply_bits (mask_mode 3)
//   xh  = (y - mean)*rstd
//   reduce:  S1 = sum g, S2 = sum g*xh [, S3 = sum g*xh2 for the downsample BN that shares g]
//   apply :  dy = scale*(g - S1/M - xh*S2/M)
// Deterministic: per-block partials [nblk][3][C], summed in fixed order by bn_bwd_finalize.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* dz, const T* z, const T* y, const float* mean,
                                                           const float* rstd, const float* scale, const float* shift,
                                                           const T* y2, const float* mean2, const float* rstd2,
                                                           float* part, long M, int C, int rows_per_block,
                                                           int mask_mode) {
  __shared__ float red[3][256][8 + 1];
  const int vpr = C >> 3;              // vectors per row (<= 256)
  const int rpar = 256 / vpr;          // rows processed in parallel
  const int tid = threadIdx.x;
  const int vc = tid % vpr, rl = tid / vpr;
  const int c = vc << 3;
  float mu[8], rs[8], sc[8], sh[8], mu2[8], rs2[8];
  load8(mean + c, mu);
  load8(rstd + c, rs);
  if (mask_mode == 2) { load8(scale + c, sc); load8(shift + c, sh); }
  if (y2) { load8(mean2 + c, mu2); load8(rstd2 + c, rs2); }
  float s1[8], s2[8], s3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; s3[e] = 0.f; }
  long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  // one row: the sums take the rows of a thread in ascending order whatever the batching below (bitwise the same partials)
  auto row = [&](const Raw8<T>& gq, const Raw8<T>& yq, const Raw8<T>& zq, unsigned b, const Raw8<T>& y2q) {
    float g[8], yv[8];
    raw8_unpack(gq, g);
    raw8_unpack(yq, yv);
    if (mask_mode == 1) {
      float zv[8];
      raw8_unpack(zq, zv);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = zv[e] > 0.f ? g[e] : 0.f;
    } else if (mask_mode == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = (yv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
    } else if (mask_mode == 3) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((b >> e) & 1u) ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += g[e];
      s2[e] += g[e] * ((yv[e] - mu[e]) * rs[e]);
    }
    if (y2) {
      float y2v[8];
      raw8_unpack(y2q, y2v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s3[e] += g[e] * ((y2v[e] - mu2[e]) * rs2[e]);
    }
  };
  if (rl < rpar) {
    // four rows per trip, every load of the trip requested before the first use and kept as loaded (16 bytes per 8 bf16): a thread has
    // 8-12 loads in flight instead of 2-3 (512 blocks of 4 waves: with one row per trip the launch read at 3.4 TB/s)
    constexpr int UR = 4;
    long r = r0 + rl;
    for (; r + (long)(UR - 1) * rpar < r1; r += (long)UR * rpar) {
      Raw8<T> gq[UR], yq[UR], zq[UR], y2q[UR];
      unsigned b[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const long off = (r + (long)u * rpar) * C + c;
        raw8_load(dz + off, gq[u]);
        raw8_load(y + off, yq[u]);
        if (mask_mode == 1) raw8_load(z + off, zq[u]); else zq[u] = gq[u];
        b[u] = mask_mode == 3 ? ((const unsigned char*)z)[off >> 3] : 0u;
        if (y2) raw8_load(y2 + off, y2q[u]); else y2q[u] = gq[u];
      }
#pragma unroll
      for (int u = 0; u < UR; ++u) row(gq[u], yq[u], zq[u], b[u], y2q[u]);
    }
    for (; r < r1; r += rpar) {
      const long off = r * C + c;
      Raw8<T> gq, yq, zq, y2q;
      raw8_load(dz + off, gq);
      raw8_load(y + off, yq);
      if (mask_mode == 1) raw8_load(z + off, zq); else zq = gq;
      const unsigned b = mask_mode == 3 ? ((const unsigned char*)z)[off >> 3] : 0u;
      if (y2) raw8_load(y2 + off, y2q); else y2q = gq;
      row(gq, yq, zq, b, y2q);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) { red[0][tid][e] = s1[e]; red[1][tid][e] = s2[e]; red[2][tid][e] = s3[e]; }
  __syncthreads();
  // fixed-order reduction over the rpar row lanes
  for (int idx = tid; idx < 3 * C; idx += 256) {
    int j = idx / C, cc = idx - j * C;
    int v = cc >> 3, e = cc & 7;
    float s = 0.f;
    for (int q = 0; q < rpar; ++q) s += red[j][q * vpr + v][e];
    part[((long)blockIdx.x * 3 + j) * C + cc] = s;
  }
}

// dgamma / dbeta (trainable affine, e.g. torchvision's BatchNorm in model/deeplabv3.py): dbeta = sum g, dgamma = sum g*xhat
// are exactly the two sums of the backward; written when the pointers are given (dgamma2: the second BN sharing g).
// (launch bounds: 8 waves per SIMD = at most 64 VGPRs.  The weight-gradient launches on the other stream hold 448 of a SIMD's 512 VGPRs on 255 CUs; at
// 73 VGPRs this kernel's workgroups fitted on the one free CU only: 212 us instead of 7 beside conv_wgrad3_multi_kernel, 12 times per step --
// profiles/tools/small_kernel_contention.py.  At 64 it takes 10 us there; the step gains little, the wait moves to the next kernel that cannot share a CU.)
__global__ __launch_bounds__(256, 8) void bn_bwd_finalize_kernel(const float* part, int nblk, int C, long count, float* coef,
                                                             float* dgamma, float* dbeta, float* dgamma2, float* dbeta2) {
  __shared__ double red[32][8][3];
  double s[3];
  part_colsum8<3>(part, nblk, C, blockIdx.x * 8, s, red);
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  if ((threadIdx.x >> 3) != 0 || c >= C) return;
#pragma unroll
  for (int j = 0; j < 3; ++j) coef[j * C + c] = (float)(s[j] / (double)count);
  if (dbeta) dbeta[c] = (float)s[0];
  if (dgamma) dgamma[c] = (float)s[1];
  if (dbeta2) dbeta2[c] = (float)s[0];
  if (dgamma2) dgamma2[c] = (float)s[2];
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* dz, const T* z, const T* y, const float* mean, const float* rstd,
                                    const float* scale, const float* shift, const float* coef, const T* y2,
                                    const float* mean2, const float* rstd2, const float* scale2, T* dy, T* dy2, T* gout,
                                    long nvec, int C, int mask_mode) {
  const int vpr = C >> 3;
  // per-channel constants once per thread (the grid stride is a multiple of vpr, see bn_apply_kernel)
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int c = (int)(i0 % vpr) << 3;
  float mu[8], rs[8], sc[8], c1[8], c2[8], sh[8], mu2[8], rs2[8], sc2[8], c3[8];
  load8(mean + c, mu);
  load8(rstd + c, rs);
  load8(scale + c, sc);
  load8(coef + c, c1);
  load8(coef + C + c, c2);
  if (mask_mode == 2) load8(shift + c, sh);
  if (y2) { load8(mean2 + c, mu2); load8(rstd2 + c, rs2); load8(scale2 + c, sc2); load8(coef + 2 * C + c, c3); }
  for (long i = i0; i < nvec; i += (long)gridDim.x * blockDim.x) {
    float g[8], yv[8];
    load8(dz + i * 8, g);
    load8(y + i * 8, yv);
    if (mask_mode == 1) {
      float zv[8];
      load8(z + i * 8, zv);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = zv[e] > 0.f ? g[e] : 0.f;
    } else if (mask_mode == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = (yv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
    } else if (mask_mode == 3) {
      const unsigned b = ((const unsigned char*)z)[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = ((b >> e) & 1u) ? g[e] : 0.f;
    }
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = sc[e] * (g[e] - c1[e] - ((yv[e] - mu[e]) * rs[e]) * c2[e]);
    store8(dy + i * 8, o);
    if (y2) {
      float y2v[8];
      load8(y2 + i * 8, y2v);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = sc2[e] * (g[e] - c1[e] - ((y2v[e] - mu2[e]) * rs2[e]) * c3[e]);
      store8(dy2 + i * 8, o);
    }
    if (gout) store8(gout + i * 8, g);
  }
}

// The same pass with FOUR channels per thread (bf16, one BatchNorm: every Bottleneck launch of the step), built for at most 64 VGPRs.  Why: the
// weight-gradient launches on the other stream hold 448 of a SIMD's 512 VGPRs on 255 CUs for ~400 us at a time; the 8-channel kernel above (114
// VGPRs: five constants x 8 channels) cannot share a CU with them and ran on whatever CUs they left -- 164 us instead of 38 beside
// conv_wgrad3_multi_kernel, 14 launches per step (profiles/tools/small_kernel_contention.py) -- although one is bound by the matrix pipe and LDS
// and the other by HBM.  Half the constants and half the data per thread fit beside them.  Same expression per element: bitwise the 8-channel
// kernel's output (tests/test_gpu_bn_pool.py).  SIMT_BNB_APPLY4=0 (compile time) keeps the 8-channel form.
#ifndef SIMT_BNB_APPLY4
#define SIMT_BNB_APPLY4 1
#endif
__device__ __forceinline__ void load4(const float* p, float* v) { const float4 a = *(const float4*)p; v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; }
__device__ __forceinline__ void unpack4(const uint2& r, float* v) {
  v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
  v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
}
// UN elements of the grid-stride loop per thread are requested before the first is used (beside the weight gradients this kernel gets ONE wave per
// SIMD: 64 of 512 VGPRs are left).  Measured, same box, four alternating rounds: UN = 1 / 2 / 4 -> 23.42 / 23.45 / 23.62 ms per step against 23.76 for
// the 8-channel kernel: the 4 096-workgroup grid leaves a thread ~9 elements, so deeper batches only lengthen the ragged last round.  MM: the mask
// mode, compile time (registers).
#ifndef SIMT_BNB_APPLY4_UN
#define SIMT_BNB_APPLY4_UN 1
#endif
template <int MM>
__global__ __launch_bounds__(256, 8) void bn_bwd_apply4_kernel(const bf16_t* dz, const bf16_t* z, const bf16_t* y, const float* mean, const float* rstd,
                                                               const float* scale, const float* shift, const float* coef, bf16_t* dy, bf16_t* gout,
                                                               long nvec4, int C) {
  constexpr int UN = SIMT_BNB_APPLY4_UN;
  const int vpr = C >> 2;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;     // (the grid stride is a multiple of vpr: host)
  const int c = (int)(i0 % vpr) << 2;
  float mu[4], rs[4], sc[4], c1[4], c2[4], sh[4];
  load4(mean + c, mu);
  load4(rstd + c, rs);
  load4(scale + c, sc);
  load4(coef + c, c1);
  load4(coef + C + c, c2);
  if (MM == 2) load4(shift + c, sh);
  // 32-bit byte offsets from uniform bases (host: the tensor is < 2 GB): one address register per element in flight
  const unsigned nb = (unsigned)nvec4 * 8u, Sb = gridDim.x * blockDim.x * 8u;
  const char* dzb = (const char*)dz; const char* yb = (const char*)y; const char* zb = (const char*)z;
  char* dyb = (char*)dy; char* gb = (char*)gout;
  for (unsigned o0 = (unsigned)i0 * 8u; o0 < nb; o0 += Sb * UN) {
    uint2 gr[UN], yr[UN], zr[UN];
    unsigned bb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {                                 // (past the end: the first element again -- loaded, never stored)
      const unsigned ou = o0 + u * Sb < nb ? o0 + u * Sb : o0;
      gr[u] = *(const uint2*)(dzb + ou);
      yr[u] = *(const uint2*)(yb + ou);
      if (MM == 1) zr[u] = *(const uint2*)(zb + ou);
      if (MM == 3) bb[u] = (unsigned)((const unsigned char*)zb)[ou >> 4] >> ((ou >> 1) & 4u);      // element ou / 8: byte (ou / 8) / 2, nibble (ou / 8) & 1
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const unsigned ou = o0 + u * Sb;
      if (ou >= nb) break;
      float g[4], yv[4];
      unpack4(gr[u], g);
      unpack4(yr[u], yv);
      if (MM == 1) {
        float zv[4];
        unpack4(zr[u], zv);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = zv[e] > 0.f ? g[e] : 0.f;
      } else if (MM == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (yv[e] * sc[e] + sh[e]) > 0.f ? g[e] : 0.f;
      } else if (MM == 3) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = ((bb[u] >> e) & 1u) ? g[e] : 0.f;
      }
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = sc[e] * (g[e] - c1[e] - ((yv[e] - mu[e]) * rs[e]) * c2[e]);
      *(uint2*)(dyb + ou) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
      if (gout) *(uint2*)(gb + ou) = make_uint2(pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]));
    }
  }
}

static inline int bn_bwd_rows_per_block(long M, int C) {
  int rpar = 256 / (C / 8);
  long target_blocks = 512;   // partials are [nblk][3][C] floats re-read by the finalize pass: keep them small
  long rpb = (M + target_blocks - 1) / target_blocks;
  rpb = ((rpb + rpar - 1) / rpar) * rpar;
  if (rpb < rpar * 4) rpb = rpar * 4;
  return (int)rpb;
}

extern "C" int simt_bn_bwd_nblk(long M, int C) {
  int rpb = bn_bwd_rows_per_block(M, C);
  return (int)((M + rpb - 1) / rpb);
}

extern "C" int simt_bn_bwd(const simt_bn_bwd_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->dz && d->y && d->mean && d->rstd && d->scale && d->part && d->coef && d->dy);
  SIMT_CHECK(d->C % 8 == 0 && d->C / 8 <= 256 && 256 % (d->C / 8) == 0);
  SIMT_CHECK((d->mask_mode != 1 && d->mask_mode != 3) || d->z);
  SIMT_CHECK(d->mask_mode != 2 || d->shift);
  SIMT_CHECK(!d->y2 || (d->mean2 && d->rstd2 && d->scale2 && d->dy2));
  hipStream_t st = (hipStream_t)stream;
  const int rpb = bn_bwd_rows_per_block(d->M, d->C);
  const int nblk = d->reduce_done_nblk > 0 ? d->reduce_done_nblk : (int)((d->M + rpb - 1) / rpb);
  const long nvec = d->M * (d->C / 8);
  if (d->reduce_done_nblk > 0) {
    SIMT_CHECK(!d->y2);     // the conv epilogue reduces for ONE BatchNorm (no downsample partner)
  } else if (d->dtype == SIMT_BF16) {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)d->dz,
                       (const bf16_t*)d->z, (const bf16_t*)d->y, d->mean, d->rstd, d->scale, d->shift,
                       (const bf16_t*)d->y2, d->mean2, d->rstd2, d->part, d->M, d->C, rpb, d->mask_mode);
  } else {
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)d->dz,
                       (const float*)d->z, (const float*)d->y, d->mean, d->rstd, d->scale, d->shift,
                       (const float*)d->y2, d->mean2, d->rstd2, d->part, d->M, d->C, rpb, d->mask_mode);
  }
  SIMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((d->C + 7) / 8), dim3(256), 0, st, d->part, nblk, d->C, d->M, d->coef,
                     d->dgamma, d->dbeta, d->dgamma2, d->dbeta2);
  SIMT_LAUNCH_CHECK();
  if (SIMT_BNB_APPLY4 && d->dtype == SIMT_BF16 && !d->y2 && nvec * 16 < (1l << 31)) {
#ifndef SIMT_BNB_APPLY4_CAP
#define SIMT_BNB_APPLY4_CAP 4096
#endif
    long gl = (nvec * 2 + 256l * SIMT_BNB_APPLY4_UN - 1) / (256l * SIMT_BNB_APPLY4_UN);
    int g4 = (int)(gl > SIMT_BNB_APPLY4_CAP ? SIMT_BNB_APPLY4_CAP : gl < 1 ? 1 : gl);
    if (d->C / 4 > 256) g4 = (g4 + 1) & ~1;                    // the grid stride (g4 * 256 threads) stays a multiple of C / 4 <= 512
#define SIMT_BNB4(MM) hipLaunchKernelGGL(bn_bwd_apply4_kernel<MM>, dim3(g4), dim3(256), 0, st, (const bf16_t*)d->dz, (const bf16_t*)d->z, (const bf16_t*)d->y, \
                                         d->mean, d->rstd, d->scale, d->shift, d->coef, (bf16_t*)d->dy, (bf16_t*)d->gout, nvec * 2, d->C)
    if (d->mask_mode == 3) SIMT_BNB4(3); else if (d->mask_mode == 2) SIMT_BNB4(2); else if (d->mask_mode == 1) SIMT_BNB4(1); else SIMT_BNB4(0);
#undef SIMT_BNB4
  } else if (d->dtype == SIMT_BF16) {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const bf16_t*)d->dz,
                       (const bf16_t*)d->z, (const bf16_t*)d->y, d->mean, d->rstd, d->scale, d->shift, d->coef,
                       (const bf16_t*)d->y2, d->mean2, d->rstd2, d->scale2, (bf16_t*)d->dy, (bf16_t*)d->dy2,
                       (bf16_t*)d->gout, nvec, d->C, d->mask_mode);
  } else {
    hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, st, (const float*)d->dz,
                       (const float*)d->z, (const float*)d->y, d->mean, d->rstd, d->scale, d->shift, d->coef,
                       (const float*)d->y2, d->mean2, d->rstd2, d->scale2, (float*)d->dy, (float*)d->dy2,
                       (float*)d->gout, nvec, d->C, d->mask_mode);
  }
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---------------------------------------------------------------------------------------------
// Stem: im2col for the 7x7 stride-2 pad-3 conv on the NCHW fp32 image (Cin = 3 is not MFMA-shaped as an
// implicit GEMM; the explicit [M][192] matrix is 0.2 GB at 4x768x768 and is read once by the GEMM).
// Column order k = ci*KH*KW + r*KW + s (matches the OIHW flattening of the weight), zero-padded to ldk.
// ---------------------------------------------------------------------------------------------
// KHC / KWC > 0: compile-time filter size (7x7 stem of the ResNets, 3x3 first conv of VGG): the per-element k -> (ci, r, s)
// decomposition becomes multiply-shift instead of three runtime integer divisions (the generic build spent 227 us on the 226 MB
// matrix of the 768x768 stem, 4x its HBM write time).
template <typename T, int KHC, int KWC>
__global__ void im2col_stem_kernel(const float* x, T* A, int B, int Cin, int H, int W, int Ho, int Wo, int KHr, int KWr,
                                   int stride, int pad, int ldk, long nvec) {
  const int KH = KHC > 0 ? KHC : KHr, KW = KWC > 0 ? KWC : KWr;
  const int vpr = ldk >> 3;
  const int K = Cin * KH * KW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    int kv = (int)(i % vpr) << 3;
    long m = i / vpr;
    int ox = (int)(m % Wo);
    long t = m / Wo;
    int oy = (int)(t % Ho);
    int b = (int)(t / Ho);
    const int iy0 = oy * stride - pad, ix0 = ox * stride - pad;
    const float* xb = x + (long)b * Cin * H * W;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      int k = kv + e;
      float val = 0.f;
      if (k < K) {
        int ci = k / (KH * KW);
        int rs = k - ci * KH * KW;
        int r = rs / KW, s = rs - r * KW;
        int iy = iy0 + r, ix = ix0 + s;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) val = xb[((long)ci * H + iy) * W + ix];
      }
      v[e] = val;
    }
    store8(A + i * 8, v);
  }
}

template <typename T>
static void launch_im2col(const float* x, T* A, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad,
                          int ldk, long nvec, hipStream_t st) {
  const dim3 grid(ew_grid(nvec)), blk(256);
  if (KH == 7 && KW == 7)
    hipLaunchKernelGGL((im2col_stem_kernel<T, 7, 7>), grid, blk, 0, st, x, A, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk, nvec);
  else if (KH == 3 && KW == 3)
    hipLaunchKernelGGL((im2col_stem_kernel<T, 3, 3>), grid, blk, 0, st, x, A, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk, nvec);
  else
    hipLaunchKernelGGL((im2col_stem_kernel<T, 0, 0>), grid, blk, 0, st, x, A, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk, nvec);
}

extern "C" int simt_im2col_stem(const float* x, void* A, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW,
                                int stride, int pad, int ldk, int dtype, simt_stream_t stream) {
  SIMT_CHECK(x && A && ldk % 8 == 0 && ldk >= Cin * KH * KW);
  long nvec = (long)B * Ho * Wo * (ldk / 8);
  if (dtype == SIMT_BF16) launch_im2col<bf16_t>(x, (bf16_t*)A, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk, nvec, (hipStream_t)stream);
  else launch_im2col<float>(x, (float*)A, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk, nvec, (hipStream_t)stream);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---------------------------------------------------------------------------------------------
// Stem pool: p = maxpool3x3/s2/p1/ceil( relu(y*scale+shift) ), first-max index kept in a byte.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void bn_relu_maxpool_kernel(const T* y, const float* scale, const float* shift, T* p, unsigned char* idx,
                                       int B, int H, int W, int C, int Hp, int Wp, long nvec) {
  const int vpr = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % vpr) << 3;
    long m = i / vpr;
    int px = (int)(m % Wp);
    long t = m / Wp;
    int py = (int)(t % Hp);
    int b = (int)(t / Hp);
    float sc[8], sh[8], best[8];
    int bi[8];
    load8(scale + c, sc);
    load8(shift + c, sh);
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
    bool first = true;
    for (int r = 0; r < 3; ++r) {
      int iy = py * 2 - 1 + r;
      if (iy < 0 || iy >= H) continue;
      for (int s = 0; s < 3; ++s) {
        int ix = px * 2 - 1 + s;
        if (ix < 0 || ix >= W) continue;
        float v[8];
        load8(y + (((long)b * H + iy) * W + ix) * C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float a = v[e] * sc[e] + sh[e];
          a = a > 0.f ? a : 0.f;
          // PyTorch: take if (val > max) || isnan(val); first element always taken
          if (first || a > best[e] || a != a) { best[e] = a; bi[e] = r * 3 + s; }
        }
        first = false;
      }
    }
    store8(p + i * 8, best);
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed |= (unsigned long long)(bi[e] & 0xff) << (8 * e);
    *(unsigned long long*)(idx + i * 8) = packed;
  }
}

// da[b,iy,ix,c] = sum over windows containing (iy,ix) whose argmax is (iy,ix) of dp
template <typename T>
__global__ void maxpool_bwd_kernel(const T* dp, const unsigned char* idx, T* da, int B, int H, int W, int C, int Hp,
                                   int Wp, long nvec) {
  const int vpr = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % vpr) << 3;
    long m = i / vpr;
    int ix = (int)(m % W);
    long t = m / W;
    int iy = (int)(t % H);
    int b = (int)(t / H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    int py0 = iy >> 1, px0 = ix >> 1;  // candidates: py in {floor(iy/2), floor(iy/2)+ (iy odd)}: windows start at 2py-1
    for (int py = py0; py <= ((iy + 1) >> 1); ++py) {
      if (py >= Hp) continue;
      int r = iy - (py * 2 - 1);
      if (r < 0 || r > 2) continue;
      for (int px = px0; px <= ((ix + 1) >> 1); ++px) {
        if (px >= Wp) continue;
        int s = ix - (px * 2 - 1);
        if (s < 0 || s > 2) continue;
        long o = (((long)b * Hp + py) * Wp + px) * C + c;
        unsigned long long packed = *(const unsigned long long*)(idx + o);
        float g[8];
        load8(dp + o, g);
        int want = r * 3 + s;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((int)((packed >> (8 * e)) & 0xff) == want) acc[e] += g[e];
      }
    }
    store8(da + i * 8, acc);
  }
}

extern "C" int simt_bn_relu_maxpool(const void* y, const float* scale, const float* shift, void* p, unsigned char* idx,
                                    int B, int H, int W, int C, int Hp, int Wp, int dtype, simt_stream_t stream) {
  SIMT_CHECK(y && scale && shift && p && idx && C % 8 == 0);
  long nvec = (long)B * Hp * Wp * (C / 8);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)y, scale, shift, (bf16_t*)p, idx, B, H, W, C, Hp, Wp, nvec);
  else
    hipLaunchKernelGGL(bn_relu_maxpool_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)y, scale, shift, (float*)p, idx, B, H, W, C, Hp, Wp, nvec);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

extern "C" int simt_maxpool_bwd(const void* dp, const unsigned char* idx, void* da, int B, int H, int W, int C, int Hp,
                                int Wp, int dtype, simt_stream_t stream) {
  SIMT_CHECK(dp && idx && da && C % 8 == 0);
  long nvec = (long)B * H * W * (C / 8);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)dp, idx, (bf16_t*)da, B, H, W, C, Hp, Wp, nvec);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)dp,
                       idx, (float*)da, B, H, W, C, Hp, Wp, nvec);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight packing: OIHW fp32 master -> K-contiguous GEMM operand in the compute dtype.
//   mode 0 (fprop): dst[(row_off+co)*ldk + (tap_off+t)*Cin + ci] = w[co][ci][t] * (cscale ? cscale[co] : 1)
//   mode 1 (dgrad): dst[ci*ldk + (tap_off+t)*Ck + row_off + co]  = w[co][ci][t]
//   mode 2 (tap-expanded fprop, one output column per (tap, cout)): dst[((tap_off+t)*Ck + row_off + co)*ldk + ci] = w[co][ci][t]
// Padding entries are never written (the buffer is zeroed once at allocation).
// `mode` bits 8.. = nt16 > 0 (SIMT_PACK_FRAG): the K-contiguous offset o = row * ldk + kcol is re-mapped to MFMA-fragment order
// (simt_conv_desc.w_frag): the weight operand of v_mfma_f32_16x16x32_bf16 for 16 rows x 32 k as 64 lanes x 16 B, contiguous.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ long long frag_offset(long long row, long long kcol, int nt16) {
  const long long kt = kcol >> 6, s = (kcol >> 5) & 1, kq = (kcol >> 3) & 3, e = kcol & 7;
  const long long lane = (row & 15) | (kq << 4);
  return ((((kt * nt16 + (row >> 4)) * 2 + s) * 64 + lane) << 3) + e;
}
template <typename T>
__global__ void pack_weight_kernel(const float* w, T* dst, int Cout, int Cin, int RS, int row_off, int tap_off, long ldk,
                                   int Ck, int mode, const float* cscale, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int t = (int)(i % RS);
    long r = i / RS;
    int ci = (int)(r % Cin);
    int co = (int)(r / Cin);
    float v = w[i];
    if (cscale) v *= cscale[co];
    const int lm = mode & 0xff, nt16 = mode >> 8;
    const long row = lm == 0 ? (long)(row_off + co) : lm == 1 ? (long)ci : (long)(tap_off + t) * Ck + row_off + co;
    const long kcol = lm == 0 ? (long)(tap_off + t) * Cin + ci : lm == 1 ? (long)(tap_off + t) * Ck + row_off + co : (long)ci;
    const long o = nt16 ? (long)frag_offset(row, kcol, nt16) : row * ldk + kcol;
    Elem<T>::st(dst + o, v);
  }
}

extern "C" int simt_pack_weight(const float* w, void* dst, int Cout, int Cin, int RS, int row_off, int tap_off, long ldk,
                                int Ck, int mode, const float* cscale, int dtype, simt_stream_t stream) {
  SIMT_CHECK(w && dst && Cout > 0 && Cin > 0 && RS > 0);
  long total = (long)Cout * Cin * RS;
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)dst,
                       Cout, Cin, RS, row_off, tap_off, ldk, Ck, mode, cscale, total);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, w, (float*)dst,
                       Cout, Cin, RS, row_off, tap_off, ldk, Ck, mode, cscale, total);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// Batched form: one launch packs every weight of a plan (306 launches of ~4 us each per training step otherwise).
// chunks[i] = (job, tile index); tiles enumerate ceil(Cout/32) x ceil(Cin/32).
struct PackJob {
  const float* w;
  void* dst;
  const float* cscale;
  long long ldk, total;
  int Cout, Cin, RS, row_off, tap_off, Ck, mode, dtype;
};
// One block = one 32 (cout) x 32 (cin) tile of one job, all RS taps: the OIHW source is read as 32 contiguous runs of
// 32*RS floats, transposed through LDS, and written with the destination's fastest index across lanes (2-byte scattered
// stores made the first version of this kernel 10x slower than the bytes it moves).
// RSC > 0: compile-time tap count (1x1 and 3x3 cover all but the 7x7 stem): the (co, ci, t) index arithmetic of the read pass is
// multiply-shift instead of runtime integer division.
template <int RSC>
__device__ __forceinline__ void pack_tile(const PackJob& j, int tidx, float* tile) {
  const int ntc = (j.Cin + 31) >> 5;
  const int co0 = (tidx / ntc) << 5, ci0 = (tidx % ntc) << 5;
  const int RS = RSC > 0 ? RSC : j.RS;
  const int nci = min(32, j.Cin - ci0), nco = min(32, j.Cout - co0);
  for (int tb = 0; tb < RS; tb += 9) {          // RS <= 9 in one pass (3x3); larger kernels in slices of 9 taps
    const int nt = min(9, RS - tb);
    __syncthreads();
    // ---- read: element e -> (co_l, ci_l, t) with (ci_l, t) contiguous in memory
    const int per_co = nci * RS;
    for (int e = threadIdx.x; e < nco * per_co; e += 256) {
      const int co_l = e / per_co, rem = e - co_l * per_co;
      const int ci_l = rem / RS, t = rem - ci_l * RS;
      if (t < tb || t >= tb + nt) continue;
      float v = j.w[((long long)(co0 + co_l) * j.Cin + ci0) * RS + rem];
      if (j.cscale) v *= j.cscale[co0 + co_l];
      tile[((t - tb) * 32 + co_l) * 33 + ci_l] = v;
    }
    __syncthreads();
    // ---- write
    for (int e = threadIdx.x; e < nt * 32 * 32; e += 256) {
      const int t = e >> 10;
      int co_l, ci_l;
      const int lm = j.mode & 0xff, nt16 = j.mode >> 8;
      if (lm == 1) { co_l = e & 31; ci_l = (e >> 5) & 31; } else { ci_l = e & 31; co_l = (e >> 5) & 31; }
      if (co_l >= nco || ci_l >= nci) continue;
      const float v = tile[(t * 32 + co_l) * 33 + ci_l];
      const int co = co0 + co_l, ci = ci0 + ci_l, tt = j.tap_off + tb + t;
      const long long row = lm == 0 ? (long long)(j.row_off + co) : lm == 1 ? (long long)ci : (long long)tt * j.Ck + j.row_off + co;
      const long long kcol = lm == 0 ? (long long)tt * j.Cin + ci : lm == 1 ? (long long)tt * j.Ck + j.row_off + co : (long long)ci;
      const long long o = nt16 ? frag_offset(row, kcol, nt16) : row * j.ldk + kcol;
      if (j.dtype == SIMT_BF16) ((bf16_t*)j.dst)[o] = f2bf(v); else ((float*)j.dst)[o] = v;
    }
  }
}
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackJob* jobs, const int* chunks, int /*chunk*/) {
  __shared__ float tile[9 * 32 * 33];
  const PackJob j = jobs[chunks[2 * blockIdx.x]];
  const int tidx = chunks[2 * blockIdx.x + 1];
  if (j.RS == 1) pack_tile<1>(j, tidx, tile);
  else if (j.RS == 9) pack_tile<9>(j, tidx, tile);
  else pack_tile<0>(j, tidx, tile);
}
extern "C" int simt_pack_weight_multi(const void* jobs, const void* chunks, int nchunks, int chunk, simt_stream_t stream) {
  SIMT_CHECK(jobs && chunks && nchunks > 0);
  hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, (const PackJob*)jobs,
                     (const int*)chunks, chunk);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// eval-mode BN folding constants: scale = gamma/sqrt(rv+eps), shift = beta - rm*scale
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}
extern "C" int simt_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                            float* scale, float* shift, int C, simt_stream_t stream) {
  SIMT_CHECK(gamma && beta && rm && rv && scale && shift);
  hipLaunchKernelGGL(bn_fold_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, gamma, beta, rm, rv, eps,
                     scale, shift, C);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---------------------------------------------------------------------------------------------
// Small glue used by the stride-2 1x1 dgrad and the head bias gradient.
// ---------------------------------------------------------------------------------------------
// dx[b, 2*oy, 2*ox, :] (+)= src[b,oy,ox,:]; every other pixel of dx is written with zero (or left, if add).
template <typename T>
__global__ void scatter_stride_kernel(const T* src, T* dx, int B, int H, int W, int C, int Ho, int Wo, int stride,
                                      long nvec) {
  const int vpr = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % vpr) << 3;
    long m = i / vpr;
    int ix = (int)(m % W);
    long t = m / W;
    int iy = (int)(t % H);
    int b = (int)(t / H);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.f;
    if (iy % stride == 0 && ix % stride == 0) {
      int oy = iy / stride, ox = ix / stride;
      if (oy < Ho && ox < Wo) load8(src + (((long)b * Ho + oy) * Wo + ox) * C + c, v);
    }
    store8(dx + i * 8, v);
  }
}
extern "C" int simt_scatter_stride(const void* src, void* dx, int B, int H, int W, int C, int Ho, int Wo, int stride,
                                   int dtype, simt_stream_t stream) {
  SIMT_CHECK(src && dx && C % 8 == 0);
  long nvec = (long)B * H * W * (C / 8);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(scatter_stride_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)src, (bf16_t*)dx, B, H, W, C, Ho, Wo, stride, nvec);
  else
    hipLaunchKernelGGL(scatter_stride_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream,
                       (const float*)src, (float*)dx, B, H, W, C, Ho, Wo, stride, nvec);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// out[c] (+)= sum_m src[m*ld + c]   (bias gradients of the head convs; c < C <= 64).  Two stages, fixed order:
// COLSUM_G blocks each sum a contiguous row range into a library-owned scratch, then one block combines them.
#define COLSUM_G 128
__device__ double g_colsum_ws[COLSUM_G * 64];

template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* src, long M, int ld, int C) {
  __shared__ double red[4][64];
  const int c = threadIdx.x & 63, lr = threadIdx.x >> 6;
  long rows = (M + COLSUM_G - 1) / COLSUM_G;
  long m0 = (long)blockIdx.x * rows, m1 = m0 + rows;
  if (m1 > M) m1 = M;
  double s = 0.0;
  if (c < C)
    for (long m = m0 + lr; m < m1; m += 4) s += (double)Elem<T>::ld(src + m * ld + c);
  red[lr][c] = s;
  __syncthreads();
  if (lr == 0) g_colsum_ws[blockIdx.x * 64 + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}
__global__ void colsum_final_kernel(float* out, int C, int accumulate) {
  int c = threadIdx.x;
  if (c >= C) return;
  double t = 0.0;
  for (int b = 0; b < COLSUM_G; ++b) t += g_colsum_ws[b * 64 + c];
  out[c] = accumulate ? out[c] + (float)t : (float)t;
}
extern "C" int simt_colsum(const void* src, float* out, long M, int ld, int C, int accumulate, int dtype,
                           simt_stream_t stream) {
  SIMT_CHECK(src && out && C <= 64);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, dim3(COLSUM_G), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, M, ld, C);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<float>, dim3(COLSUM_G), dim3(256), 0, (hipStream_t)stream, (const float*)src, M, ld, C);
  SIMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, C, accumulate);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---------------------------------------------------------------------------------------------
// VGG trunk pieces (model/deeplab_vgg.py:24-43 -> torchvision vgg16.features): MaxPool2d(2, 2) (floor mode) forward with a
// 2-bit arg-max index, and its backward fused with the ReLU mask of the conv output it follows.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void maxpool2_kernel(const T* y, T* p, unsigned char* idx, int B, int H, int W, int C, int Hp, int Wp, long nvec) {
  const int vpr = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % vpr) << 3;
    long m = i / vpr;
    int px = (int)(m % Wp);
    long t = m / Wp;
    int py = (int)(t % Hp);
    int b = (int)(t / Hp);
    float best[8];
    int bi[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
        load8(y + (((long)b * H + py * 2 + r) * W + px * 2 + s) * C + c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if ((r == 0 && s == 0) || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 2 + s; }
      }
    store8(p + i * 8, best);
    unsigned long long packed = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) packed |= (unsigned long long)bi[e] << (8 * e);
    *(unsigned long long*)(idx + i * 8) = packed;
  }
}

// da[b,iy,ix,c] = (y > 0) ? dp[b,iy/2,ix/2,c] if idx says (iy&1, ix&1) was the max : 0 ; pixels outside the pooled area -> 0
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* dp, const unsigned char* idx, const T* y, T* da, int B, int H, int W, int C, int Hp,
                                    int Wp, long nvec) {
  const int vpr = C >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % vpr) << 3;
    long m = i / vpr;
    int ix = (int)(m % W);
    long t = m / W;
    int iy = (int)(t % H);
    int b = (int)(t / H);
    float out[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = 0.f;
    const int py = iy >> 1, px = ix >> 1;
    if (py < Hp && px < Wp) {
      long o = (((long)b * Hp + py) * Wp + px) * C + c;
      unsigned long long packed = *(const unsigned long long*)(idx + o);
      float g[8], yv[8];
      load8(dp + o, g);
      load8(y + i * 8, yv);
      const int want = (iy & 1) * 2 + (ix & 1);
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((int)((packed >> (8 * e)) & 0xff) == want && (!y || yv[e] > 0.f)) out[e] = g[e];
    }
    store8(da + i * 8, out);
  }
}

extern "C" int simt_maxpool2(const void* y, void* p, unsigned char* idx, int B, int H, int W, int C, int dtype, simt_stream_t stream) {
  SIMT_CHECK(y && p && idx && C % 8 == 0);
  const int Hp = H / 2, Wp = W / 2;
  long nvec = (long)B * Hp * Wp * (C / 8);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(maxpool2_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)y, (bf16_t*)p,
                       idx, B, H, W, C, Hp, Wp, nvec);
  else
    hipLaunchKernelGGL(maxpool2_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)y, (float*)p, idx,
                       B, H, W, C, Hp, Wp, nvec);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

extern "C" int simt_maxpool2_bwd(const void* dp, const unsigned char* idx, const void* y, void* da, int B, int H, int W, int C, int dtype,
                                 simt_stream_t stream) {
  SIMT_CHECK(dp && idx && y && da && C % 8 == 0);
  const int Hp = H / 2, Wp = W / 2;
  long nvec = (long)B * H * W * (C / 8);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dp, idx,
                       (const bf16_t*)y, (bf16_t*)da, B, H, W, C, Hp, Wp, nvec);
  else
    hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(ew_grid(nvec)), dim3(256), 0, (hipStream_t)stream, (const float*)dp, idx,
                       (const float*)y, (float*)da, B, H, W, C, Hp, Wp, nvec);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// Bias gradients of wide layers: out[c] = sum_m src[m*ld + c], c < C (C a multiple of 8, <= 2048).  Two deterministic stages:
// COLSUMW_G row ranges -> scratch [G][C] (16-byte loads: a block covers 256 / (C/8) rows per pass), then a fixed-order combine.
// (The first version read 2 bytes per lane from 64 x C/64 blocks: 610 us per launch on the 2M x 64 gradient of VGG's conv1_1.)
#define COLSUMW_G 1024
#define COLSUMW_MAXC 2048
__device__ float g_colsumw_ws[COLSUMW_G * COLSUMW_MAXC];
template <typename T>
__global__ __launch_bounds__(256) void colsum_wide_partial_kernel(const T* src, long M, int ld, int C, int rows_per_block) {
  __shared__ float red[256][8 + 1];
  const int vpr = C >> 3;              // 16-byte (bf16) / 32-byte (fp32) vectors per row, <= 256
  const int rpar = 256 / vpr;          // rows in parallel
  const int tid = threadIdx.x, vc = tid % vpr, rl = tid / vpr;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  if (rl < rpar) {
    long r = r0 + rl;
    for (; r + 3 * rpar < r1; r += 4 * rpar) {        // four independent 16-byte loads in flight per thread
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) load8(src + (r + u * rpar) * ld + (vc << 3), v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[u][e];
    }
    for (; r < r1; r += rpar) {
      float v[8];
      load8(src + r * ld + (vc << 3), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[tid][e] = s[e];
  __syncthreads();
  for (int cc = tid; cc < C; cc += 256) {          // fixed-order fold of the rpar row lanes
    const int v = cc >> 3, e = cc & 7;
    float t = 0.f;
    for (int q = 0; q < rpar; ++q) t += red[q * vpr + v][e];
    g_colsumw_ws[(long)blockIdx.x * C + cc] = t;
  }
}
__global__ __launch_bounds__(256) void colsum_wide_final_kernel(float* out, int C, int nblk) {
  // 8 channels x 32 row lanes per block, at most nblk/32 dependent loads per thread (the serial version spent ~0.5 ms here)
  __shared__ double red[32][8][1];
  double s[1];
  part_colsum8<1>(g_colsumw_ws, nblk, C, blockIdx.x * 8, s, red);
  const int c = blockIdx.x * 8 + (threadIdx.x & 7);
  if ((threadIdx.x >> 3) == 0 && c < C) out[c] = (float)s[0];
}
extern "C" int simt_colsum_wide(const void* src, float* out, long M, int ld, int C, int dtype, simt_stream_t stream) {
  SIMT_CHECK(src && out && C > 0 && C <= COLSUMW_MAXC && C % 8 == 0 && 256 % (C / 8) == 0 && ld % 8 == 0);
  const int rpar = 256 / (C / 8);
  long rpb = (M + COLSUMW_G - 1) / COLSUMW_G;
  rpb = ((rpb + rpar - 1) / rpar) * rpar;
  if (rpb < rpar) rpb = rpar;
  const int nblk = (int)((M + rpb - 1) / rpb);
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(colsum_wide_partial_kernel<bf16_t>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, M, ld, C, (int)rpb);
  else
    hipLaunchKernelGGL(colsum_wide_partial_kernel<float>, dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const float*)src, M, ld, C, (int)rpb);
  SIMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(colsum_wide_final_kernel, dim3((C + 7) / 8), dim3(256), 0, (hipStream_t)stream, out, C, nblk);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
