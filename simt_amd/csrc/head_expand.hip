// Tap-expanded form of the ASPP classifier (reference model/deeplab_multi.py:104-119: the sum of two dilated 3x3 convs
// Cin -> Q with Q = 22): as an implicit GEMM it is M x 22 x (18*Cin) -- N = 22 starves the MFMA tile.  Re-associated:
//   P[m'][t*QP + n] = sum_c x[m'][c] * w[n][t][c]          one plain GEMM, N = 18*QP = 432 columns (conv_igemm2, fp32 out)
//   y[m][n]         = bias[n] + sum_t P[m + d_t][t*QP + n]  tap gather-sum (this file), each P element read once
// and for the backward   G[m'][t*QP + n] = dy[m' - d_t][n]  (this file), then
//   dx = G * Wt  (plain GEMM, K = 448)   and   dW[n][t][c] = sum_m' G[m'][t*QP+n] * x[m'][c]  (conv_wgrad2, Cd = 432).
// Same FLOPs, well-shaped tiles, and the 18 shifted re-reads of the 154 MB feature map disappear.
#include "common.h"

struct TapArgs {
  const void* src;
  const float* bias;
  void* dst;
  int B, H, W, Q, QP, lds, ldd, ntaps;
  long M;
  short dy[SIMT_MAX_TAPS], dx[SIMT_MAX_TAPS];
};

// y[m][4q..4q+3] = bias + sum_t P[(m shifted by tap t)][t*QP + 4q ..]      (fp32, one thread per pixel and column quad)
__global__ __launch_bounds__(256) void tap_gather_sum_kernel(TapArgs a) {
  const int nq = a.QP >> 2;
  const long total = a.M * nq;
  const float* P = (const float*)a.src;
  float* y = (float*)a.dst;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % nq);
    const long m = idx / nq;
    const int ox = (int)(m % a.W);
    const long t2 = m / a.W;
    const int oy = (int)(t2 % a.H);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) {
      const int c = q * 4;
      acc.x = c + 0 < a.Q ? a.bias[c + 0] : 0.f;
      acc.y = c + 1 < a.Q ? a.bias[c + 1] : 0.f;
      acc.z = c + 2 < a.Q ? a.bias[c + 2] : 0.f;
      acc.w = c + 3 < a.Q ? a.bias[c + 3] : 0.f;
    }
    for (int t = 0; t < a.ntaps; ++t) {
      const int iy = oy + a.dy[t], ix = ox + a.dx[t];
      if (iy < 0 || iy >= a.H || ix < 0 || ix >= a.W) continue;
      const float4 v = *(const float4*)(P + (m + (long)a.dy[t] * a.W + a.dx[t]) * a.lds + t * a.QP + q * 4);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *(float4*)(y + m * a.ldd + q * 4) = acc;
  }
}

// G[m'][t*QP + 8g .. +7] = dy[m' - d_t][8g .. +7]   (bf16; zero outside the image); one thread per (pixel, tap, 8 columns)
__global__ __launch_bounds__(256) void tap_scatter_kernel(TapArgs a) {
  const int ng = a.QP >> 3;
  const long total = a.M * a.ntaps * ng;
  const bf16_t* d = (const bf16_t*)a.src;
  bf16_t* G = (bf16_t*)a.dst;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int g = (int)(idx % ng);
    long r = idx / ng;
    const int t = (int)(r % a.ntaps);
    const long m = r / a.ntaps;
    const int ox = (int)(m % a.W);
    const int oy = (int)((m / a.W) % a.H);
    const int iy = oy - a.dy[t], ix = ox - a.dx[t];
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
      v = *(const uint4*)(d + (m - ((long)a.dy[t] * a.W + a.dx[t])) * a.lds + g * 8);
    st_out16(G + m * a.ldd + t * a.QP + g * 8, v);
  }
}

static int tap_grid(long total) {
  long g = (total + 255) / 256;
  if (g > 256 * 16) g = 256 * 16;
  return (int)(g < 1 ? 1 : g);
}

extern "C" int simt_tap_gather_sum(const simt_tap_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->src && d->dst && d->ntaps >= 1 && d->ntaps <= SIMT_MAX_TAPS);
  SIMT_CHECK(d->QP % 4 == 0 && d->lds % 4 == 0 && d->ldd % 4 == 0 && d->QP <= d->ldd && d->ntaps * d->QP <= d->lds);
  TapArgs a;
  a.src = d->src; a.bias = d->bias; a.dst = d->dst; a.B = d->B; a.H = d->H; a.W = d->W; a.Q = d->Q; a.QP = d->QP;
  a.lds = d->lds; a.ldd = d->ldd; a.ntaps = d->ntaps; a.M = (long)d->B * d->H * d->W;
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) { a.dy[i] = d->dy[i]; a.dx[i] = d->dx[i]; }
  hipLaunchKernelGGL(tap_gather_sum_kernel, dim3(tap_grid(a.M * (a.QP / 4))), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

extern "C" int simt_tap_scatter(const simt_tap_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->src && d->dst && d->ntaps >= 1 && d->ntaps <= SIMT_MAX_TAPS);
  SIMT_CHECK(d->QP % 8 == 0 && d->lds % 8 == 0 && d->ldd % 8 == 0 && d->QP <= d->lds && d->ntaps * d->QP <= d->ldd);
  TapArgs a;
  a.src = d->src; a.bias = nullptr; a.dst = d->dst; a.B = d->B; a.H = d->H; a.W = d->W; a.Q = d->Q; a.QP = d->QP;
  a.lds = d->lds; a.ldd = d->ldd; a.ntaps = d->ntaps; a.M = (long)d->B * d->H * d->W;
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) { a.dy[i] = d->dy[i]; a.dx[i] = d->dx[i]; }
  hipLaunchKernelGGL(tap_scatter_kernel, dim3(tap_grid(a.M * a.ntaps * (a.QP / 8))), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
