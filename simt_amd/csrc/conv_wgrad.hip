// Weight-gradient (wgrad) implicit GEMM for gfx950.
//
// Replaces the autograd weight-gradient of every nn.Conv2d on the training path
// (reference tools/trainV2_simt.py:428 loss.backward() through model/deeplab_multi.py:62,68,73,110,127,156).
//
// GEMM view:  dW[co][k] = sum_m dY[m][co] * Xs[m][k],   k = (tap, cin), Xs = tap-shifted input pixel.
// The reduction runs over PIXELS, which is the slow (row) dimension of both NHWC operands, so both MFMA
// operands need a transposed fragment.  bf16: tiles are staged [pixel][128 ch] with global_load_lds and
// read back with ds_read_b64_tr_b16 (hardware transpose, 4 pixels x 16 channels per 16-lane group);
// the 16-B chunk index is XOR-swizzled by h(pixel)<<1 on the source address so that the 8 pixel rows a
// 32-lane half touches land in 8 distinct 32-B slots (conflict-free).  f32 (parity mode): the
// 16x16x4 f32 MFMA takes one k per lane group, so plain ds_read_b32 of [pixel][ch] is already the fragment.
// Tile: 128 (co) x 128 (k) outputs, 64 (bf16) / 32 (f32) pixels per stage, 4 waves (2x2, 64x64 each),
// split over pixel ranges (split-K); partial tiles go to an f32 slab [split][Cout][Ktot] that
// simt_wgrad_reduce sums in fixed order (bitwise reproducible) into the OIHW fp32 gradient.
#include "common.h"
#include <stdio.h>

struct WgradKArgs {
  const char* dy;
  const char* x;
  float* slab;
  const char* zero;
  int B, H, W, Cin, Ho, Wo, Cd, ldd, stride, ntaps, M, Ktot;
  int nsplit, pix_per_split, cotiles, ktiles;
  float rcpWo, rcpHoWo;
  short tdy[SIMT_MAX_TAPS], tdx[SIMT_MAX_TAPS];
};

template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradKArgs a) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int BP = BF ? 64 : 32;            // pixels per stage
  constexpr int ROWB = 128 * sizeof(T);       // bytes per LDS row (128 channels)
  constexpr int CPR = ROWB / 16;              // 16-B chunks per row
  constexpr int RPI = 256 / CPR;              // rows covered by one load iteration
  constexpr int TILE_BYTES = BP * ROWB;       // 16 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sD = smem;                   // 2 x TILE_BYTES (dY tile)
  char* sX = smem + 2 * TILE_BYTES;  // 2 x TILE_BYTES (X tile)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  int bid = blockIdx.x;
  const int split = bid % a.nsplit;
  bid /= a.nsplit;
  const int kt_ = bid % a.ktiles, ct = bid / a.ktiles;
  const int co0 = ct * 128, k0 = kt_ * 128;

  // my chunk position and (swizzled) source chunk; both constant over the whole kernel
  const int c_pos = tid % CPR;
  const int row_in_iter = tid / CPR;
  int h = 0;
  if (BF) h = ((row_in_iter & 3) | (((row_in_iter >> 3) & 1) << 2)) << 1;
  const int cg = c_pos ^ h;
  // dY source channel
  const int dch = co0 + cg * (16 / (int)sizeof(T));
  const bool d_ok = dch < a.Cd;
  // X source (tap, ci)
  const int kk = k0 + cg * (16 / (int)sizeof(T));
  const bool k_ok = kk < a.Ktot;
  int tap = 0, ci = 0, tdy = 0, tdx = 0;
  if (k_ok) {
    tap = kk / a.Cin;
    ci = kk - tap * a.Cin;
    tdy = a.tdy[tap];
    tdx = a.tdx[tap];
  }
  const long pix_bytes = (long)a.Cin * sizeof(T);
  const long ldd_bytes = (long)a.ldd * sizeof(T);
  const int HoWo = a.Ho * a.Wo;

  const int m_begin = split * a.pix_per_split;
  int m_end = m_begin + a.pix_per_split;
  if (m_end > a.M) m_end = a.M;
  const int nstage = (m_end > m_begin) ? (m_end - m_begin + BP - 1) / BP : 0;

  auto stage = [&](int st, int buf) {
    int mb = m_begin + st * BP;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m = mb + i * RPI + row_in_iter;
      bool mok = m < m_end;
      const char* srcd = a.zero + c_pos * 16;
      const char* srcx = a.zero + c_pos * 16;
      if (mok) {
        if (d_ok) srcd = a.dy + (long)m * ldd_bytes + (long)dch * sizeof(T);
        if (k_ok) {
          int b, r, oy, ox;
          fast_divmod(m, HoWo, a.rcpHoWo, b, r);
          fast_divmod(r, a.Wo, a.rcpWo, oy, ox);
          int iy = oy * a.stride + tdy, ix = ox * a.stride + tdx;
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            srcx = a.x + (((long)b * a.H + iy) * a.W + ix) * pix_bytes + (long)ci * sizeof(T);
        }
      }
      int ldsoff = buf * TILE_BYTES + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(GPTR(srcd), LPTR(sD + ldsoff), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(srcx), LPTR(sX + ldsoff), 16, 0, 0);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nstage > 0) stage(0, 0);
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + 1 < nstage) stage(st + 1, buf ^ 1);
    const char* pd = sD + buf * TILE_BYTES;
    const char* px = sX + buf * TILE_BYTES;
    if constexpr (BF) {
      // lane l: g = l>>4 (k group), i = l&15, q = i>>2 (row in 4-row block), pp = i&3 (4-col group)
      const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
      const int hh = (q | ((g & 1) << 2)) << 1;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int row1 = ks * 32 + 8 * g + q;
        bf16x8 af[4], bfr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          int cba = wm * 64 + t * 16, cbb = wn * 64 + t * 16;
          int cha = ((cba >> 3) + (pp >> 1)) ^ hh;
          int chb = ((cbb >> 3) + (pp >> 1)) ^ hh;
          int offa = row1 * ROWB + cha * 16 + (pp & 1) * 8;
          int offb = row1 * ROWB + chb * 16 + (pp & 1) * 8;
          bf4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)(pd + offa));
          bf4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)(pd + offa + 4 * ROWB));
          bf4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)(px + offb));
          bf4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf4v*)(px + offb + 4 * ROWB));
          bf16x4 a0s = __builtin_bit_cast(bf16x4, a0), a1s = __builtin_bit_cast(bf16x4, a1);
          bf16x4 b0s = __builtin_bit_cast(bf16x4, b0), b1s = __builtin_bit_cast(bf16x4, b1);
          af[t] = (bf16x8){a0s[0], a0s[1], a0s[2], a0s[3], a1s[0], a1s[1], a1s[2], a1s[3]};
          bfr[t] = (bf16x8){b0s[0], b0s[1], b0s[2], b0s[3], b1s[0], b1s[1], b1s[2], b1s[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    } else {
      const int kq = lane >> 4, cl = lane & 15;
#pragma unroll
      for (int ks = 0; ks < BP / 4; ++ks) {
        const int row = ks * 4 + kq;
        float af[4], bfr[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          af[t] = *(const float*)(pd + row * ROWB + (wm * 64 + t * 16 + cl) * 4);
          bfr[t] = *(const float*)(px + row * ROWB + (wn * 64 + t * 16 + cl) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  }

  // slab[split][co][k]
  float* out = a.slab + (long)split * a.Cd * a.Ktot;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int co = co0 + wm * 64 + i * 16 + (lane >> 4) * 4 + e;
        int k = k0 + wn * 64 + j * 16 + (lane & 15);
        if (co < a.Cd && k < a.Ktot) out[(long)co * a.Ktot + k] = acc[i][j][e];
      }
}

// dst (OIHW fp32) [co][ci][r][s]  (+)=  scale * sum_split slab[split][co_off+co][(tap_off + r*S+s)*Cin + ci]
__global__ void wgrad_reduce_kernel(const float* slab, float* dst, int nsplit, int Cd, int Ktot, int Cin,
                                    int co_off, int tap_off, int Cout, int RS, int accumulate, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  // idx enumerates the slab side (co, t, ci) so reads are coalesced; writes are strided by RS (small)
  int ci = idx % Cin;
  long r = idx / Cin;
  int t = r % RS;
  int co = r / RS;
  long src = (long)(co_off + co) * Ktot + (long)(tap_off + t) * Cin + ci;
  long sstride = (long)Cd * Ktot;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += slab[k * sstride + src];
  long d = ((long)co * Cin + ci) * RS + t;
  dst[d] = accumulate ? dst[d] + s : s;
}
// 3x3 (RS = 9): one thread owns FOUR input channels x all nine taps of one output channel -- nine coalesced 16-byte slab reads per split
// and 36 CONSECUTIVE floats of the OIHW gradient (nine 16-byte stores) instead of four 4-byte stores 36 bytes apart per tap.  Per
// element the same sum in the same order (split 0, 1, ...).
__device__ __forceinline__ void wgrad_reduce9_body(const float* __restrict__ slab, float* __restrict__ dst, int nsplit, int Cd, int Ktot, int Cin,
                                                    int co_off, int tap_off, int accumulate, long idx) {
  const int c4 = Cin >> 2;
  const int ci = (int)(idx % c4) << 2;
  const int co = (int)(idx / c4);
  const float* p = slab + (long)(co_off + co) * Ktot + (long)tap_off * Cin + ci;
  const long sstride = (long)Cd * Ktot;
  float4 s[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) s[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int k = 0; k < nsplit; ++k) {
    float4 v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) v[t] = *(const float4*)(p + k * sstride + (long)t * Cin);
#pragma unroll
    for (int t = 0; t < 9; ++t) { s[t].x += v[t].x; s[t].y += v[t].y; s[t].z += v[t].z; s[t].w += v[t].w; }
  }
  // out[c * 9 + t] = s[t].{c}: 36 consecutive floats starting at ((co * Cin + ci) * 9)
  float o[36];
#pragma unroll
  for (int t = 0; t < 9; ++t) { o[t] = s[t].x; o[9 + t] = s[t].y; o[18 + t] = s[t].z; o[27 + t] = s[t].w; }
  float4* q = (float4*)(dst + ((long)co * Cin + ci) * 9);       // 16-byte aligned: (co * Cin + ci) % 4 == 0
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    float4 w = make_float4(o[4 * j], o[4 * j + 1], o[4 * j + 2], o[4 * j + 3]);
    if (accumulate) { const float4 old = q[j]; w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w; }
    q[j] = w;
  }
}
// threads a reduce needs (four elements each; 3x3: 36 each)
__device__ __forceinline__ long wgrad_reduce_threads_d(int Cout, int RS, int Cin) { return RS == 9 ? (long)Cout * (Cin / 4) : (long)Cout * RS * (Cin / 4); }
static inline long wgrad_reduce_threads(int Cout, int RS, int Cin) { return RS == 9 ? (long)Cout * (Cin / 4) : (long)Cout * RS * (Cin / 4); }
extern "C" int simt_wgrad_reduce_blocks(int Cout, int RS, int Cin) { return (int)((wgrad_reduce_threads(Cout, RS, Cin) + 255) / 256); }

// Same sums, same order (split 0, 1, ... per element), four consecutive ci per thread: 16-byte slab reads instead of 4-byte ones
// (the scalar kernel reached ~2 TB/s on 32 MB of slabs).  Needs Cin % 4 == 0 and 16-byte aligned slab rows (Ktot % 4 == 0).
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* slab, float* dst, int nsplit, int Cd, int Ktot, int Cin,
                                                            int co_off, int tap_off, int Cout, int RS, int accumulate, long total4) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total4) return;
  if (RS == 9) { wgrad_reduce9_body(slab, dst, nsplit, Cd, Ktot, Cin, co_off, tap_off, accumulate, idx); return; }
  const int c4 = Cin >> 2;
  int ci = (int)(idx % c4) << 2;
  long r = idx / c4;
  int t = r % RS;
  int co = r / RS;
  const float* p = slab + (long)(co_off + co) * Ktot + (long)(tap_off + t) * Cin + ci;
  const long sstride = (long)Cd * Ktot;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int k = 0; k < nsplit; ++k) {
    const float4 v = *(const float4*)(p + k * sstride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const long d = ((long)co * Cin + ci) * RS + t;
  if (RS == 1) {
    float4* q = (float4*)(dst + d);
    if (accumulate) { const float4 o = *q; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *q = s;
  } else {
    const float e[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[d + (long)j * RS] = accumulate ? dst[d + (long)j * RS] + e[j] : e[j];
  }
}

// Several reduces in one launch (the slabs of a grouped weight-gradient launch): job j owns blocks [block0_j, block0_{j+1}); per element
// the same sum in the same order as wgrad_reduce4_kernel.
__global__ __launch_bounds__(256) void wgrad_reduce4_multi_kernel(const simt_wgrad_reduce_job* __restrict__ jobs, int njobs) {
  int j = 0;
  for (int i = 1; i < njobs; ++i) if ((int)blockIdx.x >= jobs[i].block0) j = i;
  const simt_wgrad_reduce_job& a = jobs[j];
  const int Cin = a.Cin, RS = a.RS, Ktot = a.Ktot, nsplit = a.nsplit;
  const long idx = (long)((int)blockIdx.x - a.block0) * 256 + threadIdx.x;
  const int c4 = Cin >> 2;
  if (idx >= wgrad_reduce_threads_d(a.Cout, RS, Cin)) return;
  if (RS == 9) { wgrad_reduce9_body(a.slab, a.dst, nsplit, a.Cd, Ktot, Cin, a.co_off, a.tap_off, a.accumulate, idx); return; }
  const int ci = (int)(idx % c4) << 2;
  const long r = idx / c4;
  const int t = (int)(r % RS);
  const int co = (int)(r / RS);
  const float* p = a.slab + (long)(a.co_off + co) * Ktot + (long)(a.tap_off + t) * Cin + ci;
  const long sstride = (long)a.Cd * Ktot;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int k = 0; k < nsplit; ++k) {
    const float4 v = *(const float4*)(p + k * sstride);
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  float* dst = a.dst;
  const long d = ((long)co * Cin + ci) * RS + t;
  if (RS == 1) {
    float4* q = (float4*)(dst + d);
    if (a.accumulate) { const float4 o = *q; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *q = s;
  } else {
    const float e[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dst[d + (long)jj * RS] = a.accumulate ? dst[d + (long)jj * RS] + e[jj] : e[jj];
  }
}

extern "C" int simt_wgrad_reduce_multi(const simt_wgrad_reduce_job* jobs_dev, int n, int blocks, simt_stream_t stream) {
  SIMT_CHECK(jobs_dev && n >= 1 && n <= 2 * SIMT_WGRAD_MULTI_MAX && blocks >= 1);
  hipLaunchKernelGGL(wgrad_reduce4_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

int simt_conv_wgrad_bf16_v2(const simt_wgrad_desc* d, simt_stream_t stream);  // conv_wgrad2.hip

// Which problems run on the 128x256-tile kernel (conv_wgrad2.hip; also the condition for a grouped launch, simt_conv_wgrad_multi)
bool simt_conv_wgrad_v2_eligible(const simt_wgrad_desc* d) {
  static int min_cd = -1, min_k = 0;
  if (min_cd < 0) {
    const char* e = getenv("SIMT_WGRAD2_MIN");          // "cd,k" thresholds of the 128x256-tile kernel (experiments)
    min_cd = 64; min_k = 64;     // partial tiles are zero-filled; even at Cd = K = 64 it beats the 128x128 kernel (44 vs 59 us)
    if (e) sscanf(e, "%d,%d", &min_cd, &min_k);
  }
  return d->dtype == SIMT_BF16 && d->Cd >= min_cd && d->ntaps * d->Cin >= min_k && (d->stride != 1 || (d->H == d->Ho && d->W == d->Wo));
}

extern "C" int simt_conv_wgrad(const simt_wgrad_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->dy && d->x && d->slab);
  SIMT_CHECK(d->ntaps >= 1 && d->ntaps <= SIMT_MAX_TAPS);
  const int esz = d->dtype == SIMT_BF16 ? 2 : 4;
  const int epc = 16 / esz;
  SIMT_CHECK(d->Cin % epc == 0 && d->Cd % epc == 0 && d->ldd % epc == 0 && d->Cd <= d->ldd);
  SIMT_CHECK(d->nsplit >= 1);
  if (simt_conv_wgrad_v2_eligible(d)) return simt_conv_wgrad_bf16_v2(d, stream);
  WgradKArgs k;
  k.dy = (const char*)d->dy; k.x = (const char*)d->x; k.slab = d->slab; k.zero = (const char*)simt_zero_page();
  k.B = d->B; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Ho = d->Ho; k.Wo = d->Wo; k.Cd = d->Cd; k.ldd = d->ldd;
  k.stride = d->stride; k.ntaps = d->ntaps; k.M = d->B * d->Ho * d->Wo; k.Ktot = d->ntaps * d->Cin;
  SIMT_CHECK(k.M < (1 << 24));
  const int BP = d->dtype == SIMT_BF16 ? 64 : 32;
  k.nsplit = d->nsplit;
  int pps = (k.M + k.nsplit - 1) / k.nsplit;
  k.pix_per_split = ((pps + BP - 1) / BP) * BP;
  k.cotiles = (d->Cd + 127) / 128;
  k.ktiles = (k.Ktot + 127) / 128;
  k.rcpWo = 1.0f / (float)d->Wo;
  k.rcpHoWo = 1.0f / (float)(d->Ho * d->Wo);
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) { k.tdy[i] = d->dy_[i]; k.tdx[i] = d->dx_[i]; }
  hipStream_t st = (hipStream_t)stream;
  int grid = k.cotiles * k.ktiles * k.nsplit;
  size_t lds = 4 * 16384;
  if (d->dtype == SIMT_BF16)
    hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), lds, st, k);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<float>, dim3(grid), dim3(256), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

extern "C" int simt_wgrad_reduce(const float* slab, float* dst, int nsplit, int Cd, int Ktot, int Cin, int co_off,
                                 int tap_off, int Cout, int RS, int accumulate, simt_stream_t stream) {
  SIMT_CHECK(slab && dst && nsplit >= 1);
  long total = (long)Cout * RS * Cin;
  if (Cin % 4 == 0 && Ktot % 4 == 0 && ((uintptr_t)slab & 15) == 0 && ((uintptr_t)dst & 15) == 0) {
    const long total4 = wgrad_reduce_threads(Cout, RS, Cin);
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slab, dst, nsplit,
                       Cd, Ktot, Cin, co_off, tap_off, Cout, RS, accumulate, total4);
    SIMT_LAUNCH_CHECK();
    return SIMT_OK;
  }
  int grid = (int)((total + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, dst, nsplit, Cd, Ktot,
                     Cin, co_off, tap_off, Cout, RS, accumulate, total);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// Tap-expanded head: slab[split][(tap_off+t)*QP + row_off + co][ci]  ->  dst (OIHW fp32) [co][ci][t]
__global__ void wgrad_reduce_exp_kernel(const float* slab, float* dst, int nsplit, int Cd, int Cin, int QP, int row_off,
                                        int tap_off, int Cout, int RS, long total) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int ci = idx % Cin;
  long r = idx / Cin;
  int t = r % RS;
  int co = r / RS;
  long src = ((long)(tap_off + t) * QP + row_off + co) * Cin + ci;
  long sstride = (long)Cd * Cin;
  float s = 0.f;
  for (int k = 0; k < nsplit; ++k) s += slab[k * sstride + src];
  dst[((long)co * Cin + ci) * RS + t] = s;
}
extern "C" int simt_wgrad_reduce_exp(const float* slab, float* dst, int nsplit, int Cd, int Cin, int QP, int row_off,
                                     int tap_off, int Cout, int RS, simt_stream_t stream) {
  SIMT_CHECK(slab && dst && nsplit >= 1);
  long total = (long)Cout * RS * Cin;
  hipLaunchKernelGGL(wgrad_reduce_exp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, slab,
                     dst, nsplit, Cd, Cin, QP, row_off, tap_off, Cout, RS, total);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
