// Implicit-GEMM convolution for gfx950 (fprop and dgrad share this kernel).
//
// Replaces the nn.Conv2d calls of the reference trunk and heads:
//   reference model/deeplab_multi.py:62,68,73 (Bottleneck 1x1 / dilated 3x3 / 1x1),
//   :110,116-119 (Classifier_Module dilated 3x3 branches, summed), :127 (stem, via im2col),
//   :156 (downsample 1x1) and their autograd dgrads.
//
// GEMM view:  Y[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = output pixel (b,oy,ox), NHWC;  k = (tap, cin);  n = cout.
//   A is never materialised: each 16-byte K-chunk of a row is fetched straight into LDS with
//   global_load_lds (per-lane source address = the shifted input pixel, or a zero page when the tap
//   falls outside the image).  Wt is the packed weight [Npad][ntaps*Cin] (K contiguous).
//
// Tile: 128 (pixels) x BN (couts) x 128 bytes of K per stage (64 bf16 / 32 f32), 256 threads = 4 waves,
// double-buffered LDS, one barrier per K-stage (loads of stage k+1 fly under the MFMAs of stage k).
// LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the SOURCE address and on
// the ds_read_b128 side, so the 16-lane groups of ds_read_b128 hit 16 distinct 16-B slots.
// MFMA: v_mfma_f32_16x16x32_bf16 (bf16) or v_mfma_f32_16x16x4_f32 (f32 parity mode; exact fmaf chain).
// Epilogue: accumulators -> LDS tile -> (bias, residual, ReLU) -> 16-B vector stores; optional
// per-channel sum / sum-of-squares partials for train-mode BatchNorm (deterministic, one slot per m-tile).
#include "common.h"
#include <stdlib.h>

__device__ __attribute__((aligned(4096))) char g_simt_zero_page[4096];

const void* simt_zero_page(void) {
  static void* p = nullptr;
  if (!p) {
    (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_simt_zero_page));
  }
  return p;
}

struct ConvKArgs {
  const char* x;
  const char* w;
  void* y;
  const float* bias;
  const void* res;
  float* stats;
  const char* zero;
  const void* mask;
  const unsigned char* res_bits;
  int ldm;
  int B, H, W, Cin, Ho, Wo, Cout, Nstore, ldy, ldr, stride, ntaps, relu, M;
  int kc_per_tap;  // Cin*sizeof(T)/128
  int ntiles_n, ntiles_m;
  short dy[SIMT_MAX_TAPS], dx[SIMT_MAX_TAPS];
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[j]), __uint_as_float(b[j]), c, 0, 0, 0);
  }
};

template <typename T, typename TO, int BN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvKArgs a) {
  constexpr int BM = 128;
  constexpr int WN = (BN == 128) ? 2 : 1;
  constexpr int WM = 4 / WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  constexpr int B_ITERS = BN / 32;
  constexpr int CP = BN + 4;  // epilogue tile pitch (floats)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                // 2 x A_BYTES
  char* sB = smem + 2 * A_BYTES;  // 2 x B_BYTES

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int nwg = a.ntiles_m * a.ntiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int mt = tile / a.ntiles_n, nt = tile - mt * a.ntiles_n;
  const int m0 = mt * BM, n0 = nt * BN;

  const int esz = sizeof(T);
  const long pix_bytes = (long)a.Cin * esz;
  const long wrow_bytes = (long)a.ntaps * a.Cin * esz;

  // Per-thread A rows: q = i*256 + tid -> row = q>>3, chunk position c = q&7
  const int c_pos = tid & 7;
  int row_iy[4], row_ix[4];
  long row_base[4];  // byte offset of image b (or -1 if row beyond M)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int row = i * 32 + (tid >> 3);
    int m = m0 + row;
    if (m < a.M) {
      int hw = a.Ho * a.Wo;
      int b = m / hw;
      int r = m - b * hw;
      int oy = r / a.Wo;
      int ox = r - oy * a.Wo;
      row_iy[i] = oy * a.stride;
      row_ix[i] = ox * a.stride;
      row_base[i] = (long)b * a.H * a.W;
    } else {
      row_iy[i] = -100000;
      row_ix[i] = -100000;
      row_base[i] = 0;
    }
  }
  // swizzled source chunk for my LDS position: row>>1 & 7 with row = i*32 + (tid>>3)
  const int a_cg = c_pos ^ (((tid >> 3) >> 1) & 7);
  // B rows: q = i*256 + tid -> n row = q>>3
  const char* b_src[B_ITERS];
#pragma unroll
  for (int i = 0; i < B_ITERS; ++i) {
    int row = i * 32 + (tid >> 3);
    b_src[i] = a.w + (long)(n0 + row) * wrow_bytes + a_cg * 16;
  }

  auto stage = [&](int kt, int buf) {
    int tap = kt / a.kc_per_tap;
    int kc = kt - tap * a.kc_per_tap;
    int tdy = a.dy[tap], tdx = a.dx[tap];
    long koff = (long)kc * 128 + a_cg * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int iy = row_iy[i] + tdy, ix = row_ix[i] + tdx;
      bool ok = (iy >= 0) && (iy < a.H) && (ix >= 0) && (ix < a.W);
      const char* src = ok ? a.x + (row_base[i] + (long)iy * a.W + ix) * pix_bytes + koff : a.zero + a_cg * 16;
      char* dst = sA + buf * A_BYTES + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(dst), 16, 0, 0);
    }
    long wk = (long)kt * 128;
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      char* dst = sB + buf * B_BYTES + (i * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(GPTR(b_src[i] + wk), LPTR(dst), 16, 0, 0);
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = a.ntaps * a.kc_per_tap;
  const int sw = (lane >> 1) & 7;
  const int frag_row_off = (lane & 15) * 128;
  const int kq = lane >> 4;

  stage(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) stage(kt + 1, buf ^ 1);
    const char* pa = sA + buf * A_BYTES + (wm * TM * 16) * 128 + frag_row_off;
    const char* pb = sB + buf * B_BYTES + (wn * TN * 16) * 128 + frag_row_off;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int coff = ((4 * s + kq) ^ sw) << 4;
      u32x4 af[TM], bfr[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const u32x4*)(pa + i * 16 * 128 + coff);
#pragma unroll
      for (int j = 0; j < TN; ++j) bfr[j] = *(const u32x4*)(pb + j * 16 * 128 + coff);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(af[i], bfr[j], acc[i][j]);
    }
  }

  // ---------------- epilogue ----------------
  __syncthreads();
  float* sC = (float*)smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int r = wm * TM * 16 + i * 16 + (lane >> 4) * 4 + e;
        int c = wn * TN * 16 + j * 16 + (lane & 15);
        sC[r * CP + c] = acc[i][j][e];
      }
  __syncthreads();

  if (a.stats) {
    // column sums over the tile's 128 rows (rows >= M hold exact zeros)
    constexpr int PARTS = 256 / BN;
    constexpr int RP = BM / PARTS;
    float* sS = sC + BM * CP;  // [2][256]
    int col = tid % BN, part = tid / BN;
    float s1 = 0.f, s2 = 0.f;
    for (int r = part * RP; r < (part + 1) * RP; ++r) {
      float v = sC[r * CP + col];
      s1 += v;
      s2 += v * v;
    }
    sS[tid] = s1;
    sS[256 + tid] = s2;
    __syncthreads();
    if (tid < BN) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int p = 0; p < PARTS; ++p) {
        t1 += sS[p * BN + tid];
        t2 += sS[256 + p * BN + tid];
      }
      int n = n0 + tid;
      if (n < a.Cout) {
        a.stats[((long)mt * 2 + 0) * a.Cout + n] = t1;
        a.stats[((long)mt * 2 + 1) * a.Cout + n] = t2;
      }
    }
  }

  constexpr int VPR = BN / 8;        // 8-wide vectors per row
  constexpr int RPP = 256 / VPR;     // rows per pass
  const int vcol = (tid % VPR) * 8;
  const int n = n0 + vcol;
  if (n < a.Nstore) {
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (a.bias && (n + e) < a.Cout) ? a.bias[n + e] : 0.f;
    for (int r = tid / VPR; r < BM; r += RPP) {
      int m = m0 + r;
      if (m >= a.M) break;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = sC[r * CP + vcol + e] + bias8[e];
      if (a.res) {
        float rv[8];
        load8((const T*)a.res + (long)m * a.ldr + n, rv);
        if (a.res_bits) {
          const unsigned b = a.res_bits[((long)m * a.ldr + n) >> 3];
#pragma unroll
          for (int e = 0; e < 8; ++e) rv[e] = ((b >> e) & 1u) ? rv[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if (a.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      if (a.mask) {
        float mv[8];
        load8((const T*)a.mask + (long)m * a.ldm + n, mv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
      }
      store8((TO*)a.y + (long)m * a.ldy + n, v);
    }
  }
}

template <typename T, typename TO, int BN>
static int launch_conv(const ConvKArgs& k, hipStream_t st) {
  size_t stage_bytes = 2 * (128 * 128 + BN * 128);
  size_t epi_bytes = (size_t)128 * (BN + 4) * 4 + 2 * 256 * 4;
  size_t lds = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm_kernel<T, TO, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int nwg = k.ntiles_m * k.ntiles_n;
  hipLaunchKernelGGL((conv_igemm_kernel<T, TO, BN>), dim3(nwg), dim3(256), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

int simt_conv_fprop_bf16_v2(const simt_conv_desc* d, simt_stream_t stream);  // conv_igemm2.hip

extern "C" int simt_conv_fprop(const simt_conv_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->x && d->w && d->y);
  SIMT_CHECK(d->ntaps >= 1 && d->ntaps <= SIMT_MAX_TAPS);
  const int esz = d->dtype_in == SIMT_BF16 ? 2 : 4;
  SIMT_CHECK((d->Cin * esz) % 128 == 0);          // K-stage = 128 B of one tap
  const bool v2 = d->dtype_in == SIMT_BF16 && d->tile_n >= 64 &&
                  (d->dtype_out == SIMT_BF16 || (!d->bias && !d->res && !d->relu && !d->stats && !d->mask && d->tile_n == 256));
  SIMT_CHECK(d->tile_n == 128 || d->tile_n == 64 || d->tile_n == 32 || (v2 && d->tile_n == 256));
  SIMT_CHECK(d->Npad % d->tile_n == 0 && d->Npad >= d->Cout);
  SIMT_CHECK(d->Nstore % 8 == 0 && d->Nstore <= d->Npad && d->Nstore <= d->ldy);
  SIMT_CHECK(d->ldy % 8 == 0 && (!d->res || d->ldr % 8 == 0) && (!d->mask || d->ldm % 8 == 0));
  SIMT_CHECK(!(d->dtype_in == SIMT_F32 && d->dtype_out == SIMT_BF16));
  if (v2) return simt_conv_fprop_bf16_v2(d, stream);
  SIMT_CHECK(!d->bnr_mode);   // the fused BN-backward reduce exists in the bf16 v2 kernel only
  SIMT_CHECK(!d->in_scale && !d->in_shift && !d->in_out);      // ... and so does the operand-path BatchNorm (simt_conv_inbn_ok)
  ConvKArgs k;
  k.x = (const char*)d->x; k.w = (const char*)d->w; k.y = d->y; k.bias = d->bias; k.res = d->res;
  k.stats = d->stats; k.zero = (const char*)simt_zero_page();
  k.mask = d->mask; k.ldm = d->ldm; k.res_bits = d->res_bits;
  k.B = d->B; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout;
  k.Nstore = d->Nstore; k.ldy = d->ldy; k.ldr = d->ldr; k.stride = d->stride; k.ntaps = d->ntaps;
  k.relu = d->relu; k.M = d->B * d->Ho * d->Wo;
  k.kc_per_tap = d->Cin * esz / 128;
  k.ntiles_n = d->Npad / d->tile_n;
  k.ntiles_m = (k.M + 127) / 128;
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) { k.dy[i] = d->dy[i]; k.dx[i] = d->dx[i]; }
  hipStream_t st = (hipStream_t)stream;
  const bool bf = d->dtype_in == SIMT_BF16, obf = d->dtype_out == SIMT_BF16;
#define DISPATCH(BN)                                                           \
  if (bf && obf) return launch_conv<bf16_t, bf16_t, BN>(k, st);                \
  if (bf && !obf) return launch_conv<bf16_t, float, BN>(k, st);                \
  return launch_conv<float, float, BN>(k, st);
  if (d->tile_n == 128) { DISPATCH(128) }
  if (d->tile_n == 64) { DISPATCH(64) }
  DISPATCH(32)
#undef DISPATCH
}
