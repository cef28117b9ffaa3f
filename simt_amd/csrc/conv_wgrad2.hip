// Weight-gradient implicit GEMM, bf16 throughput kernel -- second generation (same contract as conv_wgrad.hip:
// simt_wgrad_desc -> fp32 slabs [split][Cd][Ktot] summed in fixed order by simt_wgrad_reduce; reference
// tools/trainV2_simt.py:428 through model/deeplab_multi.py:62,68,73,156).
//
// Same skeleton as conv_igemm2.hip: 8 waves, one workgroup per CU, 3-deep global_load_lds ring with counted vmcnt and
// one raw barrier per stage, the two waves of a SIMD staggered (load-then-multiply vs multiply-then-load).
// Output tile 128 (dY channels) x 256 (tap*cin columns), reduction over 64 pixels per stage.  Both MFMA operands are
// pixel-major in memory, so each stage holds three [64 px][128 ch] images (dY, X columns 0-127, X columns 128-255) that
// are read back transposed with ds_read_b64_tr_b16; the 16-B chunk index is XOR-swizzled by h(pixel)<<1 on the global
// source address so the 8 pixel rows a 32-lane half touches land in distinct 32-B slots.
// Stride-1 convolutions need no per-stage division for the tap-shifted source pixel: input pixel = m + dy*W + dx.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

struct Wgrad2KArgs {
  const char* dy;
  const char* x;
  float* slab;
  const char* zero;
  int H, W, Cin, Ho, Wo, Cd, ldd, stride, ntaps, M, Ktot;
  int nsplit, pix_per_split, cotiles, ktiles;
  float rcpWo, rcpHoWo;
  short tdy[SIMT_MAX_TAPS], tdx[SIMT_MAX_TAPS];
};

template <int N> __device__ __forceinline__ void wg_wait_vmcnt() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

// MODE 0 = product; 1 = loads only, 2 = fragment reads + MFMA only (timing ablations, env SIMT_WGRAD2_MODE)
// Round 1 measured 3x3 256<-256 83 us = loads-only 52 + MFMA-only 43 "with little overlap": a compiler-inserted s_waitcnt vmcnt(0) in
// front of the ds_read_tr builtin drained the ring every stage.  Round 2 (profiles/tools/ab_wgrad.py): asm fragment reads 84 -> 70 us,
// incremental pixel coordinates in the issue phase -> 57 us (loads-only 42, MFMA-only 43), XCD-aware block order: 1x1 1024<-256
// 44 -> 35 us and 233 -> 124 MB of HBM traffic per launch.  A ninth wave prefetching one dword per 128-B line four stages ahead made
// the 1x1 shapes slower (62-67 us): it doubles the line requests.
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_wgrad2_kernel(Wgrad2KArgs a) {
  constexpr int NT = 512, NST = 3, BP = 64;
  constexpr int ROWB = 256;                  // bytes per LDS row (128 channels)
  constexpr int SUB = BP * ROWB;             // 16 KB image
  constexpr int STAGE = 3 * SUB;             // dY | X[0:128) | X[128:256)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // 2 x 4 waves, 64 x 64 outputs each

  // Workgroup -> (pixel split, tile): split-major linear order cut into one contiguous chunk per XCD (blocks b and b + 8 share an
  // XCD), so the tiles of one pixel split run on one XCD (two at a chunk boundary): its dY / X rows are fetched from HBM once or twice
  // per launch and re-read by the other tiles from that XCD's L2.  (Round 1's order put the tiles of a split on eight XCDs: rocprofv3
  // FETCH_SIZE 233 MB per launch against 96-115 MB algorithmic.)
  const int tiles = a.cotiles * a.ktiles;
  const int lin = xcd_remap(blockIdx.x, tiles * a.nsplit);
  const int split = lin / tiles;
  const int tix = lin - split * tiles;
  const int kt_ = tix % a.ktiles, ct = tix / a.ktiles;
  const int co0 = ct * 128, k0 = kt_ * 256;

  // chunk q = i*NT + tid of an image: row = q>>4 = i*32 + (tid>>4), position q&15
  const int c_pos = tid & 15;
  const int row_in_iter = tid >> 4;
  const int h = ((row_in_iter & 3) | (((row_in_iter >> 3) & 1) << 2)) << 1;
  const int cg = c_pos ^ h;
  const int dch = co0 + cg * 8;
  const bool d_ok = dch < a.Cd;
  int xoff[2], tdy[2], tdx[2];
  bool k_ok[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kk = k0 + s * 128 + cg * 8;
    k_ok[s] = kk < a.Ktot;
    int tap = 0, ci = 0;
    if (k_ok[s]) { tap = kk / a.Cin; ci = kk - tap * a.Cin; }
    tdy[s] = a.tdy[tap];
    tdx[s] = a.tdx[tap];
    xoff[s] = ((tdy[s] * a.W + tdx[s]) * a.Cin + ci) * 2;       // byte shift of the source for stride-1 convs
  }
  const int pix_bytes = a.Cin * 2;
  const int ldd_bytes = a.ldd * 2;
  const int HoWo = a.Ho * a.Wo;
  const bool s1 = a.stride == 1;
  const bool center[2] = {tdy[0] == 0 && tdx[0] == 0, tdy[1] == 0 && tdx[1] == 0};

  const int m_begin = split * a.pix_per_split;
  int m_end = m_begin + a.pix_per_split;
  if (m_end > a.M) m_end = a.M;
  const int nk = (m_end > m_begin) ? (m_end - m_begin + BP - 1) / BP : 0;
  const char* zsrc = a.zero + c_pos * 16;

  int ld_m = m_begin;
  // Stride-1 convolutions (all but the downsample / stem ones): the source of output pixel m under tap (dy, dx) is pixel m + dy*W + dx,
  // valid iff (oy + dy, ox + dx) is inside the image.  (oy, ox) of this thread's two rows advance by 64 pixels per stage: one division
  // before the loop instead of two fast_divmod per row and stage (the issue phase is what the partner wave's MFMA shadow has to hide).
  int r_oy[2], r_ox[2];
  unsigned r_xo[2], r_do[2];                                   // running byte offsets of the row's pixel in x / dy
  // one stage = BP pixels further in the (wrapping) per-image raster: on maps of fewer than BP pixels a stage crosses several images,
  // so the advance is taken modulo Ho*Wo (then adv_y < Ho and ONE conditional subtraction below is exact for any map size)
  const int adv_r = BP % HoWo;
  const int adv_y = adv_r / a.Wo, adv_x = adv_r % a.Wo;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m_begin + i * 32 + row_in_iter;
    int b, r;
    fast_divmod(m < a.M ? m : 0, HoWo, a.rcpHoWo, b, r);
    fast_divmod(r, a.Wo, a.rcpWo, r_oy[i], r_ox[i]);
    r_xo[i] = (unsigned)m * (unsigned)pix_bytes;
    r_do[i] = (unsigned)m * (unsigned)ldd_bytes + (unsigned)(dch * 2);
  }
  const unsigned x_step = (unsigned)BP * (unsigned)pix_bytes, d_step = (unsigned)BP * (unsigned)ldd_bytes;
  auto issue = [&](int buf) {
    char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = ld_m + i * 32 + row_in_iter;
      const bool mok = m < m_end;
      const char* srcd;
      const char* srcx[2];
      if (s1) {
        srcd = (mok && d_ok) ? a.dy + r_do[i] : zsrc;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const bool in = (unsigned)(r_oy[i] + tdy[s]) < (unsigned)a.H && (unsigned)(r_ox[i] + tdx[s]) < (unsigned)a.W;
          srcx[s] = (mok && k_ok[s] && in) ? a.x + (r_xo[i] + (unsigned)xoff[s]) : zsrc;
        }
        r_do[i] += d_step; r_xo[i] += x_step;
        r_oy[i] += adv_y; r_ox[i] += adv_x;
        if (r_ox[i] >= a.Wo) { r_ox[i] -= a.Wo; r_oy[i] += 1; }
        if (r_oy[i] >= a.Ho) r_oy[i] -= a.Ho;                  // next image: the offsets simply run on
      } else {
        srcd = (mok && d_ok) ? a.dy + ((unsigned)m * (unsigned)ldd_bytes + (unsigned)(dch * 2)) : zsrc;
        int b, r, oy, ox;
        fast_divmod(m, HoWo, a.rcpHoWo, b, r);
        fast_divmod(r, a.Wo, a.rcpWo, oy, ox);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          srcx[s] = zsrc;
          const int iy = oy * a.stride + tdy[s], ix = ox * a.stride + tdx[s];
          if (mok && k_ok[s] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
            srcx[s] = a.x + ((unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)pix_bytes + (unsigned)(xoff[s] - (tdy[s] * a.W + tdx[s]) * pix_bytes));
        }
      }
      const int ldsoff = (i * NT + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds(GPTR(srcd), LPTR(sbase + ldsoff), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(srcx[0]), LPTR(sbase + SUB + ldsoff), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(srcx[1]), LPTR(sbase + 2 * SUB + ldsoff), 16, 0, 0);
    }
    ld_m += BP;
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed fragment addressing: lane l: g = l>>4 (k group), q = (l&15)>>2 (row in 4-row block), pp = l&3
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int hh = (q | ((g & 1) << 2)) << 1;
  int offa[4], offb[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int cba = wm * 64 + t * 16, cbb = (wn & 1) * 64 + t * 16;
    offa[t] = (8 * g + q) * ROWB + ((((cba >> 3) + (pp >> 1)) ^ hh) << 4) + (pp & 1) * 8;
    offb[t] = (1 + (wn >> 1)) * SUB + (8 * g + q) * ROWB + ((((cbb >> 3) + (pp >> 1)) ^ hh) << 4) + (pp & 1) * 8;
  }
  bf16x8 af[2][4], bfr[2][4];
  // The transposed fragment reads are inline asm with a hand-placed lgkmcnt wait: in front of the ds_read_tr builtin the compiler drains
  // vmcnt to 0 (it cannot tell the read from an LDS-DMA in flight), i.e. every stage waited for ALL the loads just issued -- the ring
  // never overlapped loads with MFMAs (round 1/2 measurements: product time = loads-only + MFMA-only).
  auto trd = [](unsigned addr, auto off) {
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value));
    return r;
  };
  auto load_frags = [&](int buf) {
    const unsigned base = (unsigned)(size_t)LPTR(smem + buf * STAGE);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const unsigned pa = base + (unsigned)offa[t], pb = base + (unsigned)offb[t];
      {
        const bf16x4 a0 = trd(pa, std::integral_constant<int, 0>{}), a1 = trd(pa, std::integral_constant<int, 4 * ROWB>{});
        const bf16x4 b0 = trd(pb, std::integral_constant<int, 0>{}), b1 = trd(pb, std::integral_constant<int, 4 * ROWB>{});
        af[0][t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[0][t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
      {
        const bf16x4 a0 = trd(pa, std::integral_constant<int, 32 * ROWB>{}), a1 = trd(pa, std::integral_constant<int, 36 * ROWB>{});
        const bf16x4 b0 = trd(pb, std::integral_constant<int, 32 * ROWB>{}), b1 = trd(pb, std::integral_constant<int, 36 * ROWB>{});
        af[1][t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[1][t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    }
  };
  auto frags_landed = [&]() {                                  // every fragment read has returned; the MFMAs depend on these statements
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[0][2]), "+v"(af[0][3]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[1][2]), "+v"(af[1][3]));
    asm volatile("" : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[0][2]), "+v"(bfr[0][3]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[1][2]), "+v"(bfr[1][3]));
  };
  auto mma = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);   // D = [k][co]
  };

  if (MODE != 2) {
    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
  }
  int buf = 0;
  if (wave < 4) {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wg_wait_vmcnt<6>(); else wg_wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (MODE != 1) load_frags(buf);                     // fragment reads first: their latency gates the MFMAs
      if (MODE != 2 && kt + 2 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
      if (MODE != 1) { frags_landed(); mma(); }
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wg_wait_vmcnt<6>(); else wg_wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (MODE != 1 && kt > 0) { frags_landed(); mma(); }
      if (MODE != 2 && kt + 2 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
      if (MODE != 1) load_frags(buf);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (MODE != 1 && nk > 0) { frags_landed(); mma(); }
  }

  // slab[split][co][k]: the X columns are the MFMA's row operand, so every accumulator quad is 4 consecutive k of one dY
  // channel -> one 16-byte store per quad (four lanes = 64 contiguous bytes of a slab row) instead of four 4-byte stores
  float* out = a.slab + (long)split * a.Cd * a.Ktot;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + wm * 64 + i * 16 + (lane & 15);
      const int k = k0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (co < a.Cd && k < a.Ktot) *(f32x4*)(out + (long)co * a.Ktot + k) = acc[i][j];
    }
}

// Called by simt_conv_wgrad (conv_wgrad.hip) for bf16 problems with Cd >= 128.
int simt_conv_wgrad_bf16_v2(const simt_wgrad_desc* d, simt_stream_t stream) {
  Wgrad2KArgs k;
  k.dy = (const char*)d->dy; k.x = (const char*)d->x; k.slab = d->slab; k.zero = (const char*)simt_zero_page();
  k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Ho = d->Ho; k.Wo = d->Wo; k.Cd = d->Cd; k.ldd = d->ldd;
  k.stride = d->stride; k.ntaps = d->ntaps; k.M = d->B * d->Ho * d->Wo; k.Ktot = d->ntaps * d->Cin;
  SIMT_CHECK(k.M < (1 << 24) && k.Ktot % 4 == 0);      // 16-byte slab stores
  SIMT_CHECK((long)k.M * d->ldd * 2 < (1l << 32) && (long)d->B * d->H * d->W * d->Cin * 2 < (1l << 32));
  SIMT_CHECK(d->stride == 1 ? (d->H == d->Ho && d->W == d->Wo) : true);
  k.nsplit = d->nsplit;
  const int pps = (k.M + k.nsplit - 1) / k.nsplit;
  k.pix_per_split = ((pps + 63) / 64) * 64;
  k.cotiles = (d->Cd + 127) / 128;
  k.ktiles = (k.Ktot + 255) / 256;
  k.rcpWo = 1.0f / (float)d->Wo;
  k.rcpHoWo = 1.0f / (float)(d->Ho * d->Wo);
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) { k.tdy[i] = d->dy_[i]; k.tdx[i] = d->dx_[i]; }
  const int lds = 3 * 3 * 64 * 256;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int grid = k.cotiles * k.ktiles * k.nsplit;
#ifdef SIMT_ABLATION     // timing ablations (loads only / MFMA only: MEANINGLESS outputs), never in the product library
  static const int mode = getenv("SIMT_WGRAD2_MODE") ? atoi(getenv("SIMT_WGRAD2_MODE")) : 0;
  if (mode == 1 || mode == 2) {
    (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)conv_wgrad2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (mode == 1) hipLaunchKernelGGL(conv_wgrad2_kernel<1>, dim3(grid), dim3(512), lds, (hipStream_t)stream, k);
    else hipLaunchKernelGGL(conv_wgrad2_kernel<2>, dim3(grid), dim3(512), lds, (hipStream_t)stream, k);
    SIMT_LAUNCH_CHECK();
    return SIMT_OK;
  }
#endif
  hipLaunchKernelGGL(conv_wgrad2_kernel<0>, dim3(grid), dim3(512), lds, (hipStream_t)stream, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
