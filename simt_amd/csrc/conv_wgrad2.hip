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
  int tile0;                                 // grouped launch: first tile of this problem in the group's tile list (0 otherwise)
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
// `a`: the problem (kernel argument, or one entry of a grouped launch's table in device memory -- uniform per workgroup either way);
// (split, tix): the pixel split and the output tile of this workgroup.
template <int MODE>
__device__ __forceinline__ void conv_wgrad2_body(const Wgrad2KArgs& a, const int split, const int tix) {
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
      if (co < a.Cd && k < a.Ktot) st_out16f(out + (long)co * a.Ktot + k, acc[i][j]);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// 256 (dY channels) x 256 (tap*cin columns) tile, 32 pixels per stage, 4-slot ring.
// The 128 x 256 tile above stages 48 KB per 64 pixels for 128 x 256 x 64 MACs; a CU takes in ~70 GB/s through its L2 -> LDS path
// (profiles/microbench/fillbench.hip, MI355X_MICROARCH.md "ring-gemm"), so its stage cannot go under 0.69 us while the MFMAs need
// 0.49 us: fill-bound (measured ~1.0 us).  Here a stage is 32 pixels of FOUR [32 px][128 ch] images (dY 0-127, dY 128-255, X 0-127,
// X 128-255) = 32 KB for 256 x 256 x 32 MACs: half the staged bytes per MAC, the same 32 MFMAs per wave and barrier, and a 4-slot
// ring keeps three stages (96 KB) in flight.  Each wave owns 128 channels x 64 columns: 32 accumulator quads (128 VGPRs), 12
// fragments per stage (8 dY + 4 X, one 32-pixel k-step).  Same swizzle, same transposed reads, same early / late wave stagger.
// Needs Cd % 256 == 0 to pay (layer 3 / 4 of the ResNets); everything else stays on the 128-row tile.
template <int MODE>
__device__ __forceinline__ void conv_wgrad3_body(const Wgrad2KArgs& a, const int split, const int tix) {
  constexpr int NST = 4, BP = 32;
  constexpr int ROWB = 256;                  // bytes per LDS row (128 channels)
  constexpr int SUB = BP * ROWB;             // 8 KB image
  constexpr int STAGE = 4 * SUB;             // dY[0:128) | dY[128:256) | X[0:128) | X[128:256)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;   // 2 x 4 waves, 128 channels x 64 columns each

  const int kt_ = tix % a.ktiles, ct = tix / a.ktiles;
  const int co0 = ct * 256, k0 = kt_ * 256;

  // one 16-byte chunk of every image per thread and stage: row = tid >> 4 (pixel), position tid & 15
  const int c_pos = tid & 15;
  const int row = tid >> 4;
  const int h = ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
  const int cg = c_pos ^ h;
  int dch[2], xoff[2], tdy[2], tdx[2];
  bool d_ok[2], k_ok[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    dch[s] = co0 + s * 128 + cg * 8;
    d_ok[s] = dch[s] < a.Cd;
    const int kk = k0 + s * 128 + cg * 8;
    k_ok[s] = kk < a.Ktot;
    int tap = 0, ci = 0;
    if (k_ok[s]) { tap = kk / a.Cin; ci = kk - tap * a.Cin; }
    tdy[s] = a.tdy[tap];
    tdx[s] = a.tdx[tap];
    xoff[s] = ((tdy[s] * a.W + tdx[s]) * a.Cin + ci) * 2;
  }
  const int pix_bytes = a.Cin * 2;
  const int ldd_bytes = a.ldd * 2;
  const int HoWo = a.Ho * a.Wo;
  const bool s1 = a.stride == 1;

  const int m_begin = split * a.pix_per_split;
  int m_end = m_begin + a.pix_per_split;
  if (m_end > a.M) m_end = a.M;
  const int nk = (m_end > m_begin) ? (m_end - m_begin + BP - 1) / BP : 0;
  const char* zsrc = a.zero + c_pos * 16;

  int ld_m = m_begin;
  int r_oy, r_ox;
  unsigned r_xo, r_do;
  const int adv_r = BP % HoWo;
  const int adv_y = adv_r / a.Wo, adv_x = adv_r % a.Wo;
  {
    const int m = m_begin + row;
    int b, r;
    fast_divmod(m < a.M ? m : 0, HoWo, a.rcpHoWo, b, r);
    fast_divmod(r, a.Wo, a.rcpWo, r_oy, r_ox);
    r_xo = (unsigned)m * (unsigned)pix_bytes;
    r_do = (unsigned)m * (unsigned)ldd_bytes;
  }
  const unsigned x_step = (unsigned)BP * (unsigned)pix_bytes, d_step = (unsigned)BP * (unsigned)ldd_bytes;
  auto issue = [&](int buf) {
    char* sbase = smem + buf * STAGE + wave * 1024;          // LDS-DMA: wave-uniform base, the hardware adds lane * 16
    const int m = ld_m + row;
    const bool mok = m < m_end;
    const char* srcd[2];
    const char* srcx[2];
    if (s1) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        srcd[s] = (mok && d_ok[s]) ? a.dy + (r_do + (unsigned)(dch[s] * 2)) : zsrc;
        const bool in = (unsigned)(r_oy + tdy[s]) < (unsigned)a.H && (unsigned)(r_ox + tdx[s]) < (unsigned)a.W;
        srcx[s] = (mok && k_ok[s] && in) ? a.x + (r_xo + (unsigned)xoff[s]) : zsrc;
      }
      r_do += d_step; r_xo += x_step;
      r_oy += adv_y; r_ox += adv_x;
      if (r_ox >= a.Wo) { r_ox -= a.Wo; r_oy += 1; }
      if (r_oy >= a.Ho) r_oy -= a.Ho;
    } else {
      int b, r, oy, ox;
      fast_divmod(m, HoWo, a.rcpHoWo, b, r);
      fast_divmod(r, a.Wo, a.rcpWo, oy, ox);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        srcd[s] = (mok && d_ok[s]) ? a.dy + ((unsigned)m * (unsigned)ldd_bytes + (unsigned)(dch[s] * 2)) : zsrc;
        srcx[s] = zsrc;
        const int iy = oy * a.stride + tdy[s], ix = ox * a.stride + tdx[s];
        if (mok && k_ok[s] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
          srcx[s] = a.x + ((unsigned)((b * a.H + iy) * a.W + ix) * (unsigned)pix_bytes + (unsigned)(xoff[s] - (tdy[s] * a.W + tdx[s]) * pix_bytes));
      }
    }
    if (MODE != 2) {
      __builtin_amdgcn_global_load_lds(GPTR(srcd[0]), LPTR(sbase), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(srcd[1]), LPTR(sbase + SUB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(srcx[0]), LPTR(sbase + 2 * SUB), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GPTR(srcx[1]), LPTR(sbase + 3 * SUB), 16, 0, 0);
    }
    ld_m += BP;
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // transposed fragment addressing (as above): lane l: g = l>>4 (k group), q = (l&15)>>2 (row in 4-row block), pp = l&3
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int hh = (q | ((g & 1) << 2)) << 1;
  const unsigned rowpart = (unsigned)((8 * g + q) * ROWB + (pp & 1) * 8);
  unsigned offa[8], offb[4];
#pragma unroll
  for (int t = 0; t < 8; ++t) offa[t] = (unsigned)(wm * SUB) + rowpart + (unsigned)((((t * 2) + (pp >> 1)) ^ hh) << 4);
#pragma unroll
  for (int t = 0; t < 4; ++t) offb[t] = (unsigned)((2 + (wn >> 1)) * SUB) + rowpart + (unsigned)((((((wn & 1) * 64 + t * 16) >> 3) + (pp >> 1)) ^ hh) << 4);
  bf16x8 af[8], bfr[4];
  auto trd = [](unsigned addr, auto off) {
    bf16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value));
    return r;
  };
  auto load_frags = [&](int buf) {
    const unsigned base = (unsigned)(size_t)LPTR(smem + buf * STAGE);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const bf16x4 b0 = trd(base + offb[t], std::integral_constant<int, 0>{}), b1 = trd(base + offb[t], std::integral_constant<int, 4 * ROWB>{});
      bfr[t] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const bf16x4 a0 = trd(base + offa[t], std::integral_constant<int, 0>{}), a1 = trd(base + offa[t], std::integral_constant<int, 4 * ROWB>{});
      af[t] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto frags_landed = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]));
    asm volatile("" : "+v"(bfr[0]), "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3]));
  };
  auto mma = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D = [k][co]
  };
  // stage kt has landed (this wave's pieces): up to NST - 2 younger stages (4 loads each) may still be in flight
  auto wait_stage = [&](int kt) {
    if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

#pragma unroll
  for (int s = 0; s < NST - 1; ++s) if (s < nk) issue(s);
  int buf = 0;
  if (wave < 4) {
    for (int kt = 0; kt < nk; ++kt) {
      wait_stage(kt);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (MODE != 1) load_frags(buf);
      if (kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
      if (MODE != 1) { frags_landed(); mma(); }
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      wait_stage(kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (MODE != 1 && kt > 0) { frags_landed(); mma(); }
      if (kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
      if (MODE != 1) load_frags(buf);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (MODE != 1 && nk > 0) { frags_landed(); mma(); }
  }

  float* out = a.slab + (long)split * a.Cd * a.Ktot;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = co0 + wm * 128 + i * 16 + (lane & 15);
      const int k = k0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (co < a.Cd && k < a.Ktot) st_out16f(out + (long)co * a.Ktot + k, acc[i][j]);
    }
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_wgrad2_kernel(Wgrad2KArgs a) {
  const int tiles = a.cotiles * a.ktiles;
  const int lin = xcd_remap(blockIdx.x, tiles * a.nsplit);
  const int split = lin / tiles;
  conv_wgrad2_body<MODE>(a, split, lin - split * tiles);
}

// Grouped launch: n problems (the convs of one Bottleneck: same pixels, same split count) as ONE tile list.  Per problem the output has
// only 4-18 tiles, so alone it needs 14-31 pixel splits to fill 256 CUs -- 14-31 fp32 slabs written and re-read, and ~15 us of
// launch-shaped time for 20-40 stages of work.  Together the three convs of a layer-3 block have 34 tiles: 7 splits fill the chip, each
// workgroup runs 84 stages, and the slabs shrink to a third.  Split-major order as above: the tiles of one split (of every problem)
// share an XCD.
__global__ __launch_bounds__(512, 2) void conv_wgrad2_multi_kernel(const Wgrad2KArgs* __restrict__ jobs, int njobs, int tiles) {
  const int lin = xcd_remap(blockIdx.x, (int)gridDim.x);
  const int split = lin / tiles;
  const int t = lin - split * tiles;
  int j = 0;
  for (int i = 1; i < njobs; ++i) if (t >= jobs[i].tile0) j = i;
  conv_wgrad2_body<0>(jobs[j], split, t - jobs[j].tile0);
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_wgrad3_kernel(Wgrad2KArgs a) {
  const int tiles = a.cotiles * a.ktiles;
  const int lin = xcd_remap(blockIdx.x, tiles * a.nsplit);
  const int split = lin / tiles;
  conv_wgrad3_body<MODE>(a, split, lin - split * tiles);
}
__global__ __launch_bounds__(512, 2) void conv_wgrad3_multi_kernel(const Wgrad2KArgs* __restrict__ jobs, int njobs, int tiles) {
  const int lin = xcd_remap(blockIdx.x, (int)gridDim.x);
  const int split = lin / tiles;
  const int t = lin - split * tiles;
  int j = 0;
  for (int i = 1; i < njobs; ++i) if (t >= jobs[i].tile0) j = i;
  conv_wgrad3_body<0>(jobs[j], split, t - jobs[j].tile0);
}

// Rows of dY channels per output tile: 256 (conv_wgrad3_body) when the problem has whole 256-channel tiles, else 128.
extern "C" int simt_conv_wgrad_tile_co(const simt_wgrad_desc* d) {
  static const int off = getenv("SIMT_WGRAD3") ? atoi(getenv("SIMT_WGRAD3")) == 0 : 0;         // SIMT_WGRAD3=0: the 128-row tile everywhere (A/B)
  // few pixels (DeepLabv3's stride-16 maps, M = 8 580): half the tiles means twice the splits to fill the chip -- twice the slab bytes for
  // a handful of stages per workgroup; measured 517 -> 505 images/s there, against 558 -> 582 on VGG16 at M = 33 800
  const long M = (long)d->B * d->Ho * d->Wo;
  return (!off && M >= SIMT_WGRAD3_MIN_PIXELS && d->Cd % 256 == 0 && (d->ntaps * d->Cin) % 4 == 0) ? 256 : 128;
}

static int wgrad2_fill_args(const simt_wgrad_desc* d, Wgrad2KArgs& k, int tile_co = 128) {
  k.dy = (const char*)d->dy; k.x = (const char*)d->x; k.slab = d->slab; k.zero = (const char*)simt_zero_page();
  k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Ho = d->Ho; k.Wo = d->Wo; k.Cd = d->Cd; k.ldd = d->ldd;
  k.stride = d->stride; k.ntaps = d->ntaps; k.M = d->B * d->Ho * d->Wo; k.Ktot = d->ntaps * d->Cin;
  SIMT_CHECK(k.M < (1 << 24) && k.Ktot % 4 == 0);      // 16-byte slab stores
  SIMT_CHECK((long)k.M * d->ldd * 2 < (1l << 32) && (long)d->B * d->H * d->W * d->Cin * 2 < (1l << 32));
  SIMT_CHECK(d->stride == 1 ? (d->H == d->Ho && d->W == d->Wo) : true);
  k.nsplit = d->nsplit;
  const int pps = (k.M + k.nsplit - 1) / k.nsplit;
  k.pix_per_split = ((pps + 63) / 64) * 64;
  k.cotiles = (d->Cd + tile_co - 1) / tile_co;
  k.ktiles = (k.Ktot + 255) / 256;
  k.tile0 = 0;
  k.rcpWo = 1.0f / (float)d->Wo;
  k.rcpHoWo = 1.0f / (float)(d->Ho * d->Wo);
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) { k.tdy[i] = d->dy_[i]; k.tdx[i] = d->dx_[i]; }
  return SIMT_OK;
}

static const int WGRAD2_LDS = 3 * 3 * 64 * 256;
static const int WGRAD3_LDS = 4 * 4 * 32 * 256;

bool simt_conv_wgrad_v2_eligible(const simt_wgrad_desc* d);     // conv_wgrad.hip: the dispatch rule of simt_conv_wgrad

extern "C" int simt_conv_wgrad_multi_ok(const simt_wgrad_desc* d) { return d && simt_conv_wgrad_v2_eligible(d) ? 1 : 0; }
extern "C" int simt_conv_wgrad_multi_bytes(void) { return (int)sizeof(Wgrad2KArgs); }

extern "C" int simt_conv_wgrad_multi_prepare(const simt_wgrad_desc* d, int n, void* table_host, int* grid, int* tile_co) {
  SIMT_CHECK(d && table_host && grid && tile_co && n >= 1 && n <= SIMT_WGRAD_MULTI_MAX);
  Wgrad2KArgs* k = (Wgrad2KArgs*)table_host;
  int tiles = 0, tco = 256;
  for (int i = 0; i < n; ++i) if (simt_conv_wgrad_tile_co(&d[i]) != 256) tco = 128;       // one kernel per launch: the 256-row tile only if every problem takes it
  *tile_co = tco;
  for (int i = 0; i < n; ++i) {
    SIMT_CHECK(d[i].dy && d[i].x && d[i].slab && d[i].ntaps >= 1 && d[i].ntaps <= SIMT_MAX_TAPS);
    SIMT_CHECK(d[i].Cin % 8 == 0 && d[i].Cd % 8 == 0 && d[i].ldd % 8 == 0 && d[i].Cd <= d[i].ldd);
    SIMT_CHECK(simt_conv_wgrad_v2_eligible(&d[i]) && d[i].nsplit == d[0].nsplit && d[i].nsplit >= 1);
    const int rc = wgrad2_fill_args(&d[i], k[i], tco);
    if (rc != SIMT_OK) return rc;
    k[i].tile0 = tiles;
    tiles += k[i].cotiles * k[i].ktiles;
  }
  *grid = tiles * d[0].nsplit;
  return SIMT_OK;
}

extern "C" int simt_conv_wgrad_multi(const void* table_dev, int n, int grid, int nsplit, int tile_co, simt_stream_t stream) {
  SIMT_CHECK(table_dev && n >= 1 && n <= SIMT_WGRAD_MULTI_MAX && nsplit >= 1 && grid >= nsplit && grid % nsplit == 0);
  SIMT_CHECK(tile_co == 128 || tile_co == 256);
  if (tile_co == 256) {
    static SimtLdsAttrCache attr_cache3;
    if (simt_lds_attr_needed(&attr_cache3, WGRAD3_LDS))
      (void)hipFuncSetAttribute((const void*)conv_wgrad3_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD3_LDS);
    hipLaunchKernelGGL(conv_wgrad3_multi_kernel, dim3(grid), dim3(512), WGRAD3_LDS, (hipStream_t)stream, (const Wgrad2KArgs*)table_dev, n,
                       grid / nsplit);
    SIMT_LAUNCH_CHECK();
    return SIMT_OK;
  }
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, WGRAD2_LDS))
    (void)hipFuncSetAttribute((const void*)conv_wgrad2_multi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD2_LDS);
  hipLaunchKernelGGL(conv_wgrad2_multi_kernel, dim3(grid), dim3(512), WGRAD2_LDS, (hipStream_t)stream, (const Wgrad2KArgs*)table_dev, n,
                     grid / nsplit);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// Called by simt_conv_wgrad (conv_wgrad.hip) for bf16 problems with Cd >= 64.
int simt_conv_wgrad_bf16_v2(const simt_wgrad_desc* d, simt_stream_t stream) {
  Wgrad2KArgs k;
  const int tco = simt_conv_wgrad_tile_co(d);
  { const int rc = wgrad2_fill_args(d, k, tco); if (rc != SIMT_OK) return rc; }
  if (tco == 256) {
    static SimtLdsAttrCache attr_cache3;
    if (simt_lds_attr_needed(&attr_cache3, WGRAD3_LDS))
      (void)hipFuncSetAttribute((const void*)conv_wgrad3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD3_LDS);
#ifdef SIMT_ABLATION     // timing ablations (loads only / fragment reads + MFMA only: MEANINGLESS outputs), never in the product library
    {
      static const int mode3 = getenv("SIMT_WGRAD2_MODE") ? atoi(getenv("SIMT_WGRAD2_MODE")) : 0;
      if (mode3 == 1 || mode3 == 2) {
        (void)hipFuncSetAttribute((const void*)conv_wgrad3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD3_LDS);
        (void)hipFuncSetAttribute((const void*)conv_wgrad3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD3_LDS);
        if (mode3 == 1) hipLaunchKernelGGL(conv_wgrad3_kernel<1>, dim3(k.cotiles * k.ktiles * k.nsplit), dim3(512), WGRAD3_LDS, (hipStream_t)stream, k);
        else hipLaunchKernelGGL(conv_wgrad3_kernel<2>, dim3(k.cotiles * k.ktiles * k.nsplit), dim3(512), WGRAD3_LDS, (hipStream_t)stream, k);
        SIMT_LAUNCH_CHECK();
        return SIMT_OK;
      }
    }
#endif
    hipLaunchKernelGGL(conv_wgrad3_kernel<0>, dim3(k.cotiles * k.ktiles * k.nsplit), dim3(512), WGRAD3_LDS, (hipStream_t)stream, k);
    SIMT_LAUNCH_CHECK();
    return SIMT_OK;
  }
  const int lds = WGRAD2_LDS;
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
