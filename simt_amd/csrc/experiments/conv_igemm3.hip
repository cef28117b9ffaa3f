// Implicit-GEMM convolution, bf16 throughput kernel (fprop and dgrad) for gfx950 -- third generation: ROLE-SPLIT waves.
//
// Same contract, same tile (160 or 128 pixels x 256 couts, 64-deep K stages, 3-slot global_load_lds ring, one raw barrier per stage),
// same LDS image and the same epilogue as conv_igemm2.hip; what changes is WHO does what.  Round 3 measured on conv_igemm2's stage
// (profiles/r03_conv_experiments.txt, profiles/microbench/piecebench.hip):
//   * an LDS-DMA piece costs the wave that ISSUES it 100-150 cycles of issue time inside a loaded stage, and a wave is in-order: in
//     conv_igemm2 every wave issues 6-7 pieces AND 40 MFMAs per stage, so its serial time (reads + pieces + MFMAs ~ 1 900-2 500 cycles)
//     bounds the stage whatever the partner wave does (interleaving, re-ordering, priorities: all within +-12 %);
//   * the same pieces issued by ANOTHER wave of the SIMD cost the multiplying wave nothing: 4 waves x 80 MFMAs per stage run at the
//     pipe rate (162.6 us per 288 stages) with or without four loader waves issuing all 52 pieces of the stage beside them (160.9-165).
// Hence: waves 0-3 (one per SIMD) only multiply -- a 2 x 2 grid of 80 x 128 (64 x 128) sub-tiles, 40 (32) accumulator quads in
// registers, 26 (24) fragment reads per stage instead of 2 x 18 -- and waves 4-7 only fill the ring: 13 (12) pieces each per stage,
// counted vmcnt, never a ds_read.  All eight waves stream the output tile in the epilogue.
// Round 1/2's loader-wave experiments (conv_igemm2's LW = 4 / 8) kept EIGHT multiplying waves and added loaders on top: 3-4 waves per
// SIMD, 128-170 VGPRs each, the multiplying waves' own serial time unchanged -- they measured "no gain" for that reason.
//
// RESULT (round 3, profiles/r03_conv_experiments.txt section 6): bit-identical to conv_igemm2 and within +-5 % of it on every production
// shape (3x3 256: 48.9 vs 46.5 us; 2048 -> 512: 86.4 vs 88.4; head 2048 -> 432: 83.6 vs 86.3).  Its two halves ALONE run at 35.9 us
// (loaders) and 35.6 us (multiplying waves) and together at 53 on the same box -- like conv_igemm2's 32 + 35 -> 50.  The pieces are free
// for the multiplying wave in the micro-benchmark because its source is L2-hot; in the conv the fill itself (61 GB/s per CU from beyond
// L2, 104 KB in flight per CU, the LDS holds no more) is as long as the matrix work, and matrix work beside it slows it further (the
// chip drops from 2.4 to 2.1 GHz at 1 200 W).  NOT the product path: compiled into -DSIMT_ABLATION builds only, SIMT_IGEMM3=1.
#ifdef SIMT_ABLATION
#include "../conv2_common.h"
#include "../conv2_epilogue.h"
#include <stdlib.h>
#include <type_traits>

// MODE 0 = product; 1 = loaders only, 2 = multiplying waves only, 3 = no fragment reads, 4 = no MFMAs (timing ablations, -DSIMT_ABLATION + SIMT_CONV3_MODE; outputs meaningless)
template <int BN, int TMP, int MODE = 0>
__global__ __launch_bounds__(512, 2) void conv_igemm3_kernel(Conv2KArgs a) {
  constexpr int NT = 512, NST = 3;
  constexpr int WM = 2, WN = 2;                    // multiplying waves: 2 (pixels) x 2 (couts)
  constexpr int TM = TMP, TN = BN / WN / 16;       // 5 (or 4) x 8 accumulator quads per wave
  constexpr int BM = WM * TM * 16;                 // allocated pixel rows (160 or 128)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int NL = 256;                          // loader threads (waves 4-7)
  constexpr int A_ITL = BM * 8 / NL, B_ITL = BN * 8 / NL;       // pieces per loader wave and stage: 5 (4) + 8
  static_assert((BM * 8) % NL == 0 && (BN * 8) % NL == 0 && A_ITL + B_ITL <= 16, "piece counts");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = wave >= 4;
  const int wm = (wave & 3) / WN, wn = (wave & 3) % WN;

  STAMP(0);
  const int nwg = a.ntiles_m * a.ntiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int mt = tile / a.ntiles_n, nt = tile - mt * a.ntiles_n;
  const int m0 = mt * a.rows, n0 = nt * BN;
  const int m_end = min(a.M, m0 + a.rows);
  const int nk = a.ntaps * a.kc_per_tap;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (loader) {
    // ================= loader waves: piece q = i*NL + ltid -> row q>>3, 16-B position q&7; the chunk index is XOR-swizzled with (row>>1)&7
    // on the global SOURCE address (the LDS image stays lane-linear as global_load_lds needs), exactly as in conv_igemm2.hip
    const int ltid = tid - 256, lw = wave - 4;
    const int lcg = (ltid & 7) ^ (((ltid >> 3) >> 1) & 7);       // NL/8 = 32 rows per pass: a multiple of 16, the key only depends on ltid>>3
    unsigned la_off[A_ITL];
    unsigned long long la_ok[A_ITL];
#pragma unroll
    for (int i = 0; i < A_ITL; ++i) {
      const int m = m0 + i * (NL / 8) + (ltid >> 3);
      la_ok[i] = 0ull;
      la_off[i] = 0u;
      if (m < m_end) {
        int b, r, oy, ox;
        fast_divmod(m, a.Ho * a.Wo, a.rcp_hw, b, r);
        fast_divmod(r, a.Wo, a.rcp_wo, oy, ox);
        const int iy = oy * a.stride, ix = ox * a.stride;
        la_off[i] = (unsigned)(((b * a.H + iy) * a.W + ix)) * (unsigned)a.pix_bytes + (unsigned)(lcg * 16);
        unsigned long long msk = 0ull;
        for (int t = 0; t < a.ntaps; ++t) {
          const int yy = iy + a.dy[t], xx = ix + a.dx[t];
          if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) msk |= (1ull << t);
        }
        la_ok[i] = msk;
      }
    }
    unsigned lb_off[B_ITL];
#pragma unroll
    for (int i = 0; i < B_ITL; ++i)
      lb_off[i] = (unsigned)(n0 + i * (NL / 8) + (ltid >> 3)) * (unsigned)a.wrow_bytes + (unsigned)(lcg * 16);
    const char* lz = a.zero + lcg * 16;
    int l_tap = 0, l_kc = 0, l_kt = 0;
    auto lissue = [&](int buf) {
      const int toff = a.toff[l_tap] + l_kc * 128;
      char* sbase = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < A_ITL; ++i) {
        const bool ok = (la_ok[i] >> l_tap) & 1ull;
        const char* src = ok ? a.x + (unsigned)(la_off[i] + (unsigned)toff) : lz;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sbase + (i * NL + lw * 64) * 16), 16, 0, 0);
      }
      const unsigned wk = (unsigned)l_kt * 128u;
#pragma unroll
      for (int i = 0; i < B_ITL; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(a.w + (lb_off[i] + wk)), LPTR(sbase + A_BYTES + (i * NL + lw * 64) * 16), 16, 0, 0);
      ++l_kt;
      if (++l_kc == a.kc_per_tap) { l_kc = 0; ++l_tap; }
    };
    STAMP(1);
    if (MODE != 2) {
      lissue(0);
      if (nk > 1) lissue(1);
    }
    int lbuf = 0;
    for (int kt = 0; kt < nk; ++kt) {
      if (MODE != 2) { if (kt + 1 < nk) wait_vmcnt<A_ITL + B_ITL>(); else wait_vmcnt<0>(); }       // stage kt landed, stage kt+1 may be in flight
      __builtin_amdgcn_s_barrier();
      if (MODE != 2 && kt + 2 < nk) lissue(lbuf >= 1 ? lbuf - 1 : 2);           // slot (kt+2)%3: read by the multiplying waves in stage kt-1
      lbuf = (lbuf + 1 == 3) ? 0 : lbuf + 1;
    }
  } else {
    // ================= multiplying waves
    const int sw = (lane >> 1) & 7;
    const int frag_row_off = (lane & 15) * 128;
    const int kq = lane >> 4;
    const int xbase = (wm * TM * 16) * 128 + frag_row_off;
    const int wbase = A_BYTES + (wn * TN * 16) * 128 + frag_row_off;
    // Software pipeline over 32-deep K halves.  One multiplying wave per SIMD: nothing else on the SIMD hides an LDS round trip, so every
    // fragment read is issued a whole half (8 groups of TM MFMAs = 640 cycles) before its use, as inline asm with hand-counted lgkmcnt
    // (hipcc's own counting falls back to lgkmcnt(0) at the loop head and in front of each half):
    //   * pixel fragments double-buffered (X[0] / X[1]), weight fragments ROLLING: group j = TM MFMAs of weight fragment j against all
    //     pixel fragments; once group j has issued, register W[j] is re-loaded with the NEXT half's fragment j (the MFMAs have read it);
    //   * the stage barrier sits inside the SECOND half of a stage, after its first two groups: by then every read of the stage has
    //     landed (lgkmcnt(0) costs nothing), the loaders may refill the slot, and the first half of the next stage is fetched under the
    //     remaining six groups.  Issue order of the reads is fixed (pixels of the next half, then W[0], W[1], ... W[7]); the counts below
    //     follow from it: group 0 of a half waits lgkmcnt(7), groups 1-7 of a first half lgkmcnt(TM + 7).
    bf16x8 X[2][TM], W[TN];
    static_assert(TN == 8 && (TM == 5 || TM == 4), "lgkmcnt schedule below");
    const unsigned lds0 = (unsigned)(size_t)LPTR(smem);
    auto rd = [](unsigned addr, auto OFF) {
      bf16x8 r;
      if constexpr (MODE == 3) asm volatile("; no read %0 %1 %2" : "=v"(r) : "v"(addr), "n"(decltype(OFF)::value));      // timing ablation
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(OFF)::value));
      return r;
    };
    auto load_x = [&](auto SET, unsigned addr) {          // TM reads, rows i*16 of this wave's pixel block
      constexpr int S = decltype(SET)::value;
      X[S][0] = rd(addr, std::integral_constant<int, 0>{});
      X[S][1] = rd(addr, std::integral_constant<int, 2048>{});
      X[S][2] = rd(addr, std::integral_constant<int, 4096>{});
      X[S][3] = rd(addr, std::integral_constant<int, 6144>{});
      if constexpr (TM == 5) X[S][4] = rd(addr, std::integral_constant<int, 8192>{});
    };
    auto load_w = [&](auto J, unsigned addr) {
      constexpr int j = decltype(J)::value;
      W[j] = rd(addr, std::integral_constant<int, j * 2048>{});
    };
    // wait until at most N reads are outstanding; names the registers the next group consumes (no consumer can be scheduled above it)
    auto wait_wx = [&](auto N, auto J, auto SET) {
      constexpr int j = decltype(J)::value, S = decltype(SET)::value;
      if constexpr (TM == 5)
        asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(W[j]), "+v"(X[S][0]), "+v"(X[S][1]), "+v"(X[S][2]), "+v"(X[S][3]), "+v"(X[S][4]) : "n"(decltype(N)::value));
      else
        asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(W[j]), "+v"(X[S][0]), "+v"(X[S][1]), "+v"(X[S][2]), "+v"(X[S][3]) : "n"(decltype(N)::value));
      __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_w = [&](auto N, auto J) {
      constexpr int j = decltype(J)::value;
      asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(W[j]) : "n"(decltype(N)::value));
      __builtin_amdgcn_sched_barrier(0);
    };
    auto group = [&](auto J, auto SET) {
      constexpr int j = decltype(J)::value, S = decltype(SET)::value;
      if constexpr (MODE == 4) { asm volatile("" :: "v"(W[j]), "v"(X[S][0]), "v"(X[S][TM - 1])); }                        // timing ablation: no MFMAs
      else {
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[j], X[S][i], acc[j][i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
#define IC(n) std::integral_constant<int, n>{}
    // first half of a stage: fragments W (all 8), X[0]; fetches the second half of the SAME stage from (xa1, wa1)
    auto half0 = [&](unsigned xa1, unsigned wa1) {
      wait_wx(IC(7), IC(0), IC(0));
      group(IC(0), IC(0));
      load_x(IC(1), xa1);
      load_w(IC(0), wa1);
      __builtin_amdgcn_sched_barrier(0);
#define G0(j) wait_w(IC(TM + 7), IC(j)); group(IC(j), IC(0)); load_w(IC(j), wa1); __builtin_amdgcn_sched_barrier(0);
      G0(1) G0(2) G0(3) G0(4) G0(5) G0(6) G0(7)
#undef G0
    };
    // second half of a stage with a successor: barrier after two groups, then the first half of the NEXT stage is fetched from (xa0, wa0)
    auto half1_more = [&](unsigned xa0, unsigned wa0) {
      wait_wx(IC(7), IC(0), IC(1));
      group(IC(0), IC(1));
      wait_w(IC(6), IC(1));
      group(IC(1), IC(1));
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(W[2]), "+v"(W[3]), "+v"(W[4]), "+v"(W[5]), "+v"(W[6]), "+v"(W[7]));   // the whole stage is in registers
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();                    // the next stage is in LDS; this stage's slot may be refilled
      asm volatile("" ::: "memory");
      load_x(IC(0), xa0);
      load_w(IC(0), wa0);
      load_w(IC(1), wa0);
      __builtin_amdgcn_sched_barrier(0);
#define G1(j) group(IC(j), IC(1)); load_w(IC(j), wa0); __builtin_amdgcn_sched_barrier(0);
      G1(2) G1(3) G1(4) G1(5) G1(6) G1(7)
#undef G1
    };
    auto half1_last = [&]() {
      wait_wx(IC(7), IC(0), IC(1));
      group(IC(0), IC(1));
#define GL(j) wait_w(IC(7 - j), IC(j)); group(IC(j), IC(1));
      GL(1) GL(2) GL(3) GL(4) GL(5) GL(6) GL(7)
#undef GL
    };
    // per-lane LDS addresses of this wave's fragments in slot 0, k-half 0 / 1
    const unsigned xa[2] = {lds0 + (unsigned)xbase + (unsigned)(((0 + kq) ^ sw) << 4), lds0 + (unsigned)xbase + (unsigned)(((4 + kq) ^ sw) << 4)};
    const unsigned wa[2] = {lds0 + (unsigned)wbase + (unsigned)(((0 + kq) ^ sw) << 4), lds0 + (unsigned)wbase + (unsigned)(((4 + kq) ^ sw) << 4)};
    __builtin_amdgcn_s_barrier();                    // barrier(0): the loaders' counted wait + this barrier: stage 0 is in LDS
    asm volatile("" ::: "memory");
    STAMP(2);
    if (MODE != 1) {
      load_x(IC(0), xa[0]);
      load_w(IC(0), wa[0]); load_w(IC(1), wa[0]); load_w(IC(2), wa[0]); load_w(IC(3), wa[0]);
      load_w(IC(4), wa[0]); load_w(IC(5), wa[0]); load_w(IC(6), wa[0]); load_w(IC(7), wa[0]);
      __builtin_amdgcn_sched_barrier(0);
    }
    unsigned so = 0;                                   // byte offset of the current stage's slot
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const unsigned sn = (so + STAGE == NST * STAGE) ? 0u : so + STAGE;
      if (MODE != 1) {
        half0(xa[1] + so, wa[1] + so);
        half1_more(xa[0] + sn, wa[0] + sn);
      } else {
        __builtin_amdgcn_s_barrier();
      }
      so = sn;
    }
    if (MODE != 1) {
      half0(xa[1] + so, wa[1] + so);
      half1_last();
    }
#undef IC
  }
  STAMP(3);
  conv2_epilogue<BN, BM, NT, TN, TM>(a, smem, acc, !loader, wm, wn, tid, lane, m0, n0, m_end, mt);
}

template <int BN, int TM, int MODE = 0>
static int launch_conv3(const Conv2KArgs& k, hipStream_t st) {
  constexpr int BM = 2 * TM * 16;
  const size_t ring = 3 * (size_t)(BM * 128 + BN * 128);
  const size_t epi = (size_t)BM * (BN * 2 + 8) + (size_t)(512 / (BN / 8)) * 2 * BN * 4;
  const size_t lds = ring > epi ? ring : epi;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm3_kernel<BN, TM, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm3_kernel<BN, TM, MODE>), dim3(k.ntiles_m * k.ntiles_n), dim3(512), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// Called by simt_conv_fprop_bf16_v2 (conv_igemm2.hip) for the <256, tm, 3> shapes.
int simt_conv_igemm3_launch(const Conv2KArgs& k, int tm, hipStream_t st) {
#ifdef SIMT_ABLATION
  static const int mode = getenv("SIMT_CONV3_MODE") ? atoi(getenv("SIMT_CONV3_MODE")) : 0;
  if (mode == 1) return tm == 5 ? launch_conv3<256, 5, 1>(k, st) : launch_conv3<256, 4, 1>(k, st);
  if (mode == 2) return tm == 5 ? launch_conv3<256, 5, 2>(k, st) : launch_conv3<256, 4, 2>(k, st);
  if (mode == 3) return tm == 5 ? launch_conv3<256, 5, 3>(k, st) : launch_conv3<256, 4, 3>(k, st);
  if (mode == 4) return tm == 5 ? launch_conv3<256, 5, 4>(k, st) : launch_conv3<256, 4, 4>(k, st);
#endif
  return tm == 5 ? launch_conv3<256, 5>(k, st) : launch_conv3<256, 4>(k, st);
}
bool simt_conv_igemm3_enabled() {
  static const int on = getenv("SIMT_IGEMM3") ? atoi(getenv("SIMT_IGEMM3")) : 0;     // -DSIMT_ABLATION builds: SIMT_IGEMM3=1 selects this kernel
  return on != 0;
}
#endif  // SIMT_ABLATION
