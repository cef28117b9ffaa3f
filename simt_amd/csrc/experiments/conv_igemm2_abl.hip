// EXPERIMENTS on the bf16 implicit-GEMM conv kernel (round 3; -DSIMT_ABLATION builds only, never in the shipped library): the product kernel
// (../conv_igemm2.hip) with its timing-ablation MODEs, loader waves (LW) and the weights-direct path (WD) still in the template.
// Original header of the kernel follows.
//
// Implicit-GEMM convolution, bf16 throughput kernel (fprop and dgrad) for gfx950 -- second generation.
//
// Same contract as conv_igemm.hip (simt_conv_desc; reference model/deeplab_multi.py:62,68,73,110,156 and their
// dgrads), built around what the per-shape profile of round 1 showed: the 128x128 / 2-buffer kernel kept only 64 KB in
// flight per CU and was latency-bound at ~6 TB/s of L2->LDS fill.  Here:
//   * tile 128 pixels x BN couts (BN = 256 / 128 / 64), K-stage = 128 B (64 bf16) of one tap, 8 waves (512 threads),
//     one workgroup per CU, a 3-deep global_load_lds ring (2 stages = up to 96 KB in flight while the third is
//     multiplied), counted s_waitcnt vmcnt(N) + raw s_barrier: ONE barrier per K-stage, loads never drained in the loop;
//   * A-gather addressing hoisted: per-row 32-bit pixel offset + a 64-bit tap-validity mask computed once, per stage
//     only a bit test, an add and a select per 16-B chunk (the old kernel re-derived iy/ix and divided per stage);
//   * MFMA operands swapped (weights = A operand, pixels = B operand) so every accumulator register quad is 4 consecutive
//     output channels of one pixel: the epilogue converts to bf16 in registers, writes 8 B per quad into a padded LDS
//     tile and streams it out as whole 512-B rows with bias / residual / ReLU applied on the way;
//   * BatchNorm batch statistics (sum, sum of squares of the values as stored) accumulated by the same row-streaming pass,
//     row groups combined in fixed order through LDS: one deterministic slot per (pixel tile, channel).
//   * flexible pixel tile: a workgroup owns `rows` <= BM consecutive pixels (BM = 128 or 160 allocated), rows chosen on
//     the host so that the grid is a whole number of 256-CU rounds (M = 37636: 255 tiles of 148 rows instead of 295
//     of 128 -> one round instead of two);
//   * the two waves of a SIMD run the K-stage in opposite order: waves 0-3 load-then-multiply, waves 4-7 multiply the
//     fragments they fetched in the previous stage first and load afterwards, so one wave's MFMA burst covers the other
//     wave's global_load_lds issue + ds_read latency (MI355X_MICROARCH "two waves per SIMD", item 9).
// LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the global SOURCE address and on the
// ds_read_b128 side (conflict-free 16-lane groups), the LDS image itself stays lane-linear as global_load_lds needs.
#ifdef SIMT_ABLATION
#include "../conv2_common.h"
#include "../conv2_epilogue.h"
#include <stdlib.h>
#include <type_traits>



// ---- weights-direct helpers (inline asm: hipcc must neither count these loads nor drain the LDS-DMA ring for them, guide 5.7 item 1)
template <int OFF> __device__ __forceinline__ u32x4 gload_x4(const char* p) {
  u32x4 r;
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r) : "v"(p), "n"(OFF) : "memory");
  return r;
}
// counted wait that NAMES the eight fragment registers it retires ("+v": no consumer can be scheduled above it, form (ii) of the guide)
template <int N> __device__ __forceinline__ void wait_vmcnt_regs8(u32x4* w) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7])
               : "n"(N)
               : "memory");
}

// MODE 0 = product.  MODE 1 (loads only), 2 (MFMA + fragment reads only), 3 (pixel pieces for one tap column in three), 4 (no pixel
// pieces), 5 (no weight pieces) are timing-ablation builds selected by the environment variable SIMT_CONV2_MODE; their outputs are
// meaningless.  Round 3, 3x3 256 -> 256 at M = 37 636 (us): product 50, loads only 32, MFMA + reads only 35, mode 3 49.5, mode 4 43,
// mode 5 42.6 -- neither operand's fill is "the" bound (taking ALL weight pieces away buys 15 %), re-using the pixel window across
// the dx taps could buy 4 % at most.  Also measured and removed again: one LDS-DMA piece between every five MFMAs (pinned with
// sched_barrier) instead of the burst of 6-7 pieces: +9 ... +12 % time on every shape (profiles/r03_conv_experiments.txt).
// LW = 4 / 8: wave specialisation (experiments, off by default; SIMT_CONV2_LW=4|8).  Extra waves do nothing but fill the ring;
// the eight consumer waves never issue a global_load_lds and never wait on vmcnt.  NST = 3 variants only.  Measured on MI355X
// (round-1 A/B harness, now profiles/tools/ab_conv.py; 3x3 256->256): default 48-50 us, LW=4 51.6 us (four waves issue the 13 pieces per stage more slowly
// than eight waves issue 7 each), LW=8 (8 loaders + 8 consumers at 125 VGPRs, single-buffered fragments) 50.5 us = unchanged.
// Three very different schedules, one time: the kernel is bound by a shared resource, not by issue slots.  Per 64-deep stage the
// LDS sees 52 KB of DMA writes (loads-only ablation: 31 us = ~30 B/clk/CU of L2->LDS fill) and 144 KB of fragment reads
// (562 cycles at 256 B/clk) next to 1 280 cycles of MFMA per SIMD; the fill rate is the floor.  The lever is fewer staged bytes
// per FLOP (2-D halo tiles for the 3x3 pixel operand), not scheduling.
// WD = 1 ("weights direct", round 3 experiment, -DSIMT_ABLATION builds only; measured SLOWER, see simt_conv_wants_frag): the weight
// operand never enters LDS.  The host hands it over in MFMA-fragment order
// (simt_conv_desc.w_frag) and every wave loads the fragments of ITS 64 output channels for the next K stage straight into registers:
// TN * 2 global_load_dwordx4 of 1 KB contiguous each, two register sets (stage kt is multiplied while kt + 1 lands), inline asm with
// hand-counted vmcnt (the register loads are issued BEFORE the stage's LDS-DMA pieces, so the counted wait that leaves one stage of
// pixel pieces in flight also covers them).  Per stage and wave: 2-3 LDS-DMA pieces instead of 6-7, 10 fragment reads instead of 18,
// LDS traffic 100 KB instead of 196 KB per CU; the two pixel halves of the 2 x 4 wave grid fetch the same 8 KB of weights (the second
// request hits L1/L2).  Why: profiles/microbench/fillbench.hip and DESIGN.md section 9.
// KS = 1 (round 5, SIMT_CONV2_KSTAMP=1; modes 0 and 2 of the wide tile only): s_memtime stamps INSIDE the K loop, of one middle stage, kept in
// scalar registers and written after the loop (no vector-memory instruction is added to the loop: the counted vmcnt waits stay exact) --
// where a stage's time goes for an early wave (wave 0) and a late wave (wave 4) of every workgroup.  g_kstamps[(block * 2 + late) * 8 + i]:
//   early: 0 barrier exit, 1 fragment reads + LDS-DMA pieces issued, 2 fragments landed (lgkmcnt 0), 3 MFMA burst issued, 4 next barrier exit
//   late : 0 barrier exit, 1 MFMA burst (previous stage's fragments) issued, 2 pieces + fragment reads issued, 3 fragments landed, 4 next barrier exit
//   5 / 6: s_memtime / s_memrealtime (100 MHz) at kernel start, 7: s_memtime after the loop (clock = (7 - 5) / ((realtime delta) / 100 MHz))
static __device__ unsigned long long g_kstamps[8192 * 16];
extern "C" int simt_debug_kstamps(unsigned long long* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kstamps), (size_t)nblocks * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int simt_debug_stamps_abl(unsigned long long* out, int n) {      // g_stamps of THIS translation unit (STAMP in conv2_common.h)
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int simt_debug_stamps_abl_rt(unsigned long long* out, int n) {      // s_memrealtime ticks (100 MHz) between stamps 0 and 6, one per workgroup
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_rt), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#define KST(i) do { if constexpr (KS != 0) { if (kt == kmid) { asm volatile("" ::: "memory"); kst[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } } } while (0)
#define KST_NEXT() do { if constexpr (KS != 0) { if (kt == kmid + 1) { asm volatile("" ::: "memory"); kst[4] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } } } while (0)
#define KST_LANDED(i) do { if constexpr (KS != 0) { if (kt == kmid) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); kst[i] = __builtin_amdgcn_s_memtime(); asm volatile("" ::: "memory"); } } } while (0)
template <int BN, int TMP, int NSTP, int MODE, int LW = 0, int WD = 0, int KS = 0>
__global__ __launch_bounds__(512 + LW * 64, (LW == 8 ? 4 : LW ? 1 : (NSTP == 2 ? 4 : 2))) void conv_igemm2x_kernel(Conv2KArgs a) {
  constexpr int NT = 512, NST = NSTP;   // NST = 3: one workgroup per CU, two stages in flight; NST = 2 (short-K, output-
                                        // bound shapes): two workgroups per CU so one's epilogue overlaps the other's loads
  constexpr int WM = (BN == 64) ? 4 : 2;          // waves along pixels
  constexpr int WN = 8 / WM;                       // waves along couts
  constexpr int TM = TMP, TN = BN / WN / 16;
  constexpr int BM = WM * TM * 16;                 // allocated pixel rows (128 or 160)
  static_assert(!WD || (LW == 0 && MODE == 0 && NSTP == 3 && BN == 256), "weights-direct: product build of the wide 3-slot kernel only");
  constexpr int A_BYTES = BM * 128, B_BYTES = WD ? 0 : BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = WD ? 0 : BN * 8 / NT;   // 16-B chunks per thread per stage
  constexpr bool A_TAIL = (BM * 8) % NT != 0;      // BM = 160: the third A pass is only issued by waves 0-3
  constexpr int CP = BN * 2 + 8;                   // epilogue tile pitch in bytes (bf16 row + 8 B pad)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  STAMP(0);
  [[maybe_unused]] unsigned long long kst[8] = {0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull, 0ull};
  if constexpr (KS != 0) { kst[5] = __builtin_amdgcn_s_memtime(); kst[6] = __builtin_amdgcn_s_memrealtime(); }
  const int nwg = a.ntiles_m * a.ntiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int mt = tile / a.ntiles_n, nt = tile - mt * a.ntiles_n;
  const int m0 = mt * a.rows, n0 = nt * BN;
  const int m_end = min(a.M, m0 + a.rows);

  if constexpr (LW > 0) {
    static_assert(NSTP == 3, "loader waves: 3-slot ring only");
    if (wave >= 8) {
      // ================= loader waves: piece q = i*NL + ltid -> row q>>3, 16-B position q&7 (same swizzle as below)
      constexpr int NL = LW * 64;
      constexpr int A_ITL = (BM * 8 + NL - 1) / NL, B_ITL = BN * 8 / NL;
      constexpr bool A_TAILL = (BM * 8) % NL != 0;         // last pixel pass only issued by the first waves
      static_assert((BN * 8) % NL == 0 && A_ITL + B_ITL <= 16, "piece counts");
      const int ltid = tid - NT, lw = wave - 8;
      const bool l_tail_wave = !A_TAILL || lw < (BM * 8 - (A_ITL - 1) * NL) / 64;
      const int lcg = (ltid & 7) ^ (((ltid >> 3) >> 1) & 7);     // NL/8 = 32 rows per pass: a multiple of 16, key unchanged
      unsigned la_off[A_ITL];
      unsigned long long la_ok[A_ITL];
#pragma unroll
      for (int i = 0; i < A_ITL; ++i) {
        const int m = m0 + i * (NL / 8) + (ltid >> 3);
        la_ok[i] = 0ull;
        la_off[i] = 0u;
        if (m < m_end) {
          const int hw = a.Ho * a.Wo;
          const int b = m / hw;
          const int r = m - b * hw;
          const int oy = r / a.Wo;
          const int ox = r - oy * a.Wo;
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
          if (i == A_ITL - 1 && !l_tail_wave) break;
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
      const int nkl = a.ntaps * a.kc_per_tap;
      lissue(0);
      if (nkl > 1) lissue(1);
      int lbuf = 0;
      for (int kt = 0; kt < nkl; ++kt) {
        if (kt + 1 >= nkl) wait_vmcnt<0>();                                     // stage kt landed, stage kt+1 may be in flight
        else if (A_TAILL && !l_tail_wave) wait_vmcnt<A_ITL - 1 + B_ITL>();
        else wait_vmcnt<A_ITL + B_ITL>();
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nkl) lissue(lbuf >= 1 ? lbuf - 1 : 2);                     // slot (kt+2)%3: read by everybody in step kt-1
        lbuf = (lbuf + 1 == 3) ? 0 : lbuf + 1;
      }
      if (!a.out_f32) {            // mirror the consumers' epilogue barriers (tile write, tile read, statistics)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (a.stats) __builtin_amdgcn_s_barrier();
      }
      return;
    }
  }

  // ---- hoisted A-gather metadata: chunk q = i*NT + tid -> row = q>>3, position q&7
  const int c_pos = tid & 7;
  const int a_cg = c_pos ^ (((tid >> 3) >> 1) & 7);          // (row>>1)&7 only depends on tid>>3 because NT/8 = 64 is even
  unsigned a_off[A_IT];
  unsigned long long a_ok[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int row = i * (NT / 8) + (tid >> 3);
    const int m = m0 + row;
    a_ok[i] = 0ull;
    a_off[i] = 0u;
    if (m < m_end) {
      int b, r, oy, ox;
      fast_divmod(m, a.Ho * a.Wo, a.rcp_hw, b, r);
      fast_divmod(r, a.Wo, a.rcp_wo, oy, ox);
      const int iy = oy * a.stride, ix = ox * a.stride;
      a_off[i] = (unsigned)(((b * a.H + iy) * a.W + ix)) * (unsigned)a.pix_bytes + (unsigned)(a_cg * 16);
      unsigned long long msk = 0ull;
      for (int t = 0; t < a.ntaps; ++t) {
        const int yy = iy + a.dy[t], xx = ix + a.dx[t];
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) msk |= (1ull << t);
      }
      a_ok[i] = msk;
    }
  }
  unsigned b_off[B_IT > 0 ? B_IT : 1];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = i * (NT / 8) + (tid >> 3);
    b_off[i] = (unsigned)(n0 + row) * (unsigned)a.wrow_bytes + (unsigned)(a_cg * 16);
  }
  const char* zsrc = a.zero + a_cg * 16;
  const bool a_tail_wave = !A_TAIL || wave < (BM * 8 - (A_IT - 1) * NT) / 64;

  int ld_tap = 0, ld_kc = 0;                     // position of the NEXT stage to be issued: (64-channel chunk, tap), taps innermost
  auto issue = [&](int buf) {
    const int toff = a.toff[ld_tap < SIMT_MAX_TAPS ? ld_tap : SIMT_MAX_TAPS - 1] + ld_kc * 128;     // (WD issues stages past the end: all-zero)
    char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      if (i == A_IT - 1 && !a_tail_wave) break;
      if (MODE == 3 && (ld_tap % 3) != 1) break;      // timing ablation: what would re-using the pixel window across the dx taps buy?
      if (MODE == 4) break;                             // timing ablation: no pixel pieces at all
      const bool ok = (a_ok[i] >> ld_tap) & 1ull;
      const char* src = ok ? a.x + (unsigned)(a_off[i] + (unsigned)toff) : zsrc;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sbase + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
    // K-stage order (64-channel chunk, tap): a pixel row's 128-byte line is read by all taps in CONSECUTIVE stages -- an L2 reuse distance of
    // one stage of the XCD's workgroups (~0.7 MB) instead of kc_per_tap stages (~2.8 MB of the 4 MB L2 at Cin = 256, with the weight stream on
    // top): 3x3 convs 1-2.5 % faster, the step 26.39 -> 26.10 ms.  Weights are packed K-contiguous as (tap, channel): this stage's 64 columns
    // start at (tap * chunks + chunk) * 128 bytes.  (Compile-time only: the same order behind a run-time flag cost every launch 17-28 %.)
    const unsigned wk = (unsigned)(ld_tap * a.kc_per_tap + ld_kc) * 128u;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      if (MODE == 5) break;                             // timing ablation: no weight pieces
      __builtin_amdgcn_global_load_lds(GPTR(a.w + (b_off[i] + wk)), LPTR(sbase + A_BYTES + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
    if (++ld_tap == a.ntaps) { ld_tap = 0; ++ld_kc; }
  };
  // outstanding vector-memory ops of ONE stage for this wave (the counted wait leaves exactly one stage in flight)
  auto wait_stage = [&](bool more) {
    if (NST == 2 || !more) { wait_vmcnt<0>(); return; }
    if constexpr (A_TAIL) {
      if (!a_tail_wave) { wait_vmcnt<A_IT - 1 + B_IT>(); return; }
    }
    wait_vmcnt<A_IT + B_IT>();
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = a.ntaps * a.kc_per_tap;
  [[maybe_unused]] const int kmid = nk / 2;
  const int sw = (lane >> 1) & 7;
  const int frag_row_off = (lane & 15) * 128;
  const int kq = lane >> 4;
  const int xbase = (wm * TM * 16) * 128 + frag_row_off;
  const int wbase = A_BYTES + (wn * TN * 16) * 128 + frag_row_off;

  bf16x8 xf[2][TM], wf[2][TN];
  auto load_frags = [&](int buf) {
    const char* px = smem + buf * STAGE + xbase;
    const char* pw = smem + buf * STAGE + wbase;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int coff = ((4 * s + kq) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[s][i] = *(const bf16x8*)(px + i * 16 * 128 + coff);
      if constexpr (!WD) {
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[s][j] = *(const bf16x8*)(pw + j * 16 * 128 + coff);
      }
    }
  };
  auto mma = [&]() {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
          acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], xf[s][i], acc[j][i], 0, 0, 0);
  };
  // ---- weights-direct state: two register sets of this wave's TN x 2 weight fragments, per-lane source pointer (centred: the eight
  // 1-KB fragments of a stage sit at immediate offsets -4096 ... +3072), one stage = nt16 blocks of 2 KB further.
  // Every K stage of the WD loops has the SAME instruction sequence (no run-time condition around an asm statement with register
  // outputs: hipcc would merge the paths with v_mov copies of the destination registers placed BEFORE the wait that retires them):
  //   * the counted wait is a per-half constant: waves 0-3 (the early half) are exactly the waves that issue the third pixel piece
  //     of a 160-row tile, waves 4-7 issue two;
  //   * stages past the end are still "issued": their pixel pieces come from the zero page (tap bit clear -> zsrc) into a ring slot
  //     nobody reads any more, their weight registers re-load the last stage (pointer not advanced) -- 2 + 1 dummy stages per tile;
  //   * two steps per loop trip (register set = stage parity, compile time), an odd last stage peeled.
  u32x4 wr[2][WD ? TN * 2 : 1];
  const char* wptr = nullptr;
  size_t wstep = 0;
  int w_kt = 0;                        // stage the NEXT register loads belong to
  if constexpr (WD) {
    static_assert(!WD || (TN == 4 && (!A_TAIL || (BM * 8 - (A_IT - 1) * NT) / 64 == 4)), "eight fragments per stage; tail waves = early half");
    wptr = a.wf + ((size_t)(n0 / 16 + wn * TN) * 2048 + (size_t)lane * 16 + 4096);
    wstep = (size_t)a.nt16 * 2048;
  }
  auto issue_w = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    if constexpr (WD) {
      wr[S][0] = gload_x4<-4096>(wptr); wr[S][1] = gload_x4<-3072>(wptr); wr[S][2] = gload_x4<-2048>(wptr); wr[S][3] = gload_x4<-1024>(wptr);
      wr[S][4] = gload_x4<0>(wptr);     wr[S][5] = gload_x4<1024>(wptr);  wr[S][6] = gload_x4<2048>(wptr);  wr[S][7] = gload_x4<3072>(wptr);
      ++w_kt;
      wptr += (w_kt < nk) ? wstep : (size_t)0;
    }
  };
  auto mma_w = [&](auto SET) {
    constexpr int S = decltype(SET)::value;
    if constexpr (WD) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wr[S][j * 2 + s]), xf[s][i], acc[j][i], 0, 0, 0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  STAMP(1);
  if (MODE != 2 && LW == 0) {
    issue(0);
    if (NST == 3 && (WD || nk > 1)) issue(1);
  }
  int buf = 0;
  if constexpr (WD) {
    if (wave < 4) {
      // ---- early half: [barrier] pixel fragments(kt) -> REG(kt+1) -> DMA(kt+2) -> MFMA(kt); two weight register sets (set = stage parity)
      issue_w(P0{});                                     // REG(0) behind DMA(0), DMA(1): the first step drains everything once
      auto step = [&](auto PAR, auto NWAIT) {
        constexpr int P = decltype(PAR)::value;
        wait_vmcnt_regs8<decltype(NWAIT)::value>(wr[P]); // DMA(kt) and REG(kt) landed; DMA(kt+1) (A_IT pieces of this wave) may fly
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        load_frags(buf);
        issue_w(std::integral_constant<int, 1 - P>{});
        issue(buf >= 1 ? buf - 1 : NST - 1);
        mma_w(PAR);
        buf = (buf + 1 == NST) ? 0 : buf + 1;
      };
      using NW = std::integral_constant<int, A_IT>;
      step(P0{}, std::integral_constant<int, 0>{});
      if (nk > 1) {
        step(P1{}, NW{});
        for (int kt = 2; kt + 1 < nk; kt += 2) { step(P0{}, NW{}); step(P1{}, NW{}); }
        if (nk & 1) step(P0{}, NW{});
      }
    } else {
      // ---- late half: [barrier] MFMA(kt-1) from registers -> REG(kt) into the SAME registers (the MFMAs have read them: in-order
      // issue) -> DMA(kt+2) -> pixel fragments(kt).  One weight register set: this half multiplies a stage behind.
      constexpr int NWL = A_TAIL ? A_IT - 1 : A_IT;
#pragma unroll
      for (int q = 0; q < TN * 2; ++q) wr[0][q] = (u32x4){0u, 0u, 0u, 0u};
      auto step = [&](auto FIRST) {
        wait_vmcnt_regs8<NWL>(wr[0]);                    // DMA(kt) and REG(kt-1) landed; DMA(kt+1) may fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my fragment reads of stage kt-1 are done before anyone refills
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (!decltype(FIRST)::value) mma_w(P0{});
        issue_w(P0{});
        issue(buf >= 1 ? buf - 1 : NST - 1);
        load_frags(buf);
        buf = (buf + 1 == NST) ? 0 : buf + 1;
      };
      step(std::true_type{});
      for (int kt = 1; kt < nk; ++kt) step(std::false_type{});
      wait_vmcnt_regs8<0>(wr[0]);
      mma_w(P0{});
    }
    wait_vmcnt<0>();                                   // the dummy stages: nothing of this wave may land after this point
  } else if constexpr (LW == 8) {
    // ---- consumers of the 8 + 8 build: 128 VGPRs per wave (4 waves per SIMD) -> fragments of ONE 32-deep K half at a time
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* px = smem + buf * STAGE + xbase;
      const char* pw = smem + buf * STAGE + wbase;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int coff = ((4 * s + kq) ^ sw) << 4;
        bf16x8 x1[TM], w1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) x1[i] = *(const bf16x8*)(px + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < TN; ++j) w1[j] = *(const bf16x8*)(pw + j * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j], x1[i], acc[j][i], 0, 0, 0);
      }
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else if (wave < 4) {
    // ---- early half: [barrier] issue(kt+2) -> fragments(kt) -> MFMA(kt)
    for (int kt = 0; kt < nk; ++kt) {
      if (MODE != 2 && LW == 0) wait_stage(kt + 1 < nk);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt == 0) STAMP(2);
      KST(0);
      KST_NEXT();
      if constexpr (MODE == 0 || MODE == 11 || MODE == 13 || MODE == 3 || MODE == 4 || MODE == 5) {
        load_frags(buf);
        if (LW == 0 && kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
        KST(1);
        KST_LANDED(2);
        if (MODE == 13) __builtin_amdgcn_s_setprio(1);
        mma();
        if (MODE == 13) __builtin_amdgcn_s_setprio(0);
        KST(3);
      } else {
        if (MODE != 2 && kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);     // stage kt+NST-1 -> buffer (buf-1) mod NST
        if (MODE != 1) {
          load_frags(buf);
          KST(1);
          KST_LANDED(2);
          if (MODE == 12) __builtin_amdgcn_s_setprio(1);
          mma();
          if (MODE == 12) __builtin_amdgcn_s_setprio(0);
          KST(3);
        }
      }
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else {
    // ---- late half: [barrier] MFMA(kt-1) from registers -> issue(kt+2) -> fragments(kt) (kept for the next stage)
    for (int kt = 0; kt < nk; ++kt) {
      if (MODE != 2 && LW == 0) wait_stage(kt + 1 < nk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my fragment reads of stage kt-1 are done before anyone refills
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      KST(0);
      KST_NEXT();
      if (MODE != 1 && kt > 0) {
        if (MODE == 12 || MODE == 13) __builtin_amdgcn_s_setprio(1);
        mma();
        if (MODE == 12 || MODE == 13) __builtin_amdgcn_s_setprio(0);
      }
      KST(1);
      if (MODE != 2 && LW == 0 && kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
      if (MODE != 1) load_frags(buf);
      KST(2);
      KST_LANDED(3);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    if (MODE != 1) mma();
  }
  if constexpr (KS != 0) {
    if ((tid == 0 || tid == 256) && blockIdx.x < 8192) {
      kst[7] = __builtin_amdgcn_s_memtime();
      unsigned long long* o = g_kstamps + (blockIdx.x * 2 + (tid == 256 ? 1 : 0)) * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = kst[i];
    }
  }

  STAMP(3);
  conv2_epilogue<BN, BM, NT, TN, TM>(a, smem, acc, true, wm, wn, tid, lane, m0, n0, m_end, mt);
}

template <int BN, int TM, int NST, int MODE, int LW = 0, int WD = 0, int KS = 0>
static int launch_conv2m(const Conv2KArgs& k, hipStream_t st) {
  constexpr int WM = (BN == 64) ? 4 : 2;
  constexpr int BM = WM * TM * 16;
  const size_t ring = NST * (size_t)(BM * 128 + (WD ? 0 : BN * 128));
  const size_t epi = (size_t)BM * (BN * 2 + 8) + (size_t)(512 / (BN / 8)) * 2 * BN * 4;
  const size_t lds = ring > epi ? ring : epi;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm2x_kernel<BN, TM, NST, MODE, LW, WD, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm2x_kernel<BN, TM, NST, MODE, LW, WD, KS>), dim3(k.ntiles_m * k.ntiles_n), dim3(512 + LW * 64), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

static int conv2_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
int simt_conv_igemm3_launch(const Conv2KArgs& k, int tm, hipStream_t st);   // conv_igemm3.hip (experiment)
bool simt_conv_igemm3_enabled();

template <int BN, int TM, int NST>
static bool abl_launch(const Conv2KArgs& k, hipStream_t st, int* rc) {
  static const int mode = conv2_env("SIMT_CONV2_MODE"), lw = conv2_env("SIMT_CONV2_LW"), ks = conv2_env("SIMT_CONV2_KSTAMP");
  if constexpr (BN == 256 && TM == 5 && NST == 3) {      // K-loop stamps: the dominant tile, product and compute-only builds
    if (ks && mode == 0) { *rc = launch_conv2m<BN, TM, NST, 0, 0, 0, 1>(k, st); return true; }
    if (ks && mode == 2) { *rc = launch_conv2m<BN, TM, NST, 2, 0, 0, 1>(k, st); return true; }
    if (ks && mode == 1) { *rc = launch_conv2m<BN, TM, NST, 1, 0, 0, 1>(k, st); return true; }
  }
  if (mode == 0 && conv2_env("SIMT_CONV2_ABL0")) { *rc = launch_conv2m<BN, TM, NST, 0>(k, st); return true; }      // the experiments TU's copy of the product kernel (its own g_stamps)
  if (mode == 1) { *rc = launch_conv2m<BN, TM, NST, 1>(k, st); return true; }
  if (mode == 2) { *rc = launch_conv2m<BN, TM, NST, 2>(k, st); return true; }
  if (mode == 3) { *rc = launch_conv2m<BN, TM, NST, 3>(k, st); return true; }
  if (mode == 4) { *rc = launch_conv2m<BN, TM, NST, 4>(k, st); return true; }
  if (mode == 5) { *rc = launch_conv2m<BN, TM, NST, 5>(k, st); return true; }
  if (mode == 11) { *rc = launch_conv2m<BN, TM, NST, 11>(k, st); return true; }
  if (mode == 12) { *rc = launch_conv2m<BN, TM, NST, 12>(k, st); return true; }
  if (mode == 13) { *rc = launch_conv2m<BN, TM, NST, 13>(k, st); return true; }
  if constexpr (NST == 3) {
    if (lw == 4) { *rc = launch_conv2m<BN, TM, NST, 0, 4>(k, st); return true; }
    if (lw == 8) { *rc = launch_conv2m<BN, TM, NST, 0, 8>(k, st); return true; }
  }
  if constexpr (BN == 256 && NST == 3) {
    if (k.wf) { *rc = launch_conv2m<BN, TM, NST, 0, 0, 1>(k, st); return true; }   // weights in fragment order: the LDS-free weight path
    if (simt_conv_igemm3_enabled()) { *rc = simt_conv_igemm3_launch(k, TM, st); return true; }       // role-split waves (conv_igemm3.hip)
  }
  return false;
}

// Hook called by launch_conv2 of the product file in -DSIMT_ABLATION builds: true = an experiment took the launch.
bool simt_conv2_abl_launch(const Conv2KArgs& k, int bn, int tm, int nst, hipStream_t st, int* rc) {
  if (bn == 256 && nst == 3) return tm == 5 ? abl_launch<256, 5, 3>(k, st, rc) : abl_launch<256, 4, 3>(k, st, rc);
  if (bn == 128 && nst == 3) return tm == 5 ? abl_launch<128, 5, 3>(k, st, rc) : abl_launch<128, 4, 3>(k, st, rc);
  if (bn == 128 && nst == 2) return tm == 5 ? abl_launch<128, 5, 2>(k, st, rc) : abl_launch<128, 4, 2>(k, st, rc);
  if (bn == 64) return abl_launch<64, 2, 3>(k, st, rc);
  return false;
}

// SIMT_WDIRECT=1: the wide 3-slot launches take their weights from simt_conv_desc.w_frag (measured 15-40 % slower)
int simt_conv2_abl_wants_frag(const simt_conv_desc* d) {
  static const int on = getenv("SIMT_WDIRECT") ? atoi(getenv("SIMT_WDIRECT")) : 0;
  int bn, tm, nst;
  if (!on || simt_conv_variant(d, &bn, &tm, &nst) != 2) return 0;
  return bn == 256 && nst == 3 && d->Npad % 256 == 0 && (d->Cin * 2) % 128 == 0;
}
#endif  // SIMT_ABLATION
