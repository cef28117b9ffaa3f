// Round-5 experiments on conv_igemm2_kernel<256, *, 3>, -DSIMT_ABLATION builds only (csrc/build.sh ABLATION=1), selected by
// SIMT_CONV2_ROLES = 1 (LDS-DMA pieces split by role) or 2 (+ the late waves' fragment reads pipelined into their MFMA burst).  Both are
// bit-identical to the product kernel (tests/test_gpu_conv.py, test_gpu_prod_shapes.py, test_gpu_bn_fused.py green with the switch set) and
// NEITHER is faster: the role split is neutral within +-1 % on every shape (step 23.71 vs 23.82 ms), the pipelined reads are 4-17 % SLOWER
// (3x3 256 -> 256: 42.4 -> 44.5 us, 1x1 2048 -> 512: 88 -> 101 us; step 24.08 -> 24.79 ms): profiles/r05_conv_attribution.txt.
// SIMT_CONV2_INTER=1: the third schedule of the round, conv_igemm2i_kernel below -- every wave issues its LDS-DMA pieces INSIDE its MFMA burst
// (one piece behind every fifth MFMA, sources computed in front of the burst): bit-identical, 1-1.5 % SLOWER on every shape (3x3 256: 41.8 ->
// 42.4 us; step 23.80 vs 23.83 ms).  Three schedules, one time: what bounds the stage is not the order of issue.
#include "../conv2_common.h"
#include "../conv2_epilogue.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

// Round 5: the same tile, ring and MFMA chain with the LDS-DMA pieces split BY ROLE (SIMT_CONV2_ROLES).  In-loop s_memtime
// stamps of the kernel above (profiles/r05_conv_attribution.txt) showed what a 64-deep stage of the 3x3 256 -> 256 conv is made of:
//   early wave: barrier -> 18 fragment reads + 7 pieces issued 876 clocks -> 40 MFMAs 672 -> idle at the next barrier 304
//   late wave : barrier -> 40 MFMAs 716 -> 6 pieces + 18 fragment reads issued 880 -> fragments landed 48 -> barrier 236      = 1 880 per stage
// against 1 360 clocks of matrix pipe per SIMD: the late wave is the critical path, and what it serialises behind its MFMA burst is issue time
// (a piece costs ~17 clocks of the CU's vector-memory path whoever issues it: 4 waves issuing 28 pieces at once see ~63 each).  Here the early
// waves (0-3) issue ONLY the pixel pieces (BM * 8 / 256 = 4-5 each, the ones that need the tap-validity select) and the late waves (4-7) ONLY
// the weight pieces (BN * 8 / 256 = 8 each: one add + the load), so that neither role carries both address streams, the late waves need no
// pixel metadata at all, and the pieces of a stage leave in two bursts half a stage apart instead of all at once.  Same LDS image, same
// fragment reads, same MFMA order: bit-identical outputs.
// PIPE = 1 (SIMT_CONV2_ROLES=2): the late waves keep TWO fragment register sets and read the fragments of stage kt INSIDE the MFMA burst of
// stage kt-1 (one ds_read_b128 behind every second MFMA, pinned with sched_group_barrier) instead of behind it: the 18 reads and their LDS
// latency (880 + 48 clocks of the late wave's critical path above, with the pieces) leave the path; the weight pieces follow the burst.
template <int BN, int TMP, int FBN = 0, int EPI = 0, int PIPE = 0>
__global__ __launch_bounds__(512, 2) void conv_igemm2r_kernel(Conv2KArgs a) {
  constexpr int NT = 512, NH = 256, NST = 3;
  constexpr int WM = (BN == 64) ? 4 : 2;
  constexpr int WN = 8 / WM;
  constexpr int TM = TMP, TN = BN / WN / 16;
  constexpr int BM = WM * TM * 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = BM * 8 / NH, B_IT = BN * 8 / NH;      // 16-B chunks per thread per stage: pixel pieces (early half) / weight pieces (late half)
  static_assert((BM * 8) % NH == 0 && (BN * 8) % NH == 0 && A_IT <= 16 && B_IT <= 16, "piece counts");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const bool early = wave < 4;
  const int tr = tid & (NH - 1);                 // thread index inside the role
  const int w4 = wave & 3;

  const int nwg = a.ntiles_m * a.ntiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int mt = tile / a.ntiles_n, nt = tile - mt * a.ntiles_n;
  const int m0 = mt * a.rows, n0 = nt * BN;
  const int m_end = min(a.M, m0 + a.rows);

  // chunk q = i * NH + tr of an operand -> row q >> 3 = i * 32 + (tr >> 3), 16-B position q & 7; (row >> 1) & 7 = (tr >> 4) & 7 for every i
  const int c_pos = tr & 7;
  const int a_cg = c_pos ^ ((tr >> 4) & 7);
  const int nk = a.ntaps * a.kc_per_tap;
  int ld_tap = 0, ld_kc = 0;                     // position of the NEXT stage this role issues: (64-channel chunk, tap), taps innermost
  unsigned b_off0 = 0u;                          // weight pieces: ONE per-lane offset; row group i adds a uniform i * 32 * wrow_bytes
  unsigned a_off[A_IT];
  unsigned long long a_ok[A_IT];
  const char* zsrc = a.zero + a_cg * 16;
  auto issue_b = [&](int buf) {
    const unsigned wk = (unsigned)(ld_tap * a.kc_per_tap + ld_kc) * 128u;
    char* sbase = smem + buf * STAGE + A_BYTES;
    const unsigned vo = b_off0 + wk;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const char* wrows = a.w + (size_t)(unsigned)(n0 + i * 32) * (unsigned)a.wrow_bytes;      // uniform: a scalar base per row group
      __builtin_amdgcn_global_load_lds(GPTR(wrows + vo), LPTR(sbase + (i * NH + w4 * 64) * 16), 16, 0, 0);
    }
    if (++ld_tap == a.ntaps) { ld_tap = 0; ++ld_kc; }
  };
  auto issue_a = [&](int buf) {
    const int toff = a.toff[ld_tap] + ld_kc * 128;
    char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const bool ok = (a_ok[i] >> ld_tap) & 1ull;
      const char* src = ok ? a.x + (unsigned)(a_off[i] + (unsigned)toff) : zsrc;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sbase + (i * NH + w4 * 64) * 16), 16, 0, 0);
    }
    if (++ld_tap == a.ntaps) { ld_tap = 0; ++ld_kc; }
  };
  if (!early) {
    b_off0 = (unsigned)(tr >> 3) * (unsigned)a.wrow_bytes + (unsigned)(a_cg * 16);
    issue_b(0);                                  // the weight pieces need no pixel addressing: they leave at once
    if (nk > 1) issue_b(1);
  } else {
    int tdy[9], tdx[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) { tdy[t] = a.dy[t]; tdx[t] = a.dx[t]; }
    const int ntaps = a.ntaps;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int m = m0 + i * 32 + (tr >> 3);
      a_ok[i] = 0ull;
      a_off[i] = 0u;
      if (m < m_end) {
        int b, r, oy, ox;
        fast_divmod(m, a.Ho * a.Wo, a.rcp_hw, b, r);
        fast_divmod(r, a.Wo, a.rcp_wo, oy, ox);
        const int iy = oy * a.stride, ix = ox * a.stride;
        a_off[i] = (unsigned)(((b * a.H + iy) * a.W + ix)) * (unsigned)a.pix_bytes + (unsigned)(a_cg * 16);
        unsigned long long msk = 0ull;
        if (ntaps <= 9) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            const int yy = iy + tdy[t], xx = ix + tdx[t];
            if (t < ntaps && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) msk |= (1ull << t);
          }
        } else {
          for (int t = 0; t < ntaps; ++t) {
            const int yy = iy + a.dy[t], xx = ix + a.dx[t];
            if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) msk |= (1ull << t);
          }
        }
        a_ok[i] = msk;
      }
    }
    issue_a(0);
    if (nk > 1) issue_a(1);
  }

  f32x4 acc[TN][TM];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[s][j] = *(const bf16x8*)(pw + j * 16 * 128 + coff);
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
  int buf = 0;
  if (early) {
    // ---- early half: [barrier] fragments(kt) -> pixel pieces(kt+2) -> MFMA(kt)
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<A_IT>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      load_frags(buf);
      if (kt + NST - 1 < nk) issue_a(buf >= 1 ? buf - 1 : NST - 1);
      mma();
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else if constexpr (PIPE != 0) {
    // ---- late half, pipelined: [barrier] { MFMA(kt-1) from set 1-P | fragments(kt) -> set P, interleaved } -> weight pieces(kt+2)
    bf16x8 xg[TM * 2], wg[TN * 2];               // the second fragment set (xf / wf are the first)
    constexpr int NR = 2 * (TM + TN);            // fragment reads per stage
    auto read_one = [&](auto PAR, auto RR, const char* px, const char* pw) {
      constexpr int P = decltype(PAR)::value, R = decltype(RR)::value;
      constexpr int sh = R / (TM + TN), q = R % (TM + TN);
      const int coff = ((4 * sh + kq) ^ sw) << 4;
      if constexpr (q < TM) {
        if constexpr (P == 0) xf[sh][q] = *(const bf16x8*)(px + q * 16 * 128 + coff); else xg[sh * TM + q] = *(const bf16x8*)(px + q * 16 * 128 + coff);
      } else {
        if constexpr (P == 0) wf[sh][q - TM] = *(const bf16x8*)(pw + (q - TM) * 16 * 128 + coff); else wg[sh * TN + q - TM] = *(const bf16x8*)(pw + (q - TM) * 16 * 128 + coff);
      }
    };
    auto reads_only = [&](auto PAR, int b) {
      const char* px = smem + b * STAGE + xbase;
      const char* pw = smem + b * STAGE + wbase;
      [&]<int... R>(std::integer_sequence<int, R...>) { (read_one(PAR, std::integral_constant<int, R>{}, px, pw), ...); }(std::make_integer_sequence<int, NR>{});
    };
    // MFMAs of the set 1-P with the reads into set P behind every second one
    auto mma_reads = [&](auto PAR, int b, bool do_reads) {
      constexpr int P = decltype(PAR)::value;
      const char* px = smem + b * STAGE + xbase;
      const char* pw = smem + b * STAGE + wbase;
      [&]<int... I>(std::integer_sequence<int, I...>) {
        ([&] {
          constexpr int sh = I / (TN * TM), j = (I / TM) % TN, i = I % TM;
          if constexpr (P == 1) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[sh][j], xf[sh][i], acc[j][i], 0, 0, 0);
          else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wg[sh * TN + j], xg[sh * TM + i], acc[j][i], 0, 0, 0);
          if constexpr ((I & 1) == 1 && I / 2 < NR) {
            if (do_reads) read_one(PAR, std::integral_constant<int, I / 2>{}, px, pw);
          }
        }(), ...);
      }(std::make_integer_sequence<int, 2 * TN * TM>{});
#pragma unroll
      for (int g = 0; g < NR; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);     // two MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // one DS read
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2 * TN * TM - 2 * NR, 0);
    };
    static_assert(2 * TN * TM >= 2 * NR, "one read behind every second MFMA");
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    // stage 0: nothing to multiply yet
    if (nk > 1) wait_vmcnt<B_IT>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    reads_only(P0{}, 0);
    if (NST - 1 < nk) issue_b(NST - 1);
    buf = 1;
    auto step = [&](auto PAR, int kt) {          // stage kt >= 1: reads(kt) -> set PAR, MFMA(kt-1) from the other set
      if (kt + 1 < nk) wait_vmcnt<B_IT>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of stage kt-1 are in registers (and nobody refills a slot I still read)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      mma_reads(PAR, buf, true);
      if (kt + NST - 1 < nk) issue_b(buf >= 1 ? buf - 1 : NST - 1);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    };
    int kt = 1;
    for (; kt + 1 < nk; kt += 2) { step(P1{}, kt); step(P0{}, kt + 1); }
    if (kt < nk) {
      step(P1{}, kt);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mma_reads(P0{}, 0, false);                 // MFMA(nk-1) from set 1
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mma_reads(P1{}, 0, false);                 // MFMA(nk-1) from set 0
    }
  } else {
    // ---- late half: [barrier] MFMA(kt-1) from registers -> weight pieces(kt+2) -> fragments(kt) (kept for the next stage)
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) wait_vmcnt<B_IT>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my fragment reads of stage kt-1 are done before anyone refills
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt > 0) mma();
      if (kt + NST - 1 < nk) issue_b(buf >= 1 ? buf - 1 : NST - 1);
      load_frags(buf);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    mma();
  }
  conv2_epilogue<BN, BM, NT, TN, TM, FBN, EPI>(a, smem, acc, true, wm, wn, tid, lane, m0, n0, m_end, mt, tile);
}

// ---- experiment (SIMT_CONV2_INTER=1): the LDS-DMA pieces of a stage issued INSIDE the wave's MFMA burst (one piece behind every IV MFMAs, pinned
// with sched_group_barrier), their source addresses computed in front of the burst (no VALU between the MFMAs), weight pieces from a scalar
// row-group base + one per-lane offset.  Same ring / fragment reads / MFMA order as conv_igemm2_body: bit-identical.  Why: in-loop stamps
// (profiles/r05_conv_attribution_stamps.txt) -- per 64-deep stage each wave spends ~440 clocks issuing its 6-7 pieces behind or in front of
// its 680-clock MFMA burst, and the matrix pipe idles 26 % of the stage; a piece has no register destination, so it can leave between MFMAs.
template <int BN, int TMP, int EPI>
__global__ __launch_bounds__(512, 2) void conv_igemm2i_kernel(Conv2KArgs a) {
  constexpr int NT = 512, NST = 3;
  constexpr int WM = 2, WN = 4;
  constexpr int TM = TMP, TN = BN / WN / 16;
  constexpr int BM = WM * TM * 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = BN * 8 / NT;
  constexpr bool A_TAIL = (BM * 8) % NT != 0;
  constexpr int NP = A_IT + B_IT;                  // pieces per wave and stage (the tail pixel piece: waves 0-3 only)
  constexpr int NM = 2 * TN * TM;                  // MFMAs per wave and stage
  constexpr int IV = NM / (NP + 1);                // MFMAs in front of every piece
  static_assert(BN == 256, "experiment: the wide tile only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nwg = a.ntiles_m * a.ntiles_n;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int mt = tile / a.ntiles_n, nt = tile - mt * a.ntiles_n;
  const int m0 = mt * a.rows, n0 = nt * BN;
  const int m_end = min(a.M, m0 + a.rows);
  const int c_pos = tid & 7;
  const int a_cg = c_pos ^ (((tid >> 3) >> 1) & 7);
  const unsigned b_off0 = (unsigned)(tid >> 3) * (unsigned)a.wrow_bytes + (unsigned)(a_cg * 16);      // row group i adds the uniform i * 64 * wrow_bytes
  unsigned a_off[A_IT];
  unsigned long long a_ok[A_IT];
  int tdy[9], tdx[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { tdy[t] = a.dy[t]; tdx[t] = a.dx[t]; }
  const int ntaps = a.ntaps;
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = m0 + i * (NT / 8) + (tid >> 3);
    a_ok[i] = 0ull;
    a_off[i] = 0u;
    if (m < m_end) {
      int b, r, oy, ox;
      fast_divmod(m, a.Ho * a.Wo, a.rcp_hw, b, r);
      fast_divmod(r, a.Wo, a.rcp_wo, oy, ox);
      const int iy = oy * a.stride, ix = ox * a.stride;
      a_off[i] = (unsigned)(((b * a.H + iy) * a.W + ix)) * (unsigned)a.pix_bytes + (unsigned)(a_cg * 16);
      unsigned long long msk = 0ull;
      if (ntaps <= 9) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yy = iy + tdy[t], xx = ix + tdx[t];
          if (t < ntaps && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) msk |= (1ull << t);
        }
      } else {
        for (int t = 0; t < ntaps; ++t) {
          const int yy = iy + a.dy[t], xx = ix + a.dx[t];
          if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) msk |= (1ull << t);
        }
      }
      a_ok[i] = msk;
    }
  }
  const char* zsrc = a.zero + a_cg * 16;
  const bool a_tail_wave = !A_TAIL || wave < (BM * 8 - (A_IT - 1) * NT) / 64;
  int ld_tap = 0, ld_kc = 0;
  // sources of the NEXT stage's pieces, computed ahead of the burst that issues them
  const char* asrc[A_IT];
  unsigned bvo = 0u;
  auto prep = [&]() {
    const int toff = a.toff[ld_tap] + ld_kc * 128;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const bool ok = (a_ok[i] >> ld_tap) & 1ull;
      asrc[i] = ok ? a.x + (unsigned)(a_off[i] + (unsigned)toff) : zsrc;
    }
    bvo = b_off0 + (unsigned)(ld_tap * a.kc_per_tap + ld_kc) * 128u;
    if (++ld_tap == a.ntaps) { ld_tap = 0; ++ld_kc; }
  };
  auto fire = [&](auto PP, char* sbase) {          // piece PP of the prepared stage: pixel pieces first, then weight pieces
    constexpr int P = decltype(PP)::value;
    if constexpr (P < A_IT) {
      if (P < A_IT - 1 || a_tail_wave)
        __builtin_amdgcn_global_load_lds(GPTR(asrc[P]), LPTR(sbase + (P * NT + wave * 64) * 16), 16, 0, 0);
    } else {
      constexpr int i = P - A_IT;
      const char* wrows = a.w + (size_t)(unsigned)(n0 + i * (NT / 8)) * (unsigned)a.wrow_bytes;      // uniform
      __builtin_amdgcn_global_load_lds(GPTR(wrows + bvo), LPTR(sbase + A_BYTES + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
  };
  auto fire_all = [&](char* sbase) {
    [&]<int... P>(std::integer_sequence<int, P...>) { (fire(std::integral_constant<int, P>{}, sbase), ...); }(std::make_integer_sequence<int, NP>{});
  };
  auto wait_stage = [&](bool more) {
    if (!more) { wait_vmcnt<0>(); return; }
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
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[s][j] = *(const bf16x8*)(pw + j * 16 * 128 + coff);
    }
  };
  // the MFMA burst of one stage with the prepared pieces (if any) inside it
  auto burst = [&](char* sbase, bool pieces) {
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        constexpr int sh = I / (TN * TM), j = (I / TM) % TN, i = I % TM;
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[sh][j], xf[sh][i], acc[j][i], 0, 0, 0);
        if constexpr ((I + 1) % IV == 0 && (I + 1) / IV <= NP) {
          if (pieces) fire(std::integral_constant<int, (I + 1) / IV - 1>{}, sbase);
        }
      }(), ...);
    }(std::make_integer_sequence<int, NM>{});
#pragma unroll
    for (int g = 0; g < NP; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, IV, 0);      // IV MFMAs
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one vector-memory read (the LDS-DMA piece)
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM - NP * IV, 0);
  };
  prep();
  fire_all(smem);
  if (nk > 1) { prep(); fire_all(smem + STAGE); }
  int buf = 0;
  if (wave < 4) {
    // ---- early half: [barrier] fragments(kt) -> sources(kt+2) -> { MFMA(kt) | pieces(kt+2) }
    for (int kt = 0; kt < nk; ++kt) {
      wait_stage(kt + 1 < nk);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      load_frags(buf);
      const bool more = kt + NST - 1 < nk;
      if (more) prep();
      burst(smem + (buf >= 1 ? buf - 1 : NST - 1) * STAGE, more);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else {
    // ---- late half: [barrier] { MFMA(kt-1) | pieces(kt+2) } -> fragments(kt) -> sources(kt+3)
    bool have = false;
    {
      // stage 0 has nothing to multiply: its pieces go out plainly
      wait_stage(nk > 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (NST - 1 < nk) { prep(); fire_all(smem + (NST - 1) * STAGE); }
      load_frags(0);
      buf = 1;
      have = 1 + NST - 1 < nk;
      if (have) prep();
    }
    for (int kt = 1; kt < nk; ++kt) {
      wait_stage(kt + 1 < nk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      burst(smem + (buf >= 1 ? buf - 1 : NST - 1) * STAGE, have);
      load_frags(buf);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
      have = kt + 1 + NST - 1 < nk;
      if (have) prep();
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    burst(smem, false);
  }
  conv2_epilogue<BN, BM, NT, TN, TM, 0, EPI>(a, smem, acc, true, wm, wn, tid, lane, m0, n0, m_end, mt, tile);
}

template <int TM, int EPI, int PIPE>
static int launch_roles(const Conv2KArgs& k, size_t lds, hipStream_t st) {
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm2r_kernel<256, TM, 0, EPI, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm2r_kernel<256, TM, 0, EPI, PIPE>), dim3(k.ntiles_m * k.ntiles_n), dim3(512), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

template <int TM, int EPI>
static int launch_inter(const Conv2KArgs& k, size_t lds, hipStream_t st) {
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm2i_kernel<256, TM, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm2i_kernel<256, TM, EPI>), dim3(k.ntiles_m * k.ntiles_n), dim3(512), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

bool simt_conv2_roles_launch(const Conv2KArgs& k, int tm, int epi, size_t lds, hipStream_t st, int* rc) {
  static const int v = getenv("SIMT_CONV2_ROLES") ? atoi(getenv("SIMT_CONV2_ROLES")) : 0;
  static const int inter = getenv("SIMT_CONV2_INTER") ? atoi(getenv("SIMT_CONV2_INTER")) : 0;
  if (inter && epi >= 1 && epi <= 3) {
#define SIMT_INTER_CASE(T, E) if (tm == T && epi == E) { *rc = launch_inter<T, E>(k, lds, st); return true; }
    SIMT_INTER_CASE(5, 1) SIMT_INTER_CASE(5, 2) SIMT_INTER_CASE(5, 3) SIMT_INTER_CASE(4, 1) SIMT_INTER_CASE(4, 2) SIMT_INTER_CASE(4, 3)
#undef SIMT_INTER_CASE
  }
  if (v != 1 && v != 2) return false;
#define SIMT_ROLES_CASE(T, E) if (tm == T && epi == E) { *rc = v == 2 ? launch_roles<T, E, 1>(k, lds, st) : launch_roles<T, E, 0>(k, lds, st); return true; }
  SIMT_ROLES_CASE(5, 1) SIMT_ROLES_CASE(5, 2) SIMT_ROLES_CASE(5, 3) SIMT_ROLES_CASE(5, 5)
  SIMT_ROLES_CASE(4, 1) SIMT_ROLES_CASE(4, 2) SIMT_ROLES_CASE(4, 3) SIMT_ROLES_CASE(4, 5)
#undef SIMT_ROLES_CASE
  return false;
}
