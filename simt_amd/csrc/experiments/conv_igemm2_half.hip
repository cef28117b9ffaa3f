// Round-5 experiment on conv_igemm2_kernel<256, *, 3>, -DSIMT_ABLATION builds only, SIMT_CONV2_HALF=1 | 2.
// The fourth schedule of the K stage, and the first in which NO fragment read is exposed: the two K halves of a 64-deep stage (two
// v_mfma_f32_16x16x32_bf16 K steps) use DISJOINT fragment registers (xf[s][*], wf[s][*], s = 0 | 1: 36 VGPRs each, the product kernel's 72), so the
// half that is being multiplied and the half that is being read are double-buffered against each other WITHOUT a second register set:
//     [barrier kt: stage kt landed]   { MFMA k-half 1 of stage kt-1  |  ds_read k-half 0 of stage kt  (+ the early waves' LDS-DMA pieces of kt+2) }
//                                     { MFMA k-half 0 of stage kt    |  ds_read k-half 1 of stage kt  (+ the late waves' pieces of kt+2) }
// All eight waves run the same code (no early / late phase offset), one read behind each of the first TM + TN MFMAs of a half, the pieces
// behind the following ones (sched_group_barrier pins the order).  Same ring, same LDS image, same MFMA order per accumulator (k-half 0
// then 1 of every stage): bit-identical to the product kernel.  The stamps of the product schedule (profiles/r05_conv_attribution.txt,
// section 2) put the late wave on the critical path with ~340 clocks of fragment reads behind its burst and ~230 at the barrier, the matrix pipe
// idle 26 % of the stage; the pipelined variant of conv_igemm2_roles.hip hid them with a second register set (256 VGPRs, slower).
// HALF = 2: pieces not interleaved -- issued by every wave in front of half A (what the product's early waves do).
#include "../conv2_common.h"
#include "../conv2_epilogue.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

// KS = 1 (SIMT_CONV2_HALF=3, <256, 5, statistics> only): wave 0 and wave 4 of every workgroup add up the clocks (s_memtime) they spend per stage
// in the counted vmcnt wait (the stage's LDS-DMA pieces landing) and in the barrier behind it: g_hstamps[(block * 2 + late) * 4] = {vmcnt wait,
// barrier, whole K loop, stages}.
static __device__ unsigned long long g_hstamps[8192 * 8];
extern "C" int simt_debug_hstamps(unsigned long long* out, int nblocks) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hstamps), (size_t)nblocks * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}

extern "C" int simt_debug_stamps_half(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int simt_debug_stamps_half_rt(unsigned long long* out, int n) {      // s_memrealtime ticks (100 MHz) between stamps 0 and 6, one per workgroup
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_rt), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}

template <int BN, int TMP, int EPI, int PLAIN, int KS = 0>
__global__ __launch_bounds__(512, 2) void conv_igemm2h_kernel(Conv2KArgs a) {
  constexpr int NT = 512, NST = 3;
  constexpr int WM = 2, WN = 4;
  constexpr int TM = TMP, TN = BN / WN / 16;
  constexpr int BM = WM * TM * 16;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = BN * 8 / NT;
  constexpr bool A_TAIL = (BM * 8) % NT != 0;
  constexpr int NP = A_IT + B_IT;                  // pieces per wave and stage (the tail pixel piece: waves 0-3 only)
  constexpr int NH = TN * TM;                      // MFMAs per wave and K half
  constexpr int NR = TM + TN;                      // fragment reads per wave and K half
  static_assert(BN == 256 && NR + NP <= NH, "experiment: the wide tile only; reads then pieces fit behind the MFMAs of a half");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  STAMP(0);
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
      if (ntaps == 1 && tdy[0] == 0 && tdx[0] == 0) {
        msk = 1ull;
      } else if (ntaps <= 9) {
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
  // the tap offsets as ONE register, lane t = a.toff[t]: read back with v_readlane.  (A scalar load of a.toff[ld_tap] inside the loop shares lgkmcnt
  // with the fragment reads and returns out of order: the compiler then waits lgkmcnt(0) in the middle of the MFMAs.)
  const int toff_lane = ((const __attribute__((address_space(4))) int*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                                        __builtin_offsetof(Conv2KArgs, toff)))[lane < SIMT_MAX_TAPS ? lane : 0];
  // sources of the NEXT stage's pieces, computed ahead of the half that issues them (no VALU between its MFMAs)
  const char* asrc[A_IT];
  unsigned bvo = 0u;
  auto prep = [&]() {
    const int toff = __builtin_amdgcn_readlane(toff_lane, ld_tap) + ld_kc * 128;
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
  // read number R of K half S, in the order the MFMAs of that half need them: wf[S][0], xf[S][0 .. TM-1], wf[S][1 .. TN-1]
  auto read_one = [&](auto SS, auto RR, const char* px, const char* pw, int coff) {
    constexpr int S = decltype(SS)::value, R = decltype(RR)::value;
    if constexpr (R == 0) wf[S][0] = *(const bf16x8*)(pw + coff);
    else if constexpr (R <= TM) xf[S][R - 1] = *(const bf16x8*)(px + (R - 1) * 16 * 128 + coff);
    else wf[S][R - TM] = *(const bf16x8*)(pw + (R - TM) * 16 * 128 + coff);
  };
  auto reads_only = [&](auto SS, const char* stage) {
    constexpr int S = decltype(SS)::value;
    const char* px = stage + xbase;
    const char* pw = stage + wbase;
    const int coff = ((4 * S + kq) ^ sw) << 4;
    [&]<int... R>(std::integer_sequence<int, R...>) { (read_one(SS, std::integral_constant<int, R>{}, px, pw, coff), ...); }(std::make_integer_sequence<int, NR>{});
  };
  auto mma_only = [&](auto SS) {
    constexpr int S = decltype(SS)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[S][j], xf[S][i], acc[j][i], 0, 0, 0);
  };
  // One half: the MFMAs of K half 1-S (registers), the reads of K half S of `stage` behind the first NR of them, then (PIECES) the prepared
  // pieces into `next` behind the following NP.
  auto half = [&](auto SS, auto PC, const char* stage, char* next) {
    constexpr int S = decltype(SS)::value;
    constexpr bool PIECES = decltype(PC)::value;
    const char* px = stage + xbase;
    const char* pw = stage + wbase;
    const int coff = ((4 * S + kq) ^ sw) << 4;
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ([&] {
        constexpr int j = I / TM, i = I % TM;
        acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1 - S][j], xf[1 - S][i], acc[j][i], 0, 0, 0);
        if constexpr (I < NR) read_one(SS, std::integral_constant<int, I>{}, px, pw, coff);
        else if constexpr (PIECES && I - NR < NP) fire(std::integral_constant<int, I - NR>{}, next);
      }(), ...);
    }(std::make_integer_sequence<int, NH>{});
#pragma unroll
    for (int g = 0; g < NR; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // one DS read
    }
    if constexpr (PIECES) {
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // one vector-memory read (the LDS-DMA piece)
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NH - NR - NP, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x008, NH - NR, 0);
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;

  STAMP(1);
  prep();
  fire_all(smem);
  if (nk > 1) { prep(); fire_all(smem + STAGE); }
  // the K loop for one role: EARLY waves put their pieces into half A, late waves into half B (PLAIN: every wave in front of half A)
  auto run = [&](auto EARLY) {
    constexpr bool E = decltype(EARLY)::value;
    unsigned long long t_vm = 0ull, t_bar = 0ull, t_begin = 0ull;
    if constexpr (KS != 0) t_begin = __builtin_amdgcn_s_memtime();
    // ---- stage 0: nothing to multiply yet
    wait_stage(nk > 1);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (NST - 1 < nk) { prep(); fire_all(smem + (NST - 1) * STAGE); }
    reads_only(S0{}, smem);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // { MFMA k0(0) | read k1(0) }
    half(S1{}, No{}, smem, smem);
    int buf = 1;
    int kt = 1;
    for (; kt + NST - 1 < nk; ++kt) {             // stages that still have a stage kt+2 to fetch
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of stage kt-1 are in registers: its slot may be refilled behind the barrier
      if constexpr (KS != 0) {
        const unsigned long long s0 = __builtin_amdgcn_s_memtime();
        wait_stage(true);
        const unsigned long long s1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long s2 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t_vm += s1 - s0;
        t_bar += s2 - s1;
      } else {
        wait_stage(true);
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("" ::: "memory");
      const char* stage = smem + buf * STAGE;
      char* next = smem + (buf >= 1 ? buf - 1 : NST - 1) * STAGE;
      prep();
      if constexpr (PLAIN != 0) {
        fire_all(next);
        half(S0{}, No{}, stage, next);
        half(S1{}, No{}, stage, next);
      } else if constexpr (PLAIN == 2) {          // every wave's pieces inside half A: the tail of the slower wave of a SIMD is bare MFMAs
        half(S0{}, Yes{}, stage, next);
        half(S1{}, No{}, stage, next);
      } else {
        half(S0{}, std::integral_constant<bool, E>{}, stage, next);
        half(S1{}, std::integral_constant<bool, !E>{}, stage, next);
      }
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    for (; kt < nk; ++kt) {                       // the last two stages: nothing left to fetch
      wait_stage(kt + 1 < nk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* stage = smem + buf * STAGE;
      half(S0{}, No{}, stage, smem);
      half(S1{}, No{}, stage, smem);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    mma_only(S1{});                               // k half 1 of the last stage
    if constexpr (KS != 0) {
      if ((tid == 0 || tid == 256) && blockIdx.x < 4096) {
        unsigned long long* o = g_hstamps + (blockIdx.x * 2 + (tid == 256 ? 1 : 0)) * 4;
        o[0] = t_vm; o[1] = t_bar; o[2] = __builtin_amdgcn_s_memtime() - t_begin; o[3] = (unsigned long long)nk;
      }
    }
  };
  if (wave < 4) run(Yes{}); else run(No{});
  STAMP(3);
  conv2_epilogue<BN, BM, NT, TN, TM, 0, EPI>(a, smem, acc, true, wm, wn, tid, lane, m0, n0, m_end, mt, tile);
}

template <int TM, int EPI, int PLAIN, int KS = 0>
static int launch_half(const Conv2KArgs& k, size_t lds, hipStream_t st) {
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm2h_kernel<256, TM, EPI, PLAIN, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm2h_kernel<256, TM, EPI, PLAIN, KS>), dim3(k.ntiles_m * k.ntiles_n), dim3(512), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

bool simt_conv2_half_launch(const Conv2KArgs& k, int tm, int epi, size_t lds, hipStream_t st, int* rc) {
  static const int v = getenv("SIMT_CONV2_HALF") ? atoi(getenv("SIMT_CONV2_HALF")) : 0;
  if (v == 3 && tm == 5 && epi == 1) { *rc = launch_half<5, 1, 0, 1>(k, lds, st); return true; }      // the stamped builds (measurement of the waits)
  if (v == 5 && tm == 5 && epi == 1) { *rc = launch_half<5, 1, 2, 1>(k, lds, st); return true; }
  if (v != 1 && v != 2 && v != 4) return false;
#define SIMT_HALF_CASE(T, E) if (tm == T && epi == E) { *rc = v == 2 ? launch_half<T, E, 1>(k, lds, st) : v == 4 ? launch_half<T, E, 2>(k, lds, st) : launch_half<T, E, 0>(k, lds, st); return true; }
  SIMT_HALF_CASE(5, 1) SIMT_HALF_CASE(5, 2) SIMT_HALF_CASE(5, 3) SIMT_HALF_CASE(5, 5)
  SIMT_HALF_CASE(4, 1) SIMT_HALF_CASE(4, 2) SIMT_HALF_CASE(4, 3) SIMT_HALF_CASE(4, 5)
#undef SIMT_HALF_CASE
  return false;
}
