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
//     the host so that the grid is a whole number of 256-CU rounds (M = 37636: 236 tiles of 160 rows instead of 295
//     of 128 -> one round instead of two);
//   * the two waves of a SIMD run the K-stage in opposite order: waves 0-3 load-then-multiply, waves 4-7 multiply the
//     fragments they fetched in the previous stage first and load afterwards, so one wave's MFMA burst covers the other
//     wave's global_load_lds issue + ds_read latency (MI355X_MICROARCH "two waves per SIMD", item 9).
// LDS rows are 128 B; the 16-B chunk index is XOR-swizzled with (row>>1)&7 on the global SOURCE address and on the
// ds_read_b128 side (conflict-free 16-lane groups), the LDS image itself stays lane-linear as global_load_lds needs.
#ifndef SIMT_NT_STORES
#define SIMT_NT_STORES 1      // output rows as non-temporal stores (common.h)
#endif
#include "conv2_common.h"
#include "conv2_epilogue.h"
#include <stdlib.h>
#include <type_traits>

#ifdef SIMT_ABLATION       // in-kernel s_memtime stamps of THIS kernel (diagnostic builds only; conv2_common.h STAMP)
extern "C" int simt_debug_stamps(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
extern "C" int simt_debug_stamps_rt(unsigned long long* out, int n) {      // s_memrealtime ticks (100 MHz) between stamps 0 and 6, one per workgroup
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_rt), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// Round-3 experiments on this kernel (timing-ablation MODEs, loader waves LW = 4 / 8, weights straight into registers WD = 1, and the
// role-split conv_igemm3) live in csrc/experiments/ and are compiled only into -DSIMT_ABLATION builds (csrc/build.sh ABLATION=1);
// what they measured is in DESIGN.md section 9 and profiles/r03_conv_experiments.txt.  This file is the product kernel only.
// FBN = 1: the same kernel with the fused train-mode BatchNorm tail compiled in (conv2_epilogue.h; a.fbn_mode selects forward / backward).
// A separate instantiation so that the plain kernels keep their code (the main loop is sensitive to what surrounds it).
// EPI: compile-time epilogue flavour (conv2_epilogue.h): 0 generic, 1 statistics, 2 BatchNorm-backward reduce, 3 bias + ReLU, 4-8 the dgrad forms.
template <int BN, int TMP, int NSTP, int FBN, int EPI>
__device__ __forceinline__ void conv_igemm2_body(const Conv2KArgs& a, const int bid) {
  constexpr int NT = 512, NST = NSTP;   // NST = 3: one workgroup per CU, two stages in flight; NST = 2 (short-K, output-
                                        // bound shapes): two workgroups per CU so one's epilogue overlaps the other's loads
  constexpr int WM = (BN == 64) ? 4 : 2;          // waves along pixels
  constexpr int WN = 8 / WM;                       // waves along couts
  constexpr int TM = TMP, TN = BN / WN / 16;
  constexpr int BM = WM * TM * 16;                 // allocated pixel rows (128 or 160)
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = (BM * 8 + NT - 1) / NT, B_IT = BN * 8 / NT;   // 16-B chunks per thread per stage
  constexpr bool A_TAIL = (BM * 8) % NT != 0;      // BM = 160: the third A pass is only issued by waves 0-3
  constexpr int CP = BN * 2 + 8;                   // epilogue tile pitch in bytes (bf16 row + 8 B pad)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  STAMP(0);
  const int nwg = a.ntiles_m * a.ntiles_n;
  const int tile = xcd_remap(bid, nwg);
  const int mt = tile / a.ntiles_n, nt = tile - mt * a.ntiles_n;
  const int m0 = mt * a.rows, n0 = nt * BN;
  const int m_end = min(a.M, m0 + a.rows);

  // ---- hoisted A-gather metadata: chunk q = i*NT + tid -> row = q>>3, position q&7
  const int c_pos = tid & 7;
  const int a_cg = c_pos ^ (((tid >> 3) >> 1) & 7);          // (row>>1)&7 only depends on tid>>3 because NT/8 = 64 is even
  unsigned b_off[B_IT];
#pragma unroll
  for (int i = 0; i < B_IT; ++i) {
    const int row = i * (NT / 8) + (tid >> 3);
    b_off[i] = (unsigned)(n0 + row) * (unsigned)a.wrow_bytes + (unsigned)(a_cg * 16);
  }
  // Stage 0's weight pieces need no pixel addressing: they leave NOW, so that their L2 latency runs under the address arithmetic below
  // (stamps, round 4: 2.1-3.4 us between "addressing done" and "first stage landed").  Issue order B(0), A(0), A(1), B(1): the counted wait of
  // the first K step (<= one stage outstanding) still retires exactly stage 0.
#pragma unroll
  for (int i = 0; i < B_IT; ++i)
    __builtin_amdgcn_global_load_lds(GPTR(a.w + b_off[i]), LPTR(smem + A_BYTES + (i * NT + wave * 64) * 16), 16, 0, 0);
  unsigned a_off[A_IT];
  unsigned long long a_ok[A_IT];
  // The tap offsets as scalars, fetched ONCE with wide scalar loads (round 4: stamps showed 14.7 k clocks = 6.7 us between kernel start and
  // "addressing done" for a 3x3 conv -- the validity loop below re-read a.dy[t] / a.dx[t] from the kernel arguments with a dependent
  // s_load + wait per tap and row, 27 round trips).  Nine pairs cover every 1x1 / 3x3 conv; longer tap lists keep the loop.
  int tdy[9], tdx[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) { tdy[t] = a.dy[t]; tdx[t] = a.dx[t]; }
  const int ntaps = a.ntaps;
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
      if (ntaps == 1 && tdy[0] == 0 && tdx[0] == 0) {
        msk = 1ull;          // 1x1 convs (two thirds of the launches): the one tap is the pixel itself, always inside -- the nine predicated tap tests
                             // below were ~190 of this prologue's ~250 vector instructions per thread (uniform branch)
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

  int ld_tap = 0, ld_kc = 0;                     // position of the NEXT stage to be issued: (64-channel chunk, tap), taps innermost
  auto issue = [&](int buf, bool weights = true) {
    const int toff = a.toff[ld_tap] + ld_kc * 128;
    char* sbase = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      if (i == A_IT - 1 && !a_tail_wave) break;
      const bool ok = (a_ok[i] >> ld_tap) & 1ull;
      const char* src = ok ? a.x + (unsigned)(a_off[i] + (unsigned)toff) : zsrc;
      __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sbase + (i * NT + wave * 64) * 16), 16, 0, 0);
    }
    // K-stage order (64-channel chunk, tap): a pixel row's 128-byte line is read by all taps in CONSECUTIVE stages -- an L2 reuse distance of
    // one stage of the XCD's workgroups (~0.7 MB) instead of kc_per_tap stages (~2.8 MB of the 4 MB L2 at Cin = 256, with the weight stream on
    // top): 3x3 convs 1-2.5 % faster, the step 26.39 -> 26.10 ms.  Weights are packed K-contiguous as (tap, channel): this stage's 64 columns
    // start at (tap * chunks + chunk) * 128 bytes.  (Compile-time only: the same order behind a run-time flag cost every launch 17-28 %.)
    const unsigned wk = (unsigned)(ld_tap * a.kc_per_tap + ld_kc) * 128u;
    if (weights) {
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
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
  STAMP(1);
  issue(0, false);                               // (its weight pieces are already in flight)
  if (NST == 3 && nk > 1) issue(1);
  int buf = 0;
  if (wave < 4) {
    // ---- early half: [barrier] issue(kt+2) -> fragments(kt) -> MFMA(kt)
    for (int kt = 0; kt < nk; ++kt) {
      wait_stage(kt + 1 < nk);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt == 0) STAMP(2);
      load_frags(buf);
      if (kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);     // stage kt+NST-1 -> buffer (buf-1) mod NST
      mma();
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
  } else {
    // ---- late half: [barrier] MFMA(kt-1) from registers -> issue(kt+2) -> fragments(kt) (kept for the next stage)
    for (int kt = 0; kt < nk; ++kt) {
      wait_stage(kt + 1 < nk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my fragment reads of stage kt-1 are done before anyone refills
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt > 0) mma();
      if (kt + NST - 1 < nk) issue(buf >= 1 ? buf - 1 : NST - 1);
      load_frags(buf);
      buf = (buf + 1 == NST) ? 0 : buf + 1;
    }
    mma();
  }

  STAMP(3);
  conv2_epilogue<BN, BM, NT, TN, TM, FBN, EPI>(a, smem, acc, true, wm, wn, tid, lane, m0, n0, m_end, mt, tile);
}

template <int BN, int TMP, int NSTP, int FBN = 0, int EPI = 0>
__global__ __launch_bounds__(512, (NSTP == 2 ? 4 : 2)) void conv_igemm2_kernel(Conv2KArgs a) {
  conv_igemm2_body<BN, TMP, NSTP, FBN, EPI>(a, (int)blockIdx.x);
}

// Round 5: TWO independent convs of identical geometry in ONE launch -- the trainable net's and the frozen net's conv of the
// same layer see the same shapes (tools/trainV2_simt.py:351-353 and :370 call the same ResNetMulti.forward, model/deeplab_multi.py:172-192).
// Workgroups [0, nwg) run problem 0, [nwg, 2 nwg) problem 1, each with its own compile-time epilogue flavour (the branch is workgroup-uniform
// and sits in front of everything: nothing inside the K loop knows about it).  Measured on the production shapes (one launch of twice the
// workgroups against two launches): 11 us per pair on the 3x3 256 -> 256, 2.5 us on 1x1 1024 -> 256 (profiles/r05_conv_attribution.txt section 5): the second
// round of workgroups starts as the first drains, one launch ramp / drain / boundary instead of two.
template <int BN, int TMP, int NSTP, int EPI0, int EPI1>
__global__ __launch_bounds__(512, (NSTP == 2 ? 4 : 2)) void conv_igemm2_pair_kernel(Conv2KArgs a0, Conv2KArgs a1) {
  // problem 1 starts at the next multiple of 8 (the <= 7 workgroups in between exit at once): xcd_remap reads the XCD off the low three bits of
  // the id it is given, and blockIdx.x - nwg0 has the hardware's only when nwg0 % 8 == 0 (255 tiles: it was shifted by one XCD)
  const int nwg0 = a0.ntiles_m * a0.ntiles_n, base1 = (nwg0 + 7) & ~7;
  if ((int)blockIdx.x < nwg0) conv_igemm2_body<BN, TMP, NSTP, 0, EPI0>(a0, (int)blockIdx.x);
  else if ((int)blockIdx.x >= base1) conv_igemm2_body<BN, TMP, NSTP, 0, EPI1>(a1, (int)blockIdx.x - base1);
}

#ifdef SIMT_ABLATION
// csrc/experiments/conv_igemm2_abl.hip: returns true when an experiment build (SIMT_CONV2_MODE / SIMT_CONV2_LW / SIMT_WDIRECT / SIMT_IGEMM3)
// took the launch; *rc is its result
bool simt_conv2_abl_launch(const Conv2KArgs& k, int bn, int tm, int nst, hipStream_t st, int* rc);
int simt_conv2_abl_wants_frag(const simt_conv_desc* d);
#endif
#ifdef SIMT_ABLATION
bool simt_conv2_roles_launch(const Conv2KArgs& k, int tm, int epi, size_t lds, hipStream_t st, int* rc); bool simt_conv2_half_launch(const Conv2KArgs& k, int tm, int epi, size_t lds, hipStream_t st, int* rc);      // round-5 experiments, ablation builds only
#endif
template <int BN, int TM, int NST = 3, int FBN = 0, int EPI = 0>
static int launch_conv2e(const Conv2KArgs& k, hipStream_t st) {
  constexpr int WM = (BN == 64) ? 4 : 2;
  constexpr int BM = WM * TM * 16;
  const size_t ring = NST * (size_t)(BM * 128 + BN * 128);
  const size_t epi = (size_t)BM * (BN * 2 + 8) + (size_t)(512 / (BN / 8)) * 2 * BN * 4 + (FBN ? 32 * 8 * 3 * sizeof(double) + 16 + 2 * BN * sizeof(float) : 0);
  const size_t lds = ring > epi ? ring : epi;
#ifdef SIMT_ABLATION
  if constexpr (NST == 3 && BN == 256 && FBN == 0 && (EPI == 1 || EPI == 2 || EPI == 3 || EPI == 5)) {
    int rc;
    if (simt_conv2_half_launch(k, TM, EPI, lds, st, &rc) || simt_conv2_roles_launch(k, TM, EPI, lds, st, &rc)) return rc;      // csrc/experiments/conv_igemm2_roles.hip (SIMT_CONV2_ROLES=1 | 2, SIMT_CONV2_INTER=1)
  }
#endif
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm2_kernel<BN, TM, NST, FBN, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm2_kernel<BN, TM, NST, FBN, EPI>), dim3(k.ntiles_m * k.ntiles_n), dim3(512), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// Which compile-time epilogue flavour computes exactly what the descriptor asks for (0: only the generic one does)
static int conv2_flavour(const Conv2KArgs& k) {
  static const int off = getenv("SIMT_CONV2_GENERIC_EPI") ? atoi(getenv("SIMT_CONV2_GENERIC_EPI")) : 0;     // A/B switch
  if (off || k.out_f32 || k.Nstore != k.Cout || k.Cout % 8) return 0;
  if (k.mask) return (!k.res && !k.bias && !k.relu && !k.stats && !k.bnr_mode) ? 8 : 0;
  if (k.bnr_mode == 3 && !k.res_bits && !k.bias && !k.relu && !k.stats) return 7;       // (res or not: a run-time pointer test)
  if (k.res) {
    if (k.bias || k.relu || k.stats) return 0;
    if (k.res_bits && k.bnr_mode == 3) return 4;
    if (!k.res_bits && !k.bnr_mode) return 6;
    return 0;
  }
  if (k.stats && !k.bias && !k.relu && !k.bnr_mode) return 1;
  if (k.bnr_mode == 2 && !k.stats && !k.bias && !k.relu) return 2;
  if (k.bias && k.relu && !k.stats && !k.bnr_mode) return 3;
  if (!k.bias && !k.relu && !k.stats && !k.bnr_mode) return 5;
  return 0;
}

template <int BN, int TM, int NST = 3>
static int launch_conv2(const Conv2KArgs& k, hipStream_t st) {
#ifdef SIMT_ABLATION
  { int rc; if (!k.fbn_mode && simt_conv2_abl_launch(k, BN, TM, NST, st, &rc)) return rc; }
#endif
  const int e = conv2_flavour(k);
  if constexpr (NST == 3) {
    if constexpr (BN == 256) {
      if (k.fbn_mode == 1) return launch_conv2e<BN, TM, NST, 1, 1>(k, st);
    }
    if (k.fbn_mode == 2) return launch_conv2e<BN, TM, NST, 1, 2>(k, st);
    if (e == 1) return launch_conv2e<BN, TM, NST, 0, 1>(k, st);
    if (e == 2) return launch_conv2e<BN, TM, NST, 0, 2>(k, st);
    if (e == 3) return launch_conv2e<BN, TM, NST, 0, 3>(k, st);
  }
  // the dgrad flavours: all tile shapes (the 2-slot short-K kernels run nothing else in a training step)
  if (e == 4) return launch_conv2e<BN, TM, NST, 0, 4>(k, st);
  if (e == 5) return launch_conv2e<BN, TM, NST, 0, 5>(k, st);
  if (e == 6) return launch_conv2e<BN, TM, NST, 0, 6>(k, st);
  if (e == 7) return launch_conv2e<BN, TM, NST, 0, 7>(k, st);
  if (e == 8) return launch_conv2e<BN, TM, NST, 0, 8>(k, st);
  return launch_conv2e<BN, TM, NST, 0, 0>(k, st);
}

// Pixel rows per tile: 128 (TM = 4) or, for the 2x4 wave layouts, up to 160 (TM = 5) when that saves a whole round of
// the 256 CUs.  Cost model: rounds * allocated rows; among equal costs the LARGEST tile (round 6).  A TM = 5 workgroup multiplies 160 rows
// whatever it is given: at M = 37 636 rounds 1-5 took the first one-round size, 148 rows -> 255 tiles, i.e. 7.5 % of the matrix work of every
// wide conv spent on rows that do not exist -- on a chip whose clock is set by the energy of exactly those launches (profiles/r05_power_clock.txt)
// -- and left ONE CU to the other stream.  160 rows -> 236 tiles: no padding work, 20 CUs for the weight gradients / the frozen net beside it:
// the step 24.13 -> 23.68 ms (same box, alternating; profiles/r06_tile_rows.txt).  cu_budget = -1 restores the old choice (A/B).
static void pick_rows(int M, int ntn, bool allow160, int* rows, int* tm, int cu_budget) {
  const int CUS = (cu_budget > 0 && cu_budget < 256) ? cu_budget : 256;      // simt_conv_desc.cu_budget: CUs left beside a collective's kernels
  const bool first = cu_budget == -1;        // A/B only (engine: SIMT_PICK_ROWS_FIRST=1 at plan construction): rounds 1-5's first-minimum choice
  long best = -1;
  *rows = 128; *tm = 4;
  for (int r = 128; r <= (allow160 ? 160 : 128); r += 4) {
    const int tiles = (M + r - 1) / r;
    const long rounds = ((long)tiles * ntn + CUS - 1) / CUS;
    const long cost = rounds * (r <= 128 ? 128 : 160);
    if (best < 0 || cost < best || (cost == best && !first)) { best = cost; *rows = r; *tm = r <= 128 ? 4 : 5; }
  }
}

extern "C" int simt_conv_fprop(const simt_conv_desc* d, simt_stream_t stream);      // conv_igemm.hip
bool simt_conv_stream_eligible(const simt_conv_desc* d);                  // conv1x1_stream.hip
int simt_conv_stream_launch(Conv2KArgs k, int npad, hipStream_t st);
bool simt_conv_rows_eligible(const simt_conv_desc* d);                    // conv1x1_rows.hip
bool simt_conv_rows_inbn_ok(const simt_conv_desc* d);
int simt_conv_rows_launch(Conv2KArgs k, int npad, hipStream_t st);
static bool rows_enabled() {
  static const int off = getenv("SIMT_NO_ROWS") ? atoi(getenv("SIMT_NO_ROWS")) : 0;     // A/B switch
  return !off;
}
static bool stream_enabled() {
#ifdef SIMT_ABLATION
  static const int off = getenv("SIMT_NO_STREAM") ? atoi(getenv("SIMT_NO_STREAM")) : 0;
  return !off;
#else
  return true;
#endif
}
struct Conv2Variant { int tile_n, tm, nst, rows, ntiles_n; bool stream, rowsk; };
static Conv2Variant pick_variant(const simt_conv_desc* d) {
  Conv2Variant v;
  const int M = d->B * d->Ho * d->Wo;
  // Short reductions with wide outputs (1x1 convs 256 -> 1024 and their dgrads) are bound by the output stream, not by
  // MFMA: run them as 128-column tiles with a 2-stage ring, two workgroups per CU.
  v.tile_n = d->tile_n;
  const long Kt = (long)d->ntaps * d->Cin;
  const bool short_k = d->tile_n == 256 && d->dtype_out == SIMT_BF16 && ((Kt <= 512 && d->Cout >= 512) || (Kt <= 128 && d->Cout >= 256));
#ifdef SIMT_ABLATION
  { static const int no_short = getenv("SIMT_NO_SHORTK") ? atoi(getenv("SIMT_NO_SHORTK")) : 0; if (no_short) { Conv2Variant w; w.stream = false; w.rowsk = false; w.tile_n = d->tile_n; w.nst = 3; w.ntiles_n = d->Npad / w.tile_n; w.tm = 4; pick_rows(M, w.ntiles_n, w.tile_n != 64, &w.rows, &w.tm, d->cu_budget); if (w.tile_n == 64) w.tm = 2; return w; } }
#endif
  // Round 4 experiment (SIMT_ROWS_1024=1; default OFF): the long-reduction 1x1 convs on dense rows (conv1 of layer 3 / 4: 1024 -> 256 / 512)
  // as whole 2-KB pixel rows streamed past register-resident weights (conv1x1_rows_kernel<32, 1, 4, ..., 1, 4, 1>: 128 weight registers per
  // wave leave room for 4 compute waves and 64-column workgroups only).  Parity green, and SLOWER: 45.6 us against 29.6-31.9 on
  // conv_igemm2_kernel<256, 5, 3> (one MFMA per k-step and wave, four workgroups re-reading every pixel row), the step +0.45 ms.
  static const int rows1024 = getenv("SIMT_ROWS_1024") ? atoi(getenv("SIMT_ROWS_1024")) : 0;
  const bool long_rows = rows1024 && d->ntaps == 1 && d->Cin == 1024 && d->Cout <= 512 && d->dtype_out == SIMT_BF16;
  v.rowsk = (short_k || long_rows) && rows_enabled() && simt_conv_rows_eligible(d);
  if (v.rowsk) { v.stream = false; v.tile_n = 256; v.nst = 6; v.ntiles_n = d->Npad / 256; v.tm = 2; v.rows = 128; return v; }
  v.stream = short_k && stream_enabled() && simt_conv_stream_eligible(d);
  if (v.stream) { v.tile_n = 128; v.nst = 3; v.ntiles_n = d->Npad / 128; v.tm = 4; v.rows = 128; return v; }
  if (short_k) v.tile_n = 128;
  v.nst = short_k ? 2 : 3;
  v.ntiles_n = d->Npad / v.tile_n;
  v.tm = 4;
  pick_rows(M, short_k ? (v.ntiles_n + 1) / 2 : v.ntiles_n, v.tile_n != 64, &v.rows, &v.tm, d->cu_budget);
  if (v.tile_n == 64) v.tm = 2;
  return v;
}

// Which instantiation simt_conv_fprop will run for this descriptor (profiling / reporting): 0 = conv_igemm_kernel (v1),
// otherwise conv_igemm2_kernel<bn, tm, nst>.
extern "C" int simt_conv_variant(const simt_conv_desc* d, int* bn, int* tm, int* nst) {
  const bool v2 = d->dtype_in == SIMT_BF16 && d->tile_n >= 64 &&
                  (d->dtype_out == SIMT_BF16 || (!d->bias && !d->res && !d->relu && !d->stats && !d->mask && d->tile_n == 256));
  if (!v2) { *bn = d->tile_n; *tm = 0; *nst = 2; return 0; }
  const Conv2Variant v = pick_variant(d);
  *bn = v.tile_n; *tm = v.tm; *nst = v.nst;
  return v.rowsk ? 5 : v.stream ? 4 : 2;
}

// Does the launch for d take its weights from d->w_frag (the weights-direct experiment)?  The engine asks before it allocates / packs the
// fragment-ordered copy.  Round 3 measured that form 15-40 % SLOWER than the LDS-staged one on every production shape (DESIGN.md
// section 9): it exists only in -DSIMT_ABLATION builds (csrc/experiments/), behind SIMT_WDIRECT=1; the product library always answers 0.
extern "C" int simt_conv_wants_frag(const simt_conv_desc* d) {
#ifdef SIMT_ABLATION
  return simt_conv2_abl_wants_frag(d);
#else
  (void)d;
  return 0;
#endif
}

// Which compile-time epilogue flavour (conv2_epilogue.h EPI) the launch for d runs: 0 generic, 1 statistics, 2 BatchNorm-backward reduce,
// 3 bias + ReLU, 4-8 the dgrad forms listed there (reporting: the kernel name is conv_igemm2_kernel<bn, tm, nst, fbn, epi>)
extern "C" int simt_conv_epilogue_flavour(const simt_conv_desc* d) {
  int bn, tm, nst;
  if (!d || simt_conv_variant(d, &bn, &tm, &nst) != 2) return 0;
  if (d->fbn) return d->fbn->mode == 1 ? 1 : 2;
  Conv2KArgs k;
  k.out_f32 = d->dtype_out == SIMT_F32; k.res = (const bf16_t*)d->res; k.mask = (const bf16_t*)d->mask; k.Nstore = d->Nstore; k.Cout = d->Cout;
  k.stats = d->stats; k.bias = d->bias; k.relu = d->relu; k.bnr_mode = d->bnr_mode; k.res_bits = d->res_bits;
  const int e = conv2_flavour(k);
  return (nst != 3 && e >= 1 && e <= 3) ? 0 : e;
}

// Fused BatchNorm (simt_fbn_desc): the wide / medium 3-slot kernels, every workgroup co-resident (one workgroup per CU: the ring takes the LDS)
static int device_cus() {
  static int cus[SIMT_MAX_DEVICES];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIMT_MAX_DEVICES) return 0;
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    cus[dev] = n;
  }
  return cus[dev];
}
extern "C" int simt_conv_fbn_ok(const simt_conv_desc* d) {
  int bn, tm, nst;
  if (!d || d->dtype_in != SIMT_BF16 || d->dtype_out != SIMT_BF16 || simt_conv_variant(d, &bn, &tm, &nst) != 2) return 0;
  if (nst != 3 || !(bn == 256 || d->bnr_mode == 2)) return 0;      // (the 128- / 64-column tiles are instantiated for the backward only)
  const Conv2Variant v = pick_variant(d);
  const int M = d->B * d->Ho * d->Wo;
  const long nwg = (long)((M + v.rows - 1) / v.rows) * v.ntiles_n;
  // (every 8 channels need an owner workgroup: tiny maps have fewer tiles than that; slots per owner thread: MAXS)
  return d->Cout % 8 == 0 && d->Nstore == d->Cout && nwg <= device_cus() && nwg >= (d->Cout + 7) / 8 && (M + 127) / 128 <= 384;
}

extern "C" long simt_conv_fbn_words(const simt_conv_desc* d) {
  if (!simt_conv_fbn_ok(d)) return 0;
  const Conv2Variant v = pick_variant(d);
  const long tiles = (d->B * d->Ho * d->Wo + v.rows - 1) / v.rows;
  return SIMT_FBN_BAR_WORDS + 2l * d->Cout + tiles * 3 * d->Cout;
}

// Does the launch for d accept simt_conv_desc.in_scale / in_shift / in_out (BatchNorm + ReLU of the input applied in the operand path)?
extern "C" int simt_conv_inbn_ok(const simt_conv_desc* d) {
  if (!d || d->dtype_in != SIMT_BF16 || d->dtype_out != SIMT_BF16 || d->tile_n != 256 || d->fbn) return 0;
  const Conv2Variant v = pick_variant(d);
  return v.rowsk && simt_conv_rows_inbn_ok(d) ? 1 : 0;
}

extern "C" int simt_conv_mtiles(const simt_conv_desc* d) {
  int bn, tm, nst;
  const int gen = simt_conv_variant(d, &bn, &tm, &nst);
  if (gen != 2 && gen != 4 && gen != 5) return 0;
  const Conv2Variant v = pick_variant(d);
  const int M = d->B * d->Ho * d->Wo;
  return (M + v.rows - 1) / v.rows;
}

// Kernel arguments of the bf16 v2 family for d (every generation: conv_igemm2 / rows / stream); *vout = the variant the launch takes.
static int conv2_fill_args(const simt_conv_desc* d, Conv2KArgs& k, Conv2Variant* vout) {
  k.wf = (d->w_frag && simt_conv_wants_frag(d)) ? (const char*)d->w_frag : nullptr; k.nt16 = d->Npad / 16;
  k.x = (const char*)d->x; k.w = (const char*)d->w; k.y = (bf16_t*)d->y; k.bias = d->bias; k.res = (const bf16_t*)d->res;
  k.stats = d->stats; k.zero = (const char*)simt_zero_page();
  k.mask = (const bf16_t*)d->mask; k.ldm = d->ldm; k.res_bits = d->res_bits;
  k.bnr_mode = d->bnr_mode; k.bnr_ld = d->bnr_ld; k.bnr_y = (const bf16_t*)d->bnr_y; k.bnr_mean = d->bnr_mean; k.bnr_rstd = d->bnr_rstd;
  k.bnr_scale = d->bnr_scale; k.bnr_shift = d->bnr_shift; k.bnr_bits = d->bnr_bits; k.bnr_part = d->bnr_part;
  if (k.bnr_mode) {
    SIMT_CHECK(d->dtype_out == SIMT_BF16 && !d->stats && !d->relu && !d->mask && d->bnr_y && d->bnr_mean && d->bnr_rstd && d->bnr_part);
    SIMT_CHECK(d->bnr_ld % 8 == 0 && (d->bnr_mode == 2 ? (d->bnr_scale && d->bnr_shift) : (d->bnr_mode == 3 && d->bnr_bits)));
  }
  k.fbn_mode = 0;
  if (d->fbn) {
    const simt_fbn_desc* f = d->fbn;
    SIMT_CHECK(simt_conv_fbn_ok(d) && f->out && f->work && f->ldo % 8 == 0 && !d->bias && !d->res && !d->relu && !d->mask);
    SIMT_CHECK(((long)d->B * d->Ho * d->Wo + 127) / 128 <= 384);       // statistics slots per owner thread: conv2_epilogue.h MAXS
    if (f->mode == 1) SIMT_CHECK(d->stats && !d->bnr_mode && f->mean && f->rstd && f->scale && f->shift && (!f->running_mean || f->running_var));
    else SIMT_CHECK(f->mode == 2 && d->bnr_mode == 2 && f->coef);
    k.fbn_mode = f->mode; k.fbn_ldo = f->ldo; k.fbn_out = (bf16_t*)f->out; k.fbn_bar = (unsigned long long*)f->work;
    k.fbn_err = f->err ? (unsigned long long*)f->err : k.fbn_bar + SIMT_FBN_ERR_WORD;
    k.fbn_cgran = k.fbn_bar + SIMT_FBN_BAR_WORDS; k.fbn_slots = k.fbn_cgran + 2l * d->Cout;
    k.fbn_gamma = f->gamma; k.fbn_beta = f->beta; k.fbn_rmean = f->running_mean; k.fbn_rvar = f->running_var;
    k.fbn_momentum = f->momentum; k.fbn_eps = f->eps;
    k.fbn_mean = f->mean; k.fbn_rstd = f->rstd; k.fbn_scale = f->scale; k.fbn_shift = f->shift; k.fbn_coef = f->coef;
    k.fbn_dgamma = f->dgamma; k.fbn_dbeta = f->dbeta;
  }
  k.in_scale = d->in_scale; k.in_shift = d->in_shift; k.in_out = (bf16_t*)d->in_out;
  if (d->in_scale || d->in_shift || d->in_out)       // the operand-path BatchNorm exists in the row-streaming kernel's statistics flavour only
    SIMT_CHECK(d->in_scale && d->in_shift && d->in_out && simt_conv_inbn_ok(d));
  k.out_f32 = d->dtype_out == SIMT_F32;
  if (k.out_f32) SIMT_CHECK(!d->bias && !d->res && !d->relu && !d->stats && d->Nstore % 4 == 0 && d->ldy % 4 == 0);
  k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout; k.Nstore = d->Nstore; k.ldy = d->ldy; k.ldr = d->ldr;
  k.stride = d->stride; k.ntaps = d->ntaps; k.relu = d->relu; k.M = d->B * d->Ho * d->Wo;
  k.rcp_hw = 1.0f / (float)(d->Ho * d->Wo); k.rcp_wo = 1.0f / (float)d->Wo;
  SIMT_CHECK((long)d->B * d->Ho * d->Wo < (1l << 24));     // fast_divmod range
  k.kc_per_tap = d->Cin * 2 / 128;
  k.pix_bytes = d->Cin * 2;
  k.wrow_bytes = d->ntaps * d->Cin * 2;
  const Conv2Variant v = pick_variant(d);
  const int tile_n = v.tile_n, tm = v.tm;
  const bool short_k = v.nst == 2;
  k.ntiles_n = v.ntiles_n;
  k.rows = v.rows;
  k.ntiles_m = (k.M + k.rows - 1) / k.rows;
  k.nblk128 = (k.M + 127) / 128;
  SIMT_CHECK((long)d->B * d->H * d->W * d->Cin * 2 < (1l << 32));      // 32-bit byte offsets
  SIMT_CHECK((long)d->Npad * k.wrow_bytes < (1l << 32));
  for (int i = 0; i < SIMT_MAX_TAPS; ++i) {
    k.dy[i] = d->dy[i];
    k.dx[i] = d->dx[i];
    k.toff[i] = (d->dy[i] * d->W + d->dx[i]) * k.pix_bytes;
  }
  *vout = v;
  (void)tile_n; (void)tm; (void)short_k;
  return SIMT_OK;
}

// Called by simt_conv_fprop (conv_igemm.hip) for bf16 -> bf16 problems with tile_n in {64, 128, 256}.
int simt_conv_fprop_bf16_v2(const simt_conv_desc* d, simt_stream_t stream) {
  Conv2KArgs k;
  Conv2Variant v;
  const int rc = conv2_fill_args(d, k, &v);
  if (rc != SIMT_OK) return rc;
  const int tile_n = v.tile_n, tm = v.tm;
  const bool short_k = v.nst == 2;
  hipStream_t st = (hipStream_t)stream;
  if (v.rowsk) return simt_conv_rows_launch(k, d->Npad, st);
  if (v.stream) return simt_conv_stream_launch(k, d->Npad, st);
  if (short_k) return tm == 5 ? launch_conv2<128, 5, 2>(k, st) : launch_conv2<128, 4, 2>(k, st);
  if (tile_n == 256) return tm == 5 ? launch_conv2<256, 5>(k, st) : launch_conv2<256, 4>(k, st);
  if (tile_n == 128) return tm == 5 ? launch_conv2<128, 5>(k, st) : launch_conv2<128, 4>(k, st);
  return launch_conv2<64, 2>(k, st);
}

// ---- two convs of identical geometry in one launch (simt_conv_fprop_pair) ----------------------------------------------------------------
int simt_conv_rows_pair_launch(Conv2KArgs k0, Conv2KArgs k1, int npad, hipStream_t st, bool* taken);      // conv1x1_rows.hip
template <int BN, int TM, int E0, int E1>
static int launch_pair(const Conv2KArgs& k0, const Conv2KArgs& k1, hipStream_t st) {
  constexpr int WM = (BN == 64) ? 4 : 2;
  constexpr int BM = WM * TM * 16;
  const size_t ring = 3 * (size_t)(BM * 128 + BN * 128);
  const size_t epi = (size_t)BM * (BN * 2 + 8) + (size_t)(512 / (BN / 8)) * 2 * BN * 4;
  const size_t lds = ring > epi ? ring : epi;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, lds))
    (void)hipFuncSetAttribute((const void*)conv_igemm2_pair_kernel<BN, TM, 3, E0, E1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((conv_igemm2_pair_kernel<BN, TM, 3, E0, E1>), dim3(((k0.ntiles_m * k0.ntiles_n + 7) & ~7) + k1.ntiles_m * k1.ntiles_n), dim3(512), lds, st, k0, k1);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
// the flavour pairs of the two forwards: (trainable, frozen) = (statistics, bias + ReLU) for conv1 / conv2 / the stem, (statistics, generic) for
// the downsample convs (bias without ReLU), (generic, generic) for the fp32 tap-expanded heads
template <int BN, int TM>
static bool launch_pair_flavours(const Conv2KArgs& k0, const Conv2KArgs& k1, hipStream_t st, int* rc) {
  const int e0 = conv2_flavour(k0), e1 = conv2_flavour(k1);
  if (e0 == 1 && e1 == 3) { *rc = launch_pair<BN, TM, 1, 3>(k0, k1, st); return true; }
  if (e0 == 1 && e1 == 0) { *rc = launch_pair<BN, TM, 1, 0>(k0, k1, st); return true; }
  if (e0 == 0 && e1 == 0) { *rc = launch_pair<BN, TM, 0, 0>(k0, k1, st); return true; }
  return false;
}

static int pair_enabled() {
  static const int off = getenv("SIMT_NO_PAIR") ? atoi(getenv("SIMT_NO_PAIR")) : 0;      // A/B switch: two launches
  return !off;
}

// 1 if simt_conv_fprop_pair(d0, d1) is ONE launch (same geometry and tile variant, a supported pair of epilogue flavours), 0 if it runs the two
// convs one after the other (always correct)
extern "C" int simt_conv_pair_fused(const simt_conv_desc* d0, const simt_conv_desc* d1) {
  if (!d0 || !d1 || !pair_enabled() || d0->fbn || d1->fbn) return 0;
  int b0, t0, n0, b1, t1, n1;
  const int g0 = simt_conv_variant(d0, &b0, &t0, &n0), g1 = simt_conv_variant(d1, &b1, &t1, &n1);
  if (g0 != g1 || b0 != b1 || t0 != t1 || n0 != n1 || (g0 != 2 && g0 != 5)) return 0;
  if (d0->B != d1->B || d0->H != d1->H || d0->W != d1->W || d0->Cin != d1->Cin || d0->Ho != d1->Ho || d0->Wo != d1->Wo || d0->stride != d1->stride ||
      d0->ntaps != d1->ntaps || d0->Npad != d1->Npad || d0->Cout != d1->Cout || d0->Nstore != d1->Nstore || d0->dtype_out != d1->dtype_out) return 0;
  for (int i = 0; i < d0->ntaps; ++i) if (d0->dy[i] != d1->dy[i] || d0->dx[i] != d1->dx[i]) return 0;
  if (d0->in_scale || d1->in_scale) return 0;       // (the operand-path BatchNorm flavour has no pair instantiation)
  Conv2KArgs k0, k1;
  k0.out_f32 = d0->dtype_out == SIMT_F32; k0.res = (const bf16_t*)d0->res; k0.mask = (const bf16_t*)d0->mask; k0.Nstore = d0->Nstore; k0.Cout = d0->Cout;
  k0.stats = d0->stats; k0.bias = d0->bias; k0.relu = d0->relu; k0.bnr_mode = d0->bnr_mode; k0.res_bits = d0->res_bits;
  k1.out_f32 = d1->dtype_out == SIMT_F32; k1.res = (const bf16_t*)d1->res; k1.mask = (const bf16_t*)d1->mask; k1.Nstore = d1->Nstore; k1.Cout = d1->Cout;
  k1.stats = d1->stats; k1.bias = d1->bias; k1.relu = d1->relu; k1.bnr_mode = d1->bnr_mode; k1.res_bits = d1->res_bits;
  if (g0 == 2) {
    if (n0 != 3) return 0;
    const int e0 = conv2_flavour(k0), e1 = conv2_flavour(k1);
    return (e0 == 1 && e1 == 3) || (e0 == 1 && e1 == 0) || (e0 == 0 && e1 == 0);
  }
  // rows kernel: (statistics, bias + residual + ReLU) on the geometries both flavours share (Cin 256 / 512)
  const bool f_stats = d0->stats && !d0->bias && !d0->relu && !d0->res && !d0->bnr_mode && d0->Cout % 4 == 0;
  const bool f_brr = d1->bias && d1->relu && d1->res && !d1->res_bits && !d1->bnr_mode && !d1->stats;
  return f_stats && f_brr && (d0->Cin == 256 || d0->Cin == 512);
}

extern "C" int simt_conv_fprop_pair(const simt_conv_desc* d0, const simt_conv_desc* d1, simt_stream_t stream) {
  SIMT_CHECK(d0 && d1);
  if (!simt_conv_pair_fused(d0, d1)) {
    const int rc = simt_conv_fprop(d0, stream);
    return rc != SIMT_OK ? rc : simt_conv_fprop(d1, stream);
  }
  Conv2KArgs k0, k1;
  Conv2Variant v0, v1;
  int rc = conv2_fill_args(d0, k0, &v0);
  if (rc != SIMT_OK) return rc;
  rc = conv2_fill_args(d1, k1, &v1);
  if (rc != SIMT_OK) return rc;
  SIMT_CHECK(v0.tile_n == v1.tile_n && v0.tm == v1.tm && v0.nst == v1.nst && v0.rows == v1.rows && v0.ntiles_n == v1.ntiles_n && v0.rowsk == v1.rowsk && !v0.stream);
  SIMT_CHECK(k0.ntiles_m == k1.ntiles_m);
  hipStream_t st = (hipStream_t)stream;
  if (v0.rowsk) {
    bool taken = false;
    rc = simt_conv_rows_pair_launch(k0, k1, d0->Npad, st, &taken);
    SIMT_CHECK(taken);
    return rc;
  }
  bool ok = false;
  if (v0.tile_n == 256) ok = v0.tm == 5 ? launch_pair_flavours<256, 5>(k0, k1, st, &rc) : launch_pair_flavours<256, 4>(k0, k1, st, &rc);
  else if (v0.tile_n == 128) ok = v0.tm == 5 ? launch_pair_flavours<128, 5>(k0, k1, st, &rc) : launch_pair_flavours<128, 4>(k0, k1, st, &rc);
  else ok = launch_pair_flavours<64, 2>(k0, k1, st, &rc);
  SIMT_CHECK(ok);
  return rc;
}
