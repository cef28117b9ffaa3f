// Streaming 1x1 convolution for the short-reduction / wide-output shapes (Bottleneck conv3 256 -> 1024, 512 -> 2048, 64 -> 256,
// 128 -> 512 and the dgrads of the matching conv1: model/deeplab_multi.py:62,73) -- bf16, gfx950.
//
// Why a second kernel: in-kernel stamps of conv_igemm2_kernel<128,4,2> on 256 -> 1024 at M = 37 636 (profiles/r02_stamps.txt) show a
// workgroup lifetime of ~24 k cycles of which the 4-stage main loop is 37 %: 28 % is start-up (kernel-argument loads, addressing,
// the first stage's HBM latency), 35 % the epilogue, whose 32 KB of stores leave a CU at ~8 B/clk -- the HBM share of one CU when every
// CU stores at once, which is what lock-stepped one-tile workgroups do.  The output stream (77 of the 96 MB) only flows during the
// epilogues: 1.2 TB/s.  This kernel keeps it flowing:
//   * persistent workgroups (one per CU), each walking tiles t, t + 256, ... of 128 pixels x 128 output channels; the grid stride
//     keeps a workgroup on ONE column tile, so its bias / statistics columns are loop invariants;
//   * WAVE SPECIALISATION: 8 compute waves (LDS-DMA loads + MFMA) never store to global memory, 8 store waves never load from it.
//     vmcnt retires IN ORDER, so a wave that stores and loads sees a ring stage only after its older stores are acknowledged by HBM
//     (measured: the same kernel with the stores in the compute waves and exact counted waits ran 75 us against 65 us);
//   * ONE stream of K-stages through a 3-slot global_load_lds ring that runs across tile boundaries (the next tile's first stages are
//     in flight while the current tile finishes: no per-tile start-up latency), counted vmcnt + one raw barrier per stage;
//   * DEFERRED epilogue: a finished tile's accumulators go to a private LDS tile (bf16) and are streamed out by the store waves
//     during the K-stages of the NEXT tile (they join every stage barrier) -- stores, BatchNorm statistics and MFMAs overlap.
// Flavours: plain, + BatchNorm batch statistics (deterministic: wave shuffle over the four row groups of a wave, then the eight store
// waves in fixed order), + bias / ReLU.  Residual / bit-mask / fused BN-backward flavours stay on conv_igemm2_kernel.
#include "conv2_common.h"

// Compile-time timing ablations (scratch builds only; outputs meaningless): 1 = no fragment reads / MFMA, 2 = no global stores,
// 4 = no LDS-DMA loads, 8 = store waves idle (barriers only), 16 = no statistics, 32 = no accumulator -> LDS writes, 64 = no LDS reads in row passes
#ifndef SIMT_STREAM_ABL
#define SIMT_STREAM_ABL 0
#endif

namespace {

constexpr int NC = 512, NS = 512, NT = NC + NS;      // 8 compute waves (LDS-DMA loads + MFMA) + 8 store waves (deferred epilogue)
constexpr int BN = 128, BM = 128, WN = 4, TM = 4, TN = 2, NST = 3;
constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
constexpr int A_IT = BM * 8 / NC, B_IT = BN * 8 / NC, P = A_IT + B_IT;      // LDS-DMA pieces per compute thread per stage
constexpr int CP = BN * 2 + 8;                                             // pitch of the bf16 output tile in LDS
constexpr int VPR = BN / 8, RPP = NS / VPR, NIT = BM / RPP;                // 16 vectors per row, 32 rows per pass, 4 passes per tile
constexpr int NSW = NS / 64;                                               // store waves
constexpr int RING = NST * STAGE, SC_BYTES = BM * CP, SR_BYTES = NSW * 2 * BN * 4;
constexpr int LDS_BYTES = RING + SC_BYTES + SR_BYTES;

// v[l] + v[l^16] + v[l^32] + v[l^48] in the order ((l, l^16), (l^32, l^48)) for the lanes of the first row
__device__ __forceinline__ float quad_rows_sum(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(r.x) + __uint_as_float(r.y);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__global__ __launch_bounds__(NT, 4) void conv1x1_stream_kernel(Conv2KArgs a, int G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sC = smem + RING;
  float* sR = (float*)(smem + RING + SC_BYTES);                 // [store waves][2][BN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.ntiles_m * a.ntiles_n;
  const int my_n = (nwg - (int)blockIdx.x + G - 1) / G;          // tiles of this workgroup: blockIdx.x + i * G
  const int nk = a.kc_per_tap;                                   // K-stages per tile (ntaps == 1)
  const int S_total = my_n * nk;
  const int tile0 = xcd_remap(blockIdx.x, nwg);
  const int tstep = G >> 3;                                      // tile-id step per round: xcd_remap(b + i*G) = tile0 + i * G/8
  const int nt = tile0 % a.ntiles_n;                             // the same for every tile of this workgroup (host checks ntiles_n | G/8)
  const int n0 = nt * BN;
  STAMP(1);

  if (wave < NC / 64) {
    // =============================== compute waves: stage stream + MFMA; they never store to global memory ===============================
    const int wm = wave / WN, wn = wave % WN;
    const int hw = a.Ho * a.Wo;
    const int a_cg = (tid & 7) ^ (((tid >> 3) >> 1) & 7);
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_off[i] = (unsigned)(n0 + i * (NC / 8) + (tid >> 3)) * (unsigned)a.wrow_bytes + (unsigned)(a_cg * 16);
    const char* zsrc = a.zero + a_cg * 16;
    // per-tile pixel addressing of the ISSUE cursor (tile ii, stage ikc): runs up to two stages ahead of the compute cursor
    unsigned ia_off[A_IT];
    bool ia_ok[A_IT];
    auto tile_addr = [&](int i) {
      const int mt = (tile0 + i * tstep) / a.ntiles_n;
#pragma unroll
      for (int q = 0; q < A_IT; ++q) {
        const int m = mt * BM + q * (NC / 8) + (tid >> 3);
        ia_ok[q] = m < a.M;
        ia_off[q] = 0u;
        if (ia_ok[q]) {
          int b, r, oy, ox;
          fast_divmod(m, hw, a.rcp_hw, b, r);
          fast_divmod(r, a.Wo, a.rcp_wo, oy, ox);
          ia_off[q] = (unsigned)(((b * a.H + oy * a.stride) * a.W + ox * a.stride)) * (unsigned)a.pix_bytes + (unsigned)(a_cg * 16) + (unsigned)a.toff[0];
        }
      }
    };
    int ii = 0, ikc = 0;
    tile_addr(0);
    auto issue = [&](int slot) {
      char* sbase = smem + slot * STAGE;
#if !(SIMT_STREAM_ABL & 4)
#pragma unroll
      for (int q = 0; q < A_IT; ++q) {
        const char* src = ia_ok[q] ? a.x + (unsigned)(ia_off[q] + (unsigned)(ikc * 128)) : zsrc;
        __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sbase + (q * NC + wave * 64) * 16), 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q)
        __builtin_amdgcn_global_load_lds(GPTR(a.w + (b_off[q] + (unsigned)(ikc * 128))), LPTR(sbase + A_BYTES + (q * NC + wave * 64) * 16), 16, 0, 0);
#endif
      if (++ikc == nk) {
        ikc = 0;
        if (++ii < my_n) tile_addr(ii);
      }
    };
    // fragments / MFMA (same operand roles as conv_igemm2: weights = A operand, pixels = B operand)
    const int sw = (lane >> 1) & 7, kq = lane >> 4;
    const int xbase = (wm * TM * 16) * 128 + (lane & 15) * 128;
    const int wbase = A_BYTES + (wn * TN * 16) * 128 + (lane & 15) * 128;
    f32x4 acc[TN][TM];
    bf16x8 xf[2][TM], wf[2][TN];
    auto load_frags = [&](int slot) {
      const char* px = smem + slot * STAGE + xbase;
      const char* pw = smem + slot * STAGE + wbase;
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
          for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][j], xf[s][i], acc[j][i], 0, 0, 0);
    };
    if (S_total > 0) issue(0);
    if (S_total > 1) issue(1);
    int g = 0;
#ifdef SIMT_ABLATION
    unsigned long long t_wait = 0, t_bar = 0, t_work = 0, t_hand = 0;
#endif
    for (int ci = 0; ci < my_n; ++ci) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int kc = 0; kc < nk; ++kc, ++g) {
#ifdef SIMT_ABLATION
        const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
        if (g + 1 < S_total) wait_vmcnt<P>(); else wait_vmcnt<0>();        // stage g landed; stage g+1 may be in flight
#ifdef SIMT_ABLATION
        const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef SIMT_ABLATION
        const unsigned long long tw2 = __builtin_amdgcn_s_memtime();
        t_wait += tw1 - tw0; t_bar += tw2 - tw1;
#endif
        const int slot = g % NST;
#if !(SIMT_STREAM_ABL & 1)
        load_frags(slot);
#endif
        if (g + 2 < S_total) issue((g + 2) % NST);
#if !(SIMT_STREAM_ABL & 1)
        mma();
#endif
#ifdef SIMT_ABLATION
        asm volatile("s_nop 0" ::: "memory");
        t_work += __builtin_amdgcn_s_memtime() - tw2;
#endif
      }
#ifdef SIMT_ABLATION
      const unsigned long long th0 = __builtin_amdgcn_s_memtime();
#endif
      // hand-over: barrier A (the store waves are done with the previous tile in sC), accumulators -> sC, barrier B (sC visible)
      __builtin_amdgcn_s_barrier();
#if !(SIMT_STREAM_ABL & 32)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int r = wm * TM * 16 + i * 16 + (lane & 15);
          const int c = wn * TN * 16 + j * 16 + (lane >> 4) * 4;
          uint2 pk;
          pk.x = pack_bf16x2(acc[j][i][0], acc[j][i][1]);
          pk.y = pack_bf16x2(acc[j][i][2], acc[j][i][3]);
          // written by inline asm: for a C++ LDS store the compiler first drains vmcnt to 0 (it assumes the store may alias an LDS-DMA
          // in flight), i.e. it would wait here for the next tile's first two stages
          const unsigned addr = (unsigned)(size_t)LPTR(sC + r * CP + c * 2);
          asm volatile("ds_write_b64 %0, %1" :: "v"(addr), "v"(pk) : "memory");
        }
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#ifdef SIMT_ABLATION
      t_hand += __builtin_amdgcn_s_memtime() - th0;
#endif
    }
    STAMP(3);
#ifdef SIMT_ABLATION
    if (threadIdx.x == 0 && blockIdx.x < 8192) { g_stamps[blockIdx.x * 8 + 4] = t_wait; g_stamps[blockIdx.x * 8 + 6] = t_bar; g_stamps[blockIdx.x * 8 + 7] = (unsigned long long)my_n; g_stamps[blockIdx.x * 8 + 0] = t_work; g_stamps[blockIdx.x * 8 + 2] = t_hand; }
#endif
    // drain: the store waves finish the last tile behind three more barriers (A, B, and C for its statistics in sR)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    return;
  }

  // ================================= store waves: the deferred epilogue of the tile that sits in sC =================================
  const int st = tid - NC;                                           // 0 .. NS-1
  const int swv = wave - NC / 64;                                    // 0 .. NSW-1
  const int vcol = (st % VPR) * 8, rg = st / VPR, n = n0 + vcol;
  const bool ncol_ok = n < a.Nstore;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (a.bias && ncol_ok && (n + e) < a.Cout) ? a.bias[n + e] : 0.f;
  // Pin the bias loads' wait HERE: left to the compiler, its s_waitcnt vmcnt(0) for them lands at the head of the per-tile loop, where
  // it also waits for every output store of the previous tile to be acknowledged by HBM (2-3 us per tile: 22 of the 57 us of a launch).
#pragma unroll
  for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(bias8[e]));
  const bool plain = !a.bias && !a.relu;
#if SIMT_STREAM_ABL & 16
  a.stats = nullptr;
#endif
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  int prev_m0 = 0, prev_mt = 0, rows_done = NIT;
  bool has_prev = false;
  // One 16-byte piece of one row of the previous tile per thread: LDS -> (statistics, bias, ReLU) -> HBM.  The LDS read of the NEXT pass
  // is issued before the current pass is processed (sC only changes at hand-overs), so a pass does not start with an LDS round trip.
  uint2 nlo = make_uint2(0u, 0u), nhi = make_uint2(0u, 0u);
  auto fetch_row = [&](int pass) {
    const int r = rg + pass * RPP;
    if (!(SIMT_STREAM_ABL & 64) && pass < NIT && ncol_ok && prev_m0 + r < a.M) {
      nlo = *(const uint2*)(sC + r * CP + vcol * 2);
      nhi = *(const uint2*)(sC + r * CP + vcol * 2 + 8);
    }
  };
  auto row_pass = [&]() {
    const int r = rg + rows_done * RPP;
    const int m = prev_m0 + r;
    uint4 o = make_uint4(nlo.x, nlo.y, nhi.x, nhi.y);
    ++rows_done;
    fetch_row(rows_done);
    if (ncol_ok && m < a.M) {
      if (a.stats || !plain) {
        float v[8];
        v[0] = __uint_as_float(o.x << 16); v[1] = __uint_as_float(o.x & 0xffff0000u);
        v[2] = __uint_as_float(o.y << 16); v[3] = __uint_as_float(o.y & 0xffff0000u);
        v[4] = __uint_as_float(o.z << 16); v[5] = __uint_as_float(o.z & 0xffff0000u);
        v[6] = __uint_as_float(o.w << 16); v[7] = __uint_as_float(o.w & 0xffff0000u);
        if (!(SIMT_STREAM_ABL & 512) && a.stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
        }
        if (!plain) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { v[e] += bias8[e]; if (a.relu) v[e] = v[e] > 0.f ? v[e] : 0.f; }
          o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
          o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        }
      }
#if !(SIMT_STREAM_ABL & 2)
      st_out16(a.y + (long)m * a.ldy + n, o);
#else
      if (o.x == 0x12345678u && o.y == 0x9abcdef0u) *(uint4*)(a.y + (long)m * a.ldy + n) = o;
#endif
    }
  };
  // statistics of the previous tile: the 4 row groups of a wave by lane swaps, then the store waves in fixed order through sR
  auto stats_to_lds = [&]() {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      // lanes l, l^16, l^32, l^48 hold the four row groups of a column: v_permlane16_swap / v_permlane32_swap (VALU; a
      // ds_bpermute shuffle here cost 13 us per launch: 256 LDS-crossbar instructions per tile inside the hand-over)
      const float t1 = quad_rows_sum(s1[e]), t2 = quad_rows_sum(s2[e]);
      if (!(SIMT_STREAM_ABL & 256) && lane < 16) { sR[(swv * 2 + 0) * BN + vcol + e] = t1; sR[(swv * 2 + 1) * BN + vcol + e] = t2; }
      s1[e] = 0.f; s2[e] = 0.f;
    }
  };
  auto stats_to_hbm = [&](int mt) {
    const int nn = n0 + st;
    if (!(SIMT_STREAM_ABL & 128) && st < BN && nn < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int q = 0; q < NSW; ++q) { t1 += sR[(q * 2 + 0) * BN + st]; t2 += sR[(q * 2 + 1) * BN + st]; }
      a.stats[((long)mt * 2 + 0) * a.Cout + nn] = t1;
      a.stats[((long)mt * 2 + 1) * a.Cout + nn] = t2;
    }
  };
  const int passes_per_stage = (NIT + nk - 1) / nk;
  bool stats_pending = false;                          // sR holds the statistics of tile prev_stats_mt, visible after the next barrier
  int prev_stats_mt = 0;
  for (int ci = 0; ci <= my_n; ++ci) {                 // iteration my_n is the drain of the last tile
    if (ci < my_n) {
      for (int kc = 0; kc < nk; ++kc) {
#if !(SIMT_STREAM_ABL & 8)
        if (has_prev)
          for (int q = 0; q < passes_per_stage && rows_done < NIT; ++q) row_pass();
#endif
        if (stats_pending) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // the compute waves' stage barrier
        if (stats_pending) { stats_to_hbm(prev_stats_mt); stats_pending = false; }
      }
    }
#if !(SIMT_STREAM_ABL & 8)
    if (has_prev)
      while (rows_done < NIT) row_pass();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // barrier A: sC may be overwritten
    if (stats_pending) { stats_to_hbm(prev_stats_mt); stats_pending = false; }      // drain iteration only (no stage barrier came by)
    __builtin_amdgcn_s_barrier();                      // barrier B: sC holds tile ci
    // the statistics of tile ci - 1 leave the registers AFTER the hand-over (the compute waves are not held up by it); sR is read
    // behind the next barrier (the first stage barrier of tile ci + 1, or the drain barrier)
    if (has_prev && a.stats) { stats_to_lds(); stats_pending = true; prev_stats_mt = prev_mt; }
    if (ci < my_n) {
      const int cur_mt = (tile0 + ci * tstep) / a.ntiles_n;
      has_prev = true; prev_m0 = cur_mt * BM; prev_mt = cur_mt; rows_done = 0;
      fetch_row(0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                        // drain barrier C
  if (stats_pending) stats_to_hbm(prev_stats_mt);
#ifdef SIMT_ABLATION
  if (threadIdx.x == NC && blockIdx.x < 8192) { g_stamps[blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memtime(); }
#endif
}

}  // namespace

#ifdef SIMT_ABLATION
extern "C" int simt_debug_stamps_stream(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// Shapes this kernel takes over from conv_igemm2_kernel<128, *, 2> (called by simt_conv_fprop_bf16_v2 with the filled arguments).
bool simt_conv_stream_eligible(const simt_conv_desc* d) {
  if (d->dtype_in != SIMT_BF16 || d->dtype_out != SIMT_BF16 || d->ntaps != 1) return false;
  if (d->res || d->mask || d->bnr_mode || d->res_bits) return false;
  if (d->Cin % 64 != 0 || d->Npad % BN != 0) return false;
  const int ntn = d->Npad / BN;
  return ntn == 1 || ntn == 2 || ntn == 4 || ntn == 8 || ntn == 16 || ntn == 32;      // ntiles_n | 256 / 8
}

int simt_conv_stream_launch(Conv2KArgs k, int npad, hipStream_t st) {
  k.rows = BM;
  k.ntiles_n = npad / BN;
  k.ntiles_m = (k.M + BM - 1) / BM;
  const int nwg = k.ntiles_m * k.ntiles_n;
  const int G = nwg < 256 ? nwg : 256;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, LDS_BYTES))
    (void)hipFuncSetAttribute((const void*)conv1x1_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipLaunchKernelGGL(conv1x1_stream_kernel, dim3(G), dim3(NT), LDS_BYTES, st, k, G);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
