// Row-streaming 1x1 convolution with the weights held in registers -- bf16, gfx950.
// Shapes: the short-reduction / wide-output 1x1 convs of the Bottlenecks (model/deeplab_multi.py:62,73: conv3 256 -> 1024,
// 128 -> 512, 64 -> 256 forward, and the dgrads of the matching conv1), dense NHWC input (stride 1), Cin in {64, 128, 256}.
//
// Why (measurements: DESIGN.md section 5, profiles/microbench/ldsbench.hip): these GEMMs have K <= 256, so a 128x128 output tile needs 128 KB of
// operands for 32 KB of output.  A CU fills LDS from L2 at ~75 GB/s, which is what bounded conv1x1_stream_kernel (302 MB through
// L1 per launch of 256 -> 1024), half of it the SAME weight tile re-read for every pixel tile; and its three barrier hand-shakes per
// tile coupled the store waves' HBM write latency to the MFMA waves.  Here:
//   * a workgroup owns 256 output channels; each of its 8 compute waves keeps its 32 x Cin weight slice as MFMA A-operand fragments
//     in REGISTERS for the whole launch (Cin/32 x 2 x 4 VGPRs) -- weights never touch LDS;
//   * pixels stream through a D-slot global_load_lds ring in stages of RS whole rows (RS * Cin * 2 bytes, up to D-1 stages in
//     flight, counted vmcnt); a stage is a COMPLETE reduction, so its RS x 256 outputs leave the accumulators at once;
//   * the accumulators of stage g-1 are rounded to bf16 and written to one of two LDS slabs INSIDE the MFMA loop of stage g (two VALU
//     behind every MFMA; for the training forward the BatchNorm batch statistics are taken there too, per channel, one DPP row reduction
//     per 128-row tile); 4 (2) store waves stream the finished slabs out (bias, residual, bit-masked residual, ReLU, fused BatchNorm-
//     backward reduce) while the compute waves work on: ONE barrier per stage, no hand-over section;
//   * persistent workgroups on 128-row tiles (the statistics granule), XCD-aware: the workgroups of one XCD share pixel rows.
// L2 -> LDS traffic per launch of 256 -> 1024: 4 x 19 MB instead of 302 MB.
#include "conv2_common.h"
#include <type_traits>

// Compile-time timing ablations (scratch builds only; outputs meaningless): 1 = no fragment reads / MFMA, 2 = no global stores,
// 4 = no LDS-DMA loads, 8 = store waves idle (barriers only), 64 = no epilogue in the compute waves, 128 = no fragment reads, 256 = no statistics,
// 512 = the store waves' global operands (residual, BatchNorm-backward y) from an L2-resident window
#ifndef SIMT_ROWS_ABL
#define SIMT_ROWS_ABL 0
#endif

#ifndef SIMT_ROWS_PF1
#define SIMT_ROWS_PF1 6
#endif

namespace {


// NCW compute waves (each 32 output channels: BN = 32 * NCW per workgroup) + NSW store waves.  <8, 4>: one 768-thread workgroup per CU;
// <4, 2>: two 384-thread workgroups per CU, which drift apart so that one's store phase overlaps the other's MFMA phase.
// TN: 16-channel blocks per compute wave (weights per wave: TN x Cin/32 x 4 registers -- 2 up to Cin = 256, 1 for Cin = 512).
template <int KS, int TM, int D, int NSW, int NCW, int TN> struct Geo {
  static constexpr int NC = NCW * 64, BN = NCW * TN * 16;
  static constexpr int CP = BN * 2 + 16;             // slab pitch (bytes): 4 * odd dwords, conflict-free for the accumulator writes
  static constexpr int VPR = BN / 8;                 // 16-byte pieces per output row
  static constexpr int NS = NSW * 64, NT = NC + NS;
  static constexpr int RGS = NS / VPR;               // row groups of the store waves
  static constexpr int SR_BYTES = NSW * 2 * BN * 4;
  static constexpr int RS = TM * 16;                 // pixel rows per stage
  static constexpr int CIN = KS * 32;
  static constexpr int KC = CIN / 64;                // 64-deep (128-byte) sub-tiles per stage
  static constexpr int SB = RS * CIN * 2;            // stage bytes
  static constexpr int PT = SB / 16 / NC;            // LDS-DMA pieces per compute thread per stage
  static constexpr int SPT = 128 / RS;               // stages per 128-row tile
  static constexpr int SLAB = RS * CP;
  static constexpr int PASSES = RS / RGS;            // rows per store thread per slab
  static constexpr int LDS = D * SB + 2 * SLAB + SR_BYTES;
  static_assert(CIN % 64 == 0 && SB % (16 * NC) == 0 && PT >= 1 && 128 % RS == 0 && LDS <= 160 * 1024 && NS % BN == 0 && RS % RGS == 0, "geometry");
};

// sum over the row groups of a wave: lanes l, l ^ 32 (two groups of 32 lanes) or l, l ^ 16, l ^ 32, l ^ 48 (four groups of 16)
template <int GROUPS> __device__ __forceinline__ float row_groups_sum(float v) {
  typedef __attribute__((ext_vector_type(2))) unsigned u2;
  if (GROUPS == 4) {
    const u2 q = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(q.x) + __uint_as_float(q.y);
  }
  const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

// sum over the 16 lanes of a DPP row by rotations (8, 4, 2, 1): every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xf, 0xf, false));
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false));
  return v;
}

__device__ __forceinline__ void unpack8(const uint4& q, float* v) {
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
  v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
  v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
}

struct Aux { uint4 res, by; unsigned rbits, ybits; };
// FL_STATS_INBN (round 6): FL_STATS whose INPUT is the raw pre-BatchNorm activation of the previous conv (simt_conv_desc.in_scale / in_shift /
// in_out): the store waves normalise + ReLU every landed stage IN PLACE in its ring slot one stage period before the compute waves multiply it,
// and write the activation out on the way (each of the ntiles_n workgroups that share a pixel row writes 1 / ntiles_n of it) -- the separate
// simt_bn_apply launch between conv2 and conv3 of a Bottleneck (model/deeplab_multi.py:88-92) and its 19-MB re-read disappear.
enum { FL_GEN = 0, FL_GEN_AUX = 1, FL_STATS = 2, FL_BRR = 3, FL_BNR = 4, FL_STATS_INBN = 5 };

template <int KS, int TM, int D, int FL, int NSW, int NCW, int TN>
__device__ __forceinline__ void conv1x1_rows_body(const Conv2KArgs& a, const int G, const int bid) {
  using g = Geo<KS, TM, D, NSW, NCW, TN>;
  constexpr int WCOLS = TN * 16;                               // output channels per compute wave
  constexpr int RGS = g::RGS, NS = g::NS, NC = g::NC, BN = g::BN, CP = g::CP, VPR = g::VPR;
  // Epilogue flavour, fixed at compile time for the three the nets use (the store waves share their SIMDs with the MFMA waves: every
  // instruction they do not execute is MFMA time): FL_STATS = store + BatchNorm batch statistics (conv3, training forward), FL_BRR =
  // bias + residual + ReLU (conv3 of the frozen net), FL_BNR = residual through its bit mask + fused BN-backward reduce (dgrad of conv1);
  // FL_GEN / FL_GEN_AUX read the descriptor's flags at run time.
  constexpr bool AUX = FL == FL_BRR || FL == FL_BNR || FL == FL_GEN_AUX;
  constexpr bool GEN = FL == FL_GEN || FL == FL_GEN_AUX;
  // Store waves without global operands run TWO slabs behind the compute waves: the rows of slab g - 2 are already in registers when
  // barrier g falls, so a stage starts with its stores (the step that blocks, on HBM write back-pressure), then requests slab g - 1 from
  // LDS, then does the statistics of slab g - 2 while both are in flight.
  constexpr bool PIPE = !AUX;
  // FL_STATS: the BatchNorm statistics are taken by the COMPUTE waves from the values they round for the slab (a compute wave owns its 32
  // channels for every pixel: per-lane partial sums over a 128-row tile, one 16-lane DPP reduction per tile, no LDS, no second pass over
  // the tile) -- the store waves are left with a pure LDS -> HBM copy.
#ifndef SIMT_ROWS_CSTAT
#define SIMT_ROWS_CSTAT 1
#endif
  // SIMT_ROWS_CSTAT: where FL_STATS takes the statistics (compile-time A/B; all three pass the parity tests).  2: the store waves, on the MFMA pipe -- a transposed read of the
  // finished slab (ds_read_b64_tr_b16: 8 pixels of one channel per lane) is at once the A and the B operand of v_mfma_f32_16x16x32_bf16:
  // ones x Y gives the column sums, Y^T x Y has the sums of squares on its diagonal (bf16 products are exact in fp32): 2 MFMAs and 2 LDS
  // reads per 16 channels and 32 pixels instead of 3 VALU per element (measured 39.0 us on 256 -> 1024).  1 (default): the compute waves (VALU behind their MFMAs, DPP
  // reduce; 39.7 us).  0: the store waves' VALU (register sums, row groups through sR; 39.7-41 us).  The stage time is set by the barrier-
  // coupled pair of chains (HBM write back-pressure on the store waves, MFMA + epilogue on the compute waves), not by where these sums run.
  constexpr bool INBN = FL == FL_STATS_INBN;
  constexpr bool STATS = FL == FL_STATS || INBN;
  static_assert(!INBN || (D >= 4 && SIMT_ROWS_CSTAT == 1), "the in-place input BatchNorm needs a stage landed one period early and the compute-wave statistics");
  constexpr bool CSTAT = STATS && SIMT_ROWS_CSTAT == 1;
  constexpr bool MSTAT = STATS && SIMT_ROWS_CSTAT == 2;
  const bool has_bias = GEN ? a.bias != nullptr : FL == FL_BRR;
  const bool has_relu = GEN ? a.relu != 0 : FL == FL_BRR;
  const bool has_stats = GEN ? a.stats != nullptr : (STATS && SIMT_ROWS_CSTAT == 0);      // statistics taken by the store waves (FL_STATS: by the compute waves)
  const bool has_res = GEN ? (AUX && a.res != nullptr) : AUX;
  const bool has_rbits = GEN ? (AUX && a.res_bits != nullptr) : FL == FL_BNR;
  const bool has_bnr = GEN ? (AUX && a.bnr_mode != 0) : FL == FL_BNR;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (SIMT_ROWS_ABL & 16) return;
#ifdef SIMT_ROWS_STAGGER
  { const int ph = (blockIdx.x >> 3) & 3; for (int i = 0; i < ph * SIMT_ROWS_STAGGER; ++i) __builtin_amdgcn_s_sleep(1); }
#endif
  char* slab = smem + D * g::SB;
  float* sR = (float*)(smem + D * g::SB + 2 * g::SLAB);        // [store waves][2][BN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwg = a.ntiles_m * a.ntiles_n;
  const int my_n = (nwg - bid + G - 1) / G;                    // 128-row tiles of this workgroup: bid + i * G
  const int tile0 = xcd_remap(bid, nwg);
  const int tstep = G >> 3;                                    // xcd_remap(b + i * G) = tile0 + i * G / 8
  const int nt = tile0 % a.ntiles_n;                           // fixed per workgroup (host: ntiles_n divides G / 8)
  const int n0 = nt * BN;
  const int S_total = my_n * g::SPT;
  const int mt0 = tile0 / a.ntiles_n, mt_step = tstep / a.ntiles_n;   // tile i of this workgroup covers rows (mt0 + i * mt_step) * 128 ...

  if (wave < NCW) {
    // ========================= compute waves: weights in registers, pixel stages by LDS-DMA, MFMA, accumulators -> slab =========================
    bf16x8 wf[TN][KS];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        wf[j][ks] = *(const bf16x8*)(a.w + (size_t)(n0 + wave * WCOLS + j * 16 + (lane & 15)) * (unsigned)a.wrow_bytes + (ks * 32 + (lane >> 4) * 8) * 2);
    // LDS-DMA pieces of this thread: linear LDS position p = q * NC + tid -> sub-tile kc = p / (RS * 8), row = (p / 8) % RS, chunk = p % 8
    unsigned pconst[g::PT], pmax[g::PT], pzero[g::PT];         // byte offset of the piece inside a stage / last valid offset / its place in the zero page
#pragma unroll
    for (int q = 0; q < g::PT; ++q) {
      const int p = q * NC + tid;
      const int row = (p >> 3) % g::RS, kc = p / (g::RS * 8), c = p & 7;
      const unsigned ko = (unsigned)(kc * 128 + ((c ^ ((row >> 1) & 7)) << 4));
      pconst[q] = (unsigned)row * (unsigned)a.pix_bytes + ko;
      pmax[q] = (unsigned)(a.M - 1) * (unsigned)a.pix_bytes + ko;      // rows past the end read the zero page: exact zeros, never stored,
      pzero[q] = ko;                                                   // and they add nothing to the statistics
    }
    int ii = 0, is = 0;                                        // issue cursor: tile index of this workgroup, stage inside the tile
    unsigned ibase = (unsigned)(mt0 * 128) * (unsigned)a.pix_bytes;    // uniform: first row of the stage to issue, in bytes
    const unsigned stage_step = (unsigned)g::RS * (unsigned)a.pix_bytes, tile_step = (unsigned)(mt_step * 128) * (unsigned)a.pix_bytes;
    unsigned tile_base = ibase;
    auto issue = [&](int slot) {
      char* sb = smem + slot * g::SB;
#pragma unroll
      for (int q = 0; q < g::PT; ++q) {
        const unsigned off = ibase + pconst[q];
        const char* src = off <= pmax[q] ? a.x + off : a.zero + pzero[q];
        if (!(SIMT_ROWS_ABL & 4)) __builtin_amdgcn_global_load_lds(GPTR(src), LPTR(sb + (q * NC + wave * 64) * 16), 16, 0, 0);
      }
      ibase += stage_step;
      if (++is == g::SPT) { is = 0; ++ii; tile_base += tile_step; ibase = tile_base; }
    };
    const int sw = (lane >> 1) & 7, kq = lane >> 4;
    const int xrow = (lane & 15) * 128;
    // accumulator -> slab addresses (LDS byte addresses): pixel i * 16 + (lane & 15), channels wave * WCOLS + j * 16 + (lane >> 4) * 4
    const unsigned sl_addr = (unsigned)(size_t)LPTR(slab + (lane & 15) * CP + (wave * WCOLS + (lane >> 4) * 4) * 2);
#pragma unroll
    for (int s = 0; s < D - 1; ++s) if (s < S_total) issue(s);
    int slot_c = 0, slot_i = D - 1;
    float cs1[TN][4], cs2[TN][4];                              // CSTAT: this lane's sums over its pixels (lane & 15, every 16-row block)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) { cs1[j][e] = 0.f; cs2[j][e] = 0.f; }
    int c_is = 0, c_ci = 0;                                    // cursor of the stage whose epilogue runs: stage inside the tile, tile
#ifdef SIMT_ABLATION
    unsigned long long t_wait = 0, t_bar = 0, t_work = 0;
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // The epilogue of stage g - 1 (round to bf16, write the slab, statistics) is DEFERRED into the MFMA loop of stage g, two VALU
    // operations behind every MFMA: a wave issues in order, so its own VALU work only overlaps its MFMAs if it sits between them (measured,
    // profiles/microbench/mixbench2.hip: up to two plain VALU per MFMA are free with two waves per SIMD; v_pk_*_f32 never is -- the library is built
    // without packed fp32).  All waves of a workgroup are phase-locked by the stage barrier, so nothing else would fill the MFMA shadow.
    constexpr int NQ = TN * TM;
    f32x4 prev[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) prev[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto epi_quad = [&](int qd, unsigned sbase) {             // one accumulator quad of the previous stage: 4 channels of one pixel per lane
      const int j = qd / TM, i = qd % TM;
      uint2 pk;
      pk.x = pack_bf16x2(prev[j][i][0], prev[j][i][1]);
      pk.y = pack_bf16x2(prev[j][i][2], prev[j][i][3]);
      // LDS traffic of the compute waves is inline asm with hand-counted lgkmcnt: before a C++ LDS store the compiler drains vmcnt to 0 (it
      // must assume the store aliases an LDS-DMA in flight -- also with separate static LDS arrays), and around C++ LDS loads it waits
      // lgkmcnt(0) right after requesting the next fragments
      asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(sbase), "v"(pk), "n"(i * 16 * CP + j * 32));
      if (CSTAT && !(SIMT_ROWS_ABL & 256)) {                   // statistics of the values as stored (bf16); rows past the end are exact zeros
        const float v0 = __uint_as_float(pk.x << 16), v1 = __uint_as_float(pk.x & 0xffff0000u);
        const float v2 = __uint_as_float(pk.y << 16), v3 = __uint_as_float(pk.y & 0xffff0000u);
        cs1[j][0] += v0; cs1[j][1] += v1; cs1[j][2] += v2; cs1[j][3] += v3;
        cs2[j][0] = fmaf(v0, v0, cs2[j][0]); cs2[j][1] = fmaf(v1, v1, cs2[j][1]);
        cs2[j][2] = fmaf(v2, v2, cs2[j][2]); cs2[j][3] = fmaf(v3, v3, cs2[j][3]);
      }
    };
    auto tile_end = [&]() {                                    // after the epilogue of a stage: statistics of a finished 128-row tile
      if (!CSTAT) return;
      if (++c_is < g::SPT) return;
      // sum over the 16 pixel lanes of a DPP row (rotations: every lane ends with the total, fixed order); lane 0 of each row stores 2 x 4 channels
      const int mt = mt0 + c_ci * mt_step;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float t1[4], t2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { t1[e] = row16_sum(cs1[j][e]); t2[e] = row16_sum(cs2[j][e]); cs1[j][e] = 0.f; cs2[j][e] = 0.f; }
        const int nn = n0 + wave * WCOLS + j * 16 + (lane >> 4) * 4;
        if ((lane & 15) == 0 && nn < a.Cout) {                 // Cout % 4 == 0 (host)
          *(float4*)(a.stats + ((long)mt * 2 + 0) * a.Cout + nn) = make_float4(t1[0], t1[1], t1[2], t1[3]);
          *(float4*)(a.stats + ((long)mt * 2 + 1) * a.Cout + nn) = make_float4(t2[0], t2[1], t2[2], t2[3]);
        }
      }
      c_is = 0; ++c_ci;
    };
    auto body = [&](auto do_mma, auto do_epi, int gi) {
      constexpr bool MMA = decltype(do_mma)::value, EPI = decltype(do_epi)::value;
      const char* st = smem + slot_c * g::SB + xrow;
      if (MMA) { if (++slot_c == D) slot_c = 0; }
      const unsigned sbase = sl_addr + (unsigned)(((gi - 1) & 1) * g::SLAB);      // slab of stage gi - 1
      f32x4 acc[TN][TM];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      constexpr int KSN = (SIMT_ROWS_ABL & 1) ? 0 : KS;
      // pixel fragments: requested PF k-steps ahead into PF + 1 rotating register sets (an LDS round trip under load is longer than the
      // 4 MFMAs of one k-step)
      constexpr int PF = TM == 2 ? 2 : TM == 1 ? SIMT_ROWS_PF1 : 1;     // (TM = 1, Cin = 1024: one MFMA per k-step -- a deeper fragment queue)
      bf16x8 xf[PF + 1][TM];
      const unsigned st_addr = (unsigned)(size_t)LPTR(st);
      const unsigned ad01[2] = {st_addr + (unsigned)(((0 + kq) ^ sw) << 4), st_addr + (unsigned)(((4 + kq) ^ sw) << 4)};   // even / odd k-steps
      auto frags = [&](int ks, bf16x8* f) {                    // request the pixel fragments of k-step ks (TM x ds_read_b128, immediate offsets)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (SIMT_ROWS_ABL & 128) asm volatile("" : "=v"(f[i]) : "v"(ad01[ks & 1]));
          else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i]) : "v"(ad01[ks & 1]), "n"((ks >> 1) * (g::RS * 128) + i * 16 * 128));
        }
      };
      auto landed = [&](bf16x8* f, int younger) {              // f is complete: everything but the `younger` most recent LDS operations has returned
        static_assert(TM == 1 || TM == 2 || TM == 4, "fragment blocks");
#define SIMT_LANDED(N) if (younger == N) { if (TM == 1) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0])); \
                                             else if (TM == 2) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1])); \
                                             else asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2 % TM]), "+v"(f[3 % TM])); }
        SIMT_LANDED(0) SIMT_LANDED(1) SIMT_LANDED(2) SIMT_LANDED(3) SIMT_LANDED(4) SIMT_LANDED(5) SIMT_LANDED(6)
#undef SIMT_LANDED
      };
      if (MMA) {
#pragma unroll
        for (int k0 = 0; k0 < PF && k0 < KSN; ++k0) frags(k0, xf[k0 % (PF + 1)]);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (MMA && ks + PF < KSN) frags(ks + PF, xf[(ks + PF) % (PF + 1)]);
        if (MMA && ks < KSN) {
          const int ahead = (KSN - 1 - ks) < PF ? (KSN - 1 - ks) : PF;      // k-steps requested after this one
          landed(xf[ks % (PF + 1)], ahead * TM);
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][ks], xf[ks % (PF + 1)][i], acc[j][i], 0, 0, 0);
        }
        if (EPI) {                                             // the quads of the previous stage, spread evenly over the k-steps
#pragma unroll
          for (int qd = 0; qd < NQ; ++qd) if (!(SIMT_ROWS_ABL & 64) && qd * KS / NQ == ks) epi_quad(qd, sbase);
        }
      }
      if (MMA && EPI) {                                        // 1 MFMA, then at most 2 VALU, ... (whatever is left follows the last MFMA)
#pragma unroll
        for (int n = 0; n < KSN * NQ; ++n) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
      }
      if (EPI) tile_end();
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) prev[j][i] = acc[j][i];
    };
    if (INBN) {
      // stage 0 must be normalised by the store waves BEFORE period 0 multiplies it: one extra hand-over (landed -> barrier -> their period)
      if (D - 2 < S_total) wait_vmcnt<g::PT * (D - 2)>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    for (int gi = 0; gi <= S_total; ++gi) {
#ifdef SIMT_ABLATION
      const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
      // stage gi landed (this wave's pieces); INBN: stage gi + 1 -- the store waves normalise it during period gi
      if (INBN) { if (gi + D - 2 < S_total) wait_vmcnt<g::PT * (D - 3)>(); else wait_vmcnt<0>(); }
      else if (gi + D - 2 < S_total) wait_vmcnt<g::PT * (D - 2)>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  // this wave's slab writes (stage gi - 2)
#ifdef SIMT_ABLATION
      const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
#endif
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#ifdef SIMT_ABLATION
      const unsigned long long tw2 = __builtin_amdgcn_s_memtime();
      t_wait += tw1 - tw0; t_bar += tw2 - tw1;
#endif
      if (gi + D - 1 < S_total) issue(slot_i);                 // into the slot every wave finished reading in stage gi - 1
      if (++slot_i == D) slot_i = 0;
      if (gi == 0) body(std::true_type{}, std::false_type{}, gi);
      else if (gi < S_total) body(std::true_type{}, std::true_type{}, gi);
      else body(std::false_type{}, std::true_type{}, gi);
#ifdef SIMT_ABLATION
      asm volatile("s_nop 0" ::: "memory");
      t_work += __builtin_amdgcn_s_memtime() - tw2;
#endif
    }
#ifdef SIMT_ABLATION
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
      unsigned long long* o = g_stamps + blockIdx.x * 8;
      o[0] = t_wait; o[1] = t_bar; o[2] = t_work; o[3] = __builtin_amdgcn_s_memtime() - t_begin; o[4] = (unsigned long long)S_total;
    }
    if (lane == 0 && blockIdx.x < 256) {                       // per compute wave: work / barrier ticks (second half of the stamp array)
      unsigned long long* o = g_stamps + 4096 * 8 + (blockIdx.x * 8 + wave) * 2;
      o[0] = t_work; o[1] = t_bar;
    }
#endif
    // drain: the store waves run one (two: pipelined path) slabs behind the last slab write, then the last tile's sums go through sR
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (PIPE) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    return;
  }

  // ================================ store waves: slab g - 1 -> epilogue -> HBM while stage g is computed ================================
#ifndef SIMT_ROWS_PRIO
#define SIMT_ROWS_PRIO 0
#endif
  if (SIMT_ROWS_PRIO) __builtin_amdgcn_s_setprio(SIMT_ROWS_PRIO);
  const int st = tid - NC;
  const int swv = wave - NCW;
  const int vcol = (st % VPR) * 8, rg = st / VPR;
  const int n = n0 + vcol;
  const bool ncol_ok = n < a.Nstore;
  float bias8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias8[e] = (has_bias && ncol_ok && (n + e) < a.Cout) ? a.bias[n + e] : 0.f;
  // fused BN-backward reduce (bit-mask flavour): the mean of the BatchNorm whose dz this is.  S2 = sum g * xhat is taken as rstd * sum g * (y - mean),
  // the factor applied once per 128-row tile and column (sums_out) -- the order PyTorch's batch_norm backward uses, and one multiply per element less
  // in store waves that are bound by their vector work
  float bmu[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bmu[e] = 0.f;
  if (has_bnr && ncol_ok) load8(a.bnr_mean + n, bmu);
  // Pin the waits of these loads HERE: left to the compiler, its s_waitcnt vmcnt(0) lands at the head of the slab loop, where it would
  // also wait for every output store of the previous slab to be acknowledged by HBM.
#pragma unroll
  for (int e = 0; e < 8; ++e) { asm volatile("" : "+v"(bias8[e])); if (AUX) { asm volatile("" : "+v"(bmu[e])); } }
  const bool plain = !has_bias && !has_res && !has_relu;
  const bool want_sums = has_stats || has_bnr;
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }

  // operands the epilogue reads from global memory: requested one slab ahead (pass p of the next slab replaces pass p of this one).
  // All addressing is 32-bit byte offsets from uniform bases (host: every operand < 4 GB).
  const char* yb = (const char*)a.y;
  const char* resb = (const char*)a.res;
  const char* byb = (const char*)a.bnr_y;
  const unsigned y_pitch = (unsigned)a.ldy * 2u, r_pitch = (unsigned)a.ldr * 2u, b_pitch = (unsigned)a.bnr_ld * 2u;
  const unsigned y_c = (unsigned)rg * y_pitch + (unsigned)n * 2u;             // thread constants: row group and column of this thread
  const unsigned r_c = (unsigned)rg * r_pitch + (unsigned)n * 2u;
  const unsigned b_c = (unsigned)rg * b_pitch + (unsigned)n * 2u;
  const bool all_cols = n0 + BN <= a.Nstore;                   // uniform
  Aux q[g::PASSES];
  auto aux_load = [&](Aux& d, int slab_row0, int p, bool guard) {      // guard: uniform (false for full slabs of full-width tiles: no zero fill, no exec masking)
    if (!AUX || (guard && !(ncol_ok && slab_row0 + rg + p * RGS < a.M))) {
      d.res = d.by = make_uint4(0u, 0u, 0u, 0u);
      d.rbits = d.ybits = 0xffu;
      return;
    }
    if (!has_res) { d.res = make_uint4(0u, 0u, 0u, 0u); d.rbits = 0xffu; }
    if (!has_bnr) { d.by = make_uint4(0u, 0u, 0u, 0u); d.ybits = 0xffu; }
    if (has_res && !has_rbits) d.rbits = 0xffu;
    if (has_res) {
      unsigned o = (unsigned)(slab_row0 + p * RGS) * r_pitch + r_c;
      if (SIMT_ROWS_ABL & 512) o = (o & 0x3ff0u) + blockIdx.x * 0x4000u;     // ablation: the operands come from 16 KB per workgroup (L2-resident)
      d.res = *(const uint4*)(resb + o);
      if (has_rbits) d.rbits = a.res_bits[o >> 4];
    }
    if (has_bnr) {
      unsigned o = (unsigned)(slab_row0 + p * RGS) * b_pitch + b_c;
      if (SIMT_ROWS_ABL & 512) o = (o & 0x3ff0u) + blockIdx.x * 0x4000u;
      d.by = *(const uint4*)(byb + o);
      d.ybits = a.bnr_bits[o >> 4];
    }
  };
  int ci = 0, is = 0;                                          // slab cursor: tile of this workgroup, stage inside the tile
  int cur_mt = mt0;
  int row0 = cur_mt * 128;                                     // first pixel row of the slab being processed
  auto next_row0 = [&](int& mt_next) {                        // first row of the slab after the cursor
    if (is + 1 < g::SPT) { mt_next = cur_mt; return row0 + g::RS; }
    mt_next = mt0 + (ci + 1) * mt_step;
    return mt_next * 128;
  };
  if (AUX && S_total > 0) {
#pragma unroll
    for (int p = 0; p < g::PASSES; ++p) aux_load(q[p], row0, p, true);
  }
  bool stats_pending = false;
  int pend_mt = 0;
  auto sums_out = [&](int mt) {                                // the store waves' partial sums in fixed order -> HBM
    const int nn = n0 + st;                                    // one column per store thread (the first BN of them)
    if (st < BN && nn < a.Cout) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NSW; ++w) { t1 += sR[(w * 2 + 0) * BN + st]; t2 += sR[(w * 2 + 1) * BN + st]; }
      if (has_bnr) {                                           // [m-tile][3][Cout]: S1, S2 and the (unused) second-BN row
        a.bnr_part[((long)mt * 3 + 0) * a.Cout + nn] = t1;
        // (S2 = rstd * sum g (y - mean): rstd once per tile and column.  conv2_epilogue.h / simt_bn_bwd keep sum g ((y - mean) rstd): the
        // same BatchNorm backward through this kernel and through conv_igemm2 agrees to rounding (5e-6 per layer), not bit for bit --
        // INTEGRATION.md, SIMT_NO_ROWS)
        a.bnr_part[((long)mt * 3 + 1) * a.Cout + nn] = t2 * a.bnr_rstd[nn];
        a.bnr_part[((long)mt * 3 + 2) * a.Cout + nn] = 0.f;
      } else {
        a.stats[((long)mt * 2 + 0) * a.Cout + nn] = t1;
        a.stats[((long)mt * 2 + 1) * a.Cout + nn] = t2;
      }
    }
  };

#ifdef SIMT_ABLATION
  unsigned long long ts_bar = 0, ts_work = 0, ts_prev = 0, ts_lds = 0;
#endif
  // MSTAT: this wave's channel blocks (16 channels each): running ones x Y and Y^T x Y of the tile in progress
  constexpr int MBLK = BN / 16 / NSW;
  f32x4 m1[MBLK], m2[MBLK];
#pragma unroll
  for (int t = 0; t < MBLK; ++t) { m1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; m2[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  int ms_is = 0, ms_ci = 0;                                    // cursor of the slab the statistics are taken from (the one just finished)
  const bf16x8 ones8 = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
  // transposed-read addressing: lane 4q + p of a 16-lane group supplies row q, channels 4p .. 4p + 3 of a 4-pixel x 16-channel block;
  // group g = lane >> 4 covers pixels 8g .. 8g + 7 of a 32-pixel k-step (two reads)
  const unsigned tr_lane = (unsigned)(((lane >> 4) * 8 + ((lane & 15) >> 2)) * CP + (lane & 3) * 8);
  uint4 ov[g::PASSES];                                         // PIPE: the rows of the slab to be stored, read during the previous stage
  const int LAST = PIPE ? S_total + 1 : S_total;
  // ---- INBN: BatchNorm + ReLU of the landed input stage, in place in its ring slot (store thread st owns the linear 16-byte pieces
  // q * NS + st of a stage: row (p >> 3) % RS, 64-channel sub-tile p / (RS * 8), chunk (p & 7) ^ ((row >> 1) & 7) -- the compute waves' LDS-DMA layout)
  constexpr int PTS = INBN ? g::SB / 16 / NS : 1;               // pieces per store thread and stage
  float isc[PTS][8], ish[PTS][8];
  unsigned t_loff[PTS], t_goff[PTS];                           // piece's byte offset in the slot / in the activation (row pitch CIN * 2)
  int t_row[PTS];
  bool t_mine[PTS];                                            // the 1 / ntiles_n share of the activation this workgroup writes out
  if (INBN) {
#pragma unroll
    for (int q = 0; q < PTS; ++q) {
      const int p = q * NS + st;
      const int row = (p >> 3) % g::RS, kc = p / (g::RS * 8), cg = (p & 7) ^ ((row >> 1) & 7);
      const int ch = kc * 64 + cg * 8;
      t_loff[q] = (unsigned)p * 16u;
      t_row[q] = row;
      t_goff[q] = (unsigned)row * (unsigned)(g::CIN * 2) + (unsigned)ch * 2u;
      t_mine[q] = (q * a.ntiles_n) / PTS == nt;                 // (host: PTS % ntiles_n == 0)
      load8(a.in_scale + ch, isc[q]);
      load8(a.in_shift + ch, ish[q]);
#pragma unroll
      for (int e = 0; e < 8; ++e) { asm volatile("" : "+v"(isc[q][e])); asm volatile("" : "+v"(ish[q][e])); }      // (waits pinned here, like bias8)
    }
  }
  int t_is = 0, t_ci = 0;                                      // cursor of the next stage to normalise: stage inside its tile, tile of this workgroup
  auto in_bn = [&](int s) {                                    // stage s (landed: the compute waves waited for it before the barrier just passed)
    if (!INBN || s >= S_total) return;
    const int srow0 = (mt0 + t_ci * mt_step) * 128 + t_is * g::RS;
    if (++t_is == g::SPT) { t_is = 0; ++t_ci; }
    const unsigned slot = (unsigned)(size_t)LPTR(smem + (s % D) * g::SB);
    const bool full = srow0 + g::RS <= a.M;                    // uniform; rows past the end stay the zero page's zeros (they add nothing to the statistics)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // (a native vector: HIP's uint4 is a struct, which an asm "v" operand cannot be)
    u32x4 xin[PTS];
#pragma unroll
    for (int q = 0; q < PTS; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(xin[q]) : "v"(slot + t_loff[q]));
#pragma unroll
    for (int q = 0; q < PTS; ++q) {
      // (asm loads, asm wait tied to the registers: the compiler's waitcnt insertion does not see asm operands -- DESIGN.md 5b "a measured trap")
      if (q == 0) {
        if constexpr (PTS == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xin[0]));
        else if constexpr (PTS == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xin[0]), "+v"(xin[PTS - 1]));
        else if constexpr (PTS == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xin[0]), "+v"(xin[1]), "+v"(xin[PTS - 2]), "+v"(xin[PTS - 1]));
      }
      if (!full && srow0 + t_row[q] >= a.M) continue;
      float v[8];
      unpack8(make_uint4(xin[q][0], xin[q][1], xin[q][2], xin[q][3]), v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * isc[q][e] + ish[q][e];           // simt_bn_apply's expression (same contraction: bitwise its output)
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
      asm volatile("ds_write_b128 %0, %1" :: "v"(slot + t_loff[q]), "v"(o) : "memory");
      if (t_mine[q]) st_out16((char*)a.in_out + (size_t)srow0 * (size_t)(g::CIN * 2) + t_goff[q], make_uint4(o[0], o[1], o[2], o[3]));
    }
  };
  static_assert(!INBN || PTS == 1 || PTS == 2 || PTS == 4, "in_bn's wait names one, two or four pieces");
  if (INBN) {
    __builtin_amdgcn_s_barrier();                              // the compute waves' preamble barrier: stage 0 landed
    in_bn(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();                                // the compute waves' barrier 0: they write the slab of stage g - 1 during stage g,
                                                               // so "barrier gi" below is their barrier gi + 1
  in_bn(1);                                                    // period 0 (stage 1 landed before barrier 0)
  for (int gi = 0; gi <= LAST; ++gi) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // slab reads / sR writes of the previous iteration
#ifdef SIMT_ABLATION
    const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
#ifdef SIMT_ABLATION
    const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
    ts_bar += tb1 - tb0;
    if (gi > 0) ts_work += tb0 - ts_prev;
#endif                              // barrier gi: slab gi - 1 is complete; slab gi - 2 may be overwritten
    if (stats_pending) { sums_out(pend_mt); stats_pending = false; }
#ifdef SIMT_ABLATION
    ts_prev = tb1;
#endif
    if (gi == 0 || (SIMT_ROWS_ABL & 8)) { in_bn(gi + 2); continue; }
    const char* sl = slab + ((gi - 1) & 1) * g::SLAB;
    if (PIPE) {
      const bool guard_cur = !(all_cols && row0 + g::RS <= a.M);
      if (gi >= 2) {                                           // slab gi - 2: (bias, ReLU,) stores
#pragma unroll
        for (int p = 0; p < g::PASSES; ++p) {
          if (guard_cur && !(ncol_ok && row0 + rg + p * RGS < a.M)) continue;
          uint4 o = ov[p];
          if (!plain) {
            float v[8];
            unpack8(o, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[e] += bias8[e]; if (has_relu) v[e] = v[e] > 0.f ? v[e] : 0.f; }
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
          }
          unsigned yo = (unsigned)(row0 + p * RGS) * y_pitch + y_c;
          if (SIMT_ROWS_ABL & 32) yo = (yo & 0x3fffu) + blockIdx.x * 0x4000u;
          if (!(SIMT_ROWS_ABL & 2) || o.x == 0x12345678u) st_out16(const_cast<char*>(yb) + yo, o);
        }
      }
#ifdef SIMT_ABLATION
      asm volatile("s_nop 0" ::: "memory");
      ts_lds += __builtin_amdgcn_s_memtime() - tb1;            // (PIPE: barrier -> stores issued)
#endif
      uint4 nv[g::PASSES];
      if (gi <= S_total) {                                     // slab gi - 1: LDS -> registers (in flight during the statistics below)
#pragma unroll
        for (int p = 0; p < g::PASSES; ++p) nv[p] = *(const uint4*)(sl + (rg + p * RGS) * CP + vcol * 2);
        in_bn(gi + 2);                                         // INBN: the stage the compute waves multiply NEXT period (behind the output stores: they block on HBM)
        if (MSTAT) {
          const unsigned sl_a = (unsigned)(size_t)LPTR(sl) + tr_lane;
#pragma unroll
          for (int t = 0; t < MBLK; ++t) {
            const int c0 = (swv * MBLK + t) * 16;              // channel block inside the workgroup's columns
#pragma unroll
            for (int ks = 0; ks < g::RS / 32; ++ks) {
              bf16x4 lo, hi;
              asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(sl_a + (unsigned)(c0 * 2)), "n"(ks * 32 * CP));
              asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(sl_a + (unsigned)(c0 * 2)), "n"(ks * 32 * CP + 4 * CP));
              bf16x8 yb = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yb));
              m1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones8, yb, m1[t], 0, 0, 0);
              m2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yb, yb, m2[t], 0, 0, 0);
            }
          }
          if (++ms_is == g::SPT) {                             // the slab closes a 128-row tile: row 0 of ones x Y, diagonal of Y^T x Y
            const int mt = mt0 + ms_ci * mt_step;
#pragma unroll
            for (int t = 0; t < MBLK; ++t) {
              const int nn = n0 + (swv * MBLK + t) * 16 + (lane & 15);
              const int e = lane & 3;
              const float dg = e == 0 ? m2[t][0] : e == 1 ? m2[t][1] : e == 2 ? m2[t][2] : m2[t][3];
              if (nn < a.Cout) {
                if (lane < 16) a.stats[((long)mt * 2 + 0) * a.Cout + nn] = m1[t][0];
                if (((lane & 15) >> 2) == (lane >> 4)) a.stats[((long)mt * 2 + 1) * a.Cout + nn] = dg;
              }
              m1[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; m2[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            ms_is = 0; ++ms_ci;
          }
        }
      }
      if (gi >= 2) {
        if (has_stats) {                                       // statistics of the stored value, before bias / ReLU
#pragma unroll
          for (int p = 0; p < g::PASSES; ++p) {
            if (guard_cur && !(ncol_ok && row0 + rg + p * RGS < a.M)) continue;
            float v[8];
            unpack8(ov[p], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
          }
        }
        int mt_next;
        const int nrow0 = next_row0(mt_next);
        if (is + 1 == g::SPT) {                                // end of a 128-row tile
          if (want_sums) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float t1 = row_groups_sum<64 / VPR>(s1[e]), t2 = row_groups_sum<64 / VPR>(s2[e]);
              if (lane < VPR) { sR[(swv * 2 + 0) * BN + vcol + e] = t1; sR[(swv * 2 + 1) * BN + vcol + e] = t2; }
              s1[e] = 0.f; s2[e] = 0.f;
            }
            stats_pending = true; pend_mt = cur_mt;
          }
          is = 0; ++ci;
        } else {
          ++is;
        }
        row0 = nrow0; cur_mt = mt_next;
      }
#pragma unroll
      for (int p = 0; p < g::PASSES; ++p) ov[p] = nv[p];
      continue;
    }
    int mt_next;
    const int nrow0 = next_row0(mt_next);
    const bool more = gi < S_total;
    // all slab rows of this thread first (one LDS round trip per slab), then one 16-byte store per row from a single store site (with
    // several, the compiler sinks them into a shared tail of a 4-byte and a 12-byte store)
#pragma unroll
    for (int p = 0; p < g::PASSES; ++p) if (!AUX) ov[p] = *(const uint4*)(sl + (rg + p * RGS) * CP + vcol * 2);
#ifdef SIMT_ABLATION
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ts_lds += __builtin_amdgcn_s_memtime() - tb1;
#endif
    // full slabs of full-width tiles take the branch-free path (uniform test); only the last tile / a narrow last column tile is guarded
    const bool guard_cur = !(all_cols && row0 + g::RS <= a.M);
    const bool guard_next = !(all_cols && nrow0 + g::RS <= a.M);        // uniform
    auto pass = [&](int p, bool guard) {
      const Aux cur = q[p];
      if (AUX && more) aux_load(q[p], nrow0, p, guard_next);
      if (guard && !(ncol_ok && row0 + rg + p * RGS < a.M)) return;
      uint4 o = AUX ? *(const uint4*)(sl + (rg + p * RGS) * CP + vcol * 2) : ov[p];
      if (!plain || want_sums) {
        float v[8];
        unpack8(o, v);
        if (has_stats) {                                       // forward: statistics of the stored value, before bias / residual / ReLU
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
        }
        if (!plain) {
          if (has_bias) {                                      // (x + 0.0f is not a no-op the compiler may drop: 8 adds per row in the bias-free flavours)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
          }
          if (has_res) {
            float rv[8];
            unpack8(cur.res, rv);
            // bit e of the mask as an all-ones / all-zeros word (one v_bfe_i32) ANDed onto the value: 2 vector instructions per element where
            // test + compare + select take 3 -- these store waves are bound by their vector work (profiles/tools/stamps3.py)
#pragma unroll
            for (int e = 0; e < 8; ++e)
              v[e] += has_rbits ? __uint_as_float(__float_as_uint(rv[e]) & (unsigned)__builtin_amdgcn_sbfe((int)cur.rbits, e, 1)) : rv[e];
          }
          if (has_relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        }
        if (has_bnr) {
          // backward: S1 = sum g, S2 = sum g * xhat on the value as stored (bf16), masked like the backward masks it
          if (!plain) unpack8(o, v);
          float yv[8];
          unpack8(cur.by, yv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(__float_as_uint(v[e]) & (unsigned)__builtin_amdgcn_sbfe((int)cur.ybits, e, 1));
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * (yv[e] - bmu[e]); }
        }
      }
      unsigned yo = (unsigned)(row0 + p * RGS) * y_pitch + y_c;
      if (SIMT_ROWS_ABL & 32) yo = (yo & 0x3fffu) + blockIdx.x * 0x4000u;        // ablation: every store hits the same 16 KB per workgroup (L2-resident)
      if (!(SIMT_ROWS_ABL & 2) || o.x == 0x12345678u) st_out16(const_cast<char*>(yb) + yo, o);
    };
    if (guard_cur || FL == FL_GEN_AUX) {                       // (one copy of the pass code in the run-time-flag aux flavour: registers)
#pragma unroll
      for (int p = 0; p < g::PASSES; ++p) pass(p, true);
    } else {
#pragma unroll
      for (int p = 0; p < g::PASSES; ++p) pass(p, false);
    }
    // end of a 128-row tile: the two row groups of a wave by a lane swap, the store waves through sR (read behind the next barrier)
    if (is + 1 == g::SPT) {
      if (want_sums) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float t1 = row_groups_sum<64 / VPR>(s1[e]), t2 = row_groups_sum<64 / VPR>(s2[e]);
          if (lane < VPR) { sR[(swv * 2 + 0) * BN + vcol + e] = t1; sR[(swv * 2 + 1) * BN + vcol + e] = t2; }
          s1[e] = 0.f; s2[e] = 0.f;
        }
        stats_pending = true; pend_mt = cur_mt;
      }
      is = 0; ++ci;
    } else {
      ++is;
    }
    row0 = nrow0; cur_mt = mt_next;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (stats_pending) sums_out(pend_mt);
#ifdef SIMT_ABLATION
  if (threadIdx.x == NC && blockIdx.x < 8192) { g_stamps[blockIdx.x * 8 + 5] = ts_bar; g_stamps[blockIdx.x * 8 + 6] = ts_work; g_stamps[blockIdx.x * 8 + 7] = ts_lds; }
#endif
}

template <int KS, int TM, int D, int FL, int NSW, int NCW, int TN>
__global__ __launch_bounds__((NCW + NSW) * 64, (KS == 32 ? 2 : 3)) void conv1x1_rows_kernel(Conv2KArgs a, int G) {
  conv1x1_rows_body<KS, TM, D, FL, NSW, NCW, TN>(a, G, (int)blockIdx.x);
}

// Two problems of identical geometry in one launch (simt_conv_fprop_pair; conv_igemm2_pair_kernel has the why): persistent workgroups
// [0, G) stream problem 0 with flavour FL0, [G, 2 G) problem 1 with flavour FL1.  One workgroup per CU: the second G start as the first finish.
template <int KS, int TM, int D, int FL0, int FL1, int NSW, int NCW, int TN>
__global__ __launch_bounds__((NCW + NSW) * 64, (KS == 32 ? 2 : 3)) void conv1x1_rows_pair_kernel(Conv2KArgs a0, Conv2KArgs a1, int G) {
  if ((int)blockIdx.x < G) conv1x1_rows_body<KS, TM, D, FL0, NSW, NCW, TN>(a0, G, (int)blockIdx.x);
  else conv1x1_rows_body<KS, TM, D, FL1, NSW, NCW, TN>(a1, G, (int)blockIdx.x - G);
}

template <int KS, int TM, int D, int FL0, int FL1, int NSW, int NCW, int TN = 2>
int launch_rows_pair(Conv2KArgs k0, Conv2KArgs k1, int npad, hipStream_t st) {
  using g = Geo<KS, TM, D, NSW, NCW, TN>;
  k0.rows = k1.rows = 128;
  k0.ntiles_n = k1.ntiles_n = npad / g::BN;
  k0.ntiles_m = k1.ntiles_m = (k0.M + 127) / 128;
  const int nwg = k0.ntiles_m * k0.ntiles_n;
  const int cap = (NCW == 8 || g::LDS > 80 * 1024) ? 256 : 512;
  const int G = nwg < cap ? nwg : cap;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, g::LDS))
    (void)hipFuncSetAttribute((const void*)conv1x1_rows_pair_kernel<KS, TM, D, FL0, FL1, NSW, NCW, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, g::LDS);
  hipLaunchKernelGGL((conv1x1_rows_pair_kernel<KS, TM, D, FL0, FL1, NSW, NCW, TN>), dim3(2 * G), dim3(g::NT), g::LDS, st, k0, k1, G);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

template <int KS, int TM, int D, int FL, int NSW, int NCW, int TN = 2>
int launch_rows(Conv2KArgs k, int npad, hipStream_t st) {
  using g = Geo<KS, TM, D, NSW, NCW, TN>;
  k.rows = 128;
  k.ntiles_n = npad / g::BN;
  k.ntiles_m = (k.M + 127) / 128;
  const int nwg = k.ntiles_m * k.ntiles_n;
  const int cap = (NCW == 8 || g::LDS > 80 * 1024) ? 256 : 512;    // persistent: one or two (NCW = 4: 384 threads) workgroups per CU
  const int G = nwg < cap ? nwg : cap;
  static SimtLdsAttrCache attr_cache;
  if (simt_lds_attr_needed(&attr_cache, g::LDS))
    (void)hipFuncSetAttribute((const void*)conv1x1_rows_kernel<KS, TM, D, FL, NSW, NCW, TN>, hipFuncAttributeMaxDynamicSharedMemorySize, g::LDS);
  hipLaunchKernelGGL((conv1x1_rows_kernel<KS, TM, D, FL, NSW, NCW, TN>), dim3(G), dim3(g::NT), g::LDS, st, k, G);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

}  // namespace

#ifdef SIMT_ABLATION
extern "C" int simt_debug_stamps_rows(unsigned long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), (size_t)n * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : 1;
}
#endif

// Shapes this kernel takes over from conv_igemm2_kernel<128, *, 2> / conv1x1_stream_kernel (simt_conv_fprop_bf16_v2 fills the arguments).
bool simt_conv_rows_eligible(const simt_conv_desc* d) {
  if (d->dtype_in != SIMT_BF16 || d->dtype_out != SIMT_BF16 || d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0) return false;
  if (d->stride != 1 || d->H != d->Ho || d->W != d->Wo) return false;          // dense pixel rows
  if (d->mask || (d->bnr_mode != 0 && d->bnr_mode != 3)) return false;         // fused BN-backward reduce: bit-mask flavour only
  if (d->Cin != 64 && d->Cin != 128 && d->Cin != 256 && d->Cin != 512 && d->Cin != 1024) return false;
  if (d->Cin == 1024 && (d->bnr_mode || d->res)) return false;                 // long reduction: the forward flavours only (statistics / bias + ReLU)
  if (d->Cin == 512 && d->bnr_mode) return false;                              // measured: 214 us here vs 203 us on conv_igemm2_kernel<128, 4, 2>
  if (d->Npad % 256 != 0) return false;
  const long M = (long)d->B * d->Ho * d->Wo, lim = 1l << 32;                   // 32-bit byte offsets in the store waves
  if (M * d->ldy * 2 >= lim || (d->res && M * d->ldr * 2 >= lim) || (d->bnr_mode && M * d->bnr_ld * 2 >= lim)) return false;
  if (d->ldy % 8 != 0 || (d->res && d->ldr % 8 != 0)) return false;
  const int ntn = d->Npad / 128;                                               // column tiles of the narrowest variant
  return ntn == 2 || ntn == 4 || ntn == 8 || ntn == 16 || ntn == 32;           // ntiles_n | grid / 8
}

// simt_conv_desc.in_scale / in_shift / in_out (BatchNorm + ReLU of the input in the operand path): the statistics flavour on Cin 64 / 128 / 256,
// where the pieces a store thread normalises per stage (2 / 4 / 4) are a multiple of the column tiles that share a pixel row (Cout / 256)
bool simt_conv_rows_inbn_ok(const simt_conv_desc* d) {
  if (!simt_conv_rows_eligible(d)) return false;
  if (!d->stats || d->bias || d->relu || d->res || d->bnr_mode || d->Cout % 4 != 0) return false;
  if (d->Cin != 64 && d->Cin != 128 && d->Cin != 256) return false;
  const int pts = d->Cin == 64 ? 2 : 4, ntn = d->Npad / 256;
  return ntn >= 1 && pts % ntn == 0 && (long)d->B * d->Ho * d->Wo * d->Cin * 2 < (1l << 32);
}

int simt_conv_rows_launch(Conv2KArgs k, int npad, hipStream_t st) {
  const bool aux = k.res || k.bnr_mode;
  const int cin = k.pix_bytes / 2;
  const bool f_stats = k.stats && !k.bias && !k.relu && !aux && k.Cout % 4 == 0;
  const bool f_inbn = f_stats && k.in_scale;                    // (simt_conv_inbn_ok: checked by conv2_fill_args)
  const bool f_brr = k.bias && k.relu && k.res && !k.res_bits && !k.bnr_mode && !k.stats;
  const bool f_bnr = k.res && k.res_bits && k.bnr_mode == 3 && !k.bias && !k.relu && !k.stats;
#ifndef SIMT_ROWS_NCW
#define SIMT_ROWS_NCW 8
#endif
  constexpr int CW = SIMT_ROWS_NCW, SW = CW / 2, D256 = CW == 8 ? 6 : 3;
  // SIMT_ROWS_AUX_SW (compile-time A/B): store waves of the residual flavours at Cin = 256 -- low nibble: bias + residual + ReLU, high nibble (or the
  // same value): bit-masked residual + BatchNorm-backward reduce.  8 = the 1 024-thread form (two rows per store thread and slab, 124 / 128 VGPRs):
  // parity green, measured no change (61.7 / 42.5 us: profiles/r06_rows_aux.txt -- the store path's vector work does not hide under the MFMAs whoever issues it)
#ifndef SIMT_ROWS_AUX_SW
#define SIMT_ROWS_AUX_SW 4
#endif
  constexpr int ASW = SIMT_ROWS_AUX_SW;
  if (cin == 256) {
    if (f_inbn) return launch_rows<8, 2, 6, FL_STATS_INBN, 4, 8>(k, npad, st);
    if (f_stats) return launch_rows<8, 2, D256, FL_STATS, SW, CW>(k, npad, st);
    if (f_brr) return launch_rows<8, 2, D256, FL_BRR, (ASW & 15), CW>(k, npad, st);
    if (f_bnr) return launch_rows<8, 2, D256, FL_BNR, (ASW >> 4 ? ASW >> 4 : ASW), CW>(k, npad, st);
    return aux ? launch_rows<8, 2, 6, FL_GEN_AUX, 4, 8>(k, npad, st) : launch_rows<8, 2, 6, FL_GEN, 4, 8>(k, npad, st);
  }
  if (cin == 1024) {                                           // 16 channels per wave (128 weight registers), 16-row stages of whole 2-KB rows
    if (f_stats) return launch_rows<32, 1, 4, FL_STATS, 1, 4, 1>(k, npad, st);
    return launch_rows<32, 1, 4, FL_GEN, 1, 4, 1>(k, npad, st);
  }
  if (cin == 512) {                                            // 16 channels per wave: 128-column workgroups, 8 + 2 waves
    if (f_stats) return launch_rows<16, 2, 3, FL_STATS, 2, 8, 1>(k, npad, st);
    if (f_brr) return launch_rows<16, 2, 3, FL_BRR, 2, 8, 1>(k, npad, st);
    return aux ? launch_rows<16, 2, 3, FL_GEN_AUX, 2, 8, 1>(k, npad, st) : launch_rows<16, 2, 3, FL_GEN, 2, 8, 1>(k, npad, st);
  }
  // Cin 128 / 64: 64-row stages without global epilogue operands; with them 32-row stages (the operands of a 64-row slab do not fit
  // the store waves' registers), for Cin = 64 as two 384-thread workgroups per CU (a 32-row stage is only 4 KB)
  if (cin == 128) {
    if (f_inbn) return launch_rows<4, 4, 4, FL_STATS_INBN, 4, 8>(k, npad, st);
    if (f_stats) return launch_rows<4, 4, 4, FL_STATS, 4, 8>(k, npad, st);
    if (f_brr) return launch_rows<4, 2, 6, FL_BRR, 4, 8>(k, npad, st);
    if (f_bnr) return launch_rows<4, 2, 6, FL_BNR, 4, 8>(k, npad, st);
    return aux ? launch_rows<4, 2, 6, FL_GEN_AUX, 4, 8>(k, npad, st) : launch_rows<4, 4, 4, FL_GEN, 4, 8>(k, npad, st);
  }
  if (f_inbn) return launch_rows<2, 4, 6, FL_STATS_INBN, 4, 8>(k, npad, st);
  if (f_stats) return launch_rows<2, 4, 6, FL_STATS, 4, 8>(k, npad, st);
  if (f_brr) return launch_rows<2, 2, 6, FL_BRR, 2, 4>(k, npad, st);
  if (f_bnr) return launch_rows<2, 2, 6, FL_BNR, 2, 4>(k, npad, st);
  return aux ? launch_rows<2, 2, 6, FL_GEN_AUX, 2, 4>(k, npad, st) : launch_rows<2, 4, 6, FL_GEN, 4, 8>(k, npad, st);
}

// (trainable conv3 with BatchNorm statistics, frozen conv3 with bias + residual + ReLU) of one layer in one launch: the geometries both
// flavours share (Cin 256: layer 3, Cin 512: layer 4)
int simt_conv_rows_pair_launch(Conv2KArgs k0, Conv2KArgs k1, int npad, hipStream_t st, bool* taken) {
  const int cin = k0.pix_bytes / 2;
  constexpr int CW = SIMT_ROWS_NCW, SW = CW / 2, D256 = CW == 8 ? 6 : 3;
  *taken = true;
  if (cin == 256) return launch_rows_pair<8, 2, D256, FL_STATS, FL_BRR, SW, CW>(k0, k1, npad, st);
  if (cin == 512) return launch_rows_pair<16, 2, 3, FL_STATS, FL_BRR, 2, 8, 1>(k0, k1, npad, st);
  *taken = false;
  return SIMT_ERR_INVALID;
}
