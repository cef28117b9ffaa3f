// Epilogue shared by the bf16 throughput conv kernels (conv_igemm2.hip, conv_igemm3.hip): accumulators -> bf16 tile in LDS -> whole rows
// streamed out with bias / residual / ReLU / mask applied on the way, BatchNorm statistics or the fused BatchNorm-backward reduce summed per
// row group in fixed order.  `compute_wave`: this wave holds accumulators (conv_igemm3's loader waves do not; they take part in the row
// streaming).  acc[j][i][e]: cout = n0 + (wn*TN + j)*16 + (lane>>4)*4 + e ; pixel row = (wm*TM + i)*16 + (lane&15).
#pragma once
#include "conv2_common.h"
#include <type_traits>

// Output rows go out as NON-TEMPORAL stores (st_out16 / st_out8, common.h): see there.
// Workgroup barriers BEHIND global stores are LDS-only (lds_barrier): __syncthreads() is a workgroup-scope release + acquire fence around the barrier,
// and the release makes the compiler drain vmcnt to 0 -- every row store of the tile acknowledged by the memory system (2-3 us) before the row-group
// sums may be combined through LDS.  Nothing in these kernels communicates through global memory inside a workgroup (the fused BatchNorm's granules are
// self-validating atomics), so the barrier only has to order LDS.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// 8-byte {value, tag} granules: ONE naturally aligned write-through store / L1-bypassing load each (never torn)
__device__ __forceinline__ void st_gran(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_gran(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Polling loops below: the other workgroups of the launch must become resident for a poll to end.  If something keeps them off the chip
// for ~2 s (two waiting launches of different processes starving each other, a collective's persistent kernels holding CUs: engine.py
// SIMT_BN_GRID), the poller does NOT trap (round 5): it sets the sticky error word of the plan and leaves its loop; every other poller checks the
// word every 256 polls and leaves too, so the launch -- and every later fused launch that shares the word -- ends within a poll period.  Round 6:
// a poller that gave up writes NOTHING derived from what it read -- an owner publishes no constants and leaves mean / rstd / scale / shift / the
// running statistics / coef / d gamma / d beta untouched, a workgroup whose constants poll gave up returns without its part of `out` (workgroups
// whose polls completed before the word was set have written theirs from complete sums): `out` is partially written, never garbage-normalised.  The host reads the word (TrunkPlan.fbn_error(); the trainers' losses() raise) and the optimiser
// kernels skip their update while it is set (simt_sgd_desc.skip_if).  s_memrealtime = constant 100 MHz (s_memtime counts core clocks).
__device__ __forceinline__ bool fbn_poll_gives_up(unsigned long long* err, unsigned long long t0) {
  if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return true;
  if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
    __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return true;
  }
  return false;
}
// FBN = 1: the instantiation can also run the fused train-mode BatchNorm (simt_fbn_desc, a.fbn_mode 1 / 2): the tail of this function.
// EPI: compile-time epilogue flavour.  0 = generic (every option a run-time flag: ~4 000 instructions, of which a launch executes ~1 000 per
// wave -- at two waves per SIMD that is 8 k clocks = 3.9 us of a 50 us conv, stamps in profiles/r04_bn_fusion.txt); 1 = BatchNorm
// statistics only (train-mode forward convs); 2 = fused BatchNorm-backward reduce, mask from y * scale + shift (the dgrads of a
// Bottleneck's conv3 / conv2); 3 = bias + ReLU (the frozen model's BN-folded convs); 4 = bit-masked residual + BatchNorm-backward reduce with
// the bit mask (the dx GEMM of a Bottleneck); 5 = nothing (plain dgrads); 6 = residual only; 7 = BatchNorm-backward reduce with the bit mask + an optional
// plain residual (the tap-expanded heads' dgrads into layer 3 / 4's output gradient); 8 = ReLU mask operand only (the dgrads of the BatchNorm-free VGG trunk).  The host picks the flavour from the descriptor
// (conv2_flavour in conv_igemm2.hip); every flavour computes exactly what the generic code computes for those flags.
template <int BN, int BM, int NT, int TN, int TM, int FBN = 0, int EPI = 0>
__device__ __forceinline__ void conv2_epilogue(const Conv2KArgs& a_, char* smem, f32x4 (&acc)[TN][TM], bool compute_wave, int wm, int wn,
                                               int tid, int lane, int m0, int n0, int m_end, int mt, int tile = 0) {
  // the flags as this flavour sees them: constants for EPI != 0 (the compiler drops every other path)
  struct Flags {
    const Conv2KArgs& k;
    __device__ __forceinline__ bool bias() const { return EPI == 0 ? k.bias != nullptr : EPI == 3; }
    __device__ __forceinline__ bool res() const { return (EPI == 0 || EPI == 7) ? k.res != nullptr : (EPI == 4 || EPI == 6); }
    __device__ __forceinline__ bool res_bits() const { return EPI == 0 ? k.res_bits != nullptr : EPI == 4; }
    __device__ __forceinline__ bool relu() const { return EPI == 0 ? k.relu != 0 : EPI == 3; }
    __device__ __forceinline__ bool mask() const { return EPI == 0 ? k.mask != nullptr : EPI == 8; }
    __device__ __forceinline__ bool stats() const { return EPI == 0 ? k.stats != nullptr : EPI == 1; }
    __device__ __forceinline__ int bnr() const { return EPI == 0 ? k.bnr_mode : (EPI == 2 ? 2 : (EPI == 4 || EPI == 7) ? 3 : 0); }
    __device__ __forceinline__ bool f32() const { return EPI == 0 ? k.out_f32 != 0 : false; }
  };
  const Conv2KArgs& a = a_;
  const Flags fl{a_};
  constexpr int CP = BN * 2 + 8;                   // epilogue tile pitch in bytes (bf16 row + 8 B pad)
  // ---------------- epilogue ----------------
  // acc[j][i][e]: cout = n0 + wn*TN*16 + j*16 + (lane>>4)*4 + e ; pixel row = wm*TM*16 + i*16 + (lane&15)
  if (fl.f32()) {
    // fp32 result (tap-expanded ASPP GEMM): every accumulator quad is 16 contiguous bytes of one pixel's row; four lanes
    // cover a 64-B segment -> stored directly, no LDS round trip
    if (!compute_wave) return;
    float* yf = (float*)a.y;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * TM * 16 + i * 16 + (lane & 15);
        const int c = n0 + wn * TN * 16 + j * 16 + (lane >> 4) * 4;
        if (m < m_end && c < a.Nstore) st_out16f(yf + (long)m * a.ldy + c, acc[j][i]);
      }
    return;
  }
  // Stream the tile out as whole rows (16 B per lane, 512-B rows) with bias / residual / ReLU applied on the way, and
  // accumulate the BatchNorm statistics (sum, sum of squares of the STORED bf16 values of the valid rows) per lane.
  constexpr int VPR = BN / 8;        // 16-B vectors per row
  constexpr int RPP = NT / VPR;      // rows per pass (= number of row groups)
  constexpr int NIT = (BM + RPP - 1) / RPP;   // rows per thread
  const int vcol = (tid % VPR) * 8;
  const int rg = tid / VPR;
  const int n = n0 + vcol;
  // Operands the epilogue reads from global memory (residual, its bit mask, the saved activation of the fused BatchNorm-backward
  // reduce, the VGG ReLU mask): ALL rows of this thread are requested here, right after the accumulators left for LDS (their
  // registers are free), so the HBM latency is paid once per workgroup (one row ahead, as in round 1, exposed it once per row: 4-10
  // dependent round trips per workgroup, the largest part of the short-K kernels' time).
  __syncthreads();
  char* sC = smem;                                   // [BM][CP] bytes, bf16
  if (compute_wave)
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r = wm * TM * 16 + i * 16 + (lane & 15);
      const int c = wn * TN * 16 + j * 16 + (lane >> 4) * 4;
      uint2 pk;
      pk.x = pack_bf16x2(acc[j][i][0], acc[j][i][1]);
      pk.y = pack_bf16x2(acc[j][i][2], acc[j][i][3]);
      *(uint2*)(sC + r * CP + c * 2) = pk;
    }
  // fused BatchNorm: this launch's generation on its BatchNorm's counters = a ticket drawn NOW (one returning agent-scope add by one lane),
  // needed only when the tile sums are published a few microseconds further down
  unsigned long long fbn_ticket = 0;
  if constexpr (FBN != 0) {
    if (a.fbn_mode && tid == 0)
      fbn_ticket = __hip_atomic_fetch_add(a.fbn_bar + (blockIdx.x & 7) * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  struct Aux { uint4 res, by; unsigned rbits, ybits; };   // by: saved activation (bnr) or ReLU mask operand (VGG): exclusive
  const bool aux = (fl.res() || fl.bnr() || fl.mask()) && n < a.Nstore;
  Aux q[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    q[it].res = q[it].by = make_uint4(0u, 0u, 0u, 0u);
    q[it].rbits = q[it].ybits = 0xffu;
    const int m = m0 + rg + it * RPP;
    if (aux && rg + it * RPP < BM && m < m_end) {
      if (fl.res()) {
        q[it].res = *(const uint4*)(a.res + (long)m * a.ldr + n);
        if (fl.res_bits()) q[it].rbits = a.res_bits[((long)m * a.ldr + n) >> 3];
      }
      if (fl.bnr()) {
        q[it].by = *(const uint4*)(a.bnr_y + (long)m * a.bnr_ld + n);
        if (fl.bnr() == 3) q[it].ybits = a.bnr_bits[((long)m * a.bnr_ld + n) >> 3];
      }
      if (fl.mask()) q[it].by = *(const uint4*)(a.mask + (long)m * a.ldm + n);
    }
  }
  __syncthreads();
  STAMP(4);
  float s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  if (n < a.Nstore) {
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = (fl.bias() && (n + e) < a.Cout) ? a.bias[n + e] : 0.f;
    const bool plain = !fl.bias() && !fl.res() && !fl.relu() && !fl.mask();
    float bmu[8], brs[8], bsc[8], bsh[8];          // fused BN-backward reduce: per-channel constants of the BatchNorm whose dz this is
    if (fl.bnr()) {
      load8(a.bnr_mean + n, bmu);
      load8(a.bnr_rstd + n, brs);
      if (fl.bnr() == 2) { load8(a.bnr_scale + n, bsc); load8(a.bnr_shift + n, bsh); }
    }
    auto unpack = [](const uint4& qq, float* v) {
      v[0] = __uint_as_float(qq.x << 16); v[1] = __uint_as_float(qq.x & 0xffff0000u);
      v[2] = __uint_as_float(qq.y << 16); v[3] = __uint_as_float(qq.y & 0xffff0000u);
      v[4] = __uint_as_float(qq.z << 16); v[5] = __uint_as_float(qq.z & 0xffff0000u);
      v[6] = __uint_as_float(qq.w << 16); v[7] = __uint_as_float(qq.w & 0xffff0000u);
    };
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int r = rg + it * RPP;
      const int m = m0 + r;
      if (r >= BM || m >= m_end) break;
      const Aux cur = q[it];
      const uint2 lo = *(const uint2*)(sC + r * CP + vcol * 2);
      const uint2 hi = *(const uint2*)(sC + r * CP + vcol * 2 + 8);
      const uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
      if (plain && !fl.stats() && !fl.bnr()) {
        st_out16(a.y + (long)m * a.ldy + n, o);
        continue;
      }
      float v[8];
      unpack(o, v);
      if (fl.stats()) {                               // forward: statistics of the stored value, before bias / residual / ReLU
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      }
      if (plain) {
        if (!(FBN && a.fbn_mode)) st_out16(a.y + (long)m * a.ldy + n, o);      // fused BatchNorm: the rows go out AFTER the tile sums are published (forward) / never (backward: dy instead)
      } else {
        if (fl.bias()) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bias8[e];
        }
        if (fl.res()) {
          float rv[8];
          unpack(cur.res, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e)          // (bit e as an all-ones / all-zeros word: v_bfe_i32 + and instead of test + compare + select)
            v[e] += __uint_as_float(__float_as_uint(rv[e]) & (unsigned)__builtin_amdgcn_sbfe((int)cur.rbits, e, 1));
        }
        if (fl.relu()) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (fl.mask()) {
          float mv[8];
          unpack(cur.by, mv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
        }
        st_out8(a.y + (long)m * a.ldy + n, v);
      }
      if (fl.bnr()) {
        // backward: S1 = sum g, S2 = sum g * xhat on the value as stored (bf16), masked like the backward masks it
        if (!plain) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
        }
        float yv[8];
        unpack(cur.by, yv);
        if (fl.bnr() == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (yv[e] * bsc[e] + bsh[e]) > 0.f ? v[e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = __uint_as_float(__float_as_uint(v[e]) & (unsigned)__builtin_amdgcn_sbfe((int)cur.ybits, e, 1));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * ((yv[e] - bmu[e]) * brs[e]); }
      }
    }
  }
  STAMP(5);
  if (fl.stats() || fl.bnr()) {
    // combine the RPP row groups in fixed order: sR[rg][2][BN] floats behind the tile
    float* sR = (float*)(smem + BM * CP);
    {
      // LDS stores as inline asm: in a kernel that uses LDS-DMA the compiler drains vmcnt to 0 in front of every C++ LDS store (it cannot tell an
      // outstanding global store from a global_load_lds that may still write LDS) -- here that would be a wait for the acknowledgement of every
      // row store of the tile (2-3 us).  No LDS-DMA is in flight in the epilogue (drained before the tile went to LDS).
      typedef float f4_t __attribute__((ext_vector_type(4)));
      const unsigned ad = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(sR + (rg * 2) * BN + vcol);
      const f4_t a0 = {s1[0], s1[1], s1[2], s1[3]}, a1 = {s1[4], s1[5], s1[6], s1[7]};
      const f4_t b0 = {s2[0], s2[1], s2[2], s2[3]}, b1 = {s2[4], s2[5], s2[6], s2[7]};
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16" :: "v"(ad), "v"(a0), "v"(a1) : "memory");
      asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16" :: "v"(ad + (unsigned)(BN * 4)), "v"(b0), "v"(b1) : "memory");
    }
    [[maybe_unused]] unsigned* sTag = (unsigned*)(smem + BM * CP + RPP * 2 * BN * 4 + 32 * 8 * 3 * 8);      // behind sRed (below)
    if constexpr (FBN != 0) {
      if (a.fbn_mode && tid == 0) {          // tag of this launch's granules: generation + 1 (never 0: the buffers start zeroed)
        const unsigned cnt_s = (unsigned)(a.ntiles_m * a.ntiles_n - (int)(blockIdx.x & 7) + 7) >> 3;
        sTag[0] = (unsigned)(fbn_ticket / cnt_s) + 1u;
        sTag[1] = 0u;                        // set by an owner thread whose granule poll gave up: the owner finalizes NOTHING
        sTag[2] = 0u;                        // set by the constants poller when it gave up: the workgroup applies NOTHING
      }
    }
    lds_barrier();
    STAMP(6);
    if (tid < BN) {
      const int nn = n0 + tid;
      if (nn < a.Cout) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int q = 0; q < RPP; ++q) {
          t1 += sR[(q * 2 + 0) * BN + tid];
          t2 += sR[(q * 2 + 1) * BN + tid];
        }
        if (FBN && a.fbn_mode) {
          // fused BatchNorm: the tile sums go to the OWNER workgroups of this launch as 8-byte {value, tag} granules, one write-through
          // store each (MI355X_MICROARCH.md hand-off form R2: the reader polls the data itself -- no flag, no drain, no second trip)
          const int NJg = a.fbn_mode == 2 ? 3 : 2;
          const unsigned long long tg = (unsigned long long)sTag[0] << 32;
          unsigned long long* gs = a.fbn_slots + ((long)mt * NJg) * a.Cout + nn;
          st_gran(gs, tg | __float_as_uint(t1));
          st_gran(gs + a.Cout, tg | __float_as_uint(t2));
        } else if (fl.bnr()) {      // [m-tile][3][Cout]: S1, S2 and the (unused) second-BN row
          a.bnr_part[((long)mt * 3 + 0) * a.Cout + nn] = t1;
          a.bnr_part[((long)mt * 3 + 1) * a.Cout + nn] = t2;
          a.bnr_part[((long)mt * 3 + 2) * a.Cout + nn] = 0.f;
        } else {
          a.stats[((long)mt * 2 + 0) * a.Cout + nn] = t1;
          a.stats[((long)mt * 2 + 1) * a.Cout + nn] = t2;
          if (mt == 0)   // the caller sums ceil(M/128) slots; tiles of more than 128 rows leave the tail unused: zero it
            for (int sl = a.ntiles_m; sl < a.nblk128; ++sl) {
              a.stats[((long)sl * 2 + 0) * a.Cout + nn] = 0.f;
              a.stats[((long)sl * 2 + 1) * a.Cout + nn] = 0.f;
            }
        }
      }
    }
    if constexpr (FBN != 0) {
      if (a.fbn_mode) {
        // ---------------- fused train-mode BatchNorm: tile sums -> owners -> constants -> every workgroup, all as polled granules
        constexpr int RED_OFF = BM * CP + RPP * 2 * BN * 4;       // behind the tile and the row-group sums
        double* sRed = (double*)(smem + RED_OFF);                 // [32][8][3]
        float* sCst = (float*)(smem + RED_OFF + 32 * 8 * 3 * 8 + 16);     // [2][BN]: this tile's channels' constants
        const int G = (a.Cout + 7) >> 3;                          // owner workgroups: 8 channels each, like bn_finalize_kernel's blocks
        const bool owner = tile < G;
        const int NJ = a.fbn_mode == 2 ? 3 : 2;
        const unsigned tag = sTag[0];
        STAMP(1);
        if (a.fbn_mode == 1 && n < a.Nstore) {
          // forward: NOW stream the rows of y out (the statistics loop above only summed them) -- the owners reduce meanwhile
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int r = rg + it * RPP;
            const int m = m0 + r;
            if (r >= BM || m >= m_end) break;
            const uint2 lo = *(const uint2*)(sC + r * CP + vcol * 2);
            const uint2 hi = *(const uint2*)(sC + r * CP + vcol * 2 + 8);
            st_out16(a.y + (long)m * a.ldy + n, make_uint4(lo.x, lo.y, hi.x, hi.y));
          }
        }
        STAMP(2);
        if (owner) {
          // the same sums in the same order as part_colsum8 (bn_pool.hip): thread (r, cl) adds slots r, r + 32, ... of channel 8 * tile + cl
          // in double, then r = 0 adds the 32 partials in order -> bitwise the constants of simt_bn_finalize / simt_bn_bwd.  (The forward's
          // zero tail slots ntiles_m ... ceil(M / 128) add nothing.)  All granules of a thread are requested at once and re-requested until
          // every tag is this launch's.
          const int nblk = a.ntiles_m;
          const int cl = tid & 7, r = tid >> 3;
          const int c = tile * 8 + cl;
          if (tid < 256) {
            constexpr int MAXS = 12;                               // slots per thread: ntiles_m <= 384 (simt_conv_fbn_ok)
            double sj[3] = {0.0, 0.0, 0.0};
            if (c < a.Cout) {
              auto run = [&](auto NJC) {
                constexpr int NJc = decltype(NJC)::value == 3 ? 2 : 2;      // the third row of the backward is identically zero: not sent
                float v[MAXS][NJc];
                const unsigned off0 = (unsigned)(r * decltype(NJC)::value) * (unsigned)a.Cout + (unsigned)c;
                const unsigned ostep = 32u * decltype(NJC)::value * (unsigned)a.Cout;
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                unsigned spins = 0;
                bool ok;
                do {
                  ok = true;
#pragma unroll
                  for (int u = 0; u < MAXS; ++u)
#pragma unroll
                    for (int j = 0; j < NJc; ++j) {
                      if (r + 32 * u < nblk) {
                        const unsigned long long gq = ld_gran(a.fbn_slots + (off0 + u * ostep + j * (unsigned)a.Cout));
                        ok = ok && (unsigned)(gq >> 32) == tag;
                        v[u][j] = __uint_as_float((unsigned)gq);
                      } else {
                        v[u][j] = 0.f;
                      }
                    }
                  if (!ok) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 0u && fbn_poll_gives_up(a.fbn_err, t0)) break;      // ~2 s, or another poller's bail-out
                  }
                } while (!ok);
                if (!ok) sTag[1] = 1u;       // (benign race: every writer stores 1; read behind the barrier below)
#pragma unroll
                for (int u = 0; u < MAXS; ++u)
#pragma unroll
                  for (int j = 0; j < NJc; ++j)
                    if (r + 32 * u < nblk) sj[j] += (double)v[u][j];
              };
              if (NJ == 3) run(std::integral_constant<int, 3>{}); else run(std::integral_constant<int, 2>{});
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) sRed[(r * 8 + cl) * 3 + j] = sj[j];       // (constant trip counts: a run-time index puts the array in scratch)
          }
          lds_barrier();
          // a poll of this owner gave up (time-out, or another poller's sticky error word): partial sums -- publish no constants, touch no
          // statistics (mean / rstd / scale / shift / running statistics / coef / d gamma / d beta keep their previous values); the constants
          // pollers of every workgroup then leave through the error word
          if (tid < 8 && c < a.Cout && sTag[1] == 0u) {
            double t[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int j = 0; j < 3; ++j)
              for (int q = 0; q < 32; ++q) t[j] += sRed[(q * 8 + cl) * 3 + j];   // (row 2 is zero)
            const long count = a.M;
            const unsigned long long tg = (unsigned long long)tag << 32;
            if (a.fbn_mode == 1) {                                 // bn_finalize_kernel
              const double mean = t[0] / (double)count;
              double var = t[1] / (double)count - mean * mean;
              if (var < 0.0) var = 0.0;
              const float rstd = (float)(1.0 / sqrt(var + (double)a.fbn_eps));
              const float g = a.fbn_gamma ? a.fbn_gamma[c] : 1.f, bt = a.fbn_beta ? a.fbn_beta[c] : 0.f;
              const float sc = g * rstd;
              const float sh = bt - (float)mean * sc;
              st_gran(a.fbn_cgran + c, tg | __float_as_uint(sc));
              st_gran(a.fbn_cgran + a.Cout + c, tg | __float_as_uint(sh));
              a.fbn_mean[c] = (float)mean;                         // (for the backward launches: ordinary stores)
              a.fbn_rstd[c] = rstd;
              a.fbn_scale[c] = sc;
              a.fbn_shift[c] = sh;
              if (a.fbn_rmean) {
                const double unb = count > 1 ? var * (double)count / (double)(count - 1) : var;
                a.fbn_rmean[c] = (1.f - a.fbn_momentum) * a.fbn_rmean[c] + a.fbn_momentum * (float)mean;
                a.fbn_rvar[c] = (1.f - a.fbn_momentum) * a.fbn_rvar[c] + a.fbn_momentum * (float)unb;
              }
            } else {                                               // bn_bwd_finalize_kernel
              const float c1 = (float)(t[0] / (double)count), c2 = (float)(t[1] / (double)count);
              st_gran(a.fbn_cgran + c, tg | __float_as_uint(c1));
              st_gran(a.fbn_cgran + a.Cout + c, tg | __float_as_uint(c2));
              a.fbn_coef[c] = c1;
              a.fbn_coef[a.Cout + c] = c2;
              a.fbn_coef[2 * a.Cout + c] = 0.f;
              if (a.fbn_dbeta) a.fbn_dbeta[c] = (float)t[0];        // (bn_bwd_finalize_kernel's d beta / d gamma: trainable affine)
              if (a.fbn_dgamma) a.fbn_dgamma[c] = (float)t[1];
            }
          }
        }
        // every workgroup: ONE wave polls the constants of this tile's BN channels (two granules per channel) into LDS
        if (tid < 64) {
          constexpr int PER = (2 * BN + 63) / 64;                  // granules per lane
          const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
          unsigned spins = 0;
          float cv[PER];
          bool ok;
          do {
            ok = true;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
              const int idx = lane + 64 * u;                       // [0, 2 * BN): row idx / BN (scale | shift  or  c1 | c2), channel n0 + idx % BN
              const int ch = n0 + (idx % BN);
              if (idx < 2 * BN && ch < a.Cout) {
                const unsigned long long gq = ld_gran(a.fbn_cgran + (idx / BN) * a.Cout + ch);
                ok = ok && (unsigned)(gq >> 32) == tag;
                cv[u] = __uint_as_float((unsigned)gq);
              } else {
                cv[u] = 0.f;
              }
            }
            ok = __all(ok);
            if (!ok) {
              __builtin_amdgcn_s_sleep(1);
              if ((++spins & 255u) == 0u && __any(fbn_poll_gives_up(a.fbn_err, t0))) break;
            }
          } while (!ok);
          if (!ok && lane == 0) sTag[2] = 1u;
#pragma unroll
          for (int u = 0; u < PER; ++u)
            if (lane + 64 * u < 2 * BN) sCst[lane + 64 * u] = cv[u];
        }
        lds_barrier();
        if (sTag[2] != 0u) return;                                 // gave up (workgroup-uniform): `out` is NOT written by this workgroup
        STAMP(7);                                                  // (constants published and seen)
        // ---- apply: the tile is still in LDS (bf16, as stored)
        if (n < a.Nstore) {
          auto unpack8 = [](const uint4& qq, float* v) {
            v[0] = __uint_as_float(qq.x << 16); v[1] = __uint_as_float(qq.x & 0xffff0000u);
            v[2] = __uint_as_float(qq.y << 16); v[3] = __uint_as_float(qq.y & 0xffff0000u);
            v[4] = __uint_as_float(qq.z << 16); v[5] = __uint_as_float(qq.z & 0xffff0000u);
            v[6] = __uint_as_float(qq.w << 16); v[7] = __uint_as_float(qq.w & 0xffff0000u);
          };
          float sc[8], sh[8], mu[8], rs[8], c1[8], c2[8];
          if (a.fbn_mode == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = sCst[vcol + e]; sh[e] = sCst[BN + vcol + e]; }
          } else {
            load8(a.bnr_scale + n, sc);
            load8(a.bnr_shift + n, sh);
            load8(a.bnr_mean + n, mu);
            load8(a.bnr_rstd + n, rs);
#pragma unroll
            for (int e = 0; e < 8; ++e) { c1[e] = sCst[vcol + e]; c2[e] = sCst[BN + vcol + e]; }
          }
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const int r = rg + it * RPP;
            const int m = m0 + r;
            if (r >= BM || m >= m_end) break;
            const uint2 lo = *(const uint2*)(sC + r * CP + vcol * 2);
            const uint2 hi = *(const uint2*)(sC + r * CP + vcol * 2 + 8);
            float v[8];
            unpack8(make_uint4(lo.x, lo.y, hi.x, hi.y), v);
            if (a.fbn_mode == 1) {                                 // bn_apply_kernel: relu(y * scale + shift)
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[e] + sh[e];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
              st_out8(a.fbn_out + (long)m * a.fbn_ldo + n, v);
            } else {                                               // bn_bwd_apply_kernel, mask_mode 2
              float yv[8], o[8];
              unpack8(q[it].by, yv);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (yv[e] * sc[e] + sh[e]) > 0.f ? v[e] : 0.f;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = sc[e] * (v[e] - c1[e] - ((yv[e] - mu[e]) * rs[e]) * c2[e]);
              st_out8(a.fbn_out + (long)m * a.fbn_ldo + n, o);
            }
          }
        }
        STAMP(0);                                                  // (end of the fused tail; slot 0 = start is overwritten: read deltas)
      }
    }
  }
}
