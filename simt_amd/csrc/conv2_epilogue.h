// Epilogue shared by the bf16 throughput conv kernels (conv_igemm2.hip, conv_igemm3.hip): accumulators -> bf16 tile in LDS -> whole rows
// streamed out with bias / residual / ReLU / mask applied on the way, BatchNorm statistics or the fused BatchNorm-backward reduce summed per
// row group in fixed order.  `compute_wave`: this wave holds accumulators (conv_igemm3's loader waves do not; they take part in the row
// streaming).  acc[j][i][e]: cout = n0 + (wn*TN + j)*16 + (lane>>4)*4 + e ; pixel row = (wm*TM + i)*16 + (lane&15).
#pragma once
#include "conv2_common.h"

template <int BN, int BM, int NT, int TN, int TM>
__device__ __forceinline__ void conv2_epilogue(const Conv2KArgs& a, char* smem, f32x4 (&acc)[TN][TM], bool compute_wave, int wm, int wn,
                                               int tid, int lane, int m0, int n0, int m_end, int mt) {
  constexpr int CP = BN * 2 + 8;                   // epilogue tile pitch in bytes (bf16 row + 8 B pad)
  // ---------------- epilogue ----------------
  // acc[j][i][e]: cout = n0 + wn*TN*16 + j*16 + (lane>>4)*4 + e ; pixel row = wm*TM*16 + i*16 + (lane&15)
  if (a.out_f32) {
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
        if (m < m_end && c < a.Nstore) *(f32x4*)(yf + (long)m * a.ldy + c) = acc[j][i];
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
  struct Aux { uint4 res, by; unsigned rbits, ybits; };   // by: saved activation (bnr) or ReLU mask operand (VGG): exclusive
  const bool aux = (a.res || a.bnr_mode || a.mask) && n < a.Nstore;
  Aux q[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    q[it].res = q[it].by = make_uint4(0u, 0u, 0u, 0u);
    q[it].rbits = q[it].ybits = 0xffu;
    const int m = m0 + rg + it * RPP;
    if (aux && rg + it * RPP < BM && m < m_end) {
      if (a.res) {
        q[it].res = *(const uint4*)(a.res + (long)m * a.ldr + n);
        if (a.res_bits) q[it].rbits = a.res_bits[((long)m * a.ldr + n) >> 3];
      }
      if (a.bnr_mode) {
        q[it].by = *(const uint4*)(a.bnr_y + (long)m * a.bnr_ld + n);
        if (a.bnr_mode == 3) q[it].ybits = a.bnr_bits[((long)m * a.bnr_ld + n) >> 3];
      }
      if (a.mask) q[it].by = *(const uint4*)(a.mask + (long)m * a.ldm + n);
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
    for (int e = 0; e < 8; ++e) bias8[e] = (a.bias && (n + e) < a.Cout) ? a.bias[n + e] : 0.f;
    const bool plain = !a.bias && !a.res && !a.relu && !a.mask;
    float bmu[8], brs[8], bsc[8], bsh[8];          // fused BN-backward reduce: per-channel constants of the BatchNorm whose dz this is
    if (a.bnr_mode) {
      load8(a.bnr_mean + n, bmu);
      load8(a.bnr_rstd + n, brs);
      if (a.bnr_mode == 2) { load8(a.bnr_scale + n, bsc); load8(a.bnr_shift + n, bsh); }
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
      if (plain && !a.stats && !a.bnr_mode) {
        *(uint4*)(a.y + (long)m * a.ldy + n) = o;
        continue;
      }
      float v[8];
      unpack(o, v);
      if (a.stats) {                               // forward: statistics of the stored value, before bias / residual / ReLU
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
      }
      if (plain) {
        *(uint4*)(a.y + (long)m * a.ldy + n) = o;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bias8[e];
        if (a.res) {
          float rv[8];
          unpack(cur.res, rv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += ((cur.rbits >> e) & 1u) ? rv[e] : 0.f;
        }
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        if (a.mask) {
          float mv[8];
          unpack(cur.by, mv);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = mv[e] > 0.f ? v[e] : 0.f;
        }
        store8(a.y + (long)m * a.ldy + n, v);
      }
      if (a.bnr_mode) {
        // backward: S1 = sum g, S2 = sum g * xhat on the value as stored (bf16), masked like the backward masks it
        if (!plain) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
        }
        float yv[8];
        unpack(cur.by, yv);
        if (a.bnr_mode == 2) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (yv[e] * bsc[e] + bsh[e]) > 0.f ? v[e] : 0.f;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ((cur.ybits >> e) & 1u) ? v[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * ((yv[e] - bmu[e]) * brs[e]); }
      }
    }
  }
  STAMP(5);
  if (a.stats || a.bnr_mode) {
    // combine the RPP row groups in fixed order: sR[rg][2][BN] floats behind the tile
    float* sR = (float*)(smem + BM * CP);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sR[(rg * 2 + 0) * BN + vcol + e] = s1[e];
      sR[(rg * 2 + 1) * BN + vcol + e] = s2[e];
    }
    __syncthreads();
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
        if (a.bnr_mode) {      // [m-tile][3][Cout]: S1, S2 and the (unused) second-BN row
          a.bnr_part[((long)mt * 3 + 0) * a.Cout + nn] = t1;
          a.bnr_part[((long)mt * 3 + 1) * a.Cout + nn] = t2;
          a.bnr_part[((long)mt * 3 + 2) * a.Cout + nn] = 0.f;
          return;
        }
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
}
