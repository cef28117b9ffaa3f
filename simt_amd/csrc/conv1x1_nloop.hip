// EXPERIMENT (opt-in: SIMT_CONV_NLOOP=1; the product path is conv_igemm2.hip's short-K variant).
// 1x1 convolution with a SHORT reduction and a WIDE output (Cin <= 256, Cout >= 256; bf16 -> bf16): the Bottleneck's
// conv3 (model/deeplab_multi.py:73, 256 -> 1024), the dgrad of its conv1 (1024 <- 256) and the layer1 / layer2 analogues.
//
// These GEMMs are bound by the OUTPUT stream (M x Cout x 2 bytes), not by MFMA: at M = 37 636, K = 256, N = 1024 the
// algorithmic traffic is 96 MB = 19 us of HBM time, while the tiled kernel (128 x 128 tiles, two workgroups per CU) needs
// 42-49 us isolated and 62 us inside the training step: every 128-column tile re-stages its 64 KB pixel panel and pays its
// own prologue and epilogue around 4 K-steps.  Here ONE workgroup per CU owns a panel of up to 160 pixels for the whole launch:
//   * the pixel panel (rows x K) is gathered into LDS once and stays resident (80 KB);
//   * the workgroup loops over all 128-column tiles of the output; the weight tiles (128 x 64 channels = 16 KB per stage) stream
//     through a three-slot global_load_lds ring filled by FOUR LOADER WAVES, so that the eight compute waves never wait on
//     vmcnt: stores and loads share one per-wave counter, and a wave that does both cannot wait for a ring stage without also
//     draining its output stores (measured: that serialised the 77 MB output stream with the MFMA loop);
//   * the epilogue stages one wave row (80 pixels x 128 channels) at a time through 21 KB of LDS and streams it out as whole
//     256-byte rows with bias / masked residual / ReLU / BatchNorm statistics / fused BatchNorm-backward reduce on the way --
//     register-direct 16-byte stores (64-byte runs) reached only ~2.1 TB/s against ~3.2 TB/s for 256-byte rows.
// Measured (scratch/convbench.py, 256 -> 1024): 36.8 us without statistics (tiled kernel 41.6), 50.3 us with (48.5); ablations:
// no stores 22.6 us, no MFMA 27.0 us.  Inside the training step, where operands are not cache-warm, the 68 launches of this
// shape average 85 us against 62 us for the tiled kernel and the step drops from 123.3 to 118.4 images/s -- hence opt-in.
#include "common.h"

struct NLoopArgs {
  const char* x;
  const char* w;
  bf16_t* y;
  const float* bias;
  const bf16_t* res;
  const unsigned char* res_bits;
  const bf16_t* bnr_y;                 // fused first pass of the BatchNorm backward (simt_conv_desc.bnr_*)
  const float *bnr_mean, *bnr_rstd, *bnr_scale, *bnr_shift;
  const unsigned char* bnr_bits;
  float* bnr_part;
  int bnr_mode, bnr_ld;
  float* stats;
  const char* zero;
  int H, W, Ho, Wo, Cout, Nstore, ldy, ldr, stride, relu, M;
  int pix_bytes, wrow_bytes;
  int ntiles_n, ntiles_m, rows, nblk128;
};

// MODE 0 = product; 1 = no MFMA, 2 = no epilogue stores, 3 = no weight loads (timing ablations, env SIMT_NLOOP_MODE; their
// outputs are meaningless)
template <int KC, int MODE>   // K / 64
__global__ __launch_bounds__(768) void conv1x1_nloop_kernel(NLoopArgs a) {
  constexpr int NT = 512, NLOAD = 256, BM = 160, BN = 128, NSTG = 3;   // 8 compute waves + 4 loader waves
  constexpr int WN = 4, TM = 5, TN = 2;           // compute waves: 2 (pixels) x 4 (couts); wave tile 80 x 32
  constexpr int HM = BM / 2;                      // rows per epilogue half (= one wave row)
  constexpr int A_CHUNK = BM * 128;               // one 64-channel slice of the pixel panel
  constexpr int B_STAGE = BN * 128;
  constexpr int A_IT = 3, B_IT = BN * 8 / NLOAD;  // 16-B pieces per thread: panel 160*8/512 = 2.5 (third pass: waves 0-3); weights 4
  constexpr int CP = BN * 2 + 8;                  // staging tile pitch (bf16 row + 8 B pad)
  constexpr int VPR = BN / 8, RPP = NT / VPR;     // 16-B vectors per output row, rows per streaming pass
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                                // [KC][BM][128 B]   resident pixel panel
  char* sB = smem + KC * A_CHUNK;                 // [NSTG][BN][128 B] weight ring
  char* sC = sB + NSTG * B_STAGE;                 // [HM][CP]          output staging, one wave row at a time
  float* sR = (float*)(sC + HM * CP);             // [8 waves][2][BN]  per-wave column sums (statistics / BN-backward reduce)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mt = xcd_remap(blockIdx.x, a.ntiles_m);
  const int m0 = mt * a.rows;
  const int m_end = min(a.M, m0 + a.rows);
  const int nstages = a.ntiles_n * KC;
  const bool sums = a.stats != nullptr || a.bnr_mode != 0;

  if (wave >= 8) {
    // ================= loader waves: the weight ring.  Their vmcnt never sees a store, the compute waves' never sees a load.
    // Stage g = (column tile g / KC, K slice g % KC) goes to slot g % 3; stage g+2 is issued behind the barrier of step g
    // (its slot was read during step g-1).
    const int ltid = tid - NT;
    const int lc = (ltid & 7) ^ (((ltid >> 3) >> 1) & 7);
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) b_off[i] = (unsigned)(i * (NLOAD / 8) + (ltid >> 3)) * (unsigned)a.wrow_bytes + (unsigned)(lc * 16);
    int ld_nt = 0, ld_kc = 0, ld_slot = 0;
    auto issue = [&]() {
      char* dst = sB + ld_slot * B_STAGE;
      const unsigned base = (unsigned)(ld_nt * BN) * (unsigned)a.wrow_bytes + (unsigned)(ld_kc * 128);
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(a.w + (base + b_off[i])), LPTR(dst + (i * NLOAD + (wave - 8) * 64) * 16), 16, 0, 0);
      if (++ld_kc == KC) { ld_kc = 0; ++ld_nt; }
      ld_slot = (ld_slot + 1 == NSTG) ? 0 : ld_slot + 1;
    };
    if (MODE != 3) { issue(); if (nstages > 1) issue(); }
    int g = 0;
    for (int nt = 0; nt < a.ntiles_n; ++nt) {
      for (int kc = 0; kc < KC; ++kc, ++g) {
        if (g + 1 < nstages) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // stage g landed, stage g+1 (B_IT = 4 pieces) may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g + 2 < nstages && MODE != 3) issue();
      }
      for (int q = 0; q < (sums ? 6 : 4); ++q) __builtin_amdgcn_s_barrier();    // mirror the compute waves' epilogue barriers
    }
    return;
  }
  static_assert(B_IT == 4, "the loader's counted wait assumes four pieces per stage");

  // ================= compute waves
  // ---- resident pixel panel: piece q = i*NT + tid -> row q>>3, 16-B position q&7 (XOR-swizzled on the source side)
  const int c_pos = tid & 7;
  const int cg = c_pos ^ (((tid >> 3) >> 1) & 7);
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    if (i == A_IT - 1 && wave >= 4) break;
    const int row = i * (NT / 8) + (tid >> 3);
    const int m = m0 + row;
    const char* src = a.zero + cg * 16;
    if (m < m_end) {
      const int hw = a.Ho * a.Wo;
      const int b = m / hw;
      const int r = m - b * hw;
      const int oy = r / a.Wo;
      const int ox = r - oy * a.Wo;
      src = a.x + (unsigned)((b * a.H + oy * a.stride) * a.W + ox * a.stride) * (unsigned)a.pix_bytes + (unsigned)(cg * 16);
    }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
      __builtin_amdgcn_global_load_lds(GPTR(src + (m < m_end ? kc * 128 : 0)), LPTR(sA + kc * A_CHUNK + (i * NT + wave * 64) * 16), 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's pieces of the panel (the first barrier publishes them)

  const int sw = (lane >> 1) & 7;
  const int frag_row_off = (lane & 15) * 128;
  const int kq = lane >> 4;
  const int xbase = (wm * TM * 16) * 128 + frag_row_off;
  const int wbase = (wn * TN * 16) * 128 + frag_row_off;
  // row-streaming role of this thread in the epilogue: 16-byte vector vcol of rows rg, rg + RPP, ...
  const int vcol = (tid % VPR) * 8;
  const int rg = tid / VPR;

  int g = 0, slot = 0;
  for (int nt = 0; nt < a.ntiles_n; ++nt) {
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc, ++g) {
      __builtin_amdgcn_s_barrier();                // the loader waves arrive only after their pieces of stage g have landed
      asm volatile("" ::: "memory");
      const char* px = sA + kc * A_CHUNK + xbase;
      const char* pw = sB + slot * B_STAGE + wbase;
      slot = (slot + 1 == NSTG) ? 0 : slot + 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 xf[TM], wf[TN];
        const int coff = ((4 * s + kq) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[i] = *(const bf16x8*)(px + i * 16 * 128 + coff);
#pragma unroll
        for (int j = 0; j < TN; ++j) wf[j] = *(const bf16x8*)(pw + j * 16 * 128 + coff);
        if (MODE != 1) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
        }
      }
    }
    // ---------------- epilogue of column tile nt: one wave row (80 pixels) at a time through the staging tile, streamed out
    // as whole 256-byte rows (16 B per lane) with bias / masked residual / ReLU applied on the way; per-channel sums for the
    // BatchNorm statistics or the fused BatchNorm-backward reduce accumulated by the same pass.
    const int n0 = nt * BN;
    const int n = n0 + vcol;
    float s1[8], s2[8], bias8[8], bmu[8], brs[8], bsc[8], bsh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; bias8[e] = (a.bias && (n + e) < a.Cout) ? a.bias[n + e] : 0.f; }
    if (a.bnr_mode && n < a.Nstore) {
      load8(a.bnr_mean + n, bmu);
      load8(a.bnr_rstd + n, brs);
      if (a.bnr_mode == 2) { load8(a.bnr_scale + n, bsc); load8(a.bnr_shift + n, bsh); }
    }
    const bool plain = !a.bias && !a.res && !a.relu;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wm == half) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int r = i * 16 + (lane & 15);
            const int c = wn * TN * 16 + j * 16 + (lane >> 4) * 4;
            uint2 pk;
            pk.x = (uint32_t)f2bf(acc[j][i][0]) | ((uint32_t)f2bf(acc[j][i][1]) << 16);
            pk.y = (uint32_t)f2bf(acc[j][i][2]) | ((uint32_t)f2bf(acc[j][i][3]) << 16);
            *(uint2*)(sC + r * CP + c * 2) = pk;
          }
      }
      __syncthreads();
      if (n < a.Nstore && MODE != 2) {
        for (int r = rg; r < HM; r += RPP) {
          const int m = m0 + half * HM + r;
          if (m >= m_end) break;
          const uint2 lo = *(const uint2*)(sC + r * CP + vcol * 2);
          const uint2 hi = *(const uint2*)(sC + r * CP + vcol * 2 + 8);
          uint4 o = make_uint4(lo.x, lo.y, hi.x, hi.y);
          if (plain && !sums) { *(uint4*)(a.y + (long)m * a.ldy + n) = o; continue; }
          float v[8];
          v[0] = __uint_as_float(o.x << 16); v[1] = __uint_as_float(o.x & 0xffff0000u);
          v[2] = __uint_as_float(o.y << 16); v[3] = __uint_as_float(o.y & 0xffff0000u);
          v[4] = __uint_as_float(o.z << 16); v[5] = __uint_as_float(o.z & 0xffff0000u);
          v[6] = __uint_as_float(o.w << 16); v[7] = __uint_as_float(o.w & 0xffff0000u);
          if (a.stats) {                             // forward: statistics of the stored value, before bias / residual / ReLU
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
              load8(a.res + (long)m * a.ldr + n, rv);
              if (a.res_bits) {
                const unsigned b = a.res_bits[((long)m * a.ldr + n) >> 3];
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = ((b >> e) & 1u) ? rv[e] : 0.f;
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (a.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            store8(a.y + (long)m * a.ldy + n, v);
          }
          if (a.bnr_mode) {                          // backward: S1 = sum g, S2 = sum g * xhat on the value as stored
            if (!plain) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = bf2f(f2bf(v[e]));
            }
            float yv[8];
            load8(a.bnr_y + (long)m * a.bnr_ld + n, yv);
            if (a.bnr_mode == 2) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (yv[e] * bsc[e] + bsh[e]) > 0.f ? v[e] : 0.f;
            } else {
              const unsigned b = a.bnr_bits[((long)m * a.bnr_ld + n) >> 3];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = ((b >> e) & 1u) ? v[e] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * ((yv[e] - bmu[e]) * brs[e]); }
          }
        }
      }
      __syncthreads();                               // the staging tile is free again
    }
    if (sums) {
      // a wave holds 4 row groups per column vector (lanes l, l+16, l+32, l+48): fold them, then the 8 waves in fixed order
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], 16, 64); s1[e] += __shfl_xor(s1[e], 32, 64);
        s2[e] += __shfl_xor(s2[e], 16, 64); s2[e] += __shfl_xor(s2[e], 32, 64);
      }
      if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          sR[(wave * 2 + 0) * BN + vcol + e] = s1[e];
          sR[(wave * 2 + 1) * BN + vcol + e] = s2[e];
        }
      }
      __syncthreads();
      if (tid < BN) {
        const int nn = n0 + tid;
        if (nn < a.Cout) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) { t1 += sR[(q * 2 + 0) * BN + tid]; t2 += sR[(q * 2 + 1) * BN + tid]; }
          if (a.bnr_mode) {
            a.bnr_part[((long)mt * 3 + 0) * a.Cout + nn] = t1;
            a.bnr_part[((long)mt * 3 + 1) * a.Cout + nn] = t2;
            a.bnr_part[((long)mt * 3 + 2) * a.Cout + nn] = 0.f;
          } else {
            a.stats[((long)mt * 2 + 0) * a.Cout + nn] = t1;
            a.stats[((long)mt * 2 + 1) * a.Cout + nn] = t2;
            // the caller sums ceil(M/128) slots; panels of more than 128 rows leave the tail unused: panel mt zeroes slot
            // ntiles_m + mt (ntiles_m <= nblk128 <= 2 * ntiles_m because 128 <= rows <= 160)
            const int sl = a.ntiles_m + mt;
            if (sl < a.nblk128) {
              a.stats[((long)sl * 2 + 0) * a.Cout + nn] = 0.f;
              a.stats[((long)sl * 2 + 1) * a.Cout + nn] = 0.f;
            }
          }
        }
      }
      __syncthreads();                               // sR is free again
    }
  }
}

template <int KC, int MODE>
static int launch_nloop_m(const NLoopArgs& k, hipStream_t st) {
  const size_t lds = (size_t)KC * 160 * 128 + 3 * 128 * 128 + 80 * (128 * 2 + 8) + 8 * 2 * 128 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv1x1_nloop_kernel<KC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv1x1_nloop_kernel<KC, MODE>), dim3(k.ntiles_m), dim3(768), lds, st, k);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
#include <stdlib.h>
template <int KC>
static int launch_nloop(const NLoopArgs& k, hipStream_t st) {
  static int mode = -1;
  if (mode < 0) { const char* e = getenv("SIMT_NLOOP_MODE"); mode = e ? atoi(e) : 0; }
  if (KC == 4) {
    if (mode == 1) return launch_nloop_m<KC, 1>(k, st);
    if (mode == 2) return launch_nloop_m<KC, 2>(k, st);
    if (mode == 3) return launch_nloop_m<KC, 3>(k, st);
  }
  return launch_nloop_m<KC, 0>(k, st);
}

// Does this descriptor take the resident-panel kernel?  (1 tap at offset (0,0), Cin in {64, 128, 256}, wide bf16 output)
bool simt_conv_nloop_eligible(const simt_conv_desc* d) {
  if (d->dtype_in != SIMT_BF16 || d->dtype_out != SIMT_BF16) return false;
  if (d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0 || d->mask) return false;
  if (!(d->Cin == 64 || d->Cin == 128 || d->Cin == 256)) return false;
  if (d->Npad % 128 != 0 || d->Cout < 256 || d->Nstore % 8 != 0 || d->ldy % 8 != 0) return false;
  if (d->res && d->ldr % 8 != 0) return false;
  return true;
}

// Pixel rows per panel: the fewest 256-CU rounds, then the smallest panel.
static int nloop_rows(int M) {
  long best = -1;
  int rows = 160;
  for (int r = 128; r <= 160; r += 4) {   // >= 128: the caller allocates ceil(M/128) statistics slots
    const long tiles = (M + r - 1) / r;
    const long rounds = (tiles + 255) / 256;
    const long cost = rounds * 1000 + r;          // rounds dominate; among equal rounds prefer the smaller panel
    if (best < 0 || cost < best) { best = cost; rows = r; }
  }
  return rows;
}

int simt_conv_nloop_mtiles(const simt_conv_desc* d) {
  const int M = d->B * d->Ho * d->Wo;
  const int rows = nloop_rows(M);
  return (M + rows - 1) / rows;
}

int simt_conv_fprop_bf16_nloop(const simt_conv_desc* d, simt_stream_t stream) {
  NLoopArgs k;
  k.x = (const char*)d->x; k.w = (const char*)d->w; k.y = (bf16_t*)d->y; k.bias = d->bias; k.res = (const bf16_t*)d->res;
  k.res_bits = d->res_bits; k.stats = d->stats; k.zero = (const char*)simt_zero_page();
  k.bnr_mode = d->bnr_mode; k.bnr_ld = d->bnr_ld; k.bnr_y = (const bf16_t*)d->bnr_y; k.bnr_mean = d->bnr_mean; k.bnr_rstd = d->bnr_rstd;
  k.bnr_scale = d->bnr_scale; k.bnr_shift = d->bnr_shift; k.bnr_bits = d->bnr_bits; k.bnr_part = d->bnr_part;
  if (k.bnr_mode) {
    SIMT_CHECK(!d->stats && !d->relu && d->bnr_y && d->bnr_mean && d->bnr_rstd && d->bnr_part && d->bnr_ld % 8 == 0);
    SIMT_CHECK(d->bnr_mode == 2 ? (d->bnr_scale && d->bnr_shift) : (d->bnr_mode == 3 && d->bnr_bits));
  }
  k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout; k.Nstore = d->Nstore; k.ldy = d->ldy; k.ldr = d->ldr;
  k.stride = d->stride; k.relu = d->relu; k.M = d->B * d->Ho * d->Wo;
  k.pix_bytes = d->Cin * 2;
  k.wrow_bytes = d->Cin * 2;
  k.ntiles_n = d->Npad / 128;
  k.rows = nloop_rows(k.M);
  k.ntiles_m = (k.M + k.rows - 1) / k.rows;
  k.nblk128 = (k.M + 127) / 128;
  SIMT_CHECK(k.ntiles_m <= k.nblk128 || !d->stats);
  SIMT_CHECK((long)d->B * d->H * d->W * d->Cin * 2 < (1l << 32) && (long)d->Npad * k.wrow_bytes < (1l << 32));
  hipStream_t st = (hipStream_t)stream;
  if (d->Cin == 256) return launch_nloop<4>(k, st);
  if (d->Cin == 128) return launch_nloop<2>(k, st);
  return launch_nloop<1>(k, st);
}
