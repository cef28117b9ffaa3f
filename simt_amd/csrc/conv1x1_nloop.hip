// 1x1 convolution with a SHORT reduction and a WIDE output (Cin <= 256, Cout >= 256; bf16 -> bf16): the Bottleneck's
// conv3 (model/deeplab_multi.py:73, 256 -> 1024), the dgrad of its conv1 (1024 <- 256) and the layer1 / layer2 analogues.
//
// These GEMMs are bound by the OUTPUT stream (M x Cout x 2 bytes), not by MFMA: at M = 37 636, K = 256, N = 1024 the
// algorithmic traffic is 96 MB = 19 us of HBM time, while the tiled kernel (conv_igemm2.hip, 128 x 128 tiles, two
// workgroups per CU) needed 49-62 us: every 128-column tile re-staged its 64 KB pixel panel and paid its own prologue
// and epilogue around 4 K-steps.  Here ONE workgroup per CU owns a panel of up to 160 pixels for the whole launch:
//   * the pixel panel (rows x K) is gathered into LDS once and stays resident;
//   * the workgroup loops over all 256-column tiles of the output; the weight tiles (256 x 64 channels = 32 KB per stage)
//     stream through a two-slot global_load_lds ring that never drains between column tiles; a wave tile is 80 pixels x
//     64 channels (20 MFMAs per 9 fragment reads, half the LDS traffic of the 80 x 32 tile);
//   * the accumulators leave through registers: with weights as the MFMA A operand and the weight rows staged in a
//     permuted order every lane owns 16 consecutive output channels of one pixel -> bias / masked residual / ReLU in
//     registers, two 16-byte stores per pixel, the four lane groups cover the pixel's whole 128-B line: no LDS round trip;
//   * BatchNorm statistics (sum / sum of squares of the stored bf16 values) reduced over the 16 pixel lanes with DPP-free
//     shuffles, the two wave rows combined through 2 KB of LDS in fixed order: one deterministic slot per (panel, channel).
// vmcnt discipline: stores and loads share one per-wave counter, so a wave that stores cannot wait for a ring stage without
// also waiting for its output stores to drain (measured: that serialised the 77 MB output stream with the MFMA loop, 44 us).
// Hence 8 compute waves (fragments, MFMA, stores -- no load in their loop) + 4 loader waves (weight ring only), one
// s_barrier per K-step joining all twelve.
#include "common.h"

struct NLoopArgs {
  const char* x;
  const char* w;
  bf16_t* y;
  const float* bias;
  const bf16_t* res;
  const unsigned char* res_bits;
  float* stats;
  const char* zero;
  int H, W, Ho, Wo, Cout, Nstore, ldy, ldr, stride, relu, M;
  int pix_bytes, wrow_bytes;
  int ntiles_n, ntiles_m, rows, nblk128;
};

// MODE 0 = product; 1 = no MFMA, 2 = no epilogue stores, 3 = no weight loads, 4 = no fragment reads + no MFMA (timing ablations,
// env SIMT_NLOOP_MODE; their outputs are meaningless)
template <int KC, int MODE>   // K / 64
__global__ __launch_bounds__(768) void conv1x1_nloop_kernel(NLoopArgs a) {
  constexpr int NT = 512, NLOAD = 256, BM = 160, BN = 256, NSTG = 2;   // 8 compute waves + 4 loader waves
  constexpr int WN = 4, TM = 5, TN = 4;           // waves: 2 (pixels) x 4 (couts); wave tile 80 x 64
  constexpr int A_CHUNK = BM * 128;               // one 64-channel slice of the pixel panel
  constexpr int B_STAGE = BN * 128;
  constexpr int A_IT = 3, B_IT = BN * 8 / NLOAD;  // 16-B pieces per thread: panel 160*8/512 = 2.5 (third pass: waves 0-3); weights 8
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sA = smem;                                // [KC][BM][128 B]
  char* sB = smem + KC * A_CHUNK;                 // [NSTG][BN][128 B]
  float* sS = (float*)(sB + NSTG * B_STAGE);      // [2 (sum, sumsq)][BN] partials of wave row 1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mt = xcd_remap(blockIdx.x, a.ntiles_m);
  const int m0 = mt * a.rows;
  const int m_end = min(a.M, m0 + a.rows);

  const int nstages = a.ntiles_n * KC;
  if (wave >= 8) {
    // ================= loader waves: the weight ring.  Their vmcnt never sees a store, the compute waves' never sees a load.
    // Stage g = (column tile g / KC, K slice g % KC) goes to slot g & 1.  LDS row R of a column tile holds output channel
    // perm(R): within each wave's 64-column block, MFMA row r of 16-row block j is channel (r>>2)*16 + j*4 + (r&3), so that a
    // lane's four accumulator quads are 16 consecutive channels (epilogue below).
    const int ltid = tid - NT;
    const int lc = (ltid & 7) ^ (((ltid >> 3) >> 1) & 7);
    unsigned b_off[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      const int R = i * (NLOAD / 8) + (ltid >> 3);
      const int r = R & 15, j = (R >> 4) & 3;
      const int ch = (R & ~63) + (r >> 2) * 16 + j * 4 + (r & 3);
      b_off[i] = (unsigned)ch * (unsigned)a.wrow_bytes + (unsigned)(lc * 16);
    }
    int ld_nt = 0, ld_kc = 0;
    auto issue = [&](int slot) {
      char* dst = sB + slot * B_STAGE;
      const unsigned base = (unsigned)(ld_nt * BN) * (unsigned)a.wrow_bytes + (unsigned)(ld_kc * 128);
#pragma unroll
      for (int i = 0; i < B_IT; ++i)
        __builtin_amdgcn_global_load_lds(GPTR(a.w + (base + b_off[i])), LPTR(dst + (i * NLOAD + (wave - 8) * 64) * 16), 16, 0, 0);
      if (++ld_kc == KC) { ld_kc = 0; ++ld_nt; }
    };
    if (MODE != 3) issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int g = 0;
    for (int nt = 0; nt < a.ntiles_n; ++nt) {
      for (int kc = 0; kc < KC; ++kc, ++g) {
        __builtin_amdgcn_s_barrier();              // stage g complete for everybody; slot (g+1)&1 was read during step g-1
        if (g + 1 < nstages && MODE != 3) issue((g + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (a.stats) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }   // the two barriers of the statistics hand-over
    }
    return;
  }
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
  const int prow = lane & 15;                      // pixel within a 16-row MFMA block

  int g = 0;
  for (int nt = 0; nt < a.ntiles_n; ++nt) {
    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC; ++kc, ++g) {
      // the loader waves arrive here only after their pieces of stage g have landed
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* px = sA + kc * A_CHUNK + xbase;
      const char* pw = sB + (g & 1) * B_STAGE + wbase;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 xf[TM], wf[TN];
        const int coff = ((4 * s + kq) ^ sw) << 4;
        if (MODE != 4) {
#pragma unroll
          for (int i = 0; i < TM; ++i) xf[i] = *(const bf16x8*)(px + i * 16 * 128 + coff);
#pragma unroll
          for (int j = 0; j < TN; ++j) wf[j] = *(const bf16x8*)(pw + j * 16 * 128 + coff);
        }
        if (MODE != 1 && MODE != 4) {
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf[i], acc[j][i], 0, 0, 0);
        }
      }
    }
    // ---------------- epilogue of column tile nt, straight from the accumulators ----------------
    // lane group q = lane>>4 holds channels 16q + 4j + e of its wave's 64-column block in acc[j][.][e]: 16 consecutive
    // channels = two 16-byte stores per pixel; the four lane groups cover the pixel's whole 128-byte line.
    const int n0 = nt * BN;
    const int c0 = n0 + wn * 64 + (lane >> 4) * 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                  // the two 8-channel halves of the lane's 16 channels
      const int c = c0 + h * 8;
      float s1[8], s2[8], bias8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; bias8[e] = (a.bias && (c + e) < a.Cout) ? a.bias[c + e] : 0.f; }
      const bool plain = !a.bias && !a.res && !a.relu;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * 80 + i * 16 + prow;
        // the value as stored: bf16 of the accumulator (statistics are taken on it, before bias / residual / ReLU)
        uint4 o;
        o.x = (uint32_t)f2bf(acc[2 * h][i][0]) | ((uint32_t)f2bf(acc[2 * h][i][1]) << 16);
        o.y = (uint32_t)f2bf(acc[2 * h][i][2]) | ((uint32_t)f2bf(acc[2 * h][i][3]) << 16);
        o.z = (uint32_t)f2bf(acc[2 * h + 1][i][0]) | ((uint32_t)f2bf(acc[2 * h + 1][i][1]) << 16);
        o.w = (uint32_t)f2bf(acc[2 * h + 1][i][2]) | ((uint32_t)f2bf(acc[2 * h + 1][i][3]) << 16);
        if (m >= m_end || (MODE == 2 && a.M > 0)) continue;
        if (a.stats || !plain) {
          float v[8];
          v[0] = __uint_as_float(o.x << 16); v[1] = __uint_as_float(o.x & 0xffff0000u);
          v[2] = __uint_as_float(o.y << 16); v[3] = __uint_as_float(o.y & 0xffff0000u);
          v[4] = __uint_as_float(o.z << 16); v[5] = __uint_as_float(o.z & 0xffff0000u);
          v[6] = __uint_as_float(o.w << 16); v[7] = __uint_as_float(o.w & 0xffff0000u);
          if (a.stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] += v[e] * v[e]; }
          }
          if (!plain && c < a.Nstore) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
            if (a.res) {
              float rv[8];
              load8(a.res + (long)m * a.ldr + c, rv);
              if (a.res_bits) {
                const unsigned bb = a.res_bits[((long)m * a.ldr + c) >> 3];
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = ((bb >> e) & 1u) ? rv[e] : 0.f;
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            if (a.relu) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
            }
            store8(a.y + (long)m * a.ldy + c, v);
            continue;
          }
        }
        if (c < a.Nstore) *(uint4*)(a.y + (long)m * a.ldy + c) = o;
      }
      if (a.stats) {
        // sum over the 16 pixel lanes of each channel group (lanes with equal lane>>4); wave row 1 hands its sums to row 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) {
            s1[e] += __shfl_xor(s1[e], o, 64);
            s2[e] += __shfl_xor(s2[e], o, 64);
          }
        }
        const int cl = wn * 64 + (lane >> 4) * 16 + h * 8;
        if (wm == 1 && prow == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { sS[cl + e] = s1[e]; sS[BN + cl + e] = s2[e]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (all 12 waves: the loaders mirror it)
        if (wm == 0 && prow == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int nn = n0 + cl + e;
            if (nn < a.Cout) {
              a.stats[((long)mt * 2 + 0) * a.Cout + nn] = s1[e] + sS[cl + e];
              a.stats[((long)mt * 2 + 1) * a.Cout + nn] = s2[e] + sS[BN + cl + e];
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
        // (each half writes its own sS columns; the next column tile rewrites them only after further barriers)
      }
    }
  }
}

template <int KC, int MODE>
static int launch_nloop_m(const NLoopArgs& k, hipStream_t st) {
  const size_t lds = (size_t)KC * 160 * 128 + 2 * 256 * 128 + 2 * 256 * 4;
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
    if (mode == 4) return launch_nloop_m<KC, 4>(k, st);
  }
  return launch_nloop_m<KC, 0>(k, st);
}

// Does this descriptor take the resident-panel kernel?  (1 tap at offset (0,0), Cin in {64, 128, 256}, wide bf16 output)
bool simt_conv_nloop_eligible(const simt_conv_desc* d) {
  if (d->dtype_in != SIMT_BF16 || d->dtype_out != SIMT_BF16) return false;
  if (d->ntaps != 1 || d->dy[0] != 0 || d->dx[0] != 0 || d->mask) return false;
  if (!(d->Cin == 64 || d->Cin == 128 || d->Cin == 256)) return false;
  if (d->Npad % 256 != 0 || d->Cout < 256 || d->Nstore % 8 != 0 || d->ldy % 8 != 0) return false;
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

int simt_conv_fprop_bf16_nloop(const simt_conv_desc* d, simt_stream_t stream) {
  NLoopArgs k;
  k.x = (const char*)d->x; k.w = (const char*)d->w; k.y = (bf16_t*)d->y; k.bias = d->bias; k.res = (const bf16_t*)d->res;
  k.res_bits = d->res_bits; k.stats = d->stats; k.zero = (const char*)simt_zero_page();
  k.H = d->H; k.W = d->W; k.Ho = d->Ho; k.Wo = d->Wo; k.Cout = d->Cout; k.Nstore = d->Nstore; k.ldy = d->ldy; k.ldr = d->ldr;
  k.stride = d->stride; k.relu = d->relu; k.M = d->B * d->Ho * d->Wo;
  k.pix_bytes = d->Cin * 2;
  k.wrow_bytes = d->Cin * 2;
  k.ntiles_n = d->Npad / 256;
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
