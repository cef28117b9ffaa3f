// Direct 7x7 stride-2 pad-3 convolution of the 3-channel input image -- the ResNet stem (model/deeplab_multi.py:127 `self.conv1 =
// nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)`, forward :172-173) -- bf16, gfx950.  Round 6 (VERDICT r5 #1c).
//
// Rounds 1-5 ran the stem as a materialised im2col matrix [B*Ho*Wo][192] (226 MB at 4 x 768 x 768: a 134-us pass) followed by one 64-column GEMM
// per network that re-read it (228 MB in, 76 MB out each).  Cin = 3 is not an MFMA shape -- but a filter ROW is: the 7 taps x 3 channels of
// one filter row are 21 CONTIGUOUS values of an NHWC image row, so with the image patch staged in LDS as [row][col][channel] the B operand of
// one v_mfma_f32_16x16x32_bf16 k-step is, per lane, 8 consecutive LDS elements starting at pixel 2*ox - 3 (+ 8 * k-group) of row 2*oy - 3 + r:
// K = 7 k-steps of 32 (21 real + 11 zero-weight columns), no im2col anywhere.  The trainable and the frozen network convolve the SAME image
// (tools/trainV2_simt.py:351-353 and :370), so one launch takes up to two weight sets: 128 output channels per staged patch, each set with
// its own epilogue (BatchNorm batch statistics | folded bias + ReLU) and output buffer.
//
// Workgroup: 8 waves on an 8 x 32 tile of output pixels (patch 21 x 69 pixels x 3 channels bf16, 448-byte row pitch, pad columns zero: the
// lanes of the last k-group read past the 21 real values -- finite data times zero weights).  Wave w: 32 output channels (w & 3: two 16-row
// weight blocks of one set, 14 A fragments = 56 VGPRs resident) x 128 pixels (w >> 2: four tile rows, 8 pixel blocks).  Per tile and wave
// 112 MFMAs, 224 ds_read_b32.  HBM: the image once (fp32 NCHW, halo re-read 1.3x) + the outputs: ~30-40 us for both networks at 4 x 768 x 768.
#include "common.h"

namespace {

constexpr int TH = 8, TW = 32;                 // output pixels per tile
constexpr int PR = 2 * TH + 5, PC = 2 * TW + 5;   // patch rows / columns (input pixels)
constexpr int PITCH = 224;                     // elements per patch row (PC * 3 = 207 real + zero pad): k-group 3 of the last pixel reads up to 62 * 3 + 31 = 217

struct StemArgs {
  const float* x;
  int B, H, W, Ho, Wo, tiles_y, tiles_x, nsets;
  const bf16_t* w[2];      // [64][7][32] bf16, k' = s * 3 + c (21 real, 11 zero)
  bf16_t* y[2];            // [B*Ho*Wo][64]
  const float* bias[2];
  int relu[2];
  float* stats[2];         // [tiles][2][64]
};

__global__ __launch_bounds__(512, 2) void stem7_kernel(StemArgs a) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[PR * PITCH];
  __shared__ float sred[2][2][2][64];          // [set][pixel half][sum | sum of squares][channel]
  // output rows of one pass, both sets: [set][4 tile rows x 32 pixels][64 channels] bf16 at a 144-byte pitch -- the accumulator layout (4 channels
  // of 16 pixels per lane group) written straight to HBM is 32-byte pieces of 16 different lines per store instruction (the first version: 135 us
  // for 150 MB); through LDS every store is 16 bytes of a full 128-byte pixel row
  constexpr int OP = 144;
  __shared__ __attribute__((aligned(16))) char otile[2 * 128 * OP];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 3, ph = wave >> 2;     // channel quarter of the 128 (2 sets x 64), pixel half of the tile
  const int set = cb >> 1;
  const bool active = set < a.nsets;
  int t = blockIdx.x;
  const int tx = t % a.tiles_x; t /= a.tiles_x;
  const int ty = t % a.tiles_y;
  const int b = t / a.tiles_y;
  const int oy0 = ty * TH, ox0 = tx * TW;
  // ---- weights of this wave: 2 blocks of 16 output channels x 7 filter rows, A fragments (row = channel lane & 15, k' = (lane >> 4) * 8 ...)
  bf16x8 wf[2][7];
  if (active) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 7; ++r)
        wf[j][r] = *(const bf16x8*)(a.w[set] + ((size_t)((cb & 1) * 32 + j * 16 + (lane & 15)) * 7 + r) * 32 + (lane >> 4) * 8);
  }
  // ---- patch: fp32 NCHW -> bf16 [row][col][c] in LDS (coalesced along the image row; out-of-image pixels and the pad columns are zeros)
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
  const float* xb = a.x + (size_t)b * 3 * a.H * a.W;
  {
    // all loads of a thread are requested before the first is used (constant trip count, fully unrolled: the staging is 9 dependent-free
    // round trips otherwise -- the first version of this kernel spent ~2/3 of its time here)
    constexpr int NIT = (PR * 3 * PC + 511) / 512;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 512;
      const int col = i % PC, jc = i / PC, c = jc % 3, row = jc / 3;
      const int iy = iy0 + row, ix = ix0 + col;
      // UNCONDITIONAL loads from clamped coordinates, zeroed by a select afterwards: a load under `if` is its own basic block with its own
      // s_waitcnt vmcnt(0) -- nine serialised HBM round trips per tile (disassembly of the first version: 11 of its 16 us per tile)
      const bool inb = i < PR * 3 * PC && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int cy = min(max(iy, 0), a.H - 1), cx = min(max(ix, 0), a.W - 1), cc = min(c, 2);
      const float t = xb[((size_t)cc * a.H + cy) * a.W + cx];
      v[it] = inb ? t : 0.f;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 512;
      const int col = i % PC, jc = i / PC, c = jc % 3, row = jc / 3;
      if (i < PR * 3 * PC) patch[row * PITCH + col * 3 + c] = f2bf(v[it]);
    }
  }
  for (int i = tid; i < PR * (PITCH - PC * 3); i += 512) patch[(i / (PITCH - PC * 3)) * PITCH + PC * 3 + i % (PITCH - PC * 3)] = (bf16_t)0;
  __syncthreads();
  // ---- two passes of 4 pixel blocks each (64 accumulator registers would cost the second workgroup per CU): MFMA, then the epilogue:
  // lane holds channels (cb & 1) * 32 + j * 16 + (lane >> 4) * 4 + e of pixel (lane & 15) of each block
  float s1[2][4], s2[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[j][e] = 0.f; s2[j][e] = 0.f; }
  const bool want_stats = active && a.stats[set] != nullptr;
  const bool relu = active && a.relu[set] != 0;
  float bv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[j][e] = (active && a.bias[set]) ? a.bias[set][(cb & 1) * 32 + j * 16 + (lane >> 4) * 4 + e] : 0.f;
  const int kq = lane >> 4, pxl = lane & 15;
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (active) {
      f32x4 acc[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // pixel block p = half * 4 + q of this wave: tile row ph * 4 + p / 2, columns (p & 1) * 16 + (lane & 15)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int p = half * 4 + q;
        const int py = ph * 4 + (p >> 1), px = (p & 1) * 16 + pxl;
        const bf16_t* src = patch + (2 * py) * PITCH + (2 * px) * 3 + kq * 8;          // 4-byte aligned: (6 px + 8 kq) elements
#pragma unroll
        for (int r = 0; r < 7; ++r) {
          const unsigned* s4 = (const unsigned*)(src + r * PITCH);
          union { unsigned u[4]; bf16x8 v; } f;
          f.u[0] = s4[0]; f.u[1] = s4[1]; f.u[2] = s4[2]; f.u[3] = s4[3];
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[j][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][r], f.v, acc[j][q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c0 = (cb & 1) * 32 + j * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int p = half * 4 + q;
          const int oy = oy0 + ph * 4 + (p >> 1), ox = ox0 + (p & 1) * 16 + (lane & 15);
          if (oy >= a.Ho || ox >= a.Wo) continue;
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[j][q][e] + bv[j][e];
            if (relu) v[e] = v[e] > 0.f ? v[e] : 0.f;
          }
          uint2 o;
          o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
          *(uint2*)(otile + (set * 128 + (ph * 2 + (q >> 1)) * 32 + (p & 1) * 16 + (lane & 15)) * OP + c0 * 2) = o;
          if (want_stats) {                      // statistics of the values AS STORED (bf16), like every conv epilogue of this library
            const float q0 = __uint_as_float(o.x << 16), q1 = __uint_as_float(o.x & 0xffff0000u);
            const float q2 = __uint_as_float(o.y << 16), q3 = __uint_as_float(o.y & 0xffff0000u);
            s1[j][0] += q0; s1[j][1] += q1; s1[j][2] += q2; s1[j][3] += q3;
            s2[j][0] = fmaf(q0, q0, s2[j][0]); s2[j][1] = fmaf(q1, q1, s2[j][1]); s2[j][2] = fmaf(q2, q2, s2[j][2]); s2[j][3] = fmaf(q3, q3, s2[j][3]);
          }
        }
      }
    }
    __syncthreads();
    // copy-out of this pass: 2 sets x 128 pixels x 8 pieces of 16 bytes; row slot rs of the pass = tile row (rs >> 1) * 4 + 2 * half + (rs & 1)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = tid + it * 512;
      const int st_ = idx >> 10, rem = idx & 1023, lp = rem >> 3, ch = rem & 7;
      const int rs = lp >> 5, oy = oy0 + (rs >> 1) * 4 + 2 * half + (rs & 1), ox = ox0 + (lp & 31);
      if (st_ < a.nsets && oy < a.Ho && ox < a.Wo)
        st_out16(a.y[st_] + ((size_t)(b * a.Ho + oy) * a.Wo + ox) * 64 + ch * 8, *(const uint4*)(otile + (st_ * 128 + lp) * OP + ch * 16));
    }
    __syncthreads();
  }
  // ---- per-tile BatchNorm partial sums: 16 pixel lanes (DPP row, fixed order) -> LDS -> the two pixel halves in order -> [tile][2][64]
  if (active && a.stats[set]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t1 = s1[j][e], t2 = s2[j][e];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 64); t2 += __shfl_xor(t2, o, 64); }
        if ((lane & 15) == 0) {
          const int c = (cb & 1) * 32 + j * 16 + (lane >> 4) * 4 + e;
          sred[set][ph][0][c] = t1;
          sred[set][ph][1][c] = t2;
        }
      }
  }
  __syncthreads();
  if (tid < 256) {
    const int st_ = tid >> 7, q = (tid >> 6) & 1, c = tid & 63;
    if (st_ < a.nsets && a.stats[st_])
      a.stats[st_][((size_t)blockIdx.x * 2 + q) * 64 + c] = sred[st_][0][q][c] + sred[st_][1][q][c];
  }
}

// conv1.weight fp32 OIHW [64][3][7][7] (optionally scaled per output channel: the frozen net's folded BatchNorm) -> bf16 [64][7][32], k' = s*3 + c
__global__ void stem7_pack_kernel(const float* w, const float* cscale, bf16_t* dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 7 * 32) return;
  const int kk = i & 31, r = (i >> 5) % 7, o = i / (7 * 32);
  float v = 0.f;
  if (kk < 21) {
    const int s = kk / 3, c = kk - s * 3;
    v = w[((o * 3 + c) * 7 + r) * 7 + s];
    if (cscale) v *= cscale[o];
  }
  dst[i] = f2bf(v);
}

}  // namespace

extern "C" int simt_stem7_tiles(int B, int Ho, int Wo) { return B * ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW); }

extern "C" int simt_stem7_pack(const float* w, const float* cscale, void* dst, simt_stream_t stream) {
  SIMT_CHECK(w && dst);
  hipLaunchKernelGGL(stem7_pack_kernel, dim3((64 * 7 * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, cscale, (bf16_t*)dst);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

extern "C" int simt_stem7_fwd(const simt_stem_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->x && d->nsets >= 1 && d->nsets <= 2 && d->B > 0 && d->H > 0 && d->W > 0);
  SIMT_CHECK(d->Ho == (d->H + 6 - 7) / 2 + 1 && d->Wo == (d->W + 6 - 7) / 2 + 1);
  StemArgs a;
  a.x = d->x; a.B = d->B; a.H = d->H; a.W = d->W; a.Ho = d->Ho; a.Wo = d->Wo; a.nsets = d->nsets;
  a.tiles_y = (d->Ho + TH - 1) / TH; a.tiles_x = (d->Wo + TW - 1) / TW;
  for (int s = 0; s < 2; ++s) {
    a.w[s] = nullptr; a.y[s] = nullptr; a.bias[s] = nullptr; a.relu[s] = 0; a.stats[s] = nullptr;
    if (s >= d->nsets) continue;
    SIMT_CHECK(d->w[s] && d->y[s]);
    a.w[s] = (const bf16_t*)d->w[s]; a.y[s] = (bf16_t*)d->y[s]; a.bias[s] = d->bias[s]; a.relu[s] = d->relu[s]; a.stats[s] = d->stats[s];
  }
  hipLaunchKernelGGL(stem7_kernel, dim3(a.B * a.tiles_y * a.tiles_x), dim3(512), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
