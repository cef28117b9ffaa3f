// Fused SimT head: bilinear(align_corners) upsample + softmax + every per-pixel loss term of one training
// sub-iteration, and their gradients w.r.t. the LOW-RES logits and the transition matrices.
//
// Replaces the inline loss body of the reference, tools/trainV2_simt.py:351-409 (plus :202-230 Placeholder_loss and
// utils/loss.py:14-40 CrossEntropy2d(is_softmax=False)): ~50 full-resolution eager passes (about 10 GB/step at
// 4x768x768) become two streaming passes over the pixels that only read the low-res logits (L2-resident) and the
// labels.  Per pixel everything (2 x Q logits, C fixed-model probabilities) lives in registers; Q = C+K <= 40.
//
//   pass1 : loss sums / counts, anchor arg-max keys (first-index tie-break), Exist bitmasks, dL_y/dT partials
//   final : fixed-order reduction of the block partials (double), anchors gathered, scalar losses
//   pass2 : per-pixel dL/d(upsampled logits) with the now-known 1/N factors, reduced along x inside the block
//           (exact adjoint of the interpolation, gather form -> deterministic, no float atomics)
//   yred  : reduction along y -> dL/d(low-res logits), written fp32 and in the conv dtype (padded for the dgrad GEMM)
//
// Quirks of the reference that are reproduced on purpose:
//   * Placeholder_loss replaces the arg-max logit by -0.0 (not -1000)            (trainV2_simt.py:207-209)
//   * Placeholder_y = argmax([0]*C ++ open logits) -> class 0 when all open logits <= 0   (:219-222)
//   * CE over zero valid pixels gives NaN loss and zero gradient
#include "common.h"
#include <math.h>

#define QMAX 40
#ifndef SIMT_HEAD_ABL
#define SIMT_HEAD_ABL 0      // timing ablations (results meaningless): 1 no dT partials, 2 no anchors, 4 no x-reduction, 8 no gradient terms,
                             // 16 no run sums, 32 no gradient staging
#endif
#define NSCAL 13
#define XR_MAX 287   // low-res columns one 256-pixel chunk may touch on the run-based x-reduction of pass 2 (more: the scanning form)

struct HeadGeom {
  int B, h, w, H, W, C, Q, ldp, ldf;
  float sy, sx;  // align_corners=True: (h-1)/(H-1), (w-1)/(W-1); half-pixel: h/H, w/W -- in float like ATen area_pixel_compute_scale
  int half;      // 0: interp_target of trainV2_simt.py:301 (align_corners=True);  1: F.interpolate(bilinear) default of
                 // model/deeplabv3.py:137 (align_corners=False: src = (dst + 0.5) * scale - 0.5, clamped at 0)
  int fix_logits;  // 1: fixp holds the frozen model's LOGITS and the posterior is softmax(upsample(logits)) (a model that upsamples
                   // inside, deeplabv3.py:137 + trainV2_simt.py:354);  0: fixp holds low-res probabilities, upsampled (:354)
  int single;      // 1: one-output model (DeepLabv3 / DeepLab-VGG): there is no auxiliary head, every head-1 term is dropped
};

// source coordinate of destination index d (ATen area_pixel_compute_source_index)
__device__ __forceinline__ float src_coord(int half, float scale, int d) {
  return half ? fmaxf(((float)d + 0.5f) * scale - 0.5f, 0.f) : scale * (float)d;
}
// destination indices [lo, hi] that can touch source index l (superset; weights are re-derived per element)
__device__ __forceinline__ void dst_range(int l, int out, float scale, int& lo, int& hi) {
  if (scale > 0.f) {
    const float r = 1.f / scale;
    lo = (int)floorf(((float)l - 1.f) * r) - 2;
    hi = (int)ceilf(((float)l + 2.f) * r) + 2;
  } else { lo = 0; hi = out - 1; }
  lo = max(lo, 0); hi = min(hi, out - 1);
}

struct Taps {
  int o00, o01, o10, o11;  // pixel indices (b*h + iy)*w + ix
  float wy0, wy1, wx0, wx1;
};

__device__ __forceinline__ Taps make_taps(const HeadGeom& g, int b, int y, int x) {
  Taps t;
  float fy = src_coord(g.half, g.sy, y), fx = src_coord(g.half, g.sx, x);
  int iy0 = (int)fy, ix0 = (int)fx;
  if (iy0 > g.h - 1) iy0 = g.h - 1;
  if (ix0 > g.w - 1) ix0 = g.w - 1;
  int iy1 = iy0 + (iy0 < g.h - 1 ? 1 : 0), ix1 = ix0 + (ix0 < g.w - 1 ? 1 : 0);
  t.wy1 = fy - (float)iy0; t.wy0 = 1.f - t.wy1;
  t.wx1 = fx - (float)ix0; t.wx0 = 1.f - t.wx1;
  int base = b * g.h;
  t.o00 = (base + iy0) * g.w + ix0; t.o01 = (base + iy0) * g.w + ix1;
  t.o10 = (base + iy1) * g.w + ix0; t.o11 = (base + iy1) * g.w + ix1;
  return t;
}

// ATen order: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11)
__device__ __forceinline__ float lerp4(const Taps& t, float v00, float v01, float v10, float v11) {
  return t.wy0 * (t.wx0 * v00 + t.wx1 * v01) + t.wy1 * (t.wx0 * v10 + t.wx1 * v11);
}

// e^x for x <= 0 (every call site subtracts the running maximum first): one v_exp_f32 of x*log2(e).  The rounding of the product is
// |x| 2^-24 relative in the result, so an entry's ABSOLUTE error is at most |x| e^x 2^-24 <= 2.2e-8: below half an ulp of the softmax
// sum (>= 1) it goes into.  expf's extended-precision range reduction (13 instructions per call, ~1 600 of pass 2's 8 000) bought
// nothing those sums keep.
__device__ __forceinline__ float exp_le0(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

template <int NMAX>
__device__ __forceinline__ void interp_vec(const float* src, int ld, int n, const Taps& t, float* out) {
  const float* p00 = src + (long)t.o00 * ld;
  const float* p01 = src + (long)t.o01 * ld;
  const float* p10 = src + (long)t.o10 * ld;
  const float* p11 = src + (long)t.o11 * ld;
#pragma unroll
  for (int j4 = 0; j4 < NMAX / 4; ++j4) {
    if (j4 * 4 < n) {
      float4 a = *(const float4*)(p00 + j4 * 4), b = *(const float4*)(p01 + j4 * 4);
      float4 c = *(const float4*)(p10 + j4 * 4), d = *(const float4*)(p11 + j4 * 4);
      out[j4 * 4 + 0] = lerp4(t, a.x, b.x, c.x, d.x);
      out[j4 * 4 + 1] = lerp4(t, a.y, b.y, c.y, d.y);
      out[j4 * 4 + 2] = lerp4(t, a.z, b.z, c.z, d.z);
      out[j4 * 4 + 3] = lerp4(t, a.w, b.w, c.w, d.w);
    }
  }
}

// max / first arg-max over channels [0, n) of the interpolated vector, without materialising it
template <int NMAX>
__device__ __forceinline__ void interp_argmax(const float* src, int ld, int n, const Taps& t, float& fm, int& fa) {
  const float* p00 = src + (long)t.o00 * ld;
  const float* p01 = src + (long)t.o01 * ld;
  const float* p10 = src + (long)t.o10 * ld;
  const float* p11 = src + (long)t.o11 * ld;
  fm = -INFINITY;
  fa = 0;
#pragma unroll
  for (int j4 = 0; j4 < NMAX / 4; ++j4) {
    if (j4 * 4 < n) {
      float4 a = *(const float4*)(p00 + j4 * 4), b = *(const float4*)(p01 + j4 * 4);
      float4 c = *(const float4*)(p10 + j4 * 4), d = *(const float4*)(p11 + j4 * 4);
      const float v0 = lerp4(t, a.x, b.x, c.x, d.x), v1 = lerp4(t, a.y, b.y, c.y, d.y);
      const float v2 = lerp4(t, a.z, b.z, c.z, d.z), v3 = lerp4(t, a.w, b.w, c.w, d.w);
      if (j4 * 4 + 0 < n && v0 > fm) { fm = v0; fa = j4 * 4 + 0; }
      if (j4 * 4 + 1 < n && v1 > fm) { fm = v1; fa = j4 * 4 + 1; }
      if (j4 * 4 + 2 < n && v2 > fm) { fm = v2; fa = j4 * 4 + 2; }
      if (j4 * 4 + 3 < n && v3 > fm) { fm = v3; fa = j4 * 4 + 3; }
    }
  }
}

// frozen model's posterior at one pixel -> (max probability, its first arg-max).  fix_logits: softmax AFTER the interpolation.
template <int NMAX>
__device__ __forceinline__ void fixed_posterior(const HeadGeom& g, int C, const float* fixp, const Taps& t, float& fm, int& fa) {
  if (!g.fix_logits) { interp_argmax<NMAX>(fixp, g.ldf, C, t, fm, fa); return; }
  float v[NMAX];
  interp_vec<NMAX>(fixp, g.ldf, C, t, v);
  float mx = v[0];
  fa = 0;
#pragma unroll
  for (int j = 1; j < NMAX; ++j)
    if (j < C && v[j] > mx) { mx = v[j]; fa = j; }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NMAX; ++j)
    if (j < C) sum += exp_le0(v[j] - mx);
  fm = 1.0f / sum;                      // = exp(mx - mx) / sum, the value torch.softmax(...).max() returns
}

struct HeadEval {
  int arg;       // argmax_j v[j], first index
  float vmax, sum, lse;
  int pseudo1;   // arg if (arg < C && 1/sum > th_high) else 255
  int yopen;     // Placeholder_y (valid only when pseudo1 != 255)
  float m2, sum2, lse2;  // log-sum-exp of `predict` (arg-max logit replaced by -0.0)
};

template <int QM>
__device__ __forceinline__ void eval_head(const float* v, int Q, int C, float th_high, HeadEval& e) {
  float vmax = v[0];
  int arg = 0;
#pragma unroll
  for (int j = 1; j < QM; ++j)
    if (j < Q && v[j] > vmax) { vmax = v[j]; arg = j; }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < QM; ++j)
    if (j < Q) sum += exp_le0(v[j] - vmax);
  e.arg = arg; e.vmax = vmax; e.sum = sum; e.lse = vmax + logf(sum);
  float pm = 1.0f / sum;
  e.pseudo1 = (arg < C && pm > th_high) ? arg : 255;
  // predict = v with v[arg] := -0.0 ; Placeholder_y over [0]*C ++ predict[C:]
  float best = 0.f;
  int y = 0;
  float m2 = 0.f;  // the replaced entry contributes the value 0
#pragma unroll
  for (int j = 0; j < QM; ++j)
    if (j < Q && j != arg) {
      m2 = fmaxf(m2, v[j]);
      if (j >= C && v[j] > best) { best = v[j]; y = j; }
    }
  float s2 = exp_le0(0.f - m2);
#pragma unroll
  for (int j = 0; j < QM; ++j)
    if (j < Q && j != arg) s2 += exp_le0(v[j] - m2);
  e.yopen = y; e.m2 = m2; e.sum2 = s2; e.lse2 = m2 + logf(s2);
}

__device__ __forceinline__ unsigned int f32_ord(float f) {
  unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unord(unsigned int o) {
  unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

// --------------------------------------------------------------------------------------------------------
// low-res softmax of the fixed model's logits (reference :354 softmax BEFORE interp_target)
// --------------------------------------------------------------------------------------------------------
__global__ void softmax_rows_kernel(const float* in, int ldi, float* out, int ldo, long M, int C) {
  long m = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* p = in + m * ldi;
  float mx = p[0];
  for (int c = 1; c < C; ++c) mx = fmaxf(mx, p[c]);
  float s = 0.f;
  for (int c = 0; c < C; ++c) s += expf(p[c] - mx);
  float inv = 1.0f / s;
  float* o = out + m * ldo;
  for (int c = 0; c < C; ++c) o[c] = expf(p[c] - mx) * inv;
  for (int c = C; c < ldo; ++c) o[c] = 0.f;
}

extern "C" int simt_softmax_rows(const float* in, int ldi, float* out, int ldo, long M, int C, simt_stream_t stream) {
  SIMT_CHECK(in && out && C <= ldi && C <= ldo);
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, ldi,
                     out, ldo, M, C);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// --------------------------------------------------------------------------------------------------------
// pass 1
// --------------------------------------------------------------------------------------------------------
struct HeadArgs {
  HeadGeom g;
  const float* pred1;
  const float* pred2;
  const float* fixp;
  const long long* label;
  const float* T1;
  const float* T2;
  float th_high, th_low, lambda_seg, lambda_place;
  float* part;                 // [nblk][NSCAL + 2*Q*C]
  unsigned long long* keys;    // [2*QMAX] anchor keys, [2*QMAX .. 2*QMAX+1] exist masks (zeroed per call)
  float* hout;                 // finalize output (see simt_head_out_* offsets)
  float* g1;                   // rows == 1: [2][B][H][w][QP] x-reduced gradients;  rows > 1: [2][B][H / rows][3][w][QP], see head_pass2_kernel
  int QP;
  int rows;                    // image rows per pass-2 block (head_rows_per_block)
  float gscale;
  int mode;                    // 0 = SimT loss block; 1 = warm-up stage: plain CE of both heads against `label`
  unsigned char* conf_out;     // optional [B][H][W]: the confidence label decided per pixel (255 = none)
  unsigned char* label_ws;     // optional [B][H][W]: the checked noisy label (255 = ignored / invalid / warm-up mode), pass 1 -> pass 2
};

// Full-wave sum with DPP row operations (no LDS crossbar): result valid in lane 63.
__device__ __forceinline__ float wave_sum_dpp63(float v) {
  int x = __float_as_int(v);
#define DPP_ADD(ctrl, rmask)                                                                                  \
  x = __float_as_int(__int_as_float(x) + __int_as_float(__builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, true)))
  DPP_ADD(0xB1, 0xf);    // quad_perm [1,0,3,2]
  DPP_ADD(0x4E, 0xf);    // quad_perm [2,3,0,1]
  DPP_ADD(0x141, 0xf);   // row_half_mirror
  DPP_ADD(0x140, 0xf);   // row_mirror      -> every lane holds its 16-lane row sum
  DPP_ADD(0x142, 0xa);   // row_bcast:15    -> rows 1 and 3 += previous row
  DPP_ADD(0x143, 0xc);   // row_bcast:31    -> rows 2 and 3 += row 1 (lane 31)
#undef DPP_ADD
  return __int_as_float(x);
}

// Full-wave max / OR with DPP row operations, broadcast from lane 63 through an SGPR (v_readlane): no LDS crossbar.  The
// __shfl_xor forms of these (6 ds_bpermute per value, 2 x Q values per 64 pixels for the anchor arg-max) were the largest single cost of
// pass 1.  `old` = the lane's own value: lanes a row mask leaves out keep it (max(x, x) = x).
__device__ __forceinline__ float wave_max_f(float v) {
  int x = __float_as_int(v);
#define DPP_MAX(ctrl, rmask) x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false))))
  DPP_MAX(0xB1, 0xf);    // quad_perm [1,0,3,2]
  DPP_MAX(0x4E, 0xf);    // quad_perm [2,3,0,1]
  DPP_MAX(0x141, 0xf);   // row_half_mirror
  DPP_MAX(0x140, 0xf);   // row_mirror      -> every lane holds its 16-lane row max
  DPP_MAX(0x142, 0xa);   // row_bcast:15    -> rows 1 and 3: max with the previous row
  DPP_MAX(0x143, 0xc);   // row_bcast:31    -> rows 2 and 3: max with row 1 (lane 31)
#undef DPP_MAX
  return __int_as_float(__builtin_amdgcn_readlane(x, 63));
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
  int x = (int)v;
#define DPP_OR(ctrl, rmask) x |= __builtin_amdgcn_update_dpp(x, x, ctrl, rmask, 0xf, false)
  DPP_OR(0xB1, 0xf); DPP_OR(0x4E, 0xf); DPP_OR(0x141, 0xf); DPP_OR(0x140, 0xf); DPP_OR(0x142, 0xa); DPP_OR(0x143, 0xc);
#undef DPP_OR
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ unsigned long long wave_or_u64(unsigned long long v) {
  return ((unsigned long long)wave_or_u32((unsigned)(v >> 32)) << 32) | wave_or_u32((unsigned)v);
}

// QT / CT: the channel counts as compile-time constants (0 = taken from the geometry at run time): every `j < Q` predicate of the unrolled
// per-channel loops folds away -- with run-time counts they were a third of the instructions and spilled SGPR masks into VGPR lanes.
#ifndef SIMT_HEAD_P1_WAVES
#define SIMT_HEAD_P1_WAVES 2      // 3 waves per SIMD need 21 spilled VGPRs (335-348 us); 2 waves, no scratch: 352-375 us (profiles/tools/ab_head.py)
#endif
template <int QM, int QT, int CT>
__global__ __launch_bounds__(256, (QM <= 24 ? SIMT_HEAD_P1_WAVES : 2)) void head_pass1_kernel(HeadArgs a) {
  const HeadGeom g = a.g;
  const int Q = QT ? QT : g.Q, C = CT ? CT : g.C, QC = Q * C;
  constexpr int CM = CT ? (CT + 3) / 4 * 4 : QM;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = (float*)smem;                                        // [2][QC]
  float* sdT = sT + 2 * QC;                                        // [4 waves][2][QC]
  unsigned long long* sKey = (unsigned long long*)(sdT + 8 * QC);  // [2*QMAX]  (8-byte aligned: QC even or not -> pad)
  sKey = (unsigned long long*)(((uintptr_t)sKey + 7) & ~(uintptr_t)7);
  unsigned long long* sEx = sKey + 2 * QMAX;                       // [2]
  float* sRed = (float*)(sEx + 2);                                 // [4][NSCAL]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < QC; i += 256) { sT[i] = a.mode == 0 ? a.T1[i] : 0.f; sT[QC + i] = a.mode == 0 ? a.T2[i] : 0.f; }
  for (int i = tid; i < 8 * QC; i += 256) sdT[i] = 0.f;
  if (tid < 2 * QMAX) sKey[tid] = 0ull;
  if (tid < 2) sEx[tid] = 0ull;
  __syncthreads();

  float acc[NSCAL];
#pragma unroll
  for (int i = 0; i < NSCAL; ++i) acc[i] = 0.f;

  const long P = (long)g.B * g.H * g.W;
  const long ngroups = (P + 255) / 256;
  // Logical block id: blocks that share an XCD (b, b + 8, ...) take CONSECUTIVE logical ids, so the pixel groups of one XCD lie in a few bands
  // of the image and the low-res maps they gather from (14.5 MB in all) stay in that XCD's 4 MB L2 (round 3: 202 MB of fabric reads per launch
  // for ~30 MB of operands).  Group -> partial-sum slot assignment is unchanged: bitwise the same results.
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);
  for (long grp = lbid; grp < ngroups; grp += gridDim.x) {
    const long p = grp * 256 + tid;
    const bool live = p < P;
    int b = 0, y = 0, x = 0;
    if (live) {
      x = (int)(p % g.W);
      long t = p / g.W;
      y = (int)(t % g.H);
      b = (int)(t / g.H);
    }
    Taps tp = make_taps(g, b, y, x);
    // ---- fixed-model posterior -> confidence label (reference :354-361)
    float fm = 0.f;
    int fa = 0;
    if (a.mode == 0) fixed_posterior<CM>(g, C, a.fixp, tp, fm, fa);
    asm volatile("" ::: "memory");   // keep the next gathers from being hoisted above (register pressure)
    int conf = (fm > a.th_high) ? fa : 255;
    if (fm < a.th_low) conf = C;

    float v2[QM], v1[QM];
    interp_vec<QM>(a.pred2, g.ldp, Q, tp, v2);
    if (g.single) {
#pragma unroll
      for (int j = 0; j < QM; ++j) v1[j] = v2[j];      // no auxiliary head: its slots mirror the main head and are ignored downstream
    } else {
      interp_vec<QM>(a.pred1, g.ldp, Q, tp, v1);
    }
    HeadEval e2, e1;
    eval_head<QM>(v2, Q, C, a.th_high, e2);
    eval_head<QM>(v1, Q, C, a.th_high, e1);
    if (conf == C) conf = (e2.arg >= C) ? e2.arg : 255;  // reference :387-393

    long long lab = live ? a.label[p] : 255;
    bool lab_ok = live && lab >= 0 && lab != 255 && lab < C;
    const int labi = lab_ok ? (int)lab : 0;
    // a label outside [0, C) that is not the ignore value: the reference's nll_loss / CrossEntropyLoss raise ("Target out of bounds");
    // a kernel cannot, so the pixel is skipped and COUNTED (hout[15]): the host side raises when it reads the losses
    if (live && !lab_ok && lab != 255) acc[12] += 1.f;
    if (a.mode == 1) {
      // warm-up stage (tools/trainV1_warmup.py:217-224): CrossEntropyLoss(ignore_index=255) of both heads against the
      // label itself; no placeholder, noise-posterior or anchor terms
      conf = lab_ok ? labi : 255;
      e1.pseudo1 = 255;
      e2.pseudo1 = 255;
      lab_ok = false;
    }

    if (live) {
      if (a.conf_out) a.conf_out[p] = (unsigned char)conf;
      if (a.label_ws) a.label_ws[p] = lab_ok ? (unsigned char)labi : (unsigned char)255;
      if (conf != 255) {
        float l1 = 0.f, l2 = 0.f;
#pragma unroll
        for (int j = 0; j < QM; ++j)
          if (j == conf) { l1 = e1.lse - v1[j]; l2 = e2.lse - v2[j]; }
        acc[0] += l1; acc[1] += l2; acc[8] += 1.f;
      }
      if (e1.pseudo1 != 255) {
        acc[2] += e1.lse - e1.vmax;
        float vy = 0.f;
#pragma unroll
        for (int j = 0; j < QM; ++j)
          if (j == e1.yopen && j != e1.arg) vy = v1[j];
        acc[4] += e1.lse2 - vy;
        acc[9] += 1.f;
      }
      if (e2.pseudo1 != 255) {
        acc[3] += e2.lse - e2.vmax;
        float vy = 0.f;
#pragma unroll
        for (int j = 0; j < QM; ++j)
          if (j == e2.yopen && j != e2.arg) vy = v2[j];
        acc[5] += e2.lse2 - vy;
        acc[10] += 1.f;
      }
    }
    // ---- noise-posterior loss: q = softmax(v); r = q.T[:,label]  (reference :402-409, utils/loss.py:29-39)
    float r1 = 0.f, r2 = 0.f;
    const float inv1 = 1.0f / e1.sum, inv2 = 1.0f / e2.sum;
    float q1[QM], q2[QM];      // softmax probabilities, evaluated once and reused by the dT partials below
    if (__ballot(lab_ok)) {
#pragma unroll
      for (int j = 0; j < QM; ++j) {
        q1[j] = (j < Q) ? exp_le0(v1[j] - e1.vmax) * inv1 : 0.f;
        q2[j] = (j < Q) ? exp_le0(v2[j] - e2.vmax) * inv2 : 0.f;
      }
    }
    if (lab_ok) {
#pragma unroll
      for (int j = 0; j < QM; ++j)
        if (j < Q) {
          r1 += q1[j] * sT[j * C + labi];
          r2 += q2[j] * sT[QC + j * C + labi];
        }
      acc[6] += -logf(r1);
      acc[7] += -logf(r2);
      acc[11] += 1.f;
    }
    // dL_y/dT partials: per wave, one label value at a time (labels are spatially coherent -> few rounds)
    // (tried and measured slower, round 2: staging q / r per wave as [pixel][j] and letting lane j walk the pixels -- with a register
    // run sum and scalar branches at label changes 455 us against 330, with one LDS float add per pixel 755: ds_add_f32 is slow)
    if (!(SIMT_HEAD_ABL & 1)) {
      unsigned long long todo = __ballot(lab_ok);
      const float ir1 = lab_ok ? 1.0f / r1 : 0.f, ir2 = lab_ok ? 1.0f / r2 : 0.f;
      while (todo) {
        int src = __ffsll((long long)todo) - 1;
        int c = __shfl(labi, src, 64);
        bool mine = lab_ok && labi == c;
        todo &= ~__ballot(mine);
#pragma unroll
        for (int j = 0; j < QM; ++j)
          if (j < Q) {
            float c1 = mine ? q1[j] * ir1 : 0.f;
            float c2 = mine ? q2[j] * ir2 : 0.f;
            c1 = wave_sum_dpp63(c1);
            c2 = wave_sum_dpp63(c2);
            if (lane == 63) {
              sdT[(wave * 2 + 0) * QC + j * C + c] += c1;
              sdT[(wave * 2 + 1) * QC + j * C + c] += c2;
            }
          }
      }
    }
    // ---- anchors: arg-max over all pixels of each channel's upsampled logit (first index), Exist masks (:375-384)
    if (a.mode == 0 && !(SIMT_HEAD_ABL & 2)) {
      unsigned long long ex1 = live ? (1ull << e1.arg) : 0ull, ex2 = live ? (1ull << e2.arg) : 0ull;
      ex1 = wave_or_u64(ex1);
      ex2 = wave_or_u64(ex2);
      if (lane == 0) { atomicOr(&sEx[0], ex1); atomicOr(&sEx[1], ex2); }
      // A channel's block-wide best so far (upper half of its key in LDS) is read once per group: lane t holds channel t of both heads.  A wave only reduces a channel when one of its pixels reaches that value (>=: ties go through the key,
      // whose low half prefers the smaller pixel index) -- after the first groups that is rare, and the 2 x Q wave reductions per 64
      // pixels were a fifth of this kernel.  A stale (lower) best only makes a wave do the reduction needlessly.
      // (compared as floats: one v_cmp against a scalar per channel; -0.0 >= +0.0 only sends a wave through the reduction needlessly)
      float bestf1 = -INFINITY, bestf2 = -INFINITY;
      if (lane < Q) {
        const unsigned b1 = (unsigned)(sKey[lane] >> 32), b2 = (unsigned)(sKey[QMAX + lane] >> 32);
        if (b1) bestf1 = f32_unord(b1);
        if (b2) bestf2 = f32_unord(b2);
      }
#pragma unroll
      for (int j = 0; j < QM; ++j)
        if (j < Q) {
          const float best1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bestf1), j));
          const float best2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bestf2), j));
          if (__ballot(live && v1[j] >= best1)) {
            const float a1 = live ? v1[j] : -INFINITY;
            float m1 = wave_max_f(a1);
            unsigned long long b1 = __ballot(live && a1 == m1);
            if (lane == 0 && b1) {
              long pp = grp * 256 + wave * 64 + (__ffsll((long long)b1) - 1);
              unsigned long long key = ((unsigned long long)f32_ord(m1) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)pp);
              atomicMax(&sKey[j], key);
            }
          }
          if (__ballot(live && v2[j] >= best2)) {
            const float a2 = live ? v2[j] : -INFINITY;
            float m2 = wave_max_f(a2);
            unsigned long long b2 = __ballot(live && a2 == m2);
            if (lane == 0 && b2) {
              long pp = grp * 256 + wave * 64 + (__ffsll((long long)b2) - 1);
              unsigned long long key = ((unsigned long long)f32_ord(m2) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned int)pp);
              atomicMax(&sKey[QMAX + j], key);
            }
          }
        }
    }
  }

  // ---- block reduction (fixed order) and publication
#pragma unroll
  for (int i = 0; i < NSCAL; ++i) {
    float s = wave_sum(acc[i]);
    if (lane == 0) sRed[wave * NSCAL + i] = s;
  }
  __syncthreads();
  float* part = a.part + (long)lbid * (NSCAL + 2 * QC);
  if (tid < NSCAL) part[tid] = sRed[tid] + sRed[NSCAL + tid] + sRed[2 * NSCAL + tid] + sRed[3 * NSCAL + tid];
  for (int i = tid; i < 2 * QC; i += 256)
    part[NSCAL + i] = sdT[i] + sdT[2 * QC + i] + sdT[4 * QC + i] + sdT[6 * QC + i];
  if (tid < 2 * QMAX && sKey[tid]) atomicMax(&a.keys[tid], sKey[tid]);
  if (tid < 2 && sEx[tid]) atomicOr(&a.keys[2 * QMAX + tid], sEx[tid]);
}

// --------------------------------------------------------------------------------------------------------
// finalize (one block)
// hout layout (floats):
//   [0] loss_p1 [1] loss_p2 [2] place1 [3] place2 [4] loss_y1 [5] loss_y2
//   [6] N_p [7] N_known1 [8] N_known2 [9] N_y [10] known1 [11] known2 [12] unk1 [13] unk2
//   [16            .. 16+QC)      anchor1 [Q][C]      [16+QC   .. 16+2QC)  anchor2
//   [16+2QC        .. +QMAX)      exist1 (0/1)        then exist2 [QMAX]
//   [16+2QC+2QMAX  .. +QMAX)      anchor pixel index1 (as float bits of int), then index2 [QMAX]
//   [16+2QC+4QMAX  .. +QC)        dTy1 = d(loss_y1)/dT1 (mean-normalised, unweighted), then dTy2
// --------------------------------------------------------------------------------------------------------
// Column sums of the block partials, 8 columns x 32 row lanes per block, four independent loads per thread and pass (fixed
// combination order, double precision): a single block walking 2048 rows took 2.5 ms at 4x768x768, 32 columns x 8 lanes 71 us.
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* part, int nblk, int stride, int ncols, double* sums) {
  __shared__ double red[32][8];
  const int cl = threadIdx.x & 7, r = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s = 0.0;
  if (c < ncols) {
    int b = r;
    for (; b + 96 < nblk; b += 128) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = part[(long)(b + 32 * u) * stride + c];
#pragma unroll
      for (int u = 0; u < 4; ++u) s += (double)v[u];
    }
    for (; b < nblk; b += 32) s += (double)part[(long)b * stride + c];
  }
  red[r][cl] = s;
  __syncthreads();
  if (r == 0 && c < ncols) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) t += red[q][cl];
    sums[c] = t;
  }
}

__device__ __forceinline__ double* head_sums(float* hout, int Q, int C) {
  // 8-byte aligned region behind the documented float layout of hout
  return (double*)(hout + ((16 + 4 * Q * C + 4 * QMAX + 1) & ~1));
}

__global__ __launch_bounds__(256) void head_finalize_kernel(HeadArgs a, int nblk) {
  const HeadGeom g = a.g;
  const int Q = g.Q, C = g.C, QC = Q * C;
  const int tid = threadIdx.x;
  const double* sc = head_sums(a.hout, Q, C);
  float* o = a.hout;
  const double Np = sc[8], Nk1 = sc[9], Nk2 = sc[10], Ny = sc[11];
  if (tid == 0) {
    o[0] = (float)(sc[0] / Np); o[1] = (float)(sc[1] / Np);
    float k1 = (float)(sc[2] / Nk1), k2 = (float)(sc[3] / Nk2), u1 = (float)(sc[4] / Nk1), u2 = (float)(sc[5] / Nk2);
    o[2] = k1 + a.lambda_place * u1; o[3] = k2 + a.lambda_place * u2;
    o[4] = (float)(sc[6] / Ny); o[5] = (float)(sc[7] / Ny);
    o[6] = (float)Np; o[7] = (float)Nk1; o[8] = (float)Nk2; o[9] = (float)Ny;
    o[10] = k1; o[11] = k2; o[12] = u1; o[13] = u2;
    o[14] = o[1] + a.lambda_seg * o[0];      // warm-up total: loss_seg2 + lambda_seg * loss_seg1 (trainV1_warmup.py:224)
    o[15] = (float)sc[12];                   // pixels whose label is out of range (neither a class nor 255): the reference raises
  }
  // dTy: -(1/Ny) * sum_p [label=c] q_j / r
  float* dTy = o + 16 + 2 * QC + 4 * QMAX;
  for (int i = tid; i < 2 * QC; i += 256) dTy[i] = (float)(-sc[NSCAL + i] / Ny);
  // anchors
  float* ex = o + 16 + 2 * QC;
  float* ai = ex + 2 * QMAX;
  const unsigned long long e1 = a.keys[2 * QMAX], e2 = a.keys[2 * QMAX + 1];
  if (tid < 2 * QMAX) {
    int hd = tid / QMAX, j = tid % QMAX;
    ex[tid] = (j < Q && (((hd ? e2 : e1) >> j) & 1ull)) ? 1.f : 0.f;
    unsigned long long key = a.keys[tid];
    int pidx = (j < Q) ? (int)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull)) : 0;
    ai[tid] = __int_as_float(pidx);
  }
  for (int i = tid; i < 2 * QC; i += 256) {
    if (a.mode != 0) { o[16 + i] = 0.f; continue; }
    int hd = i / QC, r = i % QC, j = r / C, c = r % C;
    unsigned long long key = a.keys[hd * QMAX + j];
    long p = (long)(0xFFFFFFFFu - (unsigned int)(key & 0xFFFFFFFFull));
    int x = (int)(p % g.W);
    long t = p / g.W;
    int y = (int)(t % g.H);
    int b = (int)(t / g.H);
    if (b >= g.B) { b = 0; y = 0; x = 0; }
    Taps tp = make_taps(g, b, y, x);
    const float* f = a.fixp;
    const float vc = lerp4(tp, f[(long)tp.o00 * g.ldf + c], f[(long)tp.o01 * g.ldf + c], f[(long)tp.o10 * g.ldf + c],
                           f[(long)tp.o11 * g.ldf + c]);
    if (!g.fix_logits) { o[16 + i] = vc; continue; }
    float mx = -INFINITY;                 // softmax of the interpolated logits at the anchor pixel (same order as fixed_posterior)
    for (int cc = 0; cc < C; ++cc)
      mx = fmaxf(mx, lerp4(tp, f[(long)tp.o00 * g.ldf + cc], f[(long)tp.o01 * g.ldf + cc], f[(long)tp.o10 * g.ldf + cc],
                           f[(long)tp.o11 * g.ldf + cc]));
    float sum = 0.f;
    for (int cc = 0; cc < C; ++cc)
      sum += expf(lerp4(tp, f[(long)tp.o00 * g.ldf + cc], f[(long)tp.o01 * g.ldf + cc], f[(long)tp.o10 * g.ldf + cc],
                        f[(long)tp.o11 * g.ldf + cc]) - mx);
    o[16 + i] = expf(vc - mx) / sum;
  }
}

// Weighted sum of up to 4*NB terms of one run [lo, hi) of a chunk's pixels, in pixel order, every LDS read requested before the first
// use.  mode 0: right-tap weights l1;  1: left-tap weights 1 - l1;  2: last column, whose right tap is the column itself.
template <int NB>
__device__ __forceinline__ float run_sum(const float* G, int GP, const float* sL1, int lo, int hi, int mode, float s) {
  if (SIMT_HEAD_ABL & 16) return s;
  float wv[NB * 4], gv[NB * 4];
#pragma unroll
  for (int u = 0; u < NB * 4; ++u) {
    const int pp = max(min(lo + u, hi - 1), 0);
    const float l1 = sL1[pp];
    const float w = mode == 0 ? l1 : mode == 2 ? (1.f - l1) + l1 : 1.f - l1;
    wv[u] = lo + u < hi ? w : 0.f;
    gv[u] = G[pp * GP];
  }
#pragma unroll
  for (int u = 0; u < NB * 4; ++u)
    if (wv[u] != 0.f) s += wv[u] * gv[u];           // (a zero weight skips the term, as in the scanning form)
  return s;
}

// --------------------------------------------------------------------------------------------------------
// pass 2: gradient w.r.t. the upsampled logits, reduced along x inside the block.  One block per group of a.rows consecutive image rows.
// --------------------------------------------------------------------------------------------------------
// rows > 1 (round 4; the production sizes run 6 or 8): the block also folds its rows along y -- (rows - 1) * sy < 1, so they touch at most
// THREE low-res rows, first = the low-res row of the group's first image row; each thread keeps its share of the three [2][w][Q] window rows
// in registers (P2_NACC x 3) and adds every finished row's x-reduced sums with the row's two interpolation weights.  g1 shrinks from
// H to 3 * H / rows rows (57 -> 28.6 MB at 4 x 768 x 768, written once, read once by the y-reduction), in a fixed order.
#define P2_NACC 18              // window entries per thread: 2 * w * Q <= 256 * P2_NACC (else rows = 1)
// launch bounds: at most 2 waves per SIMD are asked for.  The run-time-count builds need it (at 3 the QM = 24 build spilled 272 B per lane to
// scratch: round 1's "256 MB of HBM traffic per launch" against ~30 MB algorithmic); the compile-time-count builds come out at ~160
// VGPRs without spills.  At 4 x 768 x 768 on cold operands, pass 2 + y-reduction: 1 016 us (round 1) -> 826 (no spill) -> 680 (compile-time
// counts, one-instruction exp) -> 432 (run-based x-reduction).  Pass 1 keeps 3 waves per SIMD with a small spill (faster than 2 without).
template <int QM, int QT, int CT>
__global__ __launch_bounds__(256, 2) void head_pass2_kernel(HeadArgs a) {
  const HeadGeom g = a.g;
  const int Q = QT ? QT : g.Q, C = CT ? CT : g.C, QC = Q * C, QP = a.QP;
  constexpr int CM = CT ? (CT + 3) / 4 * 4 : QM;
  const int GP = Q + 1;  // LDS pitch of the per-pixel gradient rows
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sT = (float*)smem;          // [2][QC]
  float* sG = sT + 2 * QC;           // [2][256][GP]
  float* sAcc = sG + 2 * 256 * GP;   // [2][w][Q]
  int* sI0 = (int*)(sAcc + 2 * g.w * Q);     // [256] low-res column of each pixel of the chunk
  float* sL1 = (float*)(sI0 + 256);          // [256] its right-tap weight
  int* sStart = (int*)(sL1 + 256);           // [XR_MAX + 1] first pixel of the chunk whose low-res column is >= xl_lo + k
  const int tid = threadIdx.x;
  const int lbid = xcd_remap(blockIdx.x, gridDim.x);      // an XCD's blocks take consecutive image rows (L2 locality of the gathers; pass 1)
  // (the run-time-count builds keep one row per block: the window registers would spill there; head_rows_per_block)
  constexpr bool GROUPED = QT != 0;
  constexpr int NACC = GROUPED ? P2_NACC : 1;
  const int R = GROUPED ? a.rows : 1;
  const int b = (lbid * R) / g.H, y0 = (lbid * R) % g.H;   // (R divides H: a group never straddles two images)
  const int iy_first = min((int)src_coord(g.half, g.sy, y0), g.h - 1);
  float gacc[3][NACC];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < NACC; ++i) gacc[k][i] = 0.f;
  for (int i = tid; i < QC; i += 256) { sT[i] = a.mode == 0 ? a.T1[i] : 0.f; sT[QC + i] = a.mode == 0 ? a.T2[i] : 0.f; }
  const float* o = a.hout;
  const float Np = o[6], Nk1 = o[7], Nk2 = o[8], Ny = o[9];
  const float gs = a.gscale;
  const float gp1 = gs * a.lambda_seg / Np, gp2 = gs / Np;
  const float gk1 = gs * a.lambda_seg / Nk1, gk2 = gs / Nk2;
  const float gu1 = gk1 * a.lambda_place, gu2 = gk2 * a.lambda_place;
  const float gy1 = gs * a.lambda_seg / Ny, gy2 = gs / Ny;
  for (int rr = 0; rr < R; ++rr) {
  const int y = y0 + rr;
  for (int i = tid; i < 2 * g.w * Q; i += 256) sAcc[i] = 0.f;
  __syncthreads();

  for (int x0 = 0; x0 < g.W; x0 += 256) {
    const int x = x0 + tid;
    const bool live = x < g.W;
    {
      Taps tp = make_taps(g, b, y, live ? x : 0);
      {
        const float fx = src_coord(g.half, g.sx, x);
        int i0 = (int)fx;
        if (i0 > g.w - 1) i0 = g.w - 1;
        sI0[tid] = live ? i0 : -100;
        sL1[tid] = fx - (float)i0;
        // run starts of the (monotone) column sequence, for the x-reduction below: k in (column of x-1, column of x] start at this pixel
        const int xl_lo_ = max(0, (int)src_coord(g.half, g.sx, x0) - 1);
        const int nlive = min(256, g.W - x0);
        const int nxl_ = min(g.w - 1, (int)src_coord(g.half, g.sx, x0 + nlive - 1) + 2) - xl_lo_ + 1;
        if (live && nxl_ <= XR_MAX) {
          int prev = xl_lo_ - 1;
          if (tid) { prev = (int)src_coord(g.half, g.sx, x - 1); if (prev > g.w - 1) prev = g.w - 1; }
          for (int k = prev - xl_lo_ + 1; k <= i0 - xl_lo_; ++k) sStart[k] = tid;
          if (tid == nlive - 1)
            for (int k = i0 - xl_lo_ + 1; k <= nxl_; ++k) sStart[k] = nlive;
        }
      }
      // the confidence label and the checked noisy label: read back from pass 1 when the caller gave both byte maps (1 + 1 bytes per pixel
      // instead of the frozen posterior's 4 x C gathers + arg-max and the 8-byte label), else decided again
      const bool from_p1 = a.conf_out != nullptr && a.label_ws != nullptr;
      const long pix = ((long)b * g.H + y) * g.W + x;
      float fm = 0.f;
      int fa = 0;
      if (a.mode == 0 && !from_p1) fixed_posterior<CM>(g, C, a.fixp, tp, fm, fa);
      asm volatile("" ::: "memory");
      int conf = (fm > a.th_high) ? fa : 255;
      if (fm < a.th_low) conf = C;
      float v2[QM], v1[QM];
      interp_vec<QM>(a.pred2, g.ldp, Q, tp, v2);
      if (g.single) {
#pragma unroll
        for (int j = 0; j < QM; ++j) v1[j] = v2[j];
      } else {
        interp_vec<QM>(a.pred1, g.ldp, Q, tp, v1);
      }
      HeadEval e2, e1;
      eval_head<QM>(v2, Q, C, a.th_high, e2);
      eval_head<QM>(v1, Q, C, a.th_high, e1);
      if (conf == C) conf = (e2.arg >= C) ? e2.arg : 255;
      bool lab_ok;
      int labi;
      if (from_p1) {
        conf = live ? (int)a.conf_out[pix] : 255;
        const int l8 = live ? (int)a.label_ws[pix] : 255;
        lab_ok = l8 != 255;
        labi = lab_ok ? l8 : 0;
      } else {
        const long long lab = live ? a.label[pix] : 255;
        lab_ok = live && lab >= 0 && lab != 255 && lab < C;
        labi = lab_ok ? (int)lab : 0;
      }
      if (a.mode == 1) {
        if (!from_p1) conf = lab_ok ? labi : 255;
        e1.pseudo1 = 255;
        e2.pseudo1 = 255;
        lab_ok = false;
      }
      const float inv1 = 1.0f / e1.sum, inv2 = 1.0f / e2.sum;
      float r1 = 0.f, r2 = 0.f;
      if (lab_ok) {
#pragma unroll
        for (int j = 0; j < QM; ++j)
          if (j < Q) {
            r1 += (exp_le0(v1[j] - e1.vmax) * inv1) * sT[j * C + labi];
            r2 += (exp_le0(v2[j] - e2.vmax) * inv2) * sT[QC + j * C + labi];
          }
      }
      const float is1 = 1.0f / e1.sum2, is2 = 1.0f / e2.sum2;
      const float ir1 = lab_ok ? 1.0f / r1 : 0.f, ir2 = lab_ok ? 1.0f / r2 : 0.f;
#pragma unroll
      for (int j = 0; j < QM; ++j)
        if (j < Q) {
          float q1 = exp_le0(v1[j] - e1.vmax) * inv1, q2 = exp_le0(v2[j] - e2.vmax) * inv2;
          float G1 = 0.f, G2 = 0.f;
          if (live && !(SIMT_HEAD_ABL & 8)) {
            if (conf != 255) {
              float oh = (j == conf) ? 1.f : 0.f;
              G1 += gp1 * (q1 - oh);
              G2 += gp2 * (q2 - oh);
            }
            if (e1.pseudo1 != 255) {
              G1 += gk1 * (q1 - ((j == e1.arg) ? 1.f : 0.f));
              if (j != e1.arg) G1 += gu1 * (exp_le0(v1[j] - e1.m2) * is1 - ((j == e1.yopen) ? 1.f : 0.f));
            }
            if (e2.pseudo1 != 255) {
              G2 += gk2 * (q2 - ((j == e2.arg) ? 1.f : 0.f));
              if (j != e2.arg) G2 += gu2 * (exp_le0(v2[j] - e2.m2) * is2 - ((j == e2.yopen) ? 1.f : 0.f));
            }
            if (lab_ok) {                  // (a reciprocal per pixel, not two divisions per class)
              G1 += gy1 * (q1 - q1 * sT[j * C + labi] * ir1);
              G2 += gy2 * (q2 - q2 * sT[QC + j * C + labi] * ir2);
            }
          }
          if (!(SIMT_HEAD_ABL & 32)) {
          sG[(0 * 256 + tid) * GP + j] = G1;
          sG[(1 * 256 + tid) * GP + j] = G2;
          } else { asm volatile("" :: "v"(G1), "v"(G2)); }
        }
    }
    __syncthreads();
    // x-reduction: out[hd][xl][j] += sum_x wgt(x, xl) * G[hd][x][j]  -- only the low-res columns this 256-pixel chunk
    // can touch; the tap (i0, l1) of every pixel was computed once by its thread above
    const int xend = min(x0 + 256, g.W);
    const int xl_lo = max(0, (int)src_coord(g.half, g.sx, x0) - 1);
    const int xl_hi = min(g.w - 1, (int)src_coord(g.half, g.sx, xend - 1) + 2);
    const int nxl = xl_hi - xl_lo + 1;
    if (SIMT_HEAD_ABL & 4) {
    } else if (nxl <= XR_MAX) {
      // Pixels whose left tap is column xl form one run of the chunk ([sStart[xr], sStart[xr+1])), the ones whose right tap is xl the
      // run before it: two short weighted sums instead of a scan over every pixel that could touch the column.  A wave takes one
      // column, its lanes the (head, channel) pairs: run bounds, clamps and masks are wave-uniform (scalar unit), the weights broadcast
      // LDS reads, the gradient rows contiguous reads at immediate offsets, every term of a run requested before the first use.
      // Same terms in the same order as the scanning form below (bit-identical).
      const int wv_ = tid >> 6, ln = tid & 63;
      for (int xr = wv_; xr < nxl; xr += 4) {
        const int xl = xl_lo + xr;
        const int a0 = __builtin_amdgcn_readfirstlane(sStart[xr]), a1 = __builtin_amdgcn_readfirstlane(sStart[xr + 1]);
        const int b0 = xr ? __builtin_amdgcn_readfirstlane(sStart[xr - 1]) : a0;
        const bool edge = xl == g.w - 1;          // the right tap of the last column is the column itself
        const int rlen = max(a0 - b0, a1 - a0);
        const int rb = rlen <= 8 ? 2 : rlen <= 12 ? 3 : 0;     // runs of at most 4 * rb pixels: every term requested at once (0: looped)
        for (int L = ln; L < 2 * Q; L += 64) {
          const int hd = L >= Q ? 1 : 0, j = L - hd * Q;
          const float* G = sG + hd * 256 * GP + j;
          float s = 0.f;
          if (rb == 2) { s = run_sum<2>(G, GP, sL1, b0, a0, 0, s); s = run_sum<2>(G, GP, sL1, a0, a1, edge ? 2 : 1, s); }
          else if (rb == 3) { s = run_sum<3>(G, GP, sL1, b0, a0, 0, s); s = run_sum<3>(G, GP, sL1, a0, a1, edge ? 2 : 1, s); }
          else {
            for (int p = b0; p < a0; p += 4) s = run_sum<1>(G, GP, sL1, p, a0, 0, s);
            for (int p = a0; p < a1; p += 4) s = run_sum<1>(G, GP, sL1, p, a1, edge ? 2 : 1, s);
          }
          sAcc[(hd * g.w + xl) * Q + j] += s;
        }
      }
    } else
    for (int idx = tid; idx < 2 * nxl * Q; idx += 256) {
      int hd = idx / (nxl * Q);
      int r = idx - hd * nxl * Q;
      int xr = r / Q, j = r - xr * Q;
      int xl = xl_lo + xr;
      int lo, hi;
      dst_range(xl, g.W, g.sx, lo, hi);
      lo = max(lo, x0);
      hi = min(hi, xend - 1);
      float s = 0.f;
      for (int xx = lo; xx <= hi; ++xx) {
        const int i0 = sI0[xx - x0];
        const float l1 = sL1[xx - x0];
        const int i1 = i0 + (i0 < g.w - 1 ? 1 : 0);
        const float wgt = (i0 == xl ? 1.f - l1 : 0.f) + (i1 == xl ? l1 : 0.f);
        if (wgt != 0.f) s += wgt * sG[(hd * 256 + (xx - x0)) * GP + j];
      }
      sAcc[(hd * g.w + xl) * Q + j] += s;
    }
    __syncthreads();
  }
  if (!GROUPED || R == 1) {
    // write g1[hd][b][y][xl][0..QP)
    for (int idx = tid; idx < 2 * g.w * QP; idx += 256) {
      int hd = idx / (g.w * QP);
      int r = idx - hd * g.w * QP;
      int xl = r / QP, j = r - xl * QP;
      float v = (j < Q) ? sAcc[(hd * g.w + xl) * Q + j] : 0.f;
      a.g1[((((long)hd * g.B + b) * g.H + y) * g.w + xl) * QP + j] = v;
    }
  } else {
    // fold the finished row into the window: its two taps (i0, i1) carry (1 - l1, l1), like head_yreduce_kernel weighs a row
    const float fy = src_coord(g.half, g.sy, y);
    int i0 = (int)fy;
    if (i0 > g.h - 1) i0 = g.h - 1;
    const int i1 = i0 + (i0 < g.h - 1 ? 1 : 0);
    const float l1 = fy - (float)i0, l0 = 1.f - l1;
    float wk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) wk[k] = (i0 == iy_first + k ? l0 : 0.f) + (i1 == iy_first + k ? l1 : 0.f);
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 2 * g.w * Q) {
        const float sv = sAcc[idx];
#pragma unroll
        for (int k = 0; k < 3; ++k) gacc[k][i] += wk[k] * sv;
      }
    }
    __syncthreads();       // (sAcc is zeroed for the next row)
  }
  }
  if (GROUPED && R > 1) {
    // g1[hd][b][group][k][xl][0..Q): the three window rows of this group (pad columns Q..QP are never read)
    const int NG = g.H / R, gy = y0 / R;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 2 * g.w * Q) {
        const int hd = idx / (g.w * Q);
        const int r = idx - hd * g.w * Q;
        const int xl = r / Q, j = r - xl * Q;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          a.g1[(((((long)hd * g.B + b) * NG + gy) * 3 + k) * g.w + xl) * QP + j] = gacc[k][i];
      }
    }
  }
}

// y-reduction: d[hd][b][yl][xl][j] = sum_y wgt(y, yl) * g1[hd][b][y][xl][j]
template <typename T>
__global__ void head_yreduce_kernel(const float* g1, float* d32_1, float* d32_2, T* dT_1, T* dT_2, HeadGeom g, int QP,
                                    int ldo32, int ldoT, long total, int R) {
  long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int j = (int)(idx % QP);
  long t = idx / QP;
  int xl = (int)(t % g.w); t /= g.w;
  int yl = (int)(t % g.h); t /= g.h;
  int b = (int)(t % g.B);
  int hd = (int)(t / g.B);
  if (g.single && hd == 0) return;
  int lo, hi;
  dst_range(yl, g.H, g.sy, lo, hi);
  float s = 0.f;
  if (R > 1) {
    // pass 2 folded its groups of R rows already: add the window rows that are low-res row yl, in ascending group order
    const int NG = g.H / R;
    if (j < g.Q)
      for (int gy = lo / R; gy <= hi / R; ++gy) {
        const int k = yl - min((int)src_coord(g.half, g.sy, gy * R), g.h - 1);
        if (k >= 0 && k < 3) s += g1[(((((long)hd * g.B + b) * NG + gy) * 3 + k) * g.w + xl) * QP + j];
      }
  } else
  for (int yy = lo; yy <= hi; ++yy) {
    float fy = src_coord(g.half, g.sy, yy);
    int i0 = (int)fy;
    if (i0 > g.h - 1) i0 = g.h - 1;
    int i1 = i0 + (i0 < g.h - 1 ? 1 : 0);
    float l1 = fy - (float)i0, l0 = 1.f - l1;
    float wgt = (i0 == yl ? l0 : 0.f) + (i1 == yl ? l1 : 0.f);
    if (wgt != 0.f) s += wgt * g1[((((long)hd * g.B + b) * g.H + yy) * g.w + xl) * QP + j];
  }
  long m = ((long)b * g.h + yl) * g.w + xl;
  float* d32 = hd ? d32_2 : d32_1;
  T* dT = hd ? dT_2 : dT_1;
  if (d32 && j < ldo32) d32[m * ldo32 + j] = s;
  if (dT && j < ldoT) Elem<T>::st(dT + m * ldoT + j, s);
}

static size_t pass1_lds(int Q, int C) {
  size_t QC = (size_t)Q * C;
  return (2 * QC + 8 * QC) * 4 + 8 + (2 * QMAX + 2) * 8 + 4 * NSCAL * 4 + 16;
}
static size_t pass2_lds(int Q, int C, int w) {
  size_t QC = (size_t)Q * C;
  return (2 * QC + 2 * 256 * (size_t)(Q + 1) + 2 * (size_t)w * Q + 512 + XR_MAX + 1) * 4;
}

// image rows per pass-2 block: the largest R <= 8 that divides H, keeps a group inside three low-res rows ((R - 1) * sy <= 0.95) and still
// leaves two blocks per CU; 1 when the window does not fit the per-thread registers
static int head_rows_per_block(const simt_head_desc* d, float sy) {
  const bool fixed_counts = d->C == 19 && (d->Q == 22 || d->Q == 25);      // the compile-time-count instantiations (simt_head_grad's dispatch)
  if (!fixed_counts || 2l * d->w * d->Q > 256l * P2_NACC) return 1;
  for (int R = 8; R >= 3; --R)        // (R >= 3: the three window rows per group must fit the caller's [H]-row workspace)
    if (d->H % R == 0 && (float)(R - 1) * sy <= 0.95f && (long)d->B * d->H / R >= 512) return R;
  return 1;
}

static int fill_args(const simt_head_desc* d, HeadArgs& a) {
  SIMT_CHECK(d && d->pred2 && d->label && d->part && d->keys && d->hout);
  SIMT_CHECK(d->single ? (d->mode == 0) : (d->pred1 != nullptr));
  SIMT_CHECK(d->mode == 1 || (d->fixp && d->T2 && (d->single || d->T1)));
  SIMT_CHECK(d->Q <= QMAX && d->C < d->Q + 1 && d->C >= 1 && d->Q <= 64);
  SIMT_CHECK(d->ldp % 4 == 0 && d->ldf % 4 == 0 && d->ldp >= ((d->Q + 3) / 4) * 4 && d->ldf >= ((d->C + 3) / 4) * 4);
  SIMT_CHECK((long)d->B * d->H * d->W < 0xFFFFFFFFl);
  a.g.B = d->B; a.g.h = d->h; a.g.w = d->w; a.g.H = d->H; a.g.W = d->W; a.g.C = d->C; a.g.Q = d->Q;
  a.g.ldp = d->ldp; a.g.ldf = d->ldf;
  a.g.half = d->up_half_pixel ? 1 : 0; a.g.fix_logits = d->fix_logits ? 1 : 0; a.g.single = d->single ? 1 : 0;
  if (a.g.half) {
    a.g.sy = (float)d->h / (float)d->H;
    a.g.sx = (float)d->w / (float)d->W;
  } else {
    a.g.sy = d->H > 1 ? (float)(d->h - 1) / (float)(d->H - 1) : 0.f;
    a.g.sx = d->W > 1 ? (float)(d->w - 1) / (float)(d->W - 1) : 0.f;
  }
  a.pred1 = d->single ? d->pred2 : d->pred1; a.pred2 = d->pred2; a.fixp = d->fixp; a.label = (const long long*)d->label;
  a.T1 = d->single ? d->T2 : d->T1; a.T2 = d->T2;
  a.th_high = d->th_high; a.th_low = d->th_low; a.lambda_seg = d->lambda_seg; a.lambda_place = d->lambda_place;
  a.part = d->part; a.keys = (unsigned long long*)d->keys; a.hout = d->hout; a.g1 = d->g1; a.QP = d->QP;
  a.gscale = d->gscale;
  a.rows = head_rows_per_block(d, a.g.sy);
  a.mode = d->mode;
  a.conf_out = d->conf_out;
  a.label_ws = d->label_ws;
  SIMT_CHECK(d->mode == 0 || d->mode == 1);
  return SIMT_OK;
}

extern "C" int simt_head_nblk(int B, int H, int W) {
  long P = (long)B * H * W;
  long n = (P + 255) / 256;
  if (n > 512) n = 512;      // two blocks per CU, one round (round 4: 2048 -> 512 blocks; the partial-sum rows the reduction reads shrink 4x)
  return (int)n;
}
extern "C" int simt_head_part_floats(int Q, int C) { return NSCAL + 2 * Q * C; }
extern "C" int simt_head_hout_floats(int Q, int C) {
  return ((16 + 4 * Q * C + 4 * QMAX + 1) & ~1) + 2 * (NSCAL + 2 * Q * C);   // + the double-precision column sums
}
extern "C" int simt_head_keys_count(void) { return 2 * QMAX + 2; }

// losses (pass 1 + finalize).  keys must be zeroed by this call: done here with a memset node on the stream.
extern "C" int simt_head_loss(const simt_head_desc* d, simt_stream_t stream) {
  HeadArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = simt_head_nblk(d->B, d->H, d->W);
  (void)hipMemsetAsync(d->keys, 0, (2 * QMAX + 2) * sizeof(unsigned long long), st);
  size_t lds1 = pass1_lds(d->Q, d->C);
  SIMT_CHECK(lds1 <= 160 * 1024);
  static SimtLdsAttrCache lds1_cache;
  if (simt_lds_attr_needed(&lds1_cache, lds1)) {
#define P1ATTR(QM, QT, CT) (void)hipFuncSetAttribute((const void*)head_pass1_kernel<QM, QT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1)
    P1ATTR(24, 22, 19); P1ATTR(28, 25, 19); P1ATTR(24, 0, 0); P1ATTR(QMAX, 0, 0);
#undef P1ATTR
  }
#define P1(QM, QT, CT) hipLaunchKernelGGL((head_pass1_kernel<QM, QT, CT>), dim3(nblk), dim3(256), lds1, st, a)
  if (d->Q == 22 && d->C == 19) P1(24, 22, 19);          // Cityscapes, K = 3 open classes (BASELINE configs[0..2], [4])
  else if (d->Q == 25 && d->C == 19) P1(28, 25, 19);     // K = 6 (configs[3])
  else if (d->Q <= 24) P1(24, 0, 0);
  else P1(QMAX, 0, 0);
#undef P1
  SIMT_LAUNCH_CHECK();
  {
    const int ncols = NSCAL + 2 * d->Q * d->C;
    double* sums = (double*)(d->hout + ((16 + 4 * d->Q * d->C + 4 * QMAX + 1) & ~1));
    hipLaunchKernelGGL(head_reduce_kernel, dim3((ncols + 7) / 8), dim3(256), 0, st, d->part, nblk, ncols, ncols, sums);
    SIMT_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(head_finalize_kernel, dim3(1), dim3(256), 0, st, a, nblk);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// gradients w.r.t. the low-res logits (pass 2 + y reduction); needs hout from simt_head_loss of the same inputs.
extern "C" int simt_head_grad(const simt_head_desc* d, simt_stream_t stream) {
  HeadArgs a;
  int rc = fill_args(d, a);
  if (rc) return rc;
  SIMT_CHECK(d->g1 && d->QP >= d->Q);
  hipStream_t st = (hipStream_t)stream;
  size_t lds2 = pass2_lds(d->Q, d->C, d->w);
  SIMT_CHECK(lds2 <= 160 * 1024);
  static SimtLdsAttrCache lds2_cache;
  if (simt_lds_attr_needed(&lds2_cache, lds2)) {
#define P2ATTR(QM, QT, CT) (void)hipFuncSetAttribute((const void*)head_pass2_kernel<QM, QT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)
    P2ATTR(24, 22, 19); P2ATTR(28, 25, 19); P2ATTR(24, 0, 0); P2ATTR(QMAX, 0, 0);
#undef P2ATTR
  }
#define P2(QM, QT, CT) hipLaunchKernelGGL((head_pass2_kernel<QM, QT, CT>), dim3(d->B * d->H / a.rows), dim3(256), lds2, st, a)
  if (d->Q == 22 && d->C == 19) P2(24, 22, 19);
  else if (d->Q == 25 && d->C == 19) P2(28, 25, 19);
  else if (d->Q <= 24) P2(24, 0, 0);
  else P2(QMAX, 0, 0);
#undef P2
  SIMT_LAUNCH_CHECK();
  long total = 2l * d->B * d->h * d->w * d->QP;
  unsigned grid = (unsigned)((total + 255) / 256);
  if (d->grad_dtype == SIMT_BF16)
    hipLaunchKernelGGL(head_yreduce_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, d->g1, d->dpred1_f32, d->dpred2_f32,
                       (bf16_t*)d->dpred1_t, (bf16_t*)d->dpred2_t, a.g, d->QP, d->ld_f32, d->ld_t, total, a.rows);
  else
    hipLaunchKernelGGL(head_yreduce_kernel<float>, dim3(grid), dim3(256), 0, st, d->g1, d->dpred1_f32, d->dpred2_f32,
                       (float*)d->dpred1_t, (float*)d->dpred2_t, a.g, d->QP, d->ld_f32, d->ld_t, total, a.rows);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
