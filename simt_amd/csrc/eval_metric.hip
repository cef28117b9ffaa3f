// Evaluation path of the reference (tools/evaluate_cityscapes.py:96-162, evaluate_simt): logits of the main head at two
// input scales are bilinearly upsampled (align_corners=True) to the label resolution, SUMMED, arg-maxed, and scored with
// a confusion histogram (fast_hist :81-83) -> per-class IoU / mIoU (:86-87).  The reference does the sum / argmax /
// bincount on the CPU in numpy per image; here one fused kernel gathers the 4 taps of both low-res maps (L2-resident),
// and the histogram is integer atomics (exact, order-independent).
#include "common.h"

struct EvalArgs {
  const float* la;     // [B][ha][wa][lda] logits, scale A (first C channels used)
  const float* lb;     // [B][hb][wb][ldb] logits, scale B, or NULL
  int* pred;           // [B][H][W] arg-max class
  int B, ha, wa, lda, hb, wb, ldb, H, W, C;
  float sya, sxa, syb, sxb;
};

__device__ __forceinline__ void bil_taps(int y, int x, int h, int w, float sy, float sx, int& o00, int& o01, int& o10, int& o11,
                                         float& wy0, float& wy1, float& wx0, float& wx1) {
  // ATen upsample_bilinear2d, align_corners=True: src = scale * dst, scale = (in-1)/(out-1)
  const float fy = sy * (float)y, fx = sx * (float)x;
  int iy0 = (int)fy, ix0 = (int)fx;
  if (iy0 > h - 1) iy0 = h - 1;
  if (ix0 > w - 1) ix0 = w - 1;
  const int iy1 = iy0 + (iy0 < h - 1 ? 1 : 0), ix1 = ix0 + (ix0 < w - 1 ? 1 : 0);
  wy1 = fy - (float)iy0; wy0 = 1.f - wy1;
  wx1 = fx - (float)ix0; wx0 = 1.f - wx1;
  o00 = iy0 * w + ix0; o01 = iy0 * w + ix1; o10 = iy1 * w + ix0; o11 = iy1 * w + ix1;
}

__global__ __launch_bounds__(256) void upsample_sum_argmax_kernel(EvalArgs a) {
  const long P = (long)a.B * a.H * a.W;
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const int x = (int)(p % a.W);
    const long t = p / a.W;
    const int y = (int)(t % a.H);
    const int b = (int)(t / a.H);
    int a00, a01, a10, a11, b00 = 0, b01 = 0, b10 = 0, b11 = 0;
    float ay0, ay1, ax0, ax1, by0 = 0, by1 = 0, bx0 = 0, bx1 = 0;
    bil_taps(y, x, a.ha, a.wa, a.sya, a.sxa, a00, a01, a10, a11, ay0, ay1, ax0, ax1);
    const float* pa = a.la + (long)b * a.ha * a.wa * a.lda;
    const float* pb = nullptr;
    if (a.lb) {
      bil_taps(y, x, a.hb, a.wb, a.syb, a.sxb, b00, b01, b10, b11, by0, by1, bx0, bx1);
      pb = a.lb + (long)b * a.hb * a.wb * a.ldb;
    }
    float best = -INFINITY;
    int arg = 0;
    for (int c = 0; c < a.C; ++c) {
      // same association as ATen: h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11); then numpy's a + b
      float v = ay0 * (ax0 * pa[(long)a00 * a.lda + c] + ax1 * pa[(long)a01 * a.lda + c]) +
                ay1 * (ax0 * pa[(long)a10 * a.lda + c] + ax1 * pa[(long)a11 * a.lda + c]);
      if (pb) {
        const float u = by0 * (bx0 * pb[(long)b00 * a.ldb + c] + bx1 * pb[(long)b01 * a.ldb + c]) +
                        by1 * (bx0 * pb[(long)b10 * a.ldb + c] + bx1 * pb[(long)b11 * a.ldb + c]);
        v = v + u;
      }
      if (v > best) { best = v; arg = c; }     // first index on ties, like np.argmax
    }
    a.pred[p] = arg;
  }
}

extern "C" int simt_upsample_sum_argmax(const float* la, int ha, int wa, int lda, const float* lb, int hb, int wb, int ldb,
                                        int B, int H, int W, int C, int32_t* pred, simt_stream_t stream) {
  SIMT_CHECK(la && pred && B > 0 && C > 0 && C <= lda && (!lb || C <= ldb));
  EvalArgs a;
  a.la = la; a.lb = lb; a.pred = pred; a.B = B; a.ha = ha; a.wa = wa; a.lda = lda; a.hb = hb; a.wb = wb; a.ldb = ldb;
  a.H = H; a.W = W; a.C = C;
  a.sya = H > 1 ? (float)(ha - 1) / (float)(H - 1) : 0.f;
  a.sxa = W > 1 ? (float)(wa - 1) / (float)(W - 1) : 0.f;
  a.syb = (lb && H > 1) ? (float)(hb - 1) / (float)(H - 1) : 0.f;
  a.sxb = (lb && W > 1) ? (float)(wb - 1) / (float)(W - 1) : 0.f;
  long P = (long)B * H * W;
  long grid = (P + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(upsample_sum_argmax_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// hist[n*gt + pred] += 1 for 0 <= gt < n   (fast_hist: labels outside [0, n) -- the 255 "ignore" id -- are skipped)
__global__ __launch_bounds__(256) void confusion_hist_kernel(const int64_t* gt, const int32_t* pred, long P, int n,
                                                             unsigned long long* hist) {
  __shared__ unsigned int sh[32 * 32];
  const int nn = n * n;
  for (int i = threadIdx.x; i < nn; i += 256) sh[i] = 0u;
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const long long g = gt[p];
    const int q = pred[p];
    if (g >= 0 && g < n && q >= 0 && q < n) atomicAdd(&sh[(int)g * n + q], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nn; i += 256)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

extern "C" int simt_confusion_hist(const int64_t* gt, const int32_t* pred, long P, int n, int64_t* hist,
                                   simt_stream_t stream) {
  SIMT_CHECK(gt && pred && hist && n > 0 && n <= 32);
  long grid = (P + 255) / 256;
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(confusion_hist_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, gt, pred, P, n,
                     (unsigned long long*)hist);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---- F.interpolate(mode='bilinear') of NHWC low-res maps to an NCHW fp32 tensor and its adjoint.
// model/deeplabv3.py:137 upsamples the logits INSIDE the model with the default align_corners=False
// (src = (dst + 0.5) * in/out - 0.5, clamped at 0); align_corners=True is the interp_target flavour of trainV2_simt.py:301.
struct UpArgs {
  const float* src;   // [B][h][w][lds]
  float* dst;         // [B][C][H][W]
  int B, h, w, lds, C, H, W, align;
  float sy, sx;
};
__device__ __forceinline__ void up_taps(int d, int in, float scale, int align, int& i0, int& i1, float& l0, float& l1) {
  float f = align ? scale * (float)d : fmaxf(((float)d + 0.5f) * scale - 0.5f, 0.f);
  i0 = (int)f;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = f - (float)i0;
  l0 = 1.f - l1;
}
__global__ __launch_bounds__(256) void upsample_nchw_kernel(UpArgs a) {
  const long total = (long)a.B * a.C * a.H * a.W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % a.W);
    long t = i / a.W;
    const int y = (int)(t % a.H); t /= a.H;
    const int c = (int)(t % a.C);
    const int b = (int)(t / a.C);
    int y0, y1, x0, x1; float wy0, wy1, wx0, wx1;
    up_taps(y, a.h, a.sy, a.align, y0, y1, wy0, wy1);
    up_taps(x, a.w, a.sx, a.align, x0, x1, wx0, wx1);
    const float* p = a.src + (long)b * a.h * a.w * a.lds + c;
    a.dst[i] = wy0 * (wx0 * p[((long)y0 * a.w + x0) * a.lds] + wx1 * p[((long)y0 * a.w + x1) * a.lds]) +
               wy1 * (wx0 * p[((long)y1 * a.w + x0) * a.lds] + wx1 * p[((long)y1 * a.w + x1) * a.lds]);
  }
}
// adjoint: dsrc[b][yl][xl][c] = sum over (y, x) of w(y,yl) * w(x,xl) * ddst[b][c][y][x]   (gather form, deterministic).
// Separable: pass 1 folds x (tmp[b][c][y][xl] = sum_x w(x,xl) ddst[b][c][y][x]), pass 2 folds y -- the one-pass version
// evaluated ~2 700 weight pairs per low-res element (0.67 ms at 4 x 25 x 512 x 1024).
__device__ __forceinline__ void up_range(int l, int in, int out, float scale, int& lo, int& hi) {
  if (scale > 0.f) {
    const float r = 1.f / scale;
    lo = (int)floorf(((float)l - 1.f) * r) - 2;
    hi = (int)ceilf(((float)l + 2.f) * r) + 2;
  } else { lo = 0; hi = out - 1; }
  lo = max(lo, 0); hi = min(hi, out - 1);
}
__global__ __launch_bounds__(256) void upsample_bwd_x_kernel(const float* ddst, float* tmp, UpArgs a) {
  const long total = (long)a.B * a.C * a.H * a.w;          // tmp[b][c][y][xl]
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xl = (int)(i % a.w);
    const long row = i / a.w;                              // (b*C + c)*H + y
    int xlo, xhi;
    up_range(xl, a.w, a.W, a.sx, xlo, xhi);
    const float* g = ddst + row * a.W;
    float s = 0.f;
    for (int x = xlo; x <= xhi; ++x) {
      int x0, x1; float wx0, wx1;
      up_taps(x, a.w, a.sx, a.align, x0, x1, wx0, wx1);
      const float wx = (x0 == xl ? wx0 : 0.f) + (x1 == xl ? wx1 : 0.f);
      s += wx * g[x];
    }
    tmp[i] = s;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_y_kernel(const float* tmp, T* dsrc, UpArgs a) {
  const long total = (long)a.B * a.h * a.w * a.C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % a.C);
    long t = i / a.C;
    const int xl = (int)(t % a.w); t /= a.w;
    const int yl = (int)(t % a.h);
    const int b = (int)(t / a.h);
    int ylo, yhi;
    up_range(yl, a.h, a.H, a.sy, ylo, yhi);
    const float* g = tmp + ((long)(b * a.C + c) * a.H) * a.w + xl;
    float s = 0.f;
    for (int y = ylo; y <= yhi; ++y) {
      int y0, y1; float wy0, wy1;
      up_taps(y, a.h, a.sy, a.align, y0, y1, wy0, wy1);
      const float wy = (y0 == yl ? wy0 : 0.f) + (y1 == yl ? wy1 : 0.f);
      s += wy * g[(long)y * a.w];
    }
    Elem<T>::st(dsrc + ((long)(b * a.h + yl) * a.w + xl) * a.lds + c, s);
  }
}
static void fill_up(UpArgs& a, int B, int h, int w, int lds, int C, int H, int W, int align) {
  a.B = B; a.h = h; a.w = w; a.lds = lds; a.C = C; a.H = H; a.W = W; a.align = align;
  if (align) {
    a.sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
    a.sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  } else {
    a.sy = (float)h / (float)H;
    a.sx = (float)w / (float)W;
  }
}
extern "C" int simt_upsample_nchw(const float* src, int B, int h, int w, int lds, int C, int H, int W, int align_corners, float* dst,
                                  simt_stream_t stream) {
  SIMT_CHECK(src && dst && C <= lds);
  UpArgs a; a.src = src; a.dst = dst;
  fill_up(a, B, h, w, lds, C, H, W, align_corners);
  long total = (long)B * C * H * W, grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  hipLaunchKernelGGL(upsample_nchw_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
// tmp: caller-owned scratch of B*C*H*w floats for the separable adjoint (allocated with the plan's other buffers: no allocation on
// the launch path, nothing shared between plans / streams / devices)
extern "C" int simt_upsample_nchw_bwd(const float* ddst, int B, int h, int w, int lds, int C, int H, int W, int align_corners,
                                      void* dsrc, int dtype, float* tmp, simt_stream_t stream) {
  SIMT_CHECK(ddst && dsrc && tmp && C <= lds);
  UpArgs a; a.src = nullptr; a.dst = nullptr;
  fill_up(a, B, h, w, lds, C, H, W, align_corners);
  float* g_upbwd_tmp = tmp;
  long t1 = (long)B * C * H * w, g1 = (t1 + 255) / 256;
  if (g1 > 256 * 32) g1 = 256 * 32;
  hipLaunchKernelGGL(upsample_bwd_x_kernel, dim3((unsigned)g1), dim3(256), 0, (hipStream_t)stream, ddst, g_upbwd_tmp, a);
  SIMT_LAUNCH_CHECK();
  long total = (long)B * h * w * C, grid = (total + 255) / 256;
  if (grid > 256 * 16) grid = 256 * 16;
  if (dtype == SIMT_BF16)
    hipLaunchKernelGGL(upsample_bwd_y_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, g_upbwd_tmp, (bf16_t*)dsrc, a);
  else
    hipLaunchKernelGGL(upsample_bwd_y_kernel<float>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, g_upbwd_tmp, (float*)dsrc, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---- offline NTM utilities (tools/compute_ClassDistribution.py:49-51,66-86; tools/compute_ConfusionMatrix.py:54-56,68-98) ----------
// hist[na_idx * nb + b] += 1 over uint8 label images: a = row class (optional 256-entry LUT = label_mapping; NULL a -> row 0, i.e. the
// 1-D class histogram of compute_CD), b = column class.  Entries with a (after the LUT) >= na or b >= nb are skipped: 255 = ignore.
// Integer atomics -> exact and order independent; LDS histogram per block (na * nb <= 2048), one global atomic per non-zero bin.
__global__ __launch_bounds__(256) void hist2d_u8_kernel(const unsigned char* __restrict__ a, const unsigned char* __restrict__ b, long P,
                                                        int na, int nb, const unsigned char* __restrict__ lut,
                                                        unsigned long long* __restrict__ hist) {
  __shared__ unsigned int sh[2048];
  __shared__ unsigned char sl[256];
  const int nn = na * nb;
  for (int i = threadIdx.x; i < nn; i += 256) sh[i] = 0u;
  sl[threadIdx.x] = lut ? lut[threadIdx.x] : (unsigned char)threadIdx.x;
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
    const int r = a ? (int)sl[a[p]] : 0;
    const int c = (int)b[p];
    if (r < na && c < nb) atomicAdd(&sh[r * nb + c], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nn; i += 256)
    if (sh[i]) atomicAdd(&hist[i], (unsigned long long)sh[i]);
}

extern "C" int simt_hist2d_u8(const unsigned char* a, const unsigned char* b, long P, int na, int nb, const unsigned char* lut,
                              int64_t* hist, simt_stream_t stream) {
  SIMT_CHECK(b && hist && P >= 0 && na >= 1 && nb >= 1 && na * nb <= 2048 && (a || na == 1));
  if (P == 0) return SIMT_OK;
  long grid = (P + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(hist2d_u8_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a, b, P, na, nb, lut,
                     (unsigned long long*)hist);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
