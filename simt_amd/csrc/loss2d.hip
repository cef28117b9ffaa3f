// utils/loss.py of the reference as HIP kernels: CrossEntropy2d (masked CE / NLL over NCHW predictions) and EntropyLoss.
//
// Replaces utils/loss.py:6-40 (CrossEntropy2d.forward: boolean-mask gather + F.cross_entropy / log + F.nll_loss, mean
// over valid pixels, optional per-class weight) and :42-49 (EntropyLoss: mean pixel entropy of softmax(x, dim=1)).
// One thread per pixel walks the C channels of its NCHW column (coalesced across x); block partials are combined in
// fixed order in double precision (bitwise reproducible).  Zero valid pixels -> 0/0 = NaN like the reference.
#include "common.h"
#include <math.h>

#define LOSS_NBLK 1024

struct Ce2dArgs {
  const float* pred;
  const long long* target;
  const float* weight;
  double* part;     // [LOSS_NBLK][2]
  float* out;       // [0] loss, [1] denominator (sum of weights of valid pixels)
  const float* gup; // upstream gradient scalar (device) or NULL (= 1)
  float* dpred;
  long P, HW;       // P = n*h*w
  int C, ignore_label, is_softmax;
};

// A target in [C, inf) that is not the ignore label is an ERROR in the reference (torch's nll_loss raises "Target t is out of
// bounds").  It must never index pred / weight: the pixel is skipped by the gradient and POISONS the loss with NaN (ce_oob),
// so the mistake is as loud as the reference's exception without a host sync in the hot path.
__device__ __forceinline__ bool ce_valid(long long t, int ignore, int C) { return t >= 0 && t != ignore && t < C; }
__device__ __forceinline__ bool ce_oob(long long t, int ignore, int C) { return t >= C && t != ignore; }

__global__ __launch_bounds__(256) void ce2d_fwd_kernel(Ce2dArgs a) {
  __shared__ double red[2][256];
  double num = 0.0, den = 0.0;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < a.P; p += (long)gridDim.x * 256) {
    long long t = a.target[p];
    if (ce_oob(t, a.ignore_label, a.C)) num = NAN;
    if (!ce_valid(t, a.ignore_label, a.C)) continue;
    long b = p / a.HW, r = p - b * a.HW;
    const float* col = a.pred + b * a.C * a.HW + r;
    float w = a.weight ? a.weight[t] : 1.f;
    float l;
    if (a.is_softmax) {
      float mx = col[0];
      for (int j = 1; j < a.C; ++j) mx = fmaxf(mx, col[(long)j * a.HW]);
      float s = 0.f;
      for (int j = 0; j < a.C; ++j) s += expf(col[(long)j * a.HW] - mx);
      l = (mx + logf(s)) - col[t * a.HW];
    } else {
      l = -logf(col[t * a.HW]);
    }
    num += (double)(w * l);
    den += (double)w;
  }
  red[0][threadIdx.x] = num;
  red[1][threadIdx.x] = den;
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += red[threadIdx.x][i];
    a.part[blockIdx.x * 2 + threadIdx.x] = s;
  }
}

__global__ void loss_finalize_kernel(const double* part, int nblk, float* out) {
  if (threadIdx.x == 0) {
    double num = 0.0, den = 0.0;
    for (int b = 0; b < nblk; ++b) { num += part[b * 2]; den += part[b * 2 + 1]; }
    out[0] = (float)(num / den);
    out[1] = (float)den;
  }
}

__global__ __launch_bounds__(256) void ce2d_bwd_kernel(Ce2dArgs a) {
  const float g = (a.gup ? a.gup[0] : 1.f) / a.out[1];
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < a.P; p += (long)gridDim.x * 256) {
    long long t = a.target[p];
    long b = p / a.HW, r = p - b * a.HW;
    const float* col = a.pred + b * a.C * a.HW + r;
    float* dcol = a.dpred + b * a.C * a.HW + r;
    if (!ce_valid(t, a.ignore_label, a.C)) {
      for (int j = 0; j < a.C; ++j) dcol[(long)j * a.HW] = 0.f;
      continue;
    }
    float w = (a.weight ? a.weight[t] : 1.f) * g;
    if (a.is_softmax) {
      float mx = col[0];
      for (int j = 1; j < a.C; ++j) mx = fmaxf(mx, col[(long)j * a.HW]);
      float s = 0.f;
      for (int j = 0; j < a.C; ++j) s += expf(col[(long)j * a.HW] - mx);
      float inv = 1.f / s;
      for (int j = 0; j < a.C; ++j)
        dcol[(long)j * a.HW] = w * (expf(col[(long)j * a.HW] - mx) * inv - (j == t ? 1.f : 0.f));
    } else {
      for (int j = 0; j < a.C; ++j) dcol[(long)j * a.HW] = (j == t) ? -w / col[(long)j * a.HW] : 0.f;
    }
  }
}

static int loss_grid(long P) {
  long g = (P + 255) / 256;
  if (g > LOSS_NBLK) g = LOSS_NBLK;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int simt_loss_ws_bytes(void) { return LOSS_NBLK * 2 * (int)sizeof(double); }

extern "C" int simt_ce2d_fwd(const float* pred, const int64_t* target, const float* weight, int n, int c, int h, int w,
                             int ignore_label, int is_softmax, void* ws, float* out, simt_stream_t stream) {
  SIMT_CHECK(pred && target && ws && out && n > 0 && c > 0);
  Ce2dArgs a;
  a.pred = pred; a.target = (const long long*)target; a.weight = weight; a.part = (double*)ws; a.out = out; a.gup = nullptr;
  a.dpred = nullptr; a.P = (long)n * h * w; a.HW = (long)h * w; a.C = c; a.ignore_label = ignore_label; a.is_softmax = is_softmax;
  int grid = loss_grid(a.P);
  hipLaunchKernelGGL(ce2d_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws, grid, out);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

extern "C" int simt_ce2d_bwd(const float* pred, const int64_t* target, const float* weight, int n, int c, int h, int w,
                             int ignore_label, int is_softmax, const float* out, const float* grad_out, float* dpred,
                             simt_stream_t stream) {
  SIMT_CHECK(pred && target && out && dpred);
  Ce2dArgs a;
  a.pred = pred; a.target = (const long long*)target; a.weight = weight; a.part = nullptr; a.out = (float*)out; a.gup = grad_out;
  a.dpred = dpred; a.P = (long)n * h * w; a.HW = (long)h * w; a.C = c; a.ignore_label = ignore_label; a.is_softmax = is_softmax;
  hipLaunchKernelGGL(ce2d_bwd_kernel, dim3(loss_grid(a.P)), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// ---- EntropyLoss: mean_p ( - sum_j softmax(x)_j * log_softmax(x)_j ),  x [n, c, h, w]
__global__ __launch_bounds__(256) void entropy_kernel(const float* x, long P, long HW, int C, double* part, const float* gup,
                                                      float* dx) {
  __shared__ double red[256];
  double acc = 0.0;
  const float g = dx ? (gup ? gup[0] : 1.f) / (float)P : 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < P; p += (long)gridDim.x * 256) {
    long b = p / HW, r = p - b * HW;
    const float* col = x + b * C * HW + r;
    float mx = col[0];
    for (int j = 1; j < C; ++j) mx = fmaxf(mx, col[(long)j * HW]);
    float s = 0.f;
    for (int j = 0; j < C; ++j) s += expf(col[(long)j * HW] - mx);
    float lse = mx + logf(s), ent = 0.f;
    for (int j = 0; j < C; ++j) {
      float lp = col[(long)j * HW] - lse;
      ent -= expf(lp) * lp;
    }
    acc += (double)ent;
    if (dx) {
      float* dcol = dx + b * C * HW + r;
      for (int j = 0; j < C; ++j) {
        float lp = col[(long)j * HW] - lse;
        dcol[(long)j * HW] = -g * expf(lp) * (lp + ent);
      }
    }
  }
  if (part) {
    red[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      double s = 0.0;
      for (int i = 0; i < 256; ++i) s += red[i];
      part[blockIdx.x * 2] = s;
      part[blockIdx.x * 2 + 1] = (blockIdx.x == 0) ? (double)P : 0.0;
    }
  }
}

// out[0] = mean entropy (forward, dx == NULL) ; with dx != NULL writes d(mean entropy)/dx * grad_out instead.
extern "C" int simt_entropy2d(const float* x, int n, int c, int h, int w, void* ws, float* out, const float* grad_out,
                              float* dx, simt_stream_t stream) {
  SIMT_CHECK(x && n > 0 && c > 0 && (dx || (ws && out)));
  long P = (long)n * h * w;
  int grid = loss_grid(P);
  hipLaunchKernelGGL(entropy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, P, (long)h * w, c,
                     dx ? nullptr : (double*)ws, grad_out, dx);
  SIMT_LAUNCH_CHECK();
  if (!dx) {
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws, grid, out);
    SIMT_LAUNCH_CHECK();
  }
  return SIMT_OK;
}
