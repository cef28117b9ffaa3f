// Fused multi-tensor SGD (momentum, weight decay) with the reference's duplicate-parameter semantics.
//
// Replaces torch.optim.SGD(model.optim_parameters(args)).step() (tools/trainV2_simt.py:296-297,434).  The reference's
// parameter generator lists every layer3/layer4 tensor 3-4 times (model/deeplab_multi.py:211-217) and
// the optimiser it ran (single-tensor loop, foreach=False) applies the update once PER LISTING, sequentially, sharing one
// momentum buffer.  Here the `mult` listings of an element are replayed in registers, so the whole 43 M-parameter
// update is ONE launch instead of ~3000.
#include "common.h"

struct SgdSeg {
  float* p;
  const float* g;
  float* buf;
  long long n;
  int mult;
  int group;
};

struct SgdArgs {
  const SgdSeg* segs;
  const int* chunks;  // [nchunks][2] = (segment, first element / chunk)
  int nchunks, chunk;
  float lr[4], wd[4];
  float momentum, dampening;
  int first_step;
  const unsigned long long* skip_if;   // simt_sgd_desc.skip_if: the launch changes nothing while this device word is non-zero
};

__global__ __launch_bounds__(256) void sgd_multi_kernel(SgdArgs a) {
  const int ch = blockIdx.x;
  // a fused-BatchNorm launch of this step bailed out (its polling timed out: conv2_epilogue.h): its gradients are not to be applied
  if (a.skip_if && __hip_atomic_load(a.skip_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;
  const SgdSeg s = a.segs[a.chunks[2 * ch]];
  const long long start = (long long)a.chunks[2 * ch + 1] * a.chunk;
  long long end = start + a.chunk;
  if (end > s.n) end = s.n;
  const float lr = a.lr[s.group], wd = a.wd[s.group];
  auto upd = [&](float& p, float g, float& buf) {
    for (int r = 0; r < s.mult; ++r) {
      float d = wd != 0.f ? g + wd * p : g;
      if (a.momentum != 0.f) {
        buf = a.first_step ? d : a.momentum * buf + (1.f - a.dampening) * d;
        d = buf;
      }
      p = p - lr * d;
    }
  };
  long long i0 = start;
  // 16-byte path (the flat gradient buffer packs tensors back to back, so a segment's gradient may start unaligned)
  if ((((uintptr_t)(s.p + start) | (uintptr_t)(s.g + start) | (uintptr_t)(s.buf + start)) & 15) == 0) {
    const long long n4 = (end - start) >> 2;
    for (long long q = threadIdx.x; q < n4; q += 256) {
      const long long i = start + (q << 2);
      float4 p = *(const float4*)(s.p + i);
      const float4 g = *(const float4*)(s.g + i);
      float4 b = a.first_step ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(s.buf + i);
      upd(p.x, g.x, b.x); upd(p.y, g.y, b.y); upd(p.z, g.z, b.z); upd(p.w, g.w, b.w);
      *(float4*)(s.p + i) = p;
      if (a.momentum != 0.f) *(float4*)(s.buf + i) = b;
    }
    i0 = start + (n4 << 2);
  }
  for (long long i = i0 + threadIdx.x; i < end; i += 256) {
    float p = s.p[i], g = s.g[i];
    float buf = a.first_step ? 0.f : s.buf[i];
    upd(p, g, buf);
    s.p[i] = p;
    if (a.momentum != 0.f) s.buf[i] = buf;
  }
}

extern "C" int simt_sgd_multi(const simt_sgd_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->segs && d->chunks && d->nchunks > 0 && d->chunk > 0);
  SgdArgs a;
  a.segs = (const SgdSeg*)d->segs; a.chunks = (const int*)d->chunks; a.nchunks = d->nchunks; a.chunk = d->chunk;
  for (int i = 0; i < 4; ++i) { a.lr[i] = d->lr[i]; a.wd[i] = d->wd[i]; }
  a.momentum = d->momentum; a.dampening = d->dampening; a.first_step = d->first_step;
  a.skip_if = (const unsigned long long*)d->skip_if;
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(d->nchunks), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// dst[i] (+)= src[i]   (tiny fp32 vectors: the bias of the fused ASPP GEMM is the sum of its branches' biases,
// model/deeplab_multi.py:115-119)
__global__ void vec_acc_kernel(float* dst, const float* src, int n, int accumulate) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = accumulate ? dst[i] + src[i] : src[i];
}
extern "C" int simt_vec_acc(float* dst, const float* src, int n, int accumulate, simt_stream_t stream) {
  SIMT_CHECK(dst && src && n > 0);
  hipLaunchKernelGGL(vec_acc_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, dst, src, n, accumulate);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
