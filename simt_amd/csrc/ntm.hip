// Noise-transition-matrix micro-solver: everything the reference does on the 22x19 / 22x22 matrices per iteration,
// as three tiny single-workgroup kernels (all operands live in LDS) instead of ~300 eager launches + host syncs.
//
// Replaces:
//   sig_NTM.forward / sig_W.forward and their autograd            (model/deeplab_multi.py:244-286)
//   the 10-step inner W loop incl. Adam and the grad leak into NTM (tools/trainV2_simt.py:326-339)
//   Convex / Volume (19x19 det, NaN/Inf guard) / Anchor terms      (tools/trainV2_simt.py:375-384,412-424)
//   Adam on NTM1/NTM2                                              (tools/trainV2_simt.py:435-436)
#include "common.h"
#include <math.h>

#define NQ 40   // max Q
#define NC 20   // max C

struct NtmInnerArgs {
  float* ntm[2];
  float* w[2];
  float* ntm_grad[2];
  float* w_m[2];
  float* w_v[2];
  float* T_out[2];
  const float* class_dist;
  int Q, C, steps, step0;
  float lr, beta1, beta2, eps;
  int k0;      // first NTM index this launch handles (1 for one-output models)
  const unsigned long long* skip_if;   // simt_ntm_inner_desc.skip_if
};

// T = rowL1normalize( sigmoid(N) * cd + [I;0] ), also returns sigmoid and the row sums (for the backward)
__device__ void sigT_forward(const float* N, const float* cd, int Q, int C, float* T, float* sig, float* rs, int tid) {
  for (int i = tid; i < Q * C; i += 256) {
    int j = i / C, c = i - j * C;
    float s = 1.0f / (1.0f + expf(-N[i]));
    sig[i] = s;
    T[i] = s * cd[c] + (j == c ? 1.f : 0.f);
  }
  __syncthreads();
  for (int j = tid; j < Q; j += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += fabsf(T[j * C + c]);
    rs[j] = fmaxf(s, 1e-12f);
  }
  __syncthreads();
  for (int i = tid; i < Q * C; i += 256) T[i] = T[i] / rs[i / C];
  __syncthreads();
}

// dN += sig'(N) * cd * (dT - sum_c dT*T)/rowsum      (U > 0 so d|U| = dU)
__device__ void sigT_backward_acc(const float* dT, const float* T, const float* sig, const float* rs, const float* cd,
                                  int Q, int C, float* dot, float* gradN, int tid) {
  for (int j = tid; j < Q; j += 256) {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += dT[j * C + c] * T[j * C + c];
    dot[j] = s;
  }
  __syncthreads();
  for (int i = tid; i < Q * C; i += 256) {
    int j = i / C, c = i - j * C;
    float dU = (dT[i] - dot[j]) / rs[j];
    float s = sig[i];
    gradN[i] += dU * cd[c] * s * (1.f - s);
  }
  __syncthreads();
}

// W = softmax(weight with diag := -1e4, dim=1) - I      (also writes the diag back, like the reference's in-place set)
__device__ void sigW_forward(float* wraw, int Q, float* sm, float* Wm, int tid) {
  for (int j = tid; j < Q; j += 256) {
    wraw[j * Q + j] = -10000.f;
    float mx = -INFINITY;
    for (int k = 0; k < Q; ++k) mx = fmaxf(mx, wraw[j * Q + k]);
    float s = 0.f;
    for (int k = 0; k < Q; ++k) s += expf(wraw[j * Q + k] - mx);
    for (int k = 0; k < Q; ++k) {
      float v = expf(wraw[j * Q + k] - mx) / s;
      sm[j * Q + k] = v;
      Wm[j * Q + k] = v - (j == k ? 1.f : 0.f);
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void ntm_inner_kernel(NtmInnerArgs a) {
  __shared__ float N[NQ * NC], T[NQ * NC], sig[NQ * NC], WT[NQ * NC], dT[NQ * NC], gN[NQ * NC];
  __shared__ float wraw[NQ * NQ], sm[NQ * NQ], Wm[NQ * NQ], dW[NQ * NQ], am[NQ * NQ], av[NQ * NQ];
  __shared__ float rs[NQ], dot[NQ], cd[NC];
  const int k = blockIdx.x + a.k0, tid = threadIdx.x, Q = a.Q, C = a.C;
  // a fused BatchNorm launch of an earlier iteration gave up (sticky word): W, its Adam moments, the leaked NTM gradient and T_out stay as they are
  if (a.skip_if && __hip_atomic_load(a.skip_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;
  for (int i = tid; i < Q * C; i += 256) { N[i] = a.ntm[k][i]; gN[i] = 0.f; }
  for (int i = tid; i < Q * Q; i += 256) { wraw[i] = a.w[k][i]; am[i] = a.w_m[k][i]; av[i] = a.w_v[k][i]; }
  if (tid < C) cd[tid] = a.class_dist[tid];
  __syncthreads();
  sigT_forward(N, cd, Q, C, T, sig, rs, tid);
  for (int it = 0; it < a.steps; ++it) {
    sigW_forward(wraw, Q, sm, Wm, tid);
    // WT = Wm @ T
    for (int i = tid; i < Q * C; i += 256) {
      int j = i / C, c = i - j * C;
      float s = 0.f;
      for (int q = 0; q < Q; ++q) s += Wm[j * Q + q] * T[q * C + c];
      WT[i] = s;
    }
    __syncthreads();
    // dWm = 2 WT T^T ; dT = 2 Wm^T WT
    for (int i = tid; i < Q * Q; i += 256) {
      int j = i / Q, q = i - j * Q;
      float s = 0.f;
      for (int c = 0; c < C; ++c) s += WT[j * C + c] * T[q * C + c];
      dW[i] = 2.f * s;
    }
    for (int i = tid; i < Q * C; i += 256) {
      int q = i / C, c = i - q * C;
      float s = 0.f;
      for (int j = 0; j < Q; ++j) s += Wm[j * Q + q] * WT[j * C + c];
      dT[i] = 2.f * s;
    }
    __syncthreads();
    sigT_backward_acc(dT, T, sig, rs, cd, Q, C, dot, gN, tid);  // the leak into NTM.grad (quirk 3)
    // softmax backward -> gradient w.r.t. the raw weight, then Adam (torch.optim.Adam single-tensor semantics)
    for (int j = tid; j < Q; j += 256) {
      float s = 0.f;
      for (int q = 0; q < Q; ++q) s += dW[j * Q + q] * sm[j * Q + q];
      dot[j] = s;
    }
    __syncthreads();
    const int step = a.step0 + it + 1;
    const double bc1 = 1.0 - pow((double)a.beta1, (double)step);
    const double bc2 = 1.0 - pow((double)a.beta2, (double)step);
    const float step_size = (float)((double)a.lr / bc1);
    const float bc2s = (float)sqrt(bc2);
    for (int i = tid; i < Q * Q; i += 256) {
      int j = i / Q;
      float g = sm[i] * (dW[i] - dot[j]);
      float m = am[i] + (g - am[i]) * (1.f - a.beta1);
      float v = av[i] * a.beta2 + (1.f - a.beta2) * g * g;
      am[i] = m;
      av[i] = v;
      float denom = sqrtf(v) / bc2s + a.eps;
      wraw[i] = wraw[i] - step_size * (m / denom);
    }
    __syncthreads();
  }
  for (int i = tid; i < Q * C; i += 256) {
    a.ntm_grad[k][i] += gN[i];
    a.T_out[k][i] = T[i];
  }
  for (int i = tid; i < Q * Q; i += 256) { a.w[k][i] = wraw[i]; a.w_m[k][i] = am[i]; a.w_v[k][i] = av[i]; }
}

extern "C" int simt_ntm_inner_loop(const simt_ntm_inner_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->Q <= NQ && d->C <= NC && d->C <= d->Q && d->class_dist);
  NtmInnerArgs a;
  a.k0 = d->single ? 1 : 0;
  for (int k = 0; k < 2; ++k) {
    a.ntm[k] = nullptr; a.w[k] = nullptr; a.ntm_grad[k] = nullptr; a.w_m[k] = nullptr; a.w_v[k] = nullptr; a.T_out[k] = nullptr;
    if (k < a.k0) continue;
    SIMT_CHECK(d->ntm[k] && d->w[k] && d->ntm_grad[k] && d->w_m[k] && d->w_v[k] && d->T_out[k]);
    a.ntm[k] = d->ntm[k]; a.w[k] = d->w[k]; a.ntm_grad[k] = d->ntm_grad[k]; a.w_m[k] = d->w_m[k]; a.w_v[k] = d->w_v[k];
    a.T_out[k] = d->T_out[k];
  }
  a.class_dist = d->class_dist; a.Q = d->Q; a.C = d->C; a.steps = d->steps; a.step0 = d->step0;
  a.lr = d->lr; a.beta1 = d->beta1; a.beta2 = d->beta2; a.eps = d->eps;
  a.skip_if = (const unsigned long long*)d->skip_if;
  hipLaunchKernelGGL(ntm_inner_kernel, dim3(2 - a.k0), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// --------------------------------------------------------------------------------------------------------
// post-head: Convex / Volume / Anchor, total loss, and d(loss)/dNTM accumulated into ntm_grad.
// --------------------------------------------------------------------------------------------------------
struct NtmPostArgs {
  const float* ntm[2];
  float* w[2];          // raw sig_W weights (diag rewritten, like a forward call)
  float* ntm_grad[2];
  const float* class_dist;
  const float* hout;    // simt_head_loss output
  float* lout;          // [16] scalars
  int Q, C, QMAXH;
  float lambda_seg, lambda_convex, lambda_volume, lambda_anchor, gscale;
  int k0;      // 1: one-output model, only NTM [1]
};

__global__ __launch_bounds__(256) void ntm_post_kernel(NtmPostArgs a) {
  __shared__ float N[NQ * NC], T[2][NQ * NC], sig[2][NQ * NC], WT[NQ * NC], dTc[2][NQ * NC], dTv[2][NQ * NC];
  __shared__ float wraw[NQ * NQ], sm[NQ * NQ], Wm[NQ * NQ];
  __shared__ float rs[2][NQ], dot[NQ], cd[NC];
  float* dTot = N;   // N and WT are dead once T/sig and the per-k terms exist
  float* gN = WT;
  __shared__ float G[NC * NC], Gi[NC * NC];
  __shared__ float red[256];
  __shared__ float s_convex[2], s_vol[2], s_anchor[2];
  __shared__ float fcol[NC];
  __shared__ float s_det;
  __shared__ int s_piv;
  const int tid = threadIdx.x, Q = a.Q, C = a.C, QC = Q * C;
  if (tid < C) cd[tid] = a.class_dist[tid];
  if (tid == 0) { s_convex[0] = 0.f; s_vol[0] = 0.f; s_anchor[0] = 0.f; }
  __syncthreads();
  for (int k = a.k0; k < 2; ++k) {
    for (int i = tid; i < QC; i += 256) N[i] = a.ntm[k][i];
    for (int i = tid; i < Q * Q; i += 256) wraw[i] = a.w[k][i];
    __syncthreads();
    sigT_forward(N, cd, Q, C, T[k], sig[k], rs[k], tid);
    sigW_forward(wraw, Q, sm, Wm, tid);
    for (int i = tid; i < Q * Q; i += 256) a.w[k][i] = wraw[i];
    for (int i = tid; i < QC; i += 256) {
      int j = i / C, c = i - j * C;
      float s = 0.f;
      for (int q = 0; q < Q; ++q) s += Wm[j * Q + q] * T[k][q * C + c];
      WT[i] = s;
    }
    __syncthreads();
    // convex = -||W T||^2 ; d/dT = -2 W^T W T
    float part = 0.f;
    for (int i = tid; i < QC; i += 256) part += WT[i] * WT[i];
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int i = 0; i < 256; ++i) s += red[i];
      s_convex[k] = -s;
    }
    for (int i = tid; i < QC; i += 256) {
      int q = i / C, c = i - q * C;
      float s = 0.f;
      for (int j = 0; j < Q; ++j) s += Wm[j * Q + q] * WT[j * C + c];
      dTc[k][i] = -2.f * s;
    }
    // G = T^T T
    for (int i = tid; i < C * C; i += 256) {
      int r = i / C, c = i - r * C;
      float s = 0.f;
      for (int j = 0; j < Q; ++j) s += T[k][j * C + r] * T[k][j * C + c];
      G[i] = s;
      Gi[i] = (r == c) ? 1.f : 0.f;
    }
    __syncthreads();
    // Gauss-Jordan with partial pivoting on [G | Gi]: det = prod(pivots)*sign, Gi = G^-1.  Same arithmetic per element
    // as the textbook serial loop; the row operations of one pivot step run across the block (a single thread walking
    // 19 x 19 x 38 LDS-latency-bound FMAs took 0.45 ms per matrix).
    if (tid == 0) s_det = 1.f;
    __syncthreads();
    for (int p = 0; p < C; ++p) {
      if (tid == 0) {
        int piv = p;
        float best = fabsf(G[p * C + p]);
        for (int r = p + 1; r < C; ++r) {
          float v = fabsf(G[r * C + p]);
          if (v > best) { best = v; piv = r; }
        }
        s_piv = piv;
        if (piv != p) s_det = -s_det;
      }
      __syncthreads();
      const int piv = s_piv;
      if (piv != p && tid < 2 * C) {
        float* Mx = tid < C ? G : Gi;
        const int c = tid < C ? tid : tid - C;
        float t = Mx[p * C + c]; Mx[p * C + c] = Mx[piv * C + c]; Mx[piv * C + c] = t;
      }
      __syncthreads();
      const float d = G[p * C + p];
      __syncthreads();
      if (tid == 0) s_det *= d;
      if (tid < 2 * C) {
        float* Mx = tid < C ? G : Gi;
        const int c = tid < C ? tid : tid - C;
        Mx[p * C + c] *= 1.f / d;
      }
      __syncthreads();
      if (tid < C) fcol[tid] = G[tid * C + p];
      __syncthreads();
      for (int idx = tid; idx < C * 2 * C; idx += 256) {
        const int r = idx / (2 * C), cc = idx - r * 2 * C;
        const float f = fcol[r];
        if (r != p && f != 0.f) {
          float* Mx = cc < C ? G : Gi;
          const int c = cc < C ? cc : cc - C;
          Mx[r * C + c] -= f * Mx[p * C + c];
        }
      }
      __syncthreads();
    }
    if (tid == 0) s_vol[k] = logf(sqrtf(fabsf(s_det)));
    __syncthreads();
    for (int i = tid; i < QC; i += 256) {
      int j = i / C, c = i - j * C;
      float s = 0.f;
      for (int r = 0; r < C; ++r) s += T[k][j * C + r] * Gi[r * C + c];
      dTv[k][i] = s;
    }
    // anchor: sum over existing rows of ||T[j]-A[j]||^2
    const float* A = a.hout + 16 + k * QC;
    const float* ex = a.hout + 16 + 2 * QC + k * a.QMAXH;
    part = 0.f;
    for (int i = tid; i < QC; i += 256) {
      int j = i / C;
      if (ex[j] != 0.f) { float df = T[k][i] - A[i]; part += df * df; }
    }
    __syncthreads();
    red[tid] = part;
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int i = 0; i < 256; ++i) s += red[i];
      s_anchor[k] = s;
    }
    __syncthreads();
  }
  // Volume guard (reference :420-421): NaN/Inf of the SUM -> python float 0 (no gradient)
  float vol = s_vol[0] + s_vol[1];
  bool vol_ok = !(isinf(vol) || isnan(vol));
  if (!vol_ok) vol = 0.f;
  const float convex = s_convex[0] + s_convex[1];
  const float anchor = s_anchor[0] + s_anchor[1];
  for (int k = a.k0; k < 2; ++k) {
    const float wy = (k == 0) ? a.lambda_seg : 1.f;
    const float* A = a.hout + 16 + k * QC;
    const float* ex = a.hout + 16 + 2 * QC + k * a.QMAXH;
    const float* dTy = a.hout + 16 + 2 * QC + 4 * a.QMAXH + k * QC;
    for (int i = tid; i < QC; i += 256) {
      int j = i / C;
      float g = wy * dTy[i] + a.lambda_convex * dTc[k][i];
      if (vol_ok) g += a.lambda_volume * dTv[k][i];
      if (ex[j] != 0.f) g += a.lambda_anchor * 2.f * (T[k][i] - A[i]);
      dTot[i] = a.gscale * g;
      gN[i] = 0.f;
    }
    __syncthreads();
    sigT_backward_acc(dTot, T[k], sig[k], rs[k], cd, Q, C, dot, gN, tid);
    for (int i = tid; i < QC; i += 256) a.ntm_grad[k][i] += gN[i];
    __syncthreads();
  }
  if (tid == 0) {
    const float* o = a.hout;
    const float lseg = a.k0 ? 0.f : a.lambda_seg;          // one-output model: no auxiliary-head terms
    float place = lseg * o[2] + o[3];
    float target = o[1] + o[5] + lseg * o[0] + lseg * o[4];
    float total = place + target + a.lambda_convex * convex + a.lambda_volume * vol + a.lambda_anchor * anchor;
    float* l = a.lout;
    l[0] = total * a.gscale; l[1] = a.k0 ? 0.f : o[0]; l[2] = o[1]; l[3] = a.k0 ? 0.f : o[4]; l[4] = o[5]; l[5] = place; l[6] = convex; l[7] = vol;
    l[8] = anchor; l[9] = vol_ok ? 1.f : 0.f; l[10] = s_vol[0]; l[11] = s_vol[1];
    l[12] += o[15];     // out-of-range labels since the host last cleared the slot (every micro-batch of every step counts)
  }
}

extern "C" int simt_ntm_post(const simt_ntm_post_desc* d, simt_stream_t stream) {
  SIMT_CHECK(d && d->Q <= NQ && d->C <= NC && d->hout && d->lout && d->class_dist);
  NtmPostArgs a;
  a.k0 = d->single ? 1 : 0;
  for (int k = 0; k < 2; ++k) {
    a.ntm[k] = nullptr; a.w[k] = nullptr; a.ntm_grad[k] = nullptr;
    if (k < a.k0) continue;
    SIMT_CHECK(d->ntm[k] && d->w[k] && d->ntm_grad[k]);
    a.ntm[k] = d->ntm[k]; a.w[k] = d->w[k]; a.ntm_grad[k] = d->ntm_grad[k];
  }
  a.class_dist = d->class_dist; a.hout = d->hout; a.lout = d->lout; a.Q = d->Q; a.C = d->C; a.QMAXH = 40;
  a.lambda_seg = d->lambda_seg; a.lambda_convex = d->lambda_convex; a.lambda_volume = d->lambda_volume;
  a.lambda_anchor = d->lambda_anchor; a.gscale = d->gscale;
  hipLaunchKernelGGL(ntm_post_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

// --------------------------------------------------------------------------------------------------------
// stand-alone sig_NTM / sig_W forward+backward (module API) and a small Adam step
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sig_ntm_kernel(const float* ntm, const float* class_dist, const float* dT_in,
                                                      float* T_out, float* dN_out, int Q, int C) {
  __shared__ float N[NQ * NC], T[NQ * NC], sig[NQ * NC], dT[NQ * NC], gN[NQ * NC];
  __shared__ float rs[NQ], dot[NQ], cd[NC];
  const int tid = threadIdx.x;
  for (int i = tid; i < Q * C; i += 256) { N[i] = ntm[i]; gN[i] = 0.f; if (dT_in) dT[i] = dT_in[i]; }
  if (tid < C) cd[tid] = class_dist[tid];
  __syncthreads();
  sigT_forward(N, cd, Q, C, T, sig, rs, tid);
  if (T_out) for (int i = tid; i < Q * C; i += 256) T_out[i] = T[i];
  if (dT_in) {
    sigT_backward_acc(dT, T, sig, rs, cd, Q, C, dot, gN, tid);
    for (int i = tid; i < Q * C; i += 256) dN_out[i] = gN[i];
  }
}
extern "C" int simt_sig_ntm(const float* ntm, const float* class_dist, const float* dT, float* T_out, float* dN_out,
                            int Q, int C, simt_stream_t stream) {
  SIMT_CHECK(ntm && class_dist && Q <= NQ && C <= NC && (!dT || dN_out));
  hipLaunchKernelGGL(sig_ntm_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ntm, class_dist, dT, T_out, dN_out, Q, C);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

__global__ __launch_bounds__(256) void sig_w_kernel(float* weight, const float* dW_in, float* W_out, float* dweight_out,
                                                    int Q) {
  __shared__ float wraw[NQ * NQ], sm[NQ * NQ], Wm[NQ * NQ];
  __shared__ float dot[NQ];
  const int tid = threadIdx.x;
  for (int i = tid; i < Q * Q; i += 256) wraw[i] = weight[i];
  __syncthreads();
  sigW_forward(wraw, Q, sm, Wm, tid);
  for (int j = tid; j < Q; j += 256) weight[j * Q + j] = -10000.f;  // in-place diag set (deeplab_multi.py:279-281)
  if (W_out) for (int i = tid; i < Q * Q; i += 256) W_out[i] = Wm[i];
  if (dW_in) {
    for (int j = tid; j < Q; j += 256) {
      float s = 0.f;
      for (int q = 0; q < Q; ++q) s += dW_in[j * Q + q] * sm[j * Q + q];
      dot[j] = s;
    }
    __syncthreads();
    for (int i = tid; i < Q * Q; i += 256) dweight_out[i] = sm[i] * (dW_in[i] - dot[i / Q]);
  }
}
extern "C" int simt_sig_w(float* weight, const float* dW, float* W_out, float* dweight_out, int Q, simt_stream_t stream) {
  SIMT_CHECK(weight && Q <= NQ && (!dW || dweight_out));
  hipLaunchKernelGGL(sig_w_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weight, dW, W_out, dweight_out, Q);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}

__global__ void adam_step_kernel(float* p, const float* g, float* m, float* v, long n, float step_size, float bc2s,
                                 float beta1, float beta2, float eps, const unsigned long long* skip_if) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (skip_if && __hip_atomic_load(skip_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;     // like simt_sgd_desc.skip_if
  float gi = g[i];
  float mi = m[i] + (gi - m[i]) * (1.f - beta1);
  float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2s + eps));
}
// skip_if: optional device word (simt_fbn_desc.err); while it is non-zero the launch changes neither p nor the moments
extern "C" int simt_adam_step_guarded(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                                      float eps, int step, const uint64_t* skip_if, simt_stream_t stream) {
  SIMT_CHECK(p && g && m && v && step >= 1);
  double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  float step_size = (float)((double)lr / bc1), bc2s = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     n, step_size, bc2s, beta1, beta2, eps, (const unsigned long long*)skip_if);
  SIMT_LAUNCH_CHECK();
  return SIMT_OK;
}
extern "C" int simt_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                              float eps, int step, simt_stream_t stream) {
  return simt_adam_step_guarded(p, g, m, v, n, lr, beta1, beta2, eps, step, nullptr, stream);
}
