// Declarations shared by the bf16 throughput conv kernels (conv_igemm2.hip, conv1x1_stream.hip).
#pragma once
#include "common.h"

struct Conv2KArgs {
  const char* x;
  const char* w;
  const char* wf;                      // weights in MFMA-fragment order (simt_conv_desc.w_frag) or NULL
  int nt16;                            // Npad / 16: 16-row blocks per 64-deep K stage of wf
  bf16_t* y;
  const float* bias;
  const bf16_t* res;
  const bf16_t* mask;
  const unsigned char* res_bits;
  const bf16_t* bnr_y;                 // fused first pass of the BatchNorm backward (simt_conv_desc.bnr_*)
  const float *bnr_mean, *bnr_rstd, *bnr_scale, *bnr_shift;
  const unsigned char* bnr_bits;
  float* bnr_part;
  int bnr_mode, bnr_ld;
  float* stats;
  const char* zero;
  int H, W, Ho, Wo, Cout, Nstore, ldy, ldr, stride, ntaps, relu, M;
  int kc_per_tap, pix_bytes, wrow_bytes, ldm;
  int ntiles_n, ntiles_m;
  int out_f32;     // 1: y is fp32 and is stored straight from the accumulators (no bias/residual/ReLU/stats)
  int rows;        // pixels per tile (<= BM)
  int nblk128;     // stats slots allocated by the caller: ceil(M/128) >= ntiles_m
  float rcp_hw, rcp_wo;   // 1 / (Ho*Wo), 1 / Wo for fast_divmod (exact below 2^24 pixels)
  int toff[SIMT_MAX_TAPS];   // (dy*W + dx) * pix_bytes: 32-bit so that the uniform per-stage lookup is an s_load_dword
                             // (a 16-bit table compiles to global_load_sshort, whose vmcnt(0) drains the glds ring)
  short dy[SIMT_MAX_TAPS], dx[SIMT_MAX_TAPS];
  // fused train-mode BatchNorm (simt_fbn_desc; kernels instantiated with FBN = 1 only)
  int fbn_mode, fbn_ldo;
  bf16_t* fbn_out;
  unsigned long long* fbn_bar;         // [SIMT_FBN_BAR_WORDS] ticket counters
  unsigned long long* fbn_err;         // sticky error word (simt_fbn_desc.err or work[SIMT_FBN_ERR_WORD]): set by a poller that timed out
  unsigned long long *fbn_cgran, *fbn_slots;   // [2][Cout] constants granules; [ntiles_m][2 | 3][Cout] tile-sum granules
  const float *fbn_gamma, *fbn_beta;
  float *fbn_rmean, *fbn_rvar;
  float fbn_momentum, fbn_eps;
  float *fbn_mean, *fbn_rstd, *fbn_scale, *fbn_shift, *fbn_coef, *fbn_dgamma, *fbn_dbeta;
  // BatchNorm + ReLU of the input in the operand path (simt_conv_desc.in_*; conv1x1_rows_kernel FL_STATS_INBN only)
  const float *in_scale, *in_shift;
  bf16_t* in_out;
};

#ifdef SIMT_ABLATION
// In-kernel stamps (diagnostic builds only): s_memtime of wave 0 / lane 0 of every workgroup at fixed points, read back with
// simt_debug_stamps().  Slots: 0 start, 1 addressing done, 2 first stage landed, 3 main loop done, 4 tile in LDS, 5 stores issued, 6 end.
static __device__ unsigned long long g_stamps[8192 * 8];     // one copy per translation unit (no relocatable device code)
// s_memrealtime ticks (constant 100 MHz) between stamps 0 and 6 -- with slots 0 and 6 the shader clock the launch actually ran at -- live in their
// OWN array (slot 7 of g_stamps is a regular stamp of the fused-BatchNorm tail and of the rows kernel: the delta used to be overwritten there)
static __device__ unsigned long long g_stamps_rt[8192];
#define STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) { g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    if ((i) == 0) g_stamps_rt[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); \
    if ((i) == 6) g_stamps_rt[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - g_stamps_rt[blockIdx.x]; } } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N <= 16, "unsupported vmcnt");
#define SIMT_VMCNT_CASE(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory")
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SIMT_VMCNT_CASE(1); SIMT_VMCNT_CASE(2); SIMT_VMCNT_CASE(3); SIMT_VMCNT_CASE(4); SIMT_VMCNT_CASE(5); SIMT_VMCNT_CASE(6);
  SIMT_VMCNT_CASE(7); SIMT_VMCNT_CASE(8); SIMT_VMCNT_CASE(9); SIMT_VMCNT_CASE(10); SIMT_VMCNT_CASE(11); SIMT_VMCNT_CASE(12);
  SIMT_VMCNT_CASE(13); SIMT_VMCNT_CASE(14); SIMT_VMCNT_CASE(15); SIMT_VMCNT_CASE(16);
#undef SIMT_VMCNT_CASE
}

