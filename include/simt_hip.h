/* simt_hip.h -- C ABI of libsimt_hip.so, the gfx950 (MI355X) implementation of SimT's per-iteration hot path.
 *
 * The reference (CityU-AIM-Group/SimT) has no FFI layer: its hot path is a chain of torch ops called from
 * Python (tools/trainV2_simt.py:326-436 through model/deeplab_multi.py and utils/loss.py).  Each entry point
 * below replaces the torch op(s) cited next to it.  Conventions:
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller (borrowed for the call),
 *     except descriptors, which are host structs read during the call;
 *   - `stream` is a hipStream_t; kernels are enqueued there and the call never synchronises;
 *   - activations are NHWC ([B][H][W][C], C contiguous), dtype SIMT_BF16 or SIMT_F32 (fp32 = parity mode);
 *   - master weights / gradients are fp32 in PyTorch OIHW layout (the reference's state_dict contract);
 *   - return value: SIMT_OK or an error code; simt_last_error() gives the text (mirrors the reference's
 *     assert-style failures, utils/loss.py:22-27).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 */
#ifndef SIMT_HIP_H
#define SIMT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* simt_stream_t; /* hipStream_t */

enum { SIMT_OK = 0, SIMT_ERR_INVALID = 1, SIMT_ERR_LAUNCH = 2 };
enum { SIMT_F32 = 0, SIMT_BF16 = 1 };
#define SIMT_MAX_TAPS 36

const char* simt_last_error(void);
/* Device-scope events for ordering two HIP streams of ONE device (the launch lists of simt_amd/engine.py): like hipEventCreateWithFlags /
 * hipEventRecord / hipStreamWaitEvent (which torch.cuda.Event wraps), but recorded with a device-scope release instead of the default
 * system-scope fence -- the host never inspects them.  Replaces torch.cuda.Event().record() / stream.wait_event().  `scope`:
 *   0  hipEventDisableTiming | hipEventReleaseToDevice      the DOCUMENTED device-scope release (default of the launch lists since ABI 2)
 *   1  hipEventDisableTiming                                system-scope release (what torch.cuda.Event does)
 *   2  hipEventDisableTiming | hipEventDisableSystemFence   no release at the marker at all: ordering rests on the producing kernels' own
 *      end-of-kernel agent-scope release (round 5's form; explicit opt-in, SIMT_EVENT_SCOPE=2) */
int simt_event_create(void** ev, int scope);
int simt_event_destroy(void* ev);
int simt_event_record(void* ev, simt_stream_t stream);
int simt_stream_wait_event(simt_stream_t stream, void* ev);
/* 2 since round 6: simt_sgd_desc.skip_if, simt_fbn_desc.err, simt_ntm_inner_desc.skip_if (trailing pointers the library dereferences),
 * simt_adam_step_guarded, simt_event_create's scope values, simt_conv_desc.cu_budget / in_scale / in_shift / in_out.  A caller built against an older header passes
 * shorter structs: check the version before the first call (simt_amd/_lib.py does). */
#define SIMT_ABI_VERSION 2
int simt_abi_version(void);

/* ---- convolution: fprop / dgrad (implicit GEMM, MFMA) --------------------------------------------------
 * y[m][n] = act( sum_{tap,ci} x[pixel(m)*stride + (dy,dx)[tap]][ci] * w[n][tap*Cin+ci] + bias[n] + res[m][n] )
 * replaces nn.Conv2d forward (model/deeplab_multi.py:62,68,73,110,127,156; Classifier_Module.forward :115-119 is
 * ONE call with the taps of both live dilations) and, with the dgrad-packed weight and negated taps, its dgrad.
 * stats (optional): [ceil(M/128)][2][Cout] per-tile sum / sum-of-squares of the fp32 results (BatchNorm batch stats). */
typedef struct {
  const void* x;       /* [B][H][W][Cin] dtype_in */
  const void* w;       /* [Npad][ntaps*Cin] dtype_in, K contiguous (simt_pack_weight) */
  void* y;             /* [B*Ho*Wo][ldy] dtype_out */
  const float* bias;   /* [Cout] or NULL */
  const void* res;     /* [B*Ho*Wo][ldr] dtype_in, added before the activation, or NULL */
  float* stats;        /* or NULL */
  int32_t B, H, W, Cin, Ho, Wo, Cout;
  int32_t Npad;        /* rows of w, multiple of tile_n */
  int32_t Nstore;      /* columns written, multiple of 8, <= ldy */
  int32_t ldy, ldr, stride, ntaps, relu;
  int32_t dtype_in, dtype_out, tile_n; /* tile_n in {128,64,32}; bf16 -> bf16 also 256 */
  int16_t dy[SIMT_MAX_TAPS], dx[SIMT_MAX_TAPS];
  const void* mask;    /* [B*Ho*Wo][ldm] dtype_in or NULL: the result is zeroed where mask <= 0 (ReLU backward of the layer
                        * whose output this gradient belongs to; used by the BN-free VGG trunk, model/deeplab_vgg.py) */
  int32_t ldm;
  const unsigned char* res_bits; /* or NULL: res[m][n] only counts where bit (n & 7) of res_bits[(m*ldr + n) / 8] is set -- the
                                  * identity-shortcut gradient dz * (z > 0) of a residual block (model/deeplab_multi.py:97-100)
                                  * taken straight from dz and the bit mask of simt_bn_apply_bits, never materialised */
  /* Fused first pass of the BatchNorm backward (bf16 v2 kernel only; bnr_mode 0 = off).  When this launch PRODUCES the
   * gradient dz of a BatchNorm-ed activation (it is the dgrad of the conv that consumed relu(bn(y)), or the dx of the next
   * block), the epilogue also accumulates, on the values it stores, S1 = sum g and S2 = sum g*xhat with
   * g = dz * mask, xhat = (y - mean) * rstd -- what bn_bwd_reduce_kernel would re-read dz for -- into
   * bnr_part[m-tile][3][Cout] (third row zero); simt_bn_bwd then starts at its finalize (simt_bn_bwd_desc.reduce_done_nblk =
   * simt_conv_mtiles(d)).  mask: bnr_mode 2 = y*scale+shift > 0, 3 = bit mask bnr_bits (simt_bn_apply_bits). */
  const void* bnr_y;             /* [B*Ho*Wo][bnr_ld] dtype_in: the saved pre-BN activation */
  const float *bnr_mean, *bnr_rstd, *bnr_scale, *bnr_shift;
  const unsigned char* bnr_bits;
  float* bnr_part;
  int32_t bnr_mode, bnr_ld;
  const void* w_frag;  /* optional (may be NULL): the SAME weights as `w` in MFMA-fragment order (simt_pack_weight with
                        * SIMT_PACK_FRAG(Npad / 16) in `mode`): [K / 64][Npad / 16][2][64 lanes][8] bf16, i.e. element (row, kcol) of the
                        * K-contiguous matrix sits at ((((kcol / 64) * (Npad / 16) + row / 16) * 2 + (kcol / 32) % 2) * 64 +
                        * (row % 16) + 16 * ((kcol / 8) % 4)) * 8 + kcol % 8.  When present and simt_conv_wants_frag(d) != 0 the wide
                        * bf16 kernel loads its weight operand straight into registers (1 KB contiguous per wave-instruction)
                        * instead of staging it through LDS; results are bit-identical either way */
  const struct simt_fbn_desc* fbn;   /* optional (may be NULL): the train-mode BatchNorm behind this conv fused into the launch, below */
  int32_t cu_budget;   /* ABI 2.  Compute units this launch may plan for; 0 = all of the device (256); -1: A/B only, rounds 1-5 tile choice.  Data-parallel plans pass 256 minus the
                        * CUs the collective's persistent kernels hold (NCCL_MAX_NCHANNELS): the one-workgroup-per-CU tile lists of the wide convs
                        * are planned for that many CUs (M = 37 636: 236 tiles of 160 rows for any budget >= 236 -- the default plan already leaves
                        * 20 CUs free; a smaller budget re-plans).  Changes only the pixel rows per tile: every output element is bit-identical across budgets; the per-tile
                        * BatchNorm partial sums (stats / bnr_part: one slot per tile) regroup, i.e. their fp32 rounding may differ in the last bit. */
  int32_t reserved_;
  /* ABI 2.  BatchNorm + ReLU of the INPUT applied in the operand path (round 6; model/deeplab_multi.py:88-92: out = relu(bn2(conv2)), conv3(out)).
   * When in_scale != NULL the launch reads x as the RAW pre-BatchNorm activation y2, uses a = relu(x * in_scale[c] + in_shift[c]) (rounded to
   * bf16, bitwise what simt_bn_apply writes) as its operand and ALSO writes a to in_out [B*H*W][Cin] (the weight gradient of this conv and
   * the BatchNorm backward read it later) -- the separate simt_bn_apply launch and its re-read of y2 disappear.  Only launches for which
   * simt_conv_inbn_ok(d) != 0 (the row-streaming 1x1 kernel on Cin in {64, 128, 256} with BatchNorm statistics: a Bottleneck's conv3 in the
   * training forward) accept it; everywhere else the fields must be NULL. */
  const float *in_scale, *in_shift;
  void* in_out;
} simt_conv_desc;
/* Train-mode BatchNorm2d (frozen affine; model/deeplab_multi.py:63-70,81-91) fused into the PRODUCING conv launch -- round 4.  For a
 * launch whose workgroups are all co-resident (one round of the chip: simt_conv_fbn_ok), the pixel tile stays in LDS after the epilogue;
 * every workgroup publishes its per-channel tile sums as 8-byte {value, tag} granules (write-through stores), the first C/8 workgroups
 * poll the granules of 8 channels each, add them IN THE ORDER simt_bn_finalize / simt_bn_bwd's finalize use (bitwise the same constants)
 * and publish the constants the same way; every workgroup polls its channels' constants and normalises its tile from LDS:
 *   mode 1 (forward; needs d->stats):       y as usual, out = relu(y * scale + shift)   replaces simt_bn_finalize + simt_bn_apply
 *   mode 2 (backward; needs d->bnr_mode 2): out = scale * (g - c1 - xhat * c2)          replaces simt_bn_bwd; d->y (the raw dz) is NOT written
 * (d->stats / d->bnr_part themselves are not written in fused launches.)  Only a launch on the stream that owns the plan may wait like
 * this: kernels of the other streams (frozen forward, weight gradients) never wait on anything, so the co-residency the polling needs
 * always resolves.  A workgroup that has polled for ~2 s (another process's waiting launch on the same GPU, persistent collective kernels holding
 * CUs) does NOT trap: it sets the sticky error word (`err`, or work[SIMT_FBN_ERR_WORD]), every other poller of the launch sees it and the launch ends;
 * later fused launches that share the word bail out at their first failed poll.  What such a launch guarantees: a poller that gave up writes
 * nothing derived from its incomplete reads -- no constants, no mean / rstd / scale / shift / coef / d gamma / d beta, NO running-statistics
 * update, not its rows of `out` (workgroups whose polls had completed wrote theirs from complete sums: `out` is PARTIALLY written and must be
 * treated as undefined).  The caller reads the word (simt_amd: TrunkPlan.fbn_error(), raised by losses()) and rebuilds the plan with the
 * two-pass BatchNorm; the optimiser launches given the same word (simt_sgd_desc.skip_if, simt_adam_step_guarded) leave weights, momentum
 * buffers and Adam moments untouched while it is set.  `work` belongs to ONE BatchNorm and direction: its ticket counters only ever grow and
 * give every launch its generation = the granules' tag. */
#define SIMT_FBN_BAR_WORDS 144            /* uint64 words at the head of `work`: 8 ticket counters, one 128-byte line each (+ spare) */
#define SIMT_FBN_ERR_WORD 136             /* spare word of the head used as the error word when simt_fbn_desc.err is NULL */
typedef struct simt_fbn_desc {
  int32_t mode;          /* 1 forward, 2 backward */
  int32_t ldo;           /* row pitch of out in elements */
  void* out;             /* [B*Ho*Wo][ldo] bf16 */
  uint64_t* work;        /* [simt_conv_fbn_words(d)] uint64, zeroed ONCE by the caller: counters | [2][C] constants | [tiles][2|3][C] tile sums */
  const float *gamma, *beta;              /* forward: [C] or NULL (1 / 0) */
  float *running_mean, *running_var;      /* forward: updated with `momentum` like simt_bn_finalize, or NULL */
  float momentum, eps;
  float *mean, *rstd, *scale, *shift;     /* forward: OUT [C] each (the backward reads them) */
  float* coef;                            /* backward: OUT [3][C] = (sum g, sum g*xhat, 0) / count */
  float *dgamma, *dbeta;                  /* backward, trainable affine (model/deeplabv3.py's BatchNorm): OUT [C] = sum g*xhat, sum g; or NULL */
  uint64_t* err;                          /* optional: sticky error word shared by the fused launches of a plan (zeroed once by the caller; set
                                           * non-zero by a launch whose polling timed out); NULL: work[SIMT_FBN_ERR_WORD] */
} simt_fbn_desc;
int simt_conv_fprop(const simt_conv_desc* d, simt_stream_t stream);
/* Two INDEPENDENT convs of identical geometry in one launch (round 5): the trainable and the frozen ResNetMulti run the same 104 conv shapes on
 * the same image (tools/trainV2_simt.py:351-353 and :370 -> model/deeplab_multi.py:172-192); their layer-k convs differ only in weights, output
 * and epilogue (BatchNorm statistics | folded bias + ReLU [+ residual]).  d0 / d1: same B, H, W, Cin, Cout, taps, stride, Npad, output dtype.
 * Results are bit-identical to simt_conv_fprop(d0) followed by simt_conv_fprop(d1); pairs the fused kernels do not cover (different tile
 * variant, other epilogue combinations, d->fbn set) ARE run as those two launches.  simt_conv_pair_fused says which it will be. */
int simt_conv_fprop_pair(const simt_conv_desc* d0, const simt_conv_desc* d1, simt_stream_t stream);
int simt_conv_pair_fused(const simt_conv_desc* d0, const simt_conv_desc* d1);
/* 1 if simt_conv_fprop can run d with a fused BatchNorm (d->fbn): bf16 v2 kernel with the 3-slot ring -- 256-column tiles in both directions,
 * 128- / 64-column tiles (the small maps of model/deeplabv3.py) in the backward direction (d->bnr_mode 2) -- and a grid that is co-resident on
 * the current device (tiles <= compute units); d->fbn itself need not be set yet */
int simt_conv_fbn_ok(const simt_conv_desc* d);
long simt_conv_fbn_words(const simt_conv_desc* d);      /* uint64 words simt_fbn_desc.work needs for d (0 if !simt_conv_fbn_ok) */
/* 1 if the launch for d would use d->w_frag when given (conv_igemm2_kernel<256, *, 3>: Npad tiles of 256, long reductions) */
int simt_conv_wants_frag(const simt_conv_desc* d);
/* `mode` of simt_pack_weight / PackJob: low byte = layout mode 0 / 1 / 2 below; SIMT_PACK_FRAG(nt16) additionally stores the
 * destination in MFMA-fragment order (see simt_conv_desc.w_frag) with nt16 = Npad / 16 row blocks */
#define SIMT_PACK_FRAG(nt16) ((int)(nt16) << 8)
/* which kernel instantiation simt_conv_fprop runs for d (profiling / reporting / tests that must hit a given instantiation):
 * returns 0 (conv_igemm_kernel, fp32 parity + narrow outputs), 2 (conv_igemm2_kernel<bn, tm, nst>, the bf16 throughput kernel),
 * 4 (conv1x1_stream_kernel) or 5 (conv1x1_rows_kernel): the short-reduction / wide-output 1x1 shapes */
int simt_conv_variant(const simt_conv_desc* d, int* bn, int* tm, int* nst);
/* compile-time epilogue flavour of the bf16 v2 kernel the launch for d runs (the last template argument of conv_igemm2_kernel<bn, tm, nst,
 * fbn, epi>): 0 generic (run-time flags), 1 BatchNorm statistics, 2 fused BatchNorm-backward reduce (mask from y * scale + shift), 3 bias + ReLU,
 * 4 bit-masked residual + BatchNorm-backward reduce with the bit mask, 5 plain, 6 residual only, 7 BatchNorm-backward reduce with the bit mask
 * (+ optional plain residual), 8 ReLU-mask operand only (the BatchNorm-free VGG dgrads); the 2-slot short-K kernels are instantiated for 0 and
 * 4-8 only (1-3 are reported as 0 there); same results either way */
int simt_conv_epilogue_flavour(const simt_conv_desc* d);
/* number of pixel tiles (= statistics / bnr_part slots) the launch for d uses; 0 if d does not run on the bf16 v2 kernel */
int simt_conv_mtiles(const simt_conv_desc* d);
int simt_conv_inbn_ok(const simt_conv_desc* d);        /* ABI 2: may d carry in_scale / in_shift / in_out (see simt_conv_desc)? */

/* ---- convolution: wgrad (split-K over pixels, transposed MFMA operands) -------------------------------
 * slab[split][co][tap*Cin+ci] = sum_{m in split} dy[m][co] * x[pixel(m)*stride + (dy,dx)[tap]][ci]
 * then simt_wgrad_reduce sums the splits in fixed order into the OIHW fp32 gradient.
 * replaces the weight gradient autograd computes for the convs above (tools/trainV2_simt.py:428). */
typedef struct {
  const void* dy;      /* [B*Ho*Wo][ldd] dtype; channels >= Cd are ignored */
  const void* x;       /* [B][H][W][Cin] dtype */
  float* slab;         /* [nsplit][Cd][ntaps*Cin] fp32 workspace */
  int32_t B, H, W, Cin, Ho, Wo, Cd, ldd, stride, ntaps, nsplit, dtype;
  int16_t dy_[SIMT_MAX_TAPS], dx_[SIMT_MAX_TAPS];
} simt_wgrad_desc;
int simt_conv_wgrad(const simt_wgrad_desc* d, simt_stream_t stream);
int simt_wgrad_reduce(const float* slab, float* dst, int nsplit, int Cd, int Ktot, int Cin, int co_off, int tap_off,
                      int Cout, int RS, int accumulate, simt_stream_t stream);
/* Grouped launch: n <= SIMT_WGRAD_MULTI_MAX problems with the SAME nsplit (the convs of one Bottleneck: the same pixels) as one tile list,
 * so that a much coarser pixel split fills the chip (fewer, longer slabs; one launch).  Each problem keeps its own slab and its own
 * simt_wgrad_reduce; results equal simt_conv_wgrad's with that nsplit bit for bit.  The per-problem arguments live in a DEVICE table
 * the caller owns: simt_conv_wgrad_multi_prepare fills its host image (n * simt_conv_wgrad_multi_bytes() bytes) and returns the grid;
 * the caller copies it to device memory once and passes that to simt_conv_wgrad_multi.  simt_conv_wgrad_multi_ok: does the problem
 * qualify (the problems simt_conv_wgrad runs on its 128x256-tile bf16 kernel)? */
#define SIMT_WGRAD_MULTI_MAX 16
int simt_conv_wgrad_multi_ok(const simt_wgrad_desc* d);
int simt_conv_wgrad_multi_bytes(void);
int simt_conv_wgrad_multi_prepare(const simt_wgrad_desc* d, int n, void* table_host, int* grid, int* tile_co);
int simt_conv_wgrad_multi(const void* table_dev, int n, int grid, int nsplit, int tile_co, simt_stream_t stream);
/* dY channels per output tile of the bf16 kernel for this problem: 256 (Cd a multiple of 256 and enough pixels: conv_wgrad3, 256 x 256 tile, half the
 * staged bytes per MAC) or 128; a grouped launch takes 256 only if every problem does (returned by _prepare in *tile_co).  The split
 * count that fills the chip depends on it: tiles = ceil(Cd / tile_co) * ceil(ntaps * Cin / 256). */
#define SIMT_WGRAD3_MIN_PIXELS 16384   /* the 256-row tile needs B * Ho * Wo >= this (fewer pixels: the extra splits cost more than they save) */
int simt_conv_wgrad_tile_co(const simt_wgrad_desc* d);
/* The reduces of a grouped launch as ONE launch: a device table of n <= 2 * SIMT_WGRAD_MULTI_MAX jobs (the arguments of simt_wgrad_reduce; needs Cin % 4 == 0,
 * Ktot % 4 == 0, 16-byte aligned slab and dst).  Job j owns blocks [block0, block0 + simt_wgrad_reduce_blocks(Cout, RS, Cin)); `blocks` = their
 * total.  Per element the same sum in the same order as simt_wgrad_reduce: bitwise the same gradients. */
typedef struct {
  const float* slab;
  float* dst;
  int32_t nsplit, Cd, Ktot, Cin, co_off, tap_off, Cout, RS, accumulate, block0;
} simt_wgrad_reduce_job;
int simt_wgrad_reduce_multi(const simt_wgrad_reduce_job* jobs_dev, int n, int blocks, simt_stream_t stream);
int simt_wgrad_reduce_blocks(int Cout, int RS, int Cin);   /* 256-thread blocks of one job (3x3: a thread owns 4 channels x 9 taps) */

/* ---- tap-expanded ASPP classifier (bf16 throughput path; model/deeplab_multi.py:104-119) -------------------------
 * The N = Q = 22 dilated conv is re-associated into a plain GEMM with one output column per (tap, class) plus a
 * tap gather-sum (forward) / tap scatter (backward); see csrc/head_expand.hip.
 *   simt_tap_gather_sum: dst[m][n] = bias[n] + sum_t src[m + (dy,dx)[t]][t*QP + n]       (fp32 -> fp32)
 *   simt_tap_scatter   : dst[m][t*QP + n] = src[m - (dy,dx)[t]][n], 0 outside the image   (bf16 -> bf16)
 *   simt_wgrad_reduce_exp: slab[split][(tap_off+t)*QP + row_off + co][ci] -> OIHW fp32 gradient */
typedef struct {
  const void* src;
  const float* bias;   /* [Q] or NULL (gather_sum only) */
  void* dst;
  int32_t B, H, W, Q, QP, lds, ldd, ntaps;   /* lds / ldd: row pitch of src / dst in elements */
  int16_t dy[SIMT_MAX_TAPS], dx[SIMT_MAX_TAPS];
} simt_tap_desc;
int simt_tap_gather_sum(const simt_tap_desc* d, simt_stream_t stream);
int simt_tap_scatter(const simt_tap_desc* d, simt_stream_t stream);
int simt_wgrad_reduce_exp(const float* slab, float* dst, int nsplit, int Cd, int Cin, int QP, int row_off, int tap_off,
                          int Cout, int RS, simt_stream_t stream);

/* ---- weight packing / BN folding ------------------------------------------------------------------------ */
int simt_pack_weight(const float* w, void* dst, int Cout, int Cin, int RS, int row_off, int tap_off, long ldk, int Ck,
                     int mode, const float* cscale, int dtype, simt_stream_t stream);
/* batched form: jobs = device array of {const float* w; void* dst; const float* cscale; int64 ldk, total; int32 Cout, Cin, RS,
 * row_off, tap_off, Ck, mode, dtype} (72 bytes), chunks = device int32 [nchunks][2] (job, 32x32 (cout, cin) tile index) */
int simt_pack_weight_multi(const void* jobs, const void* chunks, int nchunks, int chunk, simt_stream_t stream);
int simt_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale,
                 float* shift, int C, simt_stream_t stream);

/* ---- BatchNorm2d, train mode with frozen affine (model/deeplab_multi.py:63-76; quirk: batch stats are used and
 * back-propagated through although gamma/beta never change) ------------------------------------------------ */
int simt_bn_finalize(const float* part, int nblk, int C, long count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* mean, float* rstd,
                     float* scale, float* shift, simt_stream_t stream);
int simt_bn_apply(const void* y, const float* scale, const float* shift, const void* res, const void* y2,
                  const float* scale2, const float* shift2, void* z, long M, int C, int relu, int dtype,
                  simt_stream_t stream);
/* same, and bits[(m*C + c) / 8] bit (c & 7) = (z[m][c] > 0): the ReLU mask of the block output (model/deeplab_multi.py:99-100)
 * for simt_bn_bwd's mask_mode 3 -- the backward then reads M*C/8 bytes instead of z (M*C elements) in both of its passes */
int simt_bn_apply_bits(const void* y, const float* scale, const float* shift, const void* res, const void* y2,
                       const float* scale2, const float* shift2, void* z, unsigned char* bits, long M, int C, int relu,
                       int dtype, simt_stream_t stream);
typedef struct {
  const void* dz;      /* [M][C] upstream gradient */
  const void* z;       /* [M][C] block output (mask_mode 1) or NULL */
  const void* y;       /* [M][C] raw conv output */
  const float *mean, *rstd, *scale, *shift;
  const void* y2;      /* second BN sharing the masked gradient (downsample branch) or NULL */
  const float *mean2, *rstd2, *scale2;
  float* part;         /* [simt_bn_bwd_nblk(M,C)][3][C] workspace */
  float* coef;         /* [3][C] workspace */
  void* dy;            /* [M][C] gradient wrt y */
  void* dy2;           /* [M][C] gradient wrt y2 or NULL */
  void* gout;          /* [M][C] masked gradient (may alias dz) or NULL */
  int64_t M;
  int32_t C, mask_mode, dtype; /* mask_mode: 0 none, 1 z>0, 2 y*scale+shift>0, 3 z = bit mask of simt_bn_apply_bits */
  float *dgamma, *dbeta;   /* [C] or NULL: gradients of a TRAINABLE affine (model/deeplabv3.py's torchvision BatchNorm) */
  float *dgamma2, *dbeta2; /* same for the second BN (y2) */
  int32_t reduce_done_nblk; /* > 0: `part` already holds that many [3][C] partial slots (written by the conv that produced dz,
                             * simt_conv_desc.bnr_*): skip the reduce pass */
} simt_bn_bwd_desc;
int simt_bn_bwd_nblk(long M, int C);
int simt_bn_bwd(const simt_bn_bwd_desc* d, simt_stream_t stream);

/* ---- stem: 7x7/s2 conv via im2col, BN+ReLU+MaxPool(3,2,1,ceil) (model/deeplab_multi.py:127-133,172-176) --- */
/* ---- direct 7x7 stride-2 pad-3 stem convolution (round 6; model/deeplab_multi.py:127,172-173: conv1 of the ResNets) ----
 * Replaces simt_im2col_stem + simt_conv_fprop on the im2col matrix for the bf16 throughput mode: the image patch of an 8 x 32 output tile is
 * staged in LDS as [row][col][channel], a filter ROW (7 taps x 3 channels = 21 contiguous values) is one 32-deep MFMA k-step.  Up to TWO weight
 * sets in one launch (the trainable and the frozen network convolve the same image, tools/trainV2_simt.py:351-353,370): per set
 *   y[s][m][0..63] = act( sum_{r,s,c} x[b][c][2 oy - 3 + r][2 ox - 3 + s] * w[s][o][r][s*3 + c] + bias[s][o] ),  bf16 NHWC,
 * stats[s] (optional): [simt_stem7_tiles][2][64] per-tile sum / sum of squares of the STORED values (BatchNorm batch statistics; hand
 * simt_stem7_tiles(...) as the slot count to simt_bn_finalize).  w[s]: simt_stem7_pack(conv1.weight [64][3][7][7] fp32, per-channel scale
 * or NULL) -> bf16 [64][7][32].  The weight gradient of the stem still runs on the im2col matrix (built in the backward). */
typedef struct {
  const float* x;          /* [B][3][H][W] fp32 (the reference's NCHW input tensor) */
  int32_t B, H, W, Ho, Wo;
  int32_t nsets;           /* 1 | 2 */
  const void* w[2];
  void* y[2];              /* [B*Ho*Wo][64] bf16 */
  const float* bias[2];    /* [64] or NULL */
  int32_t relu[2];
  float* stats[2];         /* or NULL */
} simt_stem_desc;
int simt_stem7_tiles(int B, int Ho, int Wo);
int simt_stem7_pack(const float* w_oihw, const float* cscale, void* dst, simt_stream_t stream);
int simt_stem7_fwd(const simt_stem_desc* d, simt_stream_t stream);
/* Weight gradient of the stem convolution straight from the image (replaces simt_im2col_stem + simt_conv_wgrad + simt_wgrad_reduce for conv1,
 * tools/trainV2_simt.py:428): dw[o][c][r][s] = sum_pixels dy[p][o] * x[b][c][2 oy - 3 + r][2 ox - 3 + s], fp32 accumulation, the per-workgroup partials
 * added in fixed order (bitwise reproducible).  part: [simt_stem7_wgrad_workgroups(B, Ho, Wo)][64 * 7 * 32] fp32 workspace; dw is overwritten. */
int simt_stem7_wgrad_workgroups(int B, int Ho, int Wo);
int simt_stem7_wgrad(const float* x_nchw, const void* dy, float* part, float* dw, int B, int H, int W, int Ho, int Wo, simt_stream_t stream);

int simt_im2col_stem(const float* x_nchw, void* A, int B, int Cin, int H, int W, int Ho, int Wo, int KH, int KW,
                     int stride, int pad, int ldk, int dtype, simt_stream_t stream);
int simt_bn_relu_maxpool(const void* y, const float* scale, const float* shift, void* p, unsigned char* idx, int B,
                         int H, int W, int C, int Hp, int Wp, int dtype, simt_stream_t stream);
int simt_maxpool_bwd(const void* dp, const unsigned char* idx, void* da, int B, int H, int W, int C, int Hp, int Wp,
                     int dtype, simt_stream_t stream);
int simt_scatter_stride(const void* src, void* dx, int B, int H, int W, int C, int Ho, int Wo, int stride, int dtype,
                        simt_stream_t stream);
int simt_colsum(const void* src, float* out, long M, int ld, int C, int accumulate, int dtype, simt_stream_t stream);
/* VGG trunk (model/deeplab_vgg.py:24-43): MaxPool2d(2,2) forward (+ arg-max bytes) and its backward fused with the ReLU
 * mask of the conv output y it pooled; bias gradients of wide layers (C <= 2048) */
int simt_maxpool2(const void* y, void* p, unsigned char* idx, int B, int H, int W, int C, int dtype, simt_stream_t stream);
int simt_maxpool2_bwd(const void* dp, const unsigned char* idx, const void* y, void* da, int B, int H, int W, int C, int dtype,
                      simt_stream_t stream);
int simt_colsum_wide(const void* src, float* out, long M, int ld, int C, int dtype, simt_stream_t stream);

/* ---- fused SimT head: upsample + softmax + per-pixel loss terms + anchors (tools/trainV2_simt.py:351-409,
 * :202-230 Placeholder_loss, utils/loss.py:14-40) ------------------------------------------------------- */
typedef struct {
  const float* pred1;   /* [B*h*w][ldp] fp32 low-res logits of the aux head (first Q channels) */
  const float* pred2;   /* [B*h*w][ldp] main head */
  const float* fixp;    /* [B*h*w][ldf] softmax probabilities of the frozen model's main head (first C channels) */
  const int64_t* label; /* [B][H][W] noisy pseudo labels, 255 = ignore */
  const float* T1;      /* [Q][C] transition matrices (simt_ntm_inner_loop / simt_sig_ntm) */
  const float* T2;
  float* part;          /* [simt_head_nblk][simt_head_part_floats] workspace */
  void* keys;           /* [simt_head_keys_count] u64 workspace (zeroed by the call) */
  float* hout;          /* [simt_head_hout_floats] result block, layout in csrc/head_loss.hip */
  float* g1;            /* [2][B][H][w][QP] workspace (simt_head_grad) */
  float* dpred1_f32;    /* [B*h*w][ld_f32] gradient outputs (any may be NULL) */
  float* dpred2_f32;
  void* dpred1_t;       /* [B*h*w][ld_t] gradient in grad_dtype for the dgrad GEMM (pad columns untouched) */
  void* dpred2_t;
  int32_t B, h, w, H, W, C, Q, ldp, ldf, QP, ld_f32, ld_t, grad_dtype;
  float th_high, th_low, lambda_seg, lambda_place, gscale;
  int32_t mode;         /* 0: SimT loss block.  1: warm-up stage (tools/trainV1_warmup.py:217-224): CE of both heads against
                         * `label` (ignore 255); fixp/T1/T2 unused (may be NULL); hout[0],[1] = loss_seg1/2, hout[14] = total */
  int32_t single;       /* 1: one-output model (model/deeplabv3.py, model/deeplab_vgg.py return ONE tensor): pred1 / T1 / dpred1_* are
                         * unused (may be NULL) and every auxiliary-head term is dropped; pred2 / T2 / dpred2_* carry the model's head */
  int32_t up_half_pixel; /* 0: interp_target = nn.Upsample(bilinear, align_corners=True) (tools/trainV2_simt.py:301);
                          * 1: F.interpolate(bilinear), align_corners=False, the in-model upsample of model/deeplabv3.py:137 fused here */
  int32_t fix_logits;   /* 1: fixp holds the frozen model's low-res LOGITS; posterior = softmax(upsample(logits)) -- what
                         * trainV2_simt.py:354 computes for a model that upsamples inside.  0: fixp = low-res probabilities */
  uint8_t* conf_out;    /* optional (may be NULL): [B][H][W] the per-pixel `Conf_label_target` of trainV2_simt.py:357-362,387-393
                         * as simt_head_loss decided it: class index 0..Q-1, 255 = no confidence label (mode 1: the label itself) */
  uint8_t* label_ws;    /* optional workspace (may be NULL): [B][H][W].  With conf_out AND label_ws given, simt_head_loss also stores each pixel's
                         * checked noisy label and simt_head_grad reads the two byte maps back (they must still hold what simt_head_loss of the
                         * same inputs wrote) instead of deciding the labels again from fixp and the 8-byte labels: same values, fewer bytes */
} simt_head_desc;
int simt_head_nblk(int B, int H, int W);
int simt_head_part_floats(int Q, int C);
int simt_head_hout_floats(int Q, int C);
int simt_head_keys_count(void);
int simt_softmax_rows(const float* in, int ldi, float* out, int ldo, long M, int C, simt_stream_t stream);
int simt_head_loss(const simt_head_desc* d, simt_stream_t stream);
int simt_head_grad(const simt_head_desc* d, simt_stream_t stream);

/* ---- NTM micro-solver (model/deeplab_multi.py:244-286, tools/trainV2_simt.py:326-339,412-424,435-436) ---- */
typedef struct {
  float* ntm[2];        /* [Q][C] NTM1/NTM2 parameters */
  float* w[2];          /* [Q][Q] sig_W weights (updated in place by Adam; diag := -1e4) */
  float* ntm_grad[2];   /* [Q][C] accumulated (+=): the inner-loop leak, SURVEY quirk 3 */
  float* w_m[2];        /* Adam exp_avg of w */
  float* w_v[2];        /* Adam exp_avg_sq of w */
  float* T_out[2];      /* [Q][C] T = sig_NTM() */
  const float* class_dist; /* [C] */
  int32_t Q, C, steps, step0; /* step0 = Adam steps already taken on w */
  float lr, beta1, beta2, eps;
  int32_t single;       /* 1: only NTM / W number 1 (index [1]) exist -- one-output models; index [0] pointers may be NULL */
  const uint64_t* skip_if;   /* ABI 2.  Optional device word (simt_fbn_desc.err): the launch changes nothing while it is non-zero; NULL: always run */
} simt_ntm_inner_desc;
int simt_ntm_inner_loop(const simt_ntm_inner_desc* d, simt_stream_t stream);
typedef struct {
  const float* ntm[2];
  float* w[2];
  float* ntm_grad[2];
  const float* class_dist;
  const float* hout;    /* from simt_head_loss */
  float* lout;          /* [16]: total*gscale, loss_p1, loss_p2, loss_y1, loss_y2, Place, Convex, Volume, Anchor, vol_ok, log-vol x2,
                           [12] += hout[15] (labels outside [0,C) and != 255, ACCUMULATED until the caller clears it) */
  int32_t Q, C;
  float lambda_seg, lambda_convex, lambda_volume, lambda_anchor, gscale;
  int32_t single;       /* 1: Convex / Volume / Anchor / total over NTM [1] only, no lambda_seg terms */
} simt_ntm_post_desc;
int simt_ntm_post(const simt_ntm_post_desc* d, simt_stream_t stream);
int simt_sig_ntm(const float* ntm, const float* class_dist, const float* dT, float* T_out, float* dN_out, int Q, int C,
                 simt_stream_t stream);
int simt_sig_w(float* weight, const float* dW, float* W_out, float* dweight_out, int Q, simt_stream_t stream);
int simt_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                   int step, simt_stream_t stream);
/* ABI 2: the same step, skipped (parameter AND moments untouched) while *skip_if != 0 -- the guard simt_sgd_desc.skip_if gives the SGD launch */
int simt_adam_step_guarded(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                           int step, const uint64_t* skip_if, simt_stream_t stream);

/* ---- fused SGD with duplicate-listing semantics (tools/trainV2_simt.py:296-297,434; model/deeplab_multi.py:194-237) --- */
typedef struct {
  const void* segs;    /* device array of {float* p; const float* g; float* buf; int64 n; int32 mult; int32 group} */
  const void* chunks;  /* device int32 [nchunks][2]: (segment index, chunk index inside the segment) */
  int32_t nchunks, chunk;
  float lr[4], wd[4];  /* per param group */
  float momentum, dampening;
  int32_t first_step;  /* 1: momentum buffers are created (= d_p) like torch's first step */
  const uint64_t* skip_if;   /* optional device word (simt_fbn_desc.err): the launch changes nothing while it is non-zero; NULL: always update */
} simt_sgd_desc;
int simt_sgd_multi(const simt_sgd_desc* d, simt_stream_t stream);
/* dst[i] (+)= src[i], fp32: bias of the fused 2-branch ASPP GEMM = sum of the branch biases (deeplab_multi.py:115-119) */
int simt_vec_acc(float* dst, const float* src, int n, int accumulate, simt_stream_t stream);

/* ---- utils/loss.py: CrossEntropy2d (:6-40) and EntropyLoss (:42-49) on NCHW fp32 predictions ----------------------
 * ws: simt_loss_ws_bytes() bytes of device scratch.  out[0] = loss (mean over valid pixels, weighted like
 * F.cross_entropy / F.nll_loss with reduction='mean'), out[1] = denominator.  Zero valid pixels -> NaN (reference). */
int simt_loss_ws_bytes(void);
int simt_ce2d_fwd(const float* pred, const int64_t* target, const float* weight, int n, int c, int h, int w,
                  int ignore_label, int is_softmax, void* ws, float* out, simt_stream_t stream);
int simt_ce2d_bwd(const float* pred, const int64_t* target, const float* weight, int n, int c, int h, int w,
                  int ignore_label, int is_softmax, const float* out, const float* grad_out, float* dpred,
                  simt_stream_t stream);
int simt_entropy2d(const float* x, int n, int c, int h, int w, void* ws, float* out, const float* grad_out, float* dx,
                   simt_stream_t stream);

/* ---- evaluation (tools/evaluate_cityscapes.py:96-162 evaluate_simt; :81-83 fast_hist) --------------------------------
 * pred[b][y][x] = argmax_c ( up(la)[c] + up(lb)[c] ), bilinear align_corners=True to (H, W), first index on ties;
 * lb may be NULL (single scale).  hist[n*gt + pred] += 1 for 0 <= gt < n (int64, accumulates across calls). */
int simt_upsample_sum_argmax(const float* la, int ha, int wa, int lda, const float* lb, int hb, int wb, int ldb, int B, int H,
                             int W, int C, int32_t* pred, simt_stream_t stream);
int simt_confusion_hist(const int64_t* gt, const int32_t* pred, long P, int n, int64_t* hist, simt_stream_t stream);
/* F.interpolate(bilinear) of an NHWC fp32 map [B][h][w][lds] (first C channels) to NCHW fp32 [B][C][H][W] and its adjoint
 * (model/deeplabv3.py:137 upsamples inside the model with align_corners=False; align_corners=1 = interp_target) */
int simt_upsample_nchw(const float* src, int B, int h, int w, int lds, int C, int H, int W, int align_corners, float* dst,
                       simt_stream_t stream);
int simt_upsample_nchw_bwd(const float* ddst, int B, int h, int w, int lds, int C, int H, int W, int align_corners,
                           void* dsrc, int dtype, float* tmp, simt_stream_t stream); /* dsrc [B][h][w][lds] in dtype, first C
                           * channels; tmp: caller-owned scratch of B*C*H*w floats (the x-folded intermediate of the separable adjoint) */

/* ---- offline NTM utilities (tools/compute_ClassDistribution.py:49-51,66-86 `fast_hist(a, n)` = bincount of the pseudo labels;
 * tools/compute_ConfusionMatrix.py:54-56,68-98 `fast_hist(a, b, n33, n19)` = bincount(n19 * a + b) after label_mapping) -------------
 * hist[lut[a[p]] * nb + b[p]] += 1 over uint8 images (int64, accumulates across calls); a == NULL: one row (class distribution);
 * lut: 256-entry label_mapping table or NULL (identity); entries with row >= na or b >= nb (255 = ignore) are skipped. */
int simt_hist2d_u8(const unsigned char* a, const unsigned char* b, long P, int na, int nb, const unsigned char* lut, int64_t* hist,
                   simt_stream_t stream);

/* ---- input pipeline (dataset/cityscapes_dataset.py:101-120 after PNG decoding) -----------------------------------------
 * The reference resizes with Pillow on the CPU (Image.resize BICUBIC / NEAREST) and converts to float32 BGR - mean, CHW.
 * simt_resample_u8 applies ONE pass of Pillow's 8-bit separable resampler (Resample.c: 22-bit fixed-point coefficients,
 * result = clip8(((1 << 21) + sum) >> 22)) along x (axis 1: [N][H][W][C] -> [N][H][out][C]) or y (axis 0: -> [N][out][W][C]);
 * bounds [out][2] = (first source index, count) and kk [out][ksize] are the tables of precompute_coeffs /
 * normalize_coeffs_8bpc, computed by the caller in double: the output equals Pillow's byte for byte.  C in {1, 3}.
 * simt_image_to_input: [N][H][W][3] u8 RGB -> [N][3][H][W] fp32, channel c = rgb[2 - c] - mean_c (BGR - IMG_MEAN, :113-116);
 * rgb_order = 1 keeps RGB (the reference's --random-mirror branch reverses the channel axis, :108-111).
 * simt_label_nearest: dst[n][y][x] = (int64) src[n][ytab[y]][xtab[flip_x ? Wo-1-x : x]] (ImagingScaleAffine index tables). */
int simt_resample_u8(const unsigned char* src, unsigned char* dst, int N, int H, int W, int C, int out, int axis,
                     const int* bounds, const int* kk, int ksize, simt_stream_t stream);
int simt_image_to_input(const unsigned char* rgb, float* x, int N, int H, int W, float mean0, float mean1, float mean2,
                        int rgb_order, simt_stream_t stream);
int simt_label_nearest(const unsigned char* src, long long* dst, int N, int H, int W, int Ho, int Wo, const int* ytab,
                       const int* xtab, int flip_x, simt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
