"""Tensor-level wrappers over the C ABI (include/simt_hip.h).

PyTorch is used only for device memory and streams: every function takes CUDA(=HIP) tensors, passes raw device
pointers + the current stream to libsimt_hip.so and returns without synchronising.  No function here computes
anything in torch; there is no CPU fallback.
"""
import ctypes as C

import torch

from . import _lib as L

F32, BF16 = L.SIMT_F32, L.SIMT_BF16


def dt_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {dtype}")


def _p(t):
    if t is None:
        return None
    assert t.is_cuda, "libsimt_hip needs device tensors (no CPU fallback)"
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def conv_taps(R, S, dil, pad):
    """(dy, dx) per filter tap, r-major."""
    return [(r * dil - pad, s * dil - pad) for r in range(R) for s in range(S)]


def _fill_taps(arr_dy, arr_dx, taps):
    assert len(taps) <= L.MAX_TAPS
    for i, (a, b) in enumerate(taps):
        arr_dy[i] = a
        arr_dx[i] = b


def pick_tile_n(cout, dtype=None):
    """Output-channel tile of the implicit GEMM; 256 exists only in the bf16 throughput kernel (conv_igemm2.hip)."""
    if cout <= 32:
        return 32
    if cout <= 64:
        return 64
    if cout <= 128 or dtype != torch.bfloat16:
        return 128
    return 256


def round_up(a, b):
    return (a + b - 1) // b * b


def packed_rows(cout, tile_n=None, dtype=None):
    tile_n = tile_n or pick_tile_n(cout, dtype)
    return round_up(cout, tile_n)


def make_conv_desc(x, w, y, *, B, H, W, Cin, Ho, Wo, Cout, taps, stride=1, bias=None, res=None, stats=None, relu=False,
                   Npad=None, Nstore=None, ldy=None, ldr=None, tile_n=None, mask=None, ldm=None, res_bits=None, bnr=None, w_frag=None):
    """bnr: fused first pass of the BatchNorm backward -- dict(y, mean, rstd, part, mode (2 | 3), scale, shift | bits).
    w_frag: the same weights in MFMA-fragment order (simt_conv_desc.w_frag; `frag_order(w)` or a pack with PACK_FRAG) or None."""
    d = L.ConvDesc()
    tile_n = tile_n or pick_tile_n(Cout, x.dtype if x.dtype == y.dtype else None)
    d.x, d.w, d.y = _p(x), _p(w), _p(y)
    d.bias, d.res, d.stats = _p(bias), _p(res), _p(stats)
    d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, Cin, Ho, Wo, Cout
    d.Npad = Npad if Npad is not None else round_up(Cout, tile_n)
    d.ldy = ldy if ldy is not None else y.shape[-1]
    d.Nstore = Nstore if Nstore is not None else min(round_up(Cout, 8), d.ldy)
    d.ldr = ldr if ldr is not None else (res.shape[-1] if res is not None else 0)
    d.stride, d.ntaps, d.relu = stride, len(taps), int(relu)
    d.dtype_in, d.dtype_out, d.tile_n = dt_code(x.dtype), dt_code(y.dtype), tile_n
    _fill_taps(d.dy, d.dx, taps)
    d.mask = _p(mask)
    d.ldm = ldm if ldm is not None else (mask.shape[-1] if mask is not None else 0)
    d.res_bits = _p(res_bits)
    d.w_frag = _p(w_frag)
    if bnr:
        d.bnr_y, d.bnr_mean, d.bnr_rstd, d.bnr_part = _p(bnr["y"]), _p(bnr["mean"]), _p(bnr["rstd"]), _p(bnr["part"])
        d.bnr_scale, d.bnr_shift, d.bnr_bits = _p(bnr.get("scale")), _p(bnr.get("shift")), _p(bnr.get("bits"))
        d.bnr_mode, d.bnr_ld = bnr["mode"], bnr["y"].shape[-1]
    return d


def PACK_FRAG(npad):
    """`mode` flag of pack_weight: store the operand in MFMA-fragment order (include/simt_hip.h SIMT_PACK_FRAG(Npad / 16))."""
    assert npad % 16 == 0
    return (npad // 16) << 8


def frag_order(wp):
    """K-contiguous packed operand [Npad][Ktot] -> the same elements in MFMA-fragment order (simt_conv_desc.w_frag):
    [Ktot/64][Npad/16][2][lane = (k/8)%4 * 16 + row%16][8].  Host-side restatement of csrc/bn_pool.hip frag_offset (tests, A/B tools)."""
    npad, kt = wp.shape
    assert npad % 16 == 0 and kt % 64 == 0
    return wp.view(npad // 16, 16, kt // 64, 2, 4, 8).permute(2, 0, 3, 4, 1, 5).contiguous().view(npad, kt)


def conv_wants_frag(d):
    return bool(L.load().simt_conv_wants_frag(C.byref(d)))


def conv_fprop_desc(d):
    L.call("simt_conv_fprop", C.byref(d), stream_ptr())


def make_wgrad_desc(dy, x, slab, *, B, H, W, Cin, Ho, Wo, Cd, taps, stride=1, nsplit=1, ldd=None):
    d = L.WgradDesc()
    d.dy, d.x, d.slab = _p(dy), _p(x), _p(slab)
    d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cd = B, H, W, Cin, Ho, Wo, Cd
    d.ldd = ldd if ldd is not None else dy.shape[-1]
    d.stride, d.ntaps, d.nsplit, d.dtype = stride, len(taps), nsplit, dt_code(x.dtype)
    _fill_taps(d.dy_, d.dx_, taps)
    return d


def conv_wgrad_desc(d):
    L.call("simt_conv_wgrad", C.byref(d), stream_ptr())


_WGRAD_PLAN = {"cus": 128, "epi": 10}      # split-count cost model of the weight-gradient launches; plans may override (wgrad_plan)


class wgrad_plan:
    """with ops.wgrad_plan(cus, epi): ... -- the cost-model constants for the plans built inside (engine_vgg: the BatchNorm-free VGG trunk has no
    long dgrad / BatchNorm chain for the weight gradients to hide behind; its step is 6 % FASTER with the whole-chip plan (256, 6): 661 vs 620
    images/s, profiles/r06_wgrad_split.txt).  SIMT_WGRAD_CUS / SIMT_WGRAD_EPI_STAGES override everything."""

    def __init__(self, cus, epi):
        self.new = {"cus": cus, "epi": epi}

    def __enter__(self):
        self.old = dict(_WGRAD_PLAN)
        _WGRAD_PLAN.update(self.new)

    def __exit__(self, *exc):
        _WGRAD_PLAN.update(self.old)


def _wgrad_cus():
    """CUs the weight-gradient launches plan their rounds for (SIMT_WGRAD_CUS at plan construction).  Round 6: 128, not the 256 the device has --
    the weight gradients run on the side stream BESIDE the dgrad / BatchNorm chain, which is the critical path of the backward (13.5 ms of chain
    against ~4 ms of weight-gradient work): split counts planned for half the chip give the minor launches (layer 4's group, layer3.0's, the
    heads', layer 1-2's, the stem's) about half the workgroups and half the fp32 slab traffic; they take longer alone (layer3.0's group 178 ->
    329 us) and the step is 0.25-0.3 ms SHORTER (profiles/r06_wgrad_split.txt: same-box A/B on three boxes).  The seven 255-workgroup launches
    of layer 3 keep their plan (5 splits of 51 tiles) under either setting; forcing THEM small costs +1.2 ms."""
    import os
    return int(os.environ.get("SIMT_WGRAD_CUS", _WGRAD_PLAN["cus"]))


def _wgrad_epi():
    """Per-workgroup fixed cost of a weight-gradient launch in 64-pixel stages (prologue + the fp32 tile written to its slab + its share of the
    reduce pass): the constant of the split-count cost model (SIMT_WGRAD_EPI_STAGES at plan construction)."""
    import os
    return int(os.environ.get("SIMT_WGRAD_EPI_STAGES", _WGRAD_PLAN["epi"]))


def _wgrad_max_wg():
    """Upper bound on the workgroups of ONE weight-gradient launch (SIMT_WGRAD_MAX_WG at plan construction; 0 = none)."""
    import os
    return int(os.environ.get("SIMT_WGRAD_MAX_WG", "0"))


def wgrad_nsplit(M, Cd, Ktot, dtype, target_wg=768):
    """Split factor over the pixel dimension.  bf16 / Cd >= 64 / Ktot >= 64 runs the 128x256-tile kernel with one
    workgroup per CU: aim at a whole number of 256-CU rounds; otherwise the 128x128 kernel at ~3 workgroups per CU."""
    if dtype == torch.bfloat16 and Cd >= 64 and Ktot >= 64:
        tiles = wgrad_tiles(M, Cd, Ktot)
        max_split = max(1, M // (64 * 8))
        best, best_cost = 1, None
        for ns in range(1, min(max_split, 256) + 1):     # one-tile problems (layer1, stem) need up to 256 splits to fill the chip
            if _wgrad_max_wg() and ns > 1 and tiles * ns > _wgrad_max_wg():
                break
            rounds = -(-tiles * ns // _wgrad_cus())
            stages = -(-M // (ns * 64))
            cost = rounds * (stages + _wgrad_epi())            # + epilogue/prologue per workgroup
            if best_cost is None or cost < best_cost:
                best, best_cost = ns, cost
        return best
    tiles = ((Cd + 127) // 128) * ((Ktot + 127) // 128)
    bp = 64 if dtype == torch.bfloat16 else 32
    max_split = max(1, M // (bp * 4))
    return int(max(1, min(max_split, (target_wg + tiles - 1) // tiles, 64)))


def wgrad_group_nsplit(M, tiles, max_cap=256):
    """Pixel split count of a grouped weight-gradient launch with `tiles` output tiles in all (simt_conv_wgrad_multi): whole 256-CU
    rounds, the same cost model as wgrad_nsplit."""
    max_split = max(1, M // (64 * 8))
    best, best_cost = 1, None
    for ns in range(1, min(max_split, max_cap) + 1):
        if _wgrad_max_wg() and ns > 1 and tiles * ns > _wgrad_max_wg():
            break
        rounds = -(-tiles * ns // _wgrad_cus())
        stages = -(-M // (ns * 64))
        cost = rounds * (stages + _wgrad_epi())
        if best_cost is None or cost < best_cost:
            best, best_cost = ns, cost
    return best


WGRAD3_MIN_PIXELS = 16384      # include/simt_hip.h SIMT_WGRAD3_MIN_PIXELS


def wgrad_tile_co(M, Cd, Ktot):
    """dY channels per output tile of the bf16 kernel (the rule of simt_conv_wgrad_tile_co): 256 when Cd has whole 256-channel tiles
    and the problem has enough pixels."""
    import os
    return 256 if (M >= WGRAD3_MIN_PIXELS and Cd % 256 == 0 and Ktot % 4 == 0 and os.environ.get("SIMT_WGRAD3", "1") != "0") else 128


def wgrad_tiles(M, Cd, Ktot, tile_co=None):
    """Output tiles of one problem on the bf16 kernel (tile_co x 256)."""
    tile_co = tile_co or wgrad_tile_co(M, Cd, Ktot)
    return ((Cd + tile_co - 1) // tile_co) * ((Ktot + 255) // 256)


def wgrad_multi_ok(d):
    return bool(L.load().simt_conv_wgrad_multi_ok(C.byref(d)))


def wgrad_multi_table(descs, device):
    """Device argument table of a grouped launch (simt_conv_wgrad_multi_prepare) -> (uint8 tensor, grid)."""
    lib = L.load()
    n = len(descs)
    arr = (L.WgradDesc * n)(*descs)
    nb = lib.simt_conv_wgrad_multi_bytes() * n
    host = (C.c_uint8 * nb)()
    grid, tco = C.c_int(), C.c_int()
    L.check(lib.simt_conv_wgrad_multi_prepare(arr, n, C.cast(host, C.c_void_p), C.byref(grid), C.byref(tco)))
    t = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(device)
    return t, grid.value, tco.value


def wgrad_group_tile_co(M, shapes):
    """shapes: [(Cd, Ktot)] of a grouped launch over M pixels -> its tile height (256 only if every problem takes it)."""
    return 256 if all(wgrad_tile_co(M, cd, kt) == 256 for cd, kt in shapes) else 128


def wgrad_multi(table, n, grid, nsplit, tile_co):
    L.call("simt_conv_wgrad_multi", _p(table), n, grid, nsplit, tile_co, stream_ptr())


def wgrad_reduce_multi_table(jobs, device):
    """jobs: dicts(slab, dst, nsplit, Cd, Ktot, Cin, co_off, tap_off, Cout, RS[, accumulate]) -> (device table, n, blocks) for
    simt_wgrad_reduce_multi."""
    arr = (L.WgradReduceJob * len(jobs))()
    blocks = 0
    for a, j in zip(arr, jobs):
        assert j["Cin"] % 4 == 0 and j["Ktot"] % 4 == 0
        # the kernel loads / stores float4 (include/simt_hip.h: 16-byte aligned slab and dst); simt_wgrad_reduce has a scalar fallback,
        # the table form does not -- engine.layout_flat_grads keeps every gradient on a 16-byte boundary
        assert j["slab"].data_ptr() % 16 == 0 and j["dst"].data_ptr() % 16 == 0, "simt_wgrad_reduce_multi needs 16-byte aligned slab / dst"
        a.slab, a.dst = _p(j["slab"]), _p(j["dst"])
        a.nsplit, a.Cd, a.Ktot, a.Cin = j["nsplit"], j["Cd"], j["Ktot"], j["Cin"]
        a.co_off, a.tap_off, a.Cout, a.RS = j["co_off"], j["tap_off"], j["Cout"], j["RS"]
        a.accumulate, a.block0 = int(j.get("accumulate", False)), blocks
        blocks += L.load().simt_wgrad_reduce_blocks(j["Cout"], j["RS"], j["Cin"])
    t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return t, len(jobs), blocks


def wgrad_reduce_multi(table, n, blocks):
    L.call("simt_wgrad_reduce_multi", _p(table), n, blocks, stream_ptr())


def wgrad_reduce(slab, dst, *, nsplit, Cd, Ktot, Cin, co_off, tap_off, Cout, RS, accumulate=False):
    L.call("simt_wgrad_reduce", _p(slab), _p(dst), nsplit, Cd, Ktot, Cin, co_off, tap_off, Cout, RS, int(accumulate),
           stream_ptr())


def pack_weight(w, dst, *, Cout, Cin, RS, row_off=0, tap_off=0, ldk, Ck=0, mode=0, cscale=None):
    """w: OIHW fp32 master ([Cout][Cin][RS]); dst: packed operand in its own dtype (zero-initialised by caller)."""
    assert w.dtype == torch.float32 and w.is_contiguous()
    L.call("simt_pack_weight", _p(w), _p(dst), Cout, Cin, RS, row_off, tap_off, ldk, Ck, mode, _p(cscale),
           dt_code(dst.dtype), stream_ptr())


def bn_fold(gamma, beta, rm, rv, eps, scale, shift):
    L.call("simt_bn_fold", _p(gamma), _p(beta), _p(rm), _p(rv), eps, _p(scale), _p(shift), gamma.numel(), stream_ptr())


def bn_finalize(part, nblk, Cn, count, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale, shift):
    L.call("simt_bn_finalize", _p(part), nblk, Cn, count, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
           momentum, eps, _p(mean), _p(rstd), _p(scale), _p(shift), stream_ptr())


def bn_apply(y, scale, shift, z, *, M, Cn, relu=True, res=None, y2=None, scale2=None, shift2=None, bits=None):
    """bits: optional uint8 [M, Cn/8] receiving the ReLU mask (bit c&7 of byte (m*Cn+c)/8 = z>0) for bn_bwd mask_mode 3."""
    if bits is not None:
        L.call("simt_bn_apply_bits", _p(y), _p(scale), _p(shift), _p(res), _p(y2), _p(scale2), _p(shift2), _p(z), _p(bits), M, Cn,
               int(relu), dt_code(y.dtype), stream_ptr())
        return
    L.call("simt_bn_apply", _p(y), _p(scale), _p(shift), _p(res), _p(y2), _p(scale2), _p(shift2), _p(z), M, Cn,
           int(relu), dt_code(y.dtype), stream_ptr())


def bn_bwd_nblk(M, Cn):
    return L.load().simt_bn_bwd_nblk(M, Cn)


def make_bn_bwd_desc(*, dz, y, mean, rstd, scale, shift, part, coef, dy, M, Cn, mask_mode, z=None, y2=None, mean2=None,
                     rstd2=None, scale2=None, dy2=None, gout=None, dgamma=None, dbeta=None, dgamma2=None, dbeta2=None,
                     reduce_done_nblk=0):
    d = L.BnBwdDesc()
    d.dz, d.z, d.y = _p(dz), _p(z), _p(y)
    d.mean, d.rstd, d.scale, d.shift = _p(mean), _p(rstd), _p(scale), _p(shift)
    d.y2, d.mean2, d.rstd2, d.scale2 = _p(y2), _p(mean2), _p(rstd2), _p(scale2)
    d.part, d.coef, d.dy, d.dy2, d.gout = _p(part), _p(coef), _p(dy), _p(dy2), _p(gout)
    d.M, d.C, d.mask_mode, d.dtype = M, Cn, mask_mode, dt_code(y.dtype)
    d.dgamma, d.dbeta, d.dgamma2, d.dbeta2 = _p(dgamma), _p(dbeta), _p(dgamma2), _p(dbeta2)
    d.reduce_done_nblk = reduce_done_nblk
    return d


def bn_bwd_desc(d):
    L.call("simt_bn_bwd", C.byref(d), stream_ptr())


def im2col_stem(x_nchw, A, *, B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk):
    assert x_nchw.dtype == torch.float32 and x_nchw.is_contiguous()
    L.call("simt_im2col_stem", _p(x_nchw), _p(A), B, Cin, H, W, Ho, Wo, KH, KW, stride, pad, ldk, dt_code(A.dtype),
           stream_ptr())


def bn_relu_maxpool(y, scale, shift, p, idx, *, B, H, W, Cn, Hp, Wp):
    L.call("simt_bn_relu_maxpool", _p(y), _p(scale), _p(shift), _p(p), _p(idx), B, H, W, Cn, Hp, Wp, dt_code(y.dtype),
           stream_ptr())


def maxpool_bwd(dp, idx, da, *, B, H, W, Cn, Hp, Wp):
    L.call("simt_maxpool_bwd", _p(dp), _p(idx), _p(da), B, H, W, Cn, Hp, Wp, dt_code(dp.dtype), stream_ptr())


def scatter_stride(src, dx, *, B, H, W, Cn, Ho, Wo, stride):
    L.call("simt_scatter_stride", _p(src), _p(dx), B, H, W, Cn, Ho, Wo, stride, dt_code(src.dtype), stream_ptr())


def colsum(src, out, *, M, ld, Cn, accumulate=False):
    L.call("simt_colsum", _p(src), _p(out), M, ld, Cn, int(accumulate), dt_code(src.dtype), stream_ptr())


def softmax_rows(src, ldi, out, ldo, M, Cn):
    L.call("simt_softmax_rows", _p(src), ldi, _p(out), ldo, M, Cn, stream_ptr())


def adam_step(p, g, m, v, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, step, skip_if=None):
    """skip_if: device word (TrunkPlan.fbn_err); while it is non-zero the launch leaves p and the moments untouched."""
    L.call("simt_adam_step_guarded", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, step, _p(skip_if), stream_ptr())


def sig_ntm(ntm, class_dist, T_out=None, dT=None, dN_out=None):
    Q, Cn = ntm.shape
    L.call("simt_sig_ntm", _p(ntm), _p(class_dist), _p(dT), _p(T_out), _p(dN_out), Q, Cn, stream_ptr())


def sig_w(weight, W_out=None, dW=None, dweight_out=None):
    Q = weight.shape[0]
    L.call("simt_sig_w", _p(weight), _p(dW), _p(W_out), _p(dweight_out), Q, stream_ptr())
