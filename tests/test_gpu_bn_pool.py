"""GPU parity of the HBM-bound kernels around the convs (C ABI: simt_bn_finalize/apply/bwd, simt_bn_relu_maxpool,
simt_maxpool_bwd, simt_im2col_stem, simt_scatter_stride, simt_colsum, simt_sgd_multi) against torch CPU fp32
(oracle/ops_ref-style checkers).  fp32: 1e-5 relative to max|ref| (max-pool indices / masks exact); bf16: 2e-2."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from simt_amd import _lib as L
from simt_amd import ops

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def nhwc(x, dev, dtype):
    return x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)


def nchw(x):
    return x.float().cpu().permute(0, 3, 1, 2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cn,B,H,W", [(64, 2, 13, 11), (256, 1, 9, 9), (2048, 1, 9, 9), (512, 3, 17, 5), (1024, 2, 9, 11), (128, 2, 25, 33)])
@pytest.mark.parametrize("mode", ["plain", "residual", "downsample"])
def test_bn_train_forward_backward(dev, dtype, Cn, B, H, W, mode):
    g = torch.Generator().manual_seed(Cn + H)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    y = (torch.randn(B, Cn, H, W, generator=g) * 2 + 0.5).to(dtype).float()
    gamma = torch.rand(Cn, generator=g) + 0.5
    beta = torch.randn(Cn, generator=g) * 0.1
    rm, rv = torch.randn(Cn, generator=g) * 0.1, torch.rand(Cn, generator=g) + 0.5
    M = B * H * W
    yr = y.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    out = F.batch_norm(yr, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    extra = None
    if mode == "residual":
        extra = torch.randn(B, Cn, H, W, generator=g).to(dtype).float().requires_grad_(True)
        out = out + extra
    elif mode == "downsample":
        extra = (torch.randn(B, Cn, H, W, generator=g) * 3 - 1).to(dtype).float().requires_grad_(True)
        g2, b2 = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1
        out = out + F.batch_norm(extra, None, None, g2, b2, training=True, eps=1e-5)
    z = F.relu(out)
    dz = torch.randn(B, Cn, H, W, generator=g).to(dtype).float()
    z.backward(dz)

    # device: statistics from per-tile partials like the conv epilogue writes them (128-row tiles)
    def stats_part(t):
        flat = t.permute(0, 2, 3, 1).reshape(M, Cn)
        nblk = (M + 127) // 128
        part = torch.zeros(nblk, 2, Cn)
        for i in range(nblk):
            blk = flat[i * 128:(i + 1) * 128]
            part[i, 0], part[i, 1] = blk.sum(0), (blk * blk).sum(0)
        return part.to(dev), nblk

    def finalize(t, gam, bet, rmean, rvar):
        part, nblk = stats_part(t)
        outs = [torch.zeros(Cn, device=dev) for _ in range(4)]
        ops.bn_finalize(part, nblk, Cn, M, gam.to(dev), bet.to(dev), rmean, rvar, 0.1, 1e-5, *outs)
        return outs

    rm_d, rv_d = rm.to(dev), rv.to(dev)
    mean, rstd, scale, shift = finalize(y, gamma, beta, rm_d, rv_d)
    y_d = nhwc(y, dev, dtype)
    z_d = torch.empty_like(y_d)
    kw = {}
    if mode == "residual":
        kw = dict(res=nhwc(extra.detach(), dev, dtype))
    elif mode == "downsample":
        mean2, rstd2, scale2, shift2 = finalize(extra.detach(), g2, b2, None, None)
        y2_d = nhwc(extra.detach(), dev, dtype)
        kw = dict(y2=y2_d, scale2=scale2, shift2=shift2)
    bits = torch.zeros(M, Cn // 8, device=dev, dtype=torch.uint8)
    ops.bn_apply(y_d, scale, shift, z_d, M=M, Cn=Cn, relu=True, bits=bits, **kw)
    torch.cuda.synchronize()
    assert rel(nchw(z_d), z.detach()) < tol
    # the bit mask is exactly (z > 0) of the stored activation: bit (c & 7) of byte (m*Cn + c) / 8
    zb = (z_d.reshape(M, Cn // 8, 8).float() > 0).to(torch.int32)
    want = (zb << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want)
    assert rel(rm_d.cpu(), rm_ref) < 1e-5 and rel(rv_d.cpu(), rv_ref) < 1e-5

    dz_d = nhwc(dz, dev, dtype)
    nblk = ops.bn_bwd_nblk(M, Cn)
    part = torch.zeros(nblk * 3 * Cn, device=dev)
    coef = torch.zeros(3 * Cn, device=dev)
    dy_d = torch.empty_like(y_d)
    gout = torch.empty_like(y_d)
    if mode == "downsample":
        dy2_d = torch.empty_like(y_d)
        d = ops.make_bn_bwd_desc(dz=dz_d, z=z_d, y=y_d, mean=mean, rstd=rstd, scale=scale, shift=shift, part=part, coef=coef,
                                 dy=dy_d, M=M, Cn=Cn, mask_mode=1, y2=y2_d, mean2=mean2, rstd2=rstd2, scale2=scale2,
                                 dy2=dy2_d, gout=gout)
    elif mode == "residual":
        d = ops.make_bn_bwd_desc(dz=dz_d, z=z_d, y=y_d, mean=mean, rstd=rstd, scale=scale, shift=shift, part=part, coef=coef,
                                 dy=dy_d, M=M, Cn=Cn, mask_mode=1, gout=gout)
    else:
        d = ops.make_bn_bwd_desc(dz=dz_d, y=y_d, mean=mean, rstd=rstd, scale=scale, shift=shift, part=part, coef=coef,
                                 dy=dy_d, M=M, Cn=Cn, mask_mode=2)
    ops.bn_bwd_desc(d)
    torch.cuda.synchronize()
    if mode in ("downsample", "residual"):
        # mask_mode 3 (bit mask instead of z) gives bit-identical results to mask_mode 1
        first = [t.clone() for t in (dy_d, gout)] + ([dy2_d.clone()] if mode == "downsample" else [])
        d.z, d.mask_mode = bits.data_ptr(), 3
        ops.bn_bwd_desc(d)
        torch.cuda.synchronize()
        again = [dy_d, gout] + ([dy2_d] if mode == "downsample" else [])
        assert all(torch.equal(a, b) for a, b in zip(first, again))
    # bf16: the mask is taken from the bf16-rounded z, a handful of borderline elements may flip
    assert rel(nchw(dy_d), yr.grad) < (tol if dtype == torch.float32 else 0.1)
    if mode == "residual":
        assert rel(nchw(gout), extra.grad) < tol
    if mode == "downsample":
        assert rel(nchw(dy2_d), extra.grad) < (tol if dtype == torch.float32 else 0.1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W", [(2, 33, 33), (1, 32, 48), (1, 7, 9)])
def test_bn_relu_maxpool_ceil_and_backward(dev, dtype, B, H, W):
    Cn = 64
    g = torch.Generator().manual_seed(H * W)
    y = torch.randn(B, Cn, H, W, generator=g).to(dtype).float()
    sc, sh = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.2
    a = F.relu(y * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).requires_grad_(True)
    p = F.max_pool2d(a, 3, 2, 1, ceil_mode=True)
    Hp, Wp = p.shape[2:]
    dp = torch.randn(p.shape, generator=g).to(dtype).float()
    p.backward(dp)
    y_d = nhwc(y, dev, dtype)
    p_d = torch.empty(B, Hp, Wp, Cn, device=dev, dtype=dtype)
    idx = torch.empty(B, Hp, Wp, Cn, device=dev, dtype=torch.uint8)
    ops.bn_relu_maxpool(y_d, sc.to(dev), sh.to(dev), p_d, idx, B=B, H=H, W=W, Cn=Cn, Hp=Hp, Wp=Wp)
    da_d = torch.empty(B, H, W, Cn, device=dev, dtype=dtype)
    ops.maxpool_bwd(nhwc(dp, dev, dtype), idx, da_d, B=B, H=H, W=W, Cn=Cn, Hp=Hp, Wp=Wp)
    torch.cuda.synchronize()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert rel(nchw(p_d), p.detach()) < tol
    if dtype == torch.float32:
        assert rel(nchw(da_d), a.grad) < 1e-6


def test_im2col_stem_scatter_colsum(dev):
    g = torch.Generator().manual_seed(1)
    B, H, W = 2, 37, 41
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    ref = F.conv2d(x, w, stride=2, padding=3)
    Ho, Wo = ref.shape[2:]
    A = torch.empty(B * Ho * Wo, 192, device=dev)
    ops.im2col_stem(x.to(dev), A, B=B, Cin=3, H=H, W=W, Ho=Ho, Wo=Wo, KH=7, KW=7, stride=2, pad=3, ldk=192)
    torch.cuda.synchronize()
    got = (A.cpu()[:, :147] @ w.reshape(64, 147).t()).reshape(B, Ho, Wo, 64).permute(0, 3, 1, 2)
    assert rel(got, ref) < 1e-5
    assert torch.all(A[:, 147:] == 0)
    # scatter_stride: adjoint of x[:, ::2, ::2]
    src = torch.randn(B, 5, 6, 64, generator=g)
    dx = torch.empty(B, 9, 11, 64, device=dev)
    ops.scatter_stride(src.to(dev), dx, B=B, H=9, W=11, Cn=64, Ho=5, Wo=6, stride=2)
    exp = torch.zeros(B, 9, 11, 64)
    exp[:, ::2, ::2] = src
    assert torch.equal(dx.cpu(), exp)
    # colsum
    m = torch.randn(1000, 32, generator=g)
    out = torch.zeros(64, device=dev)
    ops.colsum(m.to(dev), out, M=1000, ld=32, Cn=22)
    assert rel(out.cpu()[:22], m[:, :22].sum(0)) < 1e-6


def test_sgd_multi_duplicate_listing_semantics(dev):
    """torch.optim.SGD(foreach=False) with a tensor listed `mult` times == simt_sgd_multi."""
    g = torch.Generator().manual_seed(9)
    shapes, mults, groups = [(70000,), (33, 7), (5,)], [3, 4, 1], [0, 0, 1]
    ps = [torch.randn(s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(p.clone()) for p in ps]
    listing0 = [ref[0]] * 3 + [ref[1]] * 4
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt = torch.optim.SGD([{"params": listing0, "lr": 0.1}, {"params": [ref[2]], "lr": 1.0}], lr=0.1, momentum=0.9,
                              weight_decay=5e-4, foreach=False)
    dp = [p.clone().to(dev) for p in ps]
    bufs = [torch.zeros_like(p) for p in dp]
    gd = [torch.zeros_like(p) for p in dp]
    recs = [(p.data_ptr(), gg.data_ptr(), b.data_ptr(), p.numel(), m, gr) for p, gg, b, m, gr in zip(dp, gd, bufs, mults, groups)]
    seg_dt = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("mult", "<i4"), ("group", "<i4")])
    segs = torch.from_numpy(np.array(recs, dtype=seg_dt).view(np.uint8).copy()).to(dev)
    chunk = 65536
    chunks = [(si, ci) for si, r in enumerate(recs) for ci in range((r[3] + chunk - 1) // chunk)]
    chunks_d = torch.tensor(chunks, dtype=torch.int32).to(dev)
    d = L.SgdDesc()
    d.segs, d.chunks, d.nchunks, d.chunk = segs.data_ptr(), chunks_d.data_ptr(), len(chunks), chunk
    d.momentum, d.dampening = 0.9, 0.0
    for step in range(3):
        grads = [torch.randn(s, generator=g) for s in shapes]
        for r, gr in zip(ref, grads):
            r.grad = gr.clone()
        opt.step()
        for gdev, gr in zip(gd, grads):
            gdev.copy_(gr)
        d.lr[0], d.lr[1], d.wd[0], d.wd[1] = 0.1, 1.0, 5e-4, 5e-4
        d.first_step = 1 if step == 0 else 0
        L.call("simt_sgd_multi", C.byref(d), ops.stream_ptr())
        torch.cuda.synchronize()
        for r, p in zip(ref, dp):
            assert rel(p.cpu(), r.detach()) < 2e-6, f"step {step}"


@pytest.mark.parametrize("Cn,M", [(64, 5000), (256, 37636), (1024, 1001), (2048, 777), (512, 4 * 97 * 97 + 3)])
@pytest.mark.parametrize("mask_mode", [0, 1, 2, 3])
def test_bn_bwd_apply_four_channel_kernel_is_bitwise_the_eight_channel_one(dev, Cn, M, mask_mode):
    """Round 6: simt_bn_bwd's apply pass for ONE BatchNorm in bf16 runs bn_bwd_apply4_kernel (four channels per thread, <= 64 VGPRs: it shares
    a CU with the weight-gradient launches of the other stream); with a second BatchNorm (y2: the downsample partner) the call keeps the
    eight-channel kernel, which writes the FIRST BatchNorm's dy from the same expression.  Same inputs through both: dy, the masked gradient
    (gout) bit for bit, for every mask mode, ragged row counts and the widest channel count (torch.nn.BatchNorm2d backward,
    model/deeplab_multi.py:64-76 train mode)."""
    g = torch.Generator().manual_seed(Cn + M + mask_mode)
    BF = torch.bfloat16
    dz = torch.randn(M, Cn, generator=g).to(dev, BF)
    y = (torch.randn(M, Cn, generator=g) * 2 + 0.3).to(dev, BF)
    mean = torch.randn(Cn, generator=g).mul(0.2).to(dev)
    rstd = (torch.rand(Cn, generator=g) + 0.5).to(dev)
    gamma = (torch.rand(Cn, generator=g) + 0.5).to(dev)
    scale = gamma * rstd
    shift = torch.randn(Cn, generator=g).mul(0.3).to(dev) - mean * scale
    z = None
    if mask_mode == 1:
        z = torch.randn(M, Cn, generator=g).to(dev, BF)
    elif mask_mode == 3:
        z = torch.randint(0, 256, (M * Cn // 8,), generator=g, dtype=torch.int32).to(torch.uint8).to(dev)
    nblk = ops.bn_bwd_nblk(M, Cn)
    out = []
    for two in (False, True):
        part = torch.zeros(nblk, 3, Cn, device=dev)
        coef = torch.zeros(3, Cn, device=dev)
        dy = torch.full((M, Cn), float("nan"), device=dev, dtype=BF)
        dy2 = torch.full((M, Cn), float("nan"), device=dev, dtype=BF)
        gout = torch.full((M, Cn), float("nan"), device=dev, dtype=BF)
        kw = dict(y2=y, mean2=mean, rstd2=rstd, scale2=scale, dy2=dy2) if two else {}
        d = ops.make_bn_bwd_desc(dz=dz, y=y, mean=mean, rstd=rstd, scale=scale, shift=shift, part=part, coef=coef, dy=dy, M=M, Cn=Cn,
                                 mask_mode=mask_mode, z=z, gout=gout, **kw)
        ops.bn_bwd_desc(d)
        torch.cuda.synchronize()
        out.append((dy, gout, coef[:2].clone(), dy2))
    (dy4, g4, c4, _), (dy8, g8, c8, dy8b) = out
    assert torch.equal(c4, c8)                                   # the reduce pass is the same kernel either way
    assert torch.isfinite(dy4.float()).all() and torch.equal(dy4, dy8) and torch.equal(g4, g8)
    assert torch.equal(dy8b, dy8)                                # (the partner was given the same operands)
