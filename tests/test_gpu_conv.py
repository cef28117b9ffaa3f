"""GPU parity of the implicit-GEMM conv kernels (fprop / dgrad / wgrad) through the C ABI.

Checker: torch CPU fp32 conv2d + autograd (oracle/ops_ref.py), same seeded inputs.
Tolerances: fp32 mode 2e-5 relative to max|ref| (accumulation-order noise only);
bf16 mode compares against the fp32 reference evaluated on bf16-rounded inputs, 1e-2 relative to max|ref|.
"""
import pytest
import torch

from oracle import ops_ref
from simt_amd import ops

pytestmark = pytest.mark.gpu

CASES = [
    # B, H, W, Cin, Cout, k, dil, stride
    (2, 13, 13, 64, 64, 1, 1, 1),
    (2, 13, 17, 128, 256, 1, 1, 1),
    (1, 23, 21, 64, 64, 3, 1, 1),
    (2, 17, 19, 128, 128, 3, 2, 1),
    (1, 19, 19, 64, 192, 3, 4, 1),
    (2, 21, 21, 256, 128, 1, 1, 2),
    (3, 9, 9, 64, 24, 3, 2, 1),
    # maps of fewer than 64 pixels: one 64-pixel wgrad stage crosses several images (conv_wgrad2 incremental (oy, ox))
    (24, 4, 4, 64, 64, 3, 1, 1),
    (11, 7, 7, 128, 64, 3, 2, 1),
    (9, 4, 8, 64, 128, 1, 1, 1),
    # >= 16 384 output pixels and Cout % 256 == 0: the 256 x 256 weight-gradient tile (conv_wgrad3_body), strided (per-stage pixel
    # decomposition) and with taps
    (4, 130, 130, 64, 256, 1, 1, 2),
    (4, 65, 65, 64, 256, 3, 2, 1),
]


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1e-2


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_conv_fprop_dgrad_wgrad(dev, dtype, case):
    B, H, W, Cin, Cout, k, dil, stride = case
    g = torch.Generator().manual_seed(1234 + Cin + Cout + k + dil)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    pad = dil * (k // 2)
    xq = x.to(dtype).float()
    wq = w.to(dtype).float()
    y_ref, dx_ref, dw_ref, dy = ops_ref.conv2d_fwd_bwd(xq, wq, stride=stride, pad=pad, dil=dil, seed=7, grad_dtype=dtype)
    Ho, Wo = y_ref.shape[2], y_ref.shape[3]
    taps = ops.conv_taps(k, k, dil, pad)

    x_d = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    tile = ops.pick_tile_n(Cout, dtype)
    Npad = ops.round_up(Cout, tile)
    wp = torch.zeros(Npad, len(taps) * Cin, device=dev, dtype=dtype)
    ops.pack_weight(w.to(dev).contiguous(), wp, Cout=Cout, Cin=Cin, RS=k * k, ldk=len(taps) * Cin, mode=0)
    ldy = ops.round_up(Cout, 8)
    y_d = torch.full((B, Ho, Wo, ldy), float("nan"), device=dev, dtype=dtype)
    M = B * Ho * Wo
    mt = (M + 127) // 128
    stats = torch.zeros(mt, 2, Cout, device=dev)
    d = ops.make_conv_desc(x_d, wp, y_d, B=B, H=H, W=W, Cin=Cin, Ho=Ho, Wo=Wo, Cout=Cout, taps=taps, stride=stride,
                           stats=stats)
    ops.conv_fprop_desc(d)
    torch.cuda.synchronize()
    y_got = y_d[..., :Cout].float().cpu().permute(0, 3, 1, 2)
    assert _rel(y_got, y_ref) < _tol(dtype), f"fprop rel err {_rel(y_got, y_ref)}"
    # BN statistics partials (computed from the fp32 accumulators)
    s = stats.sum(0).cpu()
    s1_ref = y_ref.sum(dim=(0, 2, 3))
    s2_ref = (y_ref * y_ref).sum(dim=(0, 2, 3))
    assert _rel(s[0], s1_ref) < 5 * _tol(dtype) + 1e-4
    assert _rel(s[1], s2_ref) < 5 * _tol(dtype)

    # ---- wgrad
    dy_d = dy.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    Cd = ops.round_up(Cout, 8)
    if Cd != Cout:
        pad_t = torch.zeros(B, Ho, Wo, Cd, device=dev, dtype=dtype)
        pad_t[..., :Cout] = dy_d
        dy_d = pad_t
    Ktot = len(taps) * Cin
    for nsplit in (1, 3):
        slab = torch.full((nsplit, Cd, Ktot), float("nan"), device=dev)
        wd = ops.make_wgrad_desc(dy_d, x_d, slab, B=B, H=H, W=W, Cin=Cin, Ho=Ho, Wo=Wo, Cd=Cd, taps=taps, stride=stride,
                                 nsplit=nsplit)
        if dtype == torch.bfloat16:          # which tile the library takes for this problem is part of its contract (include/simt_hip.h)
            from simt_amd import _lib
            import ctypes
            assert _lib.load().simt_conv_wgrad_tile_co(ctypes.byref(wd)) == ops.wgrad_tile_co(B * Ho * Wo, Cd, Ktot)
        ops.conv_wgrad_desc(wd)
        dw_d = torch.full((Cout, Cin, k, k), float("nan"), device=dev)
        ops.wgrad_reduce(slab, dw_d, nsplit=nsplit, Cd=Cd, Ktot=Ktot, Cin=Cin, co_off=0, tap_off=0, Cout=Cout, RS=k * k)
        torch.cuda.synchronize()
        assert _rel(dw_d.cpu(), dw_ref) < _tol(dtype), f"wgrad rel err {_rel(dw_d.cpu(), dw_ref)} nsplit={nsplit}"

    # ---- dgrad (stride 1 only; the stride-2 1x1 goes through scatter_stride)
    esz = 2 if dtype == torch.bfloat16 else 4
    Ck = ops.round_up(Cout, 128 // esz)
    tile_b = ops.pick_tile_n(Cin, dtype)
    wb = torch.zeros(ops.round_up(Cin, tile_b), len(taps) * Ck, device=dev, dtype=dtype)
    ops.pack_weight(w.to(dev).contiguous(), wb, Cout=Cout, Cin=Cin, RS=k * k, ldk=len(taps) * Ck, Ck=Ck, mode=1)
    dyk = torch.zeros(B, Ho, Wo, Ck, device=dev, dtype=dtype)
    dyk[..., :Cout] = dy.permute(0, 2, 3, 1).to(dev, dtype)
    ntaps = [(-a, -b) for (a, b) in taps]
    if stride == 1:
        dx_d = torch.full((B, H, W, Cin), float("nan"), device=dev, dtype=dtype)
        dd = ops.make_conv_desc(dyk, wb, dx_d, B=B, H=Ho, W=Wo, Cin=Ck, Ho=H, Wo=W, Cout=Cin, taps=ntaps, stride=1)
        ops.conv_fprop_desc(dd)
    else:
        dxs = torch.full((B, Ho, Wo, Cin), float("nan"), device=dev, dtype=dtype)
        dd = ops.make_conv_desc(dyk, wb, dxs, B=B, H=Ho, W=Wo, Cin=Ck, Ho=Ho, Wo=Wo, Cout=Cin, taps=ntaps, stride=1)
        ops.conv_fprop_desc(dd)
        dx_d = torch.full((B, H, W, Cin), float("nan"), device=dev, dtype=dtype)
        ops.scatter_stride(dxs, dx_d, B=B, H=H, W=W, Cn=Cin, Ho=Ho, Wo=Wo, stride=stride)
    torch.cuda.synchronize()
    dx_got = dx_d.float().cpu().permute(0, 3, 1, 2)
    assert _rel(dx_got, dx_ref) < _tol(dtype), f"dgrad rel err {_rel(dx_got, dx_ref)}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_epilogue_bias_res_relu_and_aspp_taps(dev, dtype):
    """Classifier_Module: two dilated 3x3 branches summed in ONE launch (18 taps), bias, Cout=22 -> fp32 logits."""
    B, H, W, Cin, Cout = 2, 15, 15, 128, 22
    g = torch.Generator().manual_seed(99)
    x = torch.randn(B, Cin, H, W, generator=g)
    w6 = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    w12 = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b6 = torch.randn(Cout, generator=g)
    b12 = torch.randn(Cout, generator=g)
    xq = x.to(dtype).float()
    y_ref = ops_ref.conv2d(xq, w6.to(dtype).float(), b6, pad=6, dil=6) + ops_ref.conv2d(xq, w12.to(dtype).float(), b12, pad=12, dil=12)
    taps = ops.conv_taps(3, 3, 6, 6) + ops.conv_taps(3, 3, 12, 12)
    x_d = x.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    wp = torch.zeros(32, 18 * Cin, device=dev, dtype=dtype)
    ops.pack_weight(w6.to(dev), wp, Cout=Cout, Cin=Cin, RS=9, tap_off=0, ldk=18 * Cin)
    ops.pack_weight(w12.to(dev), wp, Cout=Cout, Cin=Cin, RS=9, tap_off=9, ldk=18 * Cin)
    bias = (b6 + b12).to(dev)
    y_d = torch.full((B, H, W, 32), float("nan"), device=dev, dtype=torch.float32)
    d = ops.make_conv_desc(x_d, wp, y_d, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, bias=bias, tile_n=32,
                           Nstore=32)
    ops.conv_fprop_desc(d)
    torch.cuda.synchronize()
    got = y_d[..., :Cout].cpu().permute(0, 3, 1, 2)
    assert _rel(got, y_ref) < _tol(dtype)
    assert torch.all(y_d[..., Cout:] == 0)  # padded logits columns are exact zeros

    # residual + relu epilogue
    w1 = torch.randn(64, Cin, 1, 1, generator=g) * 0.1
    r = torch.randn(B, 64, H, W, generator=g)
    bb = torch.randn(64, generator=g)
    ref = torch.relu(ops_ref.conv2d(xq, w1.to(dtype).float(), bb) + r.to(dtype).float())
    wp1 = torch.zeros(64, Cin, device=dev, dtype=dtype)
    ops.pack_weight(w1.to(dev), wp1, Cout=64, Cin=Cin, RS=1, ldk=Cin)
    r_d = r.permute(0, 2, 3, 1).contiguous().to(dev, dtype)
    z_d = torch.empty(B, H, W, 64, device=dev, dtype=dtype)
    d = ops.make_conv_desc(x_d, wp1, z_d, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=64, taps=[(0, 0)], bias=bb.to(dev),
                           res=r_d, relu=True)
    ops.conv_fprop_desc(d)
    torch.cuda.synchronize()
    assert _rel(z_d.float().cpu().permute(0, 3, 1, 2), ref) < _tol(dtype)

    # residual counted only where a bit mask is set (identity-shortcut gradient dz * (z > 0), simt_conv_desc.res_bits)
    keep = torch.rand(B, 64, H, W, generator=g) > 0.4
    ref = ops_ref.conv2d(xq, w1.to(dtype).float(), bb) + r.to(dtype).float() * keep
    kb = keep.permute(0, 2, 3, 1).reshape(-1, 8, 8).to(torch.int32)                      # [M, 64/8, 8]
    bits = (kb << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8).to(dev)
    d = ops.make_conv_desc(x_d, wp1, z_d, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=64, taps=[(0, 0)], bias=bb.to(dev),
                           res=r_d, res_bits=bits)
    ops.conv_fprop_desc(d)
    torch.cuda.synchronize()
    assert _rel(z_d.float().cpu().permute(0, 3, 1, 2), ref) < _tol(dtype)


def test_tap_expanded_head_kernels(dev):
    """bf16 throughput form of the ASPP classifier (csrc/head_expand.hip): P = x @ Wexp^T (fp32 out of the bf16 GEMM),
    tap gather-sum, tap scatter, expanded weight-gradient reduce -- against the dilated convs of torch CPU fp32 evaluated
    on the bf16-rounded inputs.  1e-2 of max|ref| (bf16 operands, fp32 accumulation)."""
    import ctypes as C
    from simt_amd import _lib as L
    B, H, W, Cin, Q = 2, 15, 17, 128, 22
    QP, dils = 24, (6, 12)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    ws = [(torch.randn(19, Cin, 3, 3, generator=g) * 0.05), (torch.randn(19, Cin, 3, 3, generator=g) * 0.05),
          (torch.randn(3, Cin, 3, 3, generator=g) * 0.05), (torch.randn(3, Cin, 3, 3, generator=g) * 0.05)]
    wq = [w.bfloat16().float() for w in ws]
    bias = torch.randn(Q, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = [w.clone().requires_grad_(True) for w in wq]
    y0 = ops_ref.conv2d(xr, wr[0], None, pad=6, dil=6) + ops_ref.conv2d(xr, wr[1], None, pad=12, dil=12)
    y1 = ops_ref.conv2d(xr, wr[2], None, pad=6, dil=6) + ops_ref.conv2d(xr, wr[3], None, pad=12, dil=12)
    y_ref = torch.cat([y0, y1], 1) + bias.view(1, -1, 1, 1)
    dy = torch.randn(y_ref.shape, generator=g).bfloat16().float()
    y_ref.backward(dy)
    taps = ops.conv_taps(3, 3, 6, 6) + ops.conv_taps(3, 3, 12, 12)
    nt, M = len(taps), B * H * W
    nexp, npe = nt * QP, 512
    x_d = x.permute(0, 2, 3, 1).contiguous().to(dev, torch.bfloat16)
    wexp = torch.zeros(npe, Cin, device=dev, dtype=torch.bfloat16)
    for (w, row, i) in ((ws[0], 0, 0), (ws[1], 0, 1), (ws[2], 19, 0), (ws[3], 19, 1)):
        ops.pack_weight(w.to(dev).contiguous(), wexp, Cout=w.shape[0], Cin=Cin, RS=9, row_off=row, tap_off=9 * i, ldk=Cin, Ck=QP,
                        mode=2)
    P = torch.full((M, nexp), float("nan"), device=dev)
    d = ops.make_conv_desc(x_d, wexp, P, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=nexp, taps=[(0, 0)], Npad=npe, tile_n=256,
                           ldy=nexp, Nstore=nexp)
    ops.conv_fprop_desc(d)
    logits = torch.zeros(M, 32, device=dev)
    td = L.TapDesc()
    bias_d = bias.to(dev)
    td.src, td.bias, td.dst = P.data_ptr(), bias_d.data_ptr(), logits.data_ptr()
    td.B, td.H, td.W, td.Q, td.QP, td.lds, td.ldd, td.ntaps = B, H, W, Q, QP, nexp, 32, nt
    ops._fill_taps(td.dy, td.dx, taps)
    L.call("simt_tap_gather_sum", C.byref(td), ops.stream_ptr())
    torch.cuda.synchronize()
    got = logits[:, :Q].cpu().reshape(B, H, W, Q).permute(0, 3, 1, 2)
    assert _rel(got, y_ref.detach()) < 1e-2
    assert torch.all(logits[:, Q:] == 0)
    # ---- backward: scatter, expanded wgrad + reduce, dgrad as a plain GEMM
    dl = torch.zeros(M, 64, device=dev, dtype=torch.bfloat16)
    dl[:, :Q] = dy.permute(0, 2, 3, 1).reshape(M, Q).to(dev, torch.bfloat16)
    kexp = 448
    G = torch.zeros(M, kexp, device=dev, dtype=torch.bfloat16)
    ts = L.TapDesc()
    ts.src, ts.bias, ts.dst = dl.data_ptr(), None, G.data_ptr()
    ts.B, ts.H, ts.W, ts.Q, ts.QP, ts.lds, ts.ldd, ts.ntaps = B, H, W, Q, QP, 64, kexp, nt
    ops._fill_taps(ts.dy, ts.dx, taps)
    L.call("simt_tap_scatter", C.byref(ts), ops.stream_ptr())
    for nsplit in (1, 2):
        slab = torch.full((nsplit, nexp, Cin), float("nan"), device=dev)
        wd = ops.make_wgrad_desc(G, x_d, slab, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cd=nexp, taps=[(0, 0)], nsplit=nsplit, ldd=kexp)
        ops.conv_wgrad_desc(wd)
        for (wi, row, i) in ((0, 0, 0), (1, 0, 1), (2, 19, 0), (3, 19, 1)):
            dw = torch.full(tuple(ws[wi].shape), float("nan"), device=dev)
            L.call("simt_wgrad_reduce_exp", slab.data_ptr(), dw.data_ptr(), nsplit, nexp, Cin, QP, row, 9 * i, ws[wi].shape[0], 9,
                   ops.stream_ptr())
            torch.cuda.synchronize()
            assert _rel(dw.cpu(), wr[wi].grad) < 1e-2, (wi, nsplit)
    wt = torch.zeros(128, kexp, device=dev, dtype=torch.bfloat16)
    for (w, row, i) in ((ws[0], 0, 0), (ws[1], 0, 1), (ws[2], 19, 0), (ws[3], 19, 1)):
        ops.pack_weight(w.to(dev).contiguous(), wt, Cout=w.shape[0], Cin=Cin, RS=9, row_off=row, tap_off=9 * i, ldk=kexp, Ck=QP,
                        mode=1)
    dx_d = torch.empty(B, H, W, Cin, device=dev, dtype=torch.bfloat16)
    dd = ops.make_conv_desc(G, wt, dx_d, B=B, H=H, W=W, Cin=kexp, Ho=H, Wo=W, Cout=Cin, taps=[(0, 0)])
    ops.conv_fprop_desc(dd)
    torch.cuda.synchronize()
    assert _rel(dx_d.float().cpu().permute(0, 3, 1, 2), xr.grad) < 1e-2


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("shape", [(2, 19, 23, 256, 1024, "res_bits"), (2, 19, 23, 1024, 256, "plain"), (1, 15, 17, 128, 128, "plain3x3")])
def test_conv_epilogue_fused_bn_backward_reduce(dev, mode, shape):
    """simt_conv_desc.bnr_*: the dgrad GEMM that produces dz also accumulates S1 = sum dz*mask and S2 = sum dz*mask*xhat of the
    BatchNorm whose output gradient dz is (what bn_bwd_reduce_kernel re-reads dz for).  Checked against torch on the STORED bf16
    dz: 2e-3 of max|S|, for both mask flavours and for the plain / residual-under-bit-mask epilogues."""
    import ctypes as C
    from simt_amd import _lib as L
    B, H, W, Cin, Cout, kind = shape
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(Cin + Cout + mode)
    k = 3 if kind == "plain3x3" else 1
    taps = ops.conv_taps(3, 3, 2, 2) if k == 3 else [(0, 0)]
    M = B * H * W
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, dtype)
    tile = ops.pick_tile_n(Cout, dtype)
    npad = ops.round_up(Cout, tile)
    wp = (torch.randn(npad, len(taps) * Cin, generator=g) * (1.0 / (len(taps) * Cin)) ** 0.5).to(dev, dtype)
    dz = torch.empty(M, Cout, device=dev, dtype=dtype)
    y = torch.randn(M, Cout, generator=g).to(dev, dtype)
    mean, shift = torch.randn(Cout, generator=g).to(dev) * 0.2, torch.randn(Cout, generator=g).to(dev) * 0.3
    rstd, scale = (torch.rand(Cout, generator=g) + 0.5).to(dev), (torch.rand(Cout, generator=g) + 0.5).to(dev)
    bits = torch.randint(0, 256, (M, Cout // 8), generator=g, dtype=torch.uint8).to(dev)
    part = torch.full((M // 128 + 2, 3, Cout), float("nan"), device=dev)
    bnr = {"y": y, "mean": mean, "rstd": rstd, "scale": scale, "shift": shift, "bits": bits, "mode": mode, "part": part}
    kw = {}
    if kind == "res_bits":
        kw = dict(res=torch.randn(M, Cout, generator=g).to(dev, dtype),
                  res_bits=torch.randint(0, 256, (M, Cout // 8), generator=g, dtype=torch.uint8).to(dev))
    d = ops.make_conv_desc(x, wp, dz, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=tile, bnr=bnr, **kw)
    nblk = L.load().simt_conv_mtiles(C.byref(d))
    assert 0 < nblk <= part.shape[0]
    ops.conv_fprop_desc(d)
    d0 = ops.make_conv_desc(x, wp, torch.empty_like(dz), B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=tile, **kw)
    ref_dz = torch.empty_like(dz)
    d0.y = ref_dz.data_ptr()
    ops.conv_fprop_desc(d0)
    torch.cuda.synchronize()
    assert torch.equal(dz, ref_dz)                                 # the fusion does not change what is stored
    gz, yy = dz.double(), y.double()
    if mode == 2:
        msk = (y.float() * scale + shift) > 0
    else:
        msk = ((bits.unsqueeze(-1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).reshape(M, Cout).bool()
    gm = gz * msk
    s1, s2 = gm.sum(0), (gm * ((yy - mean.double()) * rstd.double())).sum(0)
    got = part[:nblk].double().sum(0)
    assert torch.isfinite(got).all()
    assert _rel(got[0].cpu(), s1.cpu()) < 2e-3 and _rel(got[1].cpu(), s2.cpu()) < 2e-3 and got[2].abs().max().item() == 0.0


@pytest.mark.parametrize("mode", [0, 1])
def test_pack_weight_fragment_order(dev, mode):
    """simt_pack_weight with SIMT_PACK_FRAG (single launch and the batched kernel's shared index math): the fragment-ordered operand is
    exactly the K-contiguous one permuted by ops.frag_order (the layout include/simt_hip.h documents for simt_conv_desc.w_frag)."""
    g = torch.Generator().manual_seed(3)
    Cout, Cin, k = 200, 192, 3
    w = torch.randn(Cout, Cin, k, k, generator=g).to(dev)
    if mode == 0:
        npad, ldk, ck = 256, 9 * Cin, 0
    else:
        ck = ops.round_up(Cout, 64)
        npad, ldk = 256, 9 * ck
    lin = torch.zeros(npad, ldk, device=dev, dtype=torch.bfloat16)
    frg = torch.zeros(npad, ldk, device=dev, dtype=torch.bfloat16)
    ops.pack_weight(w, lin, Cout=Cout, Cin=Cin, RS=9, ldk=ldk, Ck=ck, mode=mode)
    ops.pack_weight(w, frg, Cout=Cout, Cin=Cin, RS=9, ldk=ldk, Ck=ck, mode=mode | ops.PACK_FRAG(npad))
    torch.cuda.synchronize()
    assert torch.equal(frg, ops.frag_order(lin))


@pytest.mark.parametrize("geom", [(2, 33, 33, 256, 2), (4, 65, 65, 256, 2), (1, 40, 24, 64, 1), (3, 17, 19, 128, 4)])
def test_grouped_wgrad_launch_is_bitwise_the_single_launches(dev, geom):
    """simt_conv_wgrad_multi: the three weight-gradient GEMMs of a Bottleneck (1x1 4p -> p, 3x3 p -> p dilated, 1x1 p -> 4p) as ONE
    launch with a shared pixel split count == three simt_conv_wgrad launches with that split count, slab for slab, bit for bit
    (reference: the gradients autograd computes at tools/trainV2_simt.py:428 for model/deeplab_multi.py:62,68,73); and the reduced
    gradient against the torch CPU fp32 reference (oracle/ops_ref.py) at the bf16 tolerance of this file."""
    B, H, W, p, dil = geom
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(31 + p + H)
    M = B * H * W
    probs = []   # (dy [M, Cd], x [B,H,W,Cin], taps, Cd, Cin)
    for (cd, cin, k) in ((p, 4 * p, 1), (p, p, 3), (4 * p, p, 1)):
        dy = torch.randn(M, cd, generator=g).to(dev, dt)
        x = torch.randn(B, H, W, cin, generator=g).to(dev, dt)
        taps = ops.conv_taps(k, k, dil if k == 3 else 1, dil * (k // 2) if k == 3 else 0)
        probs.append((dy, x, taps, cd, cin))
    tco = ops.wgrad_group_tile_co(M, [(cd, len(t) * cin) for (_, _, t, cd, cin) in probs])
    assert tco == (256 if (p % 256 == 0 and M >= ops.WGRAD3_MIN_PIXELS) else 128)
    tiles = sum(ops.wgrad_tiles(M, cd, len(t) * cin, tco) for (_, _, t, cd, cin) in probs)
    ns = ops.wgrad_group_nsplit(M, tiles)
    assert ns >= 1

    def descs(slabs):
        return [ops.make_wgrad_desc(dy, x, sl, B=B, H=H, W=W, Cin=cin, Ho=H, Wo=W, Cd=cd, taps=t, nsplit=ns)
                for (dy, x, t, cd, cin), sl in zip(probs, slabs)]

    s_single = [torch.full((ns, cd, len(t) * cin), float("nan"), device=dev) for (_, _, t, cd, cin) in probs]
    s_multi = [torch.full((ns, cd, len(t) * cin), float("nan"), device=dev) for (_, _, t, cd, cin) in probs]
    for d in descs(s_single):
        assert ops.wgrad_multi_ok(d)
        ops.conv_wgrad_desc(d)
    dm = descs(s_multi)
    table, grid, tco_c = ops.wgrad_multi_table(dm, dev)
    assert grid == tiles * ns and tco_c == tco
    ops.wgrad_multi(table, len(dm), grid, ns, tco)
    torch.cuda.synchronize()
    for a, b in zip(s_single, s_multi):
        assert torch.isfinite(b).all()
        assert torch.equal(a, b)
    # the grouped reduce (simt_wgrad_reduce_multi) == the single reduces, bit for bit; then every gradient against the fp32 reference on
    # the bf16-rounded operands
    ks = [3 if len(t) == 9 else 1 for (_, _, t, _, _) in probs]
    g_multi = [torch.full((cd, cin, k, k), float("nan"), device=dev) for (_, _, _, cd, cin), k in zip(probs, ks)]
    rt, rn, rblocks = ops.wgrad_reduce_multi_table(
        [dict(slab=sl, dst=gm, nsplit=ns, Cd=cd, Ktot=k * k * cin, Cin=cin, co_off=0, tap_off=0, Cout=cd, RS=k * k)
         for (_, _, _, cd, cin), sl, gm, k in zip(probs, s_multi, g_multi, ks)], dev)
    ops.wgrad_reduce_multi(rt, rn, rblocks)
    for (dy, x, taps, cd, cin), sl, gm, k in zip(probs, s_multi, g_multi, ks):
        got = torch.empty(cd, cin, k, k, device=dev)
        ops.wgrad_reduce(sl, got, nsplit=ns, Cd=cd, Ktot=k * k * cin, Cin=cin, co_off=0, tap_off=0, Cout=cd, RS=k * k)
        assert torch.equal(got, gm)
        xr = x.float().cpu().permute(0, 3, 1, 2)
        w = torch.zeros(cd, cin, k, k, requires_grad=True)
        y = torch.nn.functional.conv2d(xr, w, padding=dil * (k // 2), dilation=dil if k == 3 else 1)
        y.backward(dy.float().cpu().view(B, H, W, cd).permute(0, 3, 1, 2))
        assert _rel(got.cpu(), w.grad) <= 1e-2


def test_wgrad_reduce_unaligned_destination(dev):
    """the float4 reduce kernels need a 16-byte aligned destination.  simt_wgrad_reduce falls back to its scalar kernel for a
    gradient that starts 4 bytes off (same sums, same order: bit-identical); the table form (no fallback) refuses it; and every plan
    lays its flat gradient buffer out on 16-byte boundaries (engine.layout_flat_grads), DeepLabv3's 19 + 6 leading biases included."""
    g = torch.Generator().manual_seed(5)
    for (cd, cin, k, ns) in ((64, 64, 3, 3), (128, 32, 1, 4)):
        slab = torch.randn(ns, cd, k * k * cin, generator=g).to(dev)
        kw = dict(nsplit=ns, Cd=cd, Ktot=k * k * cin, Cin=cin, co_off=0, tap_off=0, Cout=cd, RS=k * k)
        aligned = torch.empty(cd * cin * k * k, device=dev)
        ops.wgrad_reduce(slab, aligned, **kw)
        raw = torch.full((cd * cin * k * k + 8,), float("nan"), device=dev)
        off = raw[1:1 + aligned.numel()]
        assert off.data_ptr() % 16 == 4
        ops.wgrad_reduce(slab, off, **kw)
        torch.cuda.synchronize()
        assert torch.equal(off, aligned) and torch.isnan(raw[0]) and torch.isnan(raw[1 + aligned.numel()])
        with pytest.raises(AssertionError):
            ops.wgrad_reduce_multi_table([dict(slab=slab, dst=off, **kw)], dev)
    from simt_amd import model_spec as ms
    from simt_amd.engine_v3 import V3Plan, v3_state_shapes
    st = {k: v.to(dev) for k, v in ms.kaiming_init(v3_state_shapes(19, 6, True), seed=1).items()}
    plan = V3Plan(st, 1, 64, 64, 19, 6, True, dtype=torch.bfloat16, train=True, device=dev)
    assert all(t.data_ptr() % 16 == 0 for t in plan.grads.values())
    assert plan.grad_offsets[plan.grad_order[1]][0] % 4 == 0 and plan.p[plan.grad_order[0]].numel() % 4 != 0


# Round 5: two convs of identical geometry in ONE launch (simt_conv_fprop_pair; the trainable and the frozen net's conv of one layer)
PAIR_CASES = [
    # id, B, H, W, Cin, Cout, k, dil, (epilogue 0, epilogue 1), fused?
    ("3x3 d2 256 stats | bias+relu", 4, 97, 97, 256, 256, 3, 2, ("stats", "bias_relu"), True),
    ("1x1 1024->256 stats | bias+relu", 4, 97, 97, 1024, 256, 1, 1, ("stats", "bias_relu"), True),
    ("rows 256->1024 stats | bias+res+relu", 4, 97, 97, 256, 1024, 1, 1, ("stats", "bias_res_relu"), True),
    ("rows 512->2048 stats | bias+res+relu", 2, 33, 35, 512, 2048, 1, 1, ("stats", "bias_res_relu"), True),
    ("3x3 128 stats | bias+relu (128-column tile)", 2, 49, 49, 128, 128, 3, 1, ("stats", "bias_relu"), True),
    ("3x3 64 stats | bias+relu (64-column tile)", 1, 97, 97, 64, 64, 3, 1, ("stats", "bias_relu"), True),
    ("1x1 1024->2048 stats | bias (downsample)", 1, 33, 33, 1024, 2048, 1, 1, ("stats", "bias"), True),
    ("ragged M: 3x3 256 stats | bias+relu", 1, 13, 17, 256, 256, 3, 2, ("stats", "bias_relu"), True),
    ("rows 64->256 stats | bias+res+relu (different geometries: two launches)", 1, 33, 33, 64, 256, 1, 1, ("stats", "bias_res_relu"), False),
    ("3x3 256 bias+relu | bias+relu (unsupported flavour pair: two launches)", 1, 33, 33, 256, 256, 3, 2, ("bias_relu", "bias_relu"), False),
]


@pytest.mark.parametrize("case", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_pair_launch_is_bitwise_the_two_launches(dev, case):
    import ctypes as C
    from simt_amd import _lib as L
    _id, B, H, W, Cin, Cout, k, dil, epis, want_fused = case
    BF = torch.bfloat16
    M = B * H * W
    g = torch.Generator().manual_seed(Cin + Cout + k)
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    tile = ops.pick_tile_n(Cout, BF)
    npad = ops.round_up(Cout, tile)

    def problem(epi, seed):
        gg = torch.Generator().manual_seed(seed)
        x = torch.randn(B, H, W, Cin, generator=gg).to(dev, BF)
        wp = torch.zeros(npad, len(taps) * Cin, device=dev, dtype=BF)
        wp[:Cout] = (torch.randn(Cout, len(taps) * Cin, generator=gg) * 0.03).to(dev, BF)
        kw = {}
        if epi == "stats":
            kw["stats"] = torch.full(((M + 127) // 128, 2, Cout), float("nan"), device=dev)
        if "bias" in epi:
            kw["bias"] = torch.randn(Cout, generator=gg).to(dev)
        if "res" in epi:
            kw["res"] = torch.randn(M, Cout, generator=gg).to(dev, BF)
        kw["relu"] = "relu" in epi
        return x, wp, kw

    probs = [problem(epis[0], 1), problem(epis[1], 2)]

    def run(pair):
        ys, descs = [], []
        for x, wp, kw in probs:
            y = torch.full((M, Cout), float("nan"), device=dev, dtype=BF)
            if "stats" in kw:
                kw["stats"].fill_(float("nan"))
            ys.append(y)
            descs.append(ops.make_conv_desc(x, wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=tile, **kw))
        if pair:
            assert bool(L.load().simt_conv_pair_fused(C.byref(descs[0]), C.byref(descs[1]))) == want_fused, _id
            L.call("simt_conv_fprop_pair", C.byref(descs[0]), C.byref(descs[1]), ops.stream_ptr())
        else:
            for d in descs:
                ops.conv_fprop_desc(d)
        torch.cuda.synchronize()
        return [y.clone() for y in ys], [kw["stats"].clone() if "stats" in kw else None for _x, _w, kw in probs]
    ya, sa = run(False)
    yb, sb = run(True)
    for i in range(2):
        assert torch.isfinite(yb[i].float()).all()
        assert torch.equal(ya[i], yb[i]), f"{_id}: problem {i} differs between the pair launch and its own launch"
        if sa[i] is not None:
            assert torch.equal(sa[i], sb[i])
