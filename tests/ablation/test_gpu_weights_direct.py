"""Ablation-build only (csrc/build.sh ABLATION=1 -> simt_amd/libsimt_hip_abl.so, SIMT_LIB_PATH set to it, SIMT_WDIRECT=1): round 3's weights-direct
experiment on the wide conv kernel (csrc/experiments/conv_igemm2_abl.hip; measured 15-40 % slower).  No shipped binary
contains it, so this directory is NOT collected by default (tests/conftest.py): run it with

    SIMT_ABLATION_TESTS=1 SIMT_LIB_PATH=simt_amd/libsimt_hip_abl.so SIMT_WDIRECT=1 python -m pytest tests/ablation -m ablation
"""
import pytest
import torch

from simt_amd import ops

pytestmark = [pytest.mark.gpu, pytest.mark.ablation]


def _rel(a, b):
    return (a.double() - b.double()).abs().max().item() / max(b.double().abs().max().item(), 1e-30)


WD_CASES = [
    # B, H, W, Cin, Cout, k, dil, epilogue   (M = 5 776 / 7 396: 160-row and 128-row tiles, odd / even stage counts, one stage)
    (1, 76, 76, 256, 256, 3, 2, "stats"),
    (1, 86, 86, 192, 512, 3, 4, "bias res relu"),
    (1, 76, 76, 1024, 256, 1, 1, "stats"),
    (1, 76, 76, 64, 200, 1, 1, "plain"),
    (1, 76, 76, 512, 256, 1, 1, "plain"),
    (1, 76, 76, 2048, 432, 1, 1, "f32"),
]


@pytest.mark.parametrize("case", WD_CASES)
def test_weights_direct_kernel_is_bitwise_the_lds_kernel(dev, case):
    """conv_igemm2x_kernel<256, *, 3, 0, 0, 1> (csrc/experiments/conv_igemm2_abl.hip, -DSIMT_ABLATION builds: weights straight into registers from the fragment-ordered copy) against the same launch
    without w_frag (weights staged through LDS): same MFMA chain per accumulator -> bit-identical outputs and statistics; and both against
    torch-CPU (1e-2 of max|ref|, the bf16 bar of this file)."""
    B, H, W, Cin, Cout, k, dil, epi = case
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(99 + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, BF)
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    npad = ops.round_up(Cout, 256)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05)
    wp = torch.zeros(npad, len(taps) * Cin, device=dev, dtype=BF)
    ops.pack_weight(w.to(dev), wp, Cout=Cout, Cin=Cin, RS=k * k, ldk=len(taps) * Cin, mode=0)
    wf = torch.zeros_like(wp)
    ops.pack_weight(w.to(dev), wf, Cout=Cout, Cin=Cin, RS=k * k, ldk=len(taps) * Cin, mode=ops.PACK_FRAG(npad))
    M = B * H * W
    outs = []
    for frag in (None, wf):
        f32 = "f32" in epi
        ldy = ops.round_up(Cout, 8)
        y = torch.full((M, ldy), float("nan"), device=dev, dtype=torch.float32 if f32 else BF)
        kw = {}
        if "stats" in epi:
            kw["stats"] = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
        if "res" in epi:
            kw["res"] = torch.randn(M, Cout, generator=torch.Generator().manual_seed(5)).to(dev, BF)
        if "bias" in epi:
            kw["bias"] = torch.randn(Cout, generator=torch.Generator().manual_seed(6)).to(dev)
        kw["relu"] = "relu" in epi
        d = ops.make_conv_desc(x, wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=256, w_frag=frag, **kw)
        if not ops.conv_wants_frag(d):
            pytest.skip("weights-direct kernel: -DSIMT_ABLATION builds with SIMT_WDIRECT=1 only (it measured slower; DESIGN.md section 9)")
        ops.conv_fprop_desc(d)
        torch.cuda.synchronize()
        outs.append((y.clone(), kw.get("stats")))
    (ya, sa), (yb, sb) = outs
    assert torch.equal(ya[:, :Cout].float(), yb[:, :Cout].float()), f"max diff {(ya[:, :Cout].float() - yb[:, :Cout].float()).abs().max().item()}"
    if sa is not None:
        assert torch.equal(sa, sb)
    xr = x.float().cpu().permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xr, w.to(BF).float(), padding=dil * (k // 2), dilation=dil)
    if "bias" in epi:
        ref = ref + kw["bias"].cpu().view(1, -1, 1, 1)
    if "res" in epi:
        ref = ref + kw["res"].float().cpu().view(B, H, W, Cout).permute(0, 3, 1, 2)
    if "relu" in epi:
        ref = ref.clamp_min(0)
    got = yb[:, :Cout].float().cpu().view(B, H, W, Cout).permute(0, 3, 1, 2)
    assert _rel(got, ref) < 1e-2
