"""Ablation-build only: the K-stage schedule experiments of round 5 on the wide conv kernel (csrc/experiments/conv_igemm2_roles.hip: SIMT_CONV2_ROLES=1 | 2,
SIMT_CONV2_INTER=1; conv_igemm2_half.hip: SIMT_CONV2_HALF=1 | 2 | 4) keep the MFMA chain of every accumulator, so their outputs and BatchNorm
statistics must be BIT-identical to the shipped kernel's.  The switch is read once per process:

    SIMT_ABLATION_TESTS=1 SIMT_LIB_PATH=$PWD/simt_amd/libsimt_hip_abl.so SIMT_CONV2_HALF=1 python -m pytest tests/ablation/test_gpu_schedules.py -m ablation

The reference side is the SHIPPED library (simt_amd/libsimt_hip.so) loaded beside it."""
import ctypes as C
import os

import pytest
import torch

from simt_amd import _lib as L
from simt_amd import ops

pytestmark = [pytest.mark.gpu, pytest.mark.ablation]
SWITCHES = ("SIMT_CONV2_HALF", "SIMT_CONV2_ROLES", "SIMT_CONV2_INTER")

CASES = [
    # B, H, W, Cin, Cout, k, dil, epilogue     (160-row and 128-row tiles; 36 / 16 / 9 / 2 / 1 K stages; ragged last tile)
    (1, 76, 76, 256, 256, 3, 2, "stats"),
    (1, 86, 86, 256, 256, 3, 2, "bias relu"),
    (1, 76, 76, 1024, 256, 1, 1, "stats"),
    (1, 76, 76, 64, 256, 3, 1, "stats"),
    (1, 76, 76, 128, 256, 1, 1, "bias relu"),
    (1, 76, 76, 64, 256, 1, 1, "stats"),
    (2, 97, 97, 512, 512, 3, 4, "plain"),
]


@pytest.mark.parametrize("case", CASES)
def test_experiment_schedule_is_bitwise_the_shipped_kernel(dev, case):
    if not any(os.environ.get(s, "0") != "0" for s in SWITCHES) or "abl" not in os.path.basename(L.LIB_PATH):
        pytest.skip("needs SIMT_LIB_PATH=<ablation library> and one of " + ", ".join(SWITCHES))
    shipped = C.CDLL(os.path.join(os.path.dirname(L.__file__), "libsimt_hip.so"))
    shipped.simt_conv_fprop.restype, shipped.simt_conv_fprop.argtypes = L.SIGNATURES["simt_conv_fprop"]
    B, H, W, Cin, Cout, k, dil, epi = case
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(7 + Cin + Cout + k)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev, BF)
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    npad = ops.round_up(Cout, 256)
    wp = (torch.randn(npad, len(taps) * Cin, generator=g) * 0.05).to(dev, BF)
    M = B * H * W
    st = torch.cuda.current_stream().cuda_stream
    outs = []
    for lib in (shipped, L.load()):
        y = torch.full((M, Cout), float("nan"), device=dev, dtype=BF)
        kw = {}
        if "stats" in epi:
            kw["stats"] = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
        if "bias" in epi:
            kw["bias"] = torch.randn(Cout, generator=torch.Generator().manual_seed(6)).to(dev)
        kw["relu"] = "relu" in epi
        d = ops.make_conv_desc(x, wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=256, **kw)
        assert lib.simt_conv_fprop(C.byref(d), st) == 0
        torch.cuda.synchronize()
        outs.append((y, kw.get("stats")))
    (ya, sa), (yb, sb) = outs
    assert not torch.isnan(yb.float()).any()
    assert torch.equal(ya, yb), f"{(ya.float() - yb.float()).abs().max().item()}"
    if sa is not None:
        assert torch.equal(sa, sb)
