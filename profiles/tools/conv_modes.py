"""Timing ablations of the wide conv kernel on CLEAN-cold operands (a read pass over 600 MB evicts the caches without dirtying them; round 3's numbers
were taken behind a memset, whose write-backs the timed loads then paid for: profiles/microbench/stridebench.hip).  Needs the -DSIMT_ABLATION library;
SIMT_CONV2_MODE is read once per process:  for m in 0 1 2; do SIMT_CONV2_MODE=$m python profiles/tools/conv_modes.py <abl lib>; done
modes: 0 product (the experiments TU's copy of the kernel: generic epilogue), 1 loads only, 2 MFMA + fragment reads only."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L          # noqa: E402
from simt_amd import ops                # noqa: E402

lib = C.CDLL(sys.argv[1])
fn = lib.simt_conv_fprop
fn.restype, fn.argtypes = L.SIGNATURES["simt_conv_fprop"]
BF, dev = torch.bfloat16, torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
st = torch.cuda.current_stream().cuda_stream
big = torch.empty(600 << 20, device=dev, dtype=torch.uint8).fill_(1)
mode = os.environ.get("SIMT_CONV2_MODE", "0")
for (Cin, Cout, k, dil) in ((256, 256, 3, 2), (1024, 256, 1, 1), (512, 512, 3, 4)):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    x = torch.randn(M, Cin, device=dev).to(BF)
    npad = ops.round_up(Cout, 256)
    wp = (torch.randn(npad, len(taps) * Cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, Cout, device=dev, dtype=BF)
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=256, stats=stats)
    res = {}
    for prep in ("dirty", "clean", "fresh"):
        ts = []
        for rep in range(7):
            if prep == "dirty":
                big.zero_()
            else:
                _ = big.view(torch.int64).sum()                  # a read pass: evicts, leaves nothing dirty
                if prep == "fresh":
                    x.copy_(x.roll(1, 0))                        # the operand freshly written by a streaming kernel
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.simt_conv_fprop(C.byref(d), st) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[prep] = np.median(ts)
    print(f"mode {mode}: {k}x{k} {Cin}->{Cout} d{dil}: behind a memset {res['dirty']:.1f} us, clean-cold {res['clean']:.1f} us, operand freshly written {res['fresh']:.1f} us")
