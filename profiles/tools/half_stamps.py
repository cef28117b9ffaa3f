"""Where the stages of the half-split conv schedule (csrc/experiments/conv_igemm2_half.hip, SIMT_CONV2_HALF=3: the stamped build) wait: clocks
per stage in the counted vmcnt wait (LDS-DMA pieces of the stage landing) and in the barrier behind it, wave 0 and wave 4 of every workgroup.
usage (GPU box): SIMT_CONV2_HALF=3 python profiles/tools/half_stamps.py <ablation lib>"""
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
for (Cin, Cout, k, dil) in ((256, 256, 3, 2), (1024, 256, 1, 1)):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    x = torch.randn(M, Cin, device=dev).to(BF)
    wp = (torch.randn(256, len(taps) * Cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, Cout, device=dev, dtype=BF)
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=256, tile_n=256, stats=stats)
    for rep in range(3):
        _ = big.view(torch.int64).sum()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert fn(C.byref(d), st) == 0
        e1.record()
        torch.cuda.synchronize()
    nb = 255
    out = (C.c_uint64 * (nb * 8))()
    assert lib.simt_debug_hstamps(out, nb) == 0
    v = np.array(out, dtype=np.int64).reshape(nb, 2, 4)
    nk = int(v[0, 0, 3])
    for late in (0, 1):
        vm, bar, loop = (np.median(v[:, late, i]) for i in range(3))
        print(f"{k}x{k} {Cin}->{Cout}: wave {4 * late}: K loop {loop:.0f} clocks = {loop / nk:.0f} per stage ({nk} stages); per stage in the vmcnt wait "
              f"{vm / max(nk - 3, 1):.0f}, in the barrier {bar / max(nk - 3, 1):.0f} (stages 1 .. nk-3); launch {e0.elapsed_time(e1) * 1e3:.1f} us")
