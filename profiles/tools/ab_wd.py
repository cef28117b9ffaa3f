"""A/B of the weights-direct variant of the wide conv kernel (conv_igemm2_kernel<256, tm, 3, 0, 0, 1>: weight fragments loaded straight
into registers from the fragment-ordered copy) against the LDS-staged form <..., 0> of the SAME library, one process, one device, interleaved
rounds over rotating buffer sets (operands from HBM / L2 as in the step).
Needs a -DSIMT_ABLATION build:  SIMT_LIB_PATH=scratch/libsimt_abl.so SIMT_WDIRECT=1 python profiles/tools/ab_wd.py [filter]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simt_amd import _lib as L
from simt_amd import ops

BF = torch.bfloat16
dev = torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
NSETS = 6


def make(case, frag):
    name, Cin, Cout, k, dil, epi = case
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    sets = []
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(NSETS):
        x = torch.randn(M, Cin, device=dev, generator=g).to(BF)
        npad = ops.round_up(Cout, 256)
        wp = (torch.randn(npad, len(taps) * Cin, device=dev, generator=g) * 0.02).to(BF)
        wf = ops.frag_order(wp) if frag else None
        f32 = "f32" in epi
        y = torch.empty(M, ops.round_up(Cout, 8), device=dev, dtype=torch.float32 if f32 else BF)
        kw = {}
        if "stats" in epi:
            kw["stats"] = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
        if "bnr" in epi:
            kw["bnr"] = {"y": torch.randn(M, Cout, device=dev).to(BF), "mean": torch.randn(Cout, device=dev), "rstd": torch.rand(Cout, device=dev) + 0.5,
                         "scale": torch.rand(Cout, device=dev) + 0.5, "shift": torch.randn(Cout, device=dev), "mode": 2,
                         "part": torch.zeros(M // 128 + 2, 3, Cout, device=dev)}
        d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=256,
                               w_frag=wf, **kw)
        assert ops.conv_wants_frag(d), name
        sets.append((d, (x, wp, wf, y, kw)))
    return sets


CASES = [("3x3 256->256 d2 stats", 256, 256, 3, 2, "stats"), ("3x3 256->256 d2 bnr2", 256, 256, 3, 2, "bnr2"), ("3x3 512->512 d4 stats", 512, 512, 3, 4, "stats"),
         ("1x1 1024->256 stats", 1024, 256, 1, 1, "stats"), ("1x1 1024->256 bnr2", 1024, 256, 1, 1, "bnr2"), ("1x1 2048->512 stats", 2048, 512, 1, 1, "stats"),
         ("1x1 1024->2048 stats", 1024, 2048, 1, 1, "stats"), ("head 2048->432 f32", 2048, 432, 1, 1, "f32"), ("head 1024->432 f32", 1024, 432, 1, 1, "f32"),
         ("3x3 128->128(256 pad) d1", 128, 256, 3, 1, "stats")]
if len(sys.argv) > 1:
    CASES = [c for c in CASES if sys.argv[1] in c[0]]
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
for case in CASES:
    arms = [make(case, False), make(case, True)]
    flops = 2.0 * M * case[2] * (case[3] ** 2) * case[1]
    res = [[], []]
    for rnd in range(6):
        for li, sets in enumerate(arms):
            for d, _ in sets:
                lib.simt_conv_fprop(C.byref(d), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for d, _ in sets:
                    assert lib.simt_conv_fprop(C.byref(d), st) == 0
            e1.record()
            torch.cuda.synchronize()
            res[li].append(e0.elapsed_time(e1) / (4 * NSETS) * 1e3)
    a, b = np.median(res[0]), np.median(res[1])
    same = all(torch.equal(arms[0][i][1][3], arms[1][i][1][3]) for i in range(NSETS))
    print(f"{case[0]:30s} LDS {a:7.1f} us ({flops / a / 1e6:6.0f} TF/s)   direct {b:7.1f} us ({flops / b / 1e6:6.0f} TF/s)   direct/LDS {b / a:5.3f}   "
          f"min {min(res[0]):.1f} / {min(res[1]):.1f}   outputs bit-identical: {same}", flush=True)
    del arms
    torch.cuda.empty_cache()
