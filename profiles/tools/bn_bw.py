"""Achieved HBM rate of the BatchNorm passes at the production shapes (B = 4, 768 x 768): simt_bn_apply_bits (bn3: y + residual -> z + bit mask;
bn1 / bn2: y -> a) over rotating buffer sets, algorithmic bytes / HIP-event time.  usage: python profiles/tools/bn_bw.py [lib.so ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L          # noqa: E402
from simt_amd import ops                # noqa: E402

BF, dev = torch.bfloat16, torch.device("cuda:0")
libs = sys.argv[1:] or [L.LIB_PATH]


def load(path):
    lib = C.CDLL(path)
    for name in ("simt_bn_apply_bits", "simt_bn_apply", "simt_bn_bwd"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = L.SIGNATURES[name]
    return lib


st = torch.cuda.current_stream().cuda_stream
CASES = [("bn3 layer3 1024ch res+bits", 4 * 97 * 97, 1024, True), ("bn1/2 layer3 256ch", 4 * 97 * 97, 256, False),
         ("bn3 layer1 256ch res+bits", 4 * 193 * 193, 256, True), ("bn1/2 layer1 64ch", 4 * 193 * 193, 64, False),
         ("bn3 layer4 2048ch res+bits", 4 * 97 * 97, 2048, True)]
for name, M, Cn, full in CASES:
    nset = max(3, int(600e6 // (M * Cn * 2 * (3 if full else 2))))
    sets = []
    for _ in range(nset):
        y = torch.randn(M, Cn, device=dev).to(BF)
        z = torch.empty_like(y)
        res = torch.randn(M, Cn, device=dev).to(BF) if full else None
        bits = torch.empty(M, Cn // 8, device=dev, dtype=torch.uint8) if full else None
        sets.append((y, z, res, bits))
    sc, sh = torch.rand(Cn, device=dev) + 0.5, torch.randn(Cn, device=dev)
    nbytes = M * Cn * 2 * (3 if full else 2) + (M * Cn // 8 if full else 0)
    for path in libs:
        lib = load(path)

        def run(s):
            y, z, res, bits = s
            if full:
                assert lib.simt_bn_apply_bits(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), res.data_ptr(), None, None, None, z.data_ptr(), bits.data_ptr(), M, Cn, 1,
                                              ops.BF16, st) == 0
            else:
                assert lib.simt_bn_apply(y.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None, None, None, z.data_ptr(), M, Cn, 1, ops.BF16, st) == 0
        for s in sets:
            run(s)
        ts = []
        for rep in range(5):
            for s in sets:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(s)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
        t = float(np.median(ts))
        print(f"{os.path.basename(path):24s} {name:30s} {nbytes / 1e6:7.1f} MB  {t:7.1f} us  {nbytes / t / 1e6:5.2f} TB/s")
