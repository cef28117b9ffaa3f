"""A/B of conv kernels between two builds of libsimt_hip.so in ONE process on ONE device (guide rule 24): interleaved rounds over
rotating buffer sets (operands come from HBM, not from a warm cache).  usage: python scratch/ab_conv.py libA.so libB.so"""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simt_amd import _lib as L
from simt_amd import ops

def load(path):
    lib = C.CDLL(path)
    for name in ("simt_conv_fprop", "simt_conv_wgrad"):
        fn = getattr(lib, name); sig = L.SIGNATURES[name]; fn.restype = sig[0]; fn.argtypes = sig[1]
    return lib

BF = torch.bfloat16
dev = torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
NSETS = 6

def bits(M, C): return torch.randint(0, 256, (M, C // 8), dtype=torch.uint8, device=dev)

def make(case):
    name, Cin, Cout, k, dil, epi = case
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    sets = []
    for _ in range(NSETS):
        x = torch.randn(M, Cin, device=dev).to(BF)
        tile = ops.pick_tile_n(Cout, BF); npad = ops.round_up(Cout, tile)
        wp = (torch.randn(npad, len(taps) * Cin, device=dev) * 0.02).to(BF)
        y = torch.empty(M, Cout, device=dev, dtype=BF)
        kw = {}
        if "stats" in epi: kw["stats"] = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
        if "res" in epi: kw["res"] = torch.randn(M, Cout, device=dev).to(BF)
        if "rbits" in epi: kw["res_bits"] = bits(M, Cout)
        if "bias" in epi: kw["bias"] = torch.randn(Cout, device=dev); kw["relu"] = True
        if "bnr" in epi:
            mode = 3 if "bnr3" in epi else 2
            kw["bnr"] = {"y": torch.randn(M, Cout, device=dev).to(BF), "mean": torch.randn(Cout, device=dev), "rstd": torch.rand(Cout, device=dev) + 0.5,
                         "scale": torch.rand(Cout, device=dev) + 0.5, "shift": torch.randn(Cout, device=dev), "bits": bits(M, Cout), "mode": mode,
                         "part": torch.zeros(M // 128 + 2, 3, Cout, device=dev)}
        d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=tile, **kw)
        sets.append((d, (x, wp, y, kw)))
    flops = 2.0 * M * Cout * len(taps) * Cin
    return name, sets, flops

CASES = [("1x1 256->1024 stats", 256, 1024, 1, 1, "stats"), ("1x1 256->1024 bias+res+relu", 256, 1024, 1, 1, "bias res"),
         ("1x1 256->1024 res+rbits+bnr3", 256, 1024, 1, 1, "res rbits bnr3"), ("1x1 1024->256 stats", 1024, 256, 1, 1, "stats"),
         ("1x1 1024->256 bnr2", 1024, 256, 1, 1, "bnr2"), ("3x3 256->256 stats", 256, 256, 3, 2, "stats"), ("3x3 256->256 bnr2", 256, 256, 3, 2, "bnr2"),
         ("1x1 512->2048 stats", 512, 2048, 1, 1, "stats"), ("3x3 512->512 d4 stats", 512, 512, 3, 4, "stats"),
         ("1x1 512->2048 bias+res+relu", 512, 2048, 1, 1, "bias res"), ("1x1 512->2048 res+rbits+bnr3", 512, 2048, 1, 1, "res rbits bnr3"),
         ("1x1 2048->512 stats", 2048, 512, 1, 1, "stats"), ("1x1 1024->2048 stats", 1024, 2048, 1, 1, "stats"), ("head 2048->432 plain", 2048, 432, 1, 1, "plain"),
         ("1x1 128->512 stats", 128, 512, 1, 1, "stats"), ("1x1 128->512 bias+res+relu", 128, 512, 1, 1, "bias res"), ("1x1 128->512 res+rbits+bnr3", 128, 512, 1, 1, "res rbits bnr3")]
if len(sys.argv) > 3: CASES = [c for c in CASES if sys.argv[3] in c[0]]

libs = [load(p) for p in sys.argv[1:3]]
st = torch.cuda.current_stream().cuda_stream
for case in CASES:
    name, sets, flops = make(case)
    res = [[], []]
    for rnd in range(6):
        for li, lib in enumerate(libs):
            for d, _ in sets: lib.simt_conv_fprop(C.byref(d), st)          # warm the code path, not the data
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for d, _ in sets:
                    rc = lib.simt_conv_fprop(C.byref(d), st)
                    assert rc == 0
            e1.record(); torch.cuda.synchronize()
            res[li].append(e0.elapsed_time(e1) / (4 * NSETS) * 1e3)
    a, b = np.median(res[0]), np.median(res[1])
    print(f"{name:34s} A {a:7.1f} us ({flops/a/1e6:6.0f} TF/s)   B {b:7.1f} us ({flops/b/1e6:6.0f} TF/s)   B/A {b/a:5.3f}   min A {min(res[0]):.1f} B {min(res[1]):.1f}", flush=True)
    del sets
    torch.cuda.empty_cache()
