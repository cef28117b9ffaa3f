"""Time conv_wgrad2 (+ slab reduce) per production shape under 1..N builds of the library (rotating buffers).  usage: ab_wgrad.py lib1.so [lib2.so ...]"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simt_amd import _lib as L
from simt_amd import ops
def load(path):
    lib = C.CDLL(path)
    for name in ("simt_conv_wgrad", "simt_wgrad_reduce"):
        if name in L.SIGNATURES:
            fn = getattr(lib, name); sig = L.SIGNATURES[name]; fn.restype = sig[0]; fn.argtypes = sig[1]
    return lib
BF = torch.bfloat16; dev = torch.device("cuda:0"); B, H, W = 4, 97, 97; M = B * H * W
CASES = [("3x3 256<-256 d2", 256, 256, 3, 2), ("1x1 1024<-256", 256, 1024, 1, 1), ("1x1 256<-1024", 1024, 256, 1, 1), ("3x3 512<-512 d4", 512, 512, 3, 4),
         ("1x1 2048<-512", 512, 2048, 1, 1), ("1x1 512<-2048", 2048, 512, 1, 1)]
libs = [load(p) for p in sys.argv[1:]]
st = torch.cuda.current_stream().cuda_stream
NSETS = 4
for name, Cin, Cd, k, dil in CASES:
    taps = ops.conv_taps(k, k, dil, dil * (k // 2)); Ktot = len(taps) * Cin
    nsplit = ops.wgrad_nsplit(M, Cd, Ktot, BF)
    sets = []
    for _ in range(NSETS):
        x = torch.randn(B, H, W, Cin, device=dev).to(BF); dy = torch.randn(B, H, W, Cd, device=dev).to(BF)
        slab = torch.empty(nsplit, Cd, Ktot, device=dev)
        wd = ops.make_wgrad_desc(dy, x, slab, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cd=Cd, taps=taps, nsplit=nsplit)
        sets.append((wd, x, dy, slab))
    flops = 2.0 * M * Cd * Ktot
    res = [[] for _ in libs]
    for rnd in range(5):
        for li, lib in enumerate(libs):
            for s in sets: lib.simt_conv_wgrad(C.byref(s[0]), st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for s in sets: assert lib.simt_conv_wgrad(C.byref(s[0]), st) == 0
            e1.record(); torch.cuda.synchronize()
            res[li].append(e0.elapsed_time(e1) / (4 * NSETS) * 1e3)
    print(f"{name:18s} nsplit {nsplit:3d}  " + "   ".join(f"{np.median(r):7.1f} us ({flops / np.median(r) / 1e6:5.0f} TF/s)" for r in res), flush=True)
