"""Where the time of conv_igemm2_kernel<256, 5, 3> goes ("explain the 1.63x between the compute-only loop and the MFMA pipe
floor"), on the dominant shape (3x3 d2 256 -> 256, M = 37 636, 255 workgroups) and the long-K 1x1 (1024 -> 256), from the -DSIMT_ABLATION
library: launch duration on clean-cold operands, s_memtime stamps of every workgroup outside the K loop (prologue / loop / epilogue) and of one
middle K stage inside it (an early and a late wave), the core clock the chip held.  One process per mode (the mode is read once):

    for m in 0 2 1; do SIMT_CONV2_MODE=$m SIMT_CONV2_KSTAMP=1 python profiles/tools/attrib_conv.py simt_amd/libsimt_hip_abl.so; done

modes: 0 product schedule (the experiments TU's copy of the kernel, generic epilogue), 2 MFMA + fragment reads only, 1 loads only."""
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
lib.simt_debug_stamps_abl.argtypes = [C.c_void_p, C.c_int]
lib.simt_debug_kstamps.argtypes = [C.c_void_p, C.c_int]
BF, dev = torch.bfloat16, torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
st = torch.cuda.current_stream().cuda_stream
big = torch.empty(600 << 20, device=dev, dtype=torch.uint8).fill_(1)
mode = int(os.environ.get("SIMT_CONV2_MODE", "0"))
MFMA_CYC = {(256, 3): 36 * 1280, (1024, 1): 16 * 1280}      # MFMA cycles per SIMD of the K loop: stages x 2 waves x 40 MFMAs x 16 cycles


def med(v):
    v = np.sort(np.asarray(v))
    return int(v[len(v) // 2])


for (Cin, Cout, k, dil) in ((256, 256, 3, 2), (1024, 256, 1, 1)):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    x = torch.randn(M, Cin, device=dev).to(BF)
    wp = (torch.randn(Cout, len(taps) * Cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, Cout, device=dev, dtype=BF)
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=Cout, tile_n=256, stats=stats)
    ts = []
    for rep in range(7):
        _ = big.view(torch.int64).sum()              # a read pass: evicts, leaves nothing dirty
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        assert lib.simt_conv_fprop(C.byref(d), st) == 0
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    us = float(np.median(ts))
    nwg = 255
    sb = np.zeros((nwg, 8), np.uint64)
    kb = np.zeros((nwg, 16), np.uint64)
    assert lib.simt_debug_stamps_abl(sb.ctypes.data, nwg) == 0 and lib.simt_debug_kstamps(kb.ctypes.data, nwg) == 0
    s, kk = sb.astype(np.int64), kb.astype(np.int64).reshape(nwg, 2, 8)
    ok = s[:, 3] > s[:, 0]
    s, kk = s[ok], kk[ok]
    e = kk[:, 0]                                     # early wave
    mhz = np.median((e[:, 7] - e[:, 5]).astype(np.float64)) / max(1.0, float(us))      # loop + prologue span in clocks over ~ the launch: lower bound
    nk = len(taps) * Cin // 64
    print(f"== mode {mode}: {k}x{k} {Cin}->{Cout} d{dil}: launch {us:.1f} us on clean-cold operands (median of 7); {nk} K stages; {int(ok.sum())} workgroups stamped")
    pro, first, loop = med(s[:, 1] - s[:, 0]), med(s[:, 2] - s[:, 1]), med(s[:, 3] - s[:, 2])
    epi = med(s[:, 6] - s[:, 3]) if (s[:, 6] > s[:, 3]).all() else -1
    inker = med(s[:, 6] - s[:, 0]) if epi >= 0 else med(s[:, 3] - s[:, 0])
    print(f"   clocks (median over workgroups): start->addressing done {pro}, ->first stage landed {first}, K loop {loop} ({loop / max(nk - 1, 1):.0f} per stage), "
          f"loop done->tile sums combined {epi}; in-kernel span {inker}")
    fl = MFMA_CYC[(Cin, k)]
    print(f"   MFMA pipe floor of the K loop: {fl} cycles per SIMD -> loop / floor = {loop / fl:.2f}; in-kernel span / floor = {inker / fl:.2f}")
    for nm, w in (("early wave 0", 0), ("late wave 4", 1)):
        q = kk[:, w]
        if not (q[:, 4] > q[:, 0]).all():
            continue
        d1, d2, d3, d4 = med(q[:, 1] - q[:, 0]), med(q[:, 2] - q[:, 1]), med(q[:, 3] - q[:, 2]), med(q[:, 4] - q[:, 3])
        if w == 0:
            print(f"   {nm}, one middle stage: barrier->reads+pieces issued {d1}, ->fragments landed {d2}, ->40 MFMAs issued {d3}, ->next barrier exit {d4}; stage {med(q[:, 4] - q[:, 0])} clocks")
        else:
            print(f"   {nm}, one middle stage: barrier->40 MFMAs issued {d1}, ->pieces+reads issued {d2}, ->fragments landed {d3}, ->next barrier exit {d4}; stage {med(q[:, 4] - q[:, 0])} clocks")
    rt = med(e[:, 7] - e[:, 5])
    print(f"   wave 0: kernel start -> loop done {rt} clocks")
