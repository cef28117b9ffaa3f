"""Where the fused-BatchNorm tail of conv_igemm2_kernel<256, *, 3, 1> spends its time: s_memtime stamps (100 MHz) of every workgroup from
a -DSIMT_ABLATION build (csrc/build.sh ABLATION=1 -> simt_amd/libsimt_hip_abl.so; copy it to gpurun_out-independent place that travels).
Slots: 3 main loop done, 4 tile in LDS, 5 rows stored, 6 tile sums combined, 1 stores drained, 2 arrived (owners: everybody arrived),
7 constants seen, 0 end of the tail.

    python profiles/tools/stamps_fbn.py <path to the ablation library>
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L                    # noqa: E402
from simt_amd import ops                          # noqa: E402

lib = C.CDLL(sys.argv[1])
fn = lib.simt_conv_fprop
fn.restype, fn.argtypes = L.SIGNATURES["simt_conv_fprop"]
lib.simt_debug_stamps.argtypes = [C.c_void_p, C.c_int]
BF, dev = torch.bfloat16, torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
for (Cin, Cout, k, dil) in ((256, 256, 3, 2), (1024, 256, 1, 1)):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    x = torch.randn(M, Cin, device=dev).to(BF)
    wp = (torch.randn(Cout, len(taps) * Cin, device=dev) * 0.02).to(BF)
    y, a = torch.empty(M, Cout, device=dev, dtype=BF), torch.empty(M, Cout, device=dev, dtype=BF)
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    cst = [torch.zeros(Cout, device=dev) for _ in range(6)]
    for fused in (0, 1):
        d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=Cout, tile_n=256,
                               stats=stats)
        if fused:
            fd = L.FbnDesc()
            bar = torch.zeros(L.load().simt_conv_fbn_words(C.byref(d)), device=dev, dtype=torch.int64)
            fd.mode, fd.ldo, fd.out, fd.work = 1, Cout, a.data_ptr(), bar.data_ptr()
            fd.mean, fd.rstd, fd.scale, fd.shift, fd.running_mean, fd.running_var = (t.data_ptr() for t in cst)
            fd.momentum, fd.eps = 0.1, 1e-5
            d.fbn = C.addressof(fd)
        ts = []
        for _ in range(4):
            flush.zero_()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert lib.simt_conv_fprop(C.byref(d), st) == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print(f"== {Cin} -> {Cout} k{k} {'fused BatchNorm' if fused else 'statistics flavour'}: launch {np.median(ts):.1f} us (cold caches, median of 4)")
        buf = np.zeros((256, 8), np.uint64)
        assert lib.simt_debug_stamps(buf.ctypes.data, 256) == 0
        b = buf.astype(np.int64)
        b = b[b[:, 3] > 0]
        # s_memtime counts core clocks (~2.1-2.4 GHz) and the XCDs' counters are not aligned: only differences within one workgroup mean
        # anything.  'constants seen' (7) happens within a poll period on every workgroup: the time a workgroup spends between its arrival
        # (2) and (7) is its wait = arrival skew + protocol latency; its MINIMUM over the non-owner workgroups is the protocol latency alone
        # (the last workgroup to arrive waits only for the owners' reduce + publish + its own poll).
        def show(nm, v):
            v = np.sort(v)
            print(f"   {nm:44s} min {v[0]:7d}  p10 {v[len(v) // 10]:7d}  median {v[len(v) // 2]:7d}  p90 {v[len(v) * 9 // 10]:7d}  max {v[-1]:7d}  clocks")
        if not fused:
            show("start -> addressing done", b[:, 1] - b[:, 0])
            show("addressing done -> first stage landed", b[:, 2] - b[:, 1])
            show("first stage landed -> main loop done", b[:, 3] - b[:, 2])
        show("main loop done -> tile in LDS", b[:, 4] - b[:, 3])
        show("tile in LDS -> rows stored", b[:, 5] - b[:, 4])
        show("rows stored -> tile sums combined", b[:, 6] - b[:, 5])
        if fused:
            show("tile sums combined -> granules published", b[:, 1] - b[:, 6])
            show("granules published -> rows of y stored", b[:, 2] - b[:, 1])
            show("rows stored -> constants seen (the wait)", b[:, 7] - b[:, 2])
            show("constants seen -> end (apply from LDS)", b[:, 0] - b[:, 7])
            show("main loop done -> end", b[:, 0] - b[:, 3])
        else:
            show("main loop done -> tile sums combined", b[:, 6] - b[:, 3])
