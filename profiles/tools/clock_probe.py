"""The shader clock a conv launch actually runs at: s_memtime (shader clock) against s_memrealtime (constant 100 MHz) between kernel start and
end of every workgroup (conv2_common.h STAMP slots 0 / 6 / 7, -DSIMT_ABLATION library), after a warm-up loop of back-to-back launches.
usage (GPU box; the variant is read once per process):
  [SIMT_CONV2_HALF=1 | SIMT_CONV2_MODE=1|2] python profiles/tools/clock_probe.py <ablation lib>"""
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
which = "simt_debug_stamps_half" if os.environ.get("SIMT_CONV2_HALF") else "simt_debug_stamps_abl" if os.environ.get("SIMT_CONV2_MODE") else "simt_debug_stamps"
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("SIMT_CONV2")) or "product"
BF, dev = torch.bfloat16, torch.device("cuda:0")
B, H, W = 4, 97, 97
M = B * H * W
st = torch.cuda.current_stream().cuda_stream
for (Cin, Cout, k, dil) in ((256, 256, 3, 2), (1024, 256, 1, 1)):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2))
    x = torch.randn(M, Cin, device=dev).to(BF)
    wp = (torch.randn(256, len(taps) * Cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, Cout, device=dev, dtype=BF)
    stats = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=256, tile_n=256, stats=stats)
    N = 400
    for _ in range(N):                                   # ~20 ms of back-to-back launches: the power management has settled
        assert fn(C.byref(d), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        assert fn(C.byref(d), st) == 0
    e1.record()
    torch.cuda.synchronize()
    nb = 255

    def stamps():
        out = (C.c_uint64 * (nb * 8))()
        assert getattr(lib, which)(out, nb) == 0
        v = np.array(out, dtype=np.int64).reshape(nb, 8)
        clocks = v[:, 6] - v[:, 0]
        rt = (C.c_uint64 * nb)()                       # s_memrealtime ticks between stamps 0 and 6 (their own array since round 6)
        assert getattr(lib, which + "_rt")(rt, nb) == 0
        return clocks, clocks / (np.array(rt, dtype=np.int64) * 10e-9) / 1e6, v[:, 3] - v[:, 2]
    clocks, mhz, loop = stamps()
    # the same launch after 20 ms of idle GPU (nothing to throttle for): what the kernel takes when the chip grants the nominal clock
    import time
    ts = []
    for _ in range(9):
        torch.cuda.synchronize()
        time.sleep(0.02)
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        assert fn(C.byref(d), st) == 0
        s1.record()
        torch.cuda.synchronize()
        ts.append(s0.elapsed_time(s1) * 1e3)
    clocks1, mhz1, _ = stamps()
    print(f"{tag}: {k}x{k} {Cin}->{Cout}: ONE launch after 20 ms of idle: {np.median(ts):.1f} us (event pair; min {min(ts):.1f}); in-kernel {np.median(clocks1):.0f} clocks at "
          f"{np.median(mhz1):.0f} MHz")
    print(f"{tag}: {k}x{k} {Cin}->{Cout}: {e0.elapsed_time(e1) * 1e3 / N:.1f} us per launch back to back; in-kernel {np.median(clocks):.0f} clocks, K loop "
          f"{np.median(loop):.0f}; shader clock {np.median(mhz):.0f} MHz (5 % .. 95 % of the workgroups: {np.percentile(mhz, 5):.0f} .. {np.percentile(mhz, 95):.0f})")
