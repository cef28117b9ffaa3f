"""What would ONE launch for two convs of the same shape save?  Time 2 x conv(B=4) back to back against 1 x conv(B=8) (same kernel, 510 workgroups),
rotating buffer sets, HIP events over 40 repetitions."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L
from simt_amd import ops
BF, dev = torch.bfloat16, torch.device("cuda:0")
lib = L.load(); st = torch.cuda.current_stream().cuda_stream
H = W = 97
def mk(B, Cin, Cout, k, dil, epi):
    taps = ops.conv_taps(k, k, dil, dil * (k // 2)); M = B * H * W
    sets = []
    for _ in range(6):
        x = torch.randn(M, Cin, device=dev).to(BF)
        wp = (torch.randn(Cout, len(taps) * Cin, device=dev) * 0.02).to(BF)
        y = torch.empty(M, Cout, device=dev, dtype=BF)
        kw = dict(stats=torch.zeros((M + 127) // 128, 2, Cout, device=dev)) if epi == "stats" else dict(bias=torch.zeros(Cout, device=dev), relu=True)
        d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=Cout, tile_n=256, **kw)
        sets.append((d, (x, wp, y, kw)))
    return sets
def run(seq, reps=40):
    for d, _ in seq[:4]:
        lib.simt_conv_fprop(C.byref(d), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        for d, _ in seq:
            lib.simt_conv_fprop(C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for (Cin, Cout, k, dil) in ((256, 256, 3, 2), (1024, 256, 1, 1), (256, 1024, 1, 1)):
    s4a, s4b, s8 = mk(4, Cin, Cout, k, dil, "stats"), mk(4, Cin, Cout, k, dil, "bias"), mk(8, Cin, Cout, k, dil, "stats")
    two = [s4a[0], s4b[0], s4a[1], s4b[1], s4a[2], s4b[2]]          # T conv, F conv alternating: 6 launches
    one = [s8[0], s8[1], s8[2]]                                      # 3 launches of twice the pixels
    t2, t1 = run(two), run(one)
    print(f"{k}x{k} {Cin}->{Cout}: 6 launches of B=4 {t2:.1f} us ({t2/6:.1f} each) vs 3 launches of B=8 {t1:.1f} us ({t1/3:.1f} each): one launch per pair saves {(t2-t1)/3:.1f} us per pair")
