"""Rows kernel: where a persistent workgroup's time goes (ablation build stamps)."""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simt_amd import _lib as L
from simt_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(__file__), os.environ.get("ABL_LIB", "libsimt_rabl0.so")))
fn = lib.simt_conv_fprop; sig = L.SIGNATURES["simt_conv_fprop"]; fn.restype = sig[0]; fn.argtypes = sig[1]
lib.simt_debug_stamps_rows.argtypes = [C.c_void_p, C.c_int]
BF = torch.bfloat16; dev = torch.device("cuda:0"); B, H, W = 4, 97, 97; M = B * H * W
def make(Cin, Cout, epi):
    x = torch.randn(M, Cin, device=dev).to(BF)
    npad = ops.round_up(Cout, 256)
    wp = (torch.randn(npad, Cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, Cout, device=dev, dtype=BF)
    kw = {}
    if "stats" in epi: kw["stats"] = torch.zeros((M + 127) // 128, 2, Cout, device=dev)
    if "res" in epi: kw["res"] = torch.randn(M, Cout, device=dev).to(BF)
    if "bias" in epi: kw["bias"] = torch.randn(Cout, device=dev); kw["relu"] = True
    if "rbits" in epi: kw["res_bits"] = torch.randint(0, 256, (M, Cout // 8), dtype=torch.uint8, device=dev)
    if "bnr3" in epi:
        kw["bnr"] = {"y": torch.randn(M, Cout, device=dev).to(BF), "mean": torch.randn(Cout, device=dev) * 0.2, "rstd": torch.rand(Cout, device=dev) + 0.5,
                     "scale": torch.rand(Cout, device=dev) + 0.5, "shift": torch.randn(Cout, device=dev) * 0.3,
                     "bits": torch.randint(0, 256, (M, Cout // 8), dtype=torch.uint8, device=dev), "mode": 3, "part": torch.zeros(M // 128 + 2, 3, Cout, device=dev)}
    d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=[(0, 0)], Npad=npad, tile_n=256, **kw)
    if "inbn" in epi:            # round 6: BatchNorm + ReLU of the input in the operand path (simt_conv_desc.in_*)
        kw["in"] = (torch.rand(Cin, device=dev) + 0.5, torch.randn(Cin, device=dev) * 0.3, torch.empty(M, Cin, device=dev, dtype=BF))
        d.in_scale, d.in_shift, d.in_out = (t.data_ptr() for t in kw["in"])
    return d, (x, wp, y, kw)
st = torch.cuda.current_stream().cuda_stream
flush = torch.empty(1 << 28, device=dev, dtype=torch.uint8)
for (Cin, Cout, epi) in ((256, 1024, "stats"), (256, 1024, "stats inbn"), (256, 1024, "bias res"), (256, 1024, "res rbits bnr3")):
    d, keep = make(Cin, Cout, epi)
    for _ in range(2):
        flush.zero_(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.simt_conv_fprop(C.byref(d), st); e1.record(); torch.cuda.synchronize()
    big = np.zeros((8192, 8), np.uint64)
    assert lib.simt_debug_stamps_rows(big.ctypes.data, 8192) == 0
    buf = big[:256]
    pw = big[4096:].reshape(-1)[:256 * 8 * 2].reshape(256, 8, 2).astype(np.int64)
    print("   per compute wave (median over WGs) work/stage:", np.round(np.median(pw[:, :, 0], 0) / 20).astype(int).tolist(), " barrier/stage:", np.round(np.median(pw[:, :, 1], 0) / 20).astype(int).tolist())
    b = buf.astype(np.int64)
    S = b[:, 4]
    print(f"== {Cin}->{Cout} {epi}: kernel {e0.elapsed_time(e1)*1e3:.1f} us; stages/WG median {np.median(S)}; loop ticks median {np.median(b[:,3])}")
    print("   compute wave 0 per stage: vmcnt-wait %.0f  barrier-wait %.0f  work %.0f | store wave per stage: barrier-wait %.0f  work %.0f (of which barrier -> slab rows in registers %.0f)" % (
        np.median(b[:, 0] / S), np.median(b[:, 1] / S), np.median(b[:, 2] / S), np.median(b[:, 5] / S), np.median(b[:, 6] / S), np.median(b[:, 7] / S)))
