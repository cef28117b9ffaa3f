"""One conv shape, launched over rotating buffer sets (operands from HBM), for rocprofv3 --pmc runs (profiles/tools/conv_traffic.sh).
    python profiles/tools/one_conv.py B H W Cin Cout k dil [tile_n]      (tile_n: 256 (default, the wide kernel) | 128 | 64 = the production pick for narrow convs)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L          # noqa: E402
from simt_amd import ops                # noqa: E402

B, H, W, Cin, Cout, k, dil = (int(v) for v in sys.argv[1:8])
TILE = int(sys.argv[8]) if len(sys.argv) > 8 else 256
BF, dev = torch.bfloat16, torch.device("cuda:0")
M = B * H * W
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
taps = ops.conv_taps(k, k, dil, dil * (k // 2))
sets = []
for _ in range(6):
    x = torch.randn(M, Cin, device=dev).to(BF)
    npad = ops.round_up(Cout, TILE)
    wp = (torch.randn(npad, len(taps) * Cin, device=dev) * 0.02).to(BF)
    y = torch.empty(M, Cout, device=dev, dtype=BF)
    bias = torch.zeros(Cout, device=dev)
    d = ops.make_conv_desc(x.view(B, H, W, Cin), wp, y, B=B, H=H, W=W, Cin=Cin, Ho=H, Wo=W, Cout=Cout, taps=taps, Npad=npad, tile_n=TILE, bias=bias,
                           relu=True)
    sets.append((d, (x, wp, y, bias)))
for rep in range(3):
    for d, _ in sets:
        assert lib.simt_conv_fprop(C.byref(d), st) == 0
torch.cuda.synchronize()
