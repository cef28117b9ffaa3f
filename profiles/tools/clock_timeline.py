"""The shader clock over the production step (-DSIMT_ABLATION library: csrc/experiments/clock_sampler.hip): one wave on a third stream samples
s_memtime / s_memrealtime every 20 us while the trainer runs steps on its two streams; markers on the main stream give the step boundaries.
usage (GPU box):  SIMT_LIB_PATH=<ablation lib> python profiles/tools/clock_timeline.py [steps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L                      # noqa: E402
from simt_amd import model_spec as ms               # noqa: E402
from simt_amd.step import Hyper, SimTTrainer        # noqa: E402

lib = C.CDLL(L.LIB_PATH)
lib.simt_debug_clock_sampler.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.simt_debug_mark.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda:0")
K, B, H, W = 3, 4, 768, 768
cd = ms.load_class_dist()
tr = SimTTrainer(ms.kaiming_init(ms.state_shapes(19, K, True), seed=1234), ms.kaiming_init(ms.state_shapes(19, 0, False), seed=1234), ms.ntm_init(19, K, 1),
                 ms.ntm_init(19, K, 2), Hyper(open_classes=K), cd, B, H, W, dtype=torch.bfloat16, device=dev)
img, lab = ms.synthetic_batch(B, H, W, cd, seed=1234, device=dev)
for it in range(40):                                 # warm: the power management has settled
    tr.step(img, lab, it)
torch.cuda.synchronize()
NST = int(sys.argv[1]) if len(sys.argv) > 1 else 4
PERIOD = 2000                                        # 20 us of the 100 MHz counter
N = int((NST * 26e-3 + 4e-3) / 20e-6)
buf = torch.zeros(2 * N, device=dev, dtype=torch.int64)
marks = torch.zeros(NST + 1, device=dev, dtype=torch.int64)
third = torch.cuda.Stream()
main = torch.cuda.current_stream()
assert lib.simt_debug_clock_sampler(buf.data_ptr(), N, PERIOD, third.cuda_stream) == 0
for i in range(NST):
    assert lib.simt_debug_mark(marks[i:].data_ptr(), main.cuda_stream) == 0
    tr.step(img, lab, 40 + i)
assert lib.simt_debug_mark(marks[NST:].data_ptr(), main.cuda_stream) == 0
torch.cuda.synchronize()
v = buf.cpu().numpy().reshape(N, 2)
m = marks.cpu().numpy()
ok = v[:, 1] > 0
v = v[ok]
mhz = np.diff(v[:, 0]) / (np.diff(v[:, 1]) * 10e-9) / 1e6
t = (v[1:, 1] + v[:-1, 1]) / 2
print(f"{len(mhz)} intervals of {np.median(np.diff(v[:, 1])) / 100:.1f} us; steps of {np.diff(m) / 1e5} ms")
inside = (t >= m[0]) & (t < m[-1])
print(f"shader clock over {NST} steps: mean {mhz[inside].mean():.0f} MHz, 5 % {np.percentile(mhz[inside], 5):.0f}, median {np.median(mhz[inside]):.0f}, 95 % {np.percentile(mhz[inside], 95):.0f}, "
      f"min {mhz[inside].min():.0f}, max {mhz[inside].max():.0f}")
# timeline of the LAST step in 1-ms buckets (0 = its marker)
s0, s1 = m[-2], m[-1]
print("last step, 1-ms buckets from its start (mean MHz): " + " ".join(
    f"{mhz[(t >= s0 + b * 1e5) & (t < s0 + (b + 1) * 1e5)].mean():.0f}" for b in range(int((s1 - s0) / 1e5) + 1) if ((t >= s0 + b * 1e5) & (t < s0 + (b + 1) * 1e5)).any()))
