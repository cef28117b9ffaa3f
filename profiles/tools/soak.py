"""Two-stream soak of the production step: N iterations of BASELINE configs[1] (B = 4, 768 x 768, bf16, K = 3) on the default schedule (main stream:
trainable forward / head / backward chain with the BatchNorm backward fused into layer 3's dgrad launches, i.e. launches that WAIT grid-wide; side
stream: frozen forward, weight gradients, SGD / re-pack), with a SECOND set of inputs every other step.  Passes when every step completes (a starved
waiting launch traps after ~2 s instead of hanging) and the losses stay finite; prints the rate.

    python profiles/tools/soak.py [steps]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import model_spec as ms                     # noqa: E402
from simt_amd.step import Hyper, SimTTrainer              # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
K, B, H, W = 3, 4, 768, 768
cd = ms.load_class_dist("bapa")
tr = SimTTrainer(ms.trained_like_init(ms.state_shapes(19, K, True), seed=1234), ms.trained_like_init(ms.state_shapes(19, 0, False), seed=1234),
                 ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), Hyper(open_classes=K, lr=2.5e-4, lr_T=6e-3, num_steps=steps + 100), cd, B, H, W,
                 dtype=torch.bfloat16, device=dev)
batches = [ms.synthetic_batch(B, H, W, cd, seed=s, device=dev) for s in (5, 6)]
assert tr.plan._fbn_on and tr.plan._fbn_dirs == (2,), "default schedule expected (SIMT_BN_GRID unset)"
for it in range(5):
    tr.step(*batches[it & 1], it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(5, 5 + steps):
    tr.step(*batches[it & 1], it)
    if (it - 4) % 250 == 0:
        lo = tr.losses()
        assert all(torch.isfinite(torch.as_tensor(float(v))) for v in lo.values()), lo
        print(f"step {it - 4:5d}: total {lo['total']:.4f}  ({(time.perf_counter() - t0) / (it - 4) * 1e3:.2f} ms/step incl. the read-outs)", flush=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{steps} steps in {dt:.2f} s = {dt / steps * 1e3:.3f} ms/step = {B * steps / dt:.1f} images/s; fused BatchNorm-backward launches per step: "
      f"{sum(1 for i in tr.plan.bwd_list.items if i.tag and ', 1, 2>' in i.tag)}")
