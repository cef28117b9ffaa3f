"""Where the PCIe-inclusive step loses time against the resident one.  MEASUREMENT ONLY.

    python profiles/tools/h2d_gap.py > profiles/r06_h2d_gap.txt

One SimTTrainer at BASELINE configs[1]; the batch reaches `step()` in three ways, timed in alternating rounds (median ms per step):
  resident     the same device tensors every step (bench.py's headline)
  full         DevicePrefetcher: pinned uint8 frames -> H2D on the copy stream -> uint8 -> fp32 / int64 conversion kernels -> events (h2d_inclusive)
  no_prep      the same as full with the conversion kernels skipped (the slot's outputs keep the previous contents): PCIe + events only
(full - no_prep = the conversion kernels beside the step; no_prep - resident = the H2D copies, the events and the host work)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import model_spec as ms                     # noqa: E402
from simt_amd.data.pipeline import DevicePrefetcher, InputPrep      # noqa: E402
from simt_amd.step import Hyper, SimTTrainer              # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, H, W, K = 4, 768, 768, 3
    cd = ms.load_class_dist("bapa")
    tr = SimTTrainer(ms.reference_init(ms.state_shapes(19, K, True), seed=1234), ms.reference_init(ms.state_shapes(19, 0, False), seed=1234),
                     ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), Hyper(open_classes=K, lr=6e-4, lr_T=6e-3), cd, B, H, W, dtype=torch.bfloat16, device=dev)
    img, lab = ms.synthetic_batch(B, H, W, cd, seed=1234, device=dev)
    rgb, lab8 = ms.synthetic_batch_u8(B, H, W, cd, seed=1234)
    rgb_p, lab_p = rgb.pin_memory(), lab8.pin_memory()
    rgb_d, lab_d = rgb.to(dev), lab8.to(dev)

    class NoPrep(InputPrep):
        def run(self, *a, **k):
            return

    class NoCopyPrefetcher(DevicePrefetcher):
        @staticmethod
        def _host(t):
            return t

    def feed(kind):
        if kind == "resident":
            while True:
                yield img, lab
        prep = (NoPrep if kind in ("no_prep", "events_only") else InputPrep)(B, (H, W), (W, H), dev)

        def src():
            while True:
                yield (rgb_d, lab_d, None) if kind in ("no_h2d",) else (rgb_p, lab_p, None)
        pf = DevicePrefetcher(src(), prep)
        if kind == "no_h2d":
            # device-resident "host" frames: `is_pinned()` is False for them, so route around the pinned staging copy
            for s in pf.slots:
                s["rgb_h"], s["lab_h"] = s["rgb_d"], s["lab_d"]
        if kind == "events_only":
            for s in pf.slots:
                s["rgb_d"], s["lab_d"] = torch.empty(0, dtype=torch.uint8, device=dev), torch.empty(0, dtype=torch.uint8, device=dev)
        for x, l, _m in pf:
            yield (x, l) if kind not in ("no_prep", "events_only") else (img, lab)

    kinds = ["resident", "full", "no_prep"]
    feeds = {k: feed(k) for k in kinds}
    for k in kinds:
        for _ in range(4):
            tr.step(*next(feeds[k]))
    torch.cuda.synchronize()
    res = {k: [] for k in kinds}
    for rnd in range(5):
        for k in kinds:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                tr.step(*next(feeds[k]))
            torch.cuda.synchronize()
            res[k].append((time.perf_counter() - t0) / 20 * 1e3)
    base = float(np.median(res["resident"]))
    print(f"# B={B} {H}x{W} bf16; 5 alternating rounds x 20 steps; ms per step (median), difference to the resident batch")
    for k in kinds:
        m = float(np.median(res[k]))
        print(f"{k:12s} " + " ".join(f"{v:7.3f}" for v in res[k]) + f"   median {m:7.3f}  ({m - base:+.3f})")


if __name__ == "__main__":
    main()
