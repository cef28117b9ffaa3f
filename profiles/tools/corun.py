"""Do an HBM-bound BatchNorm pass and an MFMA-bound conv finish sooner TOGETHER than one after the other?  MEASUREMENT ONLY.

    python profiles/tools/corun.py > profiles/r06_corun.txt

Takes launches out of the production plans (BASELINE configs[1]: the frozen net's layer-3 3x3 conv, the trainable net's layer-3 bn3 apply + residual
pass, its bn1 apply pass, the backward's BatchNorm apply pass, a rows-kernel 256 -> 1024 conv) and replays N of each: A alone, B alone, and A on the main
stream beside B on the side stream (both queues fed before the GPU starts: a host-side gate kernel holds them).  alone_sum = T(A) + T(B); the
co-run time says how much of the shorter one hides."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import _lib as L                            # noqa: E402
from simt_amd import model_spec as ms                     # noqa: E402
from simt_amd.engine import side_stream                   # noqa: E402
from simt_amd.step import Hyper, SimTTrainer              # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, H, W, K = 4, 768, 768, 3
    cd = ms.load_class_dist("bapa")
    tr = SimTTrainer(ms.reference_init(ms.state_shapes(19, K, True), seed=1234), ms.reference_init(ms.state_shapes(19, 0, False), seed=1234),
                     ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), Hyper(open_classes=K, lr=6e-4, lr_T=6e-3), cd, B, H, W, dtype=torch.bfloat16, device=dev)
    img, lab = ms.synthetic_batch(B, H, W, cd, seed=1234, device=dev)
    for _ in range(3):
        tr.step(img, lab)
    torch.cuda.synchronize()

    def find(lst, tag_part, shape_part=None, nth=0):
        k = 0
        for it in lst.items:
            if it.fn is not None and tag_part in (it.tag or "") and (shape_part is None or shape_part in (it.shape or "")):
                if k == nth:
                    return it
                k += 1
        raise KeyError((tag_part, shape_part))

    cands = {
        "conv3x3 (frozen, 256->256 d2, M 37636)": find(tr.fixed.fwd_list, "conv_igemm2_kernel<256, 5, 3", "taps9", 8),
        "conv1x1 1024->256 (frozen)": find(tr.fixed.fwd_list, "conv_igemm2_kernel<256, 5, 3", "K1024", 8),
        "rows 256->1024 bias+res+relu (frozen)": find(tr.fixed.fwd_list, "conv1x1_rows_kernel", "N1024 K256", 8),
        "bn_apply bn3+res+relu [M,1024]": None, "bn_apply bn1 [M,256]": None, "bn_bwd apply [M,1024]": None,
        "wgrad3 group (3 Bottlenecks)": find(tr.plan.bwd_list, "conv_wgrad<bf16>", "group9", 3),
    }
    # BatchNorm passes: by bytes
    # BatchNorm passes: time every one alone once, pick by duration (the items carry no byte counts)
    def t_alone(it):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            assert it.fn(*it.args, torch.cuda.current_stream().cuda_stream) == 0
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 5
    ap = [it for it in tr.plan.fwd_list.items if it.fn is not None and (it.tag or "").startswith("simt_bn_apply")]
    ap = sorted(((t_alone(it), i, it) for i, it in enumerate(ap)), key=lambda x: x[0])
    cands["bn_apply bn3+res+relu [M,1024]"] = ap[len(ap) * 3 // 4][2]        # the 33 wide ones are ~43 us
    cands["bn_apply bn1 [M,256]"] = ap[len(ap) // 4][2]                       # the narrow ones ~9 us
    bw = [it for it in tr.plan.bwd_list.items if it.fn is not None and (it.tag or "").startswith("simt_bn_bwd")]
    bw = sorted(((t_alone(it), i, it) for i, it in enumerate(bw)), key=lambda x: x[0])
    cands["bn_bwd apply [M,1024]"] = bw[len(bw) * 3 // 4][2]
    cands = {k: v for k, v in cands.items() if v is not None}

    mainS, sideS = torch.cuda.current_stream(), side_stream(dev)
    N = 40

    def run(a, b):
        """N launches of a on main, N of b on side (either may be None); wall time from a common start event to both ends, ms"""
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        gate = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
        gate.zero_()                       # ~30 us of GPU work on main while the host enqueues: both queues are fed when it ends
        torch.cuda._sleep(2_000_000)       # ~1 ms spin on main: the enqueue loops below finish behind it
        e0.record(mainS)
        sideS.wait_event(e0)
        for i in range(N):
            if a is not None:
                rc = a.fn(*a.args, mainS.cuda_stream)
                assert rc == 0
            if b is not None:
                rc = b.fn(*b.args, sideS.cuda_stream)
                assert rc == 0
        ea.record(mainS)
        eb.record(sideS)
        torch.cuda.synchronize()
        return e0.elapsed_time(ea) / N * 1e3, e0.elapsed_time(eb) / N * 1e3

    print(f"# B={B} {H}x{W} bf16; {N} launches each; us per launch (span from the common start / N); 'together' = main | side ends")
    alone = {}
    for k, it in cands.items():
        run(it, None)
        alone[k] = min(run(it, None)[0] for _ in range(3))
        print(f"alone   {k:44s} {alone[k]:7.1f} us   (tag {it.tag[:40]})")
    pairs = [("conv3x3 (frozen, 256->256 d2, M 37636)", "bn_apply bn3+res+relu [M,1024]"), ("conv3x3 (frozen, 256->256 d2, M 37636)", "bn_apply bn1 [M,256]"),
             ("conv1x1 1024->256 (frozen)", "bn_apply bn3+res+relu [M,1024]"), ("rows 256->1024 bias+res+relu (frozen)", "bn_apply bn1 [M,256]"),
             ("wgrad3 group (3 Bottlenecks)", "bn_bwd apply [M,1024]"), ("conv3x3 (frozen, 256->256 d2, M 37636)", "conv1x1 1024->256 (frozen)")]
    for ka, kb in pairs:
        if ka not in cands or kb not in cands:
            continue
        r = [run(cands[ka], cands[kb]) for _ in range(3)]
        ta, tb = min(x[0] for x in r), min(x[1] for x in r)
        both = max(ta, tb)
        print(f"together {ka:40s} | {kb:32s}: {ta:7.1f} | {tb:7.1f} us; both done {both:7.1f} vs alone_sum {alone[ka] + alone[kb]:7.1f} vs max {max(alone[ka], alone[kb]):7.1f}"
              f"  -> hidden {alone[ka] + alone[kb] - both:6.1f} us of {min(alone[ka], alone[kb]):6.1f}")


if __name__ == "__main__":
    main()
