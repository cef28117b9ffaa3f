"""The data-parallel exchange table of the production plan (BASELINE configs[1] per GPU = configs[2] per rank: DeepLabv2-R101 + SimT, B = 4,
768 x 768, bf16): gradient names in flat-buffer order, padded spans, the backward launch index at which each becomes final, the buckets
make_buckets cuts and the hook points of the replay.  Written once on a GPU box (the launch indices come from the plan's backward list):

    python profiles/tools/dump_bucket_table.py tests/golden/g16_dp_bucket_table.json

tests/test_dp_gloo.py replays it with eight gloo ranks on CPU; tests/test_gpu_dp.py checks that the live plan still produces it."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import model_spec as ms                     # noqa: E402
from simt_amd.dp import make_buckets                      # noqa: E402
from simt_amd.step import Hyper, SimTTrainer              # noqa: E402


def table(B=4, H=768, W=768):
    dev = torch.device("cuda:0")
    K = 3
    cd = ms.load_class_dist("bapa")
    st = ms.reference_init(ms.state_shapes(19, K, True), seed=1234)
    fst = ms.reference_init(ms.state_shapes(19, 0, False), seed=1234)
    tr = SimTTrainer(st, fst, ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), Hyper(open_classes=K), cd, B, H, W, dtype=torch.bfloat16, device=dev)
    order, sizes, end = tr.exchange_table()
    ready = tr.plan.grad_ready
    buckets = make_buckets(order, sizes, ready, bucket_elems=tr.BUCKET_ELEMS)
    return {"config": f"DeepLabv2-R101 + SimT(C=19,K={K}), B={B}, {H}x{W}, bf16", "bucket_elems": tr.BUCKET_ELEMS,
            "order": order, "sizes": [int(sizes[n]) for n in order], "ready": [int(ready[n]) for n in order], "end": int(end),
            "flat_elems": int(tr.plan.flat_grad.numel()), "backward_launches": len(tr.plan.bwd_list.items),
            "hook_points": sorted({int(v) for v in ready.values()}), "early_cut": int(tr._early_cut) if tr._early_cut is not None else None,
            "buckets": [[int(s), int(e), int(r)] for s, e, r in buckets]}


if __name__ == "__main__":
    t = table()
    with open(sys.argv[1], "w") as f:
        json.dump(t, f, indent=0)
    print(len(t["order"]), "tensors", t["end"] * 4 / 1e6, "MB exchanged;", len(t["buckets"]), "buckets:", [(e - s) * 4 >> 20 for s, e, _ in t["buckets"]],
          "MB; ready", [r for _s, _e, r in t["buckets"]], "of", t["backward_launches"], "launches; early cut", t["early_cut"])
