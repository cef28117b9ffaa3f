#!/usr/bin/env python3
"""Golden vectors for gradient accumulation (--iter-size 2), produced by RUNNING THE REFERENCE (build container only).

Same harness as gen_golden.py: imports /root/reference/tools/trainV2_simt.py and exec()s the reference's own loop body
(lines 308-436: W inner loop, `for sub_i in range(args.iter_size)`, backward, optimiser steps) for two iterations of two
micro-batches each.  A small trunk (layers 1,1,2,1; ResNetMulti is the reference's class) keeps the problem well
conditioned in fp32.  Only seeds, inputs and the reference's OUTPUTS go to tests/golden/g12_iter_size2.npz.
Usage: python oracle/gen_golden_iter2.py"""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gen_golden as gg  # noqa: E402
from oracle import simt_oracle as so  # noqa: E402

LAYERS = (1, 1, 2, 1)
K, B, H, W = 3, 2, 65, 65


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    train = gg.import_reference(["--open-classes", str(K), "--batch-size", str(B), "--input-size-target", f"{W},{H}",
                                 "--learning-rate", "6e-4", "--learning-rate-T", "6e-3", "--lambda-Convex", "0.1",
                                 "--lambda-Volume", "1.0", "--lambda-Anchor", "1.0", "--num-steps", "250000",
                                 "--iter-size", "2"])
    import torch.optim as optim
    from model import deeplab_multi as dm
    args = train.args
    assert args.iter_size == 2
    cd = np.load(os.path.join(gg.REF, "ClassDist", "ClassDist_bapa.npy"))
    shapes = so.state_shapes(19, K, True, layers=LAYERS)
    fshapes = so.state_shapes(19, 0, False, layers=LAYERS)
    model = dm.ResNetMulti(dm.Bottleneck, list(LAYERS), 19, K, True)
    assert list(model.state_dict().keys()) == list(shapes.keys())
    model.load_state_dict(so.recipe_state(shapes, seed=2024, head_scale=8.0))
    model.train()
    fixed = dm.ResNetMulti(dm.Bottleneck, list(LAYERS), 19)
    fixed.load_state_dict(so.recipe_state(fshapes, seed=2025, head_scale=8.0))
    fixed.eval()
    for p in fixed.parameters():
        p.requires_grad = False
    NTMs = [dm.sig_NTM(19, K), dm.sig_NTM(19, K), dm.sig_W(19, K), dm.sig_W(19, K)]
    with torch.no_grad():
        NTMs[0].NTM.copy_(so.ntm_init(19, K, 911))
        NTMs[1].NTM.copy_(so.ntm_init(19, K, 912))
    opts = [optim.Adam(m.parameters(), lr=args.learning_rate_T, weight_decay=0) for m in NTMs]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        optimizer = optim.SGD(model.optim_parameters(args), lr=args.learning_rate, momentum=args.momentum,
                              weight_decay=args.weight_decay, foreach=False)
    batches = [so.synthetic_batch(B, H, W, cd, seed=700 + i, block=8) for i in range(4)]
    ns = gg.make_ns(train, model, fixed, optimizer, NTMs, opts, batches, H, W)
    ns["net_dict"] = fixed.state_dict()
    SAMPLE_KEYS = ["layer3.1.conv2.weight", "layer4.0.downsample.0.weight", "layer6.conv2d_list.1.weight",
                   "layer5_1.conv2d_list.0.bias", "layer4.0.conv3.weight", "layer3.0.conv1.weight"]
    psamples, ntm_after = [], []

    def cap(g):
        sd = model.state_dict()
        psamples.append(np.stack([np.pad(sd[k].flatten()[:64].numpy().copy(), (0, max(0, 64 - sd[k].numel()))) for k in SAMPLE_KEYS]))
        ntm_after.append(np.stack([NTMs[0].NTM.detach().numpy().copy(), NTMs[1].NTM.detach().numpy().copy()]))
        # values of the LAST micro-batch of the iteration + the reference's own running averages over the micro-batches
        return [float(g["loss"]), float(g["loss_p1"]), float(g["loss_p2"]), float(g["loss_y1"]), float(g["loss_y2"]),
                float(g["Place_loss"]), float(g["NTM_Convex_loss"]), float(g["NTM_Volume_loss"]), float(g["NTM_Anchor_loss"]),
                float(g["loss_seg_p1"]), float(g["loss_seg_p2"]), float(g["loss_seg_y1"]), float(g["loss_seg_y2"])]
    traces = gg.run_reference_iterations(train, ns, 2, cap)
    gg.npz("g12_iter_size2", losses=np.array(traces), sample_keys=np.array(SAMPLE_KEYS), param_samples=np.stack(psamples),
           ntm_after=np.stack(ntm_after), w1=NTMs[2].weight.detach(), layers=np.array(LAYERS), meta=np.array([K, B, H, W, 2]))


if __name__ == "__main__":
    main()
