#!/usr/bin/env python3
"""Golden vectors for the warm-up stage, produced by RUNNING THE REFERENCE (build container only).

Imports /root/reference/model/deeplab_multi.py (ResNetMulti, Bottleneck) through the same harness as gen_golden.py and
exec()s the reference's own warm-up loss statements (tools/trainV1_warmup.py, the body of `for sub_i in range(...)`
from `pred1, pred2 = model(image_target)` to `loss.backward()`), read from the reference file at run time.  A small
trunk (layers 1,1,2,1) keeps the problem well conditioned in fp32 (tests/test_gpu_trunk.py explains why the 101-layer
65x65 case is not).  Only inputs, seeds and the reference's OUTPUTS are stored in tests/golden/g11_warmup.npz.
Usage: python oracle/gen_golden_v1.py"""
import os
import sys
import textwrap
import types
import warnings

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import simt_oracle as so  # noqa: E402


def main():
    for name in ["torchvision", "torchvision.models", "torchvision.transforms", "torchvision.transforms.functional"]:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules.setdefault(name, m)
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    os.chdir(os.path.join(REF, "tools"))
    sys.path.insert(0, REF)
    import model.deeplab_multi as dm
    with open(os.path.join(REF, "tools", "trainV1_warmup.py")) as f:
        lines = f.readlines()
    i0 = next(i for i, l in enumerate(lines) if l.strip() == "pred1, pred2 = model(image_target)")
    i1 = next(i for i, l in enumerate(lines) if l.strip() == "loss.backward()" and i > i0)
    body = textwrap.dedent("".join(lines[i0:i1 + 1]))
    layers = (1, 1, 2, 1)
    cd = so.load_class_dist()
    torch.manual_seed(0)
    model = dm.ResNetMulti(dm.Bottleneck, list(layers), 19)
    shapes = so.state_shapes(19, 0, False, layers=layers)
    assert list(model.state_dict().keys()) == list(shapes.keys())
    st = so.recipe_state(shapes, seed=77, head_scale=8.0)
    model.load_state_dict(st)
    model.train()
    args = types.SimpleNamespace(learning_rate=2.5e-4, lambda_seg=0.1, iter_size=1, momentum=0.9, weight_decay=5e-4,
                                 num_steps=250000, power=0.9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        optimizer = optim.SGD(model.optim_parameters(args, warmup=True), lr=args.learning_rate, momentum=args.momentum,
                              weight_decay=args.weight_decay, foreach=False)
    name_of = {id(p): k for k, p in model.named_parameters()}
    g0 = [name_of[id(p)] for p in optimizer.param_groups[0]["params"]]
    g1 = [name_of[id(p)] for p in optimizer.param_groups[1]["params"]]
    og0, og1 = so.optim_param_names(shapes, warmup=True, openset=False)
    assert g0 == og0 and g1 == og1, "warm-up optim_parameters listing differs from the oracle's"
    B, H, W = 2, 97, 97
    interp_target = nn.Upsample(size=(H, W), mode="bilinear", align_corners=True)
    seg_loss = torch.nn.CrossEntropyLoss(ignore_index=255)
    SAMPLE_KEYS = ["conv1.weight", "layer1.0.conv2.weight", "layer3.1.conv3.weight", "layer4.0.downsample.0.weight",
                   "layer6.conv2d_list.1.weight", "layer5.conv2d_list.0.bias"]
    losses, samples, imgs, labs = [], [], [], []
    for it in range(2):
        img, lab = so.synthetic_batch(B, H, W, cd.numpy(), seed=500 + it, block=8)
        imgs.append(img)
        labs.append(lab)
        optimizer.zero_grad()
        lr = so.lr_poly(args.learning_rate, it, args.num_steps, args.power)     # adjust_learning_rate (:133-137)
        optimizer.param_groups[0]["lr"] = lr
        optimizer.param_groups[1]["lr"] = lr * 10
        ns = dict(model=model, image_target=img, label_target=lab, interp_target=interp_target, seg_loss=seg_loss, args=args)
        exec(body, ns)
        optimizer.step()
        losses.append([float(ns["loss"]), float(ns["loss_seg1"]), float(ns["loss_seg2"])])
        sd = model.state_dict()
        samples.append(np.stack([np.pad(sd[k].flatten()[:64].numpy().copy(), (0, max(0, 64 - sd[k].numel()))) for k in SAMPLE_KEYS]))
    out = os.path.join(ROOT, "tests", "golden", "g11_warmup.npz")
    np.savez_compressed(out, losses=np.array(losses), sample_keys=np.array(SAMPLE_KEYS), param_samples=np.stack(samples),
                        layers=np.array(layers), seeds=np.array([500, 501]), n_listed=np.array([len(g0), len(set(g0)), len(g1)]))
    print("wrote", out, losses)


if __name__ == "__main__":
    main()
