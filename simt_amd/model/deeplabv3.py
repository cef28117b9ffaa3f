"""model.deeplabv3 of the reference over the HIP engine: DeepLabv3, sig_NTM, sig_W (BASELINE config 4).

Mirrors model/deeplabv3.py:9-166: `DeepLabv3(nc, openc=0, openset=False)`, `forward(x)` -> ONE tensor
[B, nc (+ openc), H, W] (the bilinear upsample to the input size happens inside the model, `:137`),
`get_1x_lr_params_NOscale()` = the parameters of resnet layer3 / layer4 / fc (each once: only the root listing's names
carry the `resnet_50.` prefix the filter looks for, `:147-151`), `get_10x_lr_params()` = assp + conv (+ conv_1),
`optim_parameters(args)`.  The reference builds its trunk with torchvision's `resnet50(pretrained=True)`; torchvision is
not a dependency here, so the same module tree (conv1, bn1, relu, maxpool, layer1..4, avgpool, fc: identical state_dict
keys, so a torchvision checkpoint loads with `load_state_dict`) is declared below and initialised like torchvision does
(kaiming-normal fan_out convs, unit BN) -- there is no network access for the ImageNet weights.  The nn.Module tree only
OWNS parameters; forward/backward replay a V3Plan (simt_amd/engine_v3.py).  GPU only, no CPU fallback.
Parity of this model is unpinned (no importable reference, no reference test): checked against oracle.v3_forward.

sig_NTM / sig_W (`:168-210`) are the same layers as model/deeplab_multi.py's; the reference copy reads a
`ClassDist_source.npy` that does not exist in its tree (`:179`) -- that name is tried first, then the bapa prior."""
import os

import numpy as np
import torch
import torch.nn as nn

from simt_amd import model_spec as ms
from simt_amd.engine_v3 import V3Plan
from simt_amd.model.deeplab_multi import sig_NTM as _sig_NTM
from simt_amd.model.deeplab_multi import sig_W  # noqa: F401  (same layer, re-exported under the reference's name)


class Bottleneck(nn.Module):
    """torchvision-style residual block (stride on the 3x3 conv): parameter container only."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class _ResNet50Params(nn.Module):
    """Module tree of torchvision.models.resnet50 (layers [3, 4, 6, 3]); layer4 / avgpool / fc exist but never run."""

    def __init__(self, layers=(3, 4, 6, 3), num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, layers[0], 1)
        self.layer2 = self._make_layer(128, layers[1], 2)
        self.layer3 = self._make_layer(256, layers[2], 2)
        self.layer4 = self._make_layer(512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * 4, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def _make_layer(self, planes, blocks, stride):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, kernel_size=1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
        mods = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        mods += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*mods)


class ResNet_50(nn.Module):
    def __init__(self, in_channels=3, conv1_out=64):
        super().__init__()
        self.resnet_50 = _ResNet50Params()
        self.relu = nn.ReLU(inplace=True)


class ASSP(nn.Module):
    def __init__(self, in_channels, out_channels=256):
        super().__init__()
        self.relu = nn.ReLU(inplace=True)
        for i, (k, d) in enumerate(((1, 1), (3, 6), (3, 12), (3, 18), (1, 1)), start=1):
            setattr(self, f"conv{i}", nn.Conv2d(in_channels, out_channels, kernel_size=k, stride=1, padding=d if k == 3 else 0,
                                                dilation=d if k == 3 else 1, bias=False))
            setattr(self, f"bn{i}", nn.BatchNorm2d(out_channels))
        self.convf = nn.Conv2d(out_channels * 5, out_channels, kernel_size=1, stride=1, padding=0, dilation=1, bias=False)
        self.bnf = nn.BatchNorm2d(out_channels)
        self.adapool = nn.AdaptiveAvgPool2d(1)


class _V3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        plan = module._plan(x, train=True)
        ctx.plan = plan
        return plan.forward(x.detach().float()).clone()

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        grads = plan.backward(g.contiguous().float())
        return (None, None) + tuple(grads[n].clone() if n in grads else None for n in plan._param_order)


class DeepLabv3(nn.Module):
    def __init__(self, nc, openc=0, openset=False):
        super().__init__()
        self.nc, self.openc, self.openset = nc, openc, openset
        self.resnet = ResNet_50()
        self.assp = ASSP(in_channels=1024)
        self.conv = nn.Conv2d(256, self.nc, kernel_size=1, stride=1, padding=0)
        if openset:
            self.conv_1 = nn.Conv2d(256, self.openc, kernel_size=1, stride=1, padding=0)
        self.compute_dtype = torch.bfloat16
        self._plans = {}

    # ------------------------------------------------------------------ engine
    def _plan(self, x, train):
        B, _, H, W = x.shape
        tensors = dict(self.named_parameters())
        tensors.update(dict(self.named_buffers()))
        sig = tuple(t.data_ptr() for t in tensors.values())
        key = (B, H, W, bool(train), self.compute_dtype)
        ent = self._plans.get(key)
        if ent is None or ent[1] != sig:
            assert x.is_cuda, "DeepLabv3 runs on the GPU only (HIP engine, no CPU fallback)"
            plan = V3Plan({k: v.data for k, v in tensors.items()}, B, H, W, self.nc, self.openc, self.openset,
                          dtype=self.compute_dtype, train=train)
            plan._param_order = [n for n, _ in self.named_parameters()]
            ent = (plan, sig, -1)
        plan = ent[0]
        ver = sum(t._version for t in tensors.values())
        if ver != ent[2]:
            plan.repack()
        self._plans[key] = (plan, sig, ver)
        return plan

    def forward(self, x):
        if self.training:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                out = _V3Fn.apply(self, x, *[p for _, p in self.named_parameters()])
            else:
                out = self._plan(x, train=True).forward(x.float()).clone()
            for m in self.modules():                     # nn.BatchNorm2d bookkeeping of the modules that ran
                if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None and m not in self._dead_bns():
                    m.num_batches_tracked += 1
            return out
        with torch.no_grad():
            return self._plan(x, train=False).forward(x.float()).clone()

    def _dead_bns(self):
        if not hasattr(self, "_dead"):
            self._dead = {m for m in self.resnet.resnet_50.layer4.modules() if isinstance(m, nn.BatchNorm2d)}
        return self._dead

    # ------------------------------------------------------------------ optimiser groups (model/deeplabv3.py:140-166)
    def get_1x_lr_params_NOscale(self):
        b = [self.resnet]
        for i in range(len(b)):
            for j in b[i].modules():
                for k in j.named_parameters():
                    if "resnet_50.layer3" in k[0] or "resnet_50.layer4" in k[0] or "resnet_50.fc" in k[0]:
                        yield k[1]

    def get_10x_lr_params(self):
        b = [self.assp.parameters(), self.conv.parameters()]
        if self.openset:
            b.append(self.conv_1.parameters())
        for j in range(len(b)):
            for i in b[j]:
                yield i

    def optim_parameters(self, args):
        return [{"params": self.get_1x_lr_params_NOscale(), "lr": args.learning_rate},
                {"params": self.get_10x_lr_params(), "lr": 10 * args.learning_rate}]


class sig_NTM(_sig_NTM):
    def __init__(self, num_classes, open_classes=0, init=None):
        src = os.path.join("..", "ClassDist", "ClassDist_source.npy")          # model/deeplabv3.py:179
        super().__init__(num_classes, open_classes, init, class_dist_path=src if os.path.exists(src) else None)
