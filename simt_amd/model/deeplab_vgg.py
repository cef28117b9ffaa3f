"""model.deeplab_vgg of the reference over the HIP engine: DeeplabVGG (BASELINE config 5).

Mirrors model/deeplab_vgg.py:6-54: `DeeplabVGG(num_classes, vgg16_caffe_path=None, pretrained=False)`, `forward(x)` -> ONE
tensor [B, num_classes, H/8, W/8], `optim_parameters(args)` = all parameters.  The reference file is Python-2-only
(`range(23)+range(24,30)`, `:34`) and builds its trunk from torchvision's vgg16; the layer list below restates that
construction (pool4/pool5 dropped, conv5_x dilated by 2, fc6/fc7 as dilated 3x3 convs) with the same Sequential indices,
so `state_dict()` keys are `features.{0,2,5,...,31}.{weight,bias}` + `classifier.conv2d_list.{0..3}.{weight,bias}`.
Parity of this trunk is unpinned (no importable reference, no reference test): it is checked against the CPU restatement
in oracle/simt_oracle.py::vgg_forward only."""
import torch
import torch.nn as nn

from simt_amd.engine_vgg import VGG_LAYERS, VggPlan
from simt_amd.model.deeplab_multi import Classifier_Module


class _VggFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        plan = module._plan(x, train=True)
        out = plan.forward(x.detach().float())
        ctx.plan = plan
        h = plan.heads[0]
        return out[h.name][..., :h.Q].permute(0, 3, 1, 2).clone()

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        h = plan.heads[0]
        d = plan.dlogits[h.name]
        d.zero_()
        d[:, :h.Q] = g.permute(0, 2, 3, 1).reshape(-1, h.Q).to(d.dtype)
        grads = plan.backward()
        return (None, None) + tuple(grads[n].clone() if n in grads else None for n in plan._param_order)


class DeeplabVGG(nn.Module):
    def __init__(self, num_classes, vgg16_caffe_path=None, pretrained=False):
        super().__init__()
        mods, at = [], {idx: (cin, cout, dil, pool) for (idx, cin, cout, dil, pool) in VGG_LAYERS}
        i = 0
        while i <= 32:
            if i in at:
                cin, cout, dil, pool = at[i]
                mods += [nn.Conv2d(cin, cout, kernel_size=3, padding=dil, dilation=dil), nn.ReLU(inplace=True)]
                i += 2
                if pool:
                    mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
                    i += 1
            else:
                raise AssertionError(i)
        self.features = nn.Sequential(*mods)
        self.classifier = Classifier_Module(1024, [6, 12, 18, 24], [6, 12, 18, 24], num_classes)
        self.num_classes = num_classes
        if pretrained:
            sd = torch.load(vgg16_caffe_path)
            own = self.state_dict()
            own.update({k: v for k, v in sd.items() if k in own and own[k].shape == v.shape})
            self.load_state_dict(own)
        self.compute_dtype = torch.bfloat16
        self._plans = {}

    def _plan(self, x, train):
        B, _, H, W = x.shape
        tensors = dict(self.named_parameters())
        sig = tuple(t.data_ptr() for t in tensors.values())
        key = (B, H, W, bool(train), self.compute_dtype)
        ent = self._plans.get(key)
        if ent is None or ent[1] != sig:
            assert x.is_cuda, "DeeplabVGG runs on the GPU only (HIP engine, no CPU fallback)"
            plan = VggPlan({k: v.data for k, v in tensors.items()}, B, H, W, self.num_classes, dtype=self.compute_dtype, train=train)
            plan._param_order = [n for n, _ in self.named_parameters()]
            ent = (plan, sig, -1)
        plan = ent[0]
        ver = sum(t._version for t in tensors.values())
        if ver != ent[2]:
            plan.repack()
        self._plans[key] = (plan, sig, ver)
        return plan

    def forward(self, x):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return _VggFn.apply(self, x, *[p for _, p in self.named_parameters()])
        with torch.no_grad():
            plan = self._plan(x, train=False)
            out = plan.forward(x.float())
            h = plan.heads[0]
            return out[h.name][..., :h.Q].permute(0, 3, 1, 2).clone()

    def optim_parameters(self, args):
        return self.parameters()
