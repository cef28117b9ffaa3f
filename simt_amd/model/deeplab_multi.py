"""model.deeplab_multi of the reference over the HIP engine: DeeplabMulti, sig_NTM, sig_W.

Call surface mirrored (reference, read-only): model/deeplab_multi.py:122-242 ResNetMulti / DeeplabMulti --
`forward(x) -> (x1, x2)`, `optim_parameters(args, warmup=False)`, `train()/eval()` BatchNorm semantics, a state_dict
of 656 NCHW fp32 tensors with the reference's key names -- and :244-286 sig_NTM / sig_W (`forward() -> T / W`).

The nn.Module tree below only OWNS the parameters (so state_dict keys, `.modules()` order and the duplicate listing of
`optim_parameters`, SURVEY quirk 4, come out identical); no torch op computes anything: forward/backward replay a
TrunkPlan (simt_amd/engine.py) through libsimt_hip.so.  Modules must live on the GPU; there is no CPU fallback.
"""
import numpy as np
import torch
import torch.nn as nn

from simt_amd import model_spec as ms
from simt_amd import ops
from simt_amd.engine import LAYERS, TrunkPlan, multi_heads

affine_par = True


class Bottleneck(nn.Module):
    """Parameter container of one residual block (computation lives in the plan)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes, affine=affine_par)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=dilation, bias=False, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes, affine=affine_par)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4, affine=affine_par)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        for bn in (self.bn1, self.bn2, self.bn3):
            for p in bn.parameters():
                p.requires_grad = False


class Classifier_Module(nn.Module):
    """ASPP head: four dilated 3x3 convs are constructed, two are live."""

    def __init__(self, inplanes, dilation_series, padding_series, num_classes):
        super().__init__()
        self.conv2d_list = nn.ModuleList(
            nn.Conv2d(inplanes, num_classes, kernel_size=3, stride=1, padding=p, dilation=d, bias=True)
            for d, p in zip(dilation_series, padding_series))
        for m in self.conv2d_list:
            m.weight.data.normal_(0, 0.01)


class _TrunkFn(torch.autograd.Function):
    """Autograd seam: forward/backward of the whole net are two launch-list replays."""

    @staticmethod
    def forward(ctx, module, x, *params):
        plan = module._plan(x, train=True)
        out = plan.forward(x.detach().float())
        ctx.plan = plan
        outs = tuple(out[h.name][..., :h.Q].permute(0, 3, 1, 2).clone() for h in plan.heads)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        for h, g in zip(plan.heads, gouts):
            d = plan.dlogits[h.name]
            d.zero_()
            if g is not None:
                d[:, :h.Q] = g.permute(0, 2, 3, 1).reshape(-1, h.Q).to(d.dtype)
        grads = plan.backward()
        return (None, None) + tuple(grads[n].clone() if n in grads else None for n in ctx.plan._param_order)


class ResNetMulti(nn.Module):
    def __init__(self, block, layers, num_classes, open_classes=0, openset=False):
        super().__init__()
        self.inplanes = 64
        self.openset = openset
        self.num_classes, self.open_classes, self.layers_cfg = num_classes, open_classes, tuple(layers)
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64, affine=affine_par)
        for p in self.bn1.parameters():
            p.requires_grad = False
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4)
        dil = [6, 12, 18, 24]
        self.layer5 = Classifier_Module(1024, dil, dil, num_classes)
        self.layer6 = Classifier_Module(2048, dil, dil, num_classes)
        if self.openset:
            self.layer5_1 = Classifier_Module(1024, dil, dil, open_classes)
            self.layer6_1 = Classifier_Module(2048, dil, dil, open_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0, 0.01)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
        self.compute_dtype = torch.bfloat16     # torch.float32 = parity mode
        self._plans = {}

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion or dilation == 2 or dilation == 4:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion, affine=affine_par))
            for p in downsample._modules["1"].parameters():
                p.requires_grad = False
        seq = [block(self.inplanes, planes, stride, dilation=dilation, downsample=downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            seq.append(block(self.inplanes, planes, dilation=dilation))
        return nn.Sequential(*seq)

    # ---------------------------------------------------------------- engine plumbing
    def _heads(self):
        return multi_heads(self.num_classes, self.open_classes, self.openset)

    def _tensors(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def _plan(self, x, train):
        B, _, H, W = x.shape
        tensors = self._tensors()
        sig = tuple(t.data_ptr() for t in tensors.values())
        key = (B, H, W, bool(train), self.compute_dtype)
        ent = self._plans.get(key)
        if ent is None or ent[1] != sig:
            assert x.is_cuda, "DeeplabMulti runs on the GPU only (HIP engine, no CPU fallback)"
            heads = self._heads()
            plan = TrunkPlan({k: v.data for k, v in tensors.items()}, B, H, W, heads, dtype=self.compute_dtype, train=train,
                             layers=self.layers_cfg)
            plan._param_order = [n for n, _ in self.named_parameters()]
            ent = (plan, sig, -1)
            self._plans[key] = ent
        plan = ent[0]
        ver = sum(t._version for t in tensors.values())
        if ver != ent[2]:
            plan.repack()                      # weights changed (optimiser step / load_state_dict)
            self._plans[key] = (plan, sig, sum(t._version for t in tensors.values()))
        return plan

    def _run(self, x):
        if self.training:
            params = [p for _, p in self.named_parameters()]
            outs = _TrunkFn.apply(self, x, *params)
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.num_batches_tracked += 1
            return outs
        with torch.no_grad():
            plan = self._plan(x, train=False)
            out = plan.forward(x.float())
            return tuple(out[h.name][..., :h.Q].permute(0, 3, 1, 2).clone() for h in plan.heads)

    def forward(self, x):
        x1, x2 = self._run(x)
        return x1, x2

    # ---------------------------------------------------------------- optimiser parameter groups (quirk 4)
    def get_1x_lr_params_NOscale(self, warmup=False):
        roots = ([self.conv1, self.bn1, self.layer1, self.layer2] if warmup else []) + [self.layer3, self.layer4]
        for root in roots:
            for sub in root.modules():           # every sub-module yields its whole subtree again -> duplicates
                for p in sub.parameters():
                    yield p

    def get_10x_lr_params(self):
        heads = [self.layer5, self.layer6] + ([self.layer5_1, self.layer6_1] if self.openset else [])
        for h in heads:
            for p in h.parameters():
                yield p

    def optim_parameters(self, args, warmup=False):
        return [{"params": self.get_1x_lr_params_NOscale(warmup), "lr": args.learning_rate},
                {"params": self.get_10x_lr_params(), "lr": 10 * args.learning_rate}]


def DeeplabMulti(num_classes=21, open_classes=0, openset=False):
    return ResNetMulti(Bottleneck, list(LAYERS), num_classes, open_classes, openset)


# ------------------------------------------------------------------------------------------------------------
class _SigNTMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ntm, cd):
        T = torch.empty_like(ntm)
        ops.sig_ntm(ntm.detach().contiguous(), cd, T_out=T)
        ctx.save_for_backward(ntm.detach(), cd)
        return T

    @staticmethod
    def backward(ctx, dT):
        ntm, cd = ctx.saved_tensors
        dN = torch.empty_like(ntm)
        ops.sig_ntm(ntm.contiguous(), cd, dT=dT.contiguous().float(), dN_out=dN)
        return dN, None


class sig_NTM(nn.Module):
    """T = L1-row-normalise( sigmoid(NTM) * class_prior + [I_C; 0_K] )   (model/deeplab_multi.py:244-263)."""

    def __init__(self, num_classes, open_classes=0, init=None, class_dist_path=None):
        super().__init__()
        T = torch.ones(num_classes + open_classes, num_classes)
        self.register_parameter(name="NTM", param=nn.parameter.Parameter(torch.FloatTensor(T)))
        nn.init.kaiming_normal_(self.NTM, mode="fan_out", nonlinearity="relu")
        self.Identity_prior = torch.cat([torch.eye(num_classes, num_classes), torch.zeros(open_classes, num_classes)], 0)
        Class_dist = ms.load_class_dist("bapa", class_dist_path)    # ../ClassDist/ClassDist_bapa.npy, else the packaged copy
        self.Class_dist = torch.FloatTensor(np.tile(Class_dist, (num_classes + open_classes, 1)))
        self._cd = None

    def forward(self):
        assert self.NTM.is_cuda, "sig_NTM runs on the GPU only (the reference calls .cuda() unconditionally)"
        if self._cd is None or self._cd.device != self.NTM.device:
            self._cd = self.Class_dist[0].contiguous().to(self.NTM.device)
        return _SigNTMFn.apply(self.NTM, self._cd)


class _SigWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight):
        W = torch.empty_like(weight)
        ops.sig_w(weight.data, W_out=W)          # also writes diag(weight) = -1e4 in place, like the reference
        ctx.save_for_backward(weight.detach().clone())
        return W

    @staticmethod
    def backward(ctx, dW):
        (w,) = ctx.saved_tensors
        dw = torch.empty_like(w)
        ops.sig_w(w, dW=dW.contiguous().float(), dweight_out=dw)
        return dw


class sig_W(nn.Module):
    """W = softmax(weight with diag := -1e4, dim=1) - I   (model/deeplab_multi.py:265-286)."""

    def __init__(self, num_classes, open_classes=0):
        super().__init__()
        self.classes = num_classes + open_classes
        init = 1.0 / (self.classes - 1.0)
        self.register_parameter(name="weight", param=nn.parameter.Parameter(init * torch.ones(self.classes, self.classes)))
        self.identity = torch.zeros(self.classes, self.classes) - torch.eye(self.classes)

    def forward(self):
        assert self.weight.is_cuda, "sig_W runs on the GPU only (the reference calls .cuda() unconditionally)"
        return _SigWFn.apply(self.weight)
