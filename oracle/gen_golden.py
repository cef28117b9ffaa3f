#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only; never on the GPU box).

What this does
  * imports /root/reference (model.deeplab_multi, model.deeplab, utils.loss, tools/trainV2_simt, evaluate_cityscapes)
    on CPU: torchvision & friends are stubbed in sys.modules, Tensor.cuda / Module.cuda become identity,
    cwd = reference/tools (sig_NTM loads ../ClassDist/ClassDist_bapa.npy relative to cwd);
  * the per-iteration body of the reference has no function boundary (tools/trainV2_simt.py:308-436 is inline in
    main()), so the harness READS those source lines from the reference file at run time, dedents them and exec()s
    them in a namespace it prepared -- the reference's own statements run, nothing is copied into this repo;
  * inputs / weights come from the build-owned seeded recipes in oracle/simt_oracle.py; only seeds, inputs and the
    reference's OUTPUTS are stored.
Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import simt_oracle as so  # noqa: E402


# ------------------------------------------------------------------------------------------------------------
# reference import harness
# ------------------------------------------------------------------------------------------------------------
def import_reference(argv):
    for name in ["torchvision", "torchvision.models", "torchvision.transforms", "torchvision.transforms.functional",
                 "ttach", "scipy.misc"]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    # The reference flattens NCHW->[P,C] with permute(0,2,3,1).view(-1,C) (trainV2_simt.py:357,375,380), which only works
    # for batch_size 1 (its default).  For B>1 fixtures fall back to reshape where view would raise: same values, and
    # the only sensible extension of the reference's semantics to the batched operating points of BASELINE.json.
    _view = torch.Tensor.view

    def view_or_reshape(self, *shape):
        try:
            return _view(self, *shape)
        except RuntimeError:
            return self.reshape(*shape)
    torch.Tensor.view = view_or_reshape
    if not hasattr(np, "int"):
        np.int = int
        np.str = str
        np.float = float
    os.chdir(os.path.join(REF, "tools"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "tools"))
    sys.argv = ["trainV2_simt.py"] + argv
    import importlib
    for m in ["trainV2_simt"]:
        if m in sys.modules:
            del sys.modules[m]
    train = importlib.import_module("trainV2_simt")
    return train


def reference_iteration_source():
    """Lines 308-436 of tools/trainV2_simt.py (the body of `for i_iter in range(args.num_steps):` up to and including
    the optimiser steps), dedented."""
    with open(os.path.join(REF, "tools", "trainV2_simt.py")) as f:
        lines = f.readlines()
    assert lines[306].strip().startswith("for i_iter in range(args.num_steps)"), lines[306]
    assert lines[435].strip() == "optimizer_t2.step()", lines[435]
    return textwrap.dedent("".join(lines[307:436]))


class FakeLoaderIter:
    def __init__(self, batches):
        self.batches = list(batches)
        self.i = 0

    def __next__(self):
        b = self.batches[self.i % len(self.batches)]
        self.i += 1
        return self.i, (b[0], b[1], None, ["synthetic"])


class StubNet:
    """Stands in for `model` / `fixed_model` when a fixture wants to drive the loss body with hand-made logits."""

    def __init__(self, outs):
        self.outs = outs

    def __call__(self, x):
        return self.outs

    def train(self):
        pass

    def load_state_dict(self, sd):
        pass


class StubOpt:
    param_groups = [{"lr": 0.0}]

    def zero_grad(self):
        pass

    def step(self):
        pass


def run_reference_iterations(train, ns, n_iters, capture):
    src = reference_iteration_source()
    code = compile(src, "reference:tools/trainV2_simt.py[308:436]", "exec")
    g = dict(train.__dict__)
    g.update(ns)
    outs = []
    for it in range(n_iters):
        g["i_iter"] = it
        exec(code, g)
        outs.append(capture(g))
    return outs


def make_ns(train, model, fixed_model, optimizer, NTMs, opts, batches, H, W):
    import torch.nn as nn
    from utils.loss import CrossEntropy2d
    NTM1, NTM2, NTM_W1, NTM_W2 = NTMs
    ns = dict(model=model, fixed_model=fixed_model, optimizer=optimizer, NTM1=NTM1, NTM2=NTM2, NTM_W1=NTM_W1,
              NTM_W2=NTM_W2, optimizer_t1=opts[0], optimizer_t2=opts[1], optimizer_w1=opts[2], optimizer_w2=opts[3],
              targetloader_iter=FakeLoaderIter(batches), net_dict={},
              interp_target=nn.Upsample(size=(H, W), mode="bilinear", align_corners=True),
              seg_loss=torch.nn.CrossEntropyLoss(ignore_index=255), Tseg_loss=CrossEntropy2d(is_softmax=False),
              loss_mse=torch.nn.MSELoss(reduction="sum"), h=H, w=W)
    return ns


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    conv = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    print("wrote", name, {k: v.shape for k, v in conv.items()})


def seeded(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


# ------------------------------------------------------------------------------------------------------------
def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    K_DEFAULT = 3
    train = import_reference(["--open-classes", str(K_DEFAULT), "--batch-size", "2", "--input-size-target", "33,33",
                              "--learning-rate", "6e-4", "--learning-rate-T", "6e-3", "--lambda-Convex", "0.1",
                              "--lambda-Volume", "1.0", "--lambda-Anchor", "1.0", "--num-steps", "250000"])
    import torch.optim as optim
    from model import deeplab_multi as dm
    from model import deeplab as dl
    args = train.args
    cd = np.load(os.path.join(REF, "ClassDist", "ClassDist_bapa.npy"))
    for n in ["adapt", "bapa", "dsp", "ltir", "sfdaseg"]:
        np.save(os.path.join(OUT, f"class_dist_{n}.npy"), np.load(os.path.join(REF, "ClassDist", f"ClassDist_{n}.npy")))

    # ---------------- G1 / G2: sig_NTM, sig_W forward + backward
    g1 = {}
    for K in (3, 6, 15):
        Q = 19 + K
        m = dm.sig_NTM(19, K)
        with torch.no_grad():
            m.NTM.copy_(so.ntm_init(19, K, seed=100 + K))
        T = m()
        up = seeded((Q, 19), 200 + K)
        (T * up).sum().backward()
        g1[f"ntm_{K}"] = m.NTM.detach().clone()
        g1[f"T_{K}"] = T
        g1[f"dT_{K}"] = up
        g1[f"dntm_{K}"] = m.NTM.grad
        w = dm.sig_W(19, K)
        with torch.no_grad():
            w.weight.copy_(seeded((Q, Q), 300 + K, 0.5))
        g1[f"w_{K}"] = w.weight.detach().clone()
        Wm = w()
        upw = seeded((Q, Q), 400 + K)
        (Wm * upw).sum().backward()
        g1[f"W_{K}"] = Wm
        g1[f"w_after_{K}"] = w.weight.detach().clone()
        g1[f"dW_{K}"] = upw
        g1[f"dw_{K}"] = w.weight.grad
    npz("g1_g2_ntm_w", **g1)

    # ---------------- G4: loss body on hand-made low-res logits (several variants), one reference iteration each
    def run_head_case(name, K, B, hl, H, seed, lam, variant):
        Q = 19 + K
        args.open_classes, args.batch_size, args.iter_size = K, B, 1
        args.lambda_Convex, args.lambda_Volume, args.lambda_Anchor = lam
        args.input_size_target = f"{H},{H}"
        args.learning_rate_T = 6e-3
        scale = 3.0 if variant != "lowconf" else 0.3
        p1 = seeded((B, Q, hl, hl), seed + 1, scale)
        p2 = seeded((B, Q, hl, hl), seed + 2, scale)
        fx = seeded((B, 19, hl, hl), seed + 3, 4.0 if variant != "lowconf" else 0.2)
        if variant == "open_negative":
            p1[:, 19:] = -p1[:, 19:].abs() - 0.5
            p2[:, 19:] = -p2[:, 19:].abs() - 0.5
        if variant == "ties":
            p1 = (p1 * 2).round() / 2
            p2 = (p2 * 2).round() / 2
            p1[:, 3] = p1[:, 5]
            p2[:, 20] = p2[:, 19]
        if variant == "open_wins":
            p2[:, 19:] += 4.0
            p1[:, 19:] += 2.0
            fx = fx * 0.05
        lab = torch.randint(0, 19, (B, H, H), generator=torch.Generator().manual_seed(seed + 4))
        lab[torch.rand(B, H, H, generator=torch.Generator().manual_seed(seed + 5)) < 0.15] = 255
        if variant == "no_valid":
            # one class at logit 2, the rest ~0 -> max prob 0.29, between the thresholds -> Conf 255 everywhere
            hot = torch.full((B, 1, hl, hl), 7, dtype=torch.long)   # same class everywhere: interpolation keeps 0.29
            fx = fx * 0.0025
            fx.scatter_add_(1, hot, torch.full((B, 1, hl, hl), 2.0))
            p1 = p1 * 0.05
            p2 = p2 * 0.05
        p1.requires_grad_(True)
        p2.requires_grad_(True)
        NTMs = [dm.sig_NTM(19, K), dm.sig_NTM(19, K), dm.sig_W(19, K), dm.sig_W(19, K)]
        with torch.no_grad():
            NTMs[0].NTM.copy_(so.ntm_init(19, K, seed + 6))
            NTMs[1].NTM.copy_(so.ntm_init(19, K, seed + 7))
        ntm_before = [NTMs[0].NTM.detach().clone(), NTMs[1].NTM.detach().clone()]
        opts = [optim.Adam(m.parameters(), lr=args.learning_rate_T, weight_decay=0) for m in NTMs]
        img = torch.zeros(B, 3, H, H)
        ns = make_ns(train, StubNet((p1, p2)), StubNet((None, fx)), StubOpt(), NTMs, opts, [(img, lab)], H, H)
        snap = {}

        def capture(g):
            snap.update(loss=g["loss"].detach(), loss_p1=g["loss_p1"].detach(), loss_p2=g["loss_p2"].detach(),
                        loss_y1=g["loss_y1"].detach(), loss_y2=g["loss_y2"].detach(), place=g["Place_loss"].detach(),
                        convex=g["NTM_Convex_loss"].detach(),
                        volume=torch.as_tensor(g["NTM_Volume_loss"]).detach().float(),
                        anchor=g["NTM_Anchor_loss"].detach(), conf=g["Conf_label_target"].detach())
            return None

        # NTM.grad just before optimizer_t.step() is what we want -> wrap the Adam steps to snapshot first
        orig_steps = [opts[0].step, opts[1].step]

        def snap_then_step(k):
            def f(*a, **kw):
                snap[f"ntm_grad{k + 1}"] = NTMs[k].NTM.grad.detach().clone()
                return orig_steps[k](*a, **kw)
            return f
        opts[0].step = snap_then_step(0)
        opts[1].step = snap_then_step(1)
        run_reference_iterations(train, ns, 1, capture)
        d = dict(K=K, B=B, hl=hl, H=H, lam=np.array(lam, dtype=np.float64), pred_lr1=p1.detach(), pred_lr2=p2.detach(),
                 fixed_lr2=fx, label=lab, ntm1=ntm_before[0], ntm2=ntm_before[1], dpred1=p1.grad, dpred2=p2.grad,
                 ntm1_after=NTMs[0].NTM.detach(), ntm2_after=NTMs[1].NTM.detach(), w1_after=NTMs[2].weight.detach(),
                 w2_after=NTMs[3].weight.detach(),
                 w1_m=opts[2].state[NTMs[2].weight]["exp_avg"], w1_v=opts[2].state[NTMs[2].weight]["exp_avg_sq"],
                 w2_m=opts[3].state[NTMs[3].weight]["exp_avg"], w2_v=opts[3].state[NTMs[3].weight]["exp_avg_sq"],
                 lr_T=np.float64(opts[0].param_groups[0]["lr"]), th=np.array([args.Threshold_high, args.Threshold_low]),
                 lambda_seg=np.float64(args.lambda_seg), lambda_place=np.float64(args.lambda_Place), **snap)
        npz(name, **d)

    lam_sh = (0.1, 1.0, 1.0)   # sh_simt.sh:16
    run_head_case("g4_head_base_k3", 3, 2, 9, 33, 1000, lam_sh, "base")
    run_head_case("g4_head_open_negative_k3", 3, 2, 9, 33, 1100, lam_sh, "open_negative")
    run_head_case("g4_head_ties_k3", 3, 2, 9, 33, 1200, lam_sh, "ties")
    run_head_case("g4_head_open_wins_k3", 3, 2, 9, 33, 1300, lam_sh, "open_wins")
    run_head_case("g4_head_lowconf_k3", 3, 2, 9, 33, 1400, lam_sh, "lowconf")
    run_head_case("g4_head_no_valid_k3", 3, 1, 9, 33, 1500, lam_sh, "no_valid")
    run_head_case("g4_head_base_k15", 15, 2, 9, 41, 1600, (0.5, 0.1, 0.5), "base")
    run_head_case("g4_head_base_k6", 6, 1, 5, 17, 1700, lam_sh, "open_wins")

    # ---------------- G5: Bottleneck train-mode forward/backward (stride / dilation / downsample variants)
    import torch.nn as nn
    g5 = {}
    for tag, (inpl, planes, stride, dil, down) in {"s1d1": (16, 4, 1, 1, False), "s2d1_down": (8, 4, 2, 1, True),
                                                   "s1d2_down": (8, 4, 1, 2, True), "s1d4": (16, 4, 1, 4, False)}.items():
        ds = None
        if down:
            ds = nn.Sequential(nn.Conv2d(inpl, planes * 4, 1, stride=stride, bias=False), nn.BatchNorm2d(planes * 4))
        blk = dm.Bottleneck(inpl, planes, stride=stride, dilation=dil, downsample=ds)
        blk.train()
        sd = blk.state_dict()
        new = {}
        for i, (k, v) in enumerate(sd.items()):
            if v.dtype == torch.long:
                new[k] = v
            elif "running_var" in k or (("bn" in k or "downsample.1" in k) and k.endswith("weight")):
                new[k] = torch.rand(v.shape, generator=torch.Generator().manual_seed(500 + i)) + 0.5
            else:
                new[k] = seeded(v.shape, 500 + i, 0.3)
        blk.load_state_dict(new)
        for p in blk.parameters():
            p.requires_grad_(True)
        x = seeded((2, inpl, 13, 13), 600).requires_grad_(True)
        y = blk(x)
        up = seeded(y.shape, 601)
        (y * up).sum().backward()
        for k, v in new.items():
            g5[f"{tag}.in.{k}"] = v
        g5[f"{tag}.x"] = x.detach()
        g5[f"{tag}.y"] = y
        g5[f"{tag}.up"] = up
        g5[f"{tag}.dx"] = x.grad
        for k, p in blk.named_parameters():
            if "conv" in k or "downsample.0" in k:
                g5[f"{tag}.grad.{k}"] = p.grad
        for k, v in blk.state_dict().items():
            if "running" in k:
                g5[f"{tag}.after.{k}"] = v
    npz("g5_bottleneck", **g5)

    # ---------------- G6: Classifier_Module of both files (2 live branches vs 4)
    g6 = {}
    cm = dm.Classifier_Module(32, [6, 12, 18, 24], [6, 12, 18, 24], 5)
    x = seeded((2, 32, 29, 29), 700).requires_grad_(True)
    for i, c in enumerate(cm.conv2d_list):
        with torch.no_grad():
            c.weight.copy_(seeded(c.weight.shape, 710 + i, 0.05))
            c.bias.copy_(seeded(c.bias.shape, 720 + i, 0.1))
        g6[f"w{i}"] = c.weight.detach().clone()
        g6[f"b{i}"] = c.bias.detach().clone()
    y = cm(x)
    up = seeded(y.shape, 730)
    (y * up).sum().backward()
    g6.update(x=x.detach(), y_multi=y, up=up, dx_multi=x.grad,
              live=np.array([c.weight.grad is not None for c in cm.conv2d_list]))
    for i in range(2):
        g6[f"dw{i}_multi"] = cm.conv2d_list[i].weight.grad
        g6[f"db{i}_multi"] = cm.conv2d_list[i].bias.grad
    npz("g6_classifier", **g6)

    # ---------------- G7 / G8: full DeeplabMulti(19,3,True) with recipe weights; 3 reference iterations
    K = 3
    args.open_classes, args.batch_size, args.iter_size = K, 1, 1
    args.lambda_Convex, args.lambda_Volume, args.lambda_Anchor = lam_sh
    args.learning_rate, args.learning_rate_T = 6e-4, 6e-3
    H = W = 65
    args.input_size_target = f"{W},{H}"
    shapes = so.state_shapes(19, K, True)
    model = dm.DeeplabMulti(num_classes=19, open_classes=K, openset=True)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), "state_dict key order differs"
    assert all(tuple(ref_sd[k].shape) == tuple(shapes[k]) for k in shapes)
    st = so.recipe_state(shapes, seed=1234)
    model.load_state_dict(st)
    model.train()
    fshapes = so.state_shapes(19, 0, False)
    fixed = dm.DeeplabMulti(num_classes=19)
    fst = so.recipe_state(fshapes, seed=4321)
    fixed.load_state_dict(fst)
    fixed.eval()
    for p in fixed.parameters():
        p.requires_grad = False
    img, lab = so.synthetic_batch(1, H, W, cd, seed=1234, block=8)
    # G7: plain forward (train & eval) + backward checksums
    m7 = dm.DeeplabMulti(num_classes=19, open_classes=K, openset=True)
    m7.load_state_dict(st)
    m7.eval()
    with torch.no_grad():
        e1, e2 = m7(img)
    m7.train()
    t1, t2 = m7(img)
    u1, u2 = seeded(t1.shape, 801), seeded(t2.shape, 802)
    ((t1 * u1).sum() + (t2 * u2).sum()).backward()
    g7 = dict(img=img, eval_x1=e1, eval_x2=e2, train_x1=t1, train_x2=t2, up1=u1, up2=u2)
    names, sums, asums, samples = [], [], [], []
    for k, p in m7.named_parameters():
        if p.grad is not None:
            names.append(k)
            sums.append(p.grad.double().sum().item())
            asums.append(p.grad.double().abs().sum().item())
            samples.append(p.grad.flatten()[:: max(1, p.grad.numel() // 7)][:7].double().numpy())
    g7.update(grad_names=np.array(names), grad_sum=np.array(sums), grad_abssum=np.array(asums),
              grad_samples=np.stack([np.pad(s, (0, 7 - len(s))) for s in samples]))
    rs = {k: v for k, v in m7.state_dict().items() if "running" in k and ("layer4.2" in k or k.startswith("bn1"))}
    for k, v in rs.items():
        g7["after." + k] = v
    npz("g7_deeplab_multi", **g7)

    # G8: three full reference iterations (loss body + backward + SGD(foreach=False) + Adam).
    # Head weights are scaled so that logits are decisive (max prob > 0.8 on many pixels): every loss branch is live
    # and arg-max flips from fp32 accumulation-order noise are rare (with N(0,0.01) heads the 3rd iteration is chaotic).
    st8 = so.recipe_state(shapes, seed=1234, head_scale=8.0)
    model.load_state_dict(st8)
    fixed.load_state_dict(so.recipe_state(fshapes, seed=4321, head_scale=8.0))
    NTMs = [dm.sig_NTM(19, K), dm.sig_NTM(19, K), dm.sig_W(19, K), dm.sig_W(19, K)]
    with torch.no_grad():
        NTMs[0].NTM.copy_(so.ntm_init(19, K, 901))
        NTMs[1].NTM.copy_(so.ntm_init(19, K, 902))
    opts = [optim.Adam(m.parameters(), lr=args.learning_rate_T, weight_decay=0) for m in NTMs]
    optimizer = optim.SGD(model.optim_parameters(args), lr=args.learning_rate, momentum=args.momentum,
                          weight_decay=args.weight_decay, foreach=False)
    batches = [so.synthetic_batch(1, H, W, cd, seed=1234 + i, block=8) for i in range(3)]
    ns = make_ns(train, model, fixed, optimizer, NTMs, opts, batches, H, W)
    ns["net_dict"] = fixed.state_dict()

    SAMPLE_KEYS = ["layer3.5.conv2.weight", "layer4.0.downsample.0.weight", "layer6.conv2d_list.1.weight",
                   "layer5_1.conv2d_list.0.bias", "layer4.2.conv3.weight"]
    psamples = []

    def cap8(g):
        sd = model.state_dict()
        psamples.append(np.stack([np.pad(sd[k].flatten()[:64].numpy().copy(), (0, max(0, 64 - sd[k].numel()))) for k in SAMPLE_KEYS]))
        return [float(g["loss"]), float(g["loss_p1"]), float(g["loss_p2"]), float(g["loss_y1"]), float(g["loss_y2"]),
                float(g["Place_loss"]), float(g["NTM_Convex_loss"]), float(g["NTM_Volume_loss"]),
                float(g["NTM_Anchor_loss"]), float((g["Conf_label_target"] != 255).sum())]
    traces = run_reference_iterations(train, ns, 3, cap8)
    g0, g1n = so.optim_param_names(shapes)
    ref_g0 = [id(p) for p in optimizer.param_groups[0]["params"]]
    name_of = {id(p): k for k, p in model.named_parameters()}
    assert [name_of[i] for i in ref_g0] == g0, "optim_parameters listing order differs from the oracle's"
    assert [name_of[id(p)] for p in optimizer.param_groups[1]["params"]] == g1n
    g8 = dict(losses=np.array(traces), ntm1=NTMs[0].NTM.detach(), ntm2=NTMs[1].NTM.detach(),
              w1=NTMs[2].weight.detach(), w2=NTMs[3].weight.detach())
    pn, ps, pa = [], [], []
    for k, v in model.state_dict().items():
        if v.dtype != torch.long:
            pn.append(k)
            ps.append(v.double().sum().item())
            pa.append(v.double().abs().sum().item())
    g8.update(param_names=np.array(pn), param_sum=np.array(ps), param_abssum=np.array(pa))
    g8.update(sample_keys=np.array(SAMPLE_KEYS), param_samples=np.stack(psamples))
    npz("g8_iteration", **g8)

    # ---------------- G9 / G10
    import evaluate_cityscapes as ev
    rng = np.random.RandomState(5)
    gt = rng.randint(0, 19, size=4000)
    gt[rng.rand(4000) < 0.1] = 255
    pr = rng.randint(0, 19, size=4000)
    hist = ev.fast_hist(gt, pr, 19)
    with np.errstate(all="ignore"):
        iu = ev.per_class_iu(hist)
    npz("g9_metric", gt=gt, pred=pr, hist=hist, iu=iu, miou=np.float64(round(np.nanmean(iu) * 100, 2)))
    its = np.array([0, 1, 100, 39999, 249999])
    npz("g10_lr_poly", it=its, lr=np.array([train.lr_poly(6e-4, int(i), 250000, 0.9) for i in its]))
    print("done")


if __name__ == "__main__":
    main()
