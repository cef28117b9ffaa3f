"""DeepLabv3 plan (BASELINE config 4; reference model/deeplabv3.py) on the GPU against the CPU restatement
(oracle.v3_forward -- parity UNPINNED: the reference file needs torchvision, see engine_v3.py's header).

Reduced widths / depths keep the CPU oracle fast; the structure is the real one (stride-2 3x3 convs with a downsample
branch, floor-mode max-pool, five ASSP branches with trainable BN, 5-plane convf, 1x1 heads, align_corners=False upsample).
Train-mode BN makes end-to-end fp32 ill-conditioned (see tests/test_gpu_trunk.py), so the oracle runs in float64 and the
criterion is the one used there: GPU error vs float64 within a small multiple of the fp32-CPU error vs float64."""
import os
import sys

import pytest
import torch

from oracle import simt_oracle as so
from simt_amd.engine_v3 import V3Plan, v3_state_shapes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def make_state(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    st = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            st[k] = torch.randn(shp, generator=g) * 0.1
        elif k.endswith("running_var"):
            st[k] = torch.rand(shp, generator=g) * 0.5 + 0.75
        elif len(shp) == 1 and ".bn" in k or "downsample.1" in k:
            st[k] = (torch.rand(shp, generator=g) * 0.5 + 0.75) if k.endswith("weight") else torch.randn(shp, generator=g) * 0.1
        elif k.endswith("bias"):
            st[k] = torch.randn(shp, generator=g) * 0.05
        else:
            fan = shp[1] * shp[2] * shp[3]
            st[k] = torch.randn(shp, generator=g) * (2.0 / fan) ** 0.5
    return st


def run_oracle(st, img, up, layers, openset, train, dtype, acts=None):
    s = {k: (v.clone().to(dtype) if v.is_floating_point() else v.clone()) for k, v in st.items()}
    for k, v in s.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    out = so.v3_forward(s, img.to(dtype), layers=layers, openset=openset, train=train, acts=acts)
    if up is not None:
        (out * up.to(dtype)).sum().backward()
    return out.detach(), s


def relu_flips(plan, acts, B):
    """Number of post-ReLU elements whose sign pattern differs between the GPU plan and the float64 oracle."""
    def nchw(t, h, w):
        return t.view(B, h, w, -1).permute(0, 3, 1, 2).cpu()
    n = 0
    for rec in plan.block_io:
        for key, hw in (("a1", ("Hi", "Wi")), ("a2", ("Ho", "Wo")), ("z", ("Ho", "Wo"))):
            n += ((nchw(rec[key], rec[hw[0]], rec[hw[1]]) > 0) != (acts[f"{rec['name']}.{key}"] > 0)).sum().item()
    h, w = plan.feat_hw
    for t in range(5):
        n += ((nchw(plan.acat[t], h, w) > 0) != (acts[f"assp.a{t + 1}"] > 0)).sum().item()
    n += ((nchw(plan.af, h, w) > 0) != (acts["assp.af"] > 0)).sum().item()
    return n


@pytest.mark.parametrize("openset", [False, True])
def test_v3_eval_forward(dev, openset):
    layers, width, ac, nc, oc = (1, 2, 2), 32, 32, 19, 6
    st = make_state(v3_state_shapes(nc, oc, openset, layers, width, ac), 3)
    B, H, W = 2, 96, 128
    img = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(4))
    p = {k: v.clone().to(dev) for k, v in st.items()}
    plan = V3Plan(p, B, H, W, nc, oc, openset, dtype=torch.float32, train=False, layers=layers, width=width, assp_ch=ac)
    got = plan.forward(img.to(dev)).cpu()
    ref, _ = run_oracle(st, img, None, layers, openset, False, torch.float64)
    assert got.shape == ref.shape == (B, nc + (oc if openset else 0), H, W)
    assert rel(got, ref) < 2e-5


def test_v3_train_forward_backward_fp32(dev):
    """Strict parity (every gradient within max(5x the fp32-CPU error, 2e-4) of float64) is demanded on the seeds where all
    ReLU masks agree with the float64 oracle.  An element whose pre-activation sits within fp32 noise of zero flips its mask
    -- a discrete event that, at 120 samples per BN channel, moves every upstream gradient by percents in EITHER fp32
    implementation (tests/test_gpu_trunk.py documents the same for the reference's own fp32 path); such a seed is held to a
    relative-L2 bound only, and at least one of the three (fixed, deterministic) seeds must be flip-free."""
    layers, width, ac, nc, oc = (1, 2, 2), 32, 32, 19, 6
    B, H, W = 2, 96, 160
    strict = 0
    for seed in (5, 15, 25):
        st = make_state(v3_state_shapes(nc, oc, True, layers, width, ac), seed)
        g = torch.Generator().manual_seed(seed + 1)
        img = torch.randn(B, 3, H, W, generator=g)
        up = torch.randn(B, nc + oc, H, W, generator=g) / (H * W)
        p = {k: v.clone().to(dev) for k, v in st.items()}
        plan = V3Plan(p, B, H, W, nc, oc, True, dtype=torch.float32, train=True, layers=layers, width=width, assp_ch=ac)
        got = plan.forward(img.to(dev)).cpu()
        grads = plan.backward(up.to(dev))
        torch.cuda.synchronize()
        acts = {}
        ref64, s64 = run_oracle(st, img, up, layers, True, True, torch.float64, acts)
        ref32, s32 = run_oracle(st, img, up, layers, True, True, torch.float32)
        e_gpu, e_cpu = rel(got, ref64), rel(ref32, ref64)
        assert e_gpu < max(5 * e_cpu, 2e-5), (e_gpu, e_cpu)
        # running statistics were updated in place like nn.BatchNorm2d does
        for k in ("resnet.resnet_50.bn1.running_mean", "assp.bnf.running_var", "resnet.resnet_50.layer2.0.downsample.1.running_var"):
            assert rel(p[k], s64[k]) < 1e-4, k
        assert set(grads) == {k for k, v in s64.items() if v.requires_grad}
        flips = relu_flips(plan, acts, B)
        print(f"seed {seed}: {flips} ReLU mask flips vs float64")
        for n, gt in grads.items():
            if flips == 0:
                e_g, e_c = rel(gt, s64[n].grad), rel(s32[n].grad, s64[n].grad)
                assert e_g < max(5 * e_c, 2e-4), f"seed {seed} {n}: gpu {e_g:.3e} cpu-fp32 {e_c:.3e}"
            else:
                l2 = ((gt.double().cpu() - s64[n].grad).norm() / s64[n].grad.norm()).item()
                assert l2 < 0.25, f"seed {seed} {n}: relative L2 {l2:.3e} with {flips} mask flips"
        strict += flips == 0
    assert strict >= 1, "no flip-free seed"


class _RoundBf16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.bfloat16().to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().to(g.dtype)


def test_v3_train_bf16_direction(dev, monkeypatch):
    """bf16 throughput mode.  The yardstick is the float64 oracle with every conv operand, conv output and the gradients
    flowing through them rounded to bf16 (what storing activations in bf16 does, whatever the kernel): on this small
    train-mode-BN net that alone moves gradient directions to cos ~0.9 of the exact ones.  The GPU must not be worse than
    that model by more than 0.06 on any tensor and never below 0.8."""
    layers, width, ac, nc = (1, 1, 2), 64, 64, 19
    st = make_state(v3_state_shapes(nc, 0, False, layers, width, ac), 7)
    B, H, W = 2, 128, 128
    g = torch.Generator().manual_seed(8)
    img = torch.randn(B, 3, H, W, generator=g)
    up = torch.nn.functional.interpolate(torch.randn(B, nc, 8, 8, generator=g), size=(H, W), mode="bilinear") / (H * W)
    p = {k: v.clone().to(dev) for k, v in st.items()}
    plan = V3Plan(p, B, H, W, nc, dtype=torch.bfloat16, train=True, layers=layers, width=width, assp_ch=ac)
    got = plan.forward(img.to(dev)).cpu()
    grads = plan.backward(up.to(dev))
    torch.cuda.synchronize()
    ref, s = run_oracle(st, img, up, layers, False, True, torch.float64)
    conv = so.F.conv2d
    monkeypatch.setattr(so.F, "conv2d", lambda x, w, *a, **k: _RoundBf16.apply(conv(_RoundBf16.apply(x), _RoundBf16.apply(w), *a, **k)))
    refq, sq = run_oracle(st, img, up, layers, False, True, torch.float64)
    monkeypatch.undo()
    assert rel(got, ref) < max(2 * rel(refq, ref), 3e-2)

    def cos(a, b):
        return torch.nn.functional.cosine_similarity(a.double().cpu().flatten(), b.double().flatten(), dim=0).item()
    for n, gt in grads.items():
        c_gpu, c_sim = cos(gt, s[n].grad), cos(sq[n].grad, s[n].grad)
        assert c_gpu > max(c_sim - 0.06, 0.8), f"{n}: gpu cosine {c_gpu:.4f}, bf16-rounded float64 model {c_sim:.4f}"


def test_deeplabv3_module_surface_and_autograd(dev):
    sys.path.insert(0, os.path.join(ROOT, "simt_amd"))
    from model.deeplabv3 import DeepLabv3, sig_NTM, sig_W
    m = DeepLabv3(19, openc=6, openset=True)
    keys = list(m.state_dict().keys())
    used = v3_state_shapes(19, 6, True)
    assert [k for k in keys if k in used] == list(used.keys())                   # same names, same order
    assert {tuple(m.state_dict()[k].shape) == tuple(s) for k, s in used.items()} == {True}
    assert "resnet.resnet_50.layer4.2.conv3.weight" in keys and "resnet.resnet_50.fc.bias" in keys   # present, never run

    class A:
        learning_rate = 2.5e-4
    groups = m.optim_parameters(A)
    g0, g1 = list(groups[0]["params"]), list(groups[1]["params"])
    named = dict(m.named_parameters())
    want0 = [p for n, p in named.items() if any(s in n for s in ("resnet_50.layer3", "resnet_50.layer4", "resnet_50.fc"))]
    assert len(g0) == len(want0) and all(a is b for a, b in zip(g0, want0))      # each once (no duplicate listing here)
    assert len(g1) == len(list(m.assp.parameters())) + 4 and groups[1]["lr"] == 10 * A.learning_rate
    m = m.to(dev)
    x = torch.randn(1, 3, 64, 96, device=dev)
    m.train()
    y = m(x)
    assert y.shape == (1, 25, 64, 96) and torch.isfinite(y).all()
    y.square().mean().backward()
    for n in ("resnet.resnet_50.conv1.weight", "resnet.resnet_50.bn1.weight", "assp.bnf.bias", "conv_1.bias", "assp.convf.weight"):
        assert named[n].grad is not None and torch.isfinite(named[n].grad).all() and named[n].grad.abs().sum() > 0, n
    assert named["resnet.resnet_50.layer4.0.conv1.weight"].grad is None
    assert int(m.assp.bnf.num_batches_tracked) == 1 and int(m.resnet.resnet_50.layer4[0].bn1.num_batches_tracked) == 0
    m.eval()
    with torch.no_grad():
        ye = m(x)
    assert ye.shape == y.shape and torch.isfinite(ye).all()
    T = sig_NTM(19, 6).to(dev)()
    assert T.shape == (25, 19) and torch.allclose(T.sum(1), torch.ones(25, device=dev), atol=1e-5)
    assert sig_W(19, 6).to(dev)().shape == (25, 25)
