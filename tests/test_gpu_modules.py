"""GPU parity of the drop-in modules (model.deeplab_multi.{DeeplabMulti, sig_NTM, sig_W}, model.deeplab.Res_Deeplab,
utils.loss.{CrossEntropy2d, EntropyLoss}) used the way tools/trainV2_simt.py uses them: torch autograd + torch.optim on
top, HIP kernels underneath.  Checker: the CPU oracle / torch CPU fp32 on the same inputs; golden g1_g2 for NTM/W.
fp32 parity mode; tolerances as in tests/test_gpu_trunk.py (small trunk: logits 1e-4, gradients 3e-2 of max|ref|
because a ReLU-mask flip may hit one low-variance BN channel, see there)."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import simt_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simt_amd"))
pytestmark = pytest.mark.gpu
G = os.path.join(ROOT, "tests", "golden")
CD = so.load_class_dist()


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def small_model(dev):
    from model.deeplab_multi import Bottleneck, ResNetMulti
    layers = (1, 1, 2, 1)
    m = ResNetMulti(Bottleneck, list(layers), 19, 3, True)
    st = so.recipe_state(so.state_shapes(19, 3, True, layers=layers), seed=21)
    m.load_state_dict(st)
    m.compute_dtype = torch.float32
    return m.to(dev), st, layers


def test_deeplab_multi_module_autograd_and_optimizer(dev):
    m, st, layers = small_model(dev)
    g = torch.Generator().manual_seed(2)
    img = torch.randn(2, 3, 81, 97, generator=g) * 50
    m.train()
    x1, x2 = m(img.to(dev))
    stg = {k: (v.clone().requires_grad_(True) if (("conv" in k or "downsample.0" in k) and v.dim() > 0) else v.clone())
           for k, v in st.items()}
    r1, r2 = so.deeplab_multi_forward(stg, img, True, True, layers=layers)
    assert x1.shape == r1.shape and rel(x1, r1.detach()) < 1e-4 and rel(x2, r2.detach()) < 1e-4
    up1, up2 = torch.randn(r1.shape, generator=g), torch.randn(r2.shape, generator=g)
    ((x1 * up1.to(dev)).sum() + (x2 * up2.to(dev)).sum()).backward()
    ((r1 * up1).sum() + (r2 * up2).sum()).backward()
    n_grad = 0
    for n, p in m.named_parameters():
        if stg[n].grad is None:
            assert p.grad is None, n                    # frozen BN affine and the dead 18/24 branches (quirks 1, 5)
        else:
            n_grad += 1
            assert rel(p.grad, stg[n].grad) < 3e-2, n
    assert n_grad == len([k for k in stg if stg[k].grad is not None])
    # BN running statistics and the batch counter moved like torch's
    assert rel(m.layer3[0].bn2.running_var, stg["layer3.0.bn2.running_var"]) < 1e-4
    assert int(m.bn1.num_batches_tracked) == 1
    # optimiser on the reference's duplicate listing, then a second forward sees the updated weights
    args = type("A", (), {"learning_rate": 1e-3})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt = torch.optim.SGD(m.optim_parameters(args), lr=1e-3, momentum=0.9, weight_decay=5e-4, foreach=False)
    before = m.layer4[0].conv2.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.layer4[0].conv2.weight.detach())
    m.eval()
    e1, e2 = m(img.to(dev))
    st2 = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    o1, o2 = so.deeplab_multi_forward(st2, img, False, True, layers=layers)
    assert rel(e1, o1) < 2e-5 and rel(e2, o2) < 2e-5


def test_sig_ntm_sig_w_modules_with_autograd(dev):
    from model.deeplab_multi import sig_NTM, sig_W
    d = np.load(os.path.join(G, "g1_g2_ntm_w.npz"))
    for K in (3, 15):
        ntm = sig_NTM(19, K).to(dev)
        ntm.NTM.data.copy_(torch.as_tensor(d[f"ntm_{K}"]))
        T = ntm()
        assert rel(T, torch.as_tensor(d[f"T_{K}"])) < 1e-6
        (T * torch.as_tensor(d[f"dT_{K}"]).to(dev)).sum().backward()
        assert rel(ntm.NTM.grad, torch.as_tensor(d[f"dntm_{K}"])) < 2e-6
        w = sig_W(19, K).to(dev)
        w.weight.data.copy_(torch.as_tensor(d[f"w_{K}"]))
        Wm = w()
        assert rel(Wm, torch.as_tensor(d[f"W_{K}"])) < 1e-6
        assert rel(w.weight.detach(), torch.as_tensor(d[f"w_after_{K}"])) == 0
        (Wm * torch.as_tensor(d[f"dW_{K}"]).to(dev)).sum().backward()
        assert rel(w.weight.grad, torch.as_tensor(d[f"dw_{K}"])) < 2e-6
    # the inner-loop usage pattern of the reference: loss = ||W T||^2, Adam on W, gradient leaks into NTM
    ntm, w = sig_NTM(19, 3).to(dev), sig_W(19, 3).to(dev)
    opt = torch.optim.Adam(w.parameters(), lr=6e-3)
    for _ in range(2):
        opt.zero_grad()
        loss = (w().mm(ntm()) ** 2).sum()
        loss.backward()
        opt.step()
    assert ntm.NTM.grad is not None and torch.isfinite(loss)


@pytest.mark.parametrize("is_softmax", [True, False])
@pytest.mark.parametrize("with_weight", [False, True])
def test_cross_entropy_2d(dev, is_softmax, with_weight):
    from utils.loss import CrossEntropy2d
    g = torch.Generator().manual_seed(3)
    n, c, h, w = 2, 19, 37, 41
    x = torch.randn(n, c, h, w, generator=g) * 2
    pred = x if is_softmax else torch.softmax(x, 1)
    tgt = torch.randint(0, c, (n, h, w), generator=g)
    tgt[torch.rand(n, h, w, generator=g) < 0.2] = 255
    wt = (torch.rand(c, generator=g) + 0.5) if with_weight else None
    pr = pred.clone().requires_grad_(True)
    mask = tgt != 255
    flat = pr.permute(0, 2, 3, 1)[mask]
    ref = F.cross_entropy(flat, tgt[mask], weight=wt) if is_softmax else F.nll_loss(torch.log(flat), tgt[mask], weight=wt)
    (ref * 1.7).backward()
    pd = pred.clone().to(dev).requires_grad_(True)
    crit = CrossEntropy2d(is_softmax=is_softmax)
    loss = crit(pd, tgt.to(dev), weight=None if wt is None else wt.to(dev))
    (loss * 1.7).backward()
    assert abs(loss.item() - ref.item()) < 1e-5 * (1 + abs(ref.item()))
    assert rel(pd.grad, pr.grad) < 1e-5
    # no valid pixel -> NaN
    assert torch.isnan(crit(pd.detach(), torch.full((n, h, w), 255, device=dev, dtype=torch.long)))
    with pytest.raises(AssertionError):
        crit(pd.detach()[0], tgt.to(dev))


def test_entropy_loss(dev):
    from utils.loss import EntropyLoss
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(2, 19, 23, 29, generator=g) * 3).requires_grad_(True)
    ref = (-(F.softmax(x, 1) * F.log_softmax(x, 1)).sum(1)).mean()
    ref.backward()
    xd = x.detach().clone().to(dev).requires_grad_(True)
    out = EntropyLoss()(xd)
    out.backward()
    assert abs(out.item() - ref.item()) < 1e-6 * (1 + abs(ref.item()))
    assert rel(xd.grad, x.grad) < 1e-5


def test_res_deeplab_single_head_module(dev):
    from model.deeplab import ResNet
    from model.deeplab_multi import Bottleneck
    layers = (1, 1, 1, 1)
    m = ResNet(Bottleneck, list(layers), 19)
    st = so.recipe_state(so.state_shapes(19, single_head=True, layers=layers), seed=5)
    m.load_state_dict(st)
    m.compute_dtype = torch.float32
    m = m.to(dev).train()
    g = torch.Generator().manual_seed(4)
    img = torch.randn(1, 3, 65, 81, generator=g) * 50
    y, y2 = m(img.to(dev))
    stg = {k: (v.clone().requires_grad_(True) if "conv2d_list" in k else v.clone()) for k, v in st.items()}
    r, _ = so.deeplab_single_forward(stg, img, True, layers=layers)
    assert y is y2 or torch.equal(y, y2)
    assert rel(y, r.detach()) < 1e-4
    up = torch.randn(r.shape, generator=g)
    (y * up.to(dev)).sum().backward()
    (r * up).sum().backward()
    for i in range(4):                                   # all four branches are live in deeplab.py
        assert rel(m.layer5.conv2d_list[i].weight.grad, stg[f"layer5.conv2d_list.{i}.weight"].grad) < 1e-3
        assert rel(m.layer5.conv2d_list[i].bias.grad, stg[f"layer5.conv2d_list.{i}.bias"].grad) < 1e-4


def test_cross_entropy_out_of_range_label_is_loud(dev):
    """A target in [C, 254] is an error in the reference (torch raises "Target ... is out of bounds"); here it must neither index out
    of bounds nor pass silently: the loss comes back NaN and the gradient stays finite (csrc/loss2d.hip ce_oob)."""
    from simt_amd.utils.loss import CrossEntropy2d
    g = torch.Generator().manual_seed(1)
    pred = torch.randn(2, 19, 9, 11, generator=g).to(dev).requires_grad_(True)
    tgt = torch.randint(0, 19, (2, 9, 11), generator=g)
    tgt[0, 0, 0] = 255
    ok = CrossEntropy2d()(pred, tgt.to(dev))
    assert torch.isfinite(ok)
    tgt[1, 3, 3] = 20
    bad = CrossEntropy2d()(pred, tgt.to(dev))
    assert torch.isnan(bad)
    pred.grad = None
    ok.backward()
    assert torch.isfinite(pred.grad).all()
