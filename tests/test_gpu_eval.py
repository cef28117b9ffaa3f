"""GPU parity of the evaluation path (reference tools/evaluate_cityscapes.py:81-162): two-scale upsample +
sum + arg-max, confusion histogram, mIoU -- against the reference's definition evaluated on the CPU (torch
F.interpolate align_corners=True + numpy argmax / bincount, i.e. the oracle's fast_hist / per_class_iu pinned by golden
g9).  Histogram and mIoU given identical predictions: bit-exact.  Arg-max from fp32 logits: exact except where the top
two summed logits are within 1e-5 of each other (the CPU may or may not contract a*b+c into an FMA)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import simt_oracle as so
from simt_amd import _lib as L
from simt_amd import ops
from simt_amd.tools.evaluate_cityscapes import Evaluator, fast_hist, per_class_iu

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_upsample_sum_argmax_and_histogram(dev):
    g = torch.Generator().manual_seed(3)
    B, C, H, W = 2, 19, 96, 160
    la = torch.randn(B, C, 13, 21, generator=g) * 3
    lb = torch.randn(B, C, 17, 26, generator=g) * 3
    ref_sum = (F.interpolate(la, size=(H, W), mode="bilinear", align_corners=True).numpy() +
               F.interpolate(lb, size=(H, W), mode="bilinear", align_corners=True).numpy())
    ref = np.argmax(ref_sum.transpose(0, 2, 3, 1), axis=3)

    def nhwc(t, ld=32):
        o = torch.zeros(t.shape[0], t.shape[2], t.shape[3], ld)
        o[..., :t.shape[1]] = t.permute(0, 2, 3, 1)
        return o.to(dev)
    a_d, b_d = nhwc(la), nhwc(lb)
    pred = torch.full((B, H, W), -1, device=dev, dtype=torch.int32)
    L.call("simt_upsample_sum_argmax", ops._p(a_d), 13, 21, 32, ops._p(b_d), 17, 26, 32, B, H, W, C, ops._p(pred), ops.stream_ptr())
    got = pred.cpu().numpy()
    srt = np.sort(ref_sum, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    diff = got != ref
    assert diff.sum() <= 2 and np.all(margin[diff] < 5e-5), f"{diff.sum()} mismatches, margins {margin[diff]}"
    # single scale
    L.call("simt_upsample_sum_argmax", ops._p(a_d), 13, 21, 32, None, 0, 0, 0, B, H, W, C, ops._p(pred), ops.stream_ptr())
    ref1 = F.interpolate(la, size=(H, W), mode="bilinear", align_corners=True).argmax(1).numpy()
    assert (pred.cpu().numpy() != ref1).sum() <= 2
    # confusion histogram: exact, ignores labels outside [0, n), accumulates across calls
    gt = torch.randint(0, C, (B, H, W), generator=g)
    gt[torch.rand(B, H, W, generator=g) < 0.1] = 255
    hist = torch.zeros(C * C, device=dev, dtype=torch.int64)
    for _ in range(2):
        L.call("simt_confusion_hist", ops._p(gt.to(dev)), ops._p(pred), gt.numel(), C, ops._p(hist), ops.stream_ptr())
    exp = 2 * fast_hist(gt.numpy().flatten(), pred.cpu().numpy().flatten().astype(np.int64), C)
    assert np.array_equal(hist.cpu().numpy().reshape(C, C), exp)


def test_metric_helpers_match_reference_golden():
    d = np.load(os.path.join(G, "g9_metric.npz"))
    h = fast_hist(d["gt"], d["pred"], 19)
    assert np.array_equal(h, d["hist"])
    np.testing.assert_array_equal(per_class_iu(h), d["iu"])
    assert round(float(np.nanmean(per_class_iu(h))) * 100, 2) == float(d["miou"])


@pytest.mark.parametrize("K", [3, 0])
def test_evaluator_end_to_end_small(dev, K):
    """Two eval plans (two input scales) + fused predict + histogram vs the oracle on the CPU, fp32, small trunk.
    K = 3: evaluate_simt (open-set heads present, logits[:, :19] scored, evaluate_cityscapes.py:96-162); K = 0: evaluate_warmup
    (:165-225, a warm-up checkpoint without open-set heads)."""
    layers = (1, 1, 2, 1)
    st = so.recipe_state(so.state_shapes(19, K, K > 0, layers=layers), seed=31, head_scale=8.0)
    g = torch.Generator().manual_seed(9)
    B, (H, W) = 1, (64, 96)
    s1, s2 = (33, 49), (41, 61)
    img1 = torch.randn(B, 3, *s1, generator=g) * 50
    img2 = F.interpolate(img1, size=s2, mode="bilinear", align_corners=True)
    gt = torch.randint(0, 19, (B, H, W), generator=g)
    ev = Evaluator(st, num_classes=19, open_classes=K, batch=B, label_hw=(H, W), scales=(s1, s2), dtype=torch.float32,
                   device=dev, layers=layers)
    ev.add(img1, img2, gt)
    miou, ius = ev.result()
    _, o1 = so.deeplab_multi_forward(st, img1, False, K > 0, layers=layers)
    _, o2 = so.deeplab_multi_forward(st, img2, False, K > 0, layers=layers)
    out = (F.interpolate(o1[:, :19], size=(H, W), mode="bilinear", align_corners=True).numpy() +
           F.interpolate(o2[:, :19], size=(H, W), mode="bilinear", align_corners=True).numpy())
    pred = np.argmax(out.transpose(0, 2, 3, 1), axis=3)
    got = ev.pred.cpu().numpy()
    # "arg-max bit-exact" (north_star), stated with its margin: the summed logits carry the fp32 conv parity error (2e-5 of max|logit|
    # per layer, tests/test_gpu_conv.py), so a pixel may differ ONLY where the reference's own top-2 gap is below that error budget;
    # everywhere else the label map is bit-exact.  Margin = 1e-4 * max|summed logit| (the measured logit error is ~3e-5 of it).
    top2 = np.sort(out.transpose(0, 2, 3, 1), axis=3)[..., -2:]
    gap = top2[..., 1] - top2[..., 0]
    margin = 1e-4 * np.abs(out).max()
    diff = got != pred
    print(f"K={K}: {int(diff.sum())} of {diff.size} labels differ; {int((gap < margin).sum())} pixels with a top-2 gap below {margin:.2e}")
    assert not np.any(diff & (gap >= margin)), f"{int((diff & (gap >= margin)).sum())} labels differ outside the rounding margin"
    assert (gap < margin).mean() < 5e-3          # the statement is not vacuous: fewer than 0.5 % of the pixels are exempt
    h = fast_hist(gt.numpy().flatten(), got.flatten().astype(np.int64), 19)
    assert np.array_equal(ev.hist.cpu().numpy().reshape(19, 19), h)
    assert miou == round(float(np.nanmean(per_class_iu(h))) * 100, 2)


@pytest.mark.parametrize("align", [0, 1])
@pytest.mark.parametrize("shape", [(2, 7, 9, 25, 32, 56, 72), (1, 32, 64, 19, 24, 512, 1024)])
def test_upsample_nchw_forward_and_adjoint(dev, align, shape):
    """simt_upsample_nchw / _bwd (DeepLabv3's in-model F.interpolate, model/deeplabv3.py:137, and the align_corners=True flavour
    of trainV2_simt.py:301) against torch's bilinear interpolate and its autograd adjoint: 1e-5 of max|ref| (fp32)."""
    import torch.nn.functional as F
    from simt_amd import _lib as L, ops
    B, h, w, C, lds, H, W = shape
    g = torch.Generator().manual_seed(h * w + align)
    src = torch.zeros(B, h, w, lds)
    src[..., :C] = torch.randn(B, h, w, C, generator=g)
    src_d = src.to(dev)
    dst = torch.empty(B, C, H, W, device=dev)
    L.call("simt_upsample_nchw", src_d.data_ptr(), B, h, w, lds, C, H, W, align, dst.data_ptr(), ops.stream_ptr())
    x = src[..., :C].permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    ref = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=bool(align))
    assert (dst.cpu() - ref.detach()).abs().max().item() < 1e-5 * ref.abs().max().item()
    up = torch.randn(B, C, H, W, generator=g)
    (ref * up).sum().backward()
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1e-2)):
        dsrc = torch.zeros(B, h, w, lds, device=dev, dtype=dt)
        tmp = torch.empty(B, C, H, w, device=dev)                     # caller-owned scratch of the separable adjoint
        L.call("simt_upsample_nchw_bwd", up.to(dev).data_ptr(), B, h, w, lds, C, H, W, align, dsrc.data_ptr(), ops.dt_code(dt),
               tmp.data_ptr(), ops.stream_ptr())
        got = dsrc[..., :C].float().cpu().permute(0, 3, 1, 2)
        assert (got - x.grad).abs().max().item() < tol * x.grad.abs().max().item()
        assert (dsrc[..., C:] == 0).all()


def test_upsample_sum_argmax_at_cityscapes_resolution(dev):
    """The evaluation kernel at the reference's geometry: logits 65 x 129 (1024 x 512 input) + 81 x 161 (1280 x 640) -> 1024 x 2048
    labels, against torch CPU; mismatches only where the two largest summed logits are closer than 1e-5 (FMA contraction)."""
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 8)))
    g = torch.Generator().manual_seed(12)
    B, C, H, W = 1, 19, 1024, 2048
    la = torch.randn(B, C, 65, 129, generator=g) * 3
    lb = torch.randn(B, C, 81, 161, generator=g) * 3
    ref_sum = (F.interpolate(la, size=(H, W), mode="bilinear", align_corners=True).numpy() +
               F.interpolate(lb, size=(H, W), mode="bilinear", align_corners=True).numpy())
    ref = np.argmax(ref_sum.transpose(0, 2, 3, 1), axis=3)

    def nhwc(t, ld=32):
        o = torch.zeros(t.shape[0], t.shape[2], t.shape[3], ld)
        o[..., :t.shape[1]] = t.permute(0, 2, 3, 1)
        return o.to(dev)
    a_d, b_d = nhwc(la), nhwc(lb)
    pred = torch.full((B, H, W), -1, device=dev, dtype=torch.int32)
    L.call("simt_upsample_sum_argmax", ops._p(a_d), 65, 129, 32, ops._p(b_d), 81, 161, 32, B, H, W, C, ops._p(pred), ops.stream_ptr())
    got = pred.cpu().numpy()
    srt = np.sort(ref_sum, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    diff = got != ref
    # arg-max of fp32 sums: only near-ties may differ (margin below ~40 ulp of sums of magnitude 10: the two bilinear terms are rounded in a
    # different order than torch's; which way a tie falls changes with the compiler's fma contraction)
    assert diff.sum() <= 20 and np.all(margin[diff] < 5e-5), f"{diff.sum()} mismatches of {got.size}, margins {margin[diff][:5]}"
    gt = torch.randint(0, C, (B, H, W), generator=g)
    gt[torch.rand(B, H, W, generator=g) < 0.1] = 255
    hist = torch.zeros(C * C, device=dev, dtype=torch.int64)
    L.call("simt_confusion_hist", ops._p(gt.to(dev)), ops._p(pred), gt.numel(), C, ops._p(hist), ops.stream_ptr())
    assert np.array_equal(hist.cpu().numpy().reshape(C, C), fast_hist(gt.numpy().flatten(), got.flatten().astype(np.int64), C))


def test_evaluator_full_depth_r101_fp32_at_reference_geometry(dev):
    """the metric north_star names ("mIoU/argmax bit-exact"), end to end, at FULL depth and at the reference's geometry
    (evaluate_cityscapes.py:96-162): ResNet-101 two-head model with open-set heads, one frame at 1024 x 512 and 1280 x 640 -> logits[:, :19]
    upsampled to 1024 x 2048, summed, arg-maxed -> fast_hist -> mIoU; the DEFAULT Evaluator (fp32 plans) against the oracle's forward x 2 +
    numpy arg-max on the CPU.  Weights: the checkpoint-like recipe (ms.trained_like_init, head_scale 8) with BatchNorm running statistics
    calibrated by train-mode forwards of the HIP trunk and handed to BOTH sides (with uncalibrated statistics 101 eval-mode layers collapse the
    features and every pixel gets the same label: a vacuous comparison).  Bar: labels equal wherever the oracle's own top-2 gap of the summed
    logits exceeds 1e-4 of max|summed logit| (the fp32 conv parity error through 101 layers is ~3e-5 of it); < 0.5 % of the pixels exempt;
    the device histogram equals fast_hist of the device labels; mIoU (2 decimals) equals the oracle's within one unit of the last decimal."""
    from simt_amd import model_spec as ms
    from simt_amd.engine import TrunkPlan, multi_heads
    K = 3
    cd = ms.load_class_dist("bapa")
    st = ms.trained_like_init(ms.state_shapes(19, K, True), seed=1234)
    p = {k: v.clone().to(dev) for k, v in st.items()}
    cal = TrunkPlan(p, 1, 384, 768, multi_heads(19, K, True), dtype=torch.float32, train=True)
    img_c, _ = ms.synthetic_batch(1, 384, 768, cd, seed=99, device=dev)
    cal.x_in.copy_(img_c)
    for _ in range(40):
        cal.fwd_list.run()
    torch.cuda.synchronize()
    for k in st:
        if k.endswith("running_mean") or k.endswith("running_var"):
            st[k] = p[k].detach().cpu().clone()
    del cal, p
    torch.cuda.empty_cache()
    H, W = 1024, 2048
    s1, s2 = (512, 1024), (640, 1280)
    # a frame with structure at every scale (piecewise-constant blocks + noise, BGR - mean range), resized like the loader does (bilinear
    # stands in for the bicubic resize: both sides get the identical tensors)
    g = torch.Generator().manual_seed(77)
    base = torch.randn(1, 3, 32, 64, generator=g) * 60
    full = F.interpolate(base, size=(H, W), mode="nearest") + torch.randn(1, 3, H, W, generator=g) * 12
    img1 = F.interpolate(full, size=s1, mode="bilinear", align_corners=False).contiguous()
    img2 = F.interpolate(full, size=s2, mode="bilinear", align_corners=False).contiguous()
    gt = torch.randint(0, 19, (1, H, W), generator=g)
    gt[torch.rand(1, H, W, generator=g) < 0.1] = 255
    ev = Evaluator(st, num_classes=19, open_classes=K, device=dev)             # every default: fp32 plans, the reference's geometry
    assert ev.dtype == torch.float32 and (ev.H, ev.W) == (H, W)
    ev.add(img1, img2, gt)
    miou, ius = ev.result()
    got = ev.pred.cpu().numpy()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))
    with torch.no_grad():
        _, o1 = so.deeplab_multi_forward(st, img1, False, True)
        _, o2 = so.deeplab_multi_forward(st, img2, False, True)
        out = (F.interpolate(o1[:, :19], size=(H, W), mode="bilinear", align_corners=True) +
               F.interpolate(o2[:, :19], size=(H, W), mode="bilinear", align_corners=True))
        top2 = out.topk(2, dim=1)
    pred = top2.indices[:, 0].numpy()
    gap = (top2.values[:, 0] - top2.values[:, 1]).numpy()
    margin = 1e-4 * float(out.abs().max())
    diff = got != pred
    nlab = len(np.unique(pred))
    print(f"full-depth fp32 evaluator: {int(diff.sum())} of {diff.size} labels differ; {int((gap < margin).sum())} pixels with a top-2 gap below "
          f"{margin:.2e}; {nlab} distinct labels in the oracle's map")
    assert nlab >= 4, "degenerate label map: the comparison would be vacuous"
    assert not np.any(diff & (gap >= margin)), f"{int((diff & (gap >= margin)).sum())} labels differ outside the rounding margin"
    assert (gap < margin).mean() < 5e-3
    h = fast_hist(gt.numpy().flatten(), got.flatten().astype(np.int64), 19)
    assert np.array_equal(ev.hist.cpu().numpy().reshape(19, 19), h)
    assert miou == round(float(np.nanmean(per_class_iu(h))) * 100, 2)
    h_ref = fast_hist(gt.numpy().flatten(), pred.flatten().astype(np.int64), 19)
    miou_ref = round(float(np.nanmean(per_class_iu(h_ref))) * 100, 2)
    print(f"mIoU: device {miou}, oracle {miou_ref}")
    assert abs(miou - miou_ref) <= 0.011
