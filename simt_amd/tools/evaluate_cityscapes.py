"""Evaluation of a SimT / DeepLab-v2 model on MI355X: the reference's tools/evaluate_cityscapes.py (evaluate_simt
:96-162, fast_hist :81-83, per_class_iu :86-87, label_mapping :90-94) with the per-image CPU/numpy work moved to the GPU.

    ev = Evaluator(state_dict, num_classes=19, open_classes=K, label_hw=(1024, 2048), scales=((512, 1024), (640, 1280)))
    for image_a, image_b, gt in loader:            # the image at both input scales, ground-truth label ids [B,H,W] int64
        ev.add(image_a, image_b, gt)
    miou, per_class = ev.result()

Two eval-mode plans (BN folded) produce the main head's logits at both scales; one fused kernel upsamples both to the
label resolution (align_corners=True), sums, arg-maxes; the confusion histogram is accumulated on the device with
integer atomics.  Dataset IO (PIL decoding, file lists) stays outside: `add` takes tensors."""
import numpy as np
import torch

from simt_amd import _lib as L
from simt_amd import ops
from simt_amd.engine import TrunkPlan, multi_heads


def fast_hist(a, b, n):
    k = (a >= 0) & (a < n)
    return np.bincount(n * a[k].astype(int) + b[k], minlength=n ** 2).reshape(n, n)


def per_class_iu(hist):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))


def label_mapping(inp, mapping):
    out = np.copy(inp)
    for ind in range(len(mapping)):
        out[inp == mapping[ind][0]] = mapping[ind][1]
    return np.array(out, dtype=np.int64)


class Evaluator:
    def __init__(self, state, *, num_classes=19, open_classes=0, openset=None, batch=1, label_hw=(1024, 2048),
                 scales=((512, 1024), (640, 1280)), dtype=torch.bfloat16, device="cuda:0", layers=None):
        self.dev = torch.device(device)
        self.C = num_classes
        openset = (open_classes > 0) if openset is None else openset
        params = {k: v.detach().to(self.dev, torch.float32 if v.dtype != torch.long else torch.long).clone() for k, v in state.items()}
        kw = {"layers": layers} if layers is not None else {}
        self.plans = [TrunkPlan(params, batch, h, w, multi_heads(num_classes, open_classes, openset), dtype=dtype, train=False, **kw)
                      for (h, w) in scales]
        self.B, (self.H, self.W) = batch, label_hw
        self.pred = torch.zeros(batch, self.H, self.W, device=self.dev, dtype=torch.int32)
        self.hist = torch.zeros(num_classes * num_classes, device=self.dev, dtype=torch.int64)

    def predict(self, *images):
        """images: one [B,3,h,w] fp32 tensor per scale.  Returns the arg-max label map [B,H,W] int32 (device)."""
        outs = []
        for plan, img in zip(self.plans, images):
            o = plan.forward(img.to(self.dev))["x2"]
            outs.append((o, o.shape[1], o.shape[2], o.shape[3]))
        (la, ha, wa, lda) = outs[0]
        lb, hb, wb, ldb = (outs[1] if len(outs) > 1 else (None, 0, 0, 0))
        L.call("simt_upsample_sum_argmax", ops._p(la), ha, wa, lda, ops._p(lb), hb, wb, ldb, self.B, self.H, self.W, self.C,
               ops._p(self.pred), ops.stream_ptr())
        return self.pred

    def add(self, *images_and_gt):
        *images, gt = images_and_gt
        pred = self.predict(*images)
        gt = gt.to(self.dev).long().contiguous()
        L.call("simt_confusion_hist", ops._p(gt), ops._p(pred), gt.numel(), self.C, ops._p(self.hist), ops.stream_ptr())

    def result(self):
        hist = self.hist.cpu().numpy().reshape(self.C, self.C).astype(np.float64)
        ius = per_class_iu(hist)
        return round(float(np.nanmean(ius)) * 100, 2), ius
