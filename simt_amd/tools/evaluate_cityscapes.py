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
                 scales=((512, 1024), (640, 1280)), dtype=torch.float32, device="cuda:0", layers=None):
        # dtype: fp32 by default -- the reference evaluates in fp32 (evaluate_cityscapes.py:96-162) and the metric is defined "argmax bit-exact";
        # bf16 plans are an explicit, labelled opt-in (tools: --eval-dtype bf16; < 0.2 % of the arg-max positions differ)
        self.dev = torch.device(device)
        self.dtype = dtype
        self.C = num_classes
        openset = (open_classes > 0) if openset is None else openset
        params = {k: v.detach().to(self.dev, torch.float32 if v.dtype != torch.long else torch.long).clone() for k, v in state.items()}
        kw = {"layers": layers} if layers is not None else {}
        self.plans = [TrunkPlan(params, batch, h, w, multi_heads(num_classes, open_classes, openset), dtype=dtype, train=False, **kw)
                      for (h, w) in scales]
        self.B, (self.H, self.W) = batch, label_hw
        self.pred = torch.zeros(batch, self.H, self.W, device=self.dev, dtype=torch.int32)
        self.hist = torch.zeros(num_classes * num_classes, device=self.dev, dtype=torch.int64)

    def load(self, state):
        """Refresh the weights from a (training) state dict and re-fold / re-pack both plans: the in-loop evaluation of
        tools/trainV2_simt.py:452-456 reuses one Evaluator for the whole run."""
        for plan in self.plans:
            for k, v in plan.p.items():
                if k in state and v.dtype != torch.long:
                    v.copy_(state[k])
        for plan in self.plans:                   # packed / folded operands are per plan (per input size), also when plans share `p`
            plan.repack()
        self.hist.zero_()

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


def evaluate_simt(state, data_dir, data_list, gt_dir, devkit_dir="../dataset/cityscapes_list", *, num_classes=19, open_classes=0, set_name="val",
                  device="cuda:0", dtype=torch.float32, evaluator=None, rank=0, world=1, process_group=None, verbose=True, workers=4):
    """File-based evaluation loop of the reference (evaluate_cityscapes.py:96-162): every validation frame at crop sizes (1024, 512) and
    (1280, 640) -> logits[:, :num_classes] of the main head, upsampled to 1024 x 2048, summed, arg-maxed -> fast_hist against the
    ground-truth label ids mapped with info.json's label2train -> mIoU (round(nanmean * 100, 2)).
    dtype: fp32 like the reference; torch.bfloat16 is an opt-in whose mIoU is printed with a "(bf16 plans)" label.
    Host: file lists, PNG decoding (threads), the label LUT.  Device: both resizes (Pillow-exact), BGR - mean, both forwards, the fused
    upsample + sum + arg-max, the histogram.  Data parallel: ranks take strided shards of the list and the histogram is all-reduced."""
    import json
    from concurrent.futures import ThreadPoolExecutor
    from os.path import join

    from simt_amd.data.pipeline import IMG_MEAN, InputPrep
    from simt_amd.dataset.cityscapes_dataset import cityscapesDataSet
    from simt_amd.tools.ntm_stats import mapping_lut
    dev = torch.device(device)
    with open(join(devkit_dir, "info.json"), "r") as fp:
        info = json.load(fp)
    name_classes = info.get("label", [str(i) for i in range(num_classes)])
    lut = mapping_lut(np.array(info["label2train"]))
    ds = cityscapesDataSet(data_dir, data_list, crop_size=(1024, 512), mean=IMG_MEAN, scale=False, mirror=False, set=set_name)
    ev = evaluator or Evaluator(state, num_classes=num_classes, open_classes=open_classes, dtype=dtype, device=dev)
    if evaluator is not None:
        ev.load(state)
    from PIL import Image

    def fetch(i):
        rgb, _, name = ds.decode(i)
        gt_path = "%s/%s" % (gt_dir, name.split("leftImg8bit")[0] + "gtFine_labelIds.png")
        return rgb, lut[np.array(Image.open(gt_path))], name
    idx = list(range(rank, len(ds), world))
    preps = {}
    xa = xb = None
    with ThreadPoolExecutor(max(1, workers)) as pool:
        for rgb, label, name in pool.map(fetch, idx):
            key = rgb.shape[:2]
            if key not in preps:
                preps[key] = (InputPrep(1, key, (1024, 512), dev, with_label=False), InputPrep(1, key, (1280, 640), dev, with_label=False))
                xa, xb = torch.empty(1, 3, 512, 1024, device=dev), torch.empty(1, 3, 640, 1280, device=dev)
            if label.size != ev.H * ev.W:
                print("Skipping: len(gt) = {:d}, len(pred) = {:d}, {:s}".format(label.size, ev.H * ev.W, name))
                continue
            rgb_d = torch.from_numpy(rgb[None]).to(dev)
            preps[key][0].run(rgb_d, xa)
            preps[key][1].run(rgb_d, xb)
            ev.add(xa, xb, torch.from_numpy(label[None].astype(np.int64)))
    if process_group is not None and world > 1:
        import torch.distributed as dist
        dist.all_reduce(ev.hist, group=process_group)
    miou, ius = ev.result()
    if verbose and rank == 0:
        for ind_class in range(num_classes):
            print("===>" + str(name_classes[ind_class]) + ":\t" + str(round(ius[ind_class] * 100, 2)))
        print("===> mIoU: " + str(miou) + ("" if ev.dtype == torch.float32 else "   (bf16 plans: not the reference's fp32 arithmetic)"))
    return miou


def evaluate_warmup(state, *args, **kw):
    """evaluate_cityscapes.py:165-225: the same loop for a warm-up (no open-set heads) checkpoint."""
    kw["open_classes"] = 0
    return evaluate_simt(state, *args, **kw)
