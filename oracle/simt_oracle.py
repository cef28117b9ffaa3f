"""CPU restatement of SimT's per-iteration hot path -- TEST INFRASTRUCTURE, never imported by the product.

Plain PyTorch-CPU fp32, functional style (parameters live in dicts keyed like the reference's state_dict).
Every function cites the reference lines it restates (paths relative to /root/reference).  The restatement is
pinned against the reference itself: oracle/gen_golden.py imports the reference in the build container, drives both
on identical seeded inputs and writes tests/golden/*.npz; tests/test_oracle_golden.py re-checks this file against
those vectors on every run (no GPU needed).  Known reference quirks are reproduced on purpose.
"""
import math
import zlib

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LAYERS = (3, 4, 23, 3)
IMG_MEAN = (104.00698793, 116.66876762, 122.67891434)  # tools/trainV2_simt.py:34 (BGR)
# ClassDist/ClassDist_bapa.npy (float64 [19], public data file of the reference; model/deeplab_multi.py:255)
CLASS_DIST_BAPA = None  # filled lazily from tests/golden/class_dist_bapa.npy


# ------------------------------------------------------------------------------------------------------------
# architecture description (model/deeplab_multi.py:122-167)
# ------------------------------------------------------------------------------------------------------------
def block_specs(layers=LAYERS):
    """[(name, inplanes, planes, stride, dilation, has_downsample)] for layer1..layer4."""
    out = []
    inpl = 64
    for li, (planes, nblk, stride, dil) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 1, 1), (1, 1, 2, 4)), 1):
        for b in range(nblk):
            out.append((f"layer{li}.{b}", inpl, planes, stride if b == 0 else 1, dil, b == 0))
            inpl = planes * 4
    return out


def head_names(openset):
    return ["layer5", "layer6"] + (["layer5_1", "layer6_1"] if openset else [])


def state_shapes(num_classes, open_classes=0, openset=False, layers=LAYERS, single_head=False):
    """Ordered {key: shape} equal to DeeplabMulti(...).state_dict() (656 keys for openset).
    single_head=True gives model/deeplab.py's ResNet (one 4-branch head named layer5 on layer4)."""
    sh = {}

    def bn(prefix, c):
        sh[prefix + ".weight"] = (c,)
        sh[prefix + ".bias"] = (c,)
        sh[prefix + ".running_mean"] = (c,)
        sh[prefix + ".running_var"] = (c,)
        sh[prefix + ".num_batches_tracked"] = ()

    sh["conv1.weight"] = (64, 3, 7, 7)
    bn("bn1", 64)
    for name, inpl, planes, stride, dil, down in block_specs(layers):
        sh[f"{name}.conv1.weight"] = (planes, inpl, 1, 1)
        bn(f"{name}.bn1", planes)
        sh[f"{name}.conv2.weight"] = (planes, planes, 3, 3)
        bn(f"{name}.bn2", planes)
        sh[f"{name}.conv3.weight"] = (planes * 4, planes, 1, 1)
        bn(f"{name}.bn3", planes * 4)
        if down:
            sh[f"{name}.downsample.0.weight"] = (planes * 4, inpl, 1, 1)
            bn(f"{name}.downsample.1", planes * 4)
    if single_head:
        for i in range(4):
            sh[f"layer5.conv2d_list.{i}.weight"] = (num_classes, 2048, 3, 3)
            sh[f"layer5.conv2d_list.{i}.bias"] = (num_classes,)
        return sh
    for hname in head_names(openset):
        cin = 1024 if hname.startswith("layer5") else 2048
        cout = open_classes if hname.endswith("_1") else num_classes
        for i in range(4):
            sh[f"{hname}.conv2d_list.{i}.weight"] = (cout, cin, 3, 3)
            sh[f"{hname}.conv2d_list.{i}.bias"] = (cout,)
    return sh


def recipe_state(shapes, seed=1234, trained_like=True, head_scale=1.0):
    """Build-owned deterministic weights: one torch.Generator per key (seed ^ crc32(key)).
    conv ~ N(0, 0.01) like model/deeplab_multi.py:144-150; trained_like=True also randomises the BN affine and
    running statistics (a released checkpoint has non-trivial ones) so that folding / frozen-affine paths are exercised."""
    st = {}
    for k, shp in shapes.items():
        g = torch.Generator().manual_seed((seed ^ zlib.crc32(k.encode())) & 0x7FFFFFFF)
        if k.endswith("num_batches_tracked"):
            st[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_mean"):
            st[k] = torch.randn(shp, generator=g) * 0.05 if trained_like else torch.zeros(shp)
        elif k.endswith("running_var"):
            st[k] = torch.rand(shp, generator=g) * 0.5 + 0.75 if trained_like else torch.ones(shp)
        elif ".bn" in k or k.startswith("bn1") or "downsample.1" in k:
            if k.endswith("weight"):
                st[k] = torch.rand(shp, generator=g) * 0.6 + 0.7 if trained_like else torch.ones(shp)
            else:
                st[k] = torch.randn(shp, generator=g) * 0.1 if trained_like else torch.zeros(shp)
        elif k.endswith("bias"):
            st[k] = torch.randn(shp, generator=g) * 0.01
        else:
            st[k] = torch.randn(shp, generator=g) * 0.01 * (head_scale if "conv2d_list" in k else 1.0)
    return st


# ------------------------------------------------------------------------------------------------------------
# forward (model/deeplab_multi.py:81-101 Bottleneck, :115-119 Classifier_Module, :172-192 ResNetMulti.forward)
# ------------------------------------------------------------------------------------------------------------
def _bn(st, prefix, x, train):
    # BatchNorm2d with frozen affine: train mode uses batch statistics AND updates the running ones (quirk 5)
    return F.batch_norm(x, st[prefix + ".running_mean"], st[prefix + ".running_var"], st[prefix + ".weight"],
                        st[prefix + ".bias"], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _bottleneck(st, name, x, stride, dil, down, train):
    out = F.conv2d(x, st[f"{name}.conv1.weight"], stride=stride)            # stride sits on the first 1x1 (:62)
    out = F.relu(_bn(st, f"{name}.bn1", out, train))
    out = F.conv2d(out, st[f"{name}.conv2.weight"], padding=dil, dilation=dil)
    out = F.relu(_bn(st, f"{name}.bn2", out, train))
    out = _bn(st, f"{name}.bn3", F.conv2d(out, st[f"{name}.conv3.weight"]), train)
    if down:
        x = _bn(st, f"{name}.downsample.1", F.conv2d(x, st[f"{name}.downsample.0.weight"], stride=stride), train)
    return F.relu(out + x)


def _aspp(st, hname, x, branches):
    # deeplab_multi.py:115-119 returns inside the loop -> only dilations 6 and 12 are live (quirk 1);
    # deeplab.py:112-116 sums all four.
    out = None
    for i, d in list(enumerate((6, 12, 18, 24)))[:branches]:
        y = F.conv2d(x, st[f"{hname}.conv2d_list.{i}.weight"], st[f"{hname}.conv2d_list.{i}.bias"], padding=d, dilation=d)
        out = y if out is None else out + y
    return out


def trunk(st, x, train, layers=LAYERS, upto=4):
    x = F.conv2d(x, st["conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(st, "bn1", x, train))
    x = F.max_pool2d(x, 3, 2, 1, ceil_mode=True)
    feats = {}
    for name, inpl, planes, stride, dil, down in block_specs(layers):
        li = int(name[5])
        if li > upto:
            break
        x = _bottleneck(st, name, x, stride, dil, down, train)
        feats[li] = x
    return feats


def deeplab_multi_forward(st, x, train, openset, layers=LAYERS):
    """-> (x1, x2): aux head on layer3, main head on layer4; open-set heads concatenated on dim 1."""
    f = trunk(st, x, train, layers)
    x1 = _aspp(st, "layer5", f[3], 2)
    x2 = _aspp(st, "layer6", f[4], 2)
    if openset:
        x1 = torch.cat([x1, _aspp(st, "layer5_1", f[3], 2)], 1)
        x2 = torch.cat([x2, _aspp(st, "layer6_1", f[4], 2)], 1)
    return x1, x2


def deeplab_single_forward(st, x, train, layers=LAYERS):
    """model/deeplab.py ResNet.forward: one 4-branch head, returned twice."""
    f = trunk(st, x, train, layers)
    y = _aspp(st, "layer5", f[4], 4)
    return y, y


def optim_param_names(shapes, warmup=False, openset=True):
    """Listing order AND multiplicity of ResNetMulti.optim_parameters (model/deeplab_multi.py:194-237, quirk 4):
    group 0 walks j.parameters() for EVERY sub-module j of layer3/layer4 (requires_grad filter commented out),
    so a conv weight inside a Bottleneck inside a Sequential is listed 3x (downsample members 4x)."""
    keys = [k for k in shapes if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]
    roots = (["conv1", "bn1", "layer1", "layer2"] if warmup else []) + ["layer3", "layer4"]

    def modules_of(root):
        # module paths in nn.Module.modules() pre-order
        mods = [root]
        if root.startswith("layer"):
            blocks = sorted({k.split(".")[1] for k in keys if k.startswith(root + ".")}, key=int)
            for b in blocks:
                bp = f"{root}.{b}"
                mods.append(bp)
                for sub in ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3", "relu"):
                    mods.append(f"{bp}.{sub}")
                if any(k.startswith(bp + ".downsample.") for k in keys):
                    mods += [f"{bp}.downsample", f"{bp}.downsample.0", f"{bp}.downsample.1"]
        return mods

    g0 = []
    for root in roots:
        for m in modules_of(root):
            g0 += [k for k in keys if k.startswith(m + ".")]
    g1 = []
    for h in head_names(openset):
        g1 += [k for k in keys if k.startswith(h + ".")]
    # reference order of heads: layer5, layer6, layer5_1, layer6_1 (:224-229)
    return g0, g1


# ------------------------------------------------------------------------------------------------------------
# NTM modules (model/deeplab_multi.py:244-286)
# ------------------------------------------------------------------------------------------------------------
def sig_ntm_forward(ntm, class_dist, num_classes):
    """T = L1-row-normalise( sigmoid(NTM) * tile(class_dist) + [I_C; 0_K] )   (:259-263)"""
    q = ntm.shape[0]
    prior = torch.cat([torch.eye(num_classes), torch.zeros(q - num_classes, num_classes)], 0)
    t = torch.sigmoid(ntm) * class_dist.to(ntm.dtype).unsqueeze(0) + prior.to(ntm.dtype)
    return F.normalize(t, p=1, dim=1)


def sig_w_forward(weight):
    """diag(weight) := -1e4 in place (no grad), W = softmax(weight, 1) - I   (:278-286)"""
    q = weight.shape[0]
    with torch.no_grad():
        weight[torch.arange(q), torch.arange(q)] = -10000.0
    return torch.softmax(weight, dim=1) - torch.eye(q, dtype=weight.dtype)


def ntm_init(num_classes, open_classes, seed):
    """kaiming_normal_(fan_out, relu) on a [Q, C] tensor (:248-252): std = sqrt(2 / fan_out), fan_out = Q."""
    q = num_classes + open_classes
    g = torch.Generator().manual_seed(seed)
    return torch.randn(q, num_classes, generator=g) * math.sqrt(2.0 / q)


def w_init(num_classes, open_classes):
    q = num_classes + open_classes
    return torch.full((q, q), 1.0 / (q - 1.0))


# ------------------------------------------------------------------------------------------------------------
# loss block (tools/trainV2_simt.py:351-424, :202-230; utils/loss.py:14-40)
# ------------------------------------------------------------------------------------------------------------
class Hyper:
    def __init__(self, num_classes=19, open_classes=15, th_high=0.8, th_low=0.2, lambda_seg=0.1, lambda_place=0.1,
                 lambda_convex=0.5, lambda_volume=0.1, lambda_anchor=0.5, iter_size=1, lr=2.5e-4, lr_T=2.5e-4,
                 momentum=0.9, weight_decay=5e-4, power=0.9, num_steps=250000):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def upsample(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)   # interp_target, :301


def confidence_labels(fixed_lr2, size, hp):
    """Pseudo-label generation from the frozen model (:351-361). Returns (Conf0 [B,H,W] long, labelC_flat [P,C])."""
    prob = upsample(torch.softmax(fixed_lr2, 1), size)
    m, a = prob.max(1)
    conf = torch.where(m > hp.th_high, a, torch.full_like(a, 255))
    conf = torch.where(m < hp.th_low, torch.full_like(a, hp.num_classes), conf)
    return conf, prob.permute(0, 2, 3, 1).reshape(-1, hp.num_classes)


def placeholder_loss(pred, hp):
    """Placeholder_loss (:202-230) incl. quirk 2: the arg-max logit is replaced by -0.0 (`-1000 * zeros`)."""
    c = hp.num_classes
    arg = pred.argmax(1)
    onehot = F.one_hot(arg, pred.shape[1]).permute(0, 3, 1, 2).bool()
    predict = torch.where(onehot, torch.full_like(pred, -0.0), pred)
    pmax = torch.softmax(pred.detach(), 1).max(1)[0]
    pseudo1 = torch.where((arg < c) & (pmax > hp.th_high), arg, torch.full_like(arg, 255))
    known = F.cross_entropy(pred, pseudo1, ignore_index=255)
    opened = torch.zeros_like(predict)
    opened[:, c:] = predict[:, c:].detach()
    y = opened.argmax(1)
    y = torch.where(pseudo1 == 255, torch.full_like(y, 255), y)
    unknown = F.cross_entropy(predict, y, ignore_index=255)
    return known + hp.lambda_place * unknown, known, unknown


def noisy_nll(pred_up, T, label):
    """softmax -> @T -> log -> masked NLL mean (:402-409, utils/loss.py:29-39; no epsilon inside the log)."""
    b, q, h, w = pred_up.shape
    prob = torch.softmax(pred_up, 1).permute(0, 2, 3, 1).reshape(-1, q)
    r = prob @ T
    lab = label.reshape(-1)
    valid = (lab >= 0) & (lab != 255)
    return F.nll_loss(torch.log(r[valid]), lab[valid], reduction="mean")


def anchor_loss(pred_up, T, labelC_flat):
    """Anchor term for one head (:375-379): rows of T for classes that are some pixel's arg-max are pulled towards the
    frozen model's posterior at that channel's global arg-max pixel (first index)."""
    q = pred_up.shape[1]
    flat = pred_up.detach().permute(0, 2, 3, 1).reshape(-1, q)
    idx = flat.argmax(0)
    exist = torch.unique(flat.argmax(1))
    anchor = labelC_flat[idx]
    return ((T[exist] - anchor[exist]) ** 2).sum(), idx, exist


def simt_losses(pred_lr1, pred_lr2, fixed_lr2, label, T1, T2, W1, W2, hp, size):
    """All loss terms of one sub-iteration. pred_lr*: [B,Q,h,w] low-res logits (require grad upstream), label [B,H,W]."""
    c = hp.num_classes
    conf0, labelC_flat = confidence_labels(fixed_lr2.detach(), size, hp)
    p1, p2 = upsample(pred_lr1, size), upsample(pred_lr2, size)
    a1, idx1, ex1 = anchor_loss(p1, T1, labelC_flat)
    a2, idx2, ex2 = anchor_loss(p2, T2, labelC_flat)
    anchor = a1 + a2
    # class-posterior constraint (:387-395): low-confidence pixels follow the main head iff it predicts an open class
    pseudo = p2.detach().argmax(1)
    repl = torch.where(pseudo >= c, pseudo, torch.full_like(pseudo, 255))
    conf = torch.where(conf0 == c, repl, conf0)
    loss_p1 = F.cross_entropy(p1, conf, ignore_index=255)
    loss_p2 = F.cross_entropy(p2, conf, ignore_index=255)
    pl1, k1, u1 = placeholder_loss(p1, hp)
    pl2, k2, u2 = placeholder_loss(p2, hp)
    place = hp.lambda_seg * pl1 + pl2
    loss_y1 = noisy_nll(p1, T1, label)
    loss_y2 = noisy_nll(p2, T2, label)
    convex = 0.0 - ((W1 @ T1) ** 2).sum() - ((W2 @ T2) ** 2).sum()
    vol = torch.log(torch.sqrt(torch.abs(torch.linalg.det(T1.t() @ T1)))) + \
        torch.log(torch.sqrt(torch.abs(torch.linalg.det(T2.t() @ T2))))
    if torch.isinf(vol) or torch.isnan(vol):
        vol = 0.0                                                             # :420-421
    target = loss_p2 + loss_y2 + hp.lambda_seg * loss_p1 + hp.lambda_seg * loss_y1
    total = place + target + hp.lambda_convex * convex + hp.lambda_volume * vol + hp.lambda_anchor * anchor
    return {"total": total / hp.iter_size, "loss_p1": loss_p1, "loss_p2": loss_p2, "loss_y1": loss_y1, "loss_y2": loss_y2,
            "place": place, "convex": convex, "volume": vol if torch.is_tensor(vol) else torch.tensor(vol),
            "anchor": anchor, "conf": conf, "conf0": conf0, "anchor_idx1": idx1, "anchor_idx2": idx2,
            "exist1": ex1, "exist2": ex2, "known1": k1, "known2": k2, "unknown1": u1, "unknown2": u2}


def adam_step_(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (weight_decay 0, amsgrad False), single-tensor formulation; in place on p, m, v."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def inner_w_loop(ntm1, ntm2, w1, w2, state, class_dist, hp, lr_T, steps=10):
    """The 10-step W optimisation (:326-339).  ntm*/w*: leaf tensors with .grad (NTM grads ACCUMULATE: quirk 3).
    state: dict with exp_avg/exp_avg_sq per W and the shared step counter."""
    c = hp.num_classes
    for _ in range(steps):
        T1 = sig_ntm_forward(ntm1, class_dist, c)
        T2 = sig_ntm_forward(ntm2, class_dist, c)
        W1 = sig_w_forward(w1)
        W2 = sig_w_forward(w2)
        w1.grad = None
        w2.grad = None
        loss = ((W1 @ T1) ** 2).sum() + ((W2 @ T2) ** 2).sum()
        loss.backward()
        state["step"] += 1
        with torch.no_grad():
            adam_step_(w1, w1.grad, state["m1"], state["v1"], state["step"], lr_T)
            adam_step_(w2, w2.grad, state["m2"], state["v2"], state["step"], lr_T)


def sgd_step_(params, grads, bufs, mults, lr, wd, momentum, first):
    """torch.optim.SGD(foreach=False) with a parameter listed `mult` times (quirk 4): the update is applied once per
    listing, sequentially; on the very first step every listing starts its own buffer (= d_p), later steps share one."""
    for p, g, b, mult in zip(params, grads, bufs, mults):
        for _ in range(mult):
            d = g + wd * p if wd != 0 else g
            if momentum != 0:
                if first:
                    b.copy_(d)
                else:
                    b.mul_(momentum).add_(d)
                d = b
            p.sub_(lr * d)


def lr_poly(base_lr, it, max_iter, power):
    return base_lr * ((1 - float(it) / max_iter) ** power)                    # :174-175


# ------------------------------------------------------------------------------------------------------------
# evaluation metric (tools/evaluate_cityscapes.py:81-87)
# ------------------------------------------------------------------------------------------------------------
def fast_hist(gt, pred, n):
    k = (gt >= 0) & (gt < n)
    return np.bincount(n * gt[k].astype(int) + pred[k], minlength=n * n).reshape(n, n)


def per_class_iu(hist):
    d = np.diag(hist)
    return d / (hist.sum(1) + hist.sum(0) - d)


def miou(hist):
    return round(float(np.nanmean(per_class_iu(hist))) * 100, 2)


# ------------------------------------------------------------------------------------------------------------
# synthetic inputs shared by bench.py / tests: Cityscapes-shaped image + blocky noisy labels
# ------------------------------------------------------------------------------------------------------------
def synthetic_batch(B, H, W, class_dist, seed=1234, block=16):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (B, 3, H, W), generator=g).float()
    img = img - torch.tensor(IMG_MEAN).view(1, 3, 1, 1)
    hb, wb = (H + block - 1) // block, (W + block - 1) // block
    p = torch.as_tensor(np.asarray(class_dist), dtype=torch.float64)
    lab = torch.multinomial(p / p.sum(), B * hb * wb, replacement=True, generator=g).view(B, hb, wb)
    ign = torch.rand(B, hb, wb, generator=g) < 0.1
    lab = torch.where(ign, torch.full_like(lab, 255), lab)
    lab = lab.repeat_interleave(block, 1).repeat_interleave(block, 2)[:, :H, :W].contiguous()
    return img, lab.long()


# ------------------------------------------------------------------------------------------------------------
# full training iteration (tools/trainV2_simt.py:308-436) on top of the pieces above
# ------------------------------------------------------------------------------------------------------------
def load_class_dist(name="bapa"):
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    return torch.as_tensor(np.load(os.path.join(here, "..", "tests", "golden", f"class_dist_{name}.npy")))


class OracleTrainer:
    """State + one-iteration step of the SimT stage, CPU fp32.  Mirrors what main() keeps between iterations."""

    def __init__(self, st, fixed_st, ntm1, ntm2, hp, class_dist, openset=True, dtype=torch.float32, layers=LAYERS):
        """dtype=torch.float64 gives the "exact arithmetic" run used to measure how far fp32 implementations
        (the reference's CPU path included) sit from the true value of an ill-conditioned quantity."""
        self.hp = hp
        self.cd = class_dist
        self.openset = openset
        self.dtype = dtype
        self.layers = layers
        st = {k: (v.to(dtype) if v.dtype != torch.long else v) for k, v in st.items()}
        fixed_st = {k: (v.to(dtype) if v.dtype != torch.long else v) for k, v in fixed_st.items()}
        ntm1, ntm2 = ntm1.to(dtype), ntm2.to(dtype)
        self.st = {k: (v.clone().requires_grad_(True) if v.dtype != torch.long and not
                       (k.endswith("running_mean") or k.endswith("running_var")) else v.clone()) for k, v in st.items()}
        # BN affine is frozen (deeplab_multi.py:64-76,130-131,159-160)
        for k, v in self.st.items():
            if (".bn" in k or k.startswith("bn1") or "downsample.1" in k) and v.dtype != torch.long and v.requires_grad:
                v.requires_grad_(False)
        self.fixed = {k: v.clone() for k, v in fixed_st.items()}
        self.ntm = [ntm1.clone().requires_grad_(True), ntm2.clone().requires_grad_(True)]
        q = ntm1.shape[0]
        self.w = [w_init(hp.num_classes, q - hp.num_classes).to(dtype).requires_grad_(True) for _ in range(2)]
        self.wstate = {"step": 0, "m1": torch.zeros(q, q, dtype=dtype), "v1": torch.zeros(q, q, dtype=dtype),
                       "m2": torch.zeros(q, q, dtype=dtype), "v2": torch.zeros(q, q, dtype=dtype)}
        self.tstate = [{"step": 0, "m": torch.zeros_like(ntm1), "v": torch.zeros_like(ntm1)} for _ in range(2)]
        shapes = {k: tuple(v.shape) for k, v in st.items()}
        g0, g1 = optim_param_names(shapes, warmup=False, openset=openset)
        self.groups = []
        for names, lr_mult in ((g0, 1.0), (g1, 10.0)):
            uniq = list(dict.fromkeys(names))
            self.groups.append({"names": uniq, "mult": [names.count(n) for n in uniq], "lr_mult": lr_mult})
        self.bufs = {}
        self.first = True

    def step(self, image, label, it, apply=True):
        """image / label: one micro-batch, or lists of hp.iter_size micro-batches (gradient accumulation,
        trainV2_simt.py:341-432: the W loop and zero_grad run once, every sub-iteration re-evaluates T, runs both nets,
        back-propagates loss / iter_size, and the optimisers step once)."""
        hp = self.hp
        c = hp.num_classes
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        labels = list(label) if isinstance(label, (list, tuple)) else [label]
        assert len(images) == len(labels) == getattr(hp, "iter_size", 1)
        lr = lr_poly(hp.lr, it, hp.num_steps, hp.power)
        lr_T = lr_poly(hp.lr_T, it, hp.num_steps, hp.power)
        for v in self.st.values():
            if v.dtype != torch.long:
                v.grad = None
        for t in self.ntm:
            t.grad = None
        inner_w_loop(self.ntm[0], self.ntm[1], self.w[0], self.w[1], self.wstate, self.cd, hp, lr_T)
        for image, label in zip(images, labels):
            image = image.to(self.dtype)
            T1 = sig_ntm_forward(self.ntm[0], self.cd, c)
            T2 = sig_ntm_forward(self.ntm[1], self.cd, c)
            size = tuple(label.shape[1:])
            with torch.no_grad():
                _, f2 = deeplab_multi_forward(self.fixed, image, False, False, self.layers)
            x1, x2 = deeplab_multi_forward(self.st, image, True, self.openset, self.layers)
            W1 = sig_w_forward(self.w[0])
            W2 = sig_w_forward(self.w[1])
            out = simt_losses(x1, x2, f2, label, T1, T2, W1, W2, hp, size)
            out["total"].backward()                   # simt_losses already divides by hp.iter_size (:427)
        if apply:
            self.apply_update(it)
        return out

    def applied_grads(self):
        """The gradient tensors the optimisers consume (:434-436): name -> .grad of every parameter listed in the SGD groups that
        received one, plus "NTM1" / "NTM2" (which carry the inner loop's leak, quirk 3).  Data parallelism averages
        exactly these over the ranks; conv1 / layer1 / layer2 gradients are computed but never applied and stay rank-local."""
        out = {}
        for g in self.groups:
            for n in g["names"]:
                if self.st[n].grad is not None:
                    out[n] = self.st[n].grad
        out["NTM1"], out["NTM2"] = self.ntm[0].grad, self.ntm[1].grad
        return out

    def apply_update(self, it):
        """optimizer.step() + optimizer_t1/t2.step() (:434-436) on whatever .grad currently holds."""
        hp = self.hp
        lr = lr_poly(hp.lr, it, hp.num_steps, hp.power)
        lr_T = lr_poly(hp.lr_T, it, hp.num_steps, hp.power)
        with torch.no_grad():
            for g in self.groups:
                ps, gs, bs, ms = [], [], [], []
                for n, m in zip(g["names"], g["mult"]):
                    p = self.st[n]
                    if p.grad is None:
                        continue
                    if n not in self.bufs:
                        self.bufs[n] = torch.zeros_like(p)
                    ps.append(p); gs.append(p.grad); bs.append(self.bufs[n]); ms.append(m)
                sgd_step_(ps, gs, bs, ms, lr * g["lr_mult"], hp.weight_decay, hp.momentum, self.first)
            self.first = False
            for k in range(2):
                s = self.tstate[k]
                s["step"] += 1
                adam_step_(self.ntm[k], self.ntm[k].grad, s["m"], s["v"], s["step"], lr_T)


def oracle_dp_step(replicas, images, labels, it):
    """One data-parallel iteration of the SimT stage as SURVEY 8e defines it (the reference has no DP: `--gpu` is parsed and
    ignored, trainV2_simt.py:151,246): every rank runs trainV2_simt.py:308-432 on ITS micro-batch (per-rank BN statistics, anchors,
    CE means, W inner loop -- the latter replica-deterministic), the gradients the optimisers apply are replaced by their MEAN over
    the ranks (NTM gradients including the rank-identical inner-loop leak: averaged, not summed), then every rank runs :434-436.
    replicas: OracleTrainer per rank, holding identical state.  Returns the per-rank loss dicts."""
    outs = [r.step(im, lb, it, apply=False) for r, im, lb in zip(replicas, images, labels)]
    grads = [r.applied_grads() for r in replicas]
    with torch.no_grad():
        for n in grads[0]:
            mean = sum(g[n] for g in grads) / len(grads)
            for g in grads:
                g[n].copy_(mean)
    for r in replicas:
        r.apply_update(it)
    return outs


# ------------------------------------------------------------------------------------------------------------
# warm-up stage (tools/trainV1_warmup.py:156-256): DeeplabMulti(num_classes) without open-set heads, plain CE
# ------------------------------------------------------------------------------------------------------------
def warmup_losses(pred_lr1, pred_lr2, label, lambda_seg, size):
    """loss = CE(up(pred2), label) + lambda_seg * CE(up(pred1), label), ignore_index 255 (:213-224)."""
    p1, p2 = upsample(pred_lr1, size), upsample(pred_lr2, size)
    l1 = F.cross_entropy(p1, label, ignore_index=255)
    l2 = F.cross_entropy(p2, label, ignore_index=255)
    return l2 + lambda_seg * l1, l1, l2


class OracleWarmupTrainer:
    """One warm-up iteration: forward (train-mode BN), CE on both heads, backward, SGD over conv1..layer4 (duplicate
    listings, `optim_parameters(args, warmup=True)`) + heads at 10x lr (:192-193, :205-232)."""

    def __init__(self, st, hp, layers=LAYERS, dtype=torch.float32):
        self.hp, self.layers, self.dtype = hp, layers, dtype
        st = {k: (v.to(dtype) if v.dtype != torch.long else v) for k, v in st.items()}
        self.st = {k: (v.clone().requires_grad_(True) if v.dtype != torch.long and not
                       (k.endswith("running_mean") or k.endswith("running_var")) else v.clone()) for k, v in st.items()}
        for k, v in self.st.items():
            if (".bn" in k or k.startswith("bn1") or "downsample.1" in k) and v.dtype != torch.long and v.requires_grad:
                v.requires_grad_(False)
        shapes = {k: tuple(v.shape) for k, v in st.items()}
        g0, g1 = optim_param_names(shapes, warmup=True, openset=False)
        self.groups = []
        for names, lr_mult in ((g0, 1.0), (g1, 10.0)):
            uniq = list(dict.fromkeys(names))
            self.groups.append({"names": uniq, "mult": [names.count(n) for n in uniq], "lr_mult": lr_mult})
        self.bufs, self.first = {}, True

    def step(self, image, label, it):
        hp = self.hp
        lr = lr_poly(hp.lr, it, hp.num_steps, hp.power)
        for v in self.st.values():
            if v.dtype != torch.long:
                v.grad = None
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        labels = list(label) if isinstance(label, (list, tuple)) else [label]
        assert len(images) == len(labels) == hp.iter_size            # trainV1_warmup.py:212-231: loss / iter_size, grads accumulate
        for image, label in zip(images, labels):
            x1, x2 = deeplab_multi_forward(self.st, image.to(self.dtype), True, False, layers=self.layers)
            total, l1, l2 = warmup_losses(x1, x2, label, hp.lambda_seg, tuple(label.shape[1:]))
            total = total / hp.iter_size
            total.backward()
        with torch.no_grad():
            for g in self.groups:
                ps, gs, bs, ms = [], [], [], []
                for n, m in zip(g["names"], g["mult"]):
                    p = self.st[n]
                    if p.grad is None:
                        continue
                    if n not in self.bufs:
                        self.bufs[n] = torch.zeros_like(p)
                    ps.append(p); gs.append(p.grad); bs.append(self.bufs[n]); ms.append(m)
                sgd_step_(ps, gs, bs, ms, lr * g["lr_mult"], hp.weight_decay, hp.momentum, self.first)
            self.first = False
        return {"total": total.detach(), "loss_seg1": l1.detach(), "loss_seg2": l2.detach()}


# ------------------------------------------------------------------------------------------------------------
# DeepLab-VGG16 (model/deeplab_vgg.py:24-51).  PARITY UNPINNED: the reference file is Python-2-only and needs torchvision
# (absent), no reference test or script exercises it; restated from the file + torchvision's public vgg16 layer list.
# ------------------------------------------------------------------------------------------------------------
def vgg_forward(st, x, layers):
    """layers: [(features index, cin, cout, dilation, pool_after)]; every conv is 3x3, pad = dilation, bias + ReLU;
    MaxPool2d(2, 2); then the 2-branch classifier (dilations 6, 12; the early `return` of :17-21)."""
    for (idx, _cin, _cout, dil, pool) in layers:
        x = F.relu(F.conv2d(x, st[f"features.{idx}.weight"], st[f"features.{idx}.bias"], padding=dil, dilation=dil))
        if pool:
            x = F.max_pool2d(x, 2, 2)
    return _aspp(st, "classifier", x, 2)


# ------------------------------------------------------------------------------------------------------------
# DeepLabv3 (model/deeplabv3.py:9-138).  PARITY UNPINNED: the reference file imports torchvision (absent here) and no
# reference script or test uses it; restated from the file + torchvision's public resnet50 definition (Bottleneck with
# the stride on the 3x3 conv, a downsample branch in block 0 of every layer, MaxPool2d(3, 2, 1) in floor mode).
# ------------------------------------------------------------------------------------------------------------
def v3_forward(st, x, layers=(3, 4, 6), openset=False, train=True, acts=None):
    """st: {state_dict key: tensor} (keys as model/deeplabv3.py builds them: resnet.resnet_50.*, assp.*, conv, conv_1).
    acts: optional dict that receives every post-ReLU activation (tests compare ReLU masks with it)."""
    def keep(name, t):
        if acts is not None:
            acts[name] = t.detach()
        return t
    r = "resnet.resnet_50."
    H, W = x.shape[-2:]
    x = F.relu(_bn(st, r + "bn1", F.conv2d(x, st[r + "conv1.weight"], stride=2, padding=3), train))       # :16
    x = F.max_pool2d(x, 3, 2, 1)                                                                          # :17
    for li, n in enumerate(layers):                                                                       # :18-20
        for bi in range(n):
            name = f"{r}layer{li + 1}.{bi}"
            stride = 2 if (li > 0 and bi == 0) else 1
            out = keep(name + ".a1", F.relu(_bn(st, name + ".bn1", F.conv2d(x, st[name + ".conv1.weight"]), train)))
            out = keep(name + ".a2", F.relu(_bn(st, name + ".bn2", F.conv2d(out, st[name + ".conv2.weight"], stride=stride, padding=1),
                                                train)))
            out = _bn(st, name + ".bn3", F.conv2d(out, st[name + ".conv3.weight"]), train)
            if bi == 0:
                x = _bn(st, name + ".downsample.1", F.conv2d(x, st[name + ".downsample.0.weight"], stride=stride), train)
            x = keep(name + ".z", F.relu(out + x))
    br = []
    for (i, k, d) in ((1, 1, 1), (2, 3, 6), (3, 3, 12), (4, 3, 18), (5, 1, 1)):                           # ASSP :82-102
        y = F.conv2d(x, st[f"assp.conv{i}.weight"], padding=d if k == 3 else 0, dilation=d if k == 3 else 1)
        br.append(keep(f"assp.a{i}", F.relu(_bn(st, f"assp.bn{i}", y, train))))
    br[4] = F.interpolate(br[4], size=tuple(br[3].shape[-2:]), mode="bilinear")                           # :102 (identity)
    x = keep("assp.af", F.relu(_bn(st, "assp.bnf", F.conv2d(torch.cat(br, 1), st["assp.convf.weight"]), train)))   # :104-107
    y = F.conv2d(x, st["conv.weight"], st["conv.bias"])                                                   # :131
    if openset:
        y = torch.cat([y, F.conv2d(x, st["conv_1.weight"], st["conv_1.bias"])], 1)                        # :133-135
    return F.interpolate(y, size=(H, W), mode="bilinear")                                                 # :137


# ------------------------------------------------------------------------------------------------------------
# bf16 storage model of the GPU throughput path: the same network evaluated in float64 with a
# round-to-bf16 at every point where the HIP path STORES a tensor as bf16 (conv operands and outputs, BN / ReLU
# outputs); accumulation, BN statistics, head logits and losses stay wide.  It is what a bf16 run of the product is
# checked against at full depth (tests/test_gpu_prod_shapes.py): the difference that is left is accumulation order
# plus the occasional 1-ulp bf16 rounding flip, instead of the ~1e-2 of bf16 itself.  Gradients flow straight
# through the roundings.  Test infrastructure, like the rest of this file.
# ------------------------------------------------------------------------------------------------------------
class _RoundBF16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def q16(t):
    return _RoundBF16.apply(t)


def _bn_fold(st, prefix):
    scale = st[prefix + ".weight"].float() / torch.sqrt(st[prefix + ".running_var"].float() + BN_EPS)     # fp32 like simt_bn_fold
    shift = st[prefix + ".bias"].float() - st[prefix + ".running_mean"].float() * scale
    return scale.double(), shift.double()


def _q16_conv(st, xx, wname, fold=None, **kw):
    d = torch.float64
    w = st[wname].to(d)
    if fold is not None:
        sc, sh = _bn_fold(st, fold)
        return F.conv2d(xx, q16(w * sc.view(-1, 1, 1, 1)), **kw) + sh.view(1, -1, 1, 1)      # caller applies residual / ReLU, then rounds
    return q16(F.conv2d(xx, q16(w), **kw))


def _q16_bn(st, prefix, y):
    d = torch.float64
    return F.batch_norm(y, None, None, st[prefix + ".weight"].to(d), st[prefix + ".bias"].to(d), training=True, eps=BN_EPS)


def bf16_block_forward(st, name, a, stride, dil, down, train=True):
    """One Bottleneck (model/deeplab_multi.py:81-101) as the bf16 plans store it -> (a1, a2, z); a: float64 tensor holding bf16 values."""
    conv, bn = _q16_conv, _q16_bn
    if train:
        a1 = q16(F.relu(bn(st, f"{name}.bn1", conv(st, a, f"{name}.conv1.weight", stride=stride))))
        a2 = q16(F.relu(bn(st, f"{name}.bn2", conv(st, a1, f"{name}.conv2.weight", padding=dil, dilation=dil))))
        out = bn(st, f"{name}.bn3", conv(st, a2, f"{name}.conv3.weight"))
        sc = bn(st, f"{name}.downsample.1", conv(st, a, f"{name}.downsample.0.weight", stride=stride)) if down else a
        return a1, a2, q16(F.relu(out + sc))
    a1 = q16(F.relu(conv(st, a, f"{name}.conv1.weight", fold=f"{name}.bn1", stride=stride)))
    a2 = q16(F.relu(conv(st, a1, f"{name}.conv2.weight", fold=f"{name}.bn2", padding=dil, dilation=dil)))
    sc = q16(conv(st, a, f"{name}.downsample.0.weight", fold=f"{name}.downsample.1", stride=stride)) if down else a
    return a1, a2, q16(F.relu(conv(st, a2, f"{name}.conv3.weight", fold=f"{name}.bn3") + sc))


def bf16_aspp(st, hname, f):
    """Classifier_Module on a bf16 feature map: bf16 weights, fp32 logits (no rounding of the output)."""
    d = torch.float64
    out = None
    for i, dl in ((0, 6), (1, 12)):
        y = F.conv2d(f, q16(st[f"{hname}.conv2d_list.{i}.weight"].to(d)), st[f"{hname}.conv2d_list.{i}.bias"].to(d), padding=dl, dilation=dl)
        out = y if out is None else out + y
    return out


def bf16_model_forward(st, x, train, openset, layers=LAYERS, heads=("layer5", "layer6")):
    """deeplab_multi_forward as the bf16 HIP plans store it (engine.py TrunkPlan: train = batch-statistics BN applied by
    simt_bn_apply to the stored bf16 conv output; eval = BN scale folded into the bf16 weights, shift in the epilogue)."""
    d = torch.float64
    x = q16(x.to(d))                                                    # stem im2col matrix is bf16
    if train:
        a = q16(F.relu(_q16_bn(st, "bn1", _q16_conv(st, x, "conv1.weight", stride=2, padding=3))))
    else:
        a = q16(F.relu(_q16_conv(st, x, "conv1.weight", fold="bn1", stride=2, padding=3)))
    a = F.max_pool2d(a, 3, 2, 1, ceil_mode=True)
    feats = {}
    for name, inpl, planes, stride, dil, down in block_specs(layers):
        _, _, a = bf16_block_forward(st, name, a, stride, dil, down, train)
        feats[int(name[5])] = a
    x1, x2 = bf16_aspp(st, heads[0], feats[3]), bf16_aspp(st, heads[1], feats[4])
    if openset:
        x1 = torch.cat([x1, bf16_aspp(st, heads[0] + "_1", feats[3])], 1)
        x2 = torch.cat([x2, bf16_aspp(st, heads[1] + "_1", feats[4])], 1)
    return x1, x2

# ------------------------------------------------------------------------------------------------------------
# SimT iteration over a ONE-OUTPUT model (BASELINE configs[3] DeepLabv3, configs[4] DeepLab-VGG16).
# The reference has no script that trains these files; its only SimT loop is tools/trainV2_simt.py:308-436, written for the
# two-output DeeplabMulti.  The iteration is defined here as THAT loop with every auxiliary-head object removed
# (pred1, NTM1, NTM_W1, and the lambda_seg-weighted terms they feed): what is left is, term by term, the reference's code
# applied to the model's single output.  `simt_losses_single` is tied to the golden-pinned two-head `simt_losses` by an
# identity checked in tests/test_oracle_golden.py: with pred1 = pred2, T1 = T2, W1 = W2 and lambda_seg = 0,
#   total_two_head = total_single + lambda_convex * convex + lambda_volume * volume + lambda_anchor * anchor.
# ------------------------------------------------------------------------------------------------------------
def simt_losses_single(pred_up, prob_up, label, T, W, hp):
    """pred_up [B,Q,H,W]: the model's logits at label resolution (requires grad upstream); prob_up [B,C,H,W]: the frozen model's
    posterior at label resolution (trainV2_simt.py:354: `interp_target(softmax(output))`)."""
    c = hp.num_classes
    m, a = prob_up.max(1)                                                     # :355-361
    conf0 = torch.where(m > hp.th_high, a, torch.full_like(a, 255))
    conf0 = torch.where(m < hp.th_low, torch.full_like(a, c), conf0)
    labelC_flat = prob_up.permute(0, 2, 3, 1).reshape(-1, c)
    anchor, idx, ex = anchor_loss(pred_up, T, labelC_flat)                    # :375-384 (one head)
    pseudo = pred_up.detach().argmax(1)                                       # :387-393
    repl = torch.where(pseudo >= c, pseudo, torch.full_like(pseudo, 255))
    conf = torch.where(conf0 == c, repl, conf0)
    loss_p = F.cross_entropy(pred_up, conf, ignore_index=255)                 # :395
    place, known, unknown = placeholder_loss(pred_up, hp)                     # :399
    loss_y = noisy_nll(pred_up, T, label)                                     # :405-409
    convex = 0.0 - ((W @ T) ** 2).sum()                                       # :412-416
    vol = torch.log(torch.sqrt(torch.abs(torch.linalg.det(T.t() @ T))))       # :417-419
    if torch.isinf(vol) or torch.isnan(vol):
        vol = 0.0                                                             # :420-421
    total = place + loss_p + loss_y + hp.lambda_convex * convex + hp.lambda_volume * vol + hp.lambda_anchor * anchor
    return {"total": total / hp.iter_size, "loss_p": loss_p, "loss_y": loss_y, "place": place, "convex": convex,
            "volume": vol if torch.is_tensor(vol) else torch.tensor(vol), "anchor": anchor, "conf": conf, "anchor_idx": idx,
            "exist": ex, "known": known, "unknown": unknown}


def inner_w_loop_single(ntm, w, state, class_dist, hp, lr_T, steps=10):
    """:326-339 with one NTM / W pair: L = ||W T||^2, Adam on W, the gradient leaks into ntm.grad (quirk 3)."""
    for _ in range(steps):
        T = sig_ntm_forward(ntm, class_dist, hp.num_classes)
        Wm = sig_w_forward(w)
        w.grad = None
        ((Wm @ T) ** 2).sum().backward()
        state["step"] += 1
        with torch.no_grad():
            adam_step_(w, w.grad, state["m"], state["v"], state["step"], lr_T)


def v3_optim_names(shapes, openset):
    """DeepLabv3.optim_parameters (model/deeplabv3.py:139-166): group 0 = parameters of self.resnet whose name RELATIVE TO self.resnet
    contains 'resnet_50.layer3' (the walk over every sub-module only matches at the top one, so each tensor is listed once; layer4 /
    fc never run), group 1 (10x) = assp + conv (+ conv_1)."""
    g0 = [k for k in shapes if k.startswith("resnet.resnet_50.layer3.") and (k.endswith(".weight") or k.endswith(".bias"))]
    g1 = [k for k in shapes if (k.startswith("assp.") or k.startswith("conv.") or (openset and k.startswith("conv_1.")))
          and (k.endswith(".weight") or k.endswith(".bias"))]
    return g0, g1


class OracleSingleTrainer:
    """One SimT iteration over a one-output model, CPU.  model = "v3" (oracle.v3_forward: upsamples inside, align_corners=False) or
    "vgg" (oracle.vgg_forward: low-res output, interp_target outside).  Mirrors OracleTrainer."""

    def __init__(self, model, st, fixed_st, ntm, hp, class_dist, arch, dtype=torch.float32):
        self.model, self.hp, self.cd, self.arch, self.dtype = model, hp, class_dist, arch, dtype
        cv = lambda d: {k: (v.to(dtype) if v.dtype != torch.long else v) for k, v in d.items()}
        st, self.fixed = cv(st), {k: v.clone() for k, v in cv(fixed_st).items()}
        stat = lambda k: k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
        self.st = {k: (v.clone().requires_grad_(True) if not stat(k) else v.clone()) for k, v in st.items()}
        self.ntm = ntm.to(dtype).clone().requires_grad_(True)
        q = ntm.shape[0]
        self.w = w_init(hp.num_classes, q - hp.num_classes).to(dtype).requires_grad_(True)
        self.wstate = {"step": 0, "m": torch.zeros(q, q, dtype=dtype), "v": torch.zeros(q, q, dtype=dtype)}
        self.tstate = {"step": 0, "m": torch.zeros_like(self.ntm), "v": torch.zeros_like(self.ntm)}
        shapes = {k: tuple(v.shape) for k, v in st.items()}
        if model == "v3":
            g0, g1 = v3_optim_names(shapes, True)
            self.groups = [{"names": g0, "lr_mult": 1.0}, {"names": g1, "lr_mult": 10.0}]
        else:                                             # DeeplabVGG.optim_parameters = self.parameters(): one group (deeplab_vgg.py:53-54)
            self.groups = [{"names": [k for k in shapes if k.endswith(".weight") or k.endswith(".bias")], "lr_mult": 1.0}]
        self.bufs, self.first = {}, True

    def forward(self, st, image, train):
        if self.model == "v3":
            return v3_forward(st, image, layers=self.arch["layers"], openset="conv_1.weight" in st, train=train)
        return vgg_forward(st, image, self.arch["layers"])

    def step(self, image, label, it):
        hp = self.hp
        lr, lr_T = lr_poly(hp.lr, it, hp.num_steps, hp.power), lr_poly(hp.lr_T, it, hp.num_steps, hp.power)
        for v in self.st.values():
            if v.dtype != torch.long:
                v.grad = None
        self.ntm.grad = None
        inner_w_loop_single(self.ntm, self.w, self.wstate, self.cd, hp, lr_T)
        image = image.to(self.dtype)
        size = tuple(label.shape[1:])
        T = sig_ntm_forward(self.ntm, self.cd, hp.num_classes)
        with torch.no_grad():
            prob = upsample(torch.softmax(self.forward(self.fixed, image, False), 1), size)       # :352-354 (identity for v3)
        pred = upsample(self.forward(self.st, image, True), size)                                 # :370-372 (identity for v3)
        out = simt_losses_single(pred, prob, label, T, sig_w_forward(self.w), hp)
        out["total"].backward()
        with torch.no_grad():
            for g in self.groups:
                ps, gs, bs, ms = [], [], [], []
                for n in g["names"]:
                    p = self.st[n]
                    if p.grad is None:
                        continue
                    if n not in self.bufs:
                        self.bufs[n] = torch.zeros_like(p)
                    ps.append(p); gs.append(p.grad); bs.append(self.bufs[n]); ms.append(1)
                sgd_step_(ps, gs, bs, ms, lr * g["lr_mult"], hp.weight_decay, hp.momentum, self.first)
            self.first = False
            s = self.tstate
            s["step"] += 1
            adam_step_(self.ntm, self.ntm.grad, s["m"], s["v"], s["step"], lr_T)
        return out


# ------------------------------------------------------------------------------------------------------------
# offline NTM utilities (tools/compute_ClassDistribution.py:49-63,66-94; tools/compute_ConfusionMatrix.py:54-66,68-98)
# ------------------------------------------------------------------------------------------------------------
def label_mapping(inp, mapping):
    """output[input == src] = dst for every (src, dst) row, applied in order on a copy (evaluate_cityscapes.py:90-94 and both tools)."""
    out = np.copy(inp)
    for src, dst in mapping:
        out[inp == src] = dst
    return np.array(out, dtype=np.int64)


def class_hist(pred, n):
    """compute_ClassDistribution.py:49-51 `fast_hist(a, n)`: counts of the values in [0, n)."""
    k = (pred >= 0) & (pred < n)
    return np.bincount(pred[k], minlength=n)


def class_distribution(preds, n=19):
    """compute_CD + the normalisation of :88-92: sum of per-image histograms / (total + 10e-10) -> what ClassDist_*.npy hold."""
    cm = np.zeros(n)
    for p in preds:
        cm += class_hist(np.asarray(p).flatten(), n)
    return cm, cm / (np.sum(cm) + 10e-10)


def rect_hist(gt, pred, n_rows, n_cols):
    """compute_ConfusionMatrix.py:54-56 `fast_hist(a, b, n33, n19)`: rows = (mapped) ground-truth ids, columns = pseudo labels."""
    k = (gt >= 0) & (gt < n_rows)
    return np.bincount(n_cols * gt[k].astype(int) + pred[k], minlength=n_rows * n_cols).reshape(n_rows, n_cols)
