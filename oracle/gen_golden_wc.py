#!/usr/bin/env python3
"""Generate tests/golden/g8b_iteration_wc.npz: two full iterations of the REFERENCE (tools/trainV2_simt.py:308-436 exec'd from the
reference file, as oracle/gen_golden.py does) on a WELL-CONDITIONED state: checkpoint-like weights (oracle recipe, trained_like, head
scale 8) whose BatchNorm running statistics were calibrated by 40 train-mode forwards of the reference model itself, so that the frozen
model's posteriors spread over (0, 1) and thousands of pixels carry a confidence label (golden g8 has ONE, which is why its end-to-end
losses can only be held to the reference's own fp32 noise).  Build container only (imports /root/reference); only seeds, the calibrated
running statistics and the reference's OUTPUTS are stored.

Stored: the nine loss scalars + label count per iteration, the per-pixel Conf_label_target of both iterations, parameter samples after
each optimiser step, NTM / W after, the calibrated running statistics (inputs), and -- so that the per-pixel label comparison can be made
margin-aware instead of tolerating a flip rate -- the frozen model's max posterior per pixel (`pmax`: the two thresholds of :358-361 cut
it) and the top-1 minus top-2 gap of the trainable model's upsampled main-head logits (`gap2`: the open-class arg-max of :387-393).
Usage: python oracle/gen_golden_wc.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden as gg  # noqa: E402
from oracle import simt_oracle as so  # noqa: E402

H = W = 129
K = 3
N_IT = 2
SAMPLE_KEYS = ["layer3.5.conv2.weight", "layer4.0.downsample.0.weight", "layer6.conv2d_list.1.weight",
               "layer5_1.conv2d_list.0.bias", "layer4.2.conv3.weight", "layer3.22.conv1.weight", "layer5.conv2d_list.0.weight"]


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    train = gg.import_reference(["--open-classes", str(K), "--batch-size", "1", "--input-size-target", f"{W},{H}",
                                 "--learning-rate", "6e-4", "--learning-rate-T", "6e-3", "--lambda-Convex", "0.1",
                                 "--lambda-Volume", "1.0", "--lambda-Anchor", "1.0", "--num-steps", "250000"])
    import torch.optim as optim
    from model import deeplab_multi as dm
    args = train.args
    args.open_classes, args.batch_size, args.iter_size = K, 1, 1
    cd = np.load(os.path.join(gg.REF, "ClassDist", "ClassDist_bapa.npy"))
    shapes, fshapes = so.state_shapes(19, K, True), so.state_shapes(19, 0, False)
    model = dm.DeeplabMulti(num_classes=19, open_classes=K, openset=True)
    model.load_state_dict(so.recipe_state(shapes, seed=1234, head_scale=8.0))
    model.train()
    # calibration: the reference model's own train-mode forwards move its running statistics onto the batch statistics
    cal_img, _ = so.synthetic_batch(1, H, W, cd, seed=99, block=8)
    with torch.no_grad():
        for _ in range(40):
            model(cal_img)
    stats = {k: v.clone() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    fixed = dm.DeeplabMulti(num_classes=19)
    fsd = so.recipe_state(fshapes, seed=1234, head_scale=8.0)
    fsd.update({k: v.clone() for k, v in stats.items()})
    fixed.load_state_dict(fsd)
    fixed.eval()
    for p in fixed.parameters():
        p.requires_grad = False
    NTMs = [dm.sig_NTM(19, K), dm.sig_NTM(19, K), dm.sig_W(19, K), dm.sig_W(19, K)]
    with torch.no_grad():
        NTMs[0].NTM.copy_(so.ntm_init(19, K, 901))
        NTMs[1].NTM.copy_(so.ntm_init(19, K, 902))
    opts = [optim.Adam(m.parameters(), lr=args.learning_rate_T, weight_decay=0) for m in NTMs]
    optimizer = optim.SGD(model.optim_parameters(args), lr=args.learning_rate, momentum=args.momentum,
                          weight_decay=args.weight_decay, foreach=False)
    batches = [so.synthetic_batch(1, H, W, cd, seed=1234 + i, block=8) for i in range(N_IT)]
    ns = gg.make_ns(train, model, fixed, optimizer, NTMs, opts, batches, H, W)
    ns["net_dict"] = fixed.state_dict()
    psamples, confs, margins, pmaxs, gaps = [], [], [], [], []

    def cap(g):
        sd = model.state_dict()
        psamples.append(np.stack([np.pad(sd[k].flatten()[:64].numpy().copy(), (0, max(0, 64 - sd[k].numel()))) for k in SAMPLE_KEYS]))
        confs.append(g["Conf_label_target"].detach().reshape(1, H, W).numpy().astype(np.uint8).copy())
        top2 = g["pred2"].detach().topk(2, dim=1).values               # the open-class decision of :387-393 is an arg-max of pred2
        gaps.append((top2[:, 0] - top2[:, 1]).numpy().astype(np.float32).copy())
        return [float(g["loss"]), float(g["loss_p1"]), float(g["loss_p2"]), float(g["loss_y1"]), float(g["loss_y2"]),
                float(g["Place_loss"]), float(g["NTM_Convex_loss"]), float(g["NTM_Volume_loss"]),
                float(g["NTM_Anchor_loss"]), float((g["Conf_label_target"] != 255).sum())]
    # threshold margins of the frozen posterior, per batch (the frozen model never changes)
    for img, _lab in batches:
        with torch.no_grad():
            _, f2 = fixed(img)
            p = ns["interp_target"](torch.softmax(f2, 1)).max(1)[0]         # :354-355: soft-max at low resolution, THEN the upsample
        margins.append(float(torch.minimum((p - args.Threshold_high).abs(), (p - args.Threshold_low).abs()).min()))
        pmaxs.append(p.numpy().astype(np.float32).copy())
    traces = gg.run_reference_iterations(train, ns, N_IT, cap)
    out = dict(losses=np.array(traces), conf=np.concatenate(confs), margins=np.array(margins), H=H, K=K,
               pmax=np.concatenate(pmaxs), gap2=np.concatenate(gaps),
               ntm1=NTMs[0].NTM.detach(), ntm2=NTMs[1].NTM.detach(), w1=NTMs[2].weight.detach(), w2=NTMs[3].weight.detach(),
               sample_keys=np.array(SAMPLE_KEYS), param_samples=np.stack(psamples),
               stat_keys=np.array(list(stats.keys())), stat_values=np.concatenate([v.flatten().numpy() for v in stats.values()]))
    print("losses", np.array(traces))
    print("threshold margins", margins, "labelled pixels", [int(t[9]) for t in traces], "of", H * W)
    print("label histogram it0", np.bincount(confs[0].flatten(), minlength=256)[[*range(22), 255]])
    gg.npz("g8b_iteration_wc", **out)


if __name__ == "__main__":
    main()
