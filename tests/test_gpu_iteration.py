"""Full SimT training iterations on the GPU (SimTTrainer: inner W loop, frozen + trainable forward, fused head,
backward, SGD with duplicate listings, Adam on NTM) against the reference's golden run (tests/golden/g8_iteration.npz,
three consecutive iterations of tools/trainV2_simt.py:308-436 executed on CPU) -- fp32 parity mode, C ABI.

Tolerances: iteration 0 losses 1e-4 (north_star); iterations 1-2 go through arg-max / threshold decisions of a
101-layer net fed with accumulation-order noise (the reference itself moves by ~1e-3 between thread counts), 2e-2.
"""
import os

import numpy as np
import pytest
import torch

from oracle import simt_oracle as so
from simt_amd.step import Hyper, SimTTrainer

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CD = so.load_class_dist()


LOSS_KEYS = ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex", "volume", "anchor"]


def test_g8_three_iterations(dev):
    """The golden run has ONE pixel with a confidence label at iteration 0 (loss_p* are a CE over a single pixel of a
    101-layer train-mode-BN net): the reference's fp32 CPU losses sit up to 1.8e-3 from the float64 value.  Bar per
    scalar: |gpu - f64| <= 5 * (largest relative fp32-reference-vs-f64 error of the nine scalars) * (1 + |f64|) + 1e-4,
    and the integer decision count (pixels with a confidence label) is exact."""
    d = np.load(os.path.join(G, "g8_iteration.npz"))
    K = 3
    st = so.recipe_state(so.state_shapes(19, K, True), seed=1234, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False), seed=4321, head_scale=8.0)
    hp = Hyper(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3)
    ohp = so.Hyper(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3)
    tr = SimTTrainer(st, fst, so.ntm_init(19, K, 901), so.ntm_init(19, K, 902), hp, CD.numpy(), 1, 65, 65,
                     dtype=torch.float32, device=dev)
    truth = so.OracleTrainer(st, fst, so.ntm_init(19, K, 901), so.ntm_init(19, K, 902), ohp, CD, dtype=torch.float64)
    keys = [str(k) for k in d["sample_keys"]]
    for it in range(3):
        img, lab = so.synthetic_batch(1, 65, 65, CD.numpy(), seed=1234 + it, block=8)
        tr.step(img.to(dev), lab.to(dev), it)
        lo = tr.lout.cpu().double().numpy()[:9]
        gold = d["losses"][it][:9]
        out = truth.step(img, lab, it)
        t64 = np.array([float(out[k].detach()) for k in LOSS_KEYS])
        if it == 0:
            ref_err = np.max(np.abs(gold - t64) / (1 + np.abs(t64)))
            bound = 5 * ref_err * (1 + np.abs(t64)) + 1e-4
            assert np.all(np.abs(lo - t64) <= bound), f"gpu {lo} f64 {t64} reference-fp32 {gold}"
            assert np.all(np.abs(lo - gold) <= bound + np.abs(gold - t64))
            assert int(tr.hout[6].item()) == int(d["losses"][it][9])      # pixels with a confidence label: exact
        else:
            # The trajectory is chaotic from here on: the reference's fp32 CPU run, the same run on 1 thread and the
            # float64 run differ by up to 2.2 in `total` at iteration 1 (DESIGN.md "Parity").  The GPU must stay inside
            # that spread: |gpu - reference| <= 3 * |reference - f64| + 0.1 per scalar, label count within 10 %.
            spread = 3 * np.abs(gold - t64) + 0.1
            assert np.all(np.abs(lo - gold) <= spread), f"it {it}: gpu {lo} reference-fp32 {gold} f64 {t64}"
            n_gpu, n_ref = int(tr.hout[6].item()), int(d["losses"][it][9])
            assert abs(n_gpu - n_ref) <= 0.1 * n_ref + 2
        # terms that do not depend on the conv stack are tight at every iteration: Convex, Volume (NTM algebra only)
        np.testing.assert_allclose(lo[6:8], gold[6:8], rtol=2e-5)
        # parameters after the optimiser step.  Iteration 0: update = lr * (gradient through the ill-conditioned net),
        # bound = 3x the reference's own distance to the float64 update; later iterations: loose absolute bound.
        got = [tr.params[k].detach().flatten()[:64].cpu().numpy() for k in keys]
        gold_p = [d["param_samples"][it][i][: len(v)] for i, v in enumerate(got)]
        if it == 0:
            p64 = [truth.st[k].detach().flatten()[:64].numpy() for k in keys]
            e_ref = max(np.abs(a - b).max() for a, b in zip(gold_p, p64))
            e_gpu = max(np.abs(a - b).max() for a, b in zip(got, p64))
            print(f"params after it 0: gpu-vs-f64 {e_gpu:.2e}, reference-fp32-vs-f64 {e_ref:.2e}")
            assert e_gpu <= 3 * e_ref + 1e-6
        else:
            p64 = [truth.st[k].detach().flatten()[:64].numpy() for k in keys]
            for a, b, c64, k in zip(got, gold_p, p64, keys):
                assert np.abs(a - b).max() <= 3 * np.abs(b - c64).max() + 1e-4, f"{k} after it {it}"
    np.testing.assert_allclose(tr.ntm[0].cpu().numpy(), d["ntm1"], atol=2e-3)
    np.testing.assert_allclose(tr.wraw[0].cpu().numpy(), d["w1"], atol=1e-4 * (1 + np.abs(d["w1"]).max()))


def test_iteration_vs_oracle_batch2_bf16_sanity(dev):
    """B=2, 97x97, small trunk: fp32 GPU == oracle tightly; bf16 GPU stays within bf16 noise of it (throughput mode)."""
    layers = (1, 1, 2, 1)
    K = 3
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=11, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=12, head_scale=8.0)
    hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    img, lab = so.synthetic_batch(2, 97, 97, CD.numpy(), seed=5, block=8)
    ohp = so.Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        tr = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, CD.numpy(), 2, 97, 97, dtype=dtype,
                         device=dev, layers=layers)
        tr.step(img.to(dev), lab.to(dev), 0)
        res[dtype] = (tr.lout.cpu().double().numpy()[:9].copy(), {k: v.cpu().clone() for k, v in tr.plan.grads.items()})
    # oracle, same small trunk
    stg = {k: (v.clone().requires_grad_(True) if k in res[torch.float32][1] else v.clone()) for k, v in st.items()}
    n = [so.ntm_init(19, K, 1).requires_grad_(True), so.ntm_init(19, K, 2).requires_grad_(True)]
    wr = [so.w_init(19, K).requires_grad_(True) for _ in range(2)]
    Q = 22
    state = {"step": 0, "m1": torch.zeros(Q, Q), "v1": torch.zeros(Q, Q), "m2": torch.zeros(Q, Q), "v2": torch.zeros(Q, Q)}
    so.inner_w_loop(n[0], n[1], wr[0], wr[1], state, CD, ohp, so.lr_poly(6e-3, 0, ohp.num_steps, ohp.power))
    T1, T2 = so.sig_ntm_forward(n[0], CD, 19), so.sig_ntm_forward(n[1], CD, 19)
    with torch.no_grad():
        _, f2 = so.deeplab_multi_forward(fst, img, False, False, layers=layers)
    x1, x2 = so.deeplab_multi_forward(stg, img, True, True, layers=layers)
    out = so.simt_losses(x1, x2, f2, lab, T1, T2, so.sig_w_forward(wr[0]), so.sig_w_forward(wr[1]), ohp, (97, 97))
    out["total"].backward()
    ref = np.array([float(out[k].detach()) for k in ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex",
                                                     "volume", "anchor"]])
    np.testing.assert_allclose(res[torch.float32][0], ref, rtol=2e-4, atol=2e-4)
    # bf16 (the benchmarked dtype) against the float64 bf16-storage model of the same two networks (oracle.bf16_model_forward: a
    # bf16 rounding wherever the HIP plans store bf16): every loss within 2e-2 * (1 + |ref|); against the unrounded fp32 oracle only
    # the documented bf16 accuracy (0.1) can be claimed.
    with torch.no_grad():
        _, f2q = so.bf16_model_forward(fst, img, False, False, layers=layers)
        x1q, x2q = so.bf16_model_forward(st, img, True, True, layers=layers)
        outq = so.simt_losses(x1q, x2q, f2q, lab, T1.detach().double(), T2.detach().double(), so.sig_w_forward(wr[0]).detach().double(),
                              so.sig_w_forward(wr[1]).detach().double(), ohp, (97, 97))
    refq = np.array([float(outq[k]) for k in ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex", "volume", "anchor"]])
    print("bf16 losses", res[torch.bfloat16][0], "storage model", refq, "fp32 oracle", ref)
    assert np.all(np.abs(res[torch.bfloat16][0] - refq) <= 2e-2 * (1 + np.abs(refq))), (res[torch.bfloat16][0], refq)
    np.testing.assert_allclose(res[torch.bfloat16][0], ref, rtol=0.1, atol=0.1)
    for name, g in res[torch.float32][1].items():
        rg = stg[name].grad
        err = (g.double() - rg.double()).abs().max().item() / max(rg.abs().max().item(), 1e-30)
        assert err < 0.1, f"{name}: {err}"      # ill-conditioned quantities: see tests/test_gpu_trunk.py for the f64-anchored bound


def test_dp_hooked_backward_world1_equals_plain(dev):
    """The data-parallel code path (bucket hook + side stream + RCCL group of size 1) leaves the same parameters as the
    plain path: the bucketed replay of the backward launch list is the same launches in the same order."""
    import torch.distributed as dist
    layers, K = (1, 1, 2, 1), 3
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=11, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=12, head_scale=8.0)
    hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    img, lab = so.synthetic_batch(2, 97, 97, CD.numpy(), seed=5, block=8)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        outs = []
        for pg in (None, dist.group.WORLD):
            tr = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, CD.numpy(), 2, 97, 97,
                             dtype=torch.float32, device=dev, layers=layers, process_group=pg)
            if pg is not None:
                # only gradients the optimiser applies are exchanged: layer3 / layer4 / heads = a prefix of the flat buffer
                applied = sum(tr.plan.grad_offsets[n][1] for n in tr.sgd_names)      # spans (16-byte padded)
                assert len(tr.reducer.buckets) >= 1 and tr.reducer.buckets[-1][1] == applied < tr.plan.flat_grad.numel()
            for it in range(2):
                tr.step(img.to(dev), lab.to(dev), it)
            torch.cuda.synchronize()
            outs.append(({k: v.clone() for k, v in tr.params.items() if v.dtype != torch.long}, tr.lout.clone()))
        for k in outs[0][0]:
            assert torch.equal(outs[0][0][k], outs[1][0][k]), k
        assert torch.equal(outs[0][1], outs[1][1])
    finally:
        if created:
            dist.destroy_process_group()


def test_g11_warmup_stage_vs_reference_golden(dev):
    """Warm-up stage (tools/trainV1_warmup.py): two iterations of WarmupTrainer (fp32) against the
    reference's own run (tests/golden/g11_warmup.npz, small trunk).  Iteration 0: losses 1e-4 relative, sampled parameters
    after the duplicate-listing SGD 5e-6.  Iteration 1 runs on parameters that already went through one SGD step; there
    the reference's fp32 CPU path itself sits 1.9e-4 (conv1.weight) / 1.9e-3 (total loss) from the same computation in
    float64, so the bar is the float64 oracle: |gpu - f64| <= 3 * |reference fp32 - f64| + eps (measured: the GPU
    reproduces the float64 loss to 7 digits)."""
    from simt_amd.step import WarmupTrainer
    d = np.load(os.path.join(G, "g11_warmup.npz"))
    layers = tuple(int(x) for x in d["layers"])
    st = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=77, head_scale=8.0)
    tr = WarmupTrainer(st, Hyper(open_classes=0, lr=2.5e-4), 2, 97, 97, dtype=torch.float32, device=dev, layers=layers)
    truth = so.OracleWarmupTrainer(st, so.Hyper(open_classes=0, lr=2.5e-4), layers=layers, dtype=torch.float64)
    keys = [str(k) for k in d["sample_keys"]]
    for it in range(2):
        img, lab = so.synthetic_batch(2, 97, 97, CD.numpy(), seed=int(d["seeds"][it]), block=8)
        tr.step(img.to(dev), lab.to(dev), it)
        o64 = truth.step(img, lab, it)
        l = tr.losses()
        got = np.array([l["total"], l["loss_seg1"], l["loss_seg2"]])
        gold = d["losses"][it]
        t64 = np.array([float(o64["total"]), float(o64["loss_seg1"]), float(o64["loss_seg2"])])
        if it == 0:
            np.testing.assert_allclose(got, gold, rtol=1e-4)
        assert np.all(np.abs(got - t64) <= 3 * np.abs(gold - t64) + 1e-4 * np.abs(t64)), f"it {it}: gpu {got} f64 {t64} ref {gold}"
        for i, k in enumerate(keys):
            v = tr.params[k].detach().flatten()[:64].cpu().numpy()
            ref = d["param_samples"][it][i][: len(v)]
            p64 = truth.st[k].detach().flatten()[:64].numpy()
            if it == 0:
                np.testing.assert_allclose(v, ref, rtol=0, atol=5e-6, err_msg=f"{k} it {it}")
            assert np.abs(v - p64).max() <= 3 * np.abs(ref - p64).max() + 5e-6, f"{k} it {it}"


def test_g12_gradient_accumulation_iter_size2(dev):
    """--iter-size 2 (trainV2_simt.py:341-432) against the reference's own run (tests/golden/g12_iter_size2.npz: two
    iterations of two micro-batches, small trunk, fp32).  Iteration 0: losses of the last micro-batch 1e-4, sampled parameters
    after the step 1e-5 (lr 6e-4 x the sum of two micro-batch gradients), NTM 2e-5.  Iteration 1 (on parameters that already moved) is anchored on the float64 oracle like
    the warm-up test: |gpu - f64| <= 3 |reference fp32 - f64| + eps (eps = 2e-3 on the losses: the class-posterior CE counts
    pixels through the 0.8 / 0.2 confidence thresholds, a handful of which move with 1e-5 parameter differences)."""
    d = np.load(os.path.join(G, "g12_iter_size2.npz"))
    K, B, H, W, its = [int(v) for v in d["meta"]]
    layers = tuple(int(v) for v in d["layers"])
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=2024, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=2025, head_scale=8.0)
    kw = dict(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3, iter_size=its)
    tr = SimTTrainer(st, fst, so.ntm_init(19, K, 911), so.ntm_init(19, K, 912), Hyper(**kw), CD, B, H, W, dtype=torch.float32,
                     device=dev, layers=layers)
    truth = so.OracleTrainer(st, fst, so.ntm_init(19, K, 911), so.ntm_init(19, K, 912), so.Hyper(**kw), CD, layers=layers,
                             dtype=torch.float64)
    keys = [str(k) for k in d["sample_keys"]]
    with pytest.raises(ValueError):
        tr.step(torch.zeros(B, 3, H, W), torch.zeros(B, H, W, dtype=torch.long), 0)      # one micro-batch where two are due
    for it in range(2):
        mb = [so.synthetic_batch(B, H, W, CD.numpy(), seed=700 + its * it + j, block=8) for j in range(its)]
        tr.step([m[0].to(dev) for m in mb], [m[1].to(dev) for m in mb], it)
        o64 = truth.step([m[0] for m in mb], [m[1] for m in mb], it)
        l = tr.losses()
        got = np.array([l[k] for k in LOSS_KEYS])
        gold = d["losses"][it][:9]
        t64 = np.array([float(o64[k].detach()) for k in LOSS_KEYS])
        if it == 0:
            np.testing.assert_allclose(got, gold, rtol=1e-4, atol=1e-4)
        assert np.all(np.abs(got - t64) <= 3 * np.abs(gold - t64) + 2e-3 * np.maximum(np.abs(t64), 1.0)), f"it {it}: gpu {got} f64 {t64} ref {gold}"
        for i, k in enumerate(keys):
            v = tr.params[k].detach().flatten()[:64].cpu().numpy()
            ref = d["param_samples"][it][i][: len(v)]
            p64 = truth.st[k].detach().flatten()[:64].numpy()
            if it == 0:
                np.testing.assert_allclose(v, ref, rtol=0, atol=1e-5, err_msg=f"{k} it {it}")
            assert np.abs(v - p64).max() <= 3 * np.abs(ref - p64).max() + 5e-6, f"{k} it {it}"
        for j in range(2):
            n = tr.ntm[j].detach().cpu().numpy()
            if it == 0:
                np.testing.assert_allclose(n, d["ntm_after"][it][j], rtol=0, atol=2e-5)
            n64 = truth.ntm[j].detach().numpy()
            assert np.abs(n - n64).max() <= 3 * np.abs(d["ntm_after"][it][j] - n64).max() + 2e-5


def test_warmup_gradient_accumulation_iter_size2(dev):
    """trainV1_warmup.py:212-231 with --iter-size 2: WarmupTrainer on two micro-batches against the float64 oracle (whose
    loop body is pinned by g11; the accumulation is `loss / iter_size` + summed gradients): loss 1e-4; parameters within
    3x the distance of the same oracle run in fp32 from float64 (+5e-6)."""
    from simt_amd.step import WarmupTrainer
    layers = (1, 1, 2, 1)
    st = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=77, head_scale=8.0)
    tr = WarmupTrainer(st, Hyper(open_classes=0, lr=2.5e-4, iter_size=2), 2, 97, 97, dtype=torch.float32, device=dev, layers=layers)
    truth = so.OracleWarmupTrainer(st, so.Hyper(open_classes=0, lr=2.5e-4, iter_size=2), layers=layers, dtype=torch.float64)
    mb = [so.synthetic_batch(2, 97, 97, CD.numpy(), seed=500 + j, block=8) for j in range(2)]
    tr.step([m[0].to(dev) for m in mb], [m[1].to(dev) for m in mb], 0)
    o64 = truth.step([m[0] for m in mb], [m[1] for m in mb], 0)
    ref32 = so.OracleWarmupTrainer(st, so.Hyper(open_classes=0, lr=2.5e-4, iter_size=2), layers=layers)
    ref32.step([m[0] for m in mb], [m[1] for m in mb], 0)
    l = tr.losses()
    assert abs(l["total"] - float(o64["total"])) < 1e-4 * abs(float(o64["total"])), (l, o64)
    for k in ("conv1.weight", "layer1.0.conv2.weight", "layer3.1.conv3.weight", "layer6.conv2d_list.1.weight", "layer5.conv2d_list.0.bias"):
        v, p64 = tr.params[k].detach().cpu().double(), truth.st[k].detach()
        e_ref = (ref32.st[k].detach().double() - p64).abs().max().item()
        assert (v - p64).abs().max().item() < 3 * e_ref + 5e-6, (k, e_ref)


def test_skip_unapplied_grads_same_trajectory(dev):
    """Hyper.skip_unapplied_grads stops the backward at the input of layer3 (the SimT stage never applies the gradients of conv1 /
    layer1 / layer2): losses and EVERY parameter after two iterations are bit-identical to the full backward."""
    layers = (1, 1, 2, 1)
    K, B, H, W = 3, 2, 65, 65
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=2024, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=2025, head_scale=8.0)
    runs = []
    for skip in (False, True):
        hp = Hyper(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3, skip_unapplied_grads=skip)
        tr = SimTTrainer(st, fst, so.ntm_init(19, K, 911), so.ntm_init(19, K, 912), hp, CD, B, H, W, dtype=torch.float32, device=dev,
                         layers=layers)
        ls = []
        for it in range(2):
            img, lab = so.synthetic_batch(B, H, W, CD.numpy(), seed=700 + it, block=8)
            tr.step(img.to(dev), lab.to(dev), it)
            ls.append(tr.lout.clone())
        runs.append((ls, {k: v.clone() for k, v in tr.params.items()}, [n.clone() for n in tr.ntm], len(tr.plan.bwd_list)))
    (l0, p0, n0, len0), (l1, p1, n1, len1) = runs
    assert len1 < len0
    assert all(torch.equal(a, b) for a, b in zip(l0, l1))
    assert all(torch.equal(p0[k], p1[k]) for k in p0)
    assert all(torch.equal(a, b) for a, b in zip(n0, n1))


def test_early_sgd_and_schedule_switches_same_trajectory(dev, monkeypatch):
    """The host-side schedule options of SimTTrainer (SGD + re-pack of the applied layers on the side stream as soon as their gradients
    are final, hipGraph replay of the forwards, enqueue order of the two forwards) only move launches between streams / host calls:
    losses, EVERY parameter, momentum buffer and NTM after three bf16 iterations are bit-identical to the plain schedule."""
    layers = (1, 1, 2, 1)
    K, B, H, W = 3, 2, 65, 65
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=31, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=32, head_scale=8.0)
    runs = []
    for env in ({"SIMT_EARLY_SGD": "0", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "main"}, {"SIMT_EARLY_SGD": "1", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "side"},
                {"SIMT_EARLY_SGD": "1", "SIMT_GRAPHS": "1", "SIMT_FWD_ORDER": "main"}, {"SIMT_EARLY_SGD": "0", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "interleave"},
                {"SIMT_EARLY_SGD": "1", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "pair"},       # round 5, opt-in (the default order stays "main"): one launch per layer for both networks
                {"SIMT_EARLY_SGD": "1", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "bnside2"},
                {"SIMT_EARLY_SGD": "1", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "paced"},       # round 6: the frozen 3x3 convs released behind the trainable conv3s
                {"SIMT_EARLY_SGD": "1", "SIMT_GRAPHS": "-1", "SIMT_FWD_ORDER": "main", "SIMT_LIGHT_EVENTS": "0"}):      # torch.cuda.Event instead of the device-scope events
        monkeypatch.delenv("SIMT_LIGHT_EVENTS", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        hp = Hyper(open_classes=K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3)
        tr = SimTTrainer(st, fst, so.ntm_init(19, K, 911), so.ntm_init(19, K, 912), hp, CD, B, H, W, dtype=torch.bfloat16, device=dev,
                         layers=layers)
        assert tr._early_sgd == (env["SIMT_EARLY_SGD"] == "1")
        if env["SIMT_FWD_ORDER"] == "pair":
            tr._fwd_both = tr._paired_forwards()
            assert tr.fwd_pairs >= 10, f"only {tr.fwd_pairs} conv pairs fused"
        ls = []
        for it in range(3):
            img, lab = so.synthetic_batch(B, H, W, CD.numpy(), seed=900 + it, block=8)
            tr.step(img.to(dev), lab.to(dev), it)
            ls.append(tr.lout.clone())
        torch.cuda.synchronize()
        runs.append((ls, {k: v.clone() for k, v in tr.params.items()}, {k: v.clone() for k, v in tr.mom.items()}, [n.clone() for n in tr.ntm]))
    l0, p0, m0, n0 = runs[0]
    for l1, p1, m1, n1 in runs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(l0, l1))
        assert all(torch.equal(p0[k], p1[k]) for k in p0)
        assert all(torch.equal(m0[k], m1[k]) for k in m0)
        assert all(torch.equal(a, b) for a, b in zip(n0, n1))


def test_out_of_range_label_is_reported(dev):
    """A noisy label that is neither a class nor the ignore value: the reference's nll_loss raises "Target out of bounds"
    (utils/loss.py:36-40 via F.nll_loss; trainV2_simt.py:408).  The kernels skip the pixel and count it; reading the losses raises."""
    layers, K = (1, 1, 1, 1), 3
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=11, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=12, head_scale=8.0)
    tr = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), Hyper(open_classes=K), CD.numpy(), 1, 65, 65,
                     dtype=torch.float32, device=dev, layers=layers)
    img, lab = so.synthetic_batch(1, 65, 65, CD.numpy(), seed=5, block=8)
    tr.step(img.to(dev), lab.to(dev), 0)
    assert np.isfinite(tr.losses()["total"])
    lab[0, 3, 4] = 19
    lab[0, 7, 7] = 200
    tr.step(img.to(dev), lab.to(dev), 1)
    with pytest.raises(ValueError, match="2 label value"):
        tr.losses()
    # the count accumulates over steps that are not read and over EVERY micro-batch (the check used to see the last one only);
    # raising clears it
    lab_ok = lab.clone()
    lab_ok[lab_ok == 19] = 0
    lab_ok[lab_ok == 200] = 255
    tr.step(img.to(dev), lab_ok.to(dev), 2)
    assert np.isfinite(tr.losses()["total"])
    tr.step(img.to(dev), lab.to(dev), 3)              # bad labels in a step whose losses nobody reads ...
    tr.step(img.to(dev), lab_ok.to(dev), 4)           # ... followed by a clean one
    with pytest.raises(ValueError, match="2 label value"):
        tr.losses()
    tr2 = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), Hyper(open_classes=K, iter_size=2), CD.numpy(), 1, 65, 65,
                      dtype=torch.float32, device=dev, layers=layers)
    tr2.step([img.to(dev), img.to(dev)], [lab.to(dev), lab_ok.to(dev)], 0)      # bad labels in the FIRST micro-batch only
    with pytest.raises(ValueError, match="2 label value"):
        tr2.losses()


@pytest.mark.parametrize("scope", [0, 2, 1])
def test_device_scope_events_order_two_streams_under_load(dev, scope):
    """the launch lists order their two HIP streams with the library's events (simt_event_create): scope 0 = hipEventReleaseToDevice
    (the documented device-scope release, default since round 6), 2 = hipEventDisableSystemFence (round 5's form), 1 = system scope.  A
    producer / consumer stress with LARGE buffers (64 MB: far beyond the L2, so stale lines would show) and many iterations: stream A writes a new
    pattern, records; stream B waits, copies it out and checksums; B records, A waits before it overwrites (write-after-read).  Every iteration's
    checksum must be exactly that iteration's pattern."""
    import ctypes as C
    from simt_amd import _lib as L
    n = 16 << 20                                   # 16 M fp32 = 64 MB
    a, b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    buf = torch.zeros(n, device=dev)
    out = torch.zeros(n, device=dev)
    sums = torch.zeros(200, device=dev, dtype=torch.float64)
    ev_w, ev_r = C.c_void_p(), C.c_void_p()
    L.call("simt_event_create", C.byref(ev_w), scope)
    L.call("simt_event_create", C.byref(ev_r), scope)
    torch.cuda.synchronize()
    try:
        for it in range(200):
            with torch.cuda.stream(a):
                if it:
                    L.call("simt_stream_wait_event", a.cuda_stream, ev_r)      # B has read the previous pattern
                buf.fill_(float(it + 1))
                buf[::4097] += 0.5                                                 # (a second kernel on the same buffer)
                L.call("simt_event_record", ev_w, a.cuda_stream)
            with torch.cuda.stream(b):
                L.call("simt_stream_wait_event", b.cuda_stream, ev_w)
                out.copy_(buf)
                sums[it] = out.sum(dtype=torch.float64)
                L.call("simt_event_record", ev_r, b.cuda_stream)
        torch.cuda.synchronize()
        k = len(range(0, n, 4097))
        want = torch.tensor([(it + 1.0) * n + 0.5 * k for it in range(200)], dtype=torch.float64)
        assert torch.equal(sums.cpu(), want), f"scope {scope}: first bad iteration {int((sums.cpu() != want).nonzero()[0])}"
    finally:
        L.call("simt_event_destroy", ev_w)
        L.call("simt_event_destroy", ev_r)
