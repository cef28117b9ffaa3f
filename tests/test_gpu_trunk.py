"""GPU parity of the DeepLab-v2 ResNet trunk + ASPP heads (TrunkPlan: implicit-GEMM convs, train-mode BN with frozen
affine, ceil-mode max-pool, fused 2-branch heads) in fp32 parity mode, through the C ABI.

Three kinds of check:
  * block level (well conditioned): every Bottleneck of a small trunk is fed, in the CPU oracle, with the GPU's own
    block input -> forward intermediates, input gradient and weight gradients agree to 2e-5 of max|ref|.
  * eval-mode end-to-end vs the reference's golden logits (g7): 2e-5.
  * train-mode end-to-end: a 101-layer net with batch-statistics BN over 81 samples is ILL-CONDITIONED in fp32 -- the
    reference's own CPU fp32 path sits 3e-4 (logits) / up to 20 % (some trunk weight gradients) away from the same
    computation in float64, and moves by 11 % between 1 and 8 threads (measured, DESIGN.md "Parity").  So the bar is
    "as close to exact (float64 oracle) as the reference's fp32 path is": err_gpu <= 4 * err_ref32 + 2e-5 per
    tensor, and |gpu - golden| <= err_gpu + err_ref32.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import simt_oracle as so
from simt_amd.engine import LaunchList, TrunkPlan, multi_heads, single_head

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = a.double(), b.double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


def to_nchw(t, Cn):
    return t[..., :Cn].permute(0, 3, 1, 2).float().cpu()


def set_dlogits(plan, name, up):
    d = plan.dlogits[name]
    Cn = up.shape[1]
    d.zero_()
    d[:, :Cn] = up.permute(0, 2, 3, 1).reshape(-1, Cn).to(d.device, d.dtype)


def oracle_run(st, img, ups, dt, layers, grad_keys):
    stg = {k: (v.clone().to(dt).requires_grad_(True) if k in grad_keys else (v.clone().to(dt) if v.dtype != torch.long else v.clone()))
           for k, v in st.items()}
    t1, t2 = so.deeplab_multi_forward(stg, img.to(dt), True, True, layers=layers)
    ((t1 * ups[0].to(dt)).sum() + (t2 * ups[1].to(dt)).sum()).backward()
    return t1.detach(), t2.detach(), {k: stg[k].grad for k in grad_keys}, stg


def check_conditioned(name, got, ref32, truth, factor=4.0, floor=2e-5):
    e_gpu, e_ref = rel(got, truth), rel(ref32, truth)
    assert e_gpu <= factor * e_ref + floor, f"{name}: gpu-vs-f64 {e_gpu:.2e} but reference-fp32-vs-f64 only {e_ref:.2e}"
    return e_gpu, e_ref


def check_population(tag, e_gpu, e_ref, factor=3.0, floor=2e-5):
    """Errors against float64 of many tensors: a single tensor can be hit by a discrete ReLU-mask flip in either
    implementation, so the GPU is compared with the reference's fp32 path on the distribution (median and maximum)."""
    e_gpu, e_ref = np.asarray(e_gpu), np.asarray(e_ref)
    print(f"{tag}: gpu-vs-f64 median {np.median(e_gpu):.2e} max {e_gpu.max():.2e} | "
          f"reference-fp32-vs-f64 median {np.median(e_ref):.2e} max {e_ref.max():.2e}")
    assert np.median(e_gpu) <= factor * np.median(e_ref) + floor
    assert e_gpu.max() <= factor * e_ref.max() + floor


def _oracle_block(st, rec, xin, dz):
    """Bottleneck forward/backward on the CPU (reference semantics, model/deeplab_multi.py:81-101) keeping a1, a2."""
    name = rec["name"]
    sd = {k: (v.clone().requires_grad_(True) if ("conv" in k or "downsample.0" in k) else v.clone())
          for k, v in st.items() if k.startswith(name + ".")}
    a1 = F.relu(so._bn(sd, f"{name}.bn1", F.conv2d(xin, sd[f"{name}.conv1.weight"], stride=rec["stride"]), True))
    a2 = F.relu(so._bn(sd, f"{name}.bn2", F.conv2d(a1, sd[f"{name}.conv2.weight"], padding=rec["dil"], dilation=rec["dil"]), True))
    out = so._bn(sd, f"{name}.bn3", F.conv2d(a2, sd[f"{name}.conv3.weight"]), True)
    sc = xin
    if rec["down"]:
        sc = so._bn(sd, f"{name}.downsample.1", F.conv2d(xin, sd[f"{name}.downsample.0.weight"], stride=rec["stride"]), True)
    z = F.relu(out + sc)
    z.backward(dz)
    return a1.detach(), a2.detach(), z.detach(), sd


def test_block_level_parity_every_bottleneck(dev):
    """Each Bottleneck, fed with the GPU's own block input: forward 2e-5; backward 2e-5 whenever the three ReLU masks
    of the block agree between GPU and CPU.  (An element whose pre-activation is within fp32 noise of zero flips its
    mask -- a discrete event that moves a whole low-variance BN channel; blocks where that happens are held to a
    relative-L2 bound instead and at most one such block is tolerated.)"""
    layers = (1, 1, 2, 1)
    st = so.recipe_state(so.state_shapes(19, 3, True, layers=layers), seed=77)
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 65, 81
    img = torch.randn(B, 3, H, W, generator=g) * 50
    p = {k: v.clone().to(dev) for k, v in st.items()}
    tr = TrunkPlan(p, B, H, W, multi_heads(19, 3, True), dtype=torch.float32, train=True, layers=layers)
    tr.forward(img.to(dev))
    torch.cuda.synchronize()
    # stem: conv 7x7 s2 -> BN(train) -> ReLU -> ceil-mode max-pool
    x = F.conv2d(img, st["conv1.weight"], stride=2, padding=3)
    x = F.max_pool2d(F.relu(so._bn({k: v.clone() for k, v in st.items()}, "bn1", x, True)), 3, 2, 1, ceil_mode=True)
    assert rel(tr.saved["stem.pool"].view(B, tr.Hp, tr.Wp, 64).permute(0, 3, 1, 2).cpu(), x) < 2e-5
    flipped = []
    for rec in tr.block_io:
        name = rec["name"]

        def c(t, h=rec["Ho"], w=rec["Wo"]):
            return t.view(B, h, w, -1).permute(0, 3, 1, 2).float().cpu()
        xin = c(rec["x"], rec["Hi"], rec["Wi"]).requires_grad_(True)
        dz = torch.randn(B, rec["planes"] * 4, rec["Ho"], rec["Wo"], generator=g)
        a1, a2, z, sd = _oracle_block(st, rec, xin, dz)
        for got, ref, what in ((rec["a1"], a1, "a1"), (rec["a2"], a2, "a2"), (rec["z"], z, "z")):
            assert rel(c(got), ref) < 2e-5, f"{name} {what}"
        nflip = sum(int(((c(got) > 0) != (ref > 0)).sum()) for got, ref in ((rec["a1"], a1), (rec["a2"], a2), (rec["z"], z)))
        start, end, dzb, dxb = tr.bwd_marks[name]
        dzb.copy_(dz.permute(0, 2, 3, 1).reshape(dzb.shape).to(dev))
        sub = LaunchList()
        sub.items = tr.bwd_list.items[start:end]
        sub.run()
        torch.cuda.synchronize()
        pairs = [(f"{name} dx", c(dxb, rec["Hi"], rec["Wi"]), xin.grad)]
        pairs += [(k, tr.grads[k].cpu(), v.grad) for k, v in sd.items() if v.requires_grad]
        if nflip == 0:
            for what, got, ref in pairs:
                assert rel(got, ref) < 2e-5, what
        else:
            flipped.append((name, nflip))
            for what, got, ref in pairs:
                l2 = ((got.double() - ref.double()).norm() / ref.double().norm()).item()
                assert l2 < 5e-2, f"{what}: relative L2 {l2} with {nflip} flipped mask elements"
    print("blocks with a flipped ReLU mask element:", flipped)
    assert len(flipped) <= 1


def test_g7_eval_and_train_forward_backward(dev):
    d = np.load(os.path.join(G, "g7_deeplab_multi.npz"))
    st = so.recipe_state(so.state_shapes(19, 3, True), seed=1234)
    img = torch.as_tensor(d["img"])
    B, _, H, W = img.shape
    # ---- eval (BN folded into the packed weights): well conditioned -> tight against the reference's own output
    pe = {k: v.clone().to(dev) for k, v in st.items()}
    ev = TrunkPlan(pe, B, H, W, multi_heads(19, 3, True), dtype=torch.float32, train=False)
    out = ev.forward(img.to(dev))
    torch.cuda.synchronize()
    assert rel(to_nchw(out["x1"], 22), torch.as_tensor(d["eval_x1"])) < 2e-5
    assert rel(to_nchw(out["x2"], 22), torch.as_tensor(d["eval_x2"])) < 2e-5
    # ---- train forward + backward
    pt = {k: v.clone().to(dev) for k, v in st.items()}
    tr = TrunkPlan(pt, B, H, W, multi_heads(19, 3, True), dtype=torch.float32, train=True)
    out = tr.forward(img.to(dev))
    ups = (torch.as_tensor(d["up1"]), torch.as_tensor(d["up2"]))
    set_dlogits(tr, "x1", ups[0])
    set_dlogits(tr, "x2", ups[1])
    grads = tr.backward()
    torch.cuda.synchronize()
    names = [str(n) for n in d["grad_names"]]
    assert sorted(names) == sorted(grads.keys())          # the same 120 tensors receive gradients
    assert torch.all(out["x1"][..., 22:] == 0)
    r1, r2, g32, stg32 = oracle_run(st, img, ups, torch.float32, so.LAYERS, set(names))
    t1, t2, g64, _ = oracle_run(st, img, ups, torch.float64, so.LAYERS, set(names))
    for nm, got, r, t, gold in (("x1", to_nchw(out["x1"], 22), r1, t1, d["train_x1"]), ("x2", to_nchw(out["x2"], 22), r2, t2, d["train_x2"])):
        e_gpu, e_ref = check_conditioned(nm, got, r, t)
        assert rel(got, torch.as_tensor(gold)) <= e_gpu + e_ref + 2e-5
    check_population("g7 gradients (120 tensors)", [rel(grads[n].cpu(), g64[n]) for n in names],
                     [rel(g32[n], g64[n]) for n in names])
    # the reference's own numbers (fp32 CPU): within the sum of both distances to float64
    for i, n in enumerate(names):
        ref_abs = float(d["grad_abssum"][i])
        tol = rel(grads[n].cpu(), g64[n]) + rel(g32[n], g64[n]) + 1e-4
        assert abs(grads[n].double().abs().sum().item() - ref_abs) <= 3 * tol * ref_abs + 1e-6, n
    # running statistics after one train-mode forward (reference values; early layers are well conditioned)
    for k in d.files:
        if k.startswith("after."):
            assert rel(pt[k[6:]].float().cpu(), torch.as_tensor(d[k]).float()) < 1e-3, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_small_trunk_batch2_nonsquare(dev, dtype):
    """layers=(1,1,2,1), B=2, 97x129: stride-2 block, dilations 2/4, ragged tiles.  bf16 (the throughput mode) is held
    to a loose sanity bound only; parity claims are made in fp32."""
    layers = (1, 1, 2, 1)
    st = so.recipe_state(so.state_shapes(19, 3, True, layers=layers), seed=77)
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 97, 129
    img = torch.randn(B, 3, H, W, generator=g) * 50
    p = {k: v.clone().to(dev) for k, v in st.items()}
    tr = TrunkPlan(p, B, H, W, multi_heads(19, 3, True), dtype=dtype, train=True, layers=layers)
    out = tr.forward(img.to(dev))
    h, w = tr.heads[0].h, tr.heads[0].w
    ups = (torch.randn(B, 22, h, w, generator=g), torch.randn(B, 22, h, w, generator=g))
    set_dlogits(tr, "x1", ups[0])
    set_dlogits(tr, "x2", ups[1])
    grads = tr.backward()
    torch.cuda.synchronize()
    keys = set(grads.keys())
    r1, r2, g32, _ = oracle_run(st, img, ups, torch.float32, layers, keys)
    t1, t2, g64, _ = oracle_run(st, img, ups, torch.float64, layers, keys)
    if dtype == torch.float32:
        check_conditioned("x1", to_nchw(out["x1"], 22), r1, t1)
        check_conditioned("x2", to_nchw(out["x2"], 22), r2, t2)
        ks = sorted(keys)
        check_population("small trunk gradients", [rel(grads[n].cpu(), g64[n]) for n in ks], [rel(g32[n], g64[n]) for n in ks])
    else:
        assert rel(to_nchw(out["x1"], 22), t1) < 0.1 and rel(to_nchw(out["x2"], 22), t2) < 0.1
        cos = [F.cosine_similarity(grads[n].cpu().double().flatten(), g64[n].flatten(), dim=0).item() for n in sorted(keys)]
        print("bf16 gradient cosine vs f64: min %.3f median %.4f" % (min(cos), float(np.median(cos))))
        assert min(cos) > 0.8 and np.median(cos) > 0.97


def test_single_head_deeplab_py(dev):
    """model/deeplab.py: 4-branch ASPP on layer4 (all of dilations 6/12/18/24 summed)."""
    layers = (1, 1, 1, 1)
    st = so.recipe_state(so.state_shapes(19, single_head=True, layers=layers), seed=5)
    g = torch.Generator().manual_seed(4)
    img = torch.randn(1, 3, 65, 81, generator=g) * 50
    p = {k: v.clone().to(dev) for k, v in st.items()}
    ev = TrunkPlan(p, 1, 65, 81, single_head(19), dtype=torch.float32, train=False, layers=layers)
    out = ev.forward(img.to(dev))
    y, _ = so.deeplab_single_forward(st, img, False, layers=layers)
    assert rel(to_nchw(out["x"], 19), y) < 2e-5
