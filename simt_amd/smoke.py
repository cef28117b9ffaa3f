"""__graft_entry__.smoke(): one small SimT training iteration on cuda:0 through the C ABI, checked against the CPU
oracle (the only place, besides tests/ and bench.py's cpu_baseline, where oracle/ is imported)."""
import numpy as np
import torch


def run():
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs a GPU: the SimT hot path has no CPU fallback")
    from oracle import simt_oracle as so
    from simt_amd import _lib
    from simt_amd.step import Hyper, SimTTrainer
    _lib.load()
    dev = torch.device("cuda:0")
    layers, K, B, H, W = (1, 1, 2, 1), 3, 2, 97, 97
    cd = so.load_class_dist()
    st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=11, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=12, head_scale=8.0)
    img, lab = so.synthetic_batch(B, H, W, cd.numpy(), seed=5, block=8)
    hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    tr = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, cd.numpy(), B, H, W, dtype=torch.float32,
                     device=dev, layers=layers)
    tr.step(img.to(dev), lab.to(dev), 0)
    got = tr.lout.cpu().double().numpy()[:9]
    # oracle: the same iteration on the CPU
    ohp = so.Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    n = [so.ntm_init(19, K, 1).requires_grad_(True), so.ntm_init(19, K, 2).requires_grad_(True)]
    wr = [so.w_init(19, K).requires_grad_(True) for _ in range(2)]
    Q = 19 + K
    state = {"step": 0, "m1": torch.zeros(Q, Q), "v1": torch.zeros(Q, Q), "m2": torch.zeros(Q, Q), "v2": torch.zeros(Q, Q)}
    so.inner_w_loop(n[0], n[1], wr[0], wr[1], state, cd, ohp, so.lr_poly(6e-3, 0, ohp.num_steps, ohp.power))
    T1, T2 = so.sig_ntm_forward(n[0], cd, 19), so.sig_ntm_forward(n[1], cd, 19)
    with torch.no_grad():
        _, f2 = so.deeplab_multi_forward(fst, img, False, False, layers=layers)
        x1, x2 = so.deeplab_multi_forward(st, img, True, True, layers=layers)
    out = so.simt_losses(x1, x2, f2, lab, T1, T2, so.sig_w_forward(wr[0]), so.sig_w_forward(wr[1]), ohp, (H, W))
    ref = np.array([float(out[k].detach()) for k in ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex",
                                                     "volume", "anchor"]])
    err = np.abs(got - ref).max()
    print("smoke: gpu losses", np.round(got, 5).tolist())
    print("smoke: cpu oracle", np.round(ref, 5).tolist(), "max abs diff %.2e" % err)
    if not err < 5e-4:
        raise AssertionError(f"smoke parity failed: {got} vs {ref}")
    # one bf16 step of the same net must run and give finite, nearby losses
    tr16 = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, cd.numpy(), B, H, W, dtype=torch.bfloat16,
                       device=dev, layers=layers)
    tr16.step(img.to(dev), lab.to(dev), 0)
    g16 = tr16.lout.cpu().double().numpy()[:9]
    # checker for the bf16 step: the float64 bf16-storage model of both networks (a bf16 rounding wherever the HIP plans store bf16)
    with torch.no_grad():
        _, f2q = so.bf16_model_forward(fst, img, False, False, layers=layers)
        x1q, x2q = so.bf16_model_forward(st, img, True, True, layers=layers)
        outq = so.simt_losses(x1q, x2q, f2q, lab, T1.detach().double(), T2.detach().double(), so.sig_w_forward(wr[0]).detach().double(),
                              so.sig_w_forward(wr[1]).detach().double(), ohp, (H, W))
    refq = np.array([float(outq[k]) for k in ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex", "volume", "anchor"]])
    e16 = np.abs(g16 - refq) / (1 + np.abs(refq))
    print("smoke: bf16 losses", np.round(g16, 5).tolist())
    print("smoke: bf16 model ", np.round(refq, 5).tolist(), "max rel diff %.2e" % e16.max())
    if not np.all(np.isfinite(g16)) or e16.max() > 2e-2:
        raise AssertionError(f"bf16 smoke step off: {g16} vs bf16 storage model {refq}")
    print("smoke: OK (bf16 vs unrounded fp32 oracle: max abs diff %.2e)" % np.abs(g16 - ref).max())
