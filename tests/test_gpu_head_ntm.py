"""GPU parity of the fused SimT head + NTM micro-solver (C ABI: simt_ntm_inner_loop, simt_head_loss, simt_ntm_post,
simt_head_grad, simt_sig_ntm, simt_sig_w, simt_adam_step) against the golden vectors produced by the reference
(tests/golden/g1_g2*, g4_*) and against the CPU oracle on the same inputs.

Tolerance: fp32 throughout.  Scalars 1e-4 (north_star: "loss within 1e-4 fp32"); gradients 1e-5 relative to max|ref|;
integer decisions (confidence labels are not exported by the kernel, but N_valid counts are) exact.
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import simt_oracle as so
from simt_amd import _lib as L
from simt_amd import ops

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
CD = so.load_class_dist()

HEAD_CASES = ["g4_head_base_k3", "g4_head_open_negative_k3", "g4_head_ties_k3", "g4_head_open_wins_k3",
              "g4_head_lowconf_k3", "g4_head_no_valid_k3", "g4_head_base_k15", "g4_head_base_k6"]


def nhwc_pad(x, ld):
    """[B,C,h,w] -> [B*h*w, ld] fp32 (zero padded)."""
    B, Cn, h, w = x.shape
    out = torch.zeros(B * h * w, ld)
    out[:, :Cn] = x.permute(0, 2, 3, 1).reshape(-1, Cn)
    return out


def run_head(dev, d, pred1, pred2, fixed2, label, ntm, *, steps=10, lr_T=None, label_ws=False):
    """Drives the head/NTM kernels exactly like SimTTrainer.step does, on explicit low-res logits."""
    K = int(d["K"]); Cn = 19; Q = Cn + K
    B, _, h, w = pred1.shape
    H, W = label.shape[1:]
    lam = d["lam"]
    lib = L.load()
    st = ops.stream_ptr()
    ldp = max(32, ops.round_up(Q, 8))
    p1 = nhwc_pad(pred1, ldp).to(dev); p2 = nhwc_pad(pred2, ldp).to(dev)
    fl = nhwc_pad(fixed2, 32).to(dev)
    fixp = torch.zeros_like(fl)
    ops.softmax_rows(fl, 32, fixp, 32, B * h * w, Cn)
    lab = label.to(dev)
    cd = CD.float().to(dev)
    ntm_d = [n.clone().to(dev) for n in ntm]
    ngrad = [torch.zeros(Q, Cn, device=dev) for _ in range(2)]
    wraw = [so.w_init(Cn, K).to(dev) for _ in range(2)]
    wm = [torch.zeros(Q, Q, device=dev) for _ in range(2)]
    wv = [torch.zeros(Q, Q, device=dev) for _ in range(2)]
    T = [torch.zeros(Q, Cn, device=dev) for _ in range(2)]
    ni = L.NtmInnerDesc()
    for k in range(2):
        ni.ntm[k], ni.w[k], ni.ntm_grad[k] = ntm_d[k].data_ptr(), wraw[k].data_ptr(), ngrad[k].data_ptr()
        ni.w_m[k], ni.w_v[k], ni.T_out[k] = wm[k].data_ptr(), wv[k].data_ptr(), T[k].data_ptr()
    ni.class_dist, ni.Q, ni.C, ni.steps, ni.step0 = cd.data_ptr(), Q, Cn, steps, 0
    ni.lr, ni.beta1, ni.beta2, ni.eps = float(d["lr_T"]) if lr_T is None else lr_T, 0.9, 0.999, 1e-8
    L.call("simt_ntm_inner_loop", C.byref(ni), st)

    nblk = lib.simt_head_nblk(B, H, W)
    part = torch.zeros(nblk, lib.simt_head_part_floats(Q, Cn), device=dev)
    keys = torch.zeros(lib.simt_head_keys_count(), device=dev, dtype=torch.int64)
    hout = torch.zeros(lib.simt_head_hout_floats(Q, Cn), device=dev)
    lout = torch.zeros(16, device=dev)
    QP = ops.round_up(Q, 8)
    g1 = torch.zeros(2, B, H, w, QP, device=dev)
    dp1 = torch.zeros(B * h * w, ldp, device=dev)
    dp2 = torch.zeros(B * h * w, ldp, device=dev)
    hd = L.HeadDesc()
    hd.pred1, hd.pred2, hd.fixp, hd.label = p1.data_ptr(), p2.data_ptr(), fixp.data_ptr(), lab.data_ptr()
    hd.T1, hd.T2 = T[0].data_ptr(), T[1].data_ptr()
    hd.part, hd.keys, hd.hout, hd.g1 = part.data_ptr(), keys.data_ptr(), hout.data_ptr(), g1.data_ptr()
    hd.dpred1_f32, hd.dpred2_f32, hd.dpred1_t, hd.dpred2_t = dp1.data_ptr(), dp2.data_ptr(), None, None
    hd.B, hd.h, hd.w, hd.H, hd.W, hd.C, hd.Q = B, h, w, H, W, Cn, Q
    hd.ldp, hd.ldf, hd.QP, hd.ld_f32, hd.ld_t, hd.grad_dtype = ldp, 32, QP, ldp, 0, L.SIMT_F32
    hd.th_high, hd.th_low = float(d["th"][0]), float(d["th"][1])
    hd.lambda_seg, hd.lambda_place, hd.gscale = float(d["lambda_seg"]), float(d["lambda_place"]), 1.0
    conf = torch.full((B, H, W), 77, dtype=torch.uint8, device=dev)       # per-pixel Conf_label_target (optional export)
    hd.conf_out = conf.data_ptr()
    if label_ws:          # the trainers' form: the gradient pass reads both byte maps back instead of deciding the labels again
        lws = torch.full((B, H, W), 77, dtype=torch.uint8, device=dev)
        hd.label_ws = lws.data_ptr()
    L.call("simt_head_loss", C.byref(hd), st)
    npd = L.NtmPostDesc()
    for k in range(2):
        npd.ntm[k], npd.w[k], npd.ntm_grad[k] = ntm_d[k].data_ptr(), wraw[k].data_ptr(), ngrad[k].data_ptr()
    npd.class_dist, npd.hout, npd.lout, npd.Q, npd.C = cd.data_ptr(), hout.data_ptr(), lout.data_ptr(), Q, Cn
    npd.lambda_seg, npd.lambda_convex, npd.lambda_volume, npd.lambda_anchor = float(d["lambda_seg"]), float(lam[0]), float(lam[1]), float(lam[2])
    npd.gscale = 1.0
    L.call("simt_ntm_post", C.byref(npd), st)
    L.call("simt_head_grad", C.byref(hd), st)
    # Adam on NTM (first step)
    ntm_after = []
    for k in range(2):
        p = ntm_d[k].clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        ops.adam_step(p, ngrad[k], m, v, lr=float(d["lr_T"]), step=1)
        ntm_after.append(p)
    torch.cuda.synchronize()

    def back(g):  # [B*h*w, ldp] -> [B,Q,h,w]
        return g.cpu()[:, :Q].reshape(B, h, w, Q).permute(0, 3, 1, 2)
    return dict(lout=lout.cpu(), hout=hout.cpu(), dpred1=back(dp1), dpred2=back(dp2), dp1_raw=dp1.cpu(), conf=conf.cpu().long(),
                ntm_grad=[g.cpu() for g in ngrad], w=[x.cpu() for x in wraw], wm=[x.cpu() for x in wm],
                wv=[x.cpu() for x in wv], T=[x.cpu() for x in T], ntm_after=[x.cpu() for x in ntm_after])


def close(a, b, tol, what=""):
    a, b = torch.as_tensor(np.asarray(a)).double(), torch.as_tensor(np.asarray(b)).double()
    if torch.isnan(b).any():
        assert torch.equal(torch.isnan(a), torch.isnan(b)), f"{what}: NaN pattern differs"
        a, b = torch.nan_to_num(a), torch.nan_to_num(b)
    err = (a - b).abs().max().item()
    assert err <= tol * (1 + b.abs().max().item()), f"{what}: err {err} (ref max {b.abs().max().item()})"


@pytest.mark.parametrize("name", HEAD_CASES)
def test_head_against_reference_golden(dev, name):
    d = np.load(os.path.join(G, name + ".npz"))
    t = lambda k: torch.as_tensor(d[k])
    r = run_head(dev, d, t("pred_lr1"), t("pred_lr2"), t("fixed_lr2"), t("label"), [t("ntm1"), t("ntm2")])
    lo = r["lout"]
    for idx, key in [(0, "loss"), (1, "loss_p1"), (2, "loss_p2"), (3, "loss_y1"), (4, "loss_y2"), (5, "place"),
                     (6, "convex"), (7, "volume"), (8, "anchor")]:
        close(lo[idx], d[key], 1e-4, key)
    # number of pixels with a confidence label: integer-exact
    assert int(r["hout"][6]) == int((d["conf"] != 255).sum())
    # ... and the label itself, pixel by pixel (trainV2_simt.py:357-362,387-393): bit-exact against the reference's Conf_label_target
    assert torch.equal(r["conf"], t("conf").long()), f"{int((r['conf'] != t('conf')).sum())} confidence labels differ"
    close(r["dpred1"], d["dpred1"], 1e-5, "dpred1")
    close(r["dpred2"], d["dpred2"], 1e-5, "dpred2")
    Q = 19 + int(d["K"])
    assert torch.all(r["dp1_raw"][:, Q:] == 0)
    close(r["w"][0], d["w1_after"], 1e-5, "w1 after 10 Adam steps")
    close(r["w"][1], d["w2_after"], 1e-5, "w2")
    close(r["wm"][0], d["w1_m"], 1e-5, "exp_avg")
    close(r["wv"][0], d["w1_v"], 1e-5, "exp_avg_sq")
    close(r["ntm_grad"][0], d["ntm_grad1"], 2e-5, "ntm_grad1 (leak + main)")
    close(r["ntm_grad"][1], d["ntm_grad2"], 2e-5, "ntm_grad2")
    close(r["ntm_after"][0], d["ntm1_after"], 1e-5, "NTM1 after Adam")
    close(r["ntm_after"][1], d["ntm2_after"], 1e-5, "NTM2 after Adam")


# (B, h, w, H, W): pass 2's x-reduction takes a different route per geometry -- runs of <= 8 pixels per low-res column (the production
# 8x upsample: all terms of a run in flight), 9..12, longer (looped), and more low-res columns than a 256-pixel chunk's run table
# holds (logits WIDER than the image: the scanning form)
# rows4 / rows6: enough image rows that pass 2 takes groups of 4 / 6 rows per block and folds them along y in registers (the production form;
# head_rows_per_block in csrc/head_loss.hip: needs B * H / rows >= 512)
HEAD_GEOMS = {"up8_two_chunks": (2, 13, 37, 97, 289), "up10": (1, 3, 13, 16, 128), "up16_looped": (1, 3, 9, 16, 144),
              "down_scanning": (1, 3, 330, 8, 272), "rows4": (4, 65, 4, 512, 24), "rows6": (4, 97, 3, 768, 16)}
# K = 3 and 6 run the builds with compile-time channel counts (Q = 22, 25); K = 4 the run-time-count build of the same width (Q = 23),
# K = 15 the wide one (Q = 34)
HEAD_KS = [("up8_two_chunks", 3, False), ("up10", 3, False), ("up16_looped", 3, True), ("down_scanning", 3, False), ("up8_two_chunks", 4, True),
           ("up10", 6, True), ("up10", 15, False), ("rows4", 3, False), ("rows4", 6, True), ("rows6", 3, True), ("rows6", 3, False)]


@pytest.mark.parametrize("geom,K,lws", HEAD_KS)
def test_head_bigger_than_one_block_vs_oracle(dev, geom, K, lws):
    """Head kernels against the oracle on explicit low-res logits: several blocks per pass, W > 256 (two x-chunks in pass 2),
    non-square, every route of the x-reduction, every instantiation of the kernels."""
    Cn = 19
    Q = Cn + K
    g = torch.Generator().manual_seed(5)
    B, h, w, H, W = HEAD_GEOMS[geom]
    p1 = torch.randn(B, Q, h, w, generator=g) * 3
    p2 = torch.randn(B, Q, h, w, generator=g) * 3
    f2 = torch.randn(B, Cn, h, w, generator=g) * 4
    _, lab = so.synthetic_batch(B, H, W, CD.numpy(), seed=11, block=8)
    ntm = [so.ntm_init(Cn, K, 1), so.ntm_init(Cn, K, 2)]
    d = {"K": K, "lam": np.array([0.5, 0.1, 0.5]), "lr_T": 6e-3, "th": np.array([0.8, 0.2]), "lambda_seg": 0.1,
         "lambda_place": 0.1}
    r = run_head(dev, d, p1, p2, f2, lab, ntm, label_ws=lws)
    # oracle
    hp = so.Hyper(num_classes=Cn, open_classes=K, lambda_convex=0.5, lambda_volume=0.1, lambda_anchor=0.5)
    n = [x.clone().requires_grad_(True) for x in ntm]
    wr = [so.w_init(Cn, K).requires_grad_(True) for _ in range(2)]
    state = {"step": 0, "m1": torch.zeros(Q, Q), "v1": torch.zeros(Q, Q), "m2": torch.zeros(Q, Q), "v2": torch.zeros(Q, Q)}
    so.inner_w_loop(n[0], n[1], wr[0], wr[1], state, CD, hp, 6e-3)
    q1, q2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    T1, T2 = so.sig_ntm_forward(n[0], CD, Cn), so.sig_ntm_forward(n[1], CD, Cn)
    out = so.simt_losses(q1, q2, f2, lab, T1, T2, so.sig_w_forward(wr[0]), so.sig_w_forward(wr[1]), hp, (H, W))
    out["total"].backward()
    lo = r["lout"]
    for idx, key in [(0, "total"), (1, "loss_p1"), (2, "loss_p2"), (3, "loss_y1"), (4, "loss_y2"), (5, "place"),
                     (6, "convex"), (7, "volume"), (8, "anchor")]:
        close(lo[idx], out[key].detach(), 1e-4, key)
    assert int(r["hout"][6]) == int((out["conf"] != 255).sum())
    assert torch.equal(r["conf"], out["conf"].long().view_as(r["conf"])), f"{int((r['conf'] != out['conf'].view_as(r['conf'])).sum())} labels differ"
    close(r["dpred1"], q1.grad, 1e-5, "dpred1")
    close(r["dpred2"], q2.grad, 1e-5, "dpred2")
    close(r["ntm_grad"][0], n[0].grad, 2e-5, "ntm grad")
    # anchors: arg-max pixel per channel, first index
    QM = 40
    ai = r["hout"][16 + 2 * Q * Cn + 2 * QM: 16 + 2 * Q * Cn + 2 * QM + Q].view(torch.int32)
    assert torch.equal(ai.long(), out["anchor_idx1"])


@pytest.mark.parametrize("K", [3, 6, 15])
def test_sig_ntm_sig_w_modules(dev, K):
    d = np.load(os.path.join(G, "g1_g2_ntm_w.npz"))
    Q = 19 + K
    ntm = torch.as_tensor(d[f"ntm_{K}"]).to(dev)
    T = torch.zeros(Q, 19, device=dev); dN = torch.zeros(Q, 19, device=dev)
    ops.sig_ntm(ntm, CD.float().to(dev), T_out=T, dT=torch.as_tensor(d[f"dT_{K}"]).to(dev), dN_out=dN)
    w = torch.as_tensor(d[f"w_{K}"]).to(dev).clone()
    Wm = torch.zeros(Q, Q, device=dev); dw = torch.zeros(Q, Q, device=dev)
    ops.sig_w(w, W_out=Wm, dW=torch.as_tensor(d[f"dW_{K}"]).to(dev), dweight_out=dw)
    torch.cuda.synchronize()
    close(T.cpu(), d[f"T_{K}"], 1e-6, "T")
    close(dN.cpu(), d[f"dntm_{K}"], 2e-6, "dNTM")
    close(Wm.cpu(), d[f"W_{K}"], 1e-6, "W")
    close(w.cpu(), d[f"w_after_{K}"], 0, "weight after (diag = -1e4)")
    close(dw.cpu(), d[f"dw_{K}"], 2e-6, "dweight")
