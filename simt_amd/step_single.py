"""SimT training iteration over a ONE-OUTPUT model on gfx950: DeepLabv3 (model/deeplabv3.py, BASELINE configs[3]) and DeepLab-VGG16
(model/deeplab_vgg.py, configs[4]).

The reference has exactly one SimT loop (tools/trainV2_simt.py:308-436), written for the two-output DeeplabMulti; no script trains
these two model files.  The iteration run here is that loop with every auxiliary-head object removed (pred1, NTM1, NTM_W1 and the
lambda_seg-weighted terms they feed), i.e. term by term the reference's code applied to the model's single output:

    total = Place(pred) + CE(pred, Conf) + NLL(log(softmax(pred) @ T), noisy label)
            + lambda_convex * (-||W T||^2) + lambda_volume * log sqrt|det T^T T| + lambda_anchor * Anchor(T)

(oracle: `oracle.simt_oracle.simt_losses_single`, tied to the golden-pinned two-head restatement by an identity test).  Model facts
that matter to the fused head kernel (csrc/head_loss.hip, simt_head_desc.single / up_half_pixel / fix_logits):
  * DeepLabv3 upsamples INSIDE the model with F.interpolate(bilinear), align_corners=False (deeplabv3.py:137); the script's
    interp_target on the full-resolution tensor is then the identity, and the frozen model's posterior is
    softmax(upsample(logits)).  The kernel consumes the low-res logits and applies the half-pixel taps itself; the 210 MB
    full-resolution logits tensor and its adjoint are never written.
  * DeeplabVGG returns low-res logits; interp_target (align_corners=True) and softmax-then-upsample as for DeepLab-v2.
Optimiser: the model's own optim_parameters (deeplabv3.py:139-166: layer3 at lr, ASSP + classifiers at 10 lr, each tensor listed
once; deeplab_vgg.py:53-54: every parameter, one group), SGD momentum / weight decay as in trainV2_simt.py:296-297; Adam on NTM.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from . import ops
from .engine import LaunchList, side_stream
from .step import lr_poly


class SimTSingleTrainer:
    def __init__(self, model, state, fixed_state, ntm, hp, class_dist, B, H, W, *, dtype=torch.bfloat16, device="cuda:0",
                 process_group=None, arch=None):
        """model: "v3" | "vgg".  state: the trainable model's state_dict tensors (DeepLabv3(nc, openc, openset=True) /
        DeeplabVGG(nc + openc)); fixed_state: the frozen model's (nc outputs).  arch: plan keyword arguments (reduced depths / widths
        for tests): v3 -> layers, width, assp_ch; vgg -> vgg_layers."""
        assert model in ("v3", "vgg")
        self.model, self.hp, self.B, self.H, self.W, self.dtype = model, hp, B, H, W, dtype
        dev = self.dev = torch.device(device)
        self.pg = process_group
        Cn, K = hp.num_classes, hp.open_classes
        self.C, self.Q = Cn, Cn + K
        Q = self.Q
        f32 = torch.float32
        conv = lambda d: {k: v.detach().to(dev, f32 if v.dtype != torch.long else torch.long).clone() for k, v in d.items()}
        self.params, self.fixed_params = conv(state), conv(fixed_state)
        arch = dict(arch or {})
        if model == "v3":
            from .engine_v3 import V3Plan
            self.plan = V3Plan(self.params, B, H, W, Cn, K, True, dtype=dtype, train=True, data_parallel=process_group is not None, **arch)
            self.fixed = V3Plan(self.fixed_params, B, H, W, Cn, 0, False, dtype=dtype, train=False, **arch)
            # the in-model upsample (last forward launch) and its adjoint (first backward launch) are fused into the head kernel
            assert self.plan.fwd_list.items[-1].tag == "simt_upsample_nchw" and self.fixed.fwd_list.items[-1].tag == "simt_upsample_nchw"
            self._fwd, self._fix_fwd, self._bwd = LaunchList(), LaunchList(), LaunchList()
            self._fwd.items = self.plan.fwd_list.items[:-1]
            self._fix_fwd.items = self.fixed.fwd_list.items[:-1]
            i0 = next(i for i, it in enumerate(self.plan.bwd_list.items) if it.tag == "simt_upsample_nchw_bwd")
            self._bwd.items = self.plan.bwd_list.items[:i0] + self.plan.bwd_list.items[i0 + 1:]
            (h, w) = self.plan.feat_hw
            self.pred, self.ldp = self.plan.logits, self.plan.ldq
            self.fix_logits_t, self.ldf = self.fixed.logits, self.fixed.ldq
            dl = self.plan.dlogits["x"]
            half, fix_logits = 1, 1
        else:
            from .engine_vgg import VggPlan
            self.plan = VggPlan(self.params, B, H, W, Q, dtype=dtype, train=True, data_parallel=process_group is not None, **arch)
            self.fixed = VggPlan(self.fixed_params, B, H, W, Cn, dtype=dtype, train=False, **arch)
            self._fwd, self._fix_fwd, self._bwd = self.plan.fwd_list, self.fixed.fwd_list, self.plan.bwd_list
            h, w = self.plan.heads[0].h, self.plan.heads[0].w
            self.pred, self.ldp = self.plan.out["x"], self.plan.ldp["x"]
            self.fix_logits_t, self.ldf = self.fixed.out["x"], self.fixed.ldp["x"]
            dl = self.plan.dlogits["x"]
            half, fix_logits = 0, 0
        self.h, self.w = h, w
        self.grad_ready = self.plan.grad_ready
        # ---- NTM / W (index [1] of the two-slot descriptors; slot [0] = the auxiliary head, absent here)
        self.ntm = ntm.detach().to(dev, f32).clone()
        self._xchg = torch.zeros(16 + Q * Cn, device=dev)      # [lout | dNTM]: lout[12:] + dNTM is ONE collective under data parallelism (step.py)
        self.ntm_grad = self._xchg[16:].view(Q, Cn)
        self.ntm_m, self.ntm_v = torch.zeros(Q, Cn, device=dev), torch.zeros(Q, Cn, device=dev)
        self.wraw = torch.full((Q, Q), 1.0 / (Q - 1.0), device=dev)
        self.w_m, self.w_v = torch.zeros(Q, Q, device=dev), torch.zeros(Q, Q, device=dev)
        self.T = torch.zeros(Q, Cn, device=dev)
        self.cd = torch.as_tensor(np.asarray(class_dist), dtype=f32).to(dev)
        self.inner_steps = 10
        lib = L.load()
        self.part = torch.zeros(lib.simt_head_nblk(B, H, W), lib.simt_head_part_floats(Q, Cn), device=dev)
        self.keys = torch.zeros(lib.simt_head_keys_count(), device=dev, dtype=torch.int64)
        self.hout = torch.zeros(lib.simt_head_hout_floats(Q, Cn), device=dev)
        self.lout = self._xchg[:16]
        self._bad_reported = 0
        self.QP = ops.round_up(Q, 8)
        self.g1 = torch.zeros(2, B, H, w, self.QP, device=dev)
        self.fixp = self.fix_logits_t if fix_logits else torch.zeros(B * h * w, self.ldf, device=dev)
        self.label = torch.zeros(B, H, W, device=dev, dtype=torch.int64)
        hd = L.HeadDesc()
        hd.pred1, hd.pred2, hd.fixp, hd.label = None, self.pred.data_ptr(), self.fixp.data_ptr(), self.label.data_ptr()
        hd.T1, hd.T2 = None, self.T.data_ptr()
        hd.part, hd.keys, hd.hout, hd.g1 = self.part.data_ptr(), self.keys.data_ptr(), self.hout.data_ptr(), self.g1.data_ptr()
        hd.dpred1_f32, hd.dpred2_f32, hd.dpred1_t, hd.dpred2_t = None, None, None, dl.data_ptr()
        hd.B, hd.h, hd.w, hd.H, hd.W, hd.C, hd.Q = B, h, w, H, W, Cn, Q
        hd.ldp, hd.ldf, hd.QP, hd.ld_f32, hd.ld_t = self.ldp, self.ldf, self.QP, 0, dl.shape[1]
        hd.grad_dtype = ops.dt_code(dtype)
        hd.th_high, hd.th_low, hd.lambda_seg, hd.lambda_place = hp.th_high, hp.th_low, 0.0, hp.lambda_place
        hd.gscale, hd.mode, hd.single, hd.up_half_pixel, hd.fix_logits = 1.0 / hp.iter_size, 0, 1, half, fix_logits
        self.conf_label = torch.full((B, H, W), 255, device=dev, dtype=torch.uint8)      # per-pixel Conf_label_target, 255 = none
        hd.conf_out = self.conf_label.data_ptr()
        self._label_ws = torch.full((B, H, W), 255, device=dev, dtype=torch.uint8)          # checked noisy labels, loss pass -> gradient pass
        hd.label_ws = self._label_ws.data_ptr()
        self.head_desc = hd
        ni = L.NtmInnerDesc()
        ni.ntm[1], ni.w[1], ni.ntm_grad[1] = self.ntm.data_ptr(), self.wraw.data_ptr(), self.ntm_grad.data_ptr()
        ni.w_m[1], ni.w_v[1], ni.T_out[1] = self.w_m.data_ptr(), self.w_v.data_ptr(), self.T.data_ptr()
        ni.class_dist, ni.Q, ni.C, ni.steps, ni.single = self.cd.data_ptr(), Q, Cn, self.inner_steps, 1
        ni.beta1, ni.beta2, ni.eps = 0.9, 0.999, 1e-8
        # every optimiser launch is guarded by the plan's sticky fused-BatchNorm error word (engine.TrunkPlan.fbn_error): once a fused launch has
        # given up polling, SGD, both Adams and the W inner loop change nothing -- a state_dict saved after losses() raised holds the last good state
        self._skip_word = getattr(self.plan, "fbn_err", None)
        if self._skip_word is not None:
            ni.skip_if = self._skip_word.data_ptr()
        self.inner_desc = ni
        npd = L.NtmPostDesc()
        npd.ntm[1], npd.w[1], npd.ntm_grad[1] = self.ntm.data_ptr(), self.wraw.data_ptr(), self.ntm_grad.data_ptr()
        npd.class_dist, npd.hout, npd.lout, npd.Q, npd.C = self.cd.data_ptr(), self.hout.data_ptr(), self.lout.data_ptr(), Q, Cn
        npd.lambda_seg, npd.lambda_convex, npd.lambda_volume = 0.0, hp.lambda_convex, hp.lambda_volume
        npd.lambda_anchor, npd.gscale, npd.single = hp.lambda_anchor, 1.0 / hp.iter_size, 1
        self.post_desc = npd
        self._build_sgd()
        self.it_done = 0
        self.reducer = None
        if self.pg is not None:
            from .dp import BucketReducer, make_buckets
            sizes = {n: k for n, (_o, k) in self.plan.grad_offsets.items()}
            buckets = make_buckets(self.plan.grad_order, sizes, self.plan.grad_ready, bucket_elems=8 << 20)
            self.reducer = BucketReducer(self.plan.flat_grad, buckets, group=self.pg, extra=[self._xchg[12:]])

    # ------------------------------------------------------------------ optimiser
    def optim_groups(self):
        """(group 0 names, group 1 names) per the model's optim_parameters; only tensors that receive a gradient are kept."""
        names = [n for n in self.plan.grads if self.plan.grad_ready.get(n, 0) > 0]
        if self.model == "v3":
            g0 = [n for n in names if n.startswith("resnet.resnet_50.layer3.")]
            g1 = [n for n in names if n.startswith(("assp.", "conv.", "conv_1."))]
            return g0, g1
        return names, []

    def _build_sgd(self):
        g0, g1 = self.optim_groups()
        recs, self.mom = [], {}
        for group, listing in ((0, g0), (1, g1)):
            for n in listing:
                p, g = self.params[n], self.plan.grads[n]
                buf = torch.zeros_like(p)
                self.mom[n] = buf
                recs.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), 1, group))
        self.sgd_names = g0 + g1
        seg_dt = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("mult", "<i4"), ("group", "<i4")])
        chunk, chunks = 65536, []
        for si, r in enumerate(recs):
            chunks += [(si, ci) for ci in range((r[3] + chunk - 1) // chunk)]
        self.sgd_segs = torch.from_numpy(np.array(recs, dtype=seg_dt).view(np.uint8).copy()).to(self.dev)
        self.sgd_chunks = torch.tensor(chunks, dtype=torch.int32).to(self.dev)
        d = L.SgdDesc()
        d.segs, d.chunks, d.nchunks, d.chunk = self.sgd_segs.data_ptr(), self.sgd_chunks.data_ptr(), len(chunks), chunk
        d.momentum, d.dampening = self.hp.momentum, 0.0
        if getattr(self.plan, "fbn_err", None) is not None:      # a fused BatchNorm launch that gave up polling: no update (engine.TrunkPlan.fbn_error)
            d.skip_if = self.plan.fbn_err.data_ptr()
        self.sgd_desc = d

    # ------------------------------------------------------------------ one iteration
    def step(self, image, label, it=None):
        hp = self.hp
        assert hp.iter_size == 1, "gradient accumulation is implemented for the DeepLab-v2 trainer only"
        it = self.it_done if it is None else it
        lr, lr_T = lr_poly(hp.lr, it, hp.num_steps, hp.power), lr_poly(hp.lr_T, it, hp.num_steps, hp.power)
        st = ops.stream_ptr()
        main, side = torch.cuda.current_stream(), side_stream(self.dev)
        ev0 = torch.cuda.Event()
        ev0.record(main)
        self.plan.x_in.copy_(image, non_blocking=True)
        self.label.copy_(label, non_blocking=True)
        ev_in = torch.cuda.Event()
        ev_in.record(main)
        with torch.cuda.stream(side):
            # W inner loop (T, W for the head; its leak into dNTM) and the frozen model's forward, beside the trainable forward
            side.wait_event(ev0)
            self.ntm_grad.zero_()
            ni = self.inner_desc
            ni.step0, ni.lr = self.inner_steps * self.it_done, lr_T
            L.call("simt_ntm_inner_loop", C.byref(ni), side.cuda_stream)
            side.wait_event(ev_in)
            self.fixed.x_in.copy_(self.plan.x_in, non_blocking=True)
            self._fix_fwd.run()
            if not self.head_desc.fix_logits:
                ops.softmax_rows(self.fix_logits_t, self.ldf, self.fixp, self.ldf, self.B * self.h * self.w, self.C)
            ev_fix = torch.cuda.Event()
            ev_fix.record(side)
        self._fwd.run()
        main.wait_event(ev_fix)
        L.call("simt_head_loss", C.byref(self.head_desc), st)
        ev_loss = torch.cuda.Event()
        ev_loss.record(main)
        with torch.cuda.stream(side):          # the regularisers only feed Adam and the loss read-out: beside the head's gradient pass
            side.wait_event(ev_loss)
            L.call("simt_ntm_post", C.byref(self.post_desc), side.cuda_stream)
            ev_post = torch.cuda.Event()
            ev_post.record(side)
        L.call("simt_head_grad", C.byref(self.head_desc), st)
        if self.reducer is not None:
            self.reducer.start()
            self._bwd.run()
            main.wait_event(ev_post)           # the NTM gradients it exchanges (side stream)
            self.reducer.finish()
        else:
            self._bwd.run()
        d = self.sgd_desc
        d.lr[0], d.lr[1] = lr, lr * 10.0
        d.wd[0], d.wd[1] = hp.weight_decay, hp.weight_decay
        d.first_step = 1 if self.it_done == 0 else 0
        L.call("simt_sgd_multi", C.byref(d), st)
        main.wait_event(ev_post)
        ops.adam_step(self.ntm, self.ntm_grad, self.ntm_m, self.ntm_v, lr=lr_T, step=self.it_done + 1, skip_if=self._skip_word)
        self.plan.repack()
        self.it_done += 1
        return self.lout

    def timed_lists(self):
        return [self._fix_fwd, self._fwd, self._bwd]

    def losses(self):
        """Local (no collective): see SimTTrainer.losses."""
        v = self.lout.cpu().tolist()                  # ONE device-to-host copy: the scalars and the bad-label count (lout[12])
        self.plan.raise_on_fbn_error()
        # accumulated by simt_ntm_post since the last call; data parallel: the gradient exchange leaves total / world on every rank
        bad = int(round(v[12] * (self.reducer.world if self.reducer is not None else 1))) - self._bad_reported      # cumulative, never reset on the device
        self._bad_reported += max(bad, 0)
        if bad > 0:      # the reference's nll_loss raises on such a target; the kernels skip the pixel and count it
            raise ValueError(f"{bad} label value(s) outside [0, {self.hp.num_classes}) that are not the ignore value 255")
        return {"total": v[0], "loss_p": v[2], "loss_y": v[4], "place": v[5], "convex": v[6], "volume": v[7], "anchor": v[8],
                "vol_ok": v[9]}
