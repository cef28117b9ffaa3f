"""One SimT training iteration on gfx950: the seam the reference lacks (its loss body is inline in
tools/trainV2_simt.py:308-436).  `SimTTrainer.step(image, label, it)` runs, on the current HIP stream and without a
host sync:

  1. the 10-step W inner loop incl. Adam and the gradient leak into NTM      (trainV2_simt.py:326-339)  simt_ntm_inner_loop
  2. the frozen model forward (eval BN folded) -> low-res posterior          (:351-354)               TrunkPlan(train=False)
  3. the trainable forward (train-mode BN, frozen affine)                     (:370)                   TrunkPlan(train=True)
  4. the fused head: upsample + softmax + every loss term + anchors           (:354-409)               simt_head_loss
     Convex / Volume / Anchor + total + d/dNTM                                (:412-424)               simt_ntm_post
     d/d(low-res logits)                                                      (:428)                   simt_head_grad
  5. the trunk backward (dgrad + wgrad + BN backward)                         (:428)                   TrunkPlan.backward
  6. [DP] RCCL all-reduce (mean) of the flat gradient buffer and of dNTM      (new capability)
  7. SGD with duplicate-listing semantics + Adam on NTM1/NTM2                 (:434-436)               simt_sgd_multi / simt_adam_step

All state lives on the device; scalars are read back only when the caller asks (`losses()`).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib as L
from . import ops
from . import trace
from .engine import LaunchList, TrunkPlan, multi_heads, side_stream


class Hyper:
    """Hyper-parameters of tools/trainV2_simt.py:34-70 (flag defaults) / sh_simt.sh:16."""

    def __init__(self, num_classes=19, open_classes=15, th_high=0.8, th_low=0.2, lambda_seg=0.1, lambda_place=0.1,
                 lambda_convex=0.5, lambda_volume=0.1, lambda_anchor=0.5, iter_size=1, lr=2.5e-4, lr_T=2.5e-4,
                 momentum=0.9, weight_decay=5e-4, power=0.9, num_steps=250000, skip_unapplied_grads=False):
        # skip_unapplied_grads: stop the backward at the input of layer3.  The reference's autograd also produces the gradients of
        # conv1 / layer1 / layer2, which its SGD never lists (model/deeplab_multi.py:194-237) and zero_grad() discards; with this
        # switch they are not computed -- identical parameter trajectory and losses, less work.  Off by default (and in bench.py):
        # the default does everything the reference's iteration does.
        self.__dict__.update(locals())
        del self.__dict__["self"]


def lr_poly(base_lr, it, max_iter, power):
    return base_lr * ((1 - float(it) / max_iter) ** power)            # trainV2_simt.py:174-175


def optim_listing(names, layers_root=("layer3", "layer4"), head_prefixes=("layer5", "layer6", "layer5_1", "layer6_1")):
    """Multiplicity of each tensor in ResNetMulti.optim_parameters (model/deeplab_multi.py:194-237).

    get_1x_lr_params_NOscale walks `for j in layer.modules(): for k in j.parameters()` -- every sub-module yields the
    parameters of its whole subtree, so a tensor is listed once per ancestor module inside the layer (the layer
    itself, its Bottleneck, the leaf; plus the `downsample` Sequential for its members).  group 0 = those, group 1
    (10x lr) = the heads, listed once."""
    g0, g1 = {}, {}
    for n in names:
        root = n.split(".")[0]
        if root in layers_root:
            depth = len(n.split(".")) - 1          # layer3.0.conv1.weight -> 3 ancestors; downsample.0.weight -> 4
            g0[n] = depth
        elif root in head_prefixes:
            g1[n] = 1
    return g0, g1


class SimTTrainer:
    def __init__(self, state, fixed_state, ntm1, ntm2, hp, class_dist, B, H, W, *, dtype=torch.bfloat16, device="cuda:0",
                 openset=True, process_group=None, w_init=None, layers=None):
        self.hp, self.B, self.H, self.W, self.dtype = hp, B, H, W, dtype
        dev = self.dev = torch.device(device)
        self.pg = process_group
        Cn, K = hp.num_classes, hp.open_classes
        self.C, self.Q = Cn, Cn + K
        f32 = torch.float32
        self.params = {k: v.detach().to(dev, f32 if v.dtype != torch.long else torch.long).clone() for k, v in state.items()}
        self.fixed_params = {k: v.detach().to(dev, f32 if v.dtype != torch.long else torch.long).clone()
                             for k, v in fixed_state.items()}
        kw = {"layers": layers} if layers is not None else {}
        self.plan = TrunkPlan(self.params, B, H, W, multi_heads(Cn, K, openset), dtype=dtype, train=True,
                              grads_from_layer=3 if getattr(hp, "skip_unapplied_grads", False) else 0,
                              data_parallel=process_group is not None, **kw)
        # the frozen model sees the same image: it reuses the trainable plan's stem im2col matrix (one im2col per micro-batch)
        self.fixed = TrunkPlan(self.fixed_params, B, H, W, multi_heads(Cn, 0, False), dtype=dtype, train=False,
                               stem_from=self.plan, data_parallel=process_group is not None, **kw)      # (same CU budget = same tile lists as the trainable plan)
        # SIMT_FROZEN_SPLIT=n (round 6 experiment): the frozen forward as n chains of B / n images each on n side streams.  The frozen net runs in
        # eval mode with folded BatchNorm: every image is independent, so the split changes nothing but the launch geometry -- B / n images are
        # 236 / n one-per-CU workgroups per wide conv, and n such launches FIT the chip side by side (two full-batch launches do not: 2 x 236 > 256),
        # so one chain's prologue / epilogue can overlap the other's K loop.  Measured +0.65 / +1.65 ms; needs SIMT_DIRECT_STEM=0 (the parts take their
        # rows of the trainable plan's im2col matrix), ignored otherwise
        self._fixed_parts, self._side2 = None, []
        import os
        nsp = int(os.environ.get("SIMT_FROZEN_SPLIT", "1"))
        if nsp > 1 and B % nsp == 0 and dtype == torch.bfloat16 and not self.plan.direct_stem:
            self._fixed_parts = [TrunkPlan(self.fixed_params, B // nsp, H, W, multi_heads(Cn, 0, False), dtype=dtype, train=False,
                                           stem_from=(self.plan, i, nsp), data_parallel=process_group is not None, **kw) for i in range(nsp)]
            self._side2 = [torch.cuda.Stream(device=dev) for _ in range(nsp - 1)]
        # item 0 of the trainable forward feeds BOTH nets: the direct stem launch with the frozen net as its second weight set (bf16), or the im2col
        assert self.plan.fwd_list.items[0].tag in ("simt_stem7_fwd", "simt_im2col_stem")
        assert self.plan.fwd_list.items[0].tag == "simt_im2col_stem" or self.plan.stem_desc.nsets == 2
        import os
        # hipGraphs of the three launch lists (frozen forward, trainable forward, backward with its two streams): the host spends
        # ~18 us per eager launch (ctypes + Python), ~1 000 launches per step -- most of a 28 ms step, and the ORDER in which the two
        # streams' launches reach the GPU was the host's, not the plan's.  SIMT_GRAPHS=0 keeps eager launches (needed by nothing;
        # the data-parallel backward with its per-bucket hooks stays eager by construction).
        self._fixed_graph = False
        # SIMT_GRAPHS: -1 (default) eager launches; 0 frozen forward as a hipGraph, 1 + trainable forward, 2 + backward (measured: 27.6 /
        # 27.7 / 27.7 / 29.9 ms per step -- the graph executor runs the two-stream backward worse than the eager streams do)
        self._graph_level = int(os.environ.get("SIMT_GRAPHS", "-1"))
        self._graphs = self._graph_level >= 0 and torch.device(dev).type == "cuda"
        self._steps_run = 0
        self._main_hi = os.environ.get("SIMT_MAIN_PRIORITY", "0") == "1"
        # host enqueue order of the two forwards (both reach the GPU within ~2 ms; the hardware then favours the queue that was fed
        # first): "main" = trainable first (default), "side" = frozen first, "interleave" = one list alternating 1:1
        self._fwd_order = os.environ.get("SIMT_FWD_ORDER", "main")      # "pair" (round 5): one launch per layer for both networks, main stream only (measured +1.3 ms: it gives up the two-stream overlap)
        self._hi_stream = None
        fa = os.environ.get("SIMT_FROZEN_AFTER")
        self._frozen_after = next((r["fwd_start"] for r in self.plan.block_io if r["name"] == fa), None) if fa else None
        self._fwd_rest = LaunchList()
        self._fwd_rest.items = self.plan.fwd_list.items[1:]
        self._fwd_both = None                      # built on first use (needs self.fixp): both forwards interleaved, see _micro_batch
        h, w = self.plan.heads[1].h, self.plan.heads[1].w
        self.h, self.w = h, w
        Q = self.Q
        # ---- NTM / W state (model/deeplab_multi.py:244-286)
        self.ntm = [ntm1.detach().to(dev, f32).clone(), ntm2.detach().to(dev, f32).clone()]
        # ONE allocation [16 scalars (lout) | dNTM1 | dNTM2]: under data parallelism its tail from lout[12] on (the bad-label count, three unused
        # slots, both NTM gradients) is ONE contiguous collective at the end of the exchange (it used to be two)
        self._xchg = torch.zeros(16 + 2 * Q * Cn, device=dev)
        self._ntm_grad_flat = self._xchg[16:].view(2, Q, Cn)
        self.ntm_grad = [self._ntm_grad_flat[0], self._ntm_grad_flat[1]]
        self.ntm_m = [torch.zeros(Q, Cn, device=dev) for _ in range(2)]
        self.ntm_v = [torch.zeros(Q, Cn, device=dev) for _ in range(2)]
        wi = w_init if w_init is not None else torch.full((Q, Q), 1.0 / (Q - 1.0))
        self.wraw = [wi.detach().to(dev, f32).clone() for _ in range(2)]
        self.w_m = [torch.zeros(Q, Q, device=dev) for _ in range(2)]
        self.w_v = [torch.zeros(Q, Q, device=dev) for _ in range(2)]
        self.T = [torch.zeros(Q, Cn, device=dev) for _ in range(2)]
        self.cd = torch.as_tensor(np.asarray(class_dist), dtype=f32).to(dev)
        self.inner_steps = 10
        # ---- head workspaces
        lib = L.load()
        self.nblk = lib.simt_head_nblk(B, H, W)
        self.part = torch.zeros(self.nblk, lib.simt_head_part_floats(Q, Cn), device=dev)
        self.keys = torch.zeros(lib.simt_head_keys_count(), device=dev, dtype=torch.int64)
        self.hout = torch.zeros(lib.simt_head_hout_floats(Q, Cn), device=dev)
        self.lout = self._xchg[:16]
        self._bad_reported = 0                     # out-of-range labels already raised for (host side; the device counter is never reset)
        self.QP = ops.round_up(Q, 8)
        self.g1 = torch.zeros(2, B, H, w, self.QP, device=dev)
        self.ldf = self.fixed.ldp["x2"]
        self.fixp = torch.zeros(B * h * w, self.ldf, device=dev)
        self.label = torch.zeros(B, H, W, device=dev, dtype=torch.int64)
        hd = L.HeadDesc()
        p1, p2 = self.plan.out["x1"], self.plan.out["x2"]
        hd.pred1, hd.pred2, hd.fixp, hd.label = p1.data_ptr(), p2.data_ptr(), self.fixp.data_ptr(), self.label.data_ptr()
        hd.T1, hd.T2 = self.T[0].data_ptr(), self.T[1].data_ptr()
        hd.part, hd.keys, hd.hout, hd.g1 = self.part.data_ptr(), self.keys.data_ptr(), self.hout.data_ptr(), self.g1.data_ptr()
        hd.dpred1_f32, hd.dpred2_f32 = None, None
        d1, d2 = self.plan.dlogits["x1"], self.plan.dlogits["x2"]
        hd.dpred1_t, hd.dpred2_t = d1.data_ptr(), d2.data_ptr()
        hd.B, hd.h, hd.w, hd.H, hd.W, hd.C, hd.Q = B, h, w, H, W, Cn, Q
        hd.ldp, hd.ldf, hd.QP, hd.ld_f32, hd.ld_t = self.plan.ldp["x1"], self.ldf, self.QP, 0, d1.shape[1]
        hd.grad_dtype = ops.dt_code(dtype)
        hd.th_high, hd.th_low, hd.lambda_seg, hd.lambda_place = hp.th_high, hp.th_low, hp.lambda_seg, hp.lambda_place
        hd.gscale = 1.0 / hp.iter_size
        # per-pixel Conf_label_target of the last (micro-)batch (trainV2_simt.py:357-362,387-393), 255 = none: 1 byte per pixel
        self.conf_label = torch.full((B, H, W), 255, device=dev, dtype=torch.uint8)
        hd.conf_out = self.conf_label.data_ptr()
        self._label_ws = torch.full((B, H, W), 255, device=dev, dtype=torch.uint8)          # checked noisy labels, loss pass -> gradient pass
        hd.label_ws = self._label_ws.data_ptr()
        self.head_desc = hd
        # ---- NTM descriptors
        ni = L.NtmInnerDesc()
        for k in range(2):
            ni.ntm[k], ni.w[k], ni.ntm_grad[k] = self.ntm[k].data_ptr(), self.wraw[k].data_ptr(), self.ntm_grad[k].data_ptr()
            ni.w_m[k], ni.w_v[k], ni.T_out[k] = self.w_m[k].data_ptr(), self.w_v[k].data_ptr(), self.T[k].data_ptr()
        ni.class_dist, ni.Q, ni.C, ni.steps = self.cd.data_ptr(), Q, Cn, self.inner_steps
        ni.beta1, ni.beta2, ni.eps = 0.9, 0.999, 1e-8
        # every optimiser launch is guarded by the plan's sticky fused-BatchNorm error word (engine.TrunkPlan.fbn_error): once a fused launch has
        # given up polling, SGD, both Adams and the W inner loop change nothing -- a state_dict saved after losses() raised holds the last good state
        self._skip_word = getattr(self.plan, "fbn_err", None)
        if self._skip_word is not None:
            ni.skip_if = self._skip_word.data_ptr()
        self.inner_desc = ni
        npd = L.NtmPostDesc()
        for k in range(2):
            npd.ntm[k], npd.w[k], npd.ntm_grad[k] = self.ntm[k].data_ptr(), self.wraw[k].data_ptr(), self.ntm_grad[k].data_ptr()
        npd.class_dist, npd.hout, npd.lout, npd.Q, npd.C = self.cd.data_ptr(), self.hout.data_ptr(), self.lout.data_ptr(), Q, Cn
        npd.lambda_seg, npd.lambda_convex, npd.lambda_volume = hp.lambda_seg, hp.lambda_convex, hp.lambda_volume
        npd.lambda_anchor, npd.gscale = hp.lambda_anchor, 1.0 / hp.iter_size
        self.post_desc = npd
        # ---- SGD segments (duplicate listings replayed in registers)
        self._build_sgd()
        self.it_done = 0
        # ---- early optimiser step: the gradients SGD applies (layer3, layer4, heads) are final long before the backward has walked
        # through layer2 / layer1 / the stem (whose gradients this stage computes but never applies), so SGD and the re-pack of the
        # updated weights run on the side stream from that point on instead of after the backward (0.7 ms of exclusive tail).
        applied = [n for n in self.sgd_names if self.plan.grad_ready.get(n, 0) > 0]
        self._ev_post = None
        self._post_side = os.environ.get("SIMT_POST_SIDE", "1") != "0"
        self._early_cut = max(self.plan.grad_ready[n] for n in applied) if applied else None
        if self._early_cut is not None:
            # The block that produced the last applied weight gradient lists its wgrads BEFORE its input-gradient convs
            # (engine._build_backward), and those convs read the packed operands of conv1 / downsample that the early re-pack
            # overwrites: fire at the first hook point at or after the END of that block, so that the event the side stream waits
            # for covers them (WAR).  If the block has no input gradient (nothing below it is differentiated) the cut stands.
            blk = min((m for m in self.plan.bwd_marks.values() if m[1] >= self._early_cut), key=lambda m: m[1], default=None)
            if blk is not None and blk[3] is not None:
                later = [c for c in sorted(set(self.plan.grad_ready.values())) if c >= blk[1]]
                self._early_cut = later[0] if later else None
        self._early_sgd = (os.environ.get("SIMT_EARLY_SGD", "1") != "0" and self._early_cut is not None
                           and hp.iter_size == 1 and self._early_cut < len(self.plan.bwd_list.items))
        self._pack_applied = self.plan.pack_subset(self.sgd_names) if self._early_sgd else None
        # ---- data parallel: bucketed mean all-reduce overlapped with backward
        self._grad_acc = torch.zeros_like(self.plan.flat_grad) if hp.iter_size > 1 else None
        self.reducer = None
        if self.pg is not None:
            from .dp import BucketReducer, make_buckets
            order, sizes, end = self.exchange_table()
            import os as _os
            be = int(float(_os.environ.get("SIMT_DP_BUCKET_MB", "0")) * (1 << 18)) or self.BUCKET_ELEMS      # (MB of fp32 per all-reduce; default 32)
            buckets = make_buckets(order, sizes, self.plan.grad_ready, bucket_elems=be)
            # the bad-label count (lout[12], accumulated by simt_ntm_post) rides in the same exchange: after the mean every rank holds
            # total / world, so losses() needs no collective of its own and every rank raises in the same call
            self.reducer = BucketReducer(self.plan.flat_grad[:end], buckets, group=self.pg, extra=[self._xchg[12:]])
            self.reducer.total_launches = len(self.plan.bwd_list.items)

    BUCKET_ELEMS = 8 << 20      # 32 MB of fp32 per all-reduce: few large messages for xGMI rings (dp.py)

    def exchange_table(self):
        """(order, sizes, end): the gradients the data-parallel exchange carries, in flat-buffer order, their padded spans in elements and
        the length of the prefix of `plan.flat_grad` they form.  Only gradients the optimiser APPLIES cross xGMI: layer3, layer4
        and the heads = the head of the flat buffer (169 of 180 MB).  conv1 / layer1 / layer2 gradients are computed like the reference
        computes them (unless skip_unapplied_grads) but the SimT stage's SGD never lists them (model/deeplab_multi.py:194-237): they stay
        rank-local.  (tests/golden/g16_dp_bucket_table.json is this table for BASELINE configs[1..2], profiles/tools/dump_bucket_table.py.)"""
        sizes = {n: k for n, (_o, k) in self.plan.grad_offsets.items()}
        applied = set(self.sgd_names)
        order = [n for n in self.plan.grad_order if n in applied and self.plan.grad_ready.get(n, 0) > 0]
        assert order == self.plan.grad_order[:len(order)], "applied gradients must form a prefix of the flat buffer"
        return order, sizes, sum(sizes[n] for n in order)

    # ------------------------------------------------------------------ optimiser plumbing
    def _build_sgd(self, roots=("layer3", "layer4")):
        g0, g1 = optim_listing(self.plan.grads.keys(), layers_root=roots)
        recs = []
        self.mom = {}
        for group, listing in ((0, g0), (1, g1)):
            for n, mult in listing.items():
                p, g = self.params[n], self.plan.grads[n]
                buf = torch.zeros_like(p)
                self.mom[n] = buf
                recs.append((p.data_ptr(), g.data_ptr(), buf.data_ptr(), p.numel(), mult, group))
        self.sgd_names = list(g0) + list(g1)
        seg_dt = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("mult", "<i4"), ("group", "<i4")])
        segs = np.array(recs, dtype=seg_dt)
        chunk = 65536
        chunks = []
        for si, r in enumerate(recs):
            for ci in range((r[3] + chunk - 1) // chunk):
                chunks.append((si, ci))
        self.sgd_segs = torch.from_numpy(segs.view(np.uint8).copy()).to(self.dev)
        self.sgd_chunks = torch.tensor(chunks, dtype=torch.int32).to(self.dev)
        d = L.SgdDesc()
        d.segs, d.chunks, d.nchunks, d.chunk = self.sgd_segs.data_ptr(), self.sgd_chunks.data_ptr(), len(chunks), chunk
        d.momentum, d.dampening = self.hp.momentum, 0.0
        if getattr(self.plan, "fbn_err", None) is not None:      # a fused BatchNorm launch that gave up polling: no update (engine.TrunkPlan.fbn_error)
            d.skip_if = self.plan.fbn_err.data_ptr()
        self.sgd_desc = d

    # ------------------------------------------------------------------ one iteration
    def _micro_batch(self, image, label, st):
        """Steps 2-5 of the iteration for one micro-batch: both forwards, fused head, NTM terms, backward."""
        self.plan.x_in.copy_(image, non_blocking=True)
        self.label.copy_(label, non_blocking=True)
        # the stem launch (first launch of the trainable forward: direct conv of both nets, or the im2col they share) feeds BOTH nets: run it before forking
        head = self.plan.fwd_list.items[0]
        rc = head.fn(*head.args, st)
        if rc != 0:
            L.check(rc)
        # 2. frozen model -> low-res posterior, on the side stream: its eval-mode convs share the CUs with the HBM-bound
        #    BatchNorm passes of the trainable forward (3.) instead of running before it.  The host enqueues ~350 launches per forward
        #    at ~5 us each, so the ORDER of the two enqueue loops matters: the trainable forward (critical path) goes first -- enqueued
        #    second, its first conv reached the GPU 3.6 ms into the step (rocprofv3 kernel trace) -- and the frozen forward is one
        #    hipGraph launch when SIMT_FIXED_GRAPH != 0.
        order = self._fwd_order
        if order == "pair" and not self._fixed_graph:
            if self._fwd_both is None:
                self._fwd_both = self._paired_forwards()
            with trace.range("forward + frozen-forward (one launch per layer for both networks)"):
                self._fwd_both.run()
        elif order == "paced" and not self._fixed_graph:
            if self._fwd_both is None:
                self._fwd_both = self._paced_forwards()
            with trace.range("forward + frozen-forward (frozen 3x3 convs paced behind the trainable conv3s)"):
                self._fwd_both.run()
        elif order in ("bnside", "bnside2") and not self._fixed_graph:
            if self._fwd_both is None:
                self._fwd_both = self._bn_side_forwards()
            with trace.range("forward + frozen-forward (convs on the main stream, BatchNorm passes on the side stream)"):
                self._fwd_both.run()
        elif self._fixed_graph or order != "interleave":
            main = torch.cuda.current_stream()
            side = side_stream(self.dev)
            ev_in = torch.cuda.Event()
            ev_in.record(main)

            def frozen():
                if self._fixed_parts is not None:
                    Mp = (self.B // len(self._fixed_parts)) * self.h * self.w
                    evs = []
                    for i, part in enumerate(self._fixed_parts):
                        stq = side if i == 0 else self._side2[i - 1]
                        with torch.cuda.stream(stq), trace.range("frozen-forward part"):
                            stq.wait_event(ev_in)
                            part.forward()
                            ops.softmax_rows(part.out["x2"], self.ldf, self.fixp[i * Mp:(i + 1) * Mp], self.ldf, Mp, self.C)
                            if i:
                                e = torch.cuda.Event()
                                e.record(stq)
                                evs.append(e)
                    with torch.cuda.stream(side):
                        for e in evs:
                            side.wait_event(e)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    return ev
                with torch.cuda.stream(side), trace.range("frozen-forward"):
                    side.wait_event(ev_in)
                    self.fixed.forward()
                    ops.softmax_rows(self.fixed.out["x2"], self.ldf, self.fixp, self.ldf, self.B * self.h * self.w, self.C)
                    ev = torch.cuda.Event()
                    ev.record(side)
                return ev
            if self._fixed_graph or order == "side":
                ev_fix = frozen()
                with trace.range("forward"):
                    self._fwd_rest.run()           # 3. trainable forward (its im2col already ran above)
            elif self._frozen_after is not None:
                # experiment (SIMT_FROZEN_AFTER=<block name>): the frozen forward is released when the trainable forward reaches that block
                cut = self._frozen_after - 1       # (_fwd_rest starts behind the im2col)
                a_, b_ = LaunchList(), LaunchList()
                a_.items, b_.items = self._fwd_rest.items[:cut], self._fwd_rest.items[cut:]
                with trace.range("forward"):
                    a_.run()
                    ev_in.record(main)
                    ev_fix = frozen()
                    b_.run()
            else:
                with trace.range("forward"):
                    self._fwd_rest.run()
                ev_fix = frozen()
            main.wait_event(ev_fix)
        else:
            # ONE list with the launches of the two forwards interleaved 1:1
            if self._fwd_both is None:
                self._fwd_both = self._interleaved_forwards()
            with trace.range("forward + frozen-forward (interleaved)"):
                self._fwd_both.run()
        # 4. fused head + NTM regularisers + gradients of the low-res logits (every term scaled by 1 / iter_size, :427;
        #    the NTM gradients accumulate on top of the inner loop's leak and of earlier micro-batches)
        #    The regularisers only feed the Adam step and the loss read-out: they run on the side stream (idle between the frozen forward
        #    and the first weight gradients) beside the head's gradient pass instead of in front of it (a 100 us single-block kernel).
        main = torch.cuda.current_stream()
        side = side_stream(self.dev) if self._post_side else main
        if self._ev_post is not None:
            main.wait_event(self._ev_post)     # an earlier micro-batch's regularisers still read the head's outputs
        with trace.range("head"):
            L.call("simt_head_loss", C.byref(self.head_desc), st)
            ev_loss = torch.cuda.Event()
            ev_loss.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_loss)
                L.call("simt_ntm_post", C.byref(self.post_desc), side.cuda_stream)
                self._ev_post = torch.cuda.Event()
                self._ev_post.record(side)
            L.call("simt_head_grad", C.byref(self.head_desc), st)

    def _interleaved_forwards(self):
        """fwd_rest (main stream) and the frozen forward + its soft-max (side stream) as one launch list, alternating."""
        from .engine import _Launch
        lib = L.load()
        both = LaunchList()
        ev_in = both.record(0)
        both.wait(ev_in, 1)
        a = list(self._fwd_rest.items)
        b = []
        for it in self.fixed.fwd_list.items:
            assert it.stream == 0 and it.fn is not None
            b.append(_Launch(it.fn, it.args, it.keep, it.tag, it.flops, it.bytes, it.shape, stream=1))
        b.append(_Launch(lib.simt_softmax_rows, (ops._p(self.fixed.out["x2"]), self.ldf, ops._p(self.fixp), self.ldf, self.B * self.h * self.w,
                                                 self.C), (self.fixed.out["x2"], self.fixp), "simt_softmax_rows", stream=1))
        assert all(it.stream == 0 for it in a)
        # paced: the launches of each stream in chunks of `pace`; chunk c of one stream waits for chunk c - 2 of the other, so neither
        # queue can run more than two chunks ahead (left alone, the GPU serves one queue exclusively for milliseconds: the two forwards
        # then run one after the other and fill none of each other's tails)
        import os
        pace = int(os.environ.get("SIMT_FWD_PACE", "8"))
        ca = [a[k:k + pace] for k in range(0, len(a), pace)]
        cb = [b[k:k + pace] for k in range(0, len(b), pace)]
        # proportional pairing: both streams finish their last chunk together
        n = max(len(ca), len(cb))
        ia = [min(len(ca), (k + 1) * len(ca) // n) for k in range(n)]
        ib = [min(len(cb), (k + 1) * len(cb) // n) for k in range(n)]
        ev_a, ev_b = [], []
        pa = pb = 0
        for k in range(n):
            if pace > 0 and k >= 2:
                if ev_b[k - 2] is not None: both.wait(ev_b[k - 2], 0)
                if ev_a[k - 2] is not None: both.wait(ev_a[k - 2], 1)
            for c in cb[pb:ib[k]]: both.items.extend(c)
            for c in ca[pa:ia[k]]: both.items.extend(c)
            ev_b.append(both.record(1) if ib[k] > pb else None)
            ev_a.append(both.record(0) if ia[k] > pa else None)
            pa, pb = ia[k], ib[k]
        ev_fix = both.record(1)
        both.wait(ev_fix, 0)
        return both

    def _paired_forwards(self):
        """Both forwards as ONE launch list on the main stream with ONE launch per layer for both networks (round 5; OPT-IN,
        SIMT_FWD_ORDER=pair: measured +1.3 ms per step, the default stays "main"): the trainable and the frozen ResNetMulti run the same conv shapes on the same image
        (tools/trainV2_simt.py:351-353 and :370 -> model/deeplab_multi.py:172-192), so conv k of one and conv k of the other go into one
        simt_conv_fprop_pair launch (510 workgroups: the second 255 start as the first drain), each half with its own compile-time epilogue
        (BatchNorm statistics | folded bias + ReLU [+ residual]).  The two lists are aligned by a longest-common-subsequence match on the
        conv geometries (their orders differ around the downsample convs: conv3 needs the frozen net's downsample output as its residual),
        each list's own order is kept, everything that has no partner is launched as before.  The two-stream schedule this replaces ran the
        forwards one after the other anyway (the GPU serves one conv queue at a time: profiles/r02_step_timeline.txt); what the pairing removes
        is one launch ramp / drain per layer: 11 us per 3x3 pair, 12 us per conv3 pair, 2.5 us per 1x1 1024 -> 256 pair
        (profiles/r05_conv_attribution.txt section 5).  Outputs are bit-identical to the separate launches (tests/test_gpu_iteration.py)."""
        from .engine import _Launch
        lib = L.load()
        conv_fn = lib.simt_conv_fprop
        t_items = list(self._fwd_rest.items)
        f_items = list(self.fixed.fwd_list.items)
        f_items.append(_Launch(lib.simt_softmax_rows, (ops._p(self.fixed.out["x2"]), self.ldf, ops._p(self.fixp), self.ldf, self.B * self.h * self.w,
                                                       self.C), (self.fixed.out["x2"], self.fixp), "simt_softmax_rows"))
        assert all(it.fn is not None and it.stream == 0 for it in t_items + f_items)

        def segments(items):
            """[(conv item or None, [the launches that follow it up to the next conv])]; the first segment's conv is None (leading launches)"""
            segs = [(None, [])]
            for it in items:
                if it.fn is conv_fn:
                    segs.append((it, []))
                else:
                    segs[-1][1].append(it)
            return segs
        ts, fs = segments(t_items), segments(f_items)
        tc, fc = ts[1:], fs[1:]
        fused = [[bool(lib.simt_conv_pair_fused(C.byref(a[0].keep), C.byref(b[0].keep))) for b in fc] for a in tc]
        # longest common subsequence over the fusable pairs (104 x 104)
        n, m = len(tc), len(fc)
        dp = [[0] * (m + 1) for _ in range(n + 1)]
        for i in range(n - 1, -1, -1):
            for j in range(m - 1, -1, -1):
                dp[i][j] = max(dp[i + 1][j], dp[i][j + 1], (1 + dp[i + 1][j + 1]) if fused[i][j] else 0)
        both = LaunchList()
        both.items += ts[0][1] + fs[0][1]
        i = j = 0
        self.fwd_pairs = 0

        def single(seg):
            both.items.append(seg[0])
            both.items.extend(seg[1])
        while i < n or j < m:
            if i < n and j < m and fused[i][j] and dp[i][j] == 1 + dp[i + 1][j + 1]:
                a, b = tc[i][0], fc[j][0]
                if a.tag.startswith("conv_igemm2_kernel<") and b.tag.startswith("conv_igemm2_kernel<"):
                    tag = ", ".join(a.tag.split(", ")[:4]) + f", {a.tag.split(', ')[4].rstrip('>')}+{b.tag.split(', ')[4].rstrip('>')}>"
                else:
                    tag = a.tag + " (pair)"
                both.items.append(_Launch(lib.simt_conv_fprop_pair, (C.byref(a.keep), C.byref(b.keep)), (a.keep, b.keep), tag,
                                          a.flops + b.flops, a.bytes + b.bytes, a.shape + " x2 (trainable + frozen)"))
                both.items.extend(tc[i][1])
                both.items.extend(fc[j][1])
                self.fwd_pairs += 1
                i, j = i + 1, j + 1
            elif j < m and (i >= n or dp[i][j] == dp[i][j + 1]):
                single(fc[j])
                j += 1
            else:
                single(tc[i])
                i += 1
        assert len(both.items) == len(t_items) + len(f_items) - self.fwd_pairs
        return both

    def _bn_side_forwards(self):
        """Both forwards as ONE launch list (round 5, SIMT_FWD_ORDER=bnside): EVERY conv -- the trainable net's and the frozen net's, alternating
        layer by layer -- on the main stream, and only the trainable net's BatchNorm launches (finalize, apply, the stem's BN + max-pool) on
        the side stream, released by an event behind the conv that feeds them and awaited by the trainable net's next conv.  While the
        statistics are finalised and the HBM-bound apply pass streams (53 VGPRs, no LDS: it shares a CU with a conv workgroup), the main
        queue runs the FROZEN net's conv of the same layer, which depends on nothing of it.  The two queues never hold two convs at once
        (each conv workgroup takes a whole CU's LDS, so two conv queues only take turns: profiles/r02_step_timeline.txt), which is what the
        alternating-convs experiments of round 4 (SIMT_FWD_ORDER=interleave) paid for; this is the backward's pattern (weight gradients
        released by events) applied to the forward."""
        from .engine import _Launch
        lib = L.load()
        both = LaunchList()
        conv_fns = (lib.simt_conv_fprop,)
        t_items = list(self._fwd_rest.items)
        f_items = list(self.fixed.fwd_list.items)
        f_items.append(_Launch(lib.simt_softmax_rows, (ops._p(self.fixed.out["x2"]), self.ldf, ops._p(self.fixp), self.ldf, self.B * self.h * self.w,
                                                       self.C), (self.fixed.out["x2"], self.fixp), "simt_softmax_rows"))
        assert all(it.fn is not None and it.stream == 0 for it in t_items + f_items)
        fi = 0

        def frozen_upto_next_conv():
            """the frozen net's next conv and whatever non-conv launches follow it (main stream)"""
            nonlocal fi
            if fi >= len(f_items):
                return
            both.items.append(f_items[fi])
            fi += 1
            while fi < len(f_items) and f_items[fi].fn not in conv_fns:
                both.items.append(f_items[fi])
                fi += 1
        # the frozen net may start with non-conv launches (none today): flush them
        while fi < len(f_items) and f_items[fi].fn not in conv_fns:
            both.items.append(f_items[fi])
            fi += 1
        ev_bn = None
        i = 0
        while i < len(t_items):
            it = t_items[i]
            if it.fn in conv_fns:
                if ev_bn is not None:
                    both.wait(ev_bn, 0)
                    ev_bn = None
                both.items.append(it)
                j = i + 1
                while j < len(t_items) and t_items[j].fn not in conv_fns:
                    j += 1
                group = t_items[i + 1:j]
                # "bnside2": the (tiny, LDS-using) finalize stays on the main stream right behind its conv; only the streaming pass moves
                if self._fwd_order == "bnside2":
                    while group and group[0].fn is lib.simt_bn_finalize:
                        both.items.append(group.pop(0))
                if group:
                    ev_conv = both.record(0)
                    both.wait(ev_conv, 1)
                    for g in group:
                        both.items.append(_Launch(g.fn, g.args, g.keep, g.tag, g.flops, g.bytes, g.shape, stream=1))
                    ev_bn = both.record(1)
                frozen_upto_next_conv()
                i = j
            else:                                   # (a non-conv launch before the first conv: main stream)
                both.items.append(it)
                i += 1
        while fi < len(f_items):
            frozen_upto_next_conv()
        if ev_bn is not None:
            both.wait(ev_bn, 0)
        return both

    def _paced_forwards(self):
        """Both forwards as ONE launch list on the two streams (round 6, SIMT_FWD_ORDER=paced): the trainable net on the main stream exactly as in
        the default order, the frozen net on the side stream -- but its k-th 3x3 conv waits for an event recorded behind the trainable net's k-th
        conv3.  What follows that conv3 on the main stream is the finalize and the wide bn3 + residual + ReLU pass (231 MB, HBM-bound, no LDS, 53
        VGPRs: it shares a CU with a conv workgroup), and a BatchNorm pass hides half of itself under a 3x3 conv and nothing under the K = 1024
        1x1 conv (profiles/r06_corun.txt).  In the default order the host enqueues the frozen forward behind the trainable one and it runs after
        it, alone (profiles/tools/queue_gaps.py: 4.1 ms), so that pairing never happens.  ONE cross-queue edge per Bottleneck, on the queue that
        has the slack (the frozen chain is ~100 us per Bottleneck against ~170 us): the main stream never waits for the side stream before the end
        of the forward (model/deeplab_multi.py:81-101; tools/trainV2_simt.py:351-353,370)."""
        import re
        from .engine import _Launch
        lib = L.load()
        both = LaunchList()
        conv = lib.simt_conv_fprop
        t_items = list(self._fwd_rest.items)
        f_items = [_Launch(g.fn, g.args, g.keep, g.tag, g.flops, g.bytes, g.shape, stream=1) for g in self.fixed.fwd_list.items]
        f_items.append(_Launch(lib.simt_softmax_rows, (ops._p(self.fixed.out["x2"]), self.ldf, ops._p(self.fixp), self.ldf, self.B * self.h * self.w,
                                                       self.C), (self.fixed.out["x2"], self.fixp), "simt_softmax_rows", stream=1))
        assert all(it.fn is not None and it.stream == 0 for it in t_items)

        def is_conv2(it):
            return it.fn is conv and " taps9 " in f" {it.shape or ''} " and "tap-expanded" not in (it.shape or "")

        def is_conv3(it):
            m = re.search(r"N(\d+) K(\d+) taps1 ", f"{it.shape or ''} ") if it.fn is conv else None
            return bool(m) and int(m.group(1)) == 4 * int(m.group(2))
        # (layer1.0's downsample conv has conv3's shape: a Bottleneck's conv3 is the first such conv BEHIND its 3x3 conv)
        marks, armed = set(), False
        for i, it in enumerate(t_items):
            if is_conv2(it):
                armed = True
            elif armed and is_conv3(it):
                marks.add(i)
                armed = False
        n2 = sum(1 for it in f_items if is_conv2(it))
        assert n2 == len(marks) and n2 > 0, (n2, len(marks))
        fi = 0

        def frozen_until_next_conv2(ev):
            """the frozen net's launches up to (not including) its next 3x3 conv; the first of them -- a 3x3 conv -- behind `ev`"""
            nonlocal fi
            if fi < len(f_items) and ev is not None:
                both.wait(ev, 1)
            first = True
            while fi < len(f_items) and (first or not is_conv2(f_items[fi])):
                both.items.append(f_items[fi])
                fi += 1
                first = False
        # the frozen net's head of list (stem pool, conv1 of the first Bottleneck): free-running beside the trainable stem
        while fi < len(f_items) and not is_conv2(f_items[fi]):
            both.items.append(f_items[fi])
            fi += 1
        for i, it in enumerate(t_items):
            both.items.append(it)
            if i in marks:
                frozen_until_next_conv2(both.record(0))
        while fi < len(f_items):
            both.items.append(f_items[fi])
            fi += 1
        both.wait(both.record(1), 0)                 # the head needs the frozen posterior
        return both

    def _capture_graphs(self):
        """Record (not run) the three launch lists into hipGraphs, once, after the first eager step created every workspace."""
        self._graphs = False
        try:
            torch.cuda.synchronize()
            with torch.cuda.stream(side_stream(self.dev)):
                self.fixed.fwd_list.capture(warm=False)
            if self._graph_level >= 1:
                self._fwd_rest.capture(warm=False)
            if self.reducer is None and self._graph_level >= 2:
                self.plan.bwd_list.capture(warm=False)
            self._fixed_graph = True
        except Exception as e:                     # capture is an optimisation of the host side only
            self.fixed.fwd_list.graph = self._fwd_rest.graph = self.plan.bwd_list.graph = None
            self._fixed_graph = False
            print(f"[simt] hipGraph capture failed ({e}); eager launches", flush=True)

    def step(self, image, label, it=None):
        """Runs _step (below); with SIMT_MAIN_PRIORITY=1 on a high-priority HIP stream ordered after / before the caller's stream."""
        if not self._main_hi:
            return self._step(image, label, it)
        caller = torch.cuda.current_stream()
        if self._hi_stream is None:
            self._hi_stream = torch.cuda.Stream(device=self.dev, priority=-1)
        self._hi_stream.wait_stream(caller)
        with torch.cuda.stream(self._hi_stream):
            out = self._step(image, label, it)
        caller.wait_stream(self._hi_stream)
        return out

    def _step(self, image, label, it=None):
        """image [B,3,H,W] fp32 (device or host), label [B,H,W] int64 -- or, with hp.iter_size > 1 (gradient accumulation,
        trainV2_simt.py:341-432), sequences of iter_size micro-batches.  Returns the device tensor `lout`
        (total, loss_p1, loss_p2, loss_y1, loss_y2, Place, Convex, Volume, Anchor, vol_ok, ...) of the last micro-batch."""
        hp = self.hp
        if self._graphs and self._steps_run >= 1:
            self._capture_graphs()
        self._steps_run += 1
        it = self.it_done if it is None else it
        lr = lr_poly(hp.lr, it, hp.num_steps, hp.power)
        lr_T = lr_poly(hp.lr_T, it, hp.num_steps, hp.power)
        st = ops.stream_ptr()
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        labels = list(label) if isinstance(label, (list, tuple)) else [label]
        if len(images) != hp.iter_size or len(labels) != hp.iter_size:
            raise ValueError(f"step() needs {hp.iter_size} micro-batch(es) (hp.iter_size), got {len(images)}")
        # 1. inner W loop (NTM grads start from zero each iteration: optimizer_t*.zero_grad(), :314-318)
        # It only feeds the head (T, W) and the Adam step: it runs on the side stream, ahead of the frozen model's forward, instead
        # of in front of the trainable forward (150 us of a single-block kernel off the critical path).
        main, side = torch.cuda.current_stream(), side_stream(self.dev)
        ev0 = torch.cuda.Event()
        ev0.record(main)                       # the previous step's Adam / head kernels (main stream) come first
        with torch.cuda.stream(side):
            side.wait_event(ev0)
            self._ntm_grad_flat.zero_()
            ni = self.inner_desc
            ni.step0, ni.lr = self.inner_steps * self.it_done, lr_T
            L.call("simt_ntm_inner_loop", C.byref(ni), side.cuda_stream)
        flat = self.plan.flat_grad
        for mi, (img, lab) in enumerate(zip(images, labels)):
            last = mi == hp.iter_size - 1
            self._micro_batch(img, lab, st)
            # 5. trunk backward (+ 6. data-parallel mean of the gradients, bucket by bucket, on a side stream)
            if self._early_sgd:            # (with a reducer: the exchange completes on the side stream too, ahead of its SGD)
                with trace.range("backward (+ exchange + early optimiser step)"):
                    self._backward_early_sgd(lr, st)
                continue
            if self.reducer is not None and hp.iter_size == 1:
                self.reducer.start()
                with trace.range("backward (+ bucketed exchange)"):
                    self.plan.backward(hook=self.reducer.ready_upto)
                torch.cuda.current_stream().wait_event(self._ev_post)      # the NTM gradients it exchanges last (side stream)
                with trace.range("exchange (flush + wait)"):
                    self.reducer.finish()
                continue
            with trace.range("backward"):
                self.plan.backward()
            if hp.iter_size > 1:      # loss.backward() accumulates into .grad (:428): keep the running sum beside the plan's buffer
                if mi == 0:
                    self._grad_acc.copy_(flat)
                elif not last:
                    L.call("simt_vec_acc", self._grad_acc.data_ptr(), flat.data_ptr(), flat.numel(), 1, st)
                else:
                    L.call("simt_vec_acc", flat.data_ptr(), self._grad_acc.data_ptr(), flat.numel(), 1, st)
                    if self.reducer is not None:   # one exchange of the accumulated gradient (no overlap with backward)
                        torch.cuda.current_stream().wait_event(self._ev_post)
                        self.reducer.start()
                        self.reducer.finish()
        # 7. optimisers
        if not self._early_sgd:
            with trace.range("optimiser"):
                self._sgd(lr, st)
        torch.cuda.current_stream().wait_event(self._ev_post)      # regularisers (side stream): NTM gradients and the losses are final
        for k in range(2):
            ops.adam_step(self.ntm[k], self.ntm_grad[k], self.ntm_m[k], self.ntm_v[k], lr=lr_T, step=self.it_done + 1, skip_if=self._skip_word)
        if not self._early_sgd:
            self.plan.repack()
        self.it_done += 1
        return self.lout

    def _sgd(self, lr, st):
        d = self.sgd_desc
        d.lr[0], d.lr[1] = lr, lr * 10.0
        d.wd[0], d.wd[1] = self.hp.weight_decay, self.hp.weight_decay
        d.first_step = 1 if self.it_done == 0 else 0
        L.call("simt_sgd_multi", C.byref(d), st)

    def _backward_early_sgd(self, lr, st):
        """Backward with the SGD step and the re-pack of the updated layers enqueued on the side stream once the block that produced
        the last applied gradient has been enqueued COMPLETELY (its input-gradient convs read the packed weights the re-pack
        overwrites; `_early_cut` is the first hook point at or after that block's end).  The side stream first waits for the main
        stream's launches up to that point; the list's final join makes the main stream wait for all of it."""
        main, side = torch.cuda.current_stream(), side_stream(self.dev)
        done = [False]
        red = self.reducer
        if red is not None:
            red.start()

        def hook(n, ev):
            if red is not None and not done[0]:
                red.ready_upto(n, ev)      # data parallel: the buckets whose gradients are final start their all-reduce
            if done[0] or n < self._early_cut:
                return
            done[0] = True
            ev_main = torch.cuda.Event()
            ev_main.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_main)
                if red is not None:
                    # last buckets + NTM gradients, then THIS stream waits for the exchange: the main stream keeps walking through
                    # layer2 / layer1 / the stem (rank-local gradients) while the mean gradients arrive and are applied
                    red.finish()
                self._sgd(lr, side.cuda_stream)
                self._pack_applied.run()
        self.plan.backward(hook=hook)
        assert done[0]

    def state_dict(self):
        """Host copy of the trainable model's state (the reference's `model.state_dict()` of :449,461): every key, NCHW fp32, with
        `num_batches_tracked` = number of train-mode forwards so far (nn.BatchNorm2d bumps it once per forward)."""
        sd = {}
        for k, v in self.params.items():
            sd[k] = (v.detach().cpu() if not k.endswith("num_batches_tracked") else
                     torch.tensor(int(v.item()) + self.it_done * self.hp.iter_size, dtype=torch.long))
        return sd

    def timed_lists(self):
        """The launch lists of one iteration for per-kernel timing (bench.py): the forwards as the step launches them (paired: the stem im2col +
        one list for both networks; otherwise frozen forward, trainable forward), then the backward."""
        if self._fwd_order == "pair" and not self._fixed_graph:
            if self._fwd_both is None:
                self._fwd_both = self._paired_forwards()
            head = LaunchList()
            head.items = [self.plan.fwd_list.items[0]]
            return [head, self._fwd_both, self.plan.bwd_list]
        return [self.fixed.fwd_list, self.plan.fwd_list, self.plan.bwd_list]

    def losses(self):
        """Host copy of the scalars of the last step.  Synchronises the device; LOCAL (no collective: under data parallelism the
        bad-label count travels with the gradient exchange, so a rank may call this alone, e.g. rank 0 for logging -- but a rank that
        raises alone leaves its peers blocked in their next all-reduce, so training loops call it on every rank at the same iterations).
        Raises ValueError for labels outside [0, C) other than 255 seen in ANY micro-batch on ANY rank since the last call
        (utils/loss.py:36 raises at once), RuntimeError if a fused BatchNorm launch timed out (TrunkPlan.fbn_error)."""
        v = self.lout.cpu().tolist()                  # ONE device-to-host copy: the scalars and the bad-label count (lout[12])
        self.plan.raise_on_fbn_error()
        # accumulated by simt_ntm_post over every micro-batch since the last call; data parallel: the exchange leaves total / world on every rank
        # (the counter is cumulative and never reset on the device: a local reset by one rank would turn the next mean into a fraction on all)
        bad = int(round(v[12] * (self.reducer.world if self.reducer is not None else 1))) - self._bad_reported
        self._bad_reported += max(bad, 0)
        if bad > 0:      # utils/loss.py:36 / nn.CrossEntropyLoss raise on such a target; the kernels skip the pixel and count it
            raise ValueError(f"{bad} label value(s) outside [0, {self.hp.num_classes}) that are not the ignore value 255")
        keys = ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex", "volume", "anchor", "vol_ok"]
        return dict(zip(keys, v))


class WarmupTrainer:
    """The warm-up stage of the reference (tools/trainV1_warmup.py:156-256) on gfx950: DeeplabMulti(num_classes) without
    open-set heads, loss = CE(up(pred2), label) + lambda_seg * CE(up(pred1), label) with ignore_index 255 (:217-224), SGD over
    conv1 ... layer4 with the duplicate listings of `optim_parameters(args, warmup=True)` + heads at 10x lr (:192-193).
    Same engine as SimTTrainer: TrunkPlan forward/backward, the fused head kernels in mode 1, simt_sgd_multi."""

    def __init__(self, state, hp, B, H, W, *, dtype=torch.bfloat16, device="cuda:0", process_group=None, layers=None):
        self.hp, self.B, self.H, self.W, self.dtype = hp, B, H, W, dtype
        dev = self.dev = torch.device(device)
        self.pg = process_group
        Cn = self.C = hp.num_classes
        f32 = torch.float32
        self.params = {k: v.detach().to(dev, f32 if v.dtype != torch.long else torch.long).clone() for k, v in state.items()}
        kw = {"layers": layers} if layers is not None else {}
        self.plan = TrunkPlan(self.params, B, H, W, multi_heads(Cn, 0, False), dtype=dtype, train=True,
                              data_parallel=process_group is not None, **kw)
        h, w = self.plan.heads[1].h, self.plan.heads[1].w
        lib = L.load()
        self.part = torch.zeros(lib.simt_head_nblk(B, H, W), lib.simt_head_part_floats(Cn, Cn), device=dev)
        self.keys = torch.zeros(lib.simt_head_keys_count(), device=dev, dtype=torch.int64)
        self.hout = torch.zeros(lib.simt_head_hout_floats(Cn, Cn), device=dev)
        # labels outside [0, C) other than 255, ACCUMULATED over every micro-batch of every step until losses() reads and clears it (hout[15]
        # itself is overwritten by each head launch; nn.CrossEntropyLoss(ignore_index=255) raises on the first one, trainV1_warmup.py:217-224)
        self.bad_labels = torch.zeros(1, device=dev)
        self._bad_reported = 0
        self.QP = ops.round_up(Cn, 8)
        self.g1 = torch.zeros(2, B, H, w, self.QP, device=dev)
        self.label = torch.zeros(B, H, W, device=dev, dtype=torch.int64)
        hd = L.HeadDesc()
        d1, d2 = self.plan.dlogits["x1"], self.plan.dlogits["x2"]
        hd.pred1, hd.pred2, hd.fixp, hd.label = self.plan.out["x1"].data_ptr(), self.plan.out["x2"].data_ptr(), None, self.label.data_ptr()
        hd.T1, hd.T2 = None, None
        hd.part, hd.keys, hd.hout, hd.g1 = self.part.data_ptr(), self.keys.data_ptr(), self.hout.data_ptr(), self.g1.data_ptr()
        hd.dpred1_f32, hd.dpred2_f32, hd.dpred1_t, hd.dpred2_t = None, None, d1.data_ptr(), d2.data_ptr()
        hd.B, hd.h, hd.w, hd.H, hd.W, hd.C, hd.Q = B, h, w, H, W, Cn, Cn
        hd.ldp, hd.ldf, hd.QP, hd.ld_f32, hd.ld_t = self.plan.ldp["x1"], self.plan.ldp["x1"], self.QP, 0, d1.shape[1]
        hd.grad_dtype = ops.dt_code(dtype)
        hd.th_high, hd.th_low, hd.lambda_seg, hd.lambda_place, hd.gscale = 2.0, -1.0, hp.lambda_seg, 0.0, 1.0 / hp.iter_size
        hd.mode = 1
        self.head_desc = hd
        SimTTrainer._build_sgd(self, roots=("conv1", "layer1", "layer2", "layer3", "layer4"))
        self.it_done = 0
        self.reducer = None
        if self.pg is not None:
            from .dp import BucketReducer, make_buckets
            sizes = {n: k for n, (_o, k) in self.plan.grad_offsets.items()}
            buckets = make_buckets(self.plan.grad_order, sizes, self.plan.grad_ready, bucket_elems=8 << 20)
            self.reducer = BucketReducer(self.plan.flat_grad, buckets, group=self.pg, extra=[self.bad_labels])
        self._grad_acc = None

    def step(self, image, label, it=None):
        """One micro-batch, or sequences of hp.iter_size micro-batches (trainV1_warmup.py:212-231: loss / iter_size,
        gradients accumulated, one optimiser step)."""
        hp = self.hp
        it = self.it_done if it is None else it
        lr = lr_poly(hp.lr, it, hp.num_steps, hp.power)
        st = ops.stream_ptr()
        images = list(image) if isinstance(image, (list, tuple)) else [image]
        labels = list(label) if isinstance(label, (list, tuple)) else [label]
        if len(images) != hp.iter_size or len(labels) != hp.iter_size:
            raise ValueError(f"step() needs {hp.iter_size} micro-batch(es) (hp.iter_size), got {len(images)}")
        flat = self.plan.flat_grad
        for mi, (img, lab) in enumerate(zip(images, labels)):
            self.plan.x_in.copy_(img, non_blocking=True)
            self.label.copy_(lab, non_blocking=True)
            self.plan.forward()
            L.call("simt_head_loss", C.byref(self.head_desc), st)
            L.call("simt_vec_acc", self.bad_labels.data_ptr(), self.hout.data_ptr() + 4 * 15, 1, 1, st)      # every micro-batch counts
            L.call("simt_head_grad", C.byref(self.head_desc), st)
            if self.reducer is not None and hp.iter_size == 1:
                self.reducer.start()
                self.plan.backward(hook=self.reducer.ready_upto)
                self.reducer.finish()
                continue
            self.plan.backward()
            if hp.iter_size > 1:
                if self._grad_acc is None:
                    self._grad_acc = torch.zeros_like(flat)
                if mi == 0:
                    self._grad_acc.copy_(flat)
                elif mi < hp.iter_size - 1:
                    L.call("simt_vec_acc", self._grad_acc.data_ptr(), flat.data_ptr(), flat.numel(), 1, st)
                else:
                    L.call("simt_vec_acc", flat.data_ptr(), self._grad_acc.data_ptr(), flat.numel(), 1, st)
                    if self.reducer is not None:
                        self.reducer.start()
                        self.reducer.finish()
        d = self.sgd_desc
        d.lr[0], d.lr[1] = lr, lr * 10.0
        d.wd[0], d.wd[1] = hp.weight_decay, hp.weight_decay
        d.first_step = 1 if self.it_done == 0 else 0
        L.call("simt_sgd_multi", C.byref(d), st)
        self.plan.repack()
        self.it_done += 1
        return self.hout

    def state_dict(self):
        sd = {}
        for k, v in self.params.items():
            sd[k] = (v.detach().cpu() if not k.endswith("num_batches_tracked") else
                     torch.tensor(int(v.item()) + self.it_done * self.hp.iter_size, dtype=torch.long))
        return sd

    def losses(self):
        """Host copy of the last micro-batch's scalars; local (see SimTTrainer.losses).  The bad-label count covers EVERY micro-batch of every
        step since the last call (a device-side accumulator; under data parallelism it rides in the gradient exchange)."""
        v = torch.cat([self.hout[:16], self.bad_labels]).cpu().tolist()
        self.plan.raise_on_fbn_error()
        bad = int(round(v[16] * (self.reducer.world if self.reducer is not None else 1))) - self._bad_reported      # cumulative, never reset on the device
        self._bad_reported += max(bad, 0)
        if bad > 0:      # nn.CrossEntropyLoss(ignore_index=255) raises on such a target (trainV1_warmup.py:217-224)
            raise ValueError(f"{bad} label value(s) outside [0, {self.hp.num_classes}) that are not the ignore value 255")
        # `loss = loss / args.iter_size` (trainV1_warmup.py:227): the reported total is the scaled one, like SimTTrainer's
        return {"total": v[14] / self.hp.iter_size, "loss_seg1": v[0], "loss_seg2": v[1]}
