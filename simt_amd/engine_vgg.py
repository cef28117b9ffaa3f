"""Execution plan for DeepLab-VGG16 (reference model/deeplab_vgg.py:24-51; BASELINE config 5) on the same HIP kernels.

Architecture restated from the reference file + torchvision's public VGG16 definition (the reference file itself is
Python-2-only and needs torchvision, so it cannot be imported: parity for this trunk is UNPINNED):
vgg16.features with pool4 / pool5 removed and conv5_x at dilation 2 (`:34-38`), fc6 = 3x3 conv 512->1024 dilation 4,
fc7 = 3x3 conv 1024->1024 dilation 4, each followed by ReLU (`:40-43`), then the 2-branch (early-return) ASPP classifier
on 1024 channels (`:6-21`).  No BatchNorm: every conv is bias + ReLU in the GEMM epilogue; the backward applies the ReLU
mask of the producing layer in the dgrad epilogue (simt_conv_desc.mask) or inside the 2x2 max-pool backward.
"""
import torch

from . import ops
from .engine import HeadCfg, TrunkPlan, layout_flat_grads

# (features index, cin, cout, dilation, pool_after)
VGG_LAYERS = [(0, 3, 64, 1, False), (2, 64, 64, 1, True), (5, 64, 128, 1, False), (7, 128, 128, 1, True),
              (10, 128, 256, 1, False), (12, 256, 256, 1, False), (14, 256, 256, 1, True),
              (17, 256, 512, 1, False), (19, 512, 512, 1, False), (21, 512, 512, 1, False),
              (23, 512, 512, 2, False), (25, 512, 512, 2, False), (27, 512, 512, 2, False),
              (29, 512, 1024, 4, False), (31, 1024, 1024, 4, False)]


def vgg_state_shapes(num_classes, layers=VGG_LAYERS):
    """state_dict of DeeplabVGG: features.N.{weight,bias} (N = position in the rebuilt Sequential), classifier.conv2d_list.i."""
    sh = {}
    for (idx, cin, cout, _d, _p) in layers:
        sh[f"features.{idx}.weight"] = (cout, cin, 3, 3)
        sh[f"features.{idx}.bias"] = (cout,)
    cl = layers[-1][2]
    for i in range(4):
        sh[f"classifier.conv2d_list.{i}.weight"] = (num_classes, cl, 3, 3)
        sh[f"classifier.conv2d_list.{i}.bias"] = (num_classes,)
    return sh


def vgg_head(num_classes, cin=1024):
    return [HeadCfg("x", 0, cin, [("classifier", num_classes)], (6, 12))]


class VggPlan(TrunkPlan):
    def __init__(self, params, B, H, W, num_classes, *, dtype=torch.bfloat16, train=True, device=None, vgg_layers=VGG_LAYERS, data_parallel=False):
        self.vgg_layers = list(vgg_layers)
        # (weight-gradient split counts planned for the whole chip: no BatchNorm / dgrad chain here for them to hide behind -- ops.wgrad_plan)
        with ops.wgrad_plan(256, 6):
            super().__init__(params, B, H, W, vgg_head(num_classes, self.vgg_layers[-1][2]), dtype=dtype, train=train, layers=(0, 0, 0, 0),
                             device=device, data_parallel=data_parallel)

    # ------------------------------------------------------------------ forward
    def _build_forward(self):
        B, dt, f = self.B, self.dtype, self.fwd_list
        self.x_in = self.new(B, 3, self.H, self.W, dtype=torch.float32)
        self.recs = []
        Hc, Wc = self.H, self.W
        x = None
        for li, (idx, cin, cout, dil, pool) in enumerate(self.vgg_layers):
            name = f"features.{idx}"
            M = B * Hc * Wc
            y = self.new(M, cout)
            bias = self.p[name + ".bias"]
            rec = {"name": name, "cin": cin, "cout": cout, "dil": dil, "H": Hc, "W": Wc, "M": M, "y": y, "pool": pool}
            if li == 0:
                ks = self.kq                                   # K = 27 padded to one 128-byte stage
                A = self.new(M, ks)
                f.add("simt_im2col_stem", self.x_in.data_ptr(), A.data_ptr(), B, 3, Hc, Wc, Hc, Wc, 3, 3, 1, 1, ks, ops.dt_code(dt))
                tile = ops.pick_tile_n(cout, dt)
                npad = ops.round_up(cout, tile)
                wp = self.new(npad, ks, zero=True)
                self.pack_list.add("simt_pack_weight", self.p[name + ".weight"].data_ptr(), wp.data_ptr(), cout, 27, 1, 0, 0, ks, 0, 0,
                                   None, ops.dt_code(dt))
                self._conv(f, A, (wp, tile, npad), y, Bn=1, Hi=1, Wi=M, Cin=ks, Ho=1, Wo=M, Cout=cout, taps=[(0, 0)], bias=bias,
                           relu=True, alg_k=27)
                rec.update(x=A, stem=True, ks=ks)
            else:
                wi = self._plan_pack(name, cout, cin, 3)
                self._conv(f, x, wi, y, Bn=B, Hi=Hc, Wi=Wc, Cin=cin, Ho=Hc, Wo=Wc, Cout=cout, taps=ops.conv_taps(3, 3, dil, dil),
                           bias=bias, relu=True)
                rec.update(x=x, stem=False)
            x = y
            if pool:
                Hp, Wp = Hc // 2, Wc // 2
                p = self.new(B * Hp * Wp, cout)
                pidx = self.new(B * Hp * Wp, cout, dtype=torch.uint8)
                f.add("simt_maxpool2", y.data_ptr(), p.data_ptr(), pidx.data_ptr(), B, Hc, Wc, cout, ops.dt_code(dt))
                rec.update(p=p, pidx=pidx)
                x, Hc, Wc = p, Hp, Wp
            self.recs.append(rec)
        self.feat_hw = (Hc, Wc)
        self._build_head_fwd(self.heads[0], x, Hc, Wc, self.vgg_layers[-1][2])

    # ------------------------------------------------------------------ gradients
    def grad_param_names(self):
        names = []
        for (idx, *_r) in self.vgg_layers:
            names += [f"features.{idx}.weight", f"features.{idx}.bias"]
        for i in range(len(self.heads[0].dilations)):
            names += [f"classifier.conv2d_list.{i}.weight", f"classifier.conv2d_list.{i}.bias"]
        return names

    def _alloc_grads(self, grad_names=None):
        order = [f"classifier.conv2d_list.{i}.{k}" for i in range(len(self.heads[0].dilations)) for k in ("weight", "bias")]
        for (idx, *_r) in reversed(self.vgg_layers):
            order += [f"features.{idx}.weight", f"features.{idx}.bias"]
        layout_flat_grads(self, order)

    # ------------------------------------------------------------------ backward
    def _build_backward(self):
        B, dt, b = self.B, self.dtype, self.bwd_list
        hd = self.heads[0]
        self._slab_cap, self._bnb_cap = 1, 1
        for rec in self.recs:
            kt = rec["ks"] if rec["stem"] else 9 * rec["cin"]
            self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(rec["M"], rec["cout"], kt, dt) * rec["cout"] * kt)
        Mh = B * hd.h * hd.w
        cd, kt = ops.round_up(hd.Q, 8), len(hd.taps) * hd.cin
        self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(Mh, cd, kt, dt) * cd * kt)
        if getattr(hd, "expanded", False):
            self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(Mh, hd.nexp, hd.cin, dt) * hd.nexp * hd.cin)
        hd.ck = ops.round_up(hd.Q, self.kq)
        self.dlogits = {hd.name: self.new(Mh, hd.ck, zero=True)}
        self.grad_ready, self.bwd_marks = {}, {}
        e0 = b.record(0)
        b.wait(e0, 1)
        last = self.recs[-1]
        hd.mask = last["y"]                      # ReLU of fc7
        g = self._build_head_bwd(hd, None, last["M"], last["cout"], 0)
        for li in range(len(self.recs) - 1, -1, -1):
            rec = self.recs[li]
            name, M, cin, cout = rec["name"], rec["M"], rec["cin"], rec["cout"]
            Hc, Wc = rec["H"], rec["W"]
            # g = d loss / d (conv output before ReLU), already masked
            b.wait(b.record(0), 1)
            b.add("simt_colsum_wide", g.data_ptr(), self.grads[name + ".bias"].data_ptr(), M, cout, cout, ops.dt_code(dt), stream=1)
            self.grad_ready[name + ".bias"] = len(b)
            if rec["stem"]:
                self._wgrad(b, g, rec["x"], None, Bn=1, Hi=1, Wi=M, Cin=rec["ks"], Ho=1, Wo=M, Cd=cout, ldd=cout, taps=[(0, 0)],
                            stride=1, parts=[(name + ".weight", 0, 0, cout, 1, 27)])
                break
            t3 = ops.conv_taps(3, 3, rec["dil"], rec["dil"])
            self._wgrad(b, g, rec["x"], None, Bn=B, Hi=Hc, Wi=Wc, Cin=cin, Ho=Hc, Wo=Wc, Cd=cout, ldd=cout, taps=t3, stride=1,
                        parts=[(name + ".weight", 0, 0, cout, 9, cin)])
            wt = self._plan_pack_t(name, cout, cin, 3)
            assert wt[3] == cout
            prev = self.recs[li - 1]
            dx = self.new(M, cin)
            if prev["pool"]:
                self._conv(b, g, wt[:3], dx, Bn=B, Hi=Hc, Wi=Wc, Cin=cout, Ho=Hc, Wo=Wc, Cout=cin, taps=[(-a, -c) for (a, c) in t3])
                gp = self.new(prev["M"], prev["cout"])
                b.add("simt_maxpool2_bwd", dx.data_ptr(), prev["pidx"].data_ptr(), prev["y"].data_ptr(), gp.data_ptr(), B, prev["H"],
                      prev["W"], prev["cout"], ops.dt_code(dt))
                g = gp
            else:
                self._conv(b, g, wt[:3], dx, Bn=B, Hi=Hc, Wi=Wc, Cin=cout, Ho=Hc, Wo=Wc, Cout=cin, taps=[(-a, -c) for (a, c) in t3],
                           mask=prev["y"])
                g = dx
        b.wait(b.record(1), 0)
