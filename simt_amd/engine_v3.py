"""Execution plan for DeepLabv3 (reference model/deeplabv3.py:9-138; BASELINE config 4) on the HIP kernels.

Architecture restated from the reference file + torchvision's public ResNet-50 definition (the reference file needs
torchvision, which is not installed, so it cannot be imported: parity for this model is UNPINNED):

  ResNet_50 (`:9-21`)  conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2 (floor) -> layer1..layer3 of torchvision Bottlenecks
                       (stride on the 3x3 conv, downsample in block 0 of every layer) -> [B,1024,H/16,W/16]
  ASSP (`:23-108`)     1x1, 3x3 d6, 3x3 d12, 3x3 d18, 1x1 -- each conv(no bias) -> BN -> ReLU; the bilinear resize of the
                       fifth branch to its own size (`:102`) is the identity; cat 5x256 -> 1x1 -> BN -> ReLU
  DeepLabv3 (`:111-138`) 1x1 conv 256->nc (|| 256->openc) with bias, then F.interpolate(bilinear, align_corners=False)
                       to the input size INSIDE the model: one tensor [B, nc+openc, H, W].

Every BatchNorm is a standard trainable one (batch statistics in train mode, running statistics updated, gradients for
gamma and beta).  The five branch activations are stored as five planes [5][M][256] so that the concat never exists: the
1x1 `convf` over 1280 channels is run as a 5-tap convolution over the image [1][5*B*h][w][256] whose taps step one plane
(B*h rows) each; its weight gradient is the matching 5-tap wgrad and its input gradient five 1x1 GEMMs on row blocks of the
transposed operand.
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import ops
from .engine import LaunchList, TrunkPlan, layout_flat_grads

R50 = "resnet.resnet_50."


def v3_block_specs(layers=(3, 4, 6), width=64):
    """[(name, inplanes, planes, stride (on conv2), has_downsample)] of torchvision's layer1..layer3."""
    specs, inpl = [], width
    for li, n in enumerate(layers):
        planes = width * 2 ** li
        for bi in range(n):
            specs.append((f"{R50}layer{li + 1}.{bi}", inpl, planes, (1 if li == 0 else 2) if bi == 0 else 1, bi == 0))
            inpl = planes * 4
    return specs


def v3_geometry(H, W):
    H0, W0 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = (H0 + 2 - 3) // 2 + 1, (W0 + 2 - 3) // 2 + 1
    return (H0, W0), (Hp, Wp)


def _bn_shapes(sh, name, c, tracked=True):
    sh[name + ".weight"] = (c,)
    sh[name + ".bias"] = (c,)
    sh[name + ".running_mean"] = (c,)
    sh[name + ".running_var"] = (c,)
    if tracked:
        sh[name + ".num_batches_tracked"] = ()


def v3_state_shapes(nc, openc=0, openset=False, layers=(3, 4, 6), width=64, assp_ch=256):
    """state_dict shapes of the part of DeepLabv3 that forward() uses (layer4 / fc of resnet50 exist in the module but never
    run), in module order."""
    sh = {R50 + "conv1.weight": (width, 3, 7, 7)}
    _bn_shapes(sh, R50 + "bn1", width)
    for (name, inpl, planes, _s, down) in v3_block_specs(layers, width):
        sh[name + ".conv1.weight"] = (planes, inpl, 1, 1)
        _bn_shapes(sh, name + ".bn1", planes)
        sh[name + ".conv2.weight"] = (planes, planes, 3, 3)
        _bn_shapes(sh, name + ".bn2", planes)
        sh[name + ".conv3.weight"] = (planes * 4, planes, 1, 1)
        _bn_shapes(sh, name + ".bn3", planes * 4)
        if down:
            sh[name + ".downsample.0.weight"] = (planes * 4, inpl, 1, 1)
            _bn_shapes(sh, name + ".downsample.1", planes * 4)
    cin = width * 2 ** (len(layers) - 1) * 4
    for i, k in ((1, 1), (2, 3), (3, 3), (4, 3), (5, 1)):
        sh[f"assp.conv{i}.weight"] = (assp_ch, cin, k, k)
        _bn_shapes(sh, f"assp.bn{i}", assp_ch)
    sh["assp.convf.weight"] = (assp_ch, 5 * assp_ch, 1, 1)
    _bn_shapes(sh, "assp.bnf", assp_ch)
    sh["conv.weight"], sh["conv.bias"] = (nc, assp_ch, 1, 1), (nc,)
    if openset:
        sh["conv_1.weight"], sh["conv_1.bias"] = (openc, assp_ch, 1, 1), (openc,)
    return sh


ASSP_BRANCHES = ((1, 1, 1), (2, 3, 6), (3, 3, 12), (4, 3, 18), (5, 1, 1))   # (index, kernel, dilation)


class V3Plan(TrunkPlan):
    def __init__(self, params, B, H, W, nc, openc=0, openset=False, *, dtype=torch.bfloat16, train=True, device=None,
                 layers=(3, 4, 6), width=64, assp_ch=256, data_parallel=False):
        self.nc, self.openc, self.openset = nc, (openc if openset else 0), openset
        self.Q = self.nc + self.openc
        self.v3_layers, self.width, self.assp_ch = tuple(layers), width, assp_ch
        self.groups = [("conv", nc)] + ([("conv_1", openc)] if openset else [])
        super().__init__(params, B, H, W, [], dtype=dtype, train=train, layers=(0, 0, 0, 0), device=device, data_parallel=data_parallel)

    # ------------------------------------------------------------------ helpers
    def _bn_conv_relu(self, f, x, cname, bname, y, a, *, M, cin, cout, Hi, Wi, Ho, Wo, k, dil=1, stride=1, Bn=None, taps=None,
                      wi=None, defer_apply=False):
        """train: y = conv(x) (+ stats) ; a = relu(bn(y)).  eval: a = relu(conv(x) * scale + shift) in one launch.
        defer_apply (train): leave the normalise + ReLU to the caller (the consuming conv may apply it in its operand path: TrunkPlan._conv inbn)."""
        Bn = Bn or self.B
        taps = taps or (ops.conv_taps(3, 3, dil, dil) if k == 3 else [(0, 0)])
        if self.train:
            s = self._new_bn(bname, M, cout)
            wi = wi or self._plan_pack(cname, cout, cin, k)
            self._conv(f, x, wi, y, Bn=Bn, Hi=Hi, Wi=Wi, Cin=cin, Ho=Ho, Wo=Wo, Cout=cout, taps=taps, stride=stride,
                       stats=s["part"])
            self._bn_train(f, bname, y, M, cout)
            if a is not None and not defer_apply:
                f.add("simt_bn_apply", y.data_ptr(), s["scale"].data_ptr(), s["shift"].data_ptr(), None, None, None, None,
                      a.data_ptr(), M, cout, 1, ops.dt_code(self.dtype))
        else:
            _, sh = self._plan_fold(bname, cout)
            wi = wi or self._plan_pack(cname, cout, cin, k, scale_bn=bname)
            self._conv(f, x, wi, a, Bn=Bn, Hi=Hi, Wi=Wi, Cin=cin, Ho=Ho, Wo=Wo, Cout=cout, taps=taps, stride=stride, bias=sh,
                       relu=True)

    # ------------------------------------------------------------------ forward
    def _build_forward(self):
        B, dt, f, wd = self.B, self.dtype, self.fwd_list, self.width
        (H0, W0), (Hp, Wp) = v3_geometry(self.H, self.W)
        self.H0, self.W0, self.Hp, self.Wp = H0, W0, Hp, Wp
        M0, Mp = B * H0 * W0, B * Hp * Wp
        self.x_in = self.new(B, 3, self.H, self.W, dtype=torch.float32)
        self.saved = {}
        KS = 192
        A = self.new(M0, KS)
        f.add("simt_im2col_stem", self.x_in.data_ptr(), A.data_ptr(), B, 3, self.H, self.W, H0, W0, 7, 7, 2, 3, KS, ops.dt_code(dt))
        y0, pool, pidx = self.new(M0, wd), self.new(Mp, wd), self.new(Mp, wd, dtype=torch.uint8)
        self.saved.update({"stem.A": A, "stem.y": y0, "stem.pool": pool, "stem.idx": pidx})
        tile = ops.pick_tile_n(wd, dt)
        npad = ops.round_up(wd, tile)
        wp = self.new(npad, KS, zero=True)
        if self.train:
            cs = None
        else:
            cs, sh0 = self._plan_fold(R50 + "bn1", wd)
        self.pack_list.add("simt_pack_weight", self.p[R50 + "conv1.weight"].data_ptr(), wp.data_ptr(), wd, 147, 1, 0, 0, KS, 0, 0,
                           cs.data_ptr() if cs is not None else None, ops.dt_code(dt))
        if self.train:
            s = self._new_bn(R50 + "bn1", M0, wd)
            self._conv(f, A, (wp, tile, npad), y0, Bn=1, Hi=1, Wi=M0, Cin=KS, Ho=1, Wo=M0, Cout=wd, taps=[(0, 0)], stats=s["part"],
                       alg_k=147)
            self._bn_train(f, R50 + "bn1", y0, M0, wd)
            sc, sh0 = s["scale"], s["shift"]
        else:
            self._conv(f, A, (wp, tile, npad), y0, Bn=1, Hi=1, Wi=M0, Cin=KS, Ho=1, Wo=M0, Cout=wd, taps=[(0, 0)], bias=sh0,
                       relu=True, alg_k=147)
            sc, sh0 = self.new(wd, dtype=torch.float32), self.new(wd, dtype=torch.float32, zero=True)
            sc.fill_(1.0)
        f.add("simt_bn_relu_maxpool", y0.data_ptr(), sc.data_ptr(), sh0.data_ptr(), pool.data_ptr(), pidx.data_ptr(), B, H0, W0, wd,
              Hp, Wp, ops.dt_code(dt))
        # ---- torchvision Bottlenecks (stride on conv2)
        x, Hc, Wc = pool, Hp, Wp
        self.block_io = []
        for (name, inpl, planes, stride, down) in v3_block_specs(self.v3_layers, wd):
            Ho, Wo = (Hc - 1) // stride + 1, (Wc - 1) // stride + 1
            Mi, Mo, c4 = B * Hc * Wc, B * Ho * Wo, planes * 4
            rec = {"name": name, "x": x, "Hi": Hc, "Wi": Wc, "Ho": Ho, "Wo": Wo, "inpl": inpl, "planes": planes, "stride": stride,
                   "down": down, "Mi": Mi, "Mo": Mo}
            tr = self.train
            y1, a1 = (self.new(Mi, planes) if tr else None), self.new(Mi, planes)
            y2, a2 = (self.new(Mo, planes) if tr else None), self.new(Mo, planes)
            z = self.new(Mo, c4)
            self._bn_conv_relu(f, x, name + ".conv1", name + ".bn1", y1, a1, M=Mi, cin=inpl, cout=planes, Hi=Hc, Wi=Wc, Ho=Hc, Wo=Wc, k=1)
            self._bn_conv_relu(f, a1, name + ".conv2", name + ".bn2", y2, a2, M=Mo, cin=planes, cout=planes, Hi=Hc, Wi=Wc, Ho=Ho,
                               Wo=Wo, k=3, stride=stride, defer_apply=tr)
            if tr:
                y3 = self.new(Mo, c4)
                s3 = self._new_bn(name + ".bn3", Mo, c4)
                w3 = self._plan_pack(name + ".conv3", c4, planes, 1)
                # bn2's normalise + ReLU in conv3's operand path where the row-streaming kernel takes conv3 (round 6; engine.TrunkPlan), else the
                # separate apply launch
                kw3 = dict(Bn=B, Hi=Ho, Wi=Wo, Cin=planes, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)], stats=s3["part"])
                s2 = self.bn[name + ".bn2"]
                inbn = (os.environ.get("SIMT_NO_INBN", "0") == "0"
                        and L.load().simt_conv_inbn_ok(C.byref(self._conv(LaunchList(), y2, w3, y3, **kw3))) != 0)
                if inbn:
                    self._conv(f, y2, w3, y3, inbn=(s2["scale"], s2["shift"], a2), **kw3)
                else:
                    f.add("simt_bn_apply", y2.data_ptr(), s2["scale"].data_ptr(), s2["shift"].data_ptr(), None, None, None, None,
                          a2.data_ptr(), Mo, planes, 1, ops.dt_code(dt))
                    self._conv(f, a2, w3, y3, **kw3)
                rec["inbn"] = bool(inbn)
                self._bn_train(f, name + ".bn3", y3, Mo, c4)
                zbits = self.new(Mo, c4 // 8, dtype=torch.uint8)      # ReLU mask of the block output, one bit per element
                rec.update(y1=y1, a1=a1, y2=y2, a2=a2, y3=y3, zbits=zbits)
                if down:
                    yd = self.new(Mo, c4)
                    sd = self._new_bn(name + ".downsample.1", Mo, c4)
                    wdn = self._plan_pack(name + ".downsample.0", c4, inpl, 1)
                    self._conv(f, x, wdn, yd, Bn=B, Hi=Hc, Wi=Wc, Cin=inpl, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)], stride=stride,
                               stats=sd["part"])
                    self._bn_train(f, name + ".downsample.1", yd, Mo, c4)
                    f.add("simt_bn_apply_bits", y3.data_ptr(), s3["scale"].data_ptr(), s3["shift"].data_ptr(), None, yd.data_ptr(),
                          sd["scale"].data_ptr(), sd["shift"].data_ptr(), z.data_ptr(), zbits.data_ptr(), Mo, c4, 1, ops.dt_code(dt),
                          tag="simt_bn_apply")
                    rec.update(yd=yd)
                else:
                    f.add("simt_bn_apply_bits", y3.data_ptr(), s3["scale"].data_ptr(), s3["shift"].data_ptr(), x.data_ptr(), None, None,
                          None, z.data_ptr(), zbits.data_ptr(), Mo, c4, 1, ops.dt_code(dt), tag="simt_bn_apply")
            else:
                res = x
                if down:
                    yd = self.new(Mo, c4)
                    _, shd = self._plan_fold(name + ".downsample.1", c4)
                    wdn = self._plan_pack(name + ".downsample.0", c4, inpl, 1, scale_bn=name + ".downsample.1")
                    self._conv(f, x, wdn, yd, Bn=B, Hi=Hc, Wi=Wc, Cin=inpl, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)], stride=stride,
                               bias=shd)
                    res = yd
                _, sh3 = self._plan_fold(name + ".bn3", c4)
                w3 = self._plan_pack(name + ".conv3", c4, planes, 1, scale_bn=name + ".bn3")
                self._conv(f, a2, w3, z, Bn=B, Hi=Ho, Wi=Wo, Cin=planes, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)], bias=sh3, res=res,
                           relu=True)
            rec.update(z=z, a1=a1, a2=a2)
            self.block_io.append(rec)
            x, Hc, Wc = z, Ho, Wo
        # ---- ASSP: five conv-BN-ReLU branches into five planes
        h, w, ac = Hc, Wc, self.assp_ch
        M = B * h * w
        cin = self.block_io[-1]["planes"] * 4
        self.feat, self.feat_hw, self.Mh, self.cfeat = x, (h, w), M, cin
        self.ycat = self.new(5, M, ac) if self.train else None
        self.acat = self.new(5, M, ac)
        for t, (i, k, dil) in enumerate(ASSP_BRANCHES):
            self._bn_conv_relu(f, x, f"assp.conv{i}", f"assp.bn{i}", self.ycat[t] if self.train else None, self.acat[t], M=M, cin=cin,
                               cout=ac, Hi=h, Wi=w, Ho=h, Wo=w, k=k, dil=dil)
        # convf over the (virtual) concat: 5 taps stepping one plane each
        self.ptaps = [(t * B * h, 0) for t in range(5)]
        self.yf = self.new(M, ac) if self.train else None
        self.af = self.new(M, ac)
        wf = self._plan_pack("assp.convf", ac, 5 * ac, 1, scale_bn=None if self.train else self._fold_name("assp.bnf", ac))
        self._bn_conv_relu(f, self.acat, "assp.convf", "assp.bnf", self.yf, self.af, M=M, cin=ac, cout=ac, Hi=5 * B * h, Wi=w,
                           Ho=B * h, Wo=w, k=1, Bn=1, taps=self.ptaps, wi=wf)
        # ---- 1x1 classifier(s) with bias, fp32 logits [M][ldp]; then the in-model bilinear upsample (align_corners=False)
        Q = self.Q
        tile = ops.pick_tile_n(Q)
        npad = ops.round_up(Q, tile)
        ldp = max(ops.round_up(Q, 8) if Q > 32 else 32, ops.round_up(Q, 4))
        wq = self.new(npad, ac, zero=True)
        self.bias_q = self.new(npad, dtype=torch.float32, zero=True)
        row = 0
        for prefix, cout in self.groups:
            self.pack_list.add("simt_pack_weight", self.p[prefix + ".weight"].data_ptr(), wq.data_ptr(), cout, ac, 1, row, 0, ac, 0, 0,
                               None, ops.dt_code(dt))
            self.pack_list.add("simt_vec_acc", self.bias_q.data_ptr() + 4 * row, self.p[prefix + ".bias"].data_ptr(), cout, 0)
            row += cout
        self.ldq = ldp
        self.logits = self.new(M, ldp, dtype=torch.float32, zero=True)
        self._conv(f, self.af, (wq, tile, npad), self.logits, Bn=B, Hi=h, Wi=w, Cin=ac, Ho=h, Wo=w, Cout=Q, taps=[(0, 0)],
                   bias=self.bias_q, ldy=ldp, Nstore=min(ldp, npad))
        self.out_full = self.new(B, Q, self.H, self.W, dtype=torch.float32)
        f.add("simt_upsample_nchw", self.logits.data_ptr(), B, h, w, ldp, Q, self.H, self.W, 0, self.out_full.data_ptr())

    def _fold_name(self, bname, c):
        self._plan_fold(bname, c)
        return bname

    # ------------------------------------------------------------------ gradients
    def grad_param_names(self):
        names = [R50 + "conv1.weight", R50 + "bn1.weight", R50 + "bn1.bias"]
        for (name, _i, _p, _s, down) in v3_block_specs(self.v3_layers, self.width):
            for j in (1, 2, 3):
                names += [f"{name}.conv{j}.weight", f"{name}.bn{j}.weight", f"{name}.bn{j}.bias"]
            if down:
                names += [f"{name}.downsample.0.weight", f"{name}.downsample.1.weight", f"{name}.downsample.1.bias"]
        for (i, _k, _d) in ASSP_BRANCHES:
            names += [f"assp.conv{i}.weight", f"assp.bn{i}.weight", f"assp.bn{i}.bias"]
        names += ["assp.convf.weight", "assp.bnf.weight", "assp.bnf.bias"]
        for prefix, _c in self.groups:
            names += [prefix + ".weight", prefix + ".bias"]
        return names

    def _alloc_grads(self, grad_names=None):
        order = list(reversed(self.grad_param_names()))         # reverse-topological: DP buckets become ready in order
        layout_flat_grads(self, order)

    # ------------------------------------------------------------------ backward
    def _bnb(self, b, *, dz, y, bname, dy, M, Cn, mask_mode, **kw):
        self._bn_bwd(b, dz=dz, y=y, bname=bname, dy=dy, M=M, Cn=Cn, mask_mode=mask_mode, affine=True, **kw)
        self.grad_ready[bname + ".weight"] = self.grad_ready[bname + ".bias"] = len(b)
        if kw.get("bname2"):
            self.grad_ready[kw["bname2"] + ".weight"] = self.grad_ready[kw["bname2"] + ".bias"] = len(b)

    def _build_backward(self):
        B, dt, b, ac = self.B, self.dtype, self.bwd_list, self.assp_ch
        h, w = self.feat_hw
        M, cin, Q = self.Mh, self.cfeat, self.Q
        # workspace capacities
        self._slab_cap, self._bnb_cap = 1, 1
        shapes = [(M, ops.round_up(Q, 8), ac), (M, ac, 5 * ac), (M, ac, cin), (M, ac, 9 * cin), (B * self.H0 * self.W0, self.width, 192)]
        bns = [(M, ac), (B * self.H0 * self.W0, self.width)]
        for rec in self.block_io:
            p, inpl = rec["planes"], rec["inpl"]
            shapes += [(rec["Mi"], p, inpl), (rec["Mo"], p, 9 * p), (rec["Mo"], 4 * p, p), (rec["Mo"], 4 * p, inpl)]
            bns += [(rec["Mi"], p), (rec["Mo"], p), (rec["Mo"], 4 * p)]
        for (m, cd, kt) in shapes:
            self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(m, cd, kt, dt) * cd * kt)
        if self._wgrad_grouped():       # one grouped launch per Bottleneck whose three (four) convs see the same pixels (engine._wgrad_group)
            for rec in self.block_io:
                if rec["Mi"] != rec["Mo"]:
                    continue
                p, inpl, m = rec["planes"], rec["inpl"], rec["Mo"]
                shp = [(p, inpl), (p, 9 * p), (4 * p, p)] + ([(4 * p, inpl)] if rec["down"] else [])
                tco = ops.wgrad_group_tile_co(m, shp)
                ns = ops.wgrad_group_nsplit(m, sum(ops.wgrad_tiles(m, cd, kt, tco) for cd, kt in shp))
                self._slab_cap = max(self._slab_cap, ns * sum(cd * kt for cd, kt in shp))
        for (m, cn) in bns:
            self._bnb_cap = max(self._bnb_cap, ops.bn_bwd_nblk(m, cn) * 3 * cn)
        self.grad_ready, self.bwd_marks = {}, {}
        # ---- upstream gradient [B,Q,H,W] fp32 -> adjoint of the upsample -> dlogits [M][ck] in the conv dtype (K-padded)
        self.dout_full = self.new(B, Q, self.H, self.W, dtype=torch.float32)
        ck = ops.round_up(Q, self.kq)
        dl = self.new(M, ck, zero=True)
        self.dlogits = {"x": dl}
        self.up_tmp = self.new(B, Q, self.H, w, dtype=torch.float32)        # x-folded intermediate of the separable adjoint
        b.add("simt_upsample_nchw_bwd", self.dout_full.data_ptr(), B, h, w, ck, Q, self.H, self.W, 0, dl.data_ptr(), ops.dt_code(dt),
              self.up_tmp.data_ptr())
        b.wait(b.record(0), 1)
        row, parts = 0, []
        for prefix, cout in self.groups:
            b.add("simt_colsum", dl.data_ptr() + row * self.esz, self.grads[prefix + ".bias"].data_ptr(), M, ck, cout, 0,
                  ops.dt_code(dt), stream=1)
            self.grad_ready[prefix + ".bias"] = len(b)
            parts.append((prefix + ".weight", row, 0, cout, 1, ac))
            row += cout
        self._wgrad(b, dl, self.af, None, Bn=B, Hi=h, Wi=w, Cin=ac, Ho=h, Wo=w, Cd=ops.round_up(Q, 8), ldd=ck, taps=[(0, 0)], stride=1,
                    parts=parts)
        tile = ops.pick_tile_n(ac, dt)
        npad = ops.round_up(ac, tile)
        wt = self.new(npad, ck, zero=True)
        row = 0
        for prefix, cout in self.groups:
            self.pack_list.add("simt_pack_weight", self.p[prefix + ".weight"].data_ptr(), wt.data_ptr(), cout, ac, 1, row, 0, ck, ck, 1,
                               None, ops.dt_code(dt))
            row += cout
        daf = self.new(M, ac)
        dyf = self.new(M, ac)
        coef = self.buf("bnb.coef", 3 * 2048, dtype=torch.float32)

        def conv_bnb(x, wi, dz, y, bname, dy, **kw):
            """dgrad GEMM + the backward of the (ReLU-masked, trainable-affine) BatchNorm its result belongs to: the reduce pass in the GEMM
            epilogue, and on a co-resident grid the whole BatchNorm backward in the launch (simt_fbn_desc mode 2)."""
            bnr = self._bnr(bname, y, 2)
            dsc = self._conv(b, x, wi, dz, Bn=B, Hi=h, Wi=w, Ho=h, Wo=w, Cout=ac, taps=[(0, 0)], bnr=bnr,
                             fbn=dict(mode=2, out=dy, bname=bname, coef=coef, affine=True, narrow=True) if bnr else None, **kw)
            if dsc.fbn:
                self.grad_ready[bname + ".weight"] = self.grad_ready[bname + ".bias"] = len(b)
            else:
                self._bnb(b, dz=dz, y=y, bname=bname, dy=dy, M=M, Cn=ac, mask_mode=2, reduce_done_nblk=self._fused_nblk(dsc, bnr))
        conv_bnb(dl, (wt, tile, npad), daf, self.yf, "assp.bnf", dyf, Cin=ck, alg_k=Q)
        # ---- convf: 5-tap wgrad over the planes; input gradient plane by plane
        b.wait(b.record(0), 1)
        self._wgrad(b, dyf, self.acat, None, Bn=1, Hi=5 * B * h, Wi=w, Cin=ac, Ho=B * h, Wo=w, Cd=ac, ldd=ac, taps=self.ptaps,
                    stride=1, parts=[("assp.convf.weight", 0, 0, ac, 1, 5 * ac)])
        wtf = self._plan_pack_t("assp.convf", ac, 5 * ac, 1)       # [5*ac (rows = concat channel)][ck = ac]
        assert wtf[3] == ac and wtf[2] >= 5 * ac
        dacat = self.new(5, M, ac)
        dybs = [self.new(M, ac) for _ in ASSP_BRANCHES]
        for t, (i, k, dil) in enumerate(ASSP_BRANCHES):
            conv_bnb(dyf, (wtf[0][t * ac:(t + 1) * ac], min(wtf[1], ac), ac), dacat[t], self.ycat[t], f"assp.bn{i}", dybs[t], Cin=ac)
        # ---- branches
        dfeat = None
        for t, (i, k, dil) in enumerate(ASSP_BRANCHES):
            dyb = dybs[t]
            b.wait(b.record(0), 1)
            taps = ops.conv_taps(3, 3, dil, dil) if k == 3 else [(0, 0)]
            self._wgrad(b, dyb, self.feat, None, Bn=B, Hi=h, Wi=w, Cin=cin, Ho=h, Wo=w, Cd=ac, ldd=ac, taps=taps, stride=1,
                        parts=[(f"assp.conv{i}.weight", 0, 0, ac, k * k, cin)])
            wtb = self._plan_pack_t(f"assp.conv{i}", ac, cin, k)
            assert wtb[3] == ac
            nxt = self.new(M, cin)
            self._conv(b, dyb, wtb[:3], nxt, Bn=B, Hi=h, Wi=w, Cin=ac, Ho=h, Wo=w, Cout=cin, taps=[(-a, -c) for (a, c) in taps],
                       res=dfeat)
            dfeat = nxt
        dz = dfeat
        # ---- Bottlenecks in reverse
        for bi in range(len(self.block_io) - 1, -1, -1):
            rec = self.block_io[bi]
            name, Mo, Mi, p, inpl = rec["name"], rec["Mo"], rec["Mi"], rec["planes"], rec["inpl"]
            Ho, Wo, Hi, Wi, stride, down = rec["Ho"], rec["Wo"], rec["Hi"], rec["Wi"], rec["stride"], rec["down"]
            c4 = 4 * p
            blk_start = len(b)
            dy3 = self.new(Mo, c4)
            dyd = self.new(Mo, c4) if down else None
            self._bnb(b, dz=dz, z=rec["zbits"], y=rec["y3"], bname=name + ".bn3", dy=dy3, M=Mo, Cn=c4, mask_mode=3, y2=rec.get("yd"),
                      bname2=name + ".downsample.1" if down else None, dy2=dyd)
            # weight gradients: one grouped launch at the end of the block when its convs see the same pixels (stride 1), else one launch
            # per conv behind its own event
            grouped = self._wgrad_grouped() and Mi == Mo
            wjobs = []

            def wgrad(first, **job):
                if grouped:
                    wjobs.append(job)
                    return
                if first:
                    b.wait(b.record(0), 1)
                self._wgrad(b, job["dy"], job["x"], None, **{k: v for k, v in job.items() if k not in ("dy", "x")})
            wgrad(True, dy=dy3, x=rec["a2"], Bn=B, Hi=Ho, Wi=Wo, Cin=p, Ho=Ho, Wo=Wo, Cd=c4, ldd=c4, taps=[(0, 0)], stride=1,
                  parts=[(name + ".conv3.weight", 0, 0, c4, 1, p)])
            if down:
                wgrad(False, dy=dyd, x=rec["x"], Bn=B, Hi=Hi, Wi=Wi, Cin=inpl, Ho=Ho, Wo=Wo, Cd=c4, ldd=c4, taps=[(0, 0)],
                      stride=stride, parts=[(name + ".downsample.0.weight", 0, 0, c4, 1, inpl)])
            wt3 = self._plan_pack_t(name + ".conv3", c4, p, 1)
            da2 = self.new(Mo, p)
            bnr = self._bnr(name + ".bn2", rec["y2"], 2)       # first pass of bn2's backward inside the GEMM that produces da2
            dy2 = self.new(Mo, p)
            coef = self.buf("bnb.coef", 3 * 2048, dtype=torch.float32)
            # on the small maps (the launch is one co-resident round of the chip) the whole BatchNorm backward rides in the dgrad launch:
            # simt_fbn_desc mode 2, d gamma / d beta written by its owner workgroups (engine.TrunkPlan._conv)
            dsc = self._conv(b, dy3, wt3[:3], da2, Bn=B, Hi=Ho, Wi=Wo, Cin=wt3[3], Ho=Ho, Wo=Wo, Cout=p, taps=[(0, 0)], bnr=bnr,
                             fbn=dict(mode=2, out=dy2, bname=name + ".bn2", coef=coef, affine=True, narrow=True) if bnr else None)
            if dsc.fbn:
                self.grad_ready[name + ".bn2.weight"] = self.grad_ready[name + ".bn2.bias"] = len(b)
            else:
                self._bnb(b, dz=da2, y=rec["y2"], bname=name + ".bn2", dy=dy2, M=Mo, Cn=p, mask_mode=2,
                          reduce_done_nblk=self._fused_nblk(dsc, bnr))
            # conv2: 3x3, stride s.  dgrad of a strided conv = stride-1 correlation of the zero-inserted dY with mirrored taps
            t3 = ops.conv_taps(3, 3, 1, 1)
            wgrad(True, dy=dy2, x=rec["a1"], Bn=B, Hi=Hi, Wi=Wi, Cin=p, Ho=Ho, Wo=Wo, Cd=p, ldd=p, taps=t3, stride=stride,
                  parts=[(name + ".conv2.weight", 0, 0, p, 9, p)])
            wt2 = self._plan_pack_t(name + ".conv2", p, p, 3)
            assert wt2[3] == p and wt3[3] == c4
            src = dy2
            if stride != 1:
                src = self.new(Mi, p)
                b.add("simt_scatter_stride", dy2.data_ptr(), src.data_ptr(), B, Hi, Wi, p, Ho, Wo, stride, ops.dt_code(dt))
            da1 = self.new(Mi, p)
            bnr = self._bnr(name + ".bn1", rec["y1"], 2)
            dy1 = self.new(Mi, p)
            dsc = self._conv(b, src, wt2[:3], da1, Bn=B, Hi=Hi, Wi=Wi, Cin=p, Ho=Hi, Wo=Wi, Cout=p, taps=[(-a, -c) for (a, c) in t3],
                             alg_flops=2.0 * Mo * p * 9 * p, bnr=bnr,
                             fbn=dict(mode=2, out=dy1, bname=name + ".bn1", coef=coef, affine=True, narrow=True) if bnr else None)
            if dsc.fbn:
                self.grad_ready[name + ".bn1.weight"] = self.grad_ready[name + ".bn1.bias"] = len(b)
            else:
                self._bnb(b, dz=da1, y=rec["y1"], bname=name + ".bn1", dy=dy1, M=Mi, Cn=p, mask_mode=2,
                          reduce_done_nblk=self._fused_nblk(dsc, bnr))
            wgrad(True, dy=dy1, x=rec["x"], Bn=B, Hi=Hi, Wi=Wi, Cin=inpl, Ho=Hi, Wo=Wi, Cd=p, ldd=p, taps=[(0, 0)], stride=1,
                  parts=[(name + ".conv1.weight", 0, 0, p, 1, inpl)])
            if grouped:
                b.wait(b.record(0), 1)
                self._wgrad_group(b, wjobs)
            # shortcut gradient at the block input's resolution
            if down:
                wtd = self._plan_pack_t(name + ".downsample.0", c4, inpl, 1)
                dxd = self.new(Mo, inpl)
                self._conv(b, dyd, wtd[:3], dxd, Bn=B, Hi=Ho, Wi=Wo, Cin=c4, Ho=Ho, Wo=Wo, Cout=inpl, taps=[(0, 0)])
                res = dxd
                if stride != 1:
                    res = self.new(Mi, inpl)
                    b.add("simt_scatter_stride", dxd.data_ptr(), res.data_ptr(), B, Hi, Wi, inpl, Ho, Wo, stride, ops.dt_code(dt))
                rbits = None
            else:           # identity shortcut: dz under the block-output ReLU bit mask, added by the conv epilogue
                res, rbits = dz, rec["zbits"]
            wt1 = self._plan_pack_t(name + ".conv1", p, inpl, 1)
            assert wt1[3] == p
            dx = self.new(Mi, inpl)
            self._conv(b, dy1, wt1[:3], dx, Bn=B, Hi=Hi, Wi=Wi, Cin=p, Ho=Hi, Wo=Wi, Cout=inpl, taps=[(0, 0)], res=res, res_bits=rbits)
            self.bwd_marks[name] = (blk_start, len(b), dz, dx)
            rec.update(g_dy3=dy3, g_da2=da2, g_dy2=dy2, g_da1=da1, g_dy1=dy1, g_dx=dx, g_dz=dz)
            dz = dx
        # ---- stem
        H0, W0, Hp, Wp, wd = self.H0, self.W0, self.Hp, self.Wp, self.width
        M0 = B * H0 * W0
        da0 = self.new(M0, wd)
        b.add("simt_maxpool_bwd", dz.data_ptr(), self.saved["stem.idx"].data_ptr(), da0.data_ptr(), B, H0, W0, wd, Hp, Wp, ops.dt_code(dt))
        dy0 = self.new(M0, wd)
        self._bnb(b, dz=da0, y=self.saved["stem.y"], bname=R50 + "bn1", dy=dy0, M=M0, Cn=wd, mask_mode=2)
        b.wait(b.record(0), 1)
        self._wgrad(b, dy0, self.saved["stem.A"], None, Bn=1, Hi=1, Wi=M0, Cin=192, Ho=1, Wo=M0, Cd=wd, ldd=wd, taps=[(0, 0)], stride=1,
                    parts=[(R50 + "conv1.weight", 0, 0, wd, 1, 147)])
        b.wait(b.record(1), 0)

    # ------------------------------------------------------------------ run
    def forward(self, x_nchw=None):
        """Returns the full-resolution logits [B, nc+openc, H, W] fp32 (NCHW, what DeepLabv3.forward returns)."""
        if x_nchw is not None:
            self.x_in.copy_(x_nchw)
        self.fwd_list.run()
        return self.out_full

    def backward(self, dout=None, hook=None):
        if dout is not None:
            self.dout_full.copy_(dout)
        return super().backward(hook)
