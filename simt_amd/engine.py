"""Static execution plans for the DeepLab-v2 ResNet trunk + ASPP heads on gfx950.

A plan is built once for a fixed (B, H, W, dtype): every activation / gradient / workspace buffer is allocated up
front in HBM (NHWC), every kernel launch is pre-described (ctypes descriptors of include/simt_hip.h), and a step is a
replay of that launch list on the current HIP stream -- no allocation, no host sync, capturable in a hipGraph.

Mirrors (reference, read-only): model/deeplab_multi.py:57-101 Bottleneck, :104-119 Classifier_Module (two live
branches), :122-192 ResNetMulti; model/deeplab.py:101-177 (single 4-branch head).  BatchNorm runs in train mode with
frozen affine in the trainable net and is folded into the conv weights in eval plans.

PyTorch is used for device memory and streams only; all arithmetic happens in libsimt_hip.so.
"""
import ctypes as C

import os

import torch

from . import _lib as L
from . import ops

LAYERS = (3, 4, 23, 3)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def block_specs(layers=LAYERS):
    """[(name, inplanes, planes, stride, dilation, has_downsample)] (model/deeplab_multi.py:152-167)."""
    out = []
    inpl = 64
    for li, (planes, nblk, stride, dil) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 1, 1), (1, 1, 2, 4)), 1):
        for b in range(nblk):
            out.append((f"layer{li}.{b}", inpl, planes, stride if b == 0 else 1, dil, b == 0))
            inpl = planes * 4
    return out


def trunk_geometry(H, W):
    """Spatial sizes after the stem conv (7x7 s2 p3), the ceil-mode max-pool (3,2,1) and layer2's stride 2."""
    def pool_out(n):
        o = -(-(n + 2 - 3) // 2) + 1
        if (o - 1) * 2 >= n + 1:
            o -= 1
        return o
    H0, W0 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    Hp, Wp = pool_out(H0), pool_out(W0)
    H2, W2 = (Hp - 1) // 2 + 1, (Wp - 1) // 2 + 1
    return (H0, W0), (Hp, Wp), (H2, W2)


class HeadCfg:
    """An ASPP classifier on a trunk feature: groups = [(prefix, cout)] concatenated on the channel axis."""

    def __init__(self, name, feat_layer, cin, groups, dilations):
        self.name, self.feat_layer, self.cin, self.groups, self.dilations = name, feat_layer, cin, groups, dilations
        self.Q = sum(c for _, c in groups)


def multi_heads(num_classes, open_classes, openset):
    """model/deeplab_multi.py:138-142,181-190: layer5(+layer5_1) on layer3, layer6(+layer6_1) on layer4, dil (6, 12)."""
    g1 = [("layer5", num_classes)] + ([("layer5_1", open_classes)] if openset else [])
    g2 = [("layer6", num_classes)] + ([("layer6_1", open_classes)] if openset else [])
    return [HeadCfg("x1", 3, 1024, g1, (6, 12)), HeadCfg("x2", 4, 2048, g2, (6, 12))]


def single_head(num_classes):
    """model/deeplab.py:112-116,139: one 4-branch head on layer4."""
    return [HeadCfg("x", 4, 2048, [("layer5", num_classes)], (6, 12, 18, 24))]


class _Launch:
    __slots__ = ("fn", "args", "keep", "tag", "flops", "bytes", "shape", "stream")

    def __init__(self, fn, args, keep, tag=None, flops=0.0, nbytes=0.0, shape=None, stream=0):
        self.fn, self.args, self.keep, self.tag, self.flops, self.bytes, self.shape = fn, args, keep, tag, flops, nbytes, shape
        self.stream = stream          # 0 = the caller's current stream, 1 = the plan's side stream


_SIDE_STREAMS = {}


def side_stream(device=None):
    """One extra HIP stream per device: weight-gradient GEMMs (off the critical path of backward) and the frozen
    model's forward run there, so their MFMA-bound workgroups share the CUs with the HBM-bound BatchNorm passes of the
    main stream instead of queueing behind them."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if os.environ.get("SIMT_SINGLE_STREAM") == "1":     # profiling aid: serial schedule, per-kernel durations without CU sharing
        return torch.cuda.current_stream(dev)
    if dev not in _SIDE_STREAMS:
        prio = int(os.environ.get("SIMT_SIDE_PRIORITY", "0"))
        _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev, priority=prio)
    return _SIDE_STREAMS[dev]


_RESERVED = set()


def reserve_streams(device=None):
    """Give the plan's two streams their hardware queues NOW.  HIP multiplexes every stream of a process onto 4 hardware queues, handed out in
    order of FIRST USE; two streams that share a queue run their kernels strictly one after the other.  Measured (round 6, profiles/
    r06_dp_emulation.txt): a data-parallel job that created its process group first -- RCCL and ProcessGroupNCCL make streams of their own --
    found the plan's side stream (weight gradients, frozen forward) on the MAIN stream's queue: 26.18 ms per step instead of 24.06, the whole
    two-stream overlap gone.  Call this before torch.distributed.init_process_group / before creating other streams (bench.py and the tools do);
    TrunkPlan calls it too, which is early enough in a single-GPU process.  Idempotent."""
    if not torch.cuda.is_available() or os.environ.get("SIMT_NO_RESERVE_STREAMS") == "1":      # (the switch: A/B only)
        return
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev in _RESERVED:
        return
    _RESERVED.add(dev)
    with torch.cuda.device(dev):
        main, side = torch.cuda.current_stream(dev), side_stream(dev)
        t = torch.zeros(64, device=torch.device("cuda", dev))
        t.add_(1)                                  # a kernel on the main stream ...
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            t.add_(1)                              # ... and one on the side stream: both queues exist from here on
        torch.cuda.synchronize(dev)


class DeviceEvent:
    """An event that orders the two HIP streams of a plan on ONE device: recorded with a device-scope release (simt_event_create: no
    system-scope cache write-back behind every record, which costs the recording queue ~6.5 us and a cross-stream wait ~12 us).  Same
    two methods the launch lists use of torch.cuda.Event."""
    __slots__ = ("h",)

    def __init__(self):
        h = C.c_void_p()
        # SIMT_EVENT_SCOPE: 0 (default) hipEventReleaseToDevice, the documented device-scope release; 2 = no release at the marker
        # (hipEventDisableSystemFence: round 5's form, opt-in); 1 = system scope
        L.call("simt_event_create", C.byref(h), int(os.environ.get("SIMT_EVENT_SCOPE", "0")))
        self.h = h

    def record(self, stream):
        L.call("simt_event_record", self.h, stream.cuda_stream)

    def wait(self, stream):
        L.call("simt_stream_wait_event", stream.cuda_stream, self.h)

    def __del__(self):
        try:
            if self.h:
                L.load().simt_event_destroy(self.h)
        except Exception:
            pass


def _new_event():
    """SIMT_LIGHT_EVENTS=0: torch.cuda.Event (system-scope release), the A/B switch."""
    if os.environ.get("SIMT_LIGHT_EVENTS", "1") != "0" and torch.cuda.is_available():
        return DeviceEvent()
    return torch.cuda.Event()


def wait_event(stream, ev):
    """stream waits for ev (a torch.cuda.Event or a DeviceEvent); for callers outside this module (dp.BucketReducer)."""
    _wait_event(stream, ev)


def _wait_event(stream, ev):
    if isinstance(ev, DeviceEvent):
        ev.wait(stream)
    else:
        stream.wait_event(ev)


class LaunchList:
    """A recorded sequence of C-ABI calls; run() replays it on the current stream."""

    def __init__(self):
        self.items = []
        self.graph = None

    def capture(self, warm=True):
        """Capture the whole list (both streams, with its fork/join events) into one hipGraph: run() then costs one graph
        launch instead of len(self) host calls -- what the launch-bound plans (DeepLabv3 at 512x1024: ~450 launches of
        ~15 us) need.  Every buffer and descriptor is preallocated, so the captured kernel arguments stay valid; the list
        is replayed once eagerly first so that lazily created workspaces exist before capture."""
        self.graph = None
        if warm:
            self.run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run()
        self.graph = g
        return g

    def add(self, name, *args, keep=None, tag=None, flops=0.0, nbytes=0.0, shape=None, stream=0):
        fn = getattr(L.load(), name)
        self.items.append(_Launch(fn, args, keep, tag or name, flops, nbytes, shape, stream))

    def record(self, stream):
        """Event recorded on `stream` at this point of the replay; returns it (pass to wait())."""
        ev = _new_event()
        self.items.append(_Launch(None, (ev,), None, "record", stream=stream))
        return ev

    def wait(self, ev, stream):
        self.items.append(_Launch(None, (ev,), None, "wait", stream=stream))

    def add_desc(self, name, desc, **kw):
        self.add(name, C.byref(desc), keep=desc, **kw)

    def run_timed(self, acc, by_shape=None):
        """Replay with a HIP event pair around every launch (on the stream the kernels are launched on) and add
        (milliseconds, algorithmic flops, algorithmic bytes, count) per tag into `acc`.  Measurement only."""
        stream = torch.cuda.current_stream()
        st = stream.cuda_stream
        evs = []
        for it in self.items:
            if it.fn is None:       # measured serially on one stream: stream assignments and events are not needed
                continue
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = it.fn(*it.args, st)
            e1.record(stream)
            if rc != 0:
                L.check(rc)
            evs.append((it, e0, e1))
        torch.cuda.synchronize()
        for it, e0, e1 in evs:
            a = acc.setdefault(it.tag, [0.0, 0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1)
            a[1] += it.flops
            a[2] += it.bytes
            a[3] += 1
            if by_shape is not None and it.shape is not None:
                b = by_shape.setdefault((it.tag, it.shape), [0.0, 0.0, 0.0, 0])
                b[0] += e0.elapsed_time(e1)
                b[1] += it.flops
                b[2] += it.bytes
                b[3] += 1

    def run(self, single_stream=False):
        if self.graph is not None and not single_stream:
            self.graph.replay()
            return
        main = torch.cuda.current_stream()
        if single_stream or not any(it.stream for it in self.items):
            st = main.cuda_stream
            for it in self.items:
                if it.fn is None:
                    continue
                rc = it.fn(*it.args, st)
                if rc != 0:
                    L.check(rc)
            return
        streams = (main, side_stream(main.device))
        handles = (main.cuda_stream, streams[1].cuda_stream)
        for it in self.items:
            if it.fn is None:
                if it.tag == "record":
                    it.args[0].record(streams[it.stream])
                else:
                    _wait_event(streams[it.stream], it.args[0])
                continue
            rc = it.fn(*it.args, handles[it.stream])
            if rc != 0:
                L.check(rc)

    def __len__(self):
        return len(self.items)

    def coalesce_packs(self, device):
        """Replace every simt_pack_weight entry by ONE simt_pack_weight_multi launch (placed where the last of them
        was: BN-fold launches whose scales the packs read stay in front).  Returns the device tables (kept alive)."""
        import numpy as np
        lib = L.load()
        packs = [it for it in self.items if it.fn is lib.simt_pack_weight]
        if len(packs) < 2:
            return None
        job_dt = np.dtype([("w", "<u8"), ("dst", "<u8"), ("cscale", "<u8"), ("ldk", "<i8"), ("total", "<i8"), ("Cout", "<i4"),
                           ("Cin", "<i4"), ("RS", "<i4"), ("row_off", "<i4"), ("tap_off", "<i4"), ("Ck", "<i4"), ("mode", "<i4"),
                           ("dtype", "<i4")])
        assert job_dt.itemsize == 72
        recs, chunks, chunk = [], [], 1024
        for ji, it in enumerate(packs):
            w, dst, cout, cin, rs, row_off, tap_off, ldk, ck, mode, cscale, dtype = it.args
            total = cout * cin * rs
            recs.append((w, dst, cscale or 0, ldk, total, cout, cin, rs, row_off, tap_off, ck, mode, dtype))
            chunks += [(ji, ti) for ti in range(((cout + 31) // 32) * ((cin + 31) // 32))]      # 32x32 (cout, cin) tiles
        jobs = torch.from_numpy(np.array(recs, dtype=job_dt).view(np.uint8).copy()).to(device)
        ch = torch.tensor(chunks, dtype=torch.int32).to(device)
        last = max(i for i, it in enumerate(self.items) if it.fn is lib.simt_pack_weight)
        new_items = []
        for i, it in enumerate(self.items):
            if it.fn is lib.simt_pack_weight:
                if i == last:
                    new_items.append(_Launch(lib.simt_pack_weight_multi, (jobs.data_ptr(), ch.data_ptr(), len(chunks), chunk),
                                             (jobs, ch), "simt_pack_weight_multi"))
                continue
            new_items.append(it)
        self.items = new_items
        return jobs, ch


def layout_flat_grads(plan, order):
    """ONE flat fp32 gradient buffer in backward-completion order (DP buckets are contiguous and become ready in order).  Every
    entry starts on a 16-byte boundary (its span is padded to a multiple of 4 floats; the pad stays zero): the weight-gradient reduce
    kernels store `float4` (csrc/conv_wgrad.hip wgrad_reduce4*), and e.g. DeepLabv3's 19 + 6 classifier biases at the head of the
    buffer would otherwise leave every later gradient 4-byte aligned only.  grad_offsets[name] = (offset, padded span)."""
    spans = [(plan.p[n].numel() + 3) // 4 * 4 for n in order]
    plan.flat_grad = torch.zeros(sum(spans), device=plan.dev, dtype=torch.float32)
    plan.grads, plan.grad_order, plan.grad_offsets = {}, list(order), {}
    off = 0
    for n, span in zip(order, spans):
        plan.grads[n] = plan.flat_grad[off:off + plan.p[n].numel()].view(plan.p[n].shape)
        plan.grad_offsets[n] = (off, span)
        assert plan.grads[n].data_ptr() % 16 == 0
        off += span


class TrunkPlan:
    """Forward (train or eval) and backward of ResNetMulti for one fixed input shape.

    params: {state_dict key: fp32 CUDA tensor}  (the nn.Module's own storage: optimiser updates are seen by
            repack()).  Buffers (running_mean/var) are updated in place in train mode.
    grads : {key: fp32 CUDA tensor} written by backward() (views of one flat buffer, reverse-topological order so
            that DP buckets become ready in order).
    """

    def __init__(self, params, B, H, W, heads, *, dtype=torch.bfloat16, train=True, layers=LAYERS, device=None,
                 need_input_grads=True, grad_names=None, grads_from_layer=0, stem_from=None, data_parallel=False):
        self.p = params
        self.B, self.H, self.W = B, H, W
        self.heads = heads
        self.dtype = dtype
        self.train = train
        self.layers = layers
        # 0: every gradient autograd would compute (the reference's behaviour).  3: stop the backward at the input of layer3 --
        # the SimT stage never applies the gradients of conv1 / layer1 / layer2 (optim_parameters lists layer3, layer4 and the
        # heads only, model/deeplab_multi.py:194-237), so the trajectory is identical; see Hyper.skip_unapplied_grads.
        self.grads_from_layer = grads_from_layer
        # stem_from: another plan of the same geometry / dtype fed with the SAME image (the frozen model of the SimT stage): its
        # im2col matrix is reused instead of being rebuilt; the caller orders the streams (step.py)
        self.stem_from = stem_from
        self.dev = device or next(iter(params.values())).device
        if torch.device(self.dev).type == "cuda":
            reserve_streams(self.dev)
        self.esz = 2 if dtype == torch.bfloat16 else 4
        self.kq = 128 // self.esz            # channel quantum of the K dimension (one 128-B stage)
        (self.H0, self.W0), (self.Hp, self.Wp), (self.H2, self.W2) = trunk_geometry(H, W)
        self.blocks = block_specs(layers)
        self._bufs = {}
        self._keep = []
        self.packed = {}      # conv name -> fprop operand
        self.packed_t = {}    # conv name -> dgrad operand
        self.fold = {}        # bn name -> (scale, shift) for eval plans
        self.bn = {}          # bn name -> dict(mean, rstd, scale, shift, part, nblk, count)
        # Train-mode BatchNorm fused into the producing conv launch (simt_fbn_desc; round 4): the launch's workgroups exchange their tile
        # sums through polled granules, so all of them must be resident at once.  Measured on the production step (profiles/
        # r04_bn_fusion.txt): the BACKWARD form (dgrad + BatchNorm backward in one launch) is worth -0.2 ms; the FORWARD form is +0.75 ms
        # slower in the step although it is 5 us faster per launch alone -- its waiting workgroups hold CUs the frozen forward on the side
        # stream wants.  Default 3 = backward only.  A waiting launch needs ALL its workgroups resident (236 of 256 CUs, 156 KB of LDS each); a
        # second PROCESS on the same GPU can starve it outright (bench.py forces SIMT_BN_GRID=0 there).  Rounds 4-5 also switched it off under
        # data parallelism (255-workgroup launches beside a collective's persistent kernels); see below.  A launch whose polling times out (~2 s) no longer traps: it sets the plan's sticky error word `fbn_err` and ends;
        # fbn_error() reports it, the trainers' losses() raise, and the optimiser kernels skip their update while it is set (skip_if).
        # Round 6: data-parallel plans default to 3 as well.  The fused launches are 236 workgroups (160-row tiles): with <= 20 CUs held by the
        # collective they ARE all resident, with more they wait for the collective's kernels to end (finite: those never wait for a compute kernel
        # of this process) -- no circular wait, and the time-out path no longer traps.  Measured over a one-rank RCCL group (profiles/
        # r06_dp_emulation.txt): 24.83 ms two-pass -> 24.31 ms fused.  SIMT_BN_GRID=0 remains the switch; bench.py forces it when several ranks
        # share one GPU (two waiting launches of different processes can starve each other).
        g = os.environ.get("SIMT_BN_GRID")           # 0 off, 1 forward + backward, 2 forward only, 3 backward only
        if g is None:
            g = "3"
        self.data_parallel = bool(data_parallel)
        # CU budget of the conv tile lists (simt_conv_desc.cu_budget; round 6).  The wide convs are ONE workgroup per CU (156 KB of LDS); their
        # default plan at M = 37 636 is 236 tiles of 160 rows (pick_rows: the largest one-round tile), which leaves 20 CUs to whatever runs beside
        # them -- the side stream's weight gradients, or a collective's persistent kernels.  A data-parallel plan states the CUs it may use as
        # 256 - NCCL_MAX_NCHANNELS (16 channels assumed when the variable is unset -- bench.py and the training tools set that default before the
        # process group exists; at most 20 are given up): any budget >= 236 leaves the plan as it is, so the budget costs nothing; a smaller one
        # re-plans the tile lists (profiles/r06_dp_emulation.txt).  SIMT_CU_BUDGET=n sets it explicitly (0: the whole device).
        cb = os.environ.get("SIMT_CU_BUDGET")
        if os.environ.get("SIMT_PICK_ROWS_FIRST") == "1":
            self.cu_budget = -1                        # A/B only: rounds 1-5's tile plan (255 tiles of 148 rows at M = 37 636)
        elif cb is not None:
            self.cu_budget = int(cb)
        elif data_parallel:
            self.cu_budget = 256 - max(0, min(20, int(os.environ.get("NCCL_MAX_NCHANNELS", "16"))))
        else:
            self.cu_budget = 0
        self._fbn_on = train and dtype == torch.bfloat16 and g != "0"
        self._fbn_dirs = {"1": (1, 2), "2": (1,), "3": (2,)}.get(g, ())
        self.fbn_err = None                          # one int64 device word shared by every fused launch of this plan (allocated on first use)
        self.fbn_launches = 0                        # fused BatchNorm launches in the plan's lists (0: fbn_error() never synchronises)
        self.pack_list = LaunchList()
        self.fwd_list = LaunchList()
        self.bwd_list = LaunchList()
        self.out = {}         # head name -> fp32 logits [B, h, w, ldp]
        self.ldp = {}
        self._build_forward()
        if train:
            self._alloc_grads(grad_names)
            self._build_backward()
        self._pack_items_raw = list(self.pack_list.items)          # before coalescing: pack_subset() filters these
        self._pack_tables = self.pack_list.coalesce_packs(self.dev)
        self.repack()

    def pack_subset(self, names):
        """A launch list that refreshes only the packed operands of the parameters `names` (plus every non-weight entry of the pack
        list: bias sums, BN folds) -- for a stage whose optimiser updates part of the model (SimT: layer3, layer4, heads)."""
        ptrs = {self.p[n].data_ptr() for n in names}
        lib = L.load()
        sub = LaunchList()
        sub.items = [it for it in self._pack_items_raw if it.fn is not lib.simt_pack_weight or it.args[0] in ptrs]
        sub._tables = sub.coalesce_packs(self.dev)
        return sub

    def fbn_error(self):
        """True if a fused-BatchNorm launch of this plan gave up polling (its workgroups were not all resident within ~2 s: another
        process's waiting launch or a collective's persistent kernels held CUs).  Synchronises the device; the word is sticky."""
        return self.fbn_err is not None and int(self.fbn_err.item()) != 0

    def raise_on_fbn_error(self):
        if self.fbn_error():
            raise RuntimeError("a fused BatchNorm launch (SIMT_BN_GRID) timed out waiting for its workgroups to become co-resident: its output and every "
                               "later activation / gradient of that step are undefined.  No optimiser launch (SGD, Adam on NTM1 / NTM2, the W inner loop) "
                               "has changed anything since the word was set, and the launch that gave up updated no BatchNorm statistics: the state_dict "
                               "holds the last good state (W one inner loop ahead if the time-out hit in the iteration's own forward / backward).  Rebuild the "
                               "trainer with SIMT_BN_GRID=0 (two-pass BatchNorm) -- required when several processes share a GPU, the default under data "
                               "parallelism")

    # ------------------------------------------------------------------ buffers
    def buf(self, role, *shape, dtype=None, zero=False):
        dtype = dtype or self.dtype
        key = (role,) + tuple(shape) + (dtype,)
        t = self._bufs.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.dev, dtype=dtype)
            self._bufs[key] = t
        return t

    def new(self, *shape, dtype=None, zero=False):
        dtype = dtype or self.dtype
        t = (torch.zeros if zero else torch.empty)(shape, device=self.dev, dtype=dtype)
        self._keep.append(t)
        return t

    def bytes_allocated(self):
        return sum(t.numel() * t.element_size() for t in list(self._bufs.values()) + self._keep)

    # ------------------------------------------------------------------ weight packing
    def _plan_pack(self, cname, cout, cin, k, *, scale_bn=None, K_cin=None):
        """fprop operand [Npad][ntaps*Cin_k]; with scale_bn (eval) the BN scale is folded per output channel."""
        tile = ops.pick_tile_n(cout, self.dtype)
        npad = ops.round_up(cout, tile)
        cin_k = K_cin or cin
        wp = self.new(npad, cin_k if cname == "conv1" else k * k * cin_k, zero=True)
        self.packed[cname] = (wp, tile, npad)
        w = self.p[cname + ".weight"]
        cs = self.fold[scale_bn][0] if scale_bn else None
        if cname == "conv1":
            # stem: K = ci*49 + r*7 + s (OIHW flattening), padded to cin_k; expressed as Cin=147, RS=1
            self.pack_list.add("simt_pack_weight", w.data_ptr(), wp.data_ptr(), cout, 147, 1, 0, 0, cin_k, 0, 0,
                               cs.data_ptr() if cs is not None else None, ops.dt_code(self.dtype))
        else:
            self.pack_list.add("simt_pack_weight", w.data_ptr(), wp.data_ptr(), cout, cin, k * k, 0, 0, k * k * cin, 0, 0,
                               cs.data_ptr() if cs is not None else None, ops.dt_code(self.dtype))
        return wp, tile, npad

    def _plan_pack_t(self, cname, cout, cin, k):
        """dgrad operand [Cin_pad][ntaps*Ck], Ck = cout rounded to the K quantum."""
        ck = ops.round_up(cout, self.kq)
        tile = ops.pick_tile_n(cin, self.dtype)
        npad = ops.round_up(cin, tile)
        wt = self.new(npad, k * k * ck, zero=True)
        self.packed_t[cname] = (wt, tile, npad, ck)
        w = self.p[cname + ".weight"]
        self.pack_list.add("simt_pack_weight", w.data_ptr(), wt.data_ptr(), cout, cin, k * k, 0, 0, k * k * ck, ck, 1, None,
                           ops.dt_code(self.dtype))
        return wt, tile, npad, ck

    def _frag_twin(self, wp, npad):
        """Fragment-ordered copy of the packed operand `wp` (simt_conv_desc.w_frag), kept fresh by twin pack jobs: every
        simt_pack_weight entry that writes `wp` gets a second entry with the same source and SIMT_PACK_FRAG in `mode`."""
        if not hasattr(self, "_frag"):
            self._frag = {}
        key = wp.data_ptr()
        if key not in self._frag:
            lib = L.load()
            wf = self.new(*wp.shape, zero=True)
            jobs = [it for it in self.pack_list.items if it.fn is lib.simt_pack_weight and it.args[1] == key]
            assert jobs, "a conv asked for fragment-ordered weights before its pack job was planned"
            for it in jobs:
                a = list(it.args)
                a[1] = wf.data_ptr()
                a[9] = a[9] | ops.PACK_FRAG(npad)
                self.pack_list.add("simt_pack_weight", *a)
            self._frag[key] = wf
        return self._frag[key]

    def repack(self):
        """Refresh every packed operand from the fp32 master weights (after an optimiser step / load_state_dict)."""
        self.pack_list.run()

    # ------------------------------------------------------------------ forward construction
    def _conv(self, lst, x, wp_info, y, *, Bn, Hi, Wi, Cin, Ho, Wo, Cout, taps, stride=1, bias=None, res=None, stats=None,
              relu=False, ldy=None, Nstore=None, alg_k=None, alg_flops=None, mask=None, res_bits=None, bnr=None, note="", fbn=None,
              inbn=None):
        """alg_k: algorithmic reduction length per output element (defaults to ntaps*Cin; the K-padded head dgrad and
        the stem pass their true value) -> algorithmic FLOPs = 2 * M * Cout * alg_k.
        fbn: ask for the train-mode BatchNorm behind this conv to be fused into the launch (simt_fbn_desc; dict(mode, out, bname[,
        coef, affine])).  Granted only where the launch is one co-resident round of the chip (simt_conv_fbn_ok): the returned descriptor then has
        `.fbn` set and the caller must NOT add the separate BatchNorm launches."""
        wp, tile, npad = wp_info
        # Few pixels (DeepLabv3's stride-16 maps: M = 8 192 at 512 x 1024): 128-row x 256-column tiles are 64 workgroups on 256 CUs, each
        # streaming the whole weight panel through its LDS.  Narrower column tiles fill the chip and cut the staged bytes per workgroup
        # (same packed operand: Npad is a multiple of every tile width).
        if x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and os.environ.get("SIMT_SMALL_M_TILES", "1") != "0":
            while tile > 64 and ((Bn * Ho * Wo + 127) // 128) * (npad // tile) < 256 and npad % (tile // 2) == 0:
                tile //= 2
        d = ops.make_conv_desc(x, wp, y, B=Bn, H=Hi, W=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cout=Cout, taps=taps, stride=stride,
                               bias=bias, res=res, stats=stats, relu=relu, Npad=npad, tile_n=tile, ldy=ldy,
                               Nstore=Nstore, mask=mask, res_bits=res_bits, bnr=bnr)
        d.cu_budget = self.cu_budget
        if inbn is not None:
            # x is the RAW pre-BatchNorm activation; the launch normalises + ReLUs it in its operand path and writes the activation to inbn[2]
            # (simt_conv_desc.in_*; the caller asked simt_conv_inbn_ok first: _conv3_takes_bn2)
            assert L.load().simt_conv_inbn_ok(C.byref(d))
            d.in_scale, d.in_shift, d.in_out = inbn[0].data_ptr(), inbn[1].data_ptr(), inbn[2].data_ptr()
            note = note + " (input BatchNorm in the operand path)"
        M = Bn * Ho * Wo
        k = alg_k if alg_k is not None else len(taps) * Cin
        tn = {torch.bfloat16: "bf16", torch.float32: "f32"}
        bn_, tm_, nst_ = C.c_int(), C.c_int(), C.c_int()
        gen = L.load().simt_conv_variant(C.byref(d), C.byref(bn_), C.byref(tm_), C.byref(nst_))
        fb = 0
        # narrow (128- / 64-column) tiles only where the caller asks: worth +1 % on DeepLabv3's small maps (engine_v3), -0.04 ms on layer 2 of the
        # two-stream DeepLab-v2 step (profiles/r04_bn_fusion.txt)
        if (fbn is not None and self._fbn_on and fbn["mode"] in self._fbn_dirs and gen == 2 and (bn_.value == 256 or fbn.get("narrow"))
                and L.load().simt_conv_fbn_ok(C.byref(d))):
            sb = self.bn[fbn["bname"]]
            fd = L.FbnDesc()
            fd.mode, fd.ldo, fd.out = fbn["mode"], fbn["out"].shape[-1], fbn["out"].data_ptr()
            # counters + granule buffers of THIS BatchNorm and direction (zeroed once; tags make every launch's granules its own)
            fd.work = self.new(L.load().simt_conv_fbn_words(C.byref(d)), dtype=torch.int64, zero=True).data_ptr()
            if self.fbn_err is None:
                self.fbn_err = self.new(1, dtype=torch.int64, zero=True)
            fd.err = self.fbn_err.data_ptr()
            self.fbn_launches += 1
            if fd.mode == 1:
                bname = fbn["bname"]
                fd.gamma, fd.beta = self.p[bname + ".weight"].data_ptr(), self.p[bname + ".bias"].data_ptr()
                fd.running_mean, fd.running_var = self.p[bname + ".running_mean"].data_ptr(), self.p[bname + ".running_var"].data_ptr()
                fd.momentum, fd.eps = BN_MOMENTUM, BN_EPS
                fd.mean, fd.rstd, fd.scale, fd.shift = (sb[k_].data_ptr() for k_ in ("mean", "rstd", "scale", "shift"))
            else:
                fd.coef = fbn["coef"].data_ptr()
                if fbn.get("affine"):              # trainable affine (engine_v3): d gamma / d beta leave the owners of the fused launch
                    bname = fbn["bname"]
                    fd.dgamma, fd.dbeta = self.grads[bname + ".weight"].data_ptr(), self.grads[bname + ".bias"].data_ptr()
            d.fbn = C.addressof(fd)
            d._fbn_keep = fd                       # the descriptor is read at every launch
            fb = 1
        wd = 0
        if gen == 2 and ops.conv_wants_frag(d):
            # -DSIMT_ABLATION builds with SIMT_WDIRECT=1 only: weight operand from a fragment-ordered copy (csrc/experiments/conv_igemm2_abl.hip)
            d.w_frag = self._frag_twin(wp, npad).data_ptr()
            wd = 1
        epi = L.load().simt_conv_epilogue_flavour(C.byref(d)) if gen == 2 else 0
        tag = (f"conv_igemm2_kernel<{bn_.value}, {tm_.value}, {nst_.value}, {fb}, {epi}>{' [weights-direct experiment]' if wd else ''}" if gen == 2 else
               "conv1x1_stream_kernel" if gen == 4 else "conv1x1_rows_kernel" if gen == 5 else
               f"conv_igemm_kernel<{tn[x.dtype]}, {tn[y.dtype]}, {tile}>")
        nbytes = (Bn * Hi * Wi * Cin + npad * len(taps) * Cin) * x.element_size() + M * Cout * y.element_size()
        lst.add_desc("simt_conv_fprop", d, tag=tag, flops=alg_flops if alg_flops is not None else 2.0 * M * Cout * k,
                     nbytes=float(nbytes),
                     shape=f"M{M} N{Cout} K{len(taps) * Cin} taps{len(taps)} s{stride}{note}")
        return d

    def _bn_train(self, lst, bname, y, M, Cn):
        """stats partials were written by the conv epilogue into self.bn[bname]['part']."""
        s = self.bn[bname]
        lst.add("simt_bn_finalize", s["part"].data_ptr(), s["nblk"], Cn, M, self.p[bname + ".weight"].data_ptr(),
                self.p[bname + ".bias"].data_ptr(), self.p[bname + ".running_mean"].data_ptr(),
                self.p[bname + ".running_var"].data_ptr(), BN_MOMENTUM, BN_EPS, s["mean"].data_ptr(),
                s["rstd"].data_ptr(), s["scale"].data_ptr(), s["shift"].data_ptr())

    def _new_bn(self, bname, M, Cn):
        nblk = (M + 127) // 128
        s = {"part": self.new(nblk, 2, Cn, dtype=torch.float32), "nblk": nblk, "count": M}
        for k in ("mean", "rstd", "scale", "shift"):
            s[k] = self.new(Cn, dtype=torch.float32)
        self.bn[bname] = s
        return s

    def _plan_fold(self, bname, Cn):
        sc, sh = self.new(Cn, dtype=torch.float32), self.new(Cn, dtype=torch.float32)
        self.fold[bname] = (sc, sh)
        self.pack_list.add("simt_bn_fold", self.p[bname + ".weight"].data_ptr(), self.p[bname + ".bias"].data_ptr(),
                           self.p[bname + ".running_mean"].data_ptr(), self.p[bname + ".running_var"].data_ptr(), BN_EPS,
                           sc.data_ptr(), sh.data_ptr(), Cn)
        return sc, sh

    def _build_forward(self):
        B, dt = self.B, self.dtype
        f = self.fwd_list
        H0, W0, Hp, Wp = self.H0, self.W0, self.Hp, self.Wp
        M0, Mp = B * H0 * W0, B * Hp * Wp
        self.saved = {}
        # ---- stem (model/deeplab_multi.py:127-133,172-176)
        KS = 192
        # Round 6 (VERDICT r5 #1c): the bf16 plans convolve the image DIRECTLY (csrc/stem7.hip: the patch of an 8 x 32 output tile in LDS, one
        # filter row per MFMA k-step) instead of writing a 226-MB im2col matrix and running a 64-column GEMM over it per network; a frozen plan
        # built with stem_from=<trainable plan> joins that plan's launch as its second weight set (both networks see the same image); conv1's
        # weight gradient comes straight from the image too (simt_stem7_wgrad, pixels as the MFMA reduction dimension), so no im2col matrix
        # exists anywhere in the step.  87 us forward (both networks) + 57 + 7 us weight gradient against 134 (im2col) + 62 + 60 (two GEMMs)
        # + ~75 us (gradient GEMM + reduce); the step: -0.06 / -0.12 ms on two boxes, six alternating rounds each, every round below
        # (profiles/r06_direct_stem.txt).  The forward alone (weight gradient still through an im2col matrix built in the backward) was
        # neutral.  SIMT_DIRECT_STEM=0: rounds 1-5's im2col + GEMM stem; fp32 plans always use it.
        self.direct_stem = dt == torch.bfloat16 and os.environ.get("SIMT_DIRECT_STEM", "1") != "0"
        y0 = self.new(M0, 64)
        pool = self.new(Mp, 64)
        pidx = self.new(Mp, 64, dtype=torch.uint8)
        self.saved["stem.y"], self.saved["stem.pool"], self.saved["stem.idx"] = y0, pool, pidx
        if self.direct_stem:
            w7 = self.new(64 * 7 * 32)
            tiles = L.load().simt_stem7_tiles(B, H0, W0)
            if self.train:
                s = self._new_bn("bn1", M0, 64)
                if tiles > s["nblk"]:                  # (small ragged images: partial 8 x 32 tiles outnumber the 128-row blocks)
                    s["part"] = self.new(tiles, 2, 64, dtype=torch.float32)
                s["nblk"] = tiles                      # one statistics slot per 8 x 32 tile
                self.pack_list.add("simt_stem7_pack", self.p["conv1.weight"].data_ptr(), None, w7.data_ptr())
                mine = dict(w=w7, y=y0, bias=None, relu=0, stats=s["part"])
            else:
                sc, sh = self._plan_fold("bn1", 64)
                self.pack_list.add("simt_stem7_pack", self.p["conv1.weight"].data_ptr(), sc.data_ptr(), w7.data_ptr())
                mine = dict(w=w7, y=y0, bias=sh, relu=1, stats=None)

            def put(d, i, m):
                d.w[i], d.y[i], d.relu[i] = m["w"].data_ptr(), m["y"].data_ptr(), m["relu"]
                d.bias[i] = m["bias"].data_ptr() if m["bias"] is not None else None
                d.stats[i] = m["stats"].data_ptr() if m["stats"] is not None else None
            if self.stem_from is not None:
                o = self.stem_from
                assert (o.B, o.H, o.W, o.dtype) == (B, self.H, self.W, dt) and o.direct_stem and o.stem_desc.nsets == 1
                self.x_in = o.x_in
                put(o.stem_desc, 1, mine)              # second weight set of the partner's launch (its fwd_list item 0)
                o.stem_desc.nsets = 2
                o.fwd_list.items[0].flops *= 2.0
                o._stem_partner_keep = mine
            else:
                self.x_in = self.new(B, 3, self.H, self.W, dtype=torch.float32)
                d = L.StemDesc()
                d.x, d.B, d.H, d.W, d.Ho, d.Wo, d.nsets = self.x_in.data_ptr(), B, self.H, self.W, H0, W0, 1
                put(d, 0, mine)
                self.stem_desc = d
                f.add_desc("simt_stem7_fwd", d, tag="simt_stem7_fwd", flops=2.0 * M0 * 64 * 147,
                           nbytes=float(B * 3 * self.H * self.W * 4 + M0 * 64 * 2), shape=f"M{M0} N64 K147 direct 7x7 s2 (one launch for the networks that share the image)")
            if self.train:
                self._bn_train(f, "bn1", y0, M0, 64)
                f.add("simt_bn_relu_maxpool", y0.data_ptr(), s["scale"].data_ptr(), s["shift"].data_ptr(), pool.data_ptr(),
                      pidx.data_ptr(), B, H0, W0, 64, Hp, Wp, ops.dt_code(dt))
            else:
                one, zero = self.new(64, dtype=torch.float32), self.new(64, dtype=torch.float32, zero=True)
                one.fill_(1.0)
                f.add("simt_bn_relu_maxpool", y0.data_ptr(), one.data_ptr(), zero.data_ptr(), pool.data_ptr(),
                      pidx.data_ptr(), B, H0, W0, 64, Hp, Wp, ops.dt_code(dt))
        elif self.stem_from is not None:
            # stem_from = plan, or (plan, i, n): this plan convolves images [i * B, (i + 1) * B) of the partner's n * B (the frozen forward split
            # into n half-batch chains, step.py SIMT_FROZEN_SPLIT): its rows of the partner's im2col matrix
            o, part, nparts = self.stem_from if isinstance(self.stem_from, tuple) else (self.stem_from, 0, 1)
            assert (o.B, o.H, o.W, o.dtype) == (B * nparts, self.H, self.W, dt) and not o.direct_stem
            self.x_in, A = o.x_in, o.saved["stem.A"][part * M0:(part + 1) * M0]
            self.saved["stem.A"] = A
        else:
            self.x_in = self.new(B, 3, self.H, self.W, dtype=torch.float32)
            A = self.new(M0, KS)
            f.add("simt_im2col_stem", self.x_in.data_ptr(), A.data_ptr(), B, 3, self.H, self.W, H0, W0, 7, 7, 2, 3, KS,
                  ops.dt_code(dt))
            self.saved["stem.A"] = A
        if self.direct_stem:
            pass
        elif self.train:
            wi = self._plan_pack("conv1", 64, 3, 7, K_cin=KS)
            s = self._new_bn("bn1", M0, 64)
            self._conv(f, A, wi, y0, Bn=1, Hi=1, Wi=M0, Cin=KS, Ho=1, Wo=M0, Cout=64, taps=[(0, 0)], stats=s["part"],
                       alg_k=147)
            self._bn_train(f, "bn1", y0, M0, 64)
            f.add("simt_bn_relu_maxpool", y0.data_ptr(), s["scale"].data_ptr(), s["shift"].data_ptr(), pool.data_ptr(),
                  pidx.data_ptr(), B, H0, W0, 64, Hp, Wp, ops.dt_code(dt))
        else:
            sc, sh = self._plan_fold("bn1", 64)
            wi = self._plan_pack("conv1", 64, 3, 7, K_cin=KS, scale_bn="bn1")
            self._conv(f, A, wi, y0, Bn=1, Hi=1, Wi=M0, Cin=KS, Ho=1, Wo=M0, Cout=64, taps=[(0, 0)], bias=sh, relu=True,
                       alg_k=147)
            one, zero = self.new(64, dtype=torch.float32), self.new(64, dtype=torch.float32, zero=True)
            one.fill_(1.0)
            f.add("simt_bn_relu_maxpool", y0.data_ptr(), one.data_ptr(), zero.data_ptr(), pool.data_ptr(),
                  pidx.data_ptr(), B, H0, W0, 64, Hp, Wp, ops.dt_code(dt))
        # ---- bottlenecks
        x, Hc, Wc = pool, Hp, Wp
        self.block_io = []
        feats = {}
        for (name, inpl, planes, stride, dil, down) in self.blocks:
            Ho, Wo = ((Hc - 1) // stride + 1, (Wc - 1) // stride + 1)
            Mi, Mo = B * Hc * Wc, B * Ho * Wo
            c4 = planes * 4
            rec = {"name": name, "x": x, "Hi": Hc, "Wi": Wc, "Ho": Ho, "Wo": Wo, "inpl": inpl, "planes": planes,
                   "stride": stride, "dil": dil, "down": down, "Mi": Mi, "Mo": Mo, "fwd_start": len(f.items)}
            t3 = ops.conv_taps(3, 3, dil, dil)
            if self.train:
                y1, a1 = self.new(Mo, planes), self.new(Mo, planes)
                y2, a2 = self.new(Mo, planes), self.new(Mo, planes)
                y3, z = self.new(Mo, c4), self.new(Mo, c4)
                s1, s2, s3 = (self._new_bn(f"{name}.bn1", Mo, planes), self._new_bn(f"{name}.bn2", Mo, planes),
                              self._new_bn(f"{name}.bn3", Mo, c4))
                w1 = self._plan_pack(f"{name}.conv1", planes, inpl, 1)
                w2 = self._plan_pack(f"{name}.conv2", planes, planes, 3)
                w3 = self._plan_pack(f"{name}.conv3", c4, planes, 1)
                # bn1 / bn2: fused into the producing conv where its grid is one co-resident round (layer 3 at 4 x 768 x 768: 236 tiles):
                # the launch writes y AND a = relu(bn(y)); otherwise statistics slots -> finalize -> apply as separate launches
                dsc = self._conv(f, x, w1, y1, Bn=B, Hi=Hc, Wi=Wc, Cin=inpl, Ho=Ho, Wo=Wo, Cout=planes, taps=[(0, 0)],
                                 stride=stride, stats=s1["part"], fbn=dict(mode=1, out=a1, bname=f"{name}.bn1"))
                if not dsc.fbn:
                    self._bn_train(f, f"{name}.bn1", y1, Mo, planes)
                    f.add("simt_bn_apply", y1.data_ptr(), s1["scale"].data_ptr(), s1["shift"].data_ptr(), None, None, None,
                          None, a1.data_ptr(), Mo, planes, 1, ops.dt_code(dt))
                dsc = self._conv(f, a1, w2, y2, Bn=B, Hi=Ho, Wi=Wo, Cin=planes, Ho=Ho, Wo=Wo, Cout=planes, taps=t3,
                                 stats=s2["part"], fbn=dict(mode=1, out=a2, bname=f"{name}.bn2"))
                # bn2 -> conv3 (round 6): where the row-streaming 1x1 kernel takes conv3 (layers 1-3), bn2's normalise + ReLU runs in ITS operand
                # path (the store waves rewrite every landed stage in LDS and write a2 out on the way): no simt_bn_apply launch, no re-read of y2
                kw3 = dict(Bn=B, Hi=Ho, Wi=Wo, Cin=planes, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)], stats=s3["part"])
                inbn = (not dsc.fbn) and os.environ.get("SIMT_NO_INBN", "0") == "0" and L.load().simt_conv_inbn_ok(C.byref(self._conv(LaunchList(), y2, w3, y3, **kw3))) != 0
                if not dsc.fbn:
                    self._bn_train(f, f"{name}.bn2", y2, Mo, planes)
                    if not inbn:
                        f.add("simt_bn_apply", y2.data_ptr(), s2["scale"].data_ptr(), s2["shift"].data_ptr(), None, None, None,
                              None, a2.data_ptr(), Mo, planes, 1, ops.dt_code(dt))
                if inbn:
                    self._conv(f, y2, w3, y3, inbn=(s2["scale"], s2["shift"], a2), **kw3)
                else:
                    self._conv(f, a2, w3, y3, **kw3)
                rec["inbn"] = bool(inbn)
                self._bn_train(f, f"{name}.bn3", y3, Mo, c4)
                # ReLU mask of the block output as one bit per element: what bn3's backward reads instead of z
                zbits = self.new(Mo, c4 // 8, dtype=torch.uint8)
                rec.update(y1=y1, a1=a1, y2=y2, a2=a2, y3=y3, z=z, zbits=zbits)
                if down:
                    yd = self.new(Mo, c4)
                    sd = self._new_bn(f"{name}.downsample.1", Mo, c4)
                    wd = self._plan_pack(f"{name}.downsample.0", c4, inpl, 1)
                    self._conv(f, x, wd, yd, Bn=B, Hi=Hc, Wi=Wc, Cin=inpl, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)],
                               stride=stride, stats=sd["part"])
                    self._bn_train(f, f"{name}.downsample.1", yd, Mo, c4)
                    f.add("simt_bn_apply_bits", y3.data_ptr(), s3["scale"].data_ptr(), s3["shift"].data_ptr(), None,
                          yd.data_ptr(), sd["scale"].data_ptr(), sd["shift"].data_ptr(), z.data_ptr(), zbits.data_ptr(), Mo, c4, 1,
                          ops.dt_code(dt), tag="simt_bn_apply")
                    rec.update(yd=yd)
                else:
                    f.add("simt_bn_apply_bits", y3.data_ptr(), s3["scale"].data_ptr(), s3["shift"].data_ptr(), x.data_ptr(),
                          None, None, None, z.data_ptr(), zbits.data_ptr(), Mo, c4, 1, ops.dt_code(dt), tag="simt_bn_apply")
            else:
                # eval: BN folded into weights (scale) and bias (shift); ReLU / residual in the conv epilogue
                a1 = self.buf("e.a1", Mo, planes)
                a2 = self.buf("e.a2", Mo, planes)
                z = self.buf("e.z%d" % (len(self.block_io) & 1), Mo, c4)
                _, sh1 = self._plan_fold(f"{name}.bn1", planes)
                _, sh2 = self._plan_fold(f"{name}.bn2", planes)
                _, sh3 = self._plan_fold(f"{name}.bn3", c4)
                w1 = self._plan_pack(f"{name}.conv1", planes, inpl, 1, scale_bn=f"{name}.bn1")
                w2 = self._plan_pack(f"{name}.conv2", planes, planes, 3, scale_bn=f"{name}.bn2")
                w3 = self._plan_pack(f"{name}.conv3", c4, planes, 1, scale_bn=f"{name}.bn3")
                self._conv(f, x, w1, a1, Bn=B, Hi=Hc, Wi=Wc, Cin=inpl, Ho=Ho, Wo=Wo, Cout=planes, taps=[(0, 0)],
                           stride=stride, bias=sh1, relu=True)
                self._conv(f, a1, w2, a2, Bn=B, Hi=Ho, Wi=Wo, Cin=planes, Ho=Ho, Wo=Wo, Cout=planes, taps=t3, bias=sh2,
                           relu=True)
                res = x
                if down:
                    yd = self.buf("e.yd", Mo, c4)
                    _, shd = self._plan_fold(f"{name}.downsample.1", c4)
                    wd = self._plan_pack(f"{name}.downsample.0", c4, inpl, 1, scale_bn=f"{name}.downsample.1")
                    self._conv(f, x, wd, yd, Bn=B, Hi=Hc, Wi=Wc, Cin=inpl, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)],
                               stride=stride, bias=shd)
                    res = yd
                self._conv(f, a2, w3, z, Bn=B, Hi=Ho, Wi=Wo, Cin=planes, Ho=Ho, Wo=Wo, Cout=c4, taps=[(0, 0)], bias=sh3,
                           res=res, relu=True)
            rec["z"] = z
            self.block_io.append(rec)
            x, Hc, Wc = z, Ho, Wo
            li = int(name[5])
            feats[li] = (z, Ho, Wo, c4)
            # heads hang off the LAST block of their layer: emit them right after it so that eval ping-pong
            # buffers are still intact
            last_of_layer = (len(self.block_io) == sum(self.layers[:li]))
            if last_of_layer:
                for hd in self.heads:
                    if hd.feat_layer == li:
                        self._build_head_fwd(hd, z, Ho, Wo, c4)
        self.feat_hw = (Hc, Wc)

    def _build_head_fwd(self, hd, feat, h, w, cin):
        """Classifier_Module (+ torch.cat of the open-set head): ONE implicit GEMM, K = len(dil)*9*Cin, N = Q."""
        B, f = self.B, self.fwd_list
        Q = hd.Q
        nd = len(hd.dilations)
        taps = []
        for d in hd.dilations:
            taps += ops.conv_taps(3, 3, d, d)
        tile = ops.pick_tile_n(Q)
        npad = ops.round_up(Q, tile)
        ldp = ops.round_up(Q, 8) if Q > 32 else 32
        ldp = max(ldp, ops.round_up(Q, 4))
        expanded = self.dtype == torch.bfloat16
        wp = None if expanded else self.new(npad, len(taps) * cin, zero=True)
        bias = self.new(npad, dtype=torch.float32, zero=True)
        self.packed["head." + hd.name] = (wp, tile, npad)
        row = 0
        bias_parts = []
        for prefix, cout in hd.groups:
            for i in range(nd):
                wt = self.p[f"{prefix}.conv2d_list.{i}.weight"]
                if not expanded:
                    self.pack_list.add("simt_pack_weight", wt.data_ptr(), wp.data_ptr(), cout, cin, 9, row, 9 * i,
                                       len(taps) * cin, 0, 0, None, ops.dt_code(self.dtype))
                bias_parts.append((row, cout, self.p[f"{prefix}.conv2d_list.{i}.bias"]))
            row += cout
        hd.bias, hd.bias_parts = bias, bias_parts
        for pi, (brow, bcout, bt) in enumerate(bias_parts):
            # bias of the summed branches = sum of the branch biases (first listing of a row range overwrites)
            first = all(r != brow for (r, _c, _t) in bias_parts[:pi])
            self.pack_list.add("simt_vec_acc", bias.data_ptr() + 4 * brow, bt.data_ptr(), bcout, 0 if first else 1)
        logits = self.new(B * h * w, ldp, dtype=torch.float32, zero=True)
        self.out[hd.name] = logits.view(B, h, w, ldp)
        self.ldp[hd.name] = ldp
        hd.feat, hd.h, hd.w, hd.cin, hd.taps = feat, h, w, cin, taps
        hd.expanded = expanded
        if not hd.expanded:
            self._conv(f, feat, (wp, tile, npad), logits, Bn=B, Hi=h, Wi=w, Cin=cin, Ho=h, Wo=w, Cout=Q, taps=taps, bias=bias,
                       ldy=ldp, Nstore=min(ldp, npad))
            return
        # ---- bf16 throughput path: tap-expanded GEMM (csrc/head_expand.hip): P = feat x Wexp^T, then a tap gather-sum
        Mh, nt = B * h * w, len(taps)
        QP = ops.round_up(Q, 8)
        nexp = nt * QP
        npe = ops.round_up(nexp, 256)
        wexp = self.new(npe, cin, zero=True)
        row = 0
        for prefix, cout in hd.groups:
            for i in range(nd):
                wt = self.p[f"{prefix}.conv2d_list.{i}.weight"]
                self.pack_list.add("simt_pack_weight", wt.data_ptr(), wexp.data_ptr(), cout, cin, 9, row, 9 * i, cin, QP, 2, None,
                                   ops.dt_code(self.dtype))
            row += cout
        P = self.new(Mh, nexp, dtype=torch.float32)
        hd.QP, hd.nexp = QP, nexp
        self._conv(f, feat, (wexp, 256, npe), P, Bn=B, Hi=h, Wi=w, Cin=cin, Ho=h, Wo=w, Cout=nexp, taps=[(0, 0)], ldy=nexp,
                   Nstore=nexp, alg_flops=2.0 * Mh * Q * nt * cin, note=" (tap-expanded head)")
        td = L.TapDesc()
        td.src, td.bias, td.dst = P.data_ptr(), bias.data_ptr(), logits.data_ptr()
        td.B, td.H, td.W, td.Q, td.QP, td.lds, td.ldd, td.ntaps = B, h, w, Q, QP, nexp, ldp, nt
        ops._fill_taps(td.dy, td.dx, taps)
        f.add_desc("simt_tap_gather_sum", td)

    # ------------------------------------------------------------------ gradients
    def grad_param_names(self):
        """Every tensor that receives a gradient (104 trunk convs + live head branches)."""
        names = ["conv1.weight"]
        for (name, *_r, down) in self.blocks:
            names += [f"{name}.conv1.weight", f"{name}.conv2.weight", f"{name}.conv3.weight"]
            if down:
                names.append(f"{name}.downsample.0.weight")
        for hd in self.heads:
            for prefix, _ in hd.groups:
                for i in range(len(hd.dilations)):
                    names += [f"{prefix}.conv2d_list.{i}.weight", f"{prefix}.conv2d_list.{i}.bias"]
        return names

    def _alloc_grads(self, grad_names=None):
        names = grad_names or self.grad_param_names()
        # reverse-topological order: heads of layer4, layer4 ..., heads of layer3, layer3, ..., stem
        order = []
        by_layer = {}
        for n in names:
            if n.startswith("conv1"):
                li = 0
            elif n.startswith("layer") and n[5].isdigit() and "." in n and n.split(".")[0] in ("layer1", "layer2", "layer3",
                                                                                               "layer4"):
                li = int(n[5])
            else:
                li = None
            by_layer.setdefault(li, []).append(n)
        head_names = by_layer.get(None, [])
        head_feat = {}
        for hd in self.heads:
            for prefix, _ in hd.groups:
                head_feat[prefix] = hd.feat_layer
        for li in (4, 3, 2, 1, 0):
            order += [n for n in head_names if head_feat.get(n.split(".")[0]) == li]
            blk = by_layer.get(li, [])
            # blocks in reverse order
            order += sorted(blk, key=lambda n: -int(n.split(".")[1]) if li else 0)
        layout_flat_grads(self, order)

    # ------------------------------------------------------------------ backward construction
    def _wgrad(self, lst, dy, x, gname, *, Bn, Hi, Wi, Cin, Ho, Wo, Cd, ldd, taps, stride, parts, stream=1):
        """parts: [(param name, co_off, tap_off, Cout, RS, Cin_dst)] slices of the slab reduced into OIHW gradients."""
        M = Bn * Ho * Wo
        Ktot = len(taps) * Cin
        nsplit = ops.wgrad_nsplit(M, Cd, Ktot, self.dtype)
        assert nsplit * Cd * Ktot <= self._slab_cap
        slab = self.buf("wgrad.slab" if stream == 1 else "wgrad.slab.main", self._slab_cap, dtype=torch.float32)    # (one per stream: in-order reuse)
        d = ops.make_wgrad_desc(dy, x, slab, B=Bn, H=Hi, W=Wi, Cin=Cin, Ho=Ho, Wo=Wo, Cd=Cd, taps=taps, stride=stride,
                                nsplit=nsplit, ldd=ldd)
        alg_cd = sum(pt[3] for pt in parts) // max(1, len({pt[2] for pt in parts}))
        lst.add_desc("simt_conv_wgrad", d, tag=f"conv_wgrad<{'bf16' if self.dtype == torch.bfloat16 else 'f32'}>",
                     flops=2.0 * M * alg_cd * (147 if parts[0][0] == "conv1.weight" else Ktot),
                     nbytes=float((M * ldd + Bn * Hi * Wi * Cin) * self.esz + nsplit * Cd * Ktot * 4),
                     shape=f"M{M} Cd{Cd} K{Ktot} taps{len(taps)} split{nsplit}", stream=stream)
        for (pname, co_off, tap_off, cout, rs, cin_dst) in parts:
            lst.add("simt_wgrad_reduce", slab.data_ptr(), self.grads[pname].data_ptr(), nsplit, Cd, Ktot, cin_dst, co_off,
                    tap_off, cout, rs, 0, stream=stream)
            self.grad_ready[pname] = len(lst)

    def _wgrad_grouped(self):
        """Group the weight-gradient GEMMs of a Bottleneck into one launch (simt_conv_wgrad_multi)?  bf16 plans; SIMT_WGRAD_GROUP=0: one
        launch per conv, each behind its own event (the round-2 schedule)."""
        return self.dtype == torch.bfloat16 and os.environ.get("SIMT_WGRAD_GROUP", "1") != "0"

    def _wgrad_group(self, lst, jobs, stream=1):
        """jobs: keyword dicts of _wgrad (dy, x, Bn, Hi, Wi, Cin, Ho, Wo, Cd, ldd, taps, stride, parts), all over the same output pixels.
        One grouped launch with a shared pixel split count, then each problem's fixed-order slab reduce."""
        M = jobs[0]["Bn"] * jobs[0]["Ho"] * jobs[0]["Wo"]
        assert all(j["Bn"] * j["Ho"] * j["Wo"] == M for j in jobs)
        kt = [len(j["taps"]) * j["Cin"] for j in jobs]
        tco = ops.wgrad_group_tile_co(M, [(j["Cd"], k) for j, k in zip(jobs, kt)])
        tiles = sum(ops.wgrad_tiles(M, j["Cd"], k, tco) for j, k in zip(jobs, kt))
        ns = ops.wgrad_group_nsplit(M, tiles)
        total = sum(ns * j["Cd"] * k for j, k in zip(jobs, kt))
        assert total <= self._slab_cap
        slab = self.buf("wgrad.slab" if stream == 1 else "wgrad.slab.main", self._slab_cap, dtype=torch.float32)
        descs, slabs, off = [], [], 0
        for j, k in zip(jobs, kt):
            sl = slab[off:off + ns * j["Cd"] * k]
            off += ns * j["Cd"] * k
            slabs.append(sl)
            descs.append(ops.make_wgrad_desc(j["dy"], j["x"], sl, B=j["Bn"], H=j["Hi"], W=j["Wi"], Cin=j["Cin"], Ho=j["Ho"], Wo=j["Wo"],
                                             Cd=j["Cd"], taps=j["taps"], stride=j["stride"], nsplit=ns, ldd=j["ldd"]))
        if not all(ops.wgrad_multi_ok(d) for d in descs) or len(descs) > 16:
            for j in jobs:
                self._wgrad(lst, j["dy"], j["x"], None, **{k: v for k, v in j.items() if k not in ("dy", "x")}, stream=stream)
            return
        table, grid, tco_c = ops.wgrad_multi_table(descs, self.dev)
        assert tco_c == tco and grid == tiles * ns
        flops = sum(2.0 * M * j["Cd"] * k for j, k in zip(jobs, kt))
        nbytes = sum((M * j["ldd"] + j["Bn"] * j["Hi"] * j["Wi"] * j["Cin"]) * self.esz + ns * j["Cd"] * k * 4 for j, k in zip(jobs, kt))
        lst.add("simt_conv_wgrad_multi", table.data_ptr(), len(descs), grid, ns, tco, keep=(table, descs, slabs), tag="conv_wgrad<bf16>",
                flops=flops, nbytes=float(nbytes),
                shape=f"M{M} group{len(descs)} " + "+".join(f"Cd{j['Cd']}K{k}" for j, k in zip(jobs, kt)) + f" tile{tco} split{ns}",
                stream=stream)
        rjobs, names = [], []
        for j, k, sl in zip(jobs, kt, slabs):
            for (pname, co_off, tap_off, cout, rs, cin_dst) in j["parts"]:
                rjobs.append(dict(slab=sl, dst=self.grads[pname], nsplit=ns, Cd=j["Cd"], Ktot=k, Cin=cin_dst, co_off=co_off, tap_off=tap_off,
                                  Cout=cout, RS=rs))
                names.append(pname)
        rt, rn, rblocks = ops.wgrad_reduce_multi_table(rjobs, self.dev)
        lst.add("simt_wgrad_reduce_multi", rt.data_ptr(), rn, rblocks, keep=(rt, rjobs), tag="simt_wgrad_reduce",
                nbytes=float(sum(ns * r["Cout"] * r["RS"] * r["Cin"] * 4 for r in rjobs)), stream=stream)
        for pname in names:
            self.grad_ready[pname] = len(lst)

    def _bnr(self, bname, y, mode, bits=None):
        """Fused first pass of `bname`'s backward for the conv that produces its dz (bf16 v2 kernel; see simt_conv_desc.bnr_*).
        Returns None when the fusion does not apply (fp32 parity plans run the separate reduce kernel)."""
        if self.dtype != torch.bfloat16 or os.environ.get("SIMT_BN_FUSE") == "0":
            return None
        sb = self.bn[bname]
        return {"y": y, "mean": sb["mean"], "rstd": sb["rstd"], "scale": sb["scale"], "shift": sb["shift"], "bits": bits, "mode": mode,
                "part": self.buf("bnb.part", self._bnb_cap, dtype=torch.float32)}

    def _fused_nblk(self, d, bnr):
        """Slot count the conv wrote (0: the launch does not run on the v2 kernel -> the descriptor's fusion is switched off)."""
        if bnr is None:
            return 0
        n = L.load().simt_conv_mtiles(C.byref(d))
        if n == 0:
            d.bnr_mode = 0
        assert n * 3 * d.Cout <= self._bnb_cap
        return n

    def _bn_bwd(self, lst, *, dz, y, bname, dy, M, Cn, mask_mode, z=None, y2=None, bname2=None, dy2=None, gout=None,
                affine=False, reduce_done_nblk=0):
        """affine: also write d gamma / d beta into self.grads[bname.weight / .bias] (trainable BatchNorm, engine_v3)."""
        s = self.bn[bname]
        ag = {}
        if affine:
            ag = dict(dgamma=self.grads[bname + ".weight"], dbeta=self.grads[bname + ".bias"])
            if bname2:
                ag.update(dgamma2=self.grads[bname2 + ".weight"], dbeta2=self.grads[bname2 + ".bias"])
        s2 = self.bn[bname2] if bname2 else None
        nblk = ops.bn_bwd_nblk(M, Cn)
        part = self.buf("bnb.part", self._bnb_cap, dtype=torch.float32)
        coef = self.buf("bnb.coef", 3 * 2048, dtype=torch.float32)
        assert nblk * 3 * Cn <= self._bnb_cap
        d = ops.make_bn_bwd_desc(dz=dz, y=y, mean=s["mean"], rstd=s["rstd"], scale=s["scale"], shift=s["shift"], part=part,
                                 coef=coef, dy=dy, M=M, Cn=Cn, mask_mode=mask_mode, z=z, y2=y2,
                                 mean2=s2["mean"] if s2 else None, rstd2=s2["rstd"] if s2 else None,
                                 scale2=s2["scale"] if s2 else None, dy2=dy2, gout=gout, reduce_done_nblk=reduce_done_nblk, **ag)
        lst.add_desc("simt_bn_bwd", d)

    def _build_backward(self):
        B, dt, kq = self.B, self.dtype, self.kq
        b = self.bwd_list
        # workspace capacities
        self._slab_cap = 1
        self._bnb_cap = 1
        for rec in self.block_io:
            Mo, p, inpl = rec["Mo"], rec["planes"], rec["inpl"]
            for (cd, kt) in ((p, inpl), (p, 9 * p), (4 * p, p), (4 * p, inpl)):
                ns = ops.wgrad_nsplit(Mo, cd, kt, dt)
                self._slab_cap = max(self._slab_cap, ns * cd * kt)
            for cn in (p, 4 * p):
                self._bnb_cap = max(self._bnb_cap, ops.bn_bwd_nblk(Mo, cn) * 3 * cn)
        if self._wgrad_grouped():
            def blk_shapes(rec):
                p, inpl = rec["planes"], rec["inpl"]
                return [(p, inpl), (p, 9 * p), (4 * p, p)] + ([(4 * p, inpl)] if rec["down"] else [])
            nb = len(self.block_io)
            cands = [self.block_io[i:i + n] for n in range(1, 6) for i in range(nb - n + 1)]       # every run of up to 5 consecutive blocks
            for grp in cands:
                Mo = grp[0]["Mo"]
                shp = sum((blk_shapes(r) for r in grp), [])
                if any(r["Mo"] != Mo for r in grp) or len(shp) > 16:
                    continue
                tco = ops.wgrad_group_tile_co(Mo, shp)
                ns = ops.wgrad_group_nsplit(Mo, sum(ops.wgrad_tiles(Mo, cd, kt, tco) for cd, kt in shp))
                self._slab_cap = max(self._slab_cap, ns * sum(cd * kt for cd, kt in shp))
        M0 = B * self.H0 * self.W0
        self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(M0, 64, 192, dt) * 64 * 192, 256 * 64 * 7 * 32)     # (256 x ...: simt_stem7_wgrad's partials)
        self._bnb_cap = max(self._bnb_cap, ops.bn_bwd_nblk(M0, 64) * 3 * 64)
        for hd in self.heads:
            Mh = B * hd.h * hd.w
            cd = ops.round_up(hd.Q, 8)
            kt = len(hd.taps) * hd.cin
            self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(Mh, cd, kt, dt) * cd * kt)
            if getattr(hd, "expanded", False):
                with ops.wgrad_plan(256, 6):       # (the plan _build_head_bwd uses for the tap-expanded heads' weight gradients)
                    self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(Mh, hd.nexp, hd.cin, dt) * hd.nexp * hd.cin)
                self._slab_cap = max(self._slab_cap, ops.wgrad_nsplit(Mh, hd.nexp, hd.cin, dt) * hd.nexp * hd.cin)

        # upstream gradients of the head logits, in the conv dtype, K-padded for the dgrad GEMM
        self.dlogits = {}
        for hd in self.heads:
            ck = ops.round_up(hd.Q, kq)
            self.dlogits[hd.name] = self.new(B * hd.h * hd.w, ck, zero=True)
            hd.ck = ck

        heads_by_layer = {}
        for hd in self.heads:
            heads_by_layer.setdefault(hd.feat_layer, []).append(hd)

        # conv1's weight gradient straight from the image too (simt_stem7_wgrad): then no im2col matrix exists anywhere in the step
        self.direct_stem_wgrad = self.direct_stem and os.environ.get("SIMT_DIRECT_STEM_WGRAD", "1") != "0"
        if self.direct_stem and self.grads_from_layer == 0 and not self.direct_stem_wgrad:
            # the forward convolved the image directly: the im2col matrix exists only for the stem's weight gradient at the very end of this
            # list.  Built FIRST, on the side stream (idle between the frozen forward and the first weight gradients): an HBM-bound 134-us pass
            # beside the heads' MFMA-bound gradient GEMMs instead of on the tail of the backward
            A = self.buf("g.stemA", B * self.H0 * self.W0, 192)
            self.saved["stem.A"] = A
            b.add("simt_im2col_stem", self.x_in.data_ptr(), A.data_ptr(), B, 3, self.H, self.W, self.H0, self.W0, 7, 7, 2, 3, 192,
                  ops.dt_code(dt), stream=1)
        n_blocks = len(self.block_io)
        self.grad_ready = {}  # param name -> number of backward launches after which its gradient is final
        self.bwd_marks = {}   # block name -> (first launch, end launch, dz buffer, dx buffer): debugging / DP buckets
        dz = None  # gradient w.r.t. the current block's output z
        # Two-stream schedule: the dgrad / BN-backward chain is the critical path (stream 0); every weight-gradient GEMM
        # (+ slab reduce, bias column sums) runs on the side stream (1) behind an event recorded after the kernel that
        # produced its dY.  dY buffers rotate over the blocks; stream 0 waits for the side stream's work of
        # the block SIMT_DY_BUFFERS (default 4; memory is not the constraint) steps back before it overwrites them.
        e0 = b.record(0)
        b.wait(e0, 1)
        npar = max(2, int(os.environ.get("SIMT_DY_BUFFERS", "6" if self._wgrad_grouped() else "4")))      # dY buffer sets: the dgrad chain may run this many blocks ahead of the weight gradients
        last_side = {i: None for i in range(npar)}
        # Grouped weight gradients: the jobs of up to SIMT_WGRAD_BLOCKS (default 3) consecutive Bottlenecks over the same pixels (and the
        # same tile rule) go into one launch (SIMT_WGRAD_PAIR=0: one Bottleneck per launch): 34 tiles of 256 x 256 fill the chip with 7
        # pixel splits where one block's 17 need 15 -- half the slab bytes again; 51 tiles with 5.  Measured per step: 26.03 ms one block per
        # launch, 25.88 two, 25.80 three, 25.80 four (the weight gradients of a group start when its LAST block's dY is there: the dY
        # buffer sets, SIMT_DY_BUFFERS, must outnumber the blocks of a group by three).
        pair_ok = os.environ.get("SIMT_WGRAD_PAIR", "1") != "0"
        nblk_group = max(1, min(npar - 3, int(os.environ.get("SIMT_WGRAD_BLOCKS", "3")))) if pair_ok else 1
        pend = {"jobs": [], "pars": [], "M": None, "tco": None, "blocks": 0}

        per_block_events = os.environ.get("SIMT_WGRAD_EVENT_PER_BLOCK") == "1"      # A/B switch: round 4's one event per Bottleneck
        waited_main = set()

        def flush_wgrads():
            if not pend["jobs"]:
                return
            # ONE main -> side edge per weight-gradient GROUP (round 5), recorded here: everything the group reads (the dY / dy buffers of up
            # to three Bottlenecks) is already enqueued on the main stream.  Until round 4 every block recorded its own event although its jobs
            # only left with the group's launch: a record idles the recording queue for ~6.5 us (profiles/r05_conv_attribution.txt section 6),
            # 22 of them per step on the critical dgrad / BatchNorm chain.
            if not per_block_events:
                b.wait(b.record(0), 1)
            self._wgrad_group(b, pend["jobs"])
            ev = b.record(1)
            for q in pend["pars"]:
                last_side[q] = ev
            pend.update(jobs=[], pars=[], M=None, tco=None, blocks=0)
        pending_bn3 = 0       # slots of bn3-backward partials the previous iteration's dx GEMM already reduced (0: none)
        for bi in range(n_blocks - 1, -1, -1):
            rec = self.block_io[bi]
            name, Mo, Mi, p, inpl = rec["name"], rec["Mo"], rec["Mi"], rec["planes"], rec["inpl"]
            c4 = 4 * p
            Ho, Wo, Hi, Wi, stride, dil, down = rec["Ho"], rec["Wo"], rec["Hi"], rec["Wi"], rec["stride"], rec["dil"], rec["down"]
            li = int(name[5])
            if li < self.grads_from_layer:
                break
            first_needed = (li == self.grads_from_layer and bi == sum(self.layers[:li - 1]) and self.grads_from_layer > 0)
            last_of_layer = (bi + 1 == sum(self.layers[:li]))
            if last_of_layer and li in heads_by_layer:
                hds = heads_by_layer[li]
                for hi, hd in enumerate(hds):
                    # the LAST head GEMM of this layer produces the block's final dz: reduce for its bn3 backward there
                    bnr = self._bnr(f"{name}.bn3", rec["y3"], 3, bits=rec["zbits"]) if (hi == len(hds) - 1 and not down) else None
                    dz = self._build_head_bwd(hd, dz, Mo, c4, bi, bnr=bnr)
                    pending_bn3 = self._head_bnr_nblk
            assert dz is not None, "no gradient reaches the last block (a head must sit on the last layer)"
            blk_start = len(b)
            par = bi % npar
            if last_side[par] is not None and (per_block_events or id(last_side[par]) not in waited_main):
                # (one wait per side-stream event: the three Bottlenecks of a weight-gradient group share the event behind their group's launch)
                b.wait(last_side[par], 0)
                waited_main.add(id(last_side[par]))
            # ---- z = relu(bn3(y3) + shortcut)
            dy3 = self.buf("g.dy3.%d" % par, Mo, c4)
            dyd = self.buf("g.dyd.%d" % par, Mo, c4) if down else None
            # identity blocks: the shortcut gradient dz * (z > 0) is never written -- the conv that produces dx adds dz under
            # the bit mask (simt_conv_desc.res_bits)
            self._bn_bwd(b, dz=dz, z=rec["zbits"], y=rec["y3"], bname=f"{name}.bn3", dy=dy3, M=Mo, Cn=c4, mask_mode=3,
                         y2=rec.get("yd"), bname2=f"{name}.downsample.1" if down else None, dy2=dyd,
                         reduce_done_nblk=pending_bn3)
            pending_bn3 = 0
            # conv3
            grouped = self._wgrad_grouped()
            wjobs = [dict(dy=dy3, x=rec["a2"], Bn=B, Hi=Ho, Wi=Wo, Cin=p, Ho=Ho, Wo=Wo, Cd=c4, ldd=c4, taps=[(0, 0)], stride=1,
                          parts=[(f"{name}.conv3.weight", 0, 0, c4, 1, p)])]
            if not grouped:
                b.wait(b.record(0), 1)
                j = wjobs[0]
                self._wgrad(b, j["dy"], j["x"], None, **{k: v for k, v in j.items() if k not in ("dy", "x")})
            wt3 = self._plan_pack_t(f"{name}.conv3", c4, p, 1)
            da2 = self.buf("g.da", Mo, p)
            bnr = self._bnr(f"{name}.bn2", rec["y2"], 2)
            dy2 = self.buf("g.dy2.%d" % par, Mo, p)
            coef = self.buf("bnb.coef", 3 * 2048, dtype=torch.float32)
            # the BatchNorm backward fused into the dgrad launch (same condition as the forward): dy2 straight from the tile in LDS, the raw
            # dz (da2) is never written
            dsc = self._conv(b, dy3, wt3[:3], da2, Bn=B, Hi=Ho, Wi=Wo, Cin=wt3[3], Ho=Ho, Wo=Wo, Cout=p, taps=[(0, 0)], bnr=bnr,
                             fbn=dict(mode=2, out=dy2, bname=f"{name}.bn2", coef=coef) if bnr else None)
            if not dsc.fbn:
                self._bn_bwd(b, dz=da2, y=rec["y2"], bname=f"{name}.bn2", dy=dy2, M=Mo, Cn=p, mask_mode=2,
                             reduce_done_nblk=self._fused_nblk(dsc, bnr))
            # conv2 (3x3 dilated)
            t3 = ops.conv_taps(3, 3, dil, dil)
            wjobs.append(dict(dy=dy2, x=rec["a1"], Bn=B, Hi=Ho, Wi=Wo, Cin=p, Ho=Ho, Wo=Wo, Cd=p, ldd=p, taps=t3, stride=1,
                              parts=[(f"{name}.conv2.weight", 0, 0, p, 9, p)]))
            if not grouped:
                b.wait(b.record(0), 1)
                j = wjobs[-1]
                self._wgrad(b, j["dy"], j["x"], None, **{k: v for k, v in j.items() if k not in ("dy", "x")})
            wt2 = self._plan_pack_t(f"{name}.conv2", p, p, 3)
            da1 = self.buf("g.da", Mo, p)
            # dy2 has p channels; the dgrad operand is K-padded to ck >= p: equal here because p % kq == 0
            assert wt2[3] == p and wt3[3] == c4
            bnr = self._bnr(f"{name}.bn1", rec["y1"], 2)
            dy1 = self.buf("g.dy1.%d" % par, Mo, p)
            dsc = self._conv(b, dy2, wt2[:3], da1, Bn=B, Hi=Ho, Wi=Wo, Cin=p, Ho=Ho, Wo=Wo, Cout=p,
                             taps=[(-a, -c) for (a, c) in t3], bnr=bnr,
                             fbn=dict(mode=2, out=dy1, bname=f"{name}.bn1", coef=coef) if bnr else None)
            if not dsc.fbn:
                self._bn_bwd(b, dz=da1, y=rec["y1"], bname=f"{name}.bn1", dy=dy1, M=Mo, Cn=p, mask_mode=2,
                             reduce_done_nblk=self._fused_nblk(dsc, bnr))
            # conv1 (+ downsample) wgrads
            if not grouped or per_block_events:
                b.wait(b.record(0), 1)
            wjobs.append(dict(dy=dy1, x=rec["x"], Bn=B, Hi=Hi, Wi=Wi, Cin=inpl, Ho=Ho, Wo=Wo, Cd=p, ldd=p, taps=[(0, 0)], stride=stride,
                              parts=[(f"{name}.conv1.weight", 0, 0, p, 1, inpl)]))
            if down:
                wjobs.append(dict(dy=dyd, x=rec["x"], Bn=B, Hi=Hi, Wi=Wi, Cin=inpl, Ho=Ho, Wo=Wo, Cd=c4, ldd=c4, taps=[(0, 0)],
                                  stride=stride, parts=[(f"{name}.downsample.0.weight", 0, 0, c4, 1, inpl)]))
            if grouped:
                # ONE launch for the block's weight gradients: 34 output tiles (layer 3) instead of 18 / 8 / 8, so 7 pixel splits fill the
                # chip where the single launches need 14 / 31 / 31 (a third of the fp32 slabs, one launch-shaped overhead instead of three)
                tco_b = ops.wgrad_group_tile_co(Mo, [(j["Cd"], len(j["taps"]) * j["Cin"]) for j in wjobs])
                if pend["jobs"] and (pend["M"] != Mo or pend["tco"] != tco_b or len(pend["jobs"]) + len(wjobs) > 16):
                    flush_wgrads()
                pend["jobs"] += wjobs
                pend["pars"].append(par)
                pend.update(M=Mo, tco=tco_b, blocks=pend["blocks"] + 1)
                # ... and at every layer boundary (block 0 of a layer is the last one the backward reaches): a group must not straddle the
                # applied / unapplied line (layer3.0 with layer2's blocks at small crops), or the last APPLIED gradient -- the early
                # optimiser step and the final DP bucket wait for it -- would become final up to two blocks late
                if pend["blocks"] >= nblk_group or first_needed or bi == sum(self.layers[:li - 1]):
                    flush_wgrads()
            else:
                for j in wjobs[2:]:
                    self._wgrad(b, j["dy"], j["x"], None, **{k: v for k, v in j.items() if k not in ("dy", "x")})
                last_side[par] = b.record(1)
            if first_needed:          # nothing below this block needs a gradient
                self.bwd_marks[name] = (blk_start, len(b), dz, None)
                continue
            # input gradient
            wt1 = self._plan_pack_t(f"{name}.conv1", p, inpl, 1)
            assert wt1[3] == p
            dx = self.buf("g.dx%d" % (bi & 1), Mi, inpl)
            if stride == 1:
                if down:
                    wtd = self._plan_pack_t(f"{name}.downsample.0", c4, inpl, 1)
                    dxd = self.buf("g.dxd", Mo, inpl)
                    self._conv(b, dyd, wtd[:3], dxd, Bn=B, Hi=Ho, Wi=Wo, Cin=c4, Ho=Ho, Wo=Wo, Cout=inpl, taps=[(0, 0)])
                    res, rbits = dxd, None
                else:
                    res, rbits = dz, rec["zbits"]
                # this dx is the dz of the block below: reduce for its bn3 backward here, unless that block has a downsample
                # partner (third sum) or receives a head's gradient on top
                bnr = None
                if bi > 0:
                    prev = self.block_io[bi - 1]
                    pli = int(prev["name"][5])
                    head_fed = (bi == sum(self.layers[:pli])) and pli in heads_by_layer
                    if not prev["down"] and not head_fed and pli >= self.grads_from_layer:
                        bnr = self._bnr(f"{prev['name']}.bn3", prev["y3"], 3, bits=prev["zbits"])
                dsc = self._conv(b, dy1, wt1[:3], dx, Bn=B, Hi=Ho, Wi=Wo, Cin=p, Ho=Ho, Wo=Wo, Cout=inpl, taps=[(0, 0)], res=res,
                                 res_bits=rbits, bnr=bnr)
                pending_bn3 = self._fused_nblk(dsc, bnr)
            else:
                assert down
                wtd = self._plan_pack_t(f"{name}.downsample.0", c4, inpl, 1)
                dxd = self.buf("g.dxd", Mo, inpl)
                dxl = self.buf("g.dxl", Mo, inpl)
                self._conv(b, dyd, wtd[:3], dxd, Bn=B, Hi=Ho, Wi=Wo, Cin=c4, Ho=Ho, Wo=Wo, Cout=inpl, taps=[(0, 0)])
                self._conv(b, dy1, wt1[:3], dxl, Bn=B, Hi=Ho, Wi=Wo, Cin=p, Ho=Ho, Wo=Wo, Cout=inpl, taps=[(0, 0)], res=dxd)
                b.add("simt_scatter_stride", dxl.data_ptr(), dx.data_ptr(), B, Hi, Wi, inpl, Ho, Wo, stride, ops.dt_code(dt))
            self.bwd_marks[name] = (blk_start, len(b), dz, dx)
            dz = dx
        flush_wgrads()
        if self.grads_from_layer > 0:
            for n in self.grads:                   # never written: final (zero) from the start, for the DP bucket schedule
                self.grad_ready.setdefault(n, 0)
            b.wait(b.record(1), 0)
            return
        # ---- stem: maxpool -> relu/bn -> conv1 wgrad (computed like the reference does)
        H0, W0, Hp, Wp = self.H0, self.W0, self.Hp, self.Wp
        da0 = self.buf("g.da0", M0, 64)
        b.add("simt_maxpool_bwd", dz.data_ptr(), self.saved["stem.idx"].data_ptr(), da0.data_ptr(), B, H0, W0, 64, Hp, Wp,
              ops.dt_code(dt))
        dy0 = self.buf("g.dy0", M0, 64)
        self._bn_bwd(b, dz=da0, y=self.saved["stem.y"], bname="bn1", dy=dy0, M=M0, Cn=64, mask_mode=2)
        # the very last weight gradient runs on the main stream (which has nothing else left) beside the side stream's backlog of
        # layer1 weight gradients instead of behind it
        stem_stream = 0 if os.environ.get("SIMT_STEM_WGRAD_MAIN", "1") != "0" else 1
        if stem_stream == 1:
            b.wait(b.record(0), 1)
        if self.direct_stem_wgrad:
            nwg = L.load().simt_stem7_wgrad_workgroups(B, H0, W0)
            part = self.buf("wgrad.slab" if stem_stream == 1 else "wgrad.slab.main", self._slab_cap, dtype=torch.float32)
            assert nwg * 64 * 7 * 32 <= self._slab_cap
            b.add("simt_stem7_wgrad", self.x_in.data_ptr(), dy0.data_ptr(), part.data_ptr(), self.grads["conv1.weight"].data_ptr(), B, self.H,
                  self.W, H0, W0, stream=stem_stream, flops=2.0 * M0 * 64 * 147,
                  nbytes=float(M0 * 64 * self.esz + B * 3 * self.H * self.W * 4 + 2 * nwg * 64 * 7 * 32 * 4), shape=f"M{M0} direct")
            self.grad_ready["conv1.weight"] = len(b)
        else:
            self._wgrad(b, dy0, self.saved["stem.A"], None, Bn=1, Hi=1, Wi=M0, Cin=192, Ho=1, Wo=M0, Cd=64, ldd=64,
                        taps=[(0, 0)], stride=1, parts=[("conv1.weight", 0, 0, 64, 1, 147)], stream=stem_stream)
        b.wait(b.record(1), 0)        # join: the optimiser (stream 0) sees every gradient

    def _build_head_bwd(self, hd, dz_prev, Mo, c4, bi, bnr=None):
        """wgrad/bias grads of the fused ASPP GEMM and its dgrad into the feature gradient (added to dz_prev).  bnr: fused first
        pass of the feature block's bn3 backward (the dgrad GEMM produces that block's dz); the slot count is left in
        self._head_bnr_nblk."""
        self._head_bnr_nblk = 0
        B, b = self.B, self.bwd_list
        dl = self.dlogits[hd.name]
        Mh = B * hd.h * hd.w
        nd = len(hd.dilations)
        # bias gradients: ONE column sum of dlogits over all Q columns, then copied into every live branch's bias gradient
        # (the branches of a group are summed in the forward, so their bias gradients are equal)
        colq = self.new(hd.ck, dtype=torch.float32, zero=True)
        b.add("simt_colsum", dl.data_ptr(), colq.data_ptr(), Mh, hd.ck, hd.Q, 0, ops.dt_code(self.dtype), stream=1)
        row = 0
        for prefix, cout in hd.groups:
            for i in range(nd):
                gname = f"{prefix}.conv2d_list.{i}.bias"
                b.add("simt_vec_acc", self.grads[gname].data_ptr(), colq.data_ptr() + 4 * row, cout, 0, stream=1)
                self.grad_ready[gname] = len(b)
            row += cout
        dfeat = self.buf("g.dfeat%d" % hd.feat_layer, Mo, c4)
        tile = ops.pick_tile_n(hd.cin, self.dtype)
        npad = ops.round_up(hd.cin, tile)
        if not hd.expanded:
            cd = ops.round_up(hd.Q, 8)
            parts = []
            row = 0
            for prefix, cout in hd.groups:
                for i in range(nd):
                    parts.append((f"{prefix}.conv2d_list.{i}.weight", row, 9 * i, cout, 9, hd.cin))
                row += cout
            self._wgrad(b, dl, hd.feat, None, Bn=B, Hi=hd.h, Wi=hd.w, Cin=hd.cin, Ho=hd.h, Wo=hd.w, Cd=cd, ldd=hd.ck,
                        taps=hd.taps, stride=1, parts=parts)
            # dgrad: operand [Cin][ntaps*ck] assembled from every branch / group
            wt = self.new(npad, len(hd.taps) * hd.ck, zero=True)
            row = 0
            for prefix, cout in hd.groups:
                for i in range(nd):
                    w = self.p[f"{prefix}.conv2d_list.{i}.weight"]
                    self.pack_list.add("simt_pack_weight", w.data_ptr(), wt.data_ptr(), cout, hd.cin, 9, row, 9 * i,
                                       len(hd.taps) * hd.ck, hd.ck, 1, None, ops.dt_code(self.dtype))
                row += cout
            if getattr(hd, "mask", None) is not None:
                bnr = None
            dsc = self._conv(b, dl, (wt, tile, npad), dfeat, Bn=B, Hi=hd.h, Wi=hd.w, Cin=hd.ck, Ho=hd.h, Wo=hd.w, Cout=hd.cin,
                             taps=[(-a, -c) for (a, c) in hd.taps], res=dz_prev, alg_k=len(hd.taps) * hd.Q,
                             mask=getattr(hd, "mask", None), bnr=bnr)
            self._head_bnr_nblk = self._fused_nblk(dsc, bnr)
            return dfeat
        # ---- tap-expanded backward: G[m'][t*QP+n] = dlogits[m' - d_t][n]; dW = G^T x feat; dfeat = G x Wt (+ dz_prev)
        nt, QP, nexp = len(hd.taps), hd.QP, hd.nexp
        kexp = ops.round_up(nexp, self.kq)
        G = self.new(Mh, kexp, zero=True)
        td = L.TapDesc()
        td.src, td.bias, td.dst = dl.data_ptr(), None, G.data_ptr()
        td.B, td.H, td.W, td.Q, td.QP, td.lds, td.ldd, td.ntaps = B, hd.h, hd.w, hd.Q, QP, hd.ck, kexp, nt
        ops._fill_taps(td.dy, td.dx, hd.taps)
        b.add_desc("simt_tap_scatter", td)
        b.wait(b.record(0), 1)
        # (the heads' weight gradients are the FIRST ones of the backward: nothing of the dgrad / BatchNorm chain runs beside them yet, so they
        # keep the whole-chip split plan; SIMT_HEAD_WGRAD_HALF=1: planned like the trunk's, for half the chip -- A/B)
        if os.environ.get("SIMT_HEAD_WGRAD_HALF") == "1":
            nsplit = ops.wgrad_nsplit(Mh, nexp, hd.cin, self.dtype)
        else:
            with ops.wgrad_plan(256, 6):
                nsplit = ops.wgrad_nsplit(Mh, nexp, hd.cin, self.dtype)
        assert nsplit * nexp * hd.cin <= self._slab_cap
        slab = self.buf("wgrad.slab", self._slab_cap, dtype=torch.float32)
        wd = ops.make_wgrad_desc(G, hd.feat, slab, B=B, H=hd.h, W=hd.w, Cin=hd.cin, Ho=hd.h, Wo=hd.w, Cd=nexp, taps=[(0, 0)],
                                 stride=1, nsplit=nsplit, ldd=kexp)
        b.add_desc("simt_conv_wgrad", wd, tag="conv_wgrad<bf16>", flops=2.0 * Mh * hd.Q * nt * hd.cin,
                   nbytes=float((Mh * kexp + Mh * hd.cin) * 2 + nsplit * nexp * hd.cin * 4),
                   shape=f"M{Mh} Cd{nexp} K{hd.cin} taps1 split{nsplit} (tap-expanded head)", stream=1)
        row = 0
        for prefix, cout in hd.groups:
            for i in range(nd):
                gname = f"{prefix}.conv2d_list.{i}.weight"
                b.add("simt_wgrad_reduce_exp", slab.data_ptr(), self.grads[gname].data_ptr(), nsplit, nexp, hd.cin, QP, row,
                      9 * i, cout, 9, stream=1)
                self.grad_ready[gname] = len(b)
            row += cout
        wt = self.new(npad, kexp, zero=True)
        row = 0
        for prefix, cout in hd.groups:
            for i in range(nd):
                w = self.p[f"{prefix}.conv2d_list.{i}.weight"]
                self.pack_list.add("simt_pack_weight", w.data_ptr(), wt.data_ptr(), cout, hd.cin, 9, row, 9 * i, kexp, QP, 1,
                                   None, ops.dt_code(self.dtype))
            row += cout
        if getattr(hd, "mask", None) is not None:
            bnr = None
        dsc = self._conv(b, G, (wt, tile, npad), dfeat, Bn=B, Hi=hd.h, Wi=hd.w, Cin=kexp, Ho=hd.h, Wo=hd.w, Cout=hd.cin,
                         taps=[(0, 0)], res=dz_prev, alg_k=nt * hd.Q, mask=getattr(hd, "mask", None), bnr=bnr,
                         note=" (tap-expanded head)")
        self._head_bnr_nblk = self._fused_nblk(dsc, bnr)
        return dfeat

    # ------------------------------------------------------------------ run
    def capture_graphs(self):
        """hipGraphs of the forward and (train plans) backward launch lists; repack() stays eager."""
        self.fwd_list.capture()
        if self.train:
            self.bwd_list.capture()

    def forward(self, x_nchw=None):
        """x: [B,3,H,W] fp32 CUDA (BGR, mean-subtracted).  Returns {head name: logits [B,h,w,ldp] fp32 (NHWC)}."""
        if x_nchw is not None:
            self.x_in.copy_(x_nchw)
        self.fwd_list.run()
        return self.out

    def backward(self, hook=None):
        """Consumes self.dlogits[*] (conv dtype, K-padded); fills self.grads.  hook(n, event): called with the number of
        launch-list entries replayed so far at every point where another gradient tensor became final, and an event
        recorded on the side stream (where the weight gradients are produced) -- used by the DP bucket reducer."""
        if hook is None:
            self.bwd_list.run()
            return self.grads
        cuts = sorted(set(self.grad_ready.values()))
        items, i0 = self.bwd_list.items, 0
        seg = LaunchList()
        side = side_stream(self.dev)
        for c in cuts:
            seg.items = items[i0:c]
            seg.run()
            # the event is made only if the hook asks for it (the reducer does when a bucket leaves at this cut: 5 of 29 cuts in the production
            # plan -- round 5 recorded a system-scope torch event at every cut: ~6.5 us of idle side queue each), device scope like the
            # launch lists' own events: the consumer is the collective's kernel on THIS device
            made = []

            def make_ev():
                if not made:
                    e = _new_event()
                    e.record(side)
                    made.append(e)
                return made[0]
            hook(c, make_ev)
            i0 = c
        seg.items = items[i0:]
        seg.run()
        return self.grads
