"""Decoded uint8 frames -> network input on the device, and the loader that keeps the GPU fed.

Reference (CPU, per item, inside 4 DataLoader workers): dataset/cityscapes_dataset.py:97-120 `cityscapesPseudo.__getitem__`, :47-63
`cityscapesDataSet.__getitem__`; tools/trainV2_simt.py:287-294 DataLoader(shuffle=True, pin_memory=True), :345-348 `.cuda()`.
Here only file IO + PNG decoding stay on host threads; resize (Pillow-exact, csrc/input_prep.hip), BGR - mean, CHW and the label's
int64 conversion run on the GPU, and uploads go through pinned double buffers on a copy stream so that batch i+1 crosses PCIe
while batch i trains.  PyTorch supplies memory and streams only.
"""
import ctypes as C
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from .. import _lib as L
from . import resample as rs

IMG_MEAN = (104.00698793, 116.66876762, 122.67891434)      # tools/trainV2_simt.py:34 (BGR)


def _f32(v):
    return float(np.float32(v))


class InputPrep:
    """Device transform of a batch of decoded frames of ONE source geometry: [B, Hs, Ws, 3] u8 RGB (+ [B, Hs, Ws] u8 labels) ->
    [B, 3, h, w] fp32 (BGR - mean) (+ [B, h, w] int64).  crop = (w, h) like the reference's `crop_size` / --input-size-target."""

    def __init__(self, B, src_hw, crop_wh, device, mean=IMG_MEAN, with_label=True):
        self.B, (self.Hs, self.Ws), (self.w, self.h) = B, src_hw, crop_wh
        self.dev = torch.device(device)
        self.mean = tuple(_f32(m) for m in mean)
        self.with_label = with_label
        dev = self.dev
        self.need_x, self.need_y = self.w != self.Ws, self.h != self.Hs
        if self.need_x:
            self.kx, bx, cx = rs.bicubic_tables(self.Ws, self.w)
            self.bx, self.cx = torch.from_numpy(bx).to(dev), torch.from_numpy(cx).to(dev)
            self.tmp_x = torch.empty(B, self.Hs, self.w, 3, dtype=torch.uint8, device=dev)
        if self.need_y:
            self.ky, by, cy = rs.bicubic_tables(self.Hs, self.h)
            self.by, self.cy = torch.from_numpy(by).to(dev), torch.from_numpy(cy).to(dev)
            self.tmp_y = torch.empty(B, self.h, self.w, 3, dtype=torch.uint8, device=dev)
        if with_label:
            self.xtab = torch.from_numpy(rs.nearest_table(self.Ws, self.w)).to(dev)
            self.ytab = torch.from_numpy(rs.nearest_table(self.Hs, self.h)).to(dev)

    def run(self, rgb, x_out, lab=None, lab_out=None, mirror=False, stream=None):
        """rgb [B,Hs,Ws,3] u8, x_out [B,3,h,w] f32, lab [B,Hs,Ws] u8, lab_out [B,h,w] i64 -- all on the device; enqueues on `stream`
        (default: the current stream) and returns without synchronising.  mirror: bool or one bool per item -- the reference's
        --random-mirror branch, reproduced AS WRITTEN (cityscapes_dataset.py:108-111): the image's CHANNEL axis is reversed (net effect RGB
        order), only the label is mirrored horizontally."""
        st = stream if stream is not None else torch.cuda.current_stream(self.dev).cuda_stream
        assert rgb.is_cuda and rgb.dtype == torch.uint8 and tuple(rgb.shape) == (self.B, self.Hs, self.Ws, 3) and rgb.is_contiguous()
        assert x_out.dtype == torch.float32 and tuple(x_out.shape) == (self.B, 3, self.h, self.w) and x_out.is_contiguous()
        cur = rgb
        if self.need_x:
            L.call("simt_resample_u8", cur.data_ptr(), self.tmp_x.data_ptr(), self.B, self.Hs, self.Ws, 3, self.w, 1,
                   self.bx.data_ptr(), self.cx.data_ptr(), self.kx, st)
            cur = self.tmp_x
        if self.need_y:
            L.call("simt_resample_u8", cur.data_ptr(), self.tmp_y.data_ptr(), self.B, self.Hs, self.w, 3, self.h, 0,
                   self.by.data_ptr(), self.cy.data_ptr(), self.ky, st)
            cur = self.tmp_y
        if lab is not None:
            assert self.with_label and lab.dtype == torch.uint8 and tuple(lab.shape) == (self.B, self.Hs, self.Ws) and lab.is_contiguous()
            assert lab_out.dtype == torch.int64 and tuple(lab_out.shape) == (self.B, self.h, self.w) and lab_out.is_contiguous()
        flags = [bool(m) for m in mirror] if isinstance(mirror, (list, tuple, np.ndarray)) else [bool(mirror)] * self.B
        assert len(flags) == self.B
        # items with the same flag that sit next to each other go out in one launch (the usual case: one launch for the batch)
        b0 = 0
        while b0 < self.B:
            b1 = b0 + 1
            while b1 < self.B and flags[b1] == flags[b0]:
                b1 += 1
            n, f = b1 - b0, 1 if flags[b0] else 0
            L.call("simt_image_to_input", cur[b0].data_ptr(), x_out[b0].data_ptr(), n, self.h, self.w, self.mean[0], self.mean[1],
                   self.mean[2], f, st)
            if lab is not None:
                L.call("simt_label_nearest", lab[b0].data_ptr(), lab_out[b0].data_ptr(), n, self.Hs, self.Ws, self.h, self.w,
                       self.ytab.data_ptr(), self.xtab.data_ptr(), f, st)
            b0 = b1


class DevicePrefetcher:
    """Pinned double-buffered upload + device transform, one batch ahead of the consumer.

    source: iterator of (rgb u8 [B,Hs,Ws,3], label u8 [B,Hs,Ws] or None, meta) host arrays / tensors (numpy or torch; pinned
    tensors are uploaded in place, anything else is staged through this object's pinned buffers).
    Each __next__ returns (image f32 [B,3,h,w], label i64 [B,h,w] | None, meta) resident in HBM.

    Slot life time: the consumer may HOLD `hold` batches at once (gradient accumulation pulls `iter_size` micro-batches before the step
    that reads them is enqueued: tools/trainV2_simt.py `mb = [next(data) for _ in range(iter_size)]`).  The tensors returned by call k
    stay valid until call k + hold: only then is the slot's `free` event recorded on the consumer's stream (everything enqueued on
    that stream up to that point -- the step that consumed batch k included -- precedes the refill), and the copy stream waits for
    it.  2*hold slots, so the next group of `hold` batches is uploaded while the current one is being consumed."""

    def __init__(self, source, prep, mirror_fn=None, hold=1, depth=None):
        self.src, self.prep, self.mirror_fn = iter(source), prep, mirror_fn
        self.hold = max(1, int(hold))
        depth = 2 * self.hold if depth is None else depth
        assert depth > self.hold, "the consumer holds `hold` slots: at least one more is needed to hand out"
        dev, B = prep.dev, prep.B
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.slots = []
        for _ in range(depth):
            s = {"rgb_h": torch.empty(B, prep.Hs, prep.Ws, 3, dtype=torch.uint8).pin_memory(),
                 "rgb_d": torch.empty(B, prep.Hs, prep.Ws, 3, dtype=torch.uint8, device=dev),
                 "x": torch.empty(B, 3, prep.h, prep.w, dtype=torch.float32, device=dev),
                 "ready": torch.cuda.Event(), "free": None, "meta": None, "has_lab": False}
            if prep.with_label:
                s["lab_h"] = torch.empty(B, prep.Hs, prep.Ws, dtype=torch.uint8).pin_memory()
                s["lab_d"] = torch.empty(B, prep.Hs, prep.Ws, dtype=torch.uint8, device=dev)
                s["lab"] = torch.empty(B, prep.h, prep.w, dtype=torch.int64, device=dev)
            self.slots.append(s)
        self.head = 0          # next slot to hand out
        self.filled = 0
        self.calls = 0
        self.done = False
        for i in range(depth):
            self._fill(i)

    @staticmethod
    def _host(t):
        return torch.from_numpy(np.ascontiguousarray(t)) if isinstance(t, np.ndarray) else t

    def _fill(self, i):
        if self.done:
            return
        try:
            rgb, lab, meta = next(self.src)
        except StopIteration:
            self.done = True
            return
        s = self.slots[i]
        if s.get("used"):
            s["ready"].synchronize()           # the previous upload out of this slot's pinned buffers has completed
        rgb = self._host(rgb)
        if not rgb.is_pinned():
            s["rgb_h"].copy_(rgb)
            rgb = s["rgb_h"]
        if lab is not None:
            lab = self._host(lab)
            if not lab.is_pinned():
                s["lab_h"].copy_(lab)
                lab = s["lab_h"]
        cs = self.copy_stream
        if s["free"] is not None:
            cs.wait_event(s["free"])           # the consumer's kernels that read this slot's outputs have been enqueued and finished
        with torch.cuda.stream(cs):
            s["rgb_d"].copy_(rgb, non_blocking=True)
            if lab is not None:
                s["lab_d"].copy_(lab, non_blocking=True)
            mirror = self.mirror_fn(self.prep.B) if self.mirror_fn is not None else False
            self.prep.run(s["rgb_d"], s["x"], s["lab_d"] if lab is not None else None, s["lab"] if lab is not None else None,
                          mirror=mirror, stream=cs.cuda_stream)
            s["ready"].record(cs)
        s["meta"], s["has_lab"], s["used"] = meta, lab is not None, True
        self.filled += 1

    def __iter__(self):
        return self

    def __next__(self):
        if self.filled == 0:
            raise StopIteration
        i = self.head
        s = self.slots[i]
        cur = torch.cuda.current_stream(self.prep.dev)
        cur.wait_event(s["ready"])
        out = (s["x"], s["lab"] if s["has_lab"] else None, s["meta"])
        self.filled -= 1
        self.calls += 1
        self.head = (i + 1) % len(self.slots)
        # release + refill the slot handed out `hold` calls ago: the consumer no longer holds it, and everything it enqueued on this
        # stream so far (the step that read it) precedes the event the copy stream will wait for
        prev = (i - self.hold) % len(self.slots)
        if self.calls > self.hold and self._handed(prev):
            ev = torch.cuda.Event()
            ev.record(cur)
            self.slots[prev]["free"] = ev
            self._fill(prev)
        return out

    def _handed(self, j):
        # slot j was handed out and not refilled yet  <=>  it is not among the `filled` slots starting at head
        n = len(self.slots)
        return all(((self.head + k) % n) != j for k in range(self.filled))


class GpuLoader:
    """DataLoader(dataset, batch_size, shuffle, num_workers, pin_memory=True) of tools/trainV2_simt.py:287-294, feeding the GPU:
    `num_workers` host threads read + decode PNGs (Pillow releases the GIL while decoding), batches of decoded frames are uploaded
    and transformed by DevicePrefetcher.  Yields (images f32 [B,3,h,w], labels i64 [B,h,w] | None, sizes, names) like the
    reference's batches (`images, labels, _, _ = batch`), already on the device.  Incomplete last batches are dropped (the
    reference repeats the list to max_iters, so it never sees one)."""

    def __init__(self, dataset, batch_size, shuffle=True, num_workers=4, device="cuda:0", seed=1234, rank=0, world=1, epochs=None,
                 hold=1):
        self.ds, self.B, self.shuffle, self.workers = dataset, batch_size, shuffle, max(1, num_workers)
        self.hold = hold            # batches the consumer keeps at once (= --iter-size): see DevicePrefetcher
        self.dev, self.seed, self.rank, self.world, self.epochs = torch.device(device), seed, rank, world, epochs
        self._prep = None
        self._rng = np.random.default_rng(seed + 7919 * rank)
        self._lock = threading.Lock()

    def _order(self, epoch):
        n = len(self.ds)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + epoch)
            idx = torch.randperm(n, generator=g).tolist()
        else:
            idx = list(range(n))
        return idx[self.rank::self.world]               # data parallel: disjoint strided shards of one global order

    def _host_batches(self):
        epoch = 0
        with ThreadPoolExecutor(self.workers) as pool:
            while self.epochs is None or epoch < self.epochs:
                idx = self._order(epoch)
                nb = len(idx) // self.B
                pending = [pool.map(self.ds.decode, idx[b * self.B:(b + 1) * self.B]) for b in range(min(2, nb))]
                for b in range(nb):
                    items = list(pending.pop(0))
                    if b + 2 < nb:
                        pending.append(pool.map(self.ds.decode, idx[(b + 2) * self.B:(b + 3) * self.B]))
                    rgb = np.stack([it[0] for it in items])
                    lab = np.stack([it[1] for it in items]) if items[0][1] is not None else None
                    sizes = np.stack([np.array([self.ds.crop_size[1], self.ds.crop_size[0], 3]) for _ in items])
                    yield rgb, lab, (sizes, [it[2] for it in items])
                epoch += 1

    def __iter__(self):
        first = None
        gen = self._host_batches()
        try:
            first = next(gen)
        except StopIteration:
            return iter(())
        Hs, Ws = first[0].shape[1:3]
        self._prep = InputPrep(self.B, (Hs, Ws), tuple(self.ds.crop_size), self.dev, mean=self.ds.mean, with_label=first[1] is not None)

        def chain():
            yield first
            yield from gen
        # `flip = np.random.choice(2) * 2 - 1` per item (cityscapes_dataset.py:109)
        mirror_fn = (lambda n: (self._rng.integers(0, 2, n) == 0).tolist()) if getattr(self.ds, "is_mirror", False) else None
        pf = DevicePrefetcher(chain(), self._prep, mirror_fn=mirror_fn, hold=self.hold)
        return ((x, lab, meta[0], meta[1]) for (x, lab, meta) in pf)
