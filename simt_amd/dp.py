"""Single-node data parallelism for the SimT iteration: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  The reference has no distributed code at all; semantics chosen: every rank runs
the reference's iteration on its own micro-batch (BN statistics, anchors, CE means are per rank), gradients of the conv
stack and of NTM1/NTM2 are AVERAGED, the W inner loop is replica-deterministic and needs no exchange.

`BucketReducer` all-reduces ONE flat fp32 gradient buffer in contiguous buckets.  TrunkPlan lays the gradients out in
the order backward produces them (heads of layer4, layer4.2 ... stem), so a bucket can be launched on a side HIP stream
as soon as the launch that completes its last tensor has been enqueued -- the exchange of layer4's 60 MB overlaps the
backward of layer3.  Buckets are sized for xGMI rings (7 links x ~153 GB/s per GPU, per-link bound): few, large
messages (default 32 MB) rather than NVSwitch-style small ones.
"""
import os
import time

import torch
import torch.distributed as dist


def make_buckets(order, sizes, ready, bucket_elems):
    """order: tensor names in flat-buffer order; sizes[name] elements; ready[name] = launch index after which the tensor
    is final.  Returns [(start, end, ready_index)] covering the flat buffer contiguously."""
    out, start, cur, rdy = [], 0, 0, 0
    for n in order:
        cur += sizes[n]
        rdy = max(rdy, ready.get(n, 0))
        if cur - start >= bucket_elems:
            out.append((start, cur, rdy))
            start = cur
    if cur > start:
        out.append((start, cur, rdy))
    # ready indices must be monotone for the in-order hook
    fixed, m = [], 0
    for s, e, r in out:
        m = max(m, r)
        fixed.append((s, e, m))
    return fixed


class BucketReducer:
    def __init__(self, flat, buckets, group=None, extra=()):
        """flat: 1-D fp32 tensor; buckets from make_buckets; extra: small tensors reduced at finish() (NTM grads)."""
        self.flat, self.buckets, self.group, self.extra = flat, buckets, group, list(extra)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a one-rank group exchanges nothing -- unless SIMT_DP_FORCE=1 (tests: the only way to run the collectives of a REAL RCCL group on a
        # 1-GPU box; a mean over one rank is the identity, bit for bit)
        self.single = self.world == 1 and not (os.environ.get("SIMT_DP_FORCE") == "1" and dist.is_initialized())
        self.cuda = flat.is_cuda
        # The collectives are issued with the plan's SIDE stream current (round 6; rounds 1-5 made a stream of their own): the side stream is
        # where the weight gradients are produced, it trails the main stream through the backward (so its wait for the main stream's position
        # at a release point is free), and every extra stream is one more contender for HIP's 4 hardware queues -- a "comm" stream that lands on
        # the main stream's queue stalls the main stream at each of its cross-stream waits (engine.reserve_streams).
        self.comm = None
        if self.cuda:
            from .engine import side_stream
            self.comm = side_stream(flat.device)
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self.avg = backend == "nccl"
        self.handles = []
        self.next = 0
        self.released = [None] * len(buckets)     # per bucket: the launch-list index the backward replay had reached when it was released
        self.total_launches = None                # set by the trainer: entries of the backward launch list (for reading `released`)
        # self-diagnosis of a multi-GPU run (bench.py "comm"): per finish() an event pair around the wait of the calling stream
        self.measure = False
        self._waits = []          # [(event before the wait, event after it, host seconds spent in handle.wait())]

    def bytes_per_step(self):
        """Bytes this rank hands to the all-reduce per step (payload, not wire traffic: a ring moves 2 (N-1)/N of it per GPU)."""
        return sum((e - s) for s, e, _ in self.buckets) * self.flat.element_size() + sum(t.numel() * t.element_size() for t in self.extra)

    def report(self):
        """-> dict for bench.py's "comm" object.  Synchronises.  exposed_wait = time the stream that called finish() (the side stream
        of the early optimiser step, or the main stream) sat waiting for the exchange: the part of the all-reduce NOT hidden under
        the backward.  SURVEY 8e: >= 6.5x at 8 GPUs needs this to stay under ~2 ms."""
        if self.cuda:
            torch.cuda.synchronize()
        ms = sorted((a.elapsed_time(b) if a is not None else 0.0) + h * 1e3 for a, b, h in self._waits)
        return {"bytes_per_step": int(self.bytes_per_step()), "buckets": len(self.buckets),
                "bucket_bytes": [int((e - s) * self.flat.element_size()) for s, e, _ in self.buckets],
                "extra_tensors": len(self.extra), "world": self.world, "op": "AVG" if self.avg else "SUM+div",
                # when each bucket left for the all-reduce in the LAST step: the backward launch-list index the replay had reached (-1: at
                # finish(), after the last hook) beside the index at which its last gradient becomes final, and the length of the list --
                # a low scaling number reads as "exposed wait" (above) or as "late release" (released >> ready) without a second run
                "bucket_ready_launch": [int(r) for _s, _e, r in self.buckets],
                "bucket_released_launch": [(-1 if (x is None or x >= (1 << 59)) else int(x)) for x in self.released],
                "backward_launches": self.total_launches,
                "exposed_wait_ms_median": round(ms[len(ms) // 2], 4) if ms else None,
                "exposed_wait_ms_max": round(ms[-1], 4) if ms else None, "steps_measured": len(ms)}

    def start(self):
        self.handles, self.next = [], 0
        self.released = [None] * len(self.buckets)

    def _reduce(self, t):
        if self.single:
            return
        if self.avg:
            self.handles.append(dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            self.handles.append((dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True), t))

    def ready_upto(self, launch_index, producer_event=None):
        """Called by the backward replay after `launch_index` launch-list entries have been enqueued.  producer_event:
        event on the stream that writes the gradients (the plan's side stream) covering those entries -- or a callable that
        records and returns one (TrunkPlan.backward: only called when a bucket really leaves at this point)."""
        if callable(producer_event) and not self.single and self.next < len(self.buckets) and self.buckets[self.next][2] <= launch_index:
            producer_event = producer_event()
        elif callable(producer_event):
            producer_event = None
        while self.next < len(self.buckets) and self.buckets[self.next][2] <= launch_index:
            s, e, _ = self.buckets[self.next]
            self.released[self.next] = launch_index
            self.next += 1
            if self.single:
                continue
            if self.cuda:
                # compute stream -> comm stream: device scope is enough (the consumer is the collective's kernel on this device; a system-scope
                # torch event idles the recording queue ~6.5 us: engine.DeviceEvent).  The event BEHIND the exchange (finish) stays system scope.
                from .engine import _new_event, wait_event
                ev_stream = torch.cuda.current_stream()
                ev = _new_event()
                ev.record(ev_stream)
                with torch.cuda.stream(self.comm):
                    if torch.cuda.current_stream() != ev_stream:
                        wait_event(self.comm, ev)
                    # (producer_event was recorded on the side stream itself: in-order, nothing to wait for)
                    self._reduce(self.flat[s:e])
            else:
                self._reduce(self.flat[s:e])

    def finish(self):
        """Flush remaining buckets + the extra tensors, then make the compute stream wait for the exchange."""
        self.ready_upto(1 << 60)
        if self.single:
            return
        if self.cuda:
            from .engine import _new_event, wait_event
            ev_stream = torch.cuda.current_stream()
            ev = _new_event()
            ev.record(ev_stream)
            with torch.cuda.stream(self.comm):
                if torch.cuda.current_stream() != ev_stream:
                    wait_event(self.comm, ev)
                for t in self.extra:
                    self._reduce(t)
        else:
            for t in self.extra:
                self._reduce(t)
        e0 = e1 = None
        if self.measure and self.cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
        t0 = time.perf_counter()
        for h in self.handles:
            if isinstance(h, tuple):
                h[0].wait()
                h[1].div_(self.world)
            else:
                h.wait()
        # exposed wait: on CUDA tensors the e0 -> e1 event pair already contains whatever the host blocked for between the two records
        # (gloo on device tensors), so the host timer counts only for CPU tensors -- never both (double counting)
        host = time.perf_counter() - t0 if not self.cuda else 0.0
        if self.cuda and torch.cuda.current_stream() != self.comm:      # (the early optimiser step calls finish() ON the side stream: nothing to join)
            done = torch.cuda.Event()
            done.record(self.comm)
            torch.cuda.current_stream().wait_event(done)
        if self.measure:
            if self.cuda:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record(torch.cuda.current_stream())
            self._waits.append((e0, e1, host))
            if len(self._waits) > 256:
                self._waits.pop(0)
