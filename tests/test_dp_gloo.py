"""Data-parallel plumbing on CPU: world_size 2, gloo backend (the GPU path uses the same class with backend nccl = RCCL).
Checks the bucket layout (contiguous cover, monotone readiness) and that BucketReducer leaves the MEAN of the ranks'
gradients in the flat buffer and in the extra (NTM) tensors, when buckets are released incrementally by the backward
hook exactly like TrunkPlan.backward(hook=...) does."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from simt_amd.dp import BucketReducer, make_buckets


def test_make_buckets_cover_and_order():
    order = [f"t{i}" for i in range(10)]
    sizes = {n: (i + 1) * 100 for i, n in enumerate(order)}
    ready = {n: 5 * (i + 1) if i != 3 else 2 for i, n in enumerate(order)}     # one out-of-order entry
    b = make_buckets(order, sizes, ready, bucket_elems=700)
    assert b[0][0] == 0 and b[-1][1] == sum(sizes.values())
    for (s0, e0, r0), (s1, e1, r1) in zip(b, b[1:]):
        assert e0 == s1 and r1 >= r0 and e0 > s0
    assert all(e - s >= 700 for s, e, _ in b[:-1])


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        order = [f"p{i}" for i in range(7)]
        sizes = dict(zip(order, [1000, 30, 5000, 7, 2048, 1, 999]))
        ready = {n: 10 * (i + 1) for i, n in enumerate(order)}
        total = sum(sizes.values())
        flat = torch.randn(total, generator=g)
        extra = torch.randn(2, 22, 19, generator=g)
        mine, mine_extra = flat.clone(), extra.clone()
        red = BucketReducer(flat, make_buckets(order, sizes, ready, bucket_elems=2000), group=dist.group.WORLD, extra=[extra])
        red.measure = True                             # bench.py's "comm" object comes from this report
        red.start()
        for launches in (5, 10, 30, 35, 70):          # the backward replay reports progress at its cut points
            red.ready_upto(launches)
        red.finish()
        rep = red.report()
        assert rep["bytes_per_step"] == (total + extra.numel()) * 4 and rep["buckets"] == len(red.buckets) and rep["world"] == world
        assert sum(rep["bucket_bytes"]) == total * 4 and rep["steps_measured"] == 1 and rep["exposed_wait_ms_median"] >= 0.0
        # expected: mean over ranks
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        ge = [torch.zeros_like(mine_extra) for _ in range(world)]
        dist.all_gather(ge, mine_extra)
        ok = (torch.allclose(flat, torch.stack(gathered).mean(0), atol=1e-6)
              and torch.allclose(extra, torch.stack(ge).mean(0), atol=1e-6))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_mean_world2():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]
