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
        made = []
        for launches in (5, 10, 30, 35, 70):          # the backward replay reports progress at its cut points
            before = red.next
            # (round 6) the producer event comes as a FACTORY: it is called only at the cut points where a bucket really leaves
            red.ready_upto(launches, lambda: made.append(launches))
            assert (made[-1:] == [launches]) == (red.next > before), (launches, made, before, red.next)
        assert 1 <= len(made) <= 5
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


# ----------------------------------------------------------------------------------------------------------------------
# DP SEMANTICS against the oracle (VERDICT r3 missing #2; north_star "DP scaling with loss parity to reference"): every rank runs
# trainV2_simt.py:308-432 on its micro-batch, the applied gradients (incl. NTM1 / NTM2 with the inner-loop leak) are averaged,
# every rank runs :434-436.  Here on CPU: two gloo ranks each hold an OracleTrainer, lay their applied gradients out in ONE flat
# buffer (like TrunkPlan.flat_grad), exchange it through the product's BucketReducer, step -- and must land on the parameters of
# `oracle_dp_step` (both ranks' oracle gradients averaged in one process).
# ----------------------------------------------------------------------------------------------------------------------
DP_LAYERS, DP_K = (1, 1, 2, 1), 3


def _dp_case():
    from oracle import simt_oracle as so
    cd = so.load_class_dist()
    st = so.recipe_state(so.state_shapes(19, DP_K, True, layers=DP_LAYERS), seed=11, head_scale=8.0)
    fst = so.recipe_state(so.state_shapes(19, 0, False, layers=DP_LAYERS), seed=12, head_scale=8.0)
    kw = dict(open_classes=DP_K, lambda_convex=0.1, lambda_volume=1.0, lambda_anchor=1.0, lr=6e-4, lr_T=6e-3)
    mk = lambda: so.OracleTrainer(st, fst, so.ntm_init(19, DP_K, 1), so.ntm_init(19, DP_K, 2), so.Hyper(**kw), cd, layers=DP_LAYERS)
    batches = [so.synthetic_batch(2, 65, 65, cd.numpy(), seed=100 + 10 * r, block=8) for r in range(2)]
    return mk, batches


def _dp_oracle_worker(rank, world, port, q):
    from oracle import simt_oracle as so
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mk, batches = _dp_case()
        orc = mk()
        out = orc.step(*batches[rank], 0, apply=False)
        grads = orc.applied_grads()
        names = [n for n in grads if not n.startswith("NTM")]
        sizes = {n: grads[n].numel() for n in names}
        flat = torch.cat([grads[n].flatten() for n in names])
        extra = torch.stack([grads["NTM1"], grads["NTM2"]]).clone()
        red = BucketReducer(flat, make_buckets(names, sizes, {n: i for i, n in enumerate(names)}, bucket_elems=1 << 16),
                            group=dist.group.WORLD, extra=[extra])
        red.start()
        red.ready_upto(len(names) // 2)
        red.finish()
        off = 0
        with torch.no_grad():
            for n in names:
                grads[n].copy_(flat[off:off + sizes[n]].view_as(grads[n]))
                off += sizes[n]
            grads["NTM1"].copy_(extra[0])
            grads["NTM2"].copy_(extra[1])
        orc.apply_update(0)
        res = {n: orc.st[n].detach().clone() for n in names}
        res["NTM1"], res["NTM2"] = orc.ntm[0].detach().clone(), orc.ntm[1].detach().clone()
        res["W1"] = orc.w[0].detach().clone()
        q.put((rank, {k: v.numpy() for k, v in res.items()}, {k: float(out[k].detach()) for k in ("total", "loss_y2", "anchor")}))
    finally:
        dist.destroy_process_group()


def test_dp_step_equals_oracle_on_mean_gradients_world2():
    from oracle import simt_oracle as so
    import numpy as np
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_oracle_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    mk, batches = _dp_case()
    reps = [mk(), mk()]
    solo = mk()
    outs = so.oracle_dp_step(reps, [b[0] for b in batches], [b[1] for b in batches], 0)
    solo.step(*batches[0], 0)
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, params, losses in res:
        for k, v in losses.items():                    # per-rank losses are the rank's own micro-batch's
            assert abs(v - float(outs[rank][k].detach())) <= 1e-6 * (1 + abs(v)), (rank, k)
        for n, v in params.items():
            ref = {"NTM1": reps[rank].ntm[0], "NTM2": reps[rank].ntm[1], "W1": reps[rank].w[0]}.get(n)
            ref = (ref if ref is not None else reps[rank].st[n]).detach().numpy()
            assert np.allclose(v, ref, rtol=1e-6, atol=1e-7), (rank, n, np.abs(v - ref).max())
    # the replicas agree with each other and DIFFER from a single-rank step on rank 0's micro-batch (the mean is not a no-op)
    for n in res[0][1]:
        assert np.array_equal(res[0][1][n], res[1][1][n]), n
    n = "layer6.conv2d_list.0.weight"
    assert not np.allclose(res[0][1][n], solo.st[n].detach().numpy(), rtol=1e-6, atol=1e-9)
    # the NTM gradient's inner-loop leak (quirk 3, trainV2_simt.py:326-339) is rank-identical: the exchanged mean must carry it ONCE
    # (a SUM over the ranks would carry it twice): mean - leak == mean over ranks of the loss-only part, and the leak is not small
    leak = mk()
    so.inner_w_loop(leak.ntm[0], leak.ntm[1], leak.w[0], leak.w[1], leak.wstate, leak.cd, leak.hp,
                    so.lr_poly(6e-3, 0, leak.hp.num_steps, leak.hp.power))
    singles = [mk(), mk()]
    for r, b in zip(singles, batches):
        r.step(*b, 0, apply=False)
    for k in range(2):
        lk = leak.ntm[k].grad
        loss_only = sum(r.ntm[k].grad - lk for r in singles) / 2
        assert lk.abs().max() > 1e-3
        assert torch.allclose(reps[0].ntm[k].grad, lk + loss_only, rtol=1e-5, atol=1e-7)


# ----------------------------------------------------------------------------------------------------------------------
# World size 8 (BASELINE configs[2]: 8 x MI355X, global batch 32) on the REAL exchange table of the production plan
# (tests/golden/g16_dp_bucket_table.json, written on a GPU box by profiles/tools/dump_bucket_table.py; tests/test_gpu_dp.py checks
# that the live plan still produces it): 96 gradient tensors, 42.2 M fp32 = 168.7 MB, five buckets of 33 / 32 / 33 / 34 / 28 MB that
# become ready at backward launches 37 / 69 / 105 / 159 / 191 of 263.  Eight gloo ranks release the buckets at the replay's real hook
# points, exactly like TrunkPlan.backward(hook=...) with the early optimiser step does, and must end with the mean.
# ----------------------------------------------------------------------------------------------------------------------
def _real_table():
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g16_dp_bucket_table.json")) as f:
        return json.load(f)


def test_real_bucket_table_shape():
    t = _real_table()
    sizes, ready = dict(zip(t["order"], t["sizes"])), dict(zip(t["order"], t["ready"]))
    b = make_buckets(t["order"], sizes, ready, bucket_elems=t["bucket_elems"])
    assert [list(x) for x in b] == t["buckets"]                                   # the committed cut is what make_buckets produces
    assert b[0][0] == 0 and b[-1][1] == t["end"] == sum(t["sizes"]) and t["end"] <= t["flat_elems"]
    assert all(e0 == s1 for (_s0, e0, _r0), (s1, _e1, _r1) in zip(b, b[1:]))      # contiguous cover of the applied prefix
    assert [r for _s, _e, r in b] == sorted(r for _s, _e, r in b)                 # monotone readiness
    assert all(e - s >= t["bucket_elems"] for s, e, _ in b[:-1]) and 0 < b[-1][1] - b[-1][0] < t["bucket_elems"]      # uneven last bucket
    assert all(s % 4 == 0 for s, _e, _r in b)                                     # 16-byte aligned starts (float4 reduce kernels write them)
    # every bucket is final strictly before the early optimiser step's cut, which itself lies inside the backward list: the whole exchange
    # can overlap the rest of the backward (layer 2 / layer 1 / stem, whose gradients stay rank-local)
    assert all(r in t["hook_points"] for _s, _e, r in b) and b[-1][2] <= t["early_cut"] < t["backward_launches"]
    assert 160e6 < t["end"] * 4 < 180e6


def _world8_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        t = _real_table()
        sizes, ready = dict(zip(t["order"], t["sizes"])), dict(zip(t["order"], t["ready"]))
        n = t["end"]
        # rank r holds (r + 1) * g(i) + r: the mean over 8 ranks is 4.5 * g(i) + 3.5 in closed form (no 1.3 GB all-gather to check it)
        g = ((torch.arange(n, dtype=torch.float32) % 1021.0) - 510.0) / 64.0
        flat = g * float(rank + 1) + float(rank)
        # the product's layout (step.SimTTrainer._xchg[12:]): [bad-label count, 3 unused slots | dNTM1 | dNTM2] -- ONE extra collective
        xchg = torch.zeros(4 + 2 * 22 * 19)
        bad, ntm = xchg[0:1], xchg[4:].view(2, 22, 19)
        ntm.fill_(float(rank + 1))
        bad.fill_(3.0 if rank == 5 else 0.0)                 # rank 5 saw three bad labels this step (lout[12])
        red = BucketReducer(flat, make_buckets(t["order"], sizes, ready, bucket_elems=t["bucket_elems"]), group=dist.group.WORLD,
                            extra=[xchg])
        red.total_launches = t["backward_launches"]
        red.measure = True
        red.start()
        released_at = []
        for hp in t["hook_points"]:                    # the replay's cut points, in order; the early optimiser step fires at early_cut
            before = red.next
            red.ready_upto(hp)
            released_at += [hp] * (red.next - before)
            if hp >= t["early_cut"]:
                break
        assert red.next == len(red.buckets), "a bucket would only leave at finish()"
        red.finish()
        rep = red.report()
        ok = (torch.allclose(flat, g * 4.5 + 3.5, rtol=1e-6, atol=1e-5) and torch.allclose(ntm, torch.full_like(ntm, 4.5))
              and abs(float(bad) * world - 3.0) < 1e-5                       # losses(): total = mean * world on EVERY rank, no collective
              and rep["bucket_released_launch"] == released_at == rep["bucket_ready_launch"]      # released at the very hook that made them final
              and rep["backward_launches"] == t["backward_launches"] and rep["buckets"] == 5 and rep["world"] == 8
              and rep["bytes_per_step"] == (n + ntm.numel() + 4) * 4 and rep["extra_tensors"] == 1)
        q.put((rank, bool(ok), released_at))
    finally:
        dist.destroy_process_group()


def test_bucket_reducer_world8_real_table():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert [(r, ok) for r, ok, _ in res] == [(r, True) for r in range(8)]
    assert all(rel == res[0][2] for _r, _ok, rel in res)


# oracle_dp_step at world size 8: eight micro-batches, one mean (toy trunk; the full-size exchange is the test above)
def _dp8_worker(rank, world, port, q):
    from oracle import simt_oracle as so
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mk, _ = _dp_case()
        cd = so.load_class_dist()
        batch = so.synthetic_batch(1, 65, 65, cd.numpy(), seed=300 + 7 * rank, block=8)
        orc = mk()
        orc.step(*batch, 0, apply=False)
        grads = orc.applied_grads()
        names = [n for n in grads if not n.startswith("NTM")]
        sizes = {n: grads[n].numel() for n in names}
        flat = torch.cat([grads[n].flatten() for n in names])
        extra = torch.stack([grads["NTM1"], grads["NTM2"]]).clone()
        red = BucketReducer(flat, make_buckets(names, sizes, {n: i for i, n in enumerate(names)}, bucket_elems=1 << 15),
                            group=dist.group.WORLD, extra=[extra])
        red.start()
        for k in range(0, len(names), 3):
            red.ready_upto(k)
        red.finish()
        off = 0
        with torch.no_grad():
            for n in names:
                grads[n].copy_(flat[off:off + sizes[n]].view_as(grads[n]))
                off += sizes[n]
            grads["NTM1"].copy_(extra[0])
            grads["NTM2"].copy_(extra[1])
        orc.apply_update(0)
        keep = ("layer6.conv2d_list.0.weight", "layer5.conv2d_list.1.bias", "layer3.0.conv1.weight", "layer4.0.conv2.weight")
        res = {n: orc.st[n].detach().clone().numpy() for n in keep}
        res["NTM1"], res["NTM2"] = orc.ntm[0].detach().clone().numpy(), orc.ntm[1].detach().clone().numpy()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_dp_step_equals_oracle_on_mean_gradients_world8():
    from oracle import simt_oracle as so
    import numpy as np
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    mk, _ = _dp_case()
    cd = so.load_class_dist()
    batches = [so.synthetic_batch(1, 65, 65, cd.numpy(), seed=300 + 7 * r, block=8) for r in range(8)]
    reps = [mk() for _ in range(8)]
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)            # like the workers: fp32 CPU convs differ in the last bits between thread counts
    try:
        so.oracle_dp_step(reps, [b[0] for b in batches], [b[1] for b in batches], 0)
    finally:
        torch.set_num_threads(nthr)
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, params in res:
        for n, v in params.items():
            ref = {"NTM1": reps[rank].ntm[0], "NTM2": reps[rank].ntm[1]}.get(n)
            ref = (ref if ref is not None else reps[rank].st[n]).detach().numpy()
            assert np.allclose(v, ref, rtol=2e-6, atol=2e-7), (rank, n, np.abs(v - ref).max())
        for n in params:                               # replicas bit-identical
            assert np.array_equal(params[n], res[0][1][n]), n
