"""Data-parallel SimT iterations with TWO processes (both on cuda:0, gloo transport -- the only multi-process setup a
1-GPU box allows; the production backend is "nccl" = RCCL, same code path in simt_amd/dp.py).  Each rank trains on its
own micro-batch; after every step the replicas must hold IDENTICAL parameters / NTM / W (gradients averaged by the
bucket reducer, W loop replica-deterministic), and they must differ from a single-rank run on rank 0's data alone."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    import numpy as np
    from oracle import simt_oracle as so
    from simt_amd.step import Hyper, SimTTrainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        layers, K = (1, 1, 2, 1), 3
        cd = so.load_class_dist()
        st = so.recipe_state(so.state_shapes(19, K, True, layers=layers), seed=11, head_scale=8.0)
        fst = so.recipe_state(so.state_shapes(19, 0, False, layers=layers), seed=12, head_scale=8.0)
        hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
        tr = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, cd.numpy(), 2, 65, 65, dtype=torch.float32,
                         device=dev, layers=layers, process_group=dist.group.WORLD)
        os.environ["SIMT_EARLY_SGD"] = "0"        # same exchange with the optimiser step after the whole backward: bit-identical replicas
        late = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, cd.numpy(), 2, 65, 65, dtype=torch.float32,
                           device=dev, layers=layers, process_group=dist.group.WORLD)
        os.environ.pop("SIMT_EARLY_SGD")
        assert tr._early_sgd and not late._early_sgd
        solo = SimTTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, cd.numpy(), 2, 65, 65, dtype=torch.float32,
                           device=dev, layers=layers) if rank == 0 else None
        for it in range(2):
            img, lab = so.synthetic_batch(2, 65, 65, cd.numpy(), seed=100 + 10 * rank + it, block=8)
            tr.step(img.to(dev), lab.to(dev), it)
            late.step(img.to(dev), lab.to(dev), it)
            if solo is not None:
                solo.step(img.to(dev), lab.to(dev), it)
        torch.cuda.synchronize()
        vec = torch.cat([tr.params[k].flatten() for k in sorted(tr.params) if tr.params[k].dtype != torch.long and "running" not in k]
                        + [tr.ntm[0].flatten(), tr.ntm[1].flatten(), tr.wraw[0].flatten()]).cpu()
        lvec = torch.cat([late.params[k].flatten() for k in sorted(late.params) if late.params[k].dtype != torch.long and "running" not in k]
                         + [late.ntm[0].flatten(), late.ntm[1].flatten(), late.wraw[0].flatten()]).cpu()
        assert torch.equal(vec, lvec), "early (side-stream) and late optimiser step give different replicas"
        gathered = [torch.zeros_like(vec) for _ in range(world)]
        dist.all_gather(gathered, vec)
        same = all(torch.equal(gathered[0], g) for g in gathered[1:])
        differs = True
        if solo is not None:
            svec = torch.cat([solo.params[k].flatten() for k in sorted(solo.params) if solo.params[k].dtype != torch.long
                              and "running" not in k]).cpu()
            differs = not torch.equal(svec, vec[: svec.numel()])
        q.put((rank, bool(same), bool(differs), bool(torch.isfinite(vec).all())))
    finally:
        dist.destroy_process_group()


def test_two_rank_dp_replicas_stay_identical(dev):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res == [(0, True, True, True), (1, True, True, True)], res
