"""The RCCL footprint of the data-parallel step, emulated on ONE GPU.  MEASUREMENT ONLY.

    for p in default first; do python profiles/tools/dp_emulate.py --plan $p; done > profiles/r06_dp_emulation.txt

What it runs: the production data-parallel trainer (SimTTrainer with a REAL one-rank RCCL process group, SIMT_DP_FORCE=1: bucketed exchange on
the reducer's comm stream, released by the backward replay where `bucket_released_launch` says, early optimiser step waiting for it) at BASELINE
configs[1] (B = 4, 768 x 768, bf16, K = 3).  A one-rank all-reduce is a copy: no ring kernel, no CU held.  So each bucket's collective is
FOLLOWED, on the same comm stream, by an "occupier" launch (profiles/tools/occupier.hip): N workgroups of T threads with L bytes of LDS that stay
resident, asleep, for bucket_bytes / rate microseconds -- the time an 8-GPU ring all-reduce of that bucket would hold N channels' CUs at `rate`
GB/s per GPU (168.7 MB per step at 150-300 GB/s = 0.6-1.1 ms of all-reduce per step).  The optimiser step waits for the occupier exactly as it
would for the wire.

What it varies: occupier count N in {0, 8, 16, 32} x the assumed rate x the occupier's LDS (0: its waves can share a CU with a conv workgroup;
64 KB: they cannot -- a conv workgroup holds 156 of the CU's 160 KB), for ONE tile plan per process (--plan default: 236 tiles of 160 rows, the
round-6 default on every GPU count; --plan first: SIMT_PICK_ROWS_FIRST=1, rounds 1-5's 255 tiles of 148 rows; --bn-grid 0: two-pass
BatchNorm).  One trainer per process ON PURPOSE: the first version of this tool kept three trainers (three comm streams) alive in one process
and every number in it was 2.7 ms too high -- HIP maps streams onto 4 hardware queues, and the fifth stream shared a queue with the main stream
(a sleeping occupier on it serialised with the step: +0.9 x the all-reduce time).  The production job has main, side, comm and RCCL's own stream."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--plan", default="default", choices=["default", "first"])
    ap.add_argument("--bn-grid", default=None)
    ap.add_argument("--pg-first", action="store_true", help="create the process group before the plan's streams (rounds 1-5's order)")
    ap.add_argument("--occ", type=int, nargs="+", default=[0, 8, 16, 32])
    ap.add_argument("--rates", type=float, nargs="+", default=[150.0, 300.0])
    ap.add_argument("--lds", type=int, nargs="+", default=[0, 65536])
    ap.add_argument("--threads", type=int, default=256)
    a = ap.parse_args()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    os.environ["SIMT_DP_FORCE"] = "1"
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    if not a.pg_first:
        # the plan's streams get their hardware queues BEFORE the process group creates its streams (HIP hands out 4 hardware queues in order
        # of first use; a side stream that ends up sharing the main stream's queue loses the two-stream overlap: +2 ms per step).  This is
        # what bench.py and the tools do since round 6; --pg-first shows the old order
        from simt_amd.engine import reserve_streams
        reserve_streams(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from simt_amd import model_spec as ms
    from simt_amd.step import Hyper, SimTTrainer
    occ = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboccupier.so"))
    occ.occ_launch.restype = C.c_int
    occ.occ_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    B, H, W, K = 4, 768, 768, 3
    cd = ms.load_class_dist("bapa")
    hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    cfg = {"n": 0, "rate": 300.0, "lds": 0}
    if a.plan == "first":
        os.environ["SIMT_PICK_ROWS_FIRST"] = "1"                # read by engine.TrunkPlan at construction (cu_budget = -1)
    if a.bn_grid is not None:
        os.environ["SIMT_BN_GRID"] = a.bn_grid
    tr = SimTTrainer(ms.reference_init(ms.state_shapes(19, K, True), seed=1234), ms.reference_init(ms.state_shapes(19, 0, False), seed=1234),
                     ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), hp, cd, B, H, W, dtype=torch.bfloat16, device=dev, process_group=dist.group.WORLD)
    red = tr.reducer
    assert red is not None and not red.single
    orig = red._reduce

    occ_stream = torch.cuda.Stream(device=dev)      # stands in for RCCL's own stream: the collective's kernels run THERE, behind the issuing stream

    def patched(t):
        orig(t)                           # the real (one-rank) collective: stream semantics, nothing on the wire
        if cfg["n"] > 0 and t.numel() > 4096:
            us = t.numel() * t.element_size() / (cfg["rate"] * 1e3)          # bytes / (GB/s) -> us
            occ_stream.wait_stream(torch.cuda.current_stream())
            assert occ.occ_launch(cfg["n"], a.threads, cfg["lds"], us, occ_stream.cuda_stream) == 0
    red._reduce = patched
    orig_finish = red.finish

    def finish():
        orig_finish()
        if cfg["n"] > 0:
            torch.cuda.current_stream().wait_stream(occ_stream)             # the optimiser step waits for the "wire" like handle.wait() does
    red.finish = finish
    label = f"plan {a.plan}" + (f", SIMT_BN_GRID={a.bn_grid}" if a.bn_grid is not None else "")
    trainers = {label: tr}
    a.budgets = [label]
    img, lab = ms.synthetic_batch(B, H, W, cd, seed=1234, device=dev)
    wide = {b: sorted({int(x) for x in _wide_tiles(tr)}) for b, tr in trainers.items()}
    print(f"# torch {torch.__version__}; B={B} {H}x{W} bf16 K={K}; data-parallel trainer over a one-rank RCCL group + occupier; "
          f"{a.rounds} alternating rounds x {a.steps} steps, median per-step HIP-event ms")
    print(f"# exchange per step: {trainers[a.budgets[0]].reducer.bytes_per_step() / 1e6:.1f} MB in {len(trainers[a.budgets[0]].reducer.buckets)} buckets; "
          f"pixel tiles of the wide (256-column, M = 37 636) convs: {wide}; fused BatchNorm launches: {tr.plan.fbn_launches}")
    for t_ in trainers.values():
        for _ in range(5):
            t_.step(img, lab)
    torch.cuda.synchronize()
    res = {}
    combos = [(0, a.rates[0], 0)] + [(n, r, l) for l in a.lds for r in a.rates for n in a.occ if n > 0]
    for rnd in range(a.rounds):
        for (n, rate, lds) in combos:
            for b, tr in trainers.items():
                cfg.update(n=n, rate=rate, lds=lds)
                for _ in range(2):
                    tr.step(img, lab)
                torch.cuda.synchronize()
                evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
                evs[0].record()
                for i in range(a.steps):
                    tr.step(img, lab)
                    evs[i + 1].record()
                torch.cuda.synchronize()
                res.setdefault((n, rate, lds, b), []).extend(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
    print(f"# {'occupiers':>9s} {'rate GB/s':>9s} {'all-reduce ms':>13s} {'LDS/wg':>7s} | " + " | ".join(f"{b}: ms (vs N=0)" for b in a.budgets))
    base = {b: float(np.median(res[(0, a.rates[0], 0, b)])) for b in a.budgets}
    tot = trainers[a.budgets[0]].reducer.bytes_per_step()
    for (n, rate, lds) in combos:
        cells = []
        for b in a.budgets:
            m = float(np.median(res[(n, rate, lds, b)]))
            cells.append(f"{m:19.3f} ({m - base[b]:+.3f})")
        ar = 0.0 if n == 0 else tot / (rate * 1e6)
        print(f"  {n:9d} {rate if n else 0:9.0f} {ar:13.2f} {lds // 1024:5d}KB | " + " | ".join(cells))
    dist.destroy_process_group()


def _wide_tiles(tr):
    from simt_amd import _lib as L
    out = set()
    for lst in (tr.plan.fwd_list, tr.plan.bwd_list):
        for it in lst.items:
            if it.tag and it.tag.startswith("conv_igemm2_kernel<256, 5, 3,") and "N256" in (it.shape or ""):
                out.add(L.load().simt_conv_mtiles(C.byref(it.keep)))
    return out


if __name__ == "__main__":
    main()
