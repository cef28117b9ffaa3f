"""A/B of the whole SimT step under two settings of construction-time environment switches, in ONE process on ONE device:
one SimTTrainer per variant (the environment as it is, plus one per --env 'K=V[,K=V]' applied while it is built), timed in alternating rounds of
--steps iterations each (boxes of the pool differ by +-0.5 ms, rounds on one box by ~0.1 ms).  BASELINE configs[1] by default.

    python profiles/tools/ab_env.py --env SIMT_BN_GRID=0 --env SIMT_BN_GRID=2 --steps 20 --rounds 5
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from simt_amd import model_spec as ms                     # noqa: E402
from simt_amd.step import Hyper, SimTTrainer              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", action="append", default=[], help="one more variant: 'K=V[,K2=V2...]' applied while that trainer is constructed")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--size", type=int, nargs=2, default=[768, 768])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    K = 3
    cd = ms.load_class_dist("bapa")
    st = ms.reference_init(ms.state_shapes(19, K, True), seed=1234)
    fst = ms.reference_init(ms.state_shapes(19, 0, False), seed=1234)
    H, W = a.size

    pg = [None]

    def make(env):
        env = dict(env)
        dp = env.pop("DP", "0") == "1"          # pseudo-variable (round 6): the data-parallel trainer over a REAL one-rank RCCL group (SIMT_DP_FORCE=1)
        if dp and pg[0] is None:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29573")
            os.environ["SIMT_DP_FORCE"] = "1"
            torch.cuda.set_device(0)
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            pg[0] = dist.group.WORLD
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return SimTTrainer(st, fst, ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), Hyper(open_classes=K, lr=2.5e-4, lr_T=6e-3), cd,
                               a.batch, H, W, dtype=torch.bfloat16, device=dev, process_group=pg[0] if dp else None)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k)
                else:
                    os.environ[k] = v
    trs = {"default": make({})}
    for v in a.env:
        trs[v] = make(dict(kv.split("=", 1) for kv in v.split(",")))
    img, lab = ms.synthetic_batch(a.batch, H, W, cd, seed=1234, device=dev)
    for tr in trs.values():
        for it in range(5):
            tr.step(img, lab, it)
    torch.cuda.synchronize()
    res = {k: [] for k in trs}
    for r in range(a.rounds):
        for name, tr in trs.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(a.steps):
                tr.step(img, lab, 5 + r * a.steps + it)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
    for name, v in res.items():
        print(f"{name:40s} ms/step per round: " + " ".join(f"{x:7.3f}" for x in v) + f"   median {sorted(v)[len(v) // 2]:7.3f}")


if __name__ == "__main__":
    main()
