#!/usr/bin/env python3
"""Throughput benchmark of the SimT training iteration on MI355X (BASELINE.json metric: training images/sec at
768x768, DeepLabv2-R101+SimT).

  python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment: this process spawns `python -m torch.distributed.run --nproc-per-node N ...
bench.py` BEFORE touching the GPU and passes the child's output through (one rank per GPU over RCCL).  Launched by
torch.distributed.run directly (RANK / WORLD_SIZE set) it is one rank of that job; WORLD_SIZE must then equal --gpus.

A step = one full SimT iteration (tools/trainV2_simt.py:308-436 of the reference) on one synthetic batch per GPU:
10-step W inner loop, frozen-model forward, trainable forward (train-mode BN), fused head losses, backward (dgrad +
wgrad of all 104 trunk convs and the heads), gradient all-reduce (N>1), SGD with duplicate listings + Adam on NTM,
weight re-packing.  `value`: inputs resident in HBM when the timed region starts (the contract of this benchmark); the
PCIe-inclusive rate (uint8 frames uploaded from pinned host memory every step, converted on the device) is reported beside it
as "h2d_inclusive".  Rank 0 prints ONE JSON line.

Extra objects: "roofline" (dominant kernel class = the implicit-GEMM conv, algorithmic FLOPs / HIP-event time measured
live in a separate, untimed replay), "cpu_baseline" (the CPU oracle -- a port of the reference -- timed on the host cores on a
bounded sample; rank 0, N=1 only), "trained_like_pass" (same workload with checkpoint-like weights so that both confidence
thresholds are live), "h2d_inclusive".
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}      # dense, /opt/skills/guides/MI355X_MICROARCH.md
FLOP_PER_IMAGE_768 = 3.33e12                            # conv MACs x2, fixed fwd + fwd + bwd
PMC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json")          # rocprofv3 PMC summaries (profiles/collect.sh), stamped with the sha256 of the library they
PMC_MFMA_FILES = ("r06_pmc_mfma.json", "r05_pmc_mfma.json", "r04_pmc_mfma.json")        # were collected on: used only when that is the library this run has loaded


def pmc_lookup(files, kernel, key):
    """-> (value, source, reason).  The counters of `kernel` from the newest committed PMC summary -- only if it was collected on the very
    library loaded now (nothing used to tie the committed JSON to the measured build)."""
    import hashlib
    from simt_amd import _lib
    sha = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    reason = "no PMC summary committed for this round"
    for fn in files:
        path = os.path.join(ROOT, "profiles", fn)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        if d.get("lib_sha256") != sha:
            reason = f"profiles/{fn} was collected on another build of libsimt_hip.so (sha256 {str(d.get('lib_sha256'))[:12]}... != {sha[:12]}...)"
            continue
        # `kernel` may name a family ("conv_igemm2_kernel<256, 5, 3, *>": every epilogue flavour of one tile shape): dispatch-weighted average
        pref = kernel[:-2] if kernel.endswith("*>") else None
        hits = [v for kname, v in d["kernels"].items()
                if (kname.replace("void ", "").strip().startswith(pref) if pref else kname.replace("void ", "").strip() == kernel)]
        if hits:
            n = sum(v["dispatches"] for v in hits)
            return sum(v[key] * v["dispatches"] for v in hits) / max(n, 1), "profiles/" + fn, None
        reason = f"profiles/{fn} has no entry for {kernel}"
    return None, None, reason


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", default="v2", choices=["v2", "v3", "vgg"],
                    help="v2: DeepLabv2-R101 (BASELINE configs[1]/[2], the headline); v3: DeepLabv3 + SimT K=6 at 512x1024 (configs[3]); "
                         "vgg: DeepLab-VGG16 + SimT K=3, batch 8 at 512x512 (configs[4])")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: 4; vgg: 8)")
    ap.add_argument("--size", type=int, nargs=2, default=None, metavar=("H", "W"), help="default 768 768; v3: 512 1024; vgg: 512 512")
    ap.add_argument("--open-classes", type=int, default=None, help="default 3; v3: 6")
    ap.add_argument("--v3-layers", type=int, nargs=3, default=[3, 4, 6], metavar=("L1", "L2", "L3"),
                    help="--model v3: Bottlenecks of layer1..3.  3 4 6 = model/deeplabv3.py as written (a torchvision ResNet-50 cut after layer3); "
                         "3 4 23 = the ResNet-101 depth BASELINE.json's configs[3] NAMES ('DeepLabv3-ResNet101')")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the trained-like and PCIe-inclusive passes")
    ap.add_argument("--skip-unapplied-grads", action="store_true",
                    help="stop the backward at layer3 (the gradients of conv1/layer1/layer2 are never applied by the SimT stage); "
                         "NOT the headline configuration: the default computes everything the reference's iteration computes")
    ap.add_argument("--shapes", action="store_true", help="print a per-shape conv timing table to stderr")
    ap.add_argument("--cpu-iters", type=int, default=3, help="timed CPU-baseline iterations per point (after one warm-up)")
    a = ap.parse_args()
    dflt = {"v2": (4, [768, 768], 3), "v3": (4, [512, 1024], 6), "vgg": (8, [512, 512], 3)}[a.model]
    a.batch = a.batch if a.batch is not None else dflt[0]
    a.size = a.size if a.size is not None else dflt[1]
    a.open_classes = a.open_classes if a.open_classes is not None else dflt[2]
    return a


def spawn_ranks(n):
    """--gpus N from a plain `python bench.py`: become the launcher.  Nothing in this process has touched the GPU yet."""
    port = 29500 + (os.getpid() % 2000)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(K, iters):
    """The oracle (CPU restatement of the reference, pinned by tests/golden) on the host cores: `iters` timed iterations (after one
    warm-up) at B=1 on config c2's shape (768x768) and on config c1 (512x512) -- a bounded sample of the same workload.  16 threads:
    PyTorch's CPU convs slow down when a 2-socket box is oversubscribed (measured on this pool at 768x768: 8 thr 9.9 s, 16 thr
    7.3 s, 32 thr 9.4 s, 128 thr 58 s); `cores` reports the count actually used.  Checker-as-baseline only; never on the product path."""
    from oracle import simt_oracle as so
    cd = so.load_class_dist()
    st = so.recipe_state(so.state_shapes(19, K, True), seed=1234, trained_like=False)
    fst = so.recipe_state(so.state_shapes(19, 0, False), seed=1234, trained_like=False)
    hp = so.Hyper(open_classes=K, lr=6e-4, lr_T=6e-3)
    tr = so.OracleTrainer(st, fst, so.ntm_init(19, K, 1), so.ntm_init(19, K, 2), hp, cd)
    ncpu = os.cpu_count() or 8
    nthr = max(1, min(16, ncpu))
    torch.set_num_threads(nthr)
    img, lab = so.synthetic_batch(1, 129, 129, cd.numpy(), seed=1)
    tr.step(img, lab, 0)                                  # warm-up (thread pool, allocator)
    pts = {}
    it = 1
    for size in (768, 512):
        ts = []
        for k in range(iters):
            img, lab = so.synthetic_batch(1, size, size, cd.numpy(), seed=2 + it)
            t0 = time.perf_counter()
            tr.step(img, lab, it)
            ts.append(time.perf_counter() - t0)
            it += 1
        ts.sort()
        pts[size] = ts[len(ts) // 2]
    return {"value": round(1.0 / pts[768], 5), "unit": "images/s", "cores": nthr, "kind": "port",
            "sample": f"median of {iters} iterations, B=1, 768x768, K={K}, fp32, torch CPU, {nthr} threads of {ncpu} ({pts[768]:.1f} s each)",
            "c1_512x512": {"value": round(1.0 / pts[512], 5), "s_per_iter": round(pts[512], 2)}}


class PowerWatch:
    """Socket power and shader clock of THIS process's GPU during the timed steps, sampled from sysfs (hwmon power1_average, the starred level of
    pp_dpm_sclk) by a thread every 0.1 s.  Context for the roofline fraction, not a metric: the peaks of MI355X_MICROARCH.md are quoted at the
    nominal 2.4 GHz, and the step of this bench runs power-managed below it (profiles/r05_power_clock.txt: 1 218 W, 2 166 MHz sustained; the dominant
    conv alone 1 941 MHz)."""

    def __init__(self, device_index):
        import glob
        import threading
        self.dir, self.samples, self._stop, self._thr = None, [], threading.Event(), None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for c in glob.glob("/sys/class/drm/card[0-9]*"):
                if os.path.basename(os.path.realpath(os.path.join(c, "device"))) == addr:
                    self.dir = os.path.join(c, "device")
            if os.environ.get("SIMT_BENCH_NO_POWER") == "1":      # A/B of the sampler itself
                self.dir = None
            if self.dir:
                hw = glob.glob(os.path.join(self.dir, "hwmon", "hwmon*"))
                self.pfile = next((os.path.join(h, f) for h in hw for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, f))), None)
                self._thr = threading.Thread(target=self._run, daemon=True)
        except Exception:
            self.dir = None

    def _read(self):
        w = mhz = top = None
        try:
            if self.pfile:
                w = float(open(self.pfile).read()) / 1e6
            lv = open(os.path.join(self.dir, "pp_dpm_sclk")).read().split("\n")
            vals = [float(l.split()[1].lower().replace("mhz", "")) for l in lv if l.strip()]
            cur = [float(l.split()[1].lower().replace("mhz", "")) for l in lv if "*" in l]
            mhz, top = (cur[0] if cur else None), (max(vals) if vals else None)
        except Exception:
            pass
        return w, mhz, top

    def _run(self):
        while not self._stop.is_set():
            self.samples.append(self._read())
            self._stop.wait(0.1)

    def __enter__(self):
        if self._thr:
            self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        if self._thr:
            self._thr.join()

    def report(self):
        w = [x[0] for x in self.samples if x[0] is not None]
        m = [x[1] for x in self.samples if x[1] is not None]
        t = [x[2] for x in self.samples if x[2] is not None]
        if not w and not m:
            return None
        return {"avg_w": round(sum(w) / len(w), 1) if w else None, "sclk_mhz_avg": round(sum(m) / len(m), 1) if m else None,
                "sclk_mhz_top_level": max(t) if t else None, "samples": len(self.samples),
                "what": "socket power and shader clock of this GPU during the timed steps (sysfs, every 0.1 s): the step runs power-managed below "
                        "the nominal clock the MFMA peak is quoted at"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("SIMT_DIST_BACKEND", "nccl")      # "nccl" = RCCL over xGMI; "gloo" only for functional tests on one GPU
    if world > 1 and backend == "nccl" and ndev < world:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, this node shows {ndev}")
    local = local % max(ndev, 1)       # (several ranks share a device only in the gloo functional test)
    if world > ndev:                   # ... where the grid-barrier BatchNorm launches of two processes could starve each other (engine.py)
        os.environ["SIMT_BN_GRID"] = "0"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    # SIMT_DP_FORCE=1 under a one-rank torchrun: the data-parallel job over a REAL one-rank RCCL group (functional test of the N > 1 path -- process
    # group, barrier, bucketed all-reduce, comm report -- on a 1-GPU box; never a measurement: the exchange is the identity)
    dp_job = world > 1 or (os.environ.get("SIMT_DP_FORCE") == "1" and "RANK" in os.environ)
    # the plan's two streams take their hardware queues before anything else makes streams (RCCL / ProcessGroupNCCL do): engine.reserve_streams
    from simt_amd.engine import reserve_streams
    reserve_streams(dev)
    if dp_job:
        import torch.distributed as dist
        # The conv tile lists of a data-parallel plan leave 20 of the 256 CUs free (236 one-per-CU workgroups: engine.TrunkPlan.cu_budget); RCCL is
        # told to stay inside that margin unless the user says otherwise (must be set before the communicator exists)
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "16")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world
        pg = dist.group.WORLD
    from simt_amd import model_spec as ms
    from simt_amd.step import Hyper, SimTTrainer

    H, W = a.size
    K = a.open_classes
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    cd = ms.load_class_dist("bapa")
    hp = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3, skip_unapplied_grads=a.skip_unapplied_grads)       # sh_simt.sh:16

    def make_trainer(trained_like, dt=None, stats=None):
        """dt: storage / arithmetic mode (default: --dtype).  stats: BatchNorm running statistics to take over instead of calibrating (so that a
        bf16 and an fp32 trainer start from the IDENTICAL state)."""
        dt = dtype if dt is None else dt
        if a.model != "v2":
            from simt_amd.step_single import SimTSingleTrainer
            if a.model == "v3":
                from simt_amd.engine_v3 import v3_state_shapes
                lay = tuple(a.v3_layers)
                st = ms.kaiming_init(v3_state_shapes(19, K, True, layers=lay), seed=1234)
                fst = ms.kaiming_init(v3_state_shapes(19, 0, False, layers=lay), seed=1234)
                return SimTSingleTrainer(a.model, st, fst, ms.ntm_init(19, K, 2), hp, cd, a.batch, H, W, dtype=dtype, device=dev, process_group=pg,
                                         arch={"layers": lay})
            else:
                from simt_amd.engine_vgg import vgg_state_shapes
                st = ms.kaiming_init(vgg_state_shapes(19 + K), seed=1234)
                fst = ms.kaiming_init(vgg_state_shapes(19), seed=1234)
            return SimTSingleTrainer(a.model, st, fst, ms.ntm_init(19, K, 2), hp, cd, a.batch, H, W, dtype=dtype, device=dev, process_group=pg)
        init = ms.trained_like_init if trained_like else ms.reference_init
        st = init(ms.state_shapes(19, K, True), seed=1234)
        fst = init(ms.state_shapes(19, 0, False), seed=1234)
        t = SimTTrainer(st, fst, ms.ntm_init(19, K, 1), ms.ntm_init(19, K, 2), hp, cd, a.batch, H, W, dtype=dt, device=dev,
                        process_group=pg)
        if trained_like and stats is not None:
            for k, v in stats.items():
                t.params[k].copy_(v)
                if k in t.fixed_params:
                    t.fixed_params[k].copy_(v)
            t.fixed.repack()
            torch.cuda.synchronize()
        elif trained_like:
            # A checkpoint's running statistics describe its own activations.  Synthetic ones do not, and 101 layers of eval-mode BN
            # with mismatched statistics collapse the frozen model's features (every posterior ~1/19).  Calibrate them: 40 train-mode
            # forwards of the (identical) trainable trunk on a synthetic batch move its running statistics onto the batch statistics
            # (momentum 0.1), then the frozen model takes them over and re-folds its BatchNorms.
            img, _ = ms.synthetic_batch(a.batch, H, W, cd, seed=99, device=dev)
            t.plan.x_in.copy_(img)
            for _ in range(40):
                t.plan.fwd_list.run()
            for k, v in t.fixed_params.items():
                if k.endswith("running_mean") or k.endswith("running_var"):
                    v.copy_(t.params[k])
            t.fixed.repack()
            torch.cuda.synchronize()
        return t

    def barrier():
        torch.cuda.synchronize()
        if dp_job:
            import torch.distributed as dist
            if dist.get_backend() == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def timed(tr, batches, steps, warmup):
        """`steps` iterations between barriers; returns (seconds for all steps [max over ranks], median per-step ms from HIP events)."""
        for _ in range(warmup):
            tr.step(*next(batches))
        barrier()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        evs[0].record()
        for i in range(steps):
            tr.step(*next(batches))
            evs[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        if dp_job:
            import torch.distributed as dist
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
        return dt, per[len(per) // 2]

    def resident(seed):
        img, lab = ms.synthetic_batch(a.batch, H, W, cd, seed=seed, device=dev)
        while True:
            yield img, lab

    tr = make_trainer(False)
    red = getattr(tr, "reducer", None)
    if red is not None:
        red.measure = True
    with PowerWatch(local) as pw:
        dt, med = timed(tr, resident(1234 + rank), a.steps, a.warmup)
    # a fused BatchNorm launch that gave up polling leaves the sticky error word set and every optimiser launch skipped: never report such a run
    if getattr(tr, "plan", None) is not None and tr.plan.fbn_error():
        raise SystemExit("bench.py: a fused BatchNorm launch timed out (TrunkPlan.fbn_error): the timed steps did not train; rerun with SIMT_BN_GRID=0")
    comm = None
    if red is not None:
        comm = red.report()              # the exchange of the timed steps (plus warm-up): bytes, buckets, exposed wait
        comm["batchnorm_form"] = ("fused BatchNorm backward launches: %d -- the SAME form as the single-GPU plan (round 6; rounds 4-5 ran the two-pass "
                                  "form under a process group, a built-in N = 1 -> N > 1 loss of ~0.3-0.6 ms per step)" % tr.plan.fbn_launches)
        comm["cu_budget"] = tr.plan.cu_budget
        comm["nccl_max_nchannels"] = os.environ.get("NCCL_MAX_NCHANNELS")
        comm["what"] = ("mean all-reduce of the applied prefix of the flat gradient buffer + the NTM gradients over "
                        + ("RCCL/xGMI" if backend == "nccl" else backend) + ", bucketed, issued from the plan's side stream under the backward; "
                        "exposed_wait = time the optimiser-step stream sat waiting for it in BucketReducer.finish()")
        red.measure = False
    ms_step = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    roof, aspp = None, None
    if not a.no_roofline:
        acc, shp = {}, {}
        for _ in range(2):
            acc.clear()
            shp.clear()
            for lst in tr.timed_lists():
                lst.run_timed(acc, shp)
        if a.shapes and rank == 0:
            for (tag, shape), v in sorted(shp.items(), key=lambda kv: -kv[1][0]):
                print(f"{tag:28s} {shape:44s} n={v[3]:3d} total {v[0]:7.3f} ms  avg {v[0] / v[3] * 1e3:8.1f} us  "
                      f"{v[1] / (v[0] * 1e-3) / 1e12:7.1f} TF/s  {v[2] / (v[0] * 1e-3) / 1e9:7.0f} GB/s(alg)", file=sys.stderr)
        # dominant kernel = the conv tile shape with the largest total time, over ALL its compile-time epilogue flavours (round 4 split every
        # shape into flavours <bn, tm, nst, fused-BN, epilogue>; the family is what round 3 reported as one instantiation)
        fam = {}
        for k, v in acc.items():
            if k.startswith("conv_igemm"):
                f = (", ".join(k.split(", ")[:3]) + ", *>") if k.startswith("conv_igemm2_kernel<") else k
                a_ = fam.setdefault(f, [0.0, 0.0, 0.0, 0])
                for i in range(4):
                    a_[i] += v[i]
        dom = max(fam, key=lambda k: fam[k][0])
        ms_k, fl, by, n = fam[dom]
        flav = {k: {"ms": round(v[0], 3), "us": round(v[0] / v[3] * 1e3, 1), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1), "n": v[3]}
                for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]) if dom.endswith("*>") and k.startswith(dom[:-2])}
        ach = fl / (ms_k * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[a.dtype]
        tot_ms = sum(v[0] for v in acc.values())
        traffic, tsrc, twhy = pmc_lookup(PMC_FILES, dom, "hbm_bytes_per_launch_corrected")      # HBM bytes per launch (FETCH_SIZE / WRITE_SIZE passes)
        mfma_busy, msrc, mwhy = pmc_lookup(PMC_MFMA_FILES, dom, "mfma_util")    # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
        # ---- the step north_star puts a number on: ASPP (the tap-expanded classifier GEMMs of both nets, forward and backward) + T-matrix
        # (the fused head: softmax x T contraction, losses and their gradients).  Algorithmic FLOPs / event time of those launches.
        aspp = None
        if a.model == "v2":
            hg = [(k2, v) for k2, v in shp.items() if "(tap-expanded head)" in k2[1]]
            g_ms, g_fl, g_n = sum(v[0] for _, v in hg), sum(v[1] for _, v in hg), sum(v[3] for _, v in hg)
            aux = sum(v[0] for k2, v in acc.items() if k2 in ("simt_tap_gather_sum", "simt_tap_scatter"))
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            import ctypes
            from simt_amd import _lib as simt_lib
            cur = torch.cuda.current_stream().cuda_stream
            for _ in range(2):               # the fused head of the resident batch, alone on the stream (its position in the step: exclusive)
                evs[0].record()
                simt_lib.call("simt_head_loss", ctypes.byref(tr.head_desc), cur)
                evs[1].record()
                simt_lib.call("simt_head_grad", ctypes.byref(tr.head_desc), cur)
                evs[2].record()
                torch.cuda.synchronize()
            P = a.batch * H * W
            Q = 19 + K
            t_fl = 2 * 2.0 * P * Q * 19 * 2        # logits x T forward + its two gradient contractions, both heads (2 P Q C each)
            aspp = {"gemm_launches": g_n, "gemm_ms": round(g_ms, 4), "gemm_alg_gflop": round(g_fl / 1e9, 2),
                    "gemm_tflops": round(g_fl / (g_ms * 1e-3) / 1e12, 1) if g_ms else None,
                    "gemm_frac_of_peak": round(g_fl / (g_ms * 1e-3) / 1e12 / peak, 4) if g_ms else None,
                    "gemm_by_shape": {k2[1]: {"us": round(v[0] / v[3] * 1e3, 1), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 1), "n": v[3]} for k2, v in hg},
                    "tap_gather_scatter_ms": round(aux, 4),
                    "head_loss_ms": round(evs[0].elapsed_time(evs[1]), 4), "head_grad_ms": round(evs[1].elapsed_time(evs[2]), 4),
                    "logitsT_alg_gflop": round(t_fl / 1e9, 2),
                    # the WHOLE ASPP + T-matrix step as it runs: (GEMM + logits x T FLOPs) / (GEMMs + tap gather / scatter + fused head) / peak
                    "whole_step_ms": round(g_ms + aux + evs[0].elapsed_time(evs[2]), 4),
                    "whole_step_frac": round((g_fl + t_fl) / ((g_ms + aux + evs[0].elapsed_time(evs[2])) * 1e-3) / 1e12 / peak, 4),
                    "what": "tap-expanded ASPP classifier GEMMs (3 forward: two trainable heads + the frozen main head; 2 dgrad; wgrads are in "
                            "conv_wgrad) as launched in the step, plus the fused head kernels (upsample, softmax, logits x T, the nine losses and "
                            "their gradients: VALU per lane, 22 x 19 mat-vec, SURVEY allows) timed alone; target north_star: >= 0.5 of MFMA peak.  "
                            "Bound for THIS decomposition (DESIGN.md 5b): the GEMMs run as N = 432 columns in two 256-column tiles (19 % padding) with an fp32 "
                            "result or as K = 448 dgrads (output-bound: 154 MB bf16 result + bit mask for 66 GFLOP), at most ~0.4 ms at this kernel's best "
                            "shape rate -> gemm_frac_of_peak <= 0.3; with the 0.7-ms fused loss block (VALU, no MFMA shape) whole_step_frac <= 0.15"}
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc or twhy, "mfma_busy": mfma_busy, "mfma_busy_source": msrc or mwhy,
                "alg_bytes_per_launch": int(by / n), "launches_per_step": n,
                "avg_launch_us": round(ms_k / n * 1e3, 2), "alg_gflop_per_launch": round(fl / n / 1e9, 3),
                "share_of_step_kernel_time": round(ms_k / tot_ms, 3),
                "clock_note": ("peak is the guide's dense bf16 figure at the nominal 2.4 GHz; this kernel runs power-managed at ~1.94 GHz (in-kernel s_memtime "
                               "against s_memrealtime, profiles/r05_power_clock.txt), where the matrix pipes offer ~2.02 PFLOP/s") if a.dtype == "bf16" else None,
                "flavours": flav,      # <..., fused BatchNorm (0 | 1), epilogue (0 generic, 1 statistics, 2 BN-backward reduce, 3 bias + ReLU, 4-7 dgrads)>
                "classes": {k: {"ms": round(v[0], 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[1] else None,
                                "n": v[3]} for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0])[:16]}}

    extra = {}
    if not a.no_extra_passes and a.model == "v2":
        # ---- PCIe-inclusive: decoded uint8 frames in pinned host memory -> upload + device conversion every step, one batch ahead
        from simt_amd.data.pipeline import DevicePrefetcher, InputPrep
        rgb, lab8 = ms.synthetic_batch_u8(a.batch, H, W, cd, seed=1234 + rank)
        rgb, lab8 = rgb.pin_memory(), lab8.pin_memory()
        prep = InputPrep(a.batch, (H, W), (W, H), dev)

        def host_frames():
            while True:
                yield rgb, lab8, None
        pf = DevicePrefetcher(host_frames(), prep)
        feed = ((x, l) for (x, l, _m) in pf)
        n2 = max(5, min(a.steps, 20))
        dt2, med2 = timed(tr, feed, n2, 2)
        extra["h2d_inclusive"] = {"value": round(a.batch * world * n2 / dt2, 3), "ms_per_step": round(dt2 / n2 * 1e3, 3), "steps": n2,
                                  "bytes_per_step_per_gpu": int(rgb.numel() + lab8.numel()),
                                  "what": "uint8 RGB + uint8 labels from pinned host memory each step (double-buffered copy stream), "
                                          "BGR-mean / CHW / int64 conversion on the device (csrc/input_prep.hip)"}
        del pf, feed, prep
        # ---- checkpoint-like weights: the frozen model's posteriors cross both confidence thresholds
        del tr
        torch.cuda.empty_cache()
        tr = make_trainer(True)
        # the nine losses of ONE iteration from the calibrated checkpoint-like state, in the throughput mode (bf16) -- and, below, of the SAME
        # iteration (identical weights, statistics, batch) in the fp32 parity mode: the size of the mode's effect AT configs[1]
        tl_stats = {k: v.detach().clone() for k, v in tr.params.items() if k.endswith("running_mean") or k.endswith("running_var")}
        tl_batch = next(resident(1234 + rank))
        tr.step(*tl_batch)
        torch.cuda.synchronize()
        tl_loss = {a.dtype: [round(float(v), 6) for v in tr.lout[:9].cpu().tolist()]}
        tl_conf = {a.dtype: int(tr.hout[6].item())}
        dt3, med3 = timed(tr, resident(1234 + rank), n2, 3)
        hout = tr.hout.cpu()
        P = a.batch * H * W
        extra["trained_like_pass"] = {"value": round(a.batch * world * n2 / dt3, 3), "ms_per_step": round(dt3 / n2 * 1e3, 3), "steps": n2,
                                      "pixels_with_confidence_label": int(hout[6].item()), "pixels": P,
                                      "what": "same workload with ms.trained_like_init weights (non-trivial BN statistics, scaled "
                                              "classifiers): both --Threshold-high and --Threshold-low branches are populated"}
        # ---- NOT the headline: the same iteration with the backward stopped at the input of layer3.  The SimT stage's optimiser never lists
        # conv1 / layer1 / layer2 (model/deeplab_multi.py:194-237), so their weight gradients -- which the reference's autograd computes and the
        # headline above computes too -- are dead values: the trajectory is bit-identical without them (tests/test_gpu_iteration.py).  Reported
        # so that the price of matching the reference's work item for item is on the line.
        if not a.skip_unapplied_grads:
            del tr
            torch.cuda.empty_cache()
            hp_skip = Hyper(open_classes=K, lr=6e-4, lr_T=6e-3, skip_unapplied_grads=True)
            init = ms.reference_init
            tr = SimTTrainer(init(ms.state_shapes(19, K, True), seed=1234), init(ms.state_shapes(19, 0, False), seed=1234), ms.ntm_init(19, K, 1),
                             ms.ntm_init(19, K, 2), hp_skip, cd, a.batch, H, W, dtype=dtype, device=dev, process_group=pg)
            dt4, _ = timed(tr, resident(1234 + rank), n2, 3)
            extra["skip_unapplied_grads_pass"] = {"value": round(a.batch * world * n2 / dt4, 3), "ms_per_step": round(dt4 / n2 * 1e3, 3), "steps": n2,
                                                  "what": "NOT the headline: backward stopped at layer3's input (the gradients of conv1 / layer1 / layer2 "
                                                          "are never applied by the SimT stage; identical parameter trajectory); Hyper(skip_unapplied_grads=True)"}
        # ---- the price of the parity mode: the SAME iteration in fp32 storage / fp32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 matrix
        # rate) -- the mode whose losses the 1e-4 parity claims are made in (tests/test_gpu_configs.py, smoke()).  A labelled secondary
        # number, never the headline.
        if a.dtype == "bf16" and world == 1:
            del tr
            torch.cuda.empty_cache()
            init = ms.reference_init
            tr = SimTTrainer(init(ms.state_shapes(19, K, True), seed=1234), init(ms.state_shapes(19, 0, False), seed=1234), ms.ntm_init(19, K, 1),
                             ms.ntm_init(19, K, 2), hp, cd, a.batch, H, W, dtype=torch.float32, device=dev, process_group=pg)
            n5 = 4
            dt5, _ = timed(tr, resident(1234 + rank), n5, 2)
            del tr
            torch.cuda.empty_cache()
            tr = make_trainer(True, torch.float32, tl_stats)
            tr.step(*tl_batch)
            torch.cuda.synchronize()
            tl_loss["f32"] = [round(float(v), 6) for v in tr.lout[:9].cpu().tolist()]
            tl_conf["f32"] = int(tr.hout[6].item())
            names9 = ["total", "loss_p1", "loss_p2", "loss_y1", "loss_y2", "place", "convex", "volume", "anchor"]      # step.py lout[0:9]
            extra["trained_like_pass"]["first_iteration_losses"] = {
                "order": names9, "bf16": tl_loss["bf16"], "f32": tl_loss["f32"],
                "rel_diff": [round(abs(x - y) / (1.0 + abs(y)), 6) for x, y in zip(tl_loss["bf16"], tl_loss["f32"])],
                "pixels_with_confidence_label": tl_conf,
                "what": "the nine losses of ONE iteration from the identical calibrated checkpoint-like state and batch, throughput mode (bf16 storage) "
                        "beside the fp32 parity mode (the mode the 1e-4 claims are made in); rel_diff = |bf16 - f32| / (1 + |f32|)"}
            extra["f32_parity_mode_pass"] = {"value": round(a.batch * world * n5 / dt5, 3), "ms_per_step": round(dt5 / n5 * 1e3, 3), "steps": n5,
                                             "dtype": "f32",
                                             "what": "NOT the headline: the same iteration in the fp32 parity mode (fp32 storage, fp32 MFMA with a k-ordered "
                                                     "fmaf chain; the kernels that carry 'loss within 1e-4'): what the parity mode costs"}
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(K, a.cpu_iters)
    if rank == 0:
        if a.model == "v2":
            flop_img = FLOP_PER_IMAGE_768 * (H * W) / (768.0 * 768.0)
        else:                           # algorithmic conv FLOPs of the three launch lists (frozen forward, forward, backward) per image
            flop_img = sum(it.flops for lst in tr.timed_lists() for it in lst.items if it.fn is not None) / a.batch
        names = {"v2": ("DeepLabv2-R101+SimT", "DeepLabv2-ResNet101"), "v3": (("DeepLabv3-R50+SimT", "DeepLabv3 (torchvision-style ResNet-50 to layer3 + ASSP; model/deeplabv3.py as written)") if tuple(a.v3_layers) == (3, 4, 6)
                        else (f"DeepLabv3-R{'101' if tuple(a.v3_layers) == (3, 4, 23) else '?'}+SimT", f"DeepLabv3 (ResNet layers {tuple(a.v3_layers)} to layer3 + ASSP)")),
                 "vgg": ("DeepLab-VGG16+SimT", "DeepLab-VGG16")}[a.model]
        cfg = {"v2": "configs[1]" if world == 1 else "configs[2]", "v3": "configs[3]", "vgg": "configs[4]"}[a.model]
        mode = ("bf16 storage / fp32 accumulate = throughput mode (parity claims 'loss within 1e-4' are made by the fp32 mode of the same "
                "kernels; bf16 is held to the float64 bf16-storage model, tests/test_gpu_prod_shapes.py)") if a.dtype == "bf16" else "fp32 parity mode"
        line = {"metric": f"training images/sec at {H}x{W}, {names[0]}", "value": round(value, 3),
                "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_step, 3),
                "ms_per_step_median": round(med, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
                "config": {"workload": f"{names[1]} + SimT(C=19,K={K}) full training iteration, batch={a.batch}/GPU, "
                                       f"{H}x{W}, {a.dtype}, {world}xMI355X" + (" DP RCCL all-reduce" if dp_job else "")
                                       + (" [backward stops at layer3: unapplied gradients skipped]" if a.skip_unapplied_grads else ""),
                           "global_batch": a.batch * world, "baseline_config": cfg,
                           "numerics": mode, "inputs": "resident in HBM (see h2d_inclusive for the PCIe-inclusive rate)",
                           "step_tflops_conv_algorithmic": round(value * flop_img / 1e12, 1),
                           "frac_of_conv_roofline": round(value * flop_img / 1e12 / (world * MFMA_PEAK_TFLOPS[a.dtype]), 4)},
                "roofline": roof, "cpu_baseline": cpu}
        if roof is not None and aspp is not None:
            line["aspp_t_step"] = aspp
        if comm is not None:
            line["comm"] = comm
        if pw.report() is not None:
            line["power_clock"] = pw.report()
        line.update(extra)
        print(json.dumps(line))
    if dp_job:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
