"""bench.py as the driver runs it: `--gpus N` must produce an N-rank job and ONE JSON line with n_gpus == N carrying the
roofline object.  With >= 2 GPUs the ranks talk RCCL; on a 1-GPU box the same launcher path is exercised with two ranks sharing
the device over gloo (SIMT_DIST_BACKEND=gloo: functional check of spawn + DP step + max-over-ranks timing, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]
    return json.loads(lines[0])


def test_bench_gpus2_spawns_two_ranks(dev):
    small = ["--steps", "2", "--warmup", "1", "--batch", "1", "--size", "129", "129", "--no-cpu-baseline", "--no-extra-passes"]
    if torch.cuda.device_count() >= 2:
        line = _run({}, ["--gpus", "2"] + small)
        assert "RCCL" in line["config"]["workload"]
    else:
        line = _run({"SIMT_DIST_BACKEND": "gloo"}, ["--gpus", "2"] + small)
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 2 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    # the N > 1 line says where the exchange went: payload per step (the applied prefix of the flat gradient buffer + 2 NTM
    # gradients + the bad-label count and three spare slots in ONE extra collective; 42.2 M fp32 = 168.7 MB for the full model), number of buckets, when each bucket was released by the
    # backward replay, and the EXPOSED wait of the optimiser-step stream
    c = line["comm"]
    assert c["world"] == 2 and c["buckets"] >= 1 and sum(c["bucket_bytes"]) + 2 * 22 * 19 * 4 + 16 == c["bytes_per_step"] and c["extra_tensors"] == 1
    assert len(c["bucket_released_launch"]) == c["buckets"] == len(c["bucket_ready_launch"]) and c["backward_launches"] > 0
    # every bucket leaves at the first hook point at or after the launch that completes its last gradient (never at finish(): -1 would be a
    # bucket that only left when the whole backward had been enqueued), in order
    rel, rdy = c["bucket_released_launch"], c["bucket_ready_launch"]
    assert all(r >= q >= 0 for r, q in zip(rel, rdy)) and rel == sorted(rel) and rel[-1] <= c["backward_launches"]
    assert 160e6 < c["bytes_per_step"] < 180e6 and c["steps_measured"] >= 2
    assert c["exposed_wait_ms_median"] is not None and c["exposed_wait_ms_median"] >= 0.0


def test_bench_single_gpu_line_has_every_contract_field(dev):
    line = _run({}, ["--steps", "3", "--warmup", "1", "--batch", "1", "--size", "257", "257", "--cpu-iters", "1"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "h2d_inclusive", "trained_like_pass", "ms_per_step_median"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["roofline"]["bound"] == "mfma" and line["cpu_baseline"]["kind"] == "port"
    assert 0 <= line["trained_like_pass"]["pixels_with_confidence_label"] <= line["trained_like_pass"]["pixels"]
    # power / shader clock of the timed steps, where the box lets the process read its card's sysfs (context for the roofline fraction)
    pc = line.get("power_clock")
    if pc is not None:
        assert pc["samples"] >= 1 and (pc["sclk_mhz_avg"] is None or 50 <= pc["sclk_mhz_avg"] <= 3000)


def test_bench_dp_job_over_a_real_one_rank_rccl_group(dev):
    """The driver's N > 1 launch line with ONE rank and SIMT_DP_FORCE=1: bench.py's data-parallel job -- RCCL process group bound to the device,
    barrier, bucketed ReduceOp.AVG all-reduce on the comm stream under the backward, max-over-ranks timing, comm report -- runs end to end over
    the production backend (a 1-GPU box cannot hold two RCCL ranks; the two-rank functional test above uses gloo)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["SIMT_DP_FORCE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "1", "--size", "129", "129", "--no-cpu-baseline",
           "--no-extra-passes"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-3000:]
    line = json.loads(lines[0])
    c = line["comm"]
    assert line["n_gpus"] == 1 and "RCCL" in line["config"]["workload"] and c["op"] == "AVG" and c["world"] == 1
    assert c["steps_measured"] >= 3 and c["exposed_wait_ms_median"] >= 0.0 and all(r >= 0 for r in c["bucket_released_launch"])
    assert line["value"] > 0
