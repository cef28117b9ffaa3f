#!/bin/bash
# bench_box.sh tag: the driver's bench command on this box -> gpurun_out/boxes/<tag>.json + a one-line summary
mkdir -p gpurun_out/boxes
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/boxes/$1.json 2> gpurun_out/boxes/$1.err
python3 - "$1" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/boxes/{sys.argv[1]}.json").read().strip().split("\n")[-1])
r, p, a = d["roofline"], d.get("power_clock") or {}, d["aspp_t_step"]
print("BOX", d["value"], d["ms_per_step"], "frac", r["frac"], "launch_us", r["avg_launch_us"], "traffic", r["traffic"], "mfma_busy", r.get("mfma_busy"),
      "aspp", a["gemm_frac_of_peak"], a["whole_step_frac"], "h2d", (d.get("h2d_inclusive") or {}).get("value"), p)
PY
