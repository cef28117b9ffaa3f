set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/dp_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SIMT_DP_FORCE=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29621 rocprofv3 --kernel-trace --output-format csv -d $OUT/dp -- python3 $ROOT/bench.py --gpus 1 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-passes > $OUT/dp.log 2>&1
cd $ROOT
f=$(find $OUT/dp -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]
sel = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]
print("step window ms %.3f kernels %d" % ((b - a) / 1e6, len(sel)))
q = collections.Counter(r["Queue_Id"] for r in sel)
print("kernels per queue:", dict(q))
for r in sel:
    n = r["Kernel_Name"]
    if any(k in n.lower() for k in ("nccl", "rccl", "reduce_scatter", "allreduce", "all_reduce")) and "wgrad" not in n:
        print("  %8.3f ms  %7.1f us  q=%s  %s" % ((int(r["Start_Timestamp"]) - a) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Queue_Id"], n[:90]))
PY
find $OUT -name "*.csv" -size +5M -delete
