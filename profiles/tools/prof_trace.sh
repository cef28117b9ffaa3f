set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_t2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prod -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-passes > $OUT/prod.log 2>&1
cd $ROOT
f=$(find $OUT/prod -name "*kernel_trace.csv" | head -1); python3 - "$f" > $OUT/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
# last 40% of the trace = steady-state steps
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
# find step boundaries by sgd_multi_kernel occurrences
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
print("sgd ends:", [(s - t0) / 1e6 for s in sg])
a, b = sg[-2], sg[-1]
sel = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]
print("step window ms:", (b - a) / 1e6, "kernels:", len(sel))
key = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
per = collections.defaultdict(float); cnt = collections.Counter()
for r in sel:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    per[r[key]] += d; cnt[r[key]] += 1
for k in per: print("queue", k, "busy ms %.2f" % per[k], "kernels", cnt[k])
# per queue per kernel class
cls = collections.defaultdict(float)
for r in sel:
    n = r["Kernel_Name"].split("(")[0][:60]
    cls[(r[key], n)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (q, n), v in sorted(cls.items(), key=lambda kv: -kv[1])[:40]: print("  q%s %-62s %.2f" % (q, n, v))
# idle gaps on each queue inside the window
for q in per:
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if r[key] == q)
    gaps = sum(max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)) / 1e6
    print("queue", q, "first start %.2f last end %.2f gaps %.2f ms" % ((ev[0][0] - a) / 1e6, (ev[-1][1] - a) / 1e6, gaps))
    evn = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-40:]) for r in sel if r[key] == q)
    gl = sorted(((evn[i + 1][0] - evn[i][1]) / 1e3, (evn[i][1] - a) / 1e6, evn[i][2], evn[i + 1][2]) for i in range(len(evn) - 1))
    import numpy as np
    g = np.array([x[0] for x in gl])
    print("   gap us percentiles 10/50/90/99:", np.percentile(g, [10, 50, 90, 99]).round(1), " sum of gaps > 20 us: %.2f ms (%d)" % (g[g > 20].sum() / 1e3, (g > 20).sum()))
    for x in gl[-12:]: print("     gap %.0f us at %.2f ms: %s -> %s" % x)
# sequence around the head
hp=[r for r in sel if "head_pass1" in r["Kernel_Name"]]
if hp:
    c0=int(hp[0]["Start_Timestamp"])-400000; c1=c0+2600000
    print("--- kernels around the head (ms from window start, dur us, queue, name)")
    for r in sel:
        st=int(r["Start_Timestamp"]); en=int(r["End_Timestamp"])
        if c0<=st<c1: print("  %8.3f %7.1f q%s %s" % ((st-a)/1e6,(en-st)/1e3,r[key],r["Kernel_Name"].split("(")[0][-50:]))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/summary.txt | tail -120
