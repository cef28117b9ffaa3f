# rocprofv3 kernel trace of a production step: how much of the step is at least one kernel running (union over both HIP streams), how much two
# kernels overlap, where the GPU idles.   usage (through gpurun, from the repo root): bash profiles/tools/prof_union.sh
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_union; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/prod -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-passes > $OUT/prod.log 2>&1
cd $ROOT
f=$(find $OUT/prod -name "*kernel_trace.csv" | head -1); python3 - "$f" > $OUT/summary.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]
sel = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:], r["Queue_Id"]) for r in rows if a <= int(r["Start_Timestamp"]) < b]
print("step window ms %.3f, kernels %d" % ((b - a) / 1e6, len(sel)))
ev = []
for s, e, n, q in sel: ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, busy1, busy2 = 0, a, 0, 0
idle = []
for t, d in ev:
    if depth >= 1: busy1 += t - last
    if depth >= 2: busy2 += t - last
    if depth == 0 and t - last > 0: idle.append((t - last, last))
    depth += d; last = t
print("at least one kernel running: %.3f ms; two or more: %.3f ms; nothing running: %.3f ms" % (busy1 / 1e6, busy2 / 1e6, sum(i for i, _ in idle) / 1e6))
print("sum of kernel durations %.3f ms" % (sum(e - s for s, e, _, _ in sel) / 1e6))
idle.sort(reverse=True)
ends = sorted(sel, key=lambda r: r[1])
import bisect
endt = [r[1] for r in ends]
print("largest idle gaps (us, at ms into the step, kernel that ended before, kernel that started after):")
starts = sorted(sel)
st = [r[0] for r in starts]
for g, t0 in idle[:25]:
    i = bisect.bisect_right(endt, t0) - 1
    j = bisect.bisect_left(st, t0 + g)
    print("  %7.1f  at %6.2f   %-48s -> %s" % (g / 1e3, (t0 - a) / 1e6, ends[i][2] if i >= 0 else "-", starts[j][2] if j < len(starts) else "-"))
hist = collections.Counter()
for g, _ in idle: hist[min(int(g / 1e3) // 2 * 2, 20)] += g
print("idle time by gap length (us bucket: total ms):", {k: round(v / 1e6, 3) for k, v in sorted(hist.items())})
PY
cat $OUT/summary.txt
