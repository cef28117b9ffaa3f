# rocprofv3 kernel trace of a production step, per HIP stream (queue): busy time, first / last kernel, and a 0.5-ms-bucket timeline of which kernel
# family each queue runs -- where the side stream idles, what the main chain is doing meanwhile.   usage (through gpurun): bash profiles/tools/prof_streams.sh [bench args]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_streams; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prod -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-passes "$@" > $OUT/prod.log 2>&1
cd $ROOT
f=$(find $OUT/prod -name "*kernel_trace.csv" | head -1); python3 - "$f" > $OUT/summary.txt <<'PY'
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]
def fam(n):
    n = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))
    m = re.match(r"(\w+)", n)
    base = m.group(1) if m else n[:10]
    short = {"conv_igemm2_kernel": "ig2", "conv1x1_rows_kernel": "rows", "bn_apply_kernel": "bnA", "bn_bwd_apply_kernel": "bnB", "bn_finalize_kernel": "bnF",
             "bn_bwd_finalize_kernel": "bnBF", "bn_bwd_reduce_kernel": "bnBR", "conv_wgrad3_multi_kernel": "WG3", "conv_wgrad2_multi_kernel": "WG2",
             "wgrad_reduce4_multi_kernel": "wred", "head_pass1_kernel": "HP1", "head_pass2_kernel": "HP2", "sgd_multi_kernel": "SGD", "pack_weight_multi_kernel": "pack"}
    return short.get(base, base[:10])
sel = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"]), r["Queue_Id"]) for r in rows if a <= int(r["Start_Timestamp"]) < b]
print("step window %.3f ms, %d kernels" % ((b - a) / 1e6, len(sel)))
qs = sorted({q for *_, q in sel})
for q in qs:
    k = [r for r in sel if r[3] == q]
    print("queue %s: %d kernels, busy %.3f ms, first start %.3f ms, last end %.3f ms  (first %s, last %s)" %
          (q, len(k), sum(e - s for s, e, *_ in k) / 1e6, (k[0][0] - a) / 1e6, (max(e for _, e, *_ in k) - a) / 1e6, k[0][2], k[-1][2]))
W = 500000
nb = (b - a + W - 1) // W
print("timeline, %.1f-ms buckets: per queue the busy fraction and the family with the most time in the bucket" % (W / 1e6))
for i in range(nb):
    t0, t1 = a + i * W, a + (i + 1) * W
    line = "%5.1f ms" % (i * W / 1e6)
    for q in qs:
        acc = collections.Counter()
        for s, e, n, qq in sel:
            if qq == q and e > t0 and s < t1:
                acc[n] += min(e, t1) - max(s, t0)
        tot = sum(acc.values())
        top = acc.most_common(2)
        line += "   | q%s %3d%% %-14s" % (q, round(100 * tot / W), " ".join("%s" % n for n, _ in top))
    print(line)
PY
cat $OUT/summary.txt
