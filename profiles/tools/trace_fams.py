"""trace_fams.py <kernel_trace.csv> [rows]: one production step (between the last two sgd_multi launches) of a rocprofv3 kernel trace:
per queue the busy time and the kernel families by total time (count, average, maximum)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]
sel = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]


def fam(n):
    n = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))
    m = re.match(r"([\w]+(<[^>]*>)?)", n)
    return m.group(1)


print("step window %.3f ms, %d kernels" % ((b - a) / 1e6, len(sel)))
for q in sorted({r["Queue_Id"] for r in sel}):
    acc = collections.defaultdict(lambda: [0, 0, 0])
    for r in sel:
        if r["Queue_Id"] != q:
            continue
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        f = fam(r["Kernel_Name"])
        acc[f][0] += d
        acc[f][1] += 1
        acc[f][2] = max(acc[f][2], d)
    print("queue %s: busy %.2f ms" % (q, sum(v[0] for v in acc.values()) / 1e6))
    for f, v in sorted(acc.items(), key=lambda x: -x[1][0])[:top]:
        print("   %-52s %7.3f ms  n=%3d avg %6.1f us max %6.1f" % (f[:52], v[0] / 1e6, v[1], v[0] / v[1] / 1e3, v[2] / 1e3))


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def inter(a_, b_):
    i = j = 0
    t = 0
    while i < len(a_) and j < len(b_):
        s, e = max(a_[i][0], b_[j][0]), min(a_[i][1], b_[j][1])
        if e > s:
            t += e - s
        if a_[i][1] < b_[j][1]:
            i += 1
        else:
            j += 1
    return t


qs = sorted({r["Queue_Id"] for r in sel})
u = {q: union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel if r["Queue_Id"] == q]) for q in qs}
allu = union([(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel])
print("at least one kernel running %.2f ms of the window" % (sum(e - s for s, e in allu) / 1e6), end="")
if len(qs) >= 2:
    print("; both of the first two queues %.2f ms" % (inter(u[qs[0]], u[qs[1]]) / 1e6))
else:
    print()
