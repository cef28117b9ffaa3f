"""trace_gaps.py <kernel_trace.csv> [n]: idle gaps of the main queue inside one production step (between the last two sgd_multi launches): total
per phase and the n largest with the kernels on either side -- what an event record / cross-queue wait / host stall costs the critical chain."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]
sel = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]
qs = {}
for r in sel:
    qs[r["Queue_Id"]] = qs.get(r["Queue_Id"], 0) + 1
main = max(qs, key=qs.get)


def fam(nm):
    nm = re.sub(r"^void ", "", nm.replace("(anonymous namespace)::", ""))
    m = re.match(r"([\w]+(<[^>]*>)?)", nm)
    return m.group(1)[:40]


k = [(int(r["Start_Timestamp"]) - a, int(r["End_Timestamp"]) - a, fam(r["Kernel_Name"])) for r in sel if r["Queue_Id"] == main]
gaps = [(k[i + 1][0] - k[i][1], k[i][1], k[i][2], k[i + 1][2]) for i in range(len(k) - 1) if k[i + 1][0] > k[i][1]]
print(f"main queue {main}: {len(k)} kernels, {len(gaps)} gaps, {sum(g[0] for g in gaps) / 1e3:.1f} us idle in a {(b - a) / 1e6:.2f}-ms step")
hist = {}
for g in gaps:
    key = "<1us" if g[0] < 1000 else "1-3us" if g[0] < 3000 else "3-8us" if g[0] < 8000 else "8-20us" if g[0] < 20000 else ">20us"
    h = hist.setdefault(key, [0, 0])
    h[0] += 1
    h[1] += g[0]
for key in ("<1us", "1-3us", "3-8us", "8-20us", ">20us"):
    if key in hist:
        print(f"   {key:7s} n={hist[key][0]:4d} total {hist[key][1] / 1e3:8.1f} us")
for g in sorted(gaps, reverse=True)[:n]:
    print(f"   {g[0] / 1e3:7.1f} us at {g[1] / 1e6:7.3f} ms   {g[2]:40s} -> {g[3]}")
