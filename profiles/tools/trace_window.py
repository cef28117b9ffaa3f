"""trace_window.py <kernel_trace.csv> <t0_ms> <t1_ms>: the kernels of one production step (window between the last two sgd_multi launches) that
start in [t0, t1) ms of the step, both queues, in start order: start, end, duration, queue, grid, kernel family."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
t0, t1 = float(sys.argv[2]) * 1e6, float(sys.argv[3]) * 1e6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sg = [int(r["End_Timestamp"]) for r in rows if "sgd_multi" in r["Kernel_Name"]]
a, b = sg[-2], sg[-1]


def fam(n):
    n = re.sub(r"^void ", "", n.replace("(anonymous namespace)::", ""))
    m = re.match(r"([\w]+(<[^>]*>)?)", n)
    return m.group(1)[:44]


for r in rows:
    s, e = int(r["Start_Timestamp"]) - a, int(r["End_Timestamp"]) - a
    if a <= int(r["Start_Timestamp"]) < b and t0 <= s < t1:
        print(f"q{r['Queue_Id']} {s / 1e3:9.1f} -> {e / 1e3:9.1f} ({(e - s) / 1e3:6.1f})  wgs={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):>6} {fam(r['Kernel_Name'])}")
