"""Idle intervals of the busiest queue inside one step of a rocprofv3 kernel trace, and what the other queues ran meanwhile.
usage: queue_gaps.py kernel_trace.csv [min_gap_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
mingap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows)
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]      # noqa: E731
# one step: between the last two sgd_multi_kernel launches
sgd = [s for (s, e, n, q) in ks if n.startswith("sgd_multi_kernel")]
t0, t1 = sgd[-2], sgd[-1]
step = [k for k in ks if t0 <= k[0] < t1]
byq = defaultdict(list)
for k in step:
    byq[k[3]].append(k)
busy = {q: sum(e - s for (s, e, n, _) in v) for q, v in byq.items()}
main = max(busy, key=busy.get)
print(f"step window {(t1 - t0) / 1e6:.3f} ms; queues: " + ", ".join(f"{q}: {len(v)} kernels, busy {busy[q] / 1e6:.2f} ms" for q, v in byq.items()))
tot = 0.0
prev_e, prev_n = None, None
for (s, e, n, q) in byq[main]:
    if prev_e is not None and s - prev_e > mingap * 1e3:
        others = defaultdict(float)
        for (s2, e2, n2, q2) in step:
            if q2 != main:
                ov = min(e2, s) - max(s2, prev_e)
                if ov > 0:
                    others[short(n2)] += ov / 1e3
        o = ", ".join(f"{k} {v:.0f}" for k, v in sorted(others.items(), key=lambda kv: -kv[1])[:3])
        print(f"  at {(prev_e - t0) / 1e6:7.3f} ms  gap {(s - prev_e) / 1e3:7.1f} us  after {short(prev_n):40s} before {short(n):40s} | other queues (us): {o}")
        tot += (s - prev_e) / 1e3
    prev_e, prev_n = max(e, prev_e or 0), n
print(f"gaps >= {mingap} us on the busiest queue: {tot / 1e3:.2f} ms")
