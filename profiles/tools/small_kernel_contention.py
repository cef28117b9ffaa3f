"""For every launch of the small dependent kernels of the backward chain (bn_bwd_finalize, bn_finalize) in a rocprofv3 kernel trace of the step:
its duration, and which kernel of the OTHER queue was running when it started.  usage: small_kernel_contention.py kernel_trace.csv"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in rows]
ks.sort()
names = ("bn_bwd_finalize_kernel", "bn_finalize_kernel", "bn_bwd_apply_kernel", "bn_apply_kernel")
for nm in names:
    by = defaultdict(list)
    for (s, e, n, q) in ks:
        if not n.startswith(nm) and nm not in n[:40]:
            continue
        other = "-"
        for (s2, e2, n2, q2) in ks:
            if q2 != q and s2 <= s < e2:
                other = n2.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:40]
                break
        by[other].append((e - s) / 1e3)
    print(nm)
    for o, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
        v.sort()
        print(f"   beside {o:42s} n={len(v):4d}  median {v[len(v) // 2]:7.1f} us  mean {sum(v) / len(v):7.1f}  max {v[-1]:7.1f}")

# every kernel name: launches that started while a weight-gradient kernel ran on another queue against the rest
print("\nall kernels: started beside conv_wgrad* on another queue vs not (mean us), extra ms over the trace")
wg = [(s, e, q) for (s, e, n, q) in ks if n.startswith("conv_wgrad") or "conv_wgrad" in n[:30]]
agg = defaultdict(lambda: [[], []])
for (s, e, n, q) in ks:
    if "conv_wgrad" in n[:30] or "wgrad_reduce" in n[:30]:
        continue
    beside = any(q2 != q and s2 <= s < e2 for (s2, e2, q2) in wg)
    agg[n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]][1 if beside else 0].append((e - s) / 1e3)
out = []
for n, (a, b) in agg.items():
    if a and b:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        out.append(((mb - ma) * len(b) / 1e3, n, len(a), ma, len(b), mb))
for extra, n, na, ma, nb, mb in sorted(out, reverse=True)[:16]:
    print(f"   {n:60s} alone n={na:4d} {ma:7.1f} us | beside n={nb:4d} {mb:7.1f} us | extra {extra:6.2f} ms")

# the same split against ANY kernel of another queue (forward: the frozen network's launches on the side stream)
print("\nall kernels: started while any kernel ran on another queue vs not")
agg = defaultdict(lambda: [[], []])
for (s_, e, n, q) in ks:
    beside = any(q2 != q and s2 <= s_ < e2 for (s2, e2, n2, q2) in ks)
    nm = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
    agg[nm][1 if beside else 0].append((e - s_) / 1e3)
out = []
for n, (a, b) in agg.items():
    if a and b:
        ma, mb = sum(a) / len(a), sum(b) / len(b)
        out.append(((mb - ma) * len(b) / 1e3, n, len(a), ma, len(b), mb))
for extra, n, na, ma, nb, mb in sorted(out, reverse=True)[:14]:
    print(f"   {n:60s} alone n={na:4d} {ma:7.1f} us | beside n={nb:4d} {mb:7.1f} us | extra {extra:6.2f} ms")
