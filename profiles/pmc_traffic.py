#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY, with --kernel-trace
only), as MI355X_MICROARCH.md's HBM section prescribes:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <out>/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d <out>/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
  python3 profiles/pmc_traffic.py <out>/fetch <out>/write > profiles/rNN_pmc_traffic.json

Units: both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch (averaged per kernel name)."""
import collections
import csv
import glob
import json
import os
import sys


def lib_sha256():
    """sha256 of the shipped libsimt_hip.so the counters were collected on: bench.py emits `traffic` / `mfma_busy` from this file only
    when it matches the library it has loaded (a profile of an older build must not decorate a newer kernel's line)."""
    import hashlib
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.environ.get("SIMT_LIB_PATH") or os.path.join(here, "..", "simt_amd", "libsimt_hip.so")
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


def kernel_name(full):
    """'void conv_igemm2_kernel<256, 5, 3>(Conv2KArgs)' -> 'void conv_igemm2_kernel<256, 5, 3>': cut at the first '('
    outside template brackets ('(anonymous namespace)' prefixes and function-pointer template arguments contain parentheses too)."""
    depth = 0
    for i, ch in enumerate(full):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and i > 0 and not full[:i].rstrip().endswith("void"):
            return full[:i].strip()
    return full.strip()


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) over `bench.py --steps 2 "
                   "--warmup 1`; per-dispatch averages per kernel. Correction per MI355X_MICROARCH.md HBM section: FETCH_SIZE counts 1/2 "
                   "of wide coalesced reads on gfx950 -> bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.",
           "lib_sha256": lib_sha256(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
        fa = f[0] / f[1] if f[1] else 0.0
        wa = w[0] / w[1] if w[1] else 0.0
        name = kernel_name(k)
        out["kernels"][name] = {"dispatches": max(f[1], w[1]), "FETCH_SIZE_KB_avg": round(fa, 1), "WRITE_SIZE_KB_avg": round(wa, 1),
                                "hbm_bytes_per_launch_corrected": int((2 * fa + wa) * 1024)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
