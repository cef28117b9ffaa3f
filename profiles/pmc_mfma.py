#!/usr/bin/env python3
"""Per-kernel MFMA utilisation from ONE rocprofv3 PMC pass (counters in their own run, with --kernel-trace only):

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <out>/mfma \
      -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-passes
  python3 profiles/pmc_mfma.py <out>/mfma [<out>/mfma/..kernel_trace.csv] > profiles/rNN_pmc_mfma.json

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts shader cycles in which a SIMD's matrix pipe is
busy, summed over the SIMDs (= 16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16); GRBM_GUI_ACTIVE is summed over the 8 XCDs.
mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs): the fraction of all SIMD-cycles of the dispatch spent in the matrix pipe.
`mfma_busy_check` = busy cycles / (16 * algorithmic MFMA count) where the caller passes the kernel's MFMA count: 1.0 confirms the unit."""
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

SIMDS = 256 * 4
NAMES = ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")


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
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            c = r.get("Counter_Name")
            if c not in NAMES:
                continue
            k = kernel_name(r["Kernel_Name"])
            acc[k][c] += float(r["Counter_Value"])
            disp[k].add(r.get("Dispatch_Id"))
    out = {"note": "rocprofv3 --kernel-trace --pmc " + " ".join(NAMES) + " over `bench.py --steps 2 --warmup 1` (own pass); per-dispatch averages "
                   "per kernel.  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): share of all SIMD-cycles of the dispatch "
                   "with the matrix pipe busy (profiled passes run at a lower clock than un-profiled ones: a cycle ratio, not a time).",
           "lib_sha256": lib_sha256(), "kernels": {}}
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]):
        n = max(1, len(disp[k]))
        a = {c: acc[k][c] / n for c in NAMES}
        gui = a["GRBM_GUI_ACTIVE"] / 8.0
        out["kernels"][k] = {"dispatches": n, **{c + "_avg": round(v, 1) for c, v in a.items()},
                             "mfma_util": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * SIMDS), 4) if gui > 0 else None,
                             "mfma_busy_over_sq_busy": round(a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["SQ_BUSY_CYCLES"], 4) if a["SQ_BUSY_CYCLES"] else None}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
