#!/bin/bash
# attrib_conv.sh: the attribution runs of VERDICT r4 #2 on the -DSIMT_ABLATION library (through gpurun from the repo root): stamps per mode, then
# the SQ counter passes (rocprofv3 --pmc, counters in their own runs with --kernel-trace only) of the same launches.  Output: gpurun_out/attrib/.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/attrib; rm -rf $OUT; mkdir -p $OUT
LIB=$ROOT/profiles/ab_libs/libsimt_abl.so      # (csrc/build.sh ABLATION=1, copied here: simt_amd/libsimt_hip_abl.so is gpurun-ignored)
for m in 0 2 1; do
  SIMT_CONV2_MODE=$m SIMT_CONV2_KSTAMP=1 python3 profiles/tools/attrib_conv.py $LIB >> $OUT/stamps.txt 2>$OUT/stamps_$m.err
  SIMT_CONV2_MODE=$m SIMT_CONV2_ABL0=1 python3 profiles/tools/conv_modes.py $LIB >> $OUT/modes.txt 2>>$OUT/modes.err
done
cat $OUT/stamps.txt $OUT/modes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $OUT/counters_avail.txt | sort -u > $OUT/sq_names.txt
pmc() {   # pmc <tag> <mode> counters...
  tag=$1; m=$2; shift 2
  SIMT_LIB_PATH=$LIB SIMT_CONV2_MODE=$m SIMT_CONV2_ABL0=1 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${tag}_m$m -- python3 $ROOT/profiles/tools/one_conv.py 4 97 97 256 256 3 2 > $OUT/pmc_${tag}_m$m.log 2>&1
}
for m in 0 2; do
  pmc a $m SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE
  pmc b $m SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
  pmc c $m SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY
  pmc d $m SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA
  pmc e $m SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES
done
cd $ROOT
python3 - $OUT > $OUT/pmc_summary.txt 2>&1 <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for m in (0, 2):
    tot = collections.OrderedDict()
    for d in sorted(glob.glob(os.path.join(out, f"pmc_*_m{m}"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc, n = collections.defaultdict(float), collections.defaultdict(int)
            for r in csv.DictReader(open(f)):
                if "conv_igemm2" not in r["Kernel_Name"]:
                    continue
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            for k in acc:
                tot[k] = acc[k] / max(n[k], 1)
    print(f"mode {m}: per launch (average over the conv launches of one_conv.py: 3x3 d2 256->256, M = 37 636)")
    for k, v in tot.items():
        print(f"   {k:32s} {v:16.0f}")
PY
cat $OUT/pmc_summary.txt
find $OUT -name "*.csv" -size +2M -delete
