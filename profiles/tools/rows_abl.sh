#!/bin/bash
# rows_abl.sh lib...: serial-schedule kernel stats of the step under each library (timing ablations of conv1x1_rows: outputs meaningless); prints the rows flavours' average launch times
ROOT=$(pwd); OUT=$ROOT/gpurun_out/rows_abl; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  if [ "$lib" = default ]; then unset SIMT_LIB_PATH; else export SIMT_LIB_PATH=$ROOT/$lib; fi
  rm -rf $OUT/x
  SIMT_SINGLE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/x -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-passes > $OUT/$tag.log 2>&1
  f=$(find $OUT/x -name "*kernel_stats.csv" | head -1)
  echo "== $tag"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "conv1x1_rows_kernel<8, 2, 6" in r["Name"] or "conv1x1_rows_kernel<4, 2, 6" in r["Name"]:
        print("  %-48s n=%4s avg %6.1f us" % (r["Name"].replace("(anonymous namespace)::", "")[5:53], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
