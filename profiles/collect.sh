#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):   bash profiles/collect.sh r03
# Four rocprofv3 passes of the SAME bench command (kernel trace + stats on the production two-stream schedule, the same on the
# serial schedule that bench.py's roofline leg measures under, and the two PMC passes -- counters in their own runs, with
# --kernel-trace only, as MI355X_MICROARCH.md prescribes), then the default bench line.  Summaries land in gpurun_out/<tag>/;
# copy the *_kernel_stats.csv / *_pmc_traffic.json / *_bench_n1.json you want judged into profiles/.
set -u
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-passes"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prod -- $CMD > $OUT/prod.log 2>&1
SIMT_SINGLE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- $CMD > $OUT/serial.log 2>&1
CMD2="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-passes"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD2 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD2 > $OUT/write.log 2>&1
# MFMA utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation"): its own counter pass, program directly after `--`
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $CMD2 > $OUT/mfma.log 2>&1
cd $ROOT
python3 profiles/pmc_traffic.py $OUT/fetch $OUT/write > $OUT/pmc_traffic.json 2> $OUT/pmc.err
python3 profiles/pmc_mfma.py $OUT/mfma > $OUT/pmc_mfma.json 2> $OUT/pmc_mfma.err
for d in prod serial; do f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv; f=$(find $OUT/$d -name "*domain_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_domain_stats.csv; done
# keep the merged output small: the raw traces are large
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
head -5 $OUT/prod_kernel_stats.csv; head -c 300 $OUT/bench.json
