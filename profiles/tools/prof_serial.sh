set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_s; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SIMT_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-passes > $OUT/serial.log 2>&1
cd $ROOT
f=$(find $OUT/serial -name "*kernel_stats.csv" | head -1); cp $f $OUT/serial_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
head -3 $OUT/serial_kernel_stats.csv
