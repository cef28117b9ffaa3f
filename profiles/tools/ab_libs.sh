#!/bin/bash
# ab_libs.sh [-r rounds] [-t] lib1.so[:ENV=V[,ENV2=V2]] lib2.so ...: the default bench workload (BASELINE configs[1]) under each build of the library, alternating
# rounds on ONE box (boxes of the pool differ by +-0.5 ms); "default" = the shipped simt_amd/libsimt_hip.so.  -t: also one rocprofv3 kernel
# trace per library with the per-family time of the main queue (profiles/tools/trace_fams.py).   Through gpurun from the repo root.
R=3; TR=0
while getopts "r:t" o; do case $o in r) R=$OPTARG;; t) TR=1;; esac; done; shift $((OPTIND-1))
ROOT=$(pwd); OUT=$ROOT/gpurun_out/ab_libs; mkdir -p $OUT
for r in $(seq 1 $R); do
  for lib in "$@"; do
    envs=""; case "$lib" in *:*) envs=${lib#*:}; lib=${lib%%:*};; esac
    tag=$(basename $lib .so)${envs:+:$envs}
    if [ "$lib" = default ]; then unset SIMT_LIB_PATH; else export SIMT_LIB_PATH=$ROOT/$lib; fi
    EV=$(echo $envs | tr ',' ' ')
    env $EV python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra-passes 2>$OUT/last.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('%-28s round $r  %.3f ms/step (median %.3f)  %.1f images/s' % ('$tag', d['ms_per_step'], d['ms_per_step_median'], d['value']))"
  done
done
if [ $TR = 1 ]; then
  for lib in "$@"; do
    envs=""; case "$lib" in *:*) envs=${lib#*:}; lib=${lib%%:*};; esac
    tag=$(basename $lib .so)${envs:+:$envs}
    if [ "$lib" = default ]; then unset SIMT_LIB_PATH; else export SIMT_LIB_PATH=$ROOT/$lib; fi
    EV=$(echo $envs | tr ',' ' ')
    rm -rf $OUT/tr_x; (cd /tmp && export $EV TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_x -- python3 $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-extra-passes > $OUT/tr_x.log 2>&1)
    f=$(find $OUT/tr_x -name "*kernel_trace.csv" | head -1)
    echo "== $tag"; python3 profiles/tools/trace_fams.py "$f" 14
    rm -rf $OUT/tr_x
  done
fi
