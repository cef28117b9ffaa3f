#!/bin/bash
# power_watch.sh <seconds> <out>: sample every card's busy %, socket power (hwmon power1_average / power1_input, uW) and current sclk from sysfs every
# 0.25 s -- which card the job runs on, at what power and shader clock (rocm-smi on this pool reports card0 only, often another GPU's idle state).
T=$1; OUT=$2; : > $OUT
end=$(( $(date +%s) + T ))
while [ $(date +%s) -lt $end ]; do
  for c in /sys/class/drm/card[0-9]*; do
    d=$c/device
    [ -f $d/gpu_busy_percent ] || continue
    busy=$(cat $d/gpu_busy_percent 2>/dev/null)
    pw=$(cat $d/hwmon/hwmon*/power1_average 2>/dev/null || cat $d/hwmon/hwmon*/power1_input 2>/dev/null)
    sclk=$(grep '\*' $d/pp_dpm_sclk 2>/dev/null | tr -s ' ' | cut -d' ' -f2)
    echo "$(date +%s.%N | cut -c1-14) $(basename $c) busy=$busy power_uW=$pw sclk=$sclk" >> $OUT
  done
  sleep 0.25
done
