#!/bin/bash
# VGG (configs[4]) dominant conv: where do the 5.4x FETCH bytes come from?  L2 requests / hits / misses and the EA read requests by destination.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r4_vgg_pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --model vgg --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra-passes"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
cd $ROOT
python3 - <<'PY'
import csv,glob,collections,os
out=os.path.join("gpurun_out","r4_vgg_pmc")
for sub in ("a","b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
    for f in glob.glob(os.path.join(out,sub,"**","*counter_collection.csv"),recursive=True):
        for r in csv.DictReader(open(f)):
            a=acc[r["Kernel_Name"]][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
    for k,v in sorted(acc.items(), key=lambda kv:-sum(x[0] for x in kv[1].values()))[:8]:
        print(sub, k[:70], {c:(round(x[0]/x[1]),x[1]) for c,x in v.items()})
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
