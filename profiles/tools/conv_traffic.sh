#!/bin/bash
# Fabric traffic of ONE conv shape, exact request sizes: bytes = TCC_BUBBLE*128 + (RDREQ - BUBBLE - RDREQ_32B)*64 + RDREQ_32B*32 (the expression
# behind rocprofv3's derived read-bandwidth metric), L2 hit rate, writes.  usage: conv_traffic.sh tag B H W Cin Cout k dil
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/traffic_$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CMD="python3 $ROOT/profiles/tools/one_conv.py $*"
# (every pass under `timeout`: a counter set the hardware cannot collect makes rocprofv3 abort and then sit in its signal handler for minutes)
timeout 150 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TCC_BUBBLE_sum --output-format csv -d $OUT/a2 -- $CMD > $OUT/a2.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/a3 -- $CMD > $OUT/a3.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/a4 -- $CMD > $OUT/a4.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1
cd $ROOT
python3 - "$TAG" "$*" <<'PY'
import csv,glob,collections,os,sys
tag=sys.argv[1]
out=os.path.join("gpurun_out","traffic_"+tag)
acc=collections.defaultdict(lambda:[0.0,0])
for f in glob.glob(os.path.join(out,"*","**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_igemm2" not in r["Kernel_Name"]: continue
        a=acc[r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
v={c:x[0]/x[1] for c,x in acc.items()}
B,H,W,Cin,Cout,k,dil=[int(t) for t in sys.argv[2].split()[:7]]
M=B*H*W; alg=(M*Cin+M*Cout+Cout*k*k*Cin)*2
rd=v["TCC_BUBBLE_sum"]*128+(v["TCC_EA0_RDREQ_sum"]-v["TCC_BUBBLE_sum"]-v["TCC_EA0_RDREQ_32B_sum"])*64+v["TCC_EA0_RDREQ_32B_sum"]*32
print(f"{tag}: M={M} {Cin}->{Cout} {k}x{k} d{dil}: algorithmic {alg/1e6:.1f} MB (x {M*Cin*2/1e6:.1f} + y {M*Cout*2/1e6:.1f} + w {Cout*k*k*Cin*2/1e6:.1f}); "
      f"reads {rd/1e6:.1f} MB ({v['TCC_BUBBLE_sum']:.0f} x 128 B + {v['TCC_EA0_RDREQ_sum']-v['TCC_BUBBLE_sum']-v['TCC_EA0_RDREQ_32B_sum']:.0f} x 64 B), "
      f"FETCH_SIZE {v['FETCH_SIZE']*1024/1e6:.1f} MB, writes {v['WRITE_SIZE']*1024/1e6:.1f} MB; L2 requests {v['TCC_REQ_sum']:.0f} hits {v['TCC_HIT_sum']:.0f} "
      f"misses {v['TCC_MISS_sum']:.0f} ({v['TCC_HIT_sum']/v['TCC_REQ_sum']:.3f} hit); sector misses x 64 B = {v['TCC_MISS_sum']*64/1e6:.1f} MB, "
      f"misses per fabric read request {v['TCC_MISS_sum']/max(v['TCC_EA0_RDREQ_sum'],1):.2f} (2 = every request fetches a whole 128-B line: reads = {v['TCC_EA0_RDREQ_sum']*128/1e6:.1f} MB)")
PY
rm -rf $OUT/a $OUT/a2 $OUT/a3 $OUT/a4 $OUT/b
