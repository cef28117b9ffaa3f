#!/bin/bash
# build profiles/ab_libs/libsimt_<tag><N>.so (git-ignored, travels with gpurun) = product objects with one file compiled at -D<MACRO>=N.   usage: build_variants.sh file MACRO tag N...
set -e
cd /root/repo/simt_amd/csrc
bash build.sh >/dev/null
FILE=$1; MACRO=$2; TAG=$3; shift 3
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops"
for n in "$@"; do
  ( /opt/rocm/bin/hipcc $FLAGS -D$MACRO=$n -c $FILE.hip -o /root/repo/profiles/ab_libs/${TAG}_$n.o 2>/dev/null
    objs=$(ls ../_build/*.o | grep -v /$FILE.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /root/repo/profiles/ab_libs/${TAG}_$n.o -o /root/repo/profiles/ab_libs/libsimt_${TAG}$n.so ) &
done
wait
