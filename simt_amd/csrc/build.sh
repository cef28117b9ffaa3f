#!/bin/bash
# Build libsimt_hip.so for gfx950 (in-tree; the .so is git-ignored but travels with gpurun snapshots).
set -e
cd "$(dirname "$0")"
OUT=../libsimt_hip.so
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# -packed-fp32-ops: v_pk_*_f32 cannot issue in the shadow of an MFMA (profiles/microbench/mixbench2.hip: one per MFMA costs +77 %), plain VALU can
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops"
# ABLATION=1: also compile the timing-ablation instantiations and csrc/experiments/*.hip (round-3 experiments on the conv kernel: loader
# waves, weights-direct, role-split conv_igemm3; outputs of the timing modes are meaningless; never shipped).  OUT=path BUILD=dir: where to.
SRCS=(*.hip)
BUILD=${BUILD:-../_build}
if [ "${ABLATION:-0}" = "1" ]; then FLAGS="${FLAGS/-std=c++17/-std=c++20} -DSIMT_ABLATION"; SRCS+=(experiments/*.hip); BUILD=${BUILD}_abl; OUT=${OUT_ABL:-../libsimt_hip_abl.so}; fi
mkdir -p $BUILD
objs=()
pids=()
for f in "${SRCS[@]}"; do
  b=$(basename "$f")
  o=$BUILD/${b%.hip}.o
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ conv2_common.h -nt "$o" ] || [ conv2_epilogue.h -nt "$o" ] || [ ../../include/simt_hip.h -nt "$o" ]; then
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $OUT
echo "built $OUT"
