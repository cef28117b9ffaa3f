#!/bin/bash
# disasm.sh <file.hip|lib.so|file.o> <out.s>: gfx950 code object of a HIP source (compiled with the product flags), of an object file or of the
# shipped library, disassembled with llvm-objdump (kernel symbols, instructions; no addresses).
set -e
src=$1; out=$2
LLVM=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
case "$src" in
  *.hip)
    FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA"
    (cd "$(dirname "$src")" && /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -c "$(basename "$src")" -o $tmp/dev.o 2>/dev/null)
    $LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co ;;
  *)
    $LLVM/clang-offload-bundler --unbundle --type=o --input="$src" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co 2>/dev/null \
      || (cd $tmp && $LLVM/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin "$src" /dev/null 2>/dev/null; \
          $LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co) ;;
esac
$LLVM/llvm-objdump -d $tmp/dev.co | sed -E 's/\/\/ [0-9A-F]+:.*$//; s/[[:space:]]+$//' > "$out"
rm -rf $tmp
