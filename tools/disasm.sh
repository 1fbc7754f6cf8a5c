#!/bin/bash
# usage: tools/disasm.sh <object.o> <out.s>   -- gfx950 disassembly of a hipcc object's device code
set -e
obj=$1; out=$2
tmp=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat $obj
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat --output=$tmp/co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --mcpu=gfx950 $tmp/co > $out
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/co | grep -E "\.name:|vgpr_count|agpr_count|sgpr_count|spill|private_segment_fixed|group_segment_fixed" > $out.meta || true
rm -rf $tmp
