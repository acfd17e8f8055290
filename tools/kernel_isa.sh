#!/bin/bash
# usage: tools/kernel_isa.sh lib.so out.s [symbol substring]  -- disassembly of one kernel of a built library
LIB=$1; OUT=$2; SYM=${3:-solve_kernelILi128ELb0ELb0E}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB $T/fat.bin
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle
FULL=$(/opt/rocm/lib/llvm/bin/llvm-readelf -sW $T/k.co | grep FUNC | grep "$SYM" | head -1 | awk "{print \$8}")
/opt/rocm/lib/llvm/bin/llvm-objdump -d --disassemble-symbols=$FULL $T/k.co > $OUT
echo "$FULL: $(grep -c '^\s' $OUT) lines"
