#!/bin/bash
# usage: tools/kernel_resources.sh [lib.so] [name filter]  -- VGPRs / scratch / LDS of the kernels in a built library
LIB=${1:-nemo_amd/lib/libnemo_hip.so}; FILT=${2:-solve_kernel}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=<(/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB /dev/stdout) --output=$T/k.co --unbundle 2>/dev/null || { /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB $T/fat.bin; /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co --unbundle; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk)
    if not name or '$FILT' not in name.group(1): continue
    g=lambda k:(re.search(r'\.'+k+r':\s+(\d+)',blk) or [0,'?'])[1]
    print(name.group(1)[:90],'vgpr',g('vgpr_count'),'sgpr',g('sgpr_count'),'scratch',g('private_segment_fixed_size'),'spill',g('vgpr_spill_count'))
"
