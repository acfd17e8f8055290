#!/usr/bin/env python3
"""usage: tools/isa_compact.py in.s out.lst -- llvm-objdump output -> 'offset: instruction' lines (symbol names and encodings stripped)."""
import re, sys
out = []
base = None
for ln in open(sys.argv[1]):
    m = re.match(r'\s+(\S.*?)\s*//\s*([0-9A-F]{12}):', ln)
    if not m:
        continue
    ins, addr = m.group(1), int(m.group(2), 16)
    if base is None:
        base = addr
    ins = re.sub(r'<[^>]*\+(0x[0-9a-f]+)>', r'-> \1', ins)
    out.append('%05x: %s' % (addr - base, ins))
open(sys.argv[2], 'w').write('\n'.join(out) + '\n')
print(len(out), 'instructions')
