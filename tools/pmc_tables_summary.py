#!/usr/bin/env python3
"""HBM traffic of the two kernels of the tables-form solve from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate
runs of `tools/prof_solve.py --images N --steps S --reps 1`), per unit: bytes per covered pixel-step for solve_kernel,
bytes per job for bank_table_kernel.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes,
MI355X_MICROARCH.md).  The covered pixel-steps come from the run's own log line ("covered pixel-steps <n>").
usage: tools/pmc_tables_summary.py <fetch_dir> <write_dir> <fetch_log> <out.json> <images>"""
import json
import re
import sys

sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.abspath(__file__)))
from pmc_kernels import sums  # noqa: E402

f, fc = sums(sys.argv[1], 'FETCH_SIZE')
w, wc = sums(sys.argv[2], 'WRITE_SIZE')
cov = float(re.search(r'covered pixel-steps (\d+)', open(sys.argv[3]).read()).group(1))
images = int(sys.argv[5])
out = {'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/prof_solve.py --images %d --steps 300 --reps 1' % images,
       'note': 'bytes; FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024; one launch of each kernel', 'covered_pixel_steps': cov, 'images': images}
for k in ('solve_kernel', 'bank_table_kernel'):
    if k not in f:   # tables on demand: no bank_table_kernel
        continue
    fb, wb = f[k] * 2048, w.get(k, 0.0) * 1024
    out[k] = {'calls': fc[k], 'fetch_bytes': fb, 'write_bytes': wb, 'hbm_bytes': fb + wb}
out['solve_kernel']['hbm_bytes_per_covered_pixel_step'] = out['solve_kernel']['hbm_bytes'] / cov
if 'bank_table_kernel' in out:
    out['bank_table_kernel']['hbm_bytes_per_job'] = out['bank_table_kernel']['hbm_bytes'] / images
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print(json.dumps(out, indent=1))
