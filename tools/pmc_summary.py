#!/usr/bin/env python3
"""Turn the two PMC passes (tools/pmc.sh fetch* FETCH_SIZE / write* WRITE_SIZE of tools/prof_solve.py) into
profiles/rNN_pmc_traffic.json: per-kernel counter sums in KB and the solve kernel's HBM bytes per covered
pixel-step (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
usage: tools/pmc_summary.py <fetch_dir> <write_dir> <covered_pixel_steps> <out.json>"""
import collections
import csv
import glob
import json
import re
import sys


def sums(d, counter):
    acc = collections.defaultdict(float)
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] == counter:
                m = re.search(r'(\w+)(<|\()', r['Kernel_Name'].replace('(anonymous namespace)::', ''))
                acc[m.group(1) if m else r['Kernel_Name'][:40]] += float(r['Counter_Value'])
    return dict(acc)


fetch, write = sums(sys.argv[1], 'FETCH_SIZE'), sums(sys.argv[2], 'WRITE_SIZE')
cov = int(sys.argv[3])
fb, wb = fetch['solve_kernel'] * 1024 * 2, write['solve_kernel'] * 1024
out = {
    'command': 'rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/prof_solve.py --images 1024 --steps 300 --reps 1 --grid',
    'note': 'FETCH_SIZE doubled: on gfx950 it tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM section)',
    'counters_KB': {'FETCH_SIZE_KB': fetch, 'WRITE_SIZE_KB': write},
    'solve_kernel': {'covered_pixel_steps': cov, 'fetch_bytes_x2_corrected': fb, 'write_bytes': wb,
                     'hbm_bytes_per_covered_pixel_step': (fb + wb) / cov, 'algorithmic_bytes_per_covered_pixel_step': 516},
}
json.dump(out, open(sys.argv[4], 'w'), indent=1)
print(out['solve_kernel'])
