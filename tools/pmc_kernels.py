#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command):
calls, bytes per call.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md).
usage: tools/pmc_kernels.py <fetch_dir> <write_dir> <out.json> [command text]"""
import collections
import csv
import glob
import json
import re
import sys


def sums(d, counter):
    acc, calls = collections.defaultdict(float), collections.defaultdict(int)
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if r['Counter_Name'] == counter:
                m = re.search(r'(\w+)(<|\()', r['Kernel_Name'].replace('(anonymous namespace)::', ''))
                k = m.group(1) if m else r['Kernel_Name'][:40]
                acc[k] += float(r['Counter_Value'])
                calls[k] += 1
    return acc, calls


if __name__ == '__main__':
    f, fc = sums(sys.argv[1], 'FETCH_SIZE')
    w, wc = sums(sys.argv[2], 'WRITE_SIZE')
    out = {'command': sys.argv[4] if len(sys.argv) > 4 else '', 'note': 'bytes; FETCH_SIZE x 1024 x 2 (gfx950 correction), WRITE_SIZE x 1024', 'kernels': {}}
    for k in sorted(f, key=lambda k: -(f[k] * 2 + w.get(k, 0))):
        fb, wb = f[k] * 2048, w.get(k, 0.0) * 1024
        if fb + wb < 1e8:
            continue
        out['kernels'][k] = {'calls': fc[k], 'fetch_bytes_total': fb, 'write_bytes_total': wb,
                             'hbm_bytes_per_call': (fb + wb) / max(fc[k], 1)}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in out['kernels'].items():
        print(f"{k:40s} calls {v['calls']:4d}  per call {v['hbm_bytes_per_call'] / 1e9:9.3f} GB")
