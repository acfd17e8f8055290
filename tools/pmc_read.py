#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel name, counter sums over dispatches."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for fn in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
        for r in csv.DictReader(open(fn)):
            k = r['Kernel_Name'][:60]
            acc[k][r['Counter_Name']] += float(r['Counter_Value']); nd[k].add(r['Dispatch_Id'])
        for k, v in acc.items():
            if 'solve_kernel' in k or 'bank_table' in k or 'grid_kernel' in k or 'clutter' in k or 'grid_mfma' in k:
                print(fn.split('/')[-3], k, 'dispatches', len(nd[k]))
                for c, x in v.items():
                    print(f'    {c:28s} {x:.4g}')
