#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 PMC counters (last dispatch of each kernel): pmc_summary.py <dir> [name filter]"""
import csv, glob, sys
from collections import defaultdict
rows = defaultdict(dict)
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for path in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        name = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:60]
        if flt not in name:
            continue
        key = (name, int(r['Dispatch_Id']))
        rows[key][r['Counter_Name']] = rows[key].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
last = {}
for (name, did), c in rows.items():
    if name not in last or did > last[name][0]:
        last[name] = (did, c)
for name, (did, c) in last.items():
    print(name, 'dispatch', did)
    for k in sorted(c):
        print(f'   {k:32s} {c[k]:.4g}')
