#!/usr/bin/env python3
"""Per-kernel call count, median and mean duration from a rocprofv3 --kernel-trace directory (kernel_trace.csv): the --stats
averages are polluted by the tiny probe launches of the measured policies (144-ray forward probes, 64-ray backward probes) and by
the cold first call.  Usage: kernel_medians.py <dir> [min_us]"""
import csv, glob, statistics, sys
from collections import defaultdict
dur = defaultdict(list)
for path in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        dur[r['Kernel_Name'].replace('(anonymous namespace)::', '')].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
floor = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows = sorted(dur.items(), key=lambda kv: -sum(kv[1]))
print(f'{"kernel":90s} {"calls":>6s} {"median us":>11s} {"mean us":>11s} {"total ms":>10s}')
for name, v in rows:
    if max(v) < floor:
        continue
    print(f'{name[:90]:90s} {len(v):6d} {statistics.median(v):11.1f} {sum(v) / len(v):11.1f} {sum(v) / 1e3:10.2f}')
