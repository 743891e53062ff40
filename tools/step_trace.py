#!/usr/bin/env python3
"""Lists the kernels of the last training step in a rocprofv3 kernel trace (development aid): name, duration, gap to
the previous kernel."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last occurrence of the optimiser kernel closes a step; the one before opens it
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
lo, hi = idx[-2] + 1, idx[-1] + 1
prev_end = int(rows[lo - 1]['End_Timestamp'])
total_busy = 0
for r in rows[lo:hi]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '')
    print(f'{(st - prev_end) / 1e3:8.1f} us gap  {(en - st) / 1e3:9.1f} us  {name[:90]}')
    total_busy += en - st
    prev_end = en
span = int(rows[hi - 1]['End_Timestamp']) - int(rows[lo - 1]['End_Timestamp'])
print(f'step span {span / 1e3:.1f} us, kernels busy {total_busy / 1e3:.1f} us, {hi - lo} launches')
