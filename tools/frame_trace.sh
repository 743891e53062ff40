#!/bin/bash
# Kernel-trace stats of the full-frame two-pass inference driver (tools/frame_time.py): which launches make up a frame.
R=/root/repo
O=$R/gpurun_out/frame_trace
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/tools/frame_time.py 1024 > $O/run.log 2>&1
cat $O/run.log | tail -3
python3 - "$O" <<'PY'
import csv, glob, sys
p = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(p)))[:14]:
    print('%-70s calls %4s avg %9.1f us total %8.2f ms  %5s %%' % (r['Name'].replace('(anonymous namespace)::', '')[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6, r['Percentage']))
PY
