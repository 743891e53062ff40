#!/bin/bash
# Per-kernel time of tools/pipe_check.py (rocprofv3 --kernel-trace --stats); usage on the GPU box: tools/pipe_prof.sh TAG [pipe_check args]
R=/root/repo
TAG=$1; shift
OUT=$R/gpurun_out/pp_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/pipe_check.py "$@" > $OUT/run.log 2>&1
f=$(find $OUT -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
cp "$f" $R/gpurun_out/pp_$TAG.csv
tail -3 $OUT/run.log
