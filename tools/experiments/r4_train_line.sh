#!/bin/bash
# one training-step line of bench.py with the kernel times: tools/experiments/r4_train_line.sh [reps] [env assignments...]
cd "$(dirname "$0")/../.."
reps=${1:-3}; shift
for rep in $(seq $reps); do
  env "$@" timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.2f ms/step %.3e' % (d['ms_per_step'], d['value']), r['kernel'], '%.2f ms' % r['kernel_ms_hip_events'], 'render %.2f ms' % r['render_kernel']['kernel_ms_hip_events'], r.get('weight_precision', ''))"
done
