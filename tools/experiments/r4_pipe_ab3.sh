#!/bin/bash
# Round 4, pipelined backward, third A/B: prev = hand-off protocol on weight-gradient wave 4, cos pieces on waves 5 - 7 (commit
# "db summed in fp32 ..."); shipped = protocol on data-gradient wave 0, every data wave fetches the cos fragments of its own tile,
# the weight-gradient waves two H pieces each.  Single fp16 W^T (measured policy) in both.
cd "$(dirname "$0")/../.."
run() {
  SUNERF_HIP_LIB=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s' % '$1', '%.2f ms/step' % d['ms_per_step'], 'pipelined kernel %.2f ms' % r['kernel_ms_hip_events'], 'render %.2f ms' % r['render_kernel']['kernel_ms_hip_events'])"
}
for rep in 1 2 3; do
  run prev $PWD/build_var/libsunerf_hip_prev.so
  run shipped ""
done
