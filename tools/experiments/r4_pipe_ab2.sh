#!/bin/bash
# Round 4, pipelined backward, second A/B (single fp16 W^T chosen by the measured policy in all three): prebias = db from the
# rounded dZ by v_dot2 on the weight-gradient waves (round 3); shipped = db summed in fp32 by the data-gradient waves / the prologue;
# cosdata = shipped + the cos fragments fetched by the data wave that reads them (-DPIPE_COS_ON_DATA=1).
cd "$(dirname "$0")/../.."
run() {
  SUNERF_HIP_LIB=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s' % '$1', '%.2f ms/step' % d['ms_per_step'], 'pipelined kernel %.2f ms' % r['kernel_ms_hip_events'], 'render %.2f ms' % r['render_kernel']['kernel_ms_hip_events'], r.get('weight_precision', ''))"
}
for rep in 1 2 3; do
  run prebias $PWD/build_var/libsunerf_hip_prebias.so
  run shipped ""
  run cosdata $PWD/build_var/libsunerf_hip_cosdata.so
done
