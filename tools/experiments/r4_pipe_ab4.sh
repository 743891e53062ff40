#!/bin/bash
# Round 4, pipelined backward with a single fp16 W^T (the weight-gradient waves are now the workgroup's critical path): static wave
# priority.  prio2 = s_setprio 1 on the weight-gradient waves (-DPIPE_PRIO=2; +9 % TIME in round 3, when the data-gradient waves
# were critical), prio1 = on the data-gradient waves.
cd "$(dirname "$0")/../.."
run() {
  SUNERF_HIP_LIB=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s' % '$1', '%.2f ms/step' % d['ms_per_step'], 'pipelined kernel %.2f ms' % r['kernel_ms_hip_events'], 'render %.2f ms' % r['render_kernel']['kernel_ms_hip_events'])"
}
for rep in 1 2 3; do
  run shipped ""
  run prio2 $PWD/build_var/libsunerf_hip_prio2.so
  run prio1 $PWD/build_var/libsunerf_hip_prio1.so
done
