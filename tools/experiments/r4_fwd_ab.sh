#!/bin/bash
# Round 4, forward: A/B of library variants on one box, alternating: tools/experiments/r4_fwd_ab.sh <variant> [<variant> ...]
# ("" = the shipped library; variants are build_var/libsunerf_hip_<name>.so from tools/build_variant.sh)
cd "$(dirname "$0")/../.."
export SUNERF_FORWARD_PRECISION=fast
for rep in 1 2 3; do
  for v in "" "$@"; do
    lib=${v:+$PWD/build_var/libsunerf_hip_$v.so}
    SUNERF_HIP_LIB=$lib timeout -k 10 200 python bench.py --mode fwd --steps 4 --warmup 1 --no-cpu-baseline --no-half --no-exact 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd   %-8s' % '${v:-shipped}', '%.2f ms/frame' % d['ms_per_step'])"
    SUNERF_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train %-8s' % '${v:-shipped}', '%.2f ms/step' % d['ms_per_step'], 'render %.2f ms' % d['roofline']['render_kernel']['kernel_ms_hip_events'])"
  done
done
