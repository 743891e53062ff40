#!/bin/bash
# What does each kind of work in the forward's hidden tiles cost?  Builds of the library with ONE kind of work removed
# (render_fwd.hip, SUNERF_ABL_*; operands stay finite, outputs are wrong) against the shipped build, same box, alternating,
# inference frame 1024^2 x 128, FAST arithmetic forced (the AUTO probe would object to the wrong outputs).
# Build first:  for v in NO_AREAD NO_TRANS NO_L8; do tools/build_variant.sh abl_$v -DSUNERF_ABL_$v=1; done
cd "$(dirname "$0")/.."
export SUNERF_FORWARD_PRECISION=fast
for rep in 1 2; do
  for v in "" abl_NO_AREAD abl_NO_TRANS abl_NO_L8; do
    lib=${v:+$PWD/build_var/libsunerf_hip_$v.so}
    SUNERF_HIP_LIB=$lib timeout -k 10 200 python bench.py --mode fwd --steps 4 --warmup 1 --no-cpu-baseline --no-half 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s' % '${v:-shipped}', '%.2f ms/frame' % d['ms_per_step'], 'render kernel %.2f ms' % d['roofline']['kernel_ms_hip_events'])"
  done
done
