#!/bin/bash
# A/B of two builds of the library on the SAME box, alternating runs: tools/ab_bench.sh <libA.so|""> <libB.so|""> <bench args...>
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for v in "$A" "$B"; do
    SUNERF_HIP_LIB=${v:+$PWD/$v} timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-two-pass --no-small-batch --no-half 2>&1 | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${v:-default}', '%.4e' % d['value'], '%.2f ms' % d['ms_per_step'], 'render %.2f ms' % d['roofline']['kernel_ms_hip_events'])"
  done
done
