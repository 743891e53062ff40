#!/bin/bash
# Round 4: what do the weight stream's LDS-DMA instructions cost the forward, and does issuing them one wave at a time help?
# Builds (tools/build_variant.sh): nodma = -DSUNERF_ABL_NO_DMA=1 (timing only), rot = -DSUNERF_DMA_ROTATE=1 (correct results).
# Same box, alternating; inference frame and training step; FAST arithmetic forced.
cd "$(dirname "$0")/../.."
export SUNERF_FORWARD_PRECISION=fast
export SUNERF_BENCH_ABLATION=1
for rep in 1 2; do
  for v in "" nodma rot; do
    lib=${v:+$PWD/build_var/libsunerf_hip_$v.so}
    SUNERF_HIP_LIB=$lib timeout -k 10 200 python bench.py --mode fwd --steps 4 --warmup 1 --no-cpu-baseline --no-half --no-exact 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fwd   %-8s' % '${v:-shipped}', '%.2f ms/frame' % d['ms_per_step'], 'render kernel %.2f ms' % d['roofline']['kernel_ms_hip_events'])"
    SUNERF_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('train %-8s' % '${v:-shipped}', '%.2f ms/step' % d['ms_per_step'], 'render %.2f ms' % d['roofline']['render_kernel']['kernel_ms_hip_events'] if 'render_kernel' in d['roofline'] else '')"
  done
done
