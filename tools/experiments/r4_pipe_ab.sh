#!/bin/bash
# Round 4, pipelined backward: (a) does the in-layer stage bound the pipeline?  Build instage0 (-DPIPE_ABL_IN_STAGE=1: that stage
# keeps its protocol and DMA but computes nothing; timing only).  (b) single fp16 W^T in the data gradient (SUNERF_PIPE_HI_ONLY=1)
# on the FINAL kernel.  Same box, alternating; training step of bench.py, the pipelined kernel timed inside the C ABI.
cd "$(dirname "$0")/../.."
export SUNERF_BENCH_ABLATION=1
run() {   # name, lib, env
  SUNERF_HIP_LIB=$2 env $3 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-half --no-two-pass --no-small-batch --no-exact --no-dt 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s' % '$1', '%.2f ms/step' % d['ms_per_step'], 'pipelined kernel %.2f ms' % r['kernel_ms_hip_events'], 'render %.2f ms' % r['render_kernel']['kernel_ms_hip_events'])"
}
for rep in 1 2 3; do
  run hilo "" "SUNERF_PIPE_HI_ONLY=0"
  run instage0 $PWD/build_var/libsunerf_hip_instage0.so "SUNERF_PIPE_HI_ONLY=0"
  run hi_only "" "SUNERF_PIPE_HI_ONLY=1"
  run auto "" "A=1"
done
