#!/bin/bash
# Round 4: the forward's weight traffic as POWER.  Inference frame (FAST arithmetic forced, results of the ablations meaningless):
# shipped; nodma = ring filled once (-DSUNERF_ABL_NO_DMA=1: no L2 -> LDS stream); noaread = weight operands not re-read from LDS
# (-DSUNERF_ABL_NO_AREAD=1: the first group's stay in registers); noboth = both.  GRBM_GUI_ACTIVE of the same runs gives the clock.
R=/root/repo
cd /tmp && export TMPDIR=/tmp
export SUNERF_FORWARD_PRECISION=fast SUNERF_BENCH_ABLATION=1
for v in shipped nodma noaread noboth; do
  [ $v = shipped ] && unset SUNERF_HIP_LIB || export SUNERF_HIP_LIB=$R/build_var/libsunerf_hip_$v.so
  line=$(timeout -k 10 200 python3 $R/bench.py --mode fwd --steps 4 --warmup 1 --no-cpu-baseline --no-half --no-exact 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f ms/frame, kernel %.2f ms' % (d['ms_per_step'], d['roofline']['kernel_ms_hip_events']))")
  rm -rf $R/gpurun_out/fwdpow_$v
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/fwdpow_$v -- python3 $R/bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --no-half --no-exact > $R/gpurun_out/fwdpow_$v.log 2>&1
  echo "$v: $line; $(python3 $R/tools/pmc_summary.py $R/gpurun_out/fwdpow_$v 'render_fwd_kernel<256, 0, true' | tr '\n' ' ')"
done
