#!/bin/bash
# Round 4: are the hand-off bytes of the pipelined backward written to memory because the L2 writes through, or because the rings
# (3.6 MB of a 4 MB L2) are evicted by the streams passing through?  WRITE_SIZE of the kernel with rings of 16 (shipped) and 12 slots.
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for v in shipped ring12; do
  [ $v = shipped ] && unset SUNERF_HIP_LIB || export SUNERF_HIP_LIB=$R/build_var/libsunerf_hip_$v.so
  rm -rf $R/gpurun_out/ringw_$v
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ringw_$v -- python3 $R/bench.py --no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact --steps 3 --warmup 1 --mode train > $R/gpurun_out/ringw_$v.log 2>&1
  echo -n "$v  "; python3 $R/tools/pmc_summary.py $R/gpurun_out/ringw_$v bwd_pipe_kernel | tr '\n' ' '; echo
done
