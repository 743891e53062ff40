#!/bin/bash
# SQ counters of forward-kernel build variants (build_var/lib_<name>.so): tools/pmc_variants.sh name1 name2 ...
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export SUNERF_HIP_LIB=$R/build_var/lib_$v.so
  O=$R/gpurun_out/pmcv/$v
  mkdir -p $O
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $O/a -- python3 $R/bench.py --mode fwd --steps 1 --warmup 1 --no-cpu-baseline --no-two-pass --no-half > $O/a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA \
    --output-format csv -d $O/b -- python3 $R/bench.py --mode fwd --steps 1 --warmup 1 --no-cpu-baseline --no-two-pass --no-half > $O/b.log 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $O render_fwd
done
