#!/bin/bash
# Cycles, clock and wait split of the inference forward for the shipped build and energy-ablation builds
# (tools/energy_ablation.sh): tools/pmc_ablation.sh "" abl_NO_AREAD ...      ("" = shipped)
R=/root/repo
cd /tmp && export TMPDIR=/tmp
export SUNERF_FORWARD_PRECISION=fast
export SUNERF_BENCH_ABLATION=1
for v in "$@"; do
  if [ -n "$v" ]; then export SUNERF_HIP_LIB=$R/build_var/libsunerf_hip_$v.so; else unset SUNERF_HIP_LIB; fi
  O=$R/gpurun_out/pmca/${v:-shipped}
  rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --mode fwd --steps 2 --warmup 1 --no-cpu-baseline --no-half --no-exact > $O/t.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $O/a -- python3 $R/bench.py --mode fwd --steps 1 --warmup 1 --no-cpu-baseline --no-half --no-exact > $O/a.log 2>&1
  echo "== ${v:-shipped}"
  python3 - "$O" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + '/t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'render_fwd' in r['Name']:
            print('   kernel average %.2f ms over %s calls' % (float(r['AverageNs']) / 1e6, r['Calls']))
PY
  python3 $R/tools/pmc_summary.py $O/a render_fwd
done
