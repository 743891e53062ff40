#!/bin/bash
# Duration, cycles and clock of wgrad for the shipped build and variants: tools/pmc_wgrad.sh "" nomfma ...
R=/root/repo
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ -n "$v" ]; then export SUNERF_HIP_LIB=$R/build_var/libsunerf_hip_$v.so; else unset SUNERF_HIP_LIB; fi
  O=$R/gpurun_out/pmcw/${v:-shipped}
  rm -rf $O; mkdir -p $O
  ex="--no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-dt --steps 3 --warmup 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py $ex > $O/t.log 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d $O/a -- python3 $R/bench.py $ex > $O/a.log 2>&1
  echo "== ${v:-shipped}"
  python3 - "$O" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + '/t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'wgrad_kernel' in r['Name'] or 'dgrad' in r['Name']:
            print('   %-22s average %.3f ms over %s calls' % (r['Name'][27:48], float(r['AverageNs']) / 1e6, r['Calls']))
PY
  python3 $R/tools/pmc_summary.py $O/a wgrad_kernel
done
