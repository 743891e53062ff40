#!/bin/bash
# End-of-round measurement set (run on the GPU box through gpurun): bench JSON lines of every mode, kernel-trace stats and
# HBM counters.  Outputs under gpurun_out/final/; copy what is to be kept into profiles/.
set -e
R=/root/repo
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
b() { name=$1; shift; python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name $(python3 -c "import json;d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]);print('%.4e %.2f ms' % (d['value'], d['ms_per_step']))")"; }
b bench_train
b bench_fwd --mode fwd
SUNERF_FORWARD_PRECISION=exact b bench_train_exact --no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact
SUNERF_FORWARD_PRECISION=exact b bench_fwd_exact --mode fwd --no-cpu-baseline
b bench_train_d512 --d-filter 512 --no-two-pass --no-small-batch --no-half --no-exact
b bench_fwd_d512 --d-filter 512 --mode fwd
SUNERF_FORWARD_PRECISION=exact b bench_train_d512_exact --d-filter 512 --no-two-pass --no-small-batch --no-half --no-cpu-baseline --no-exact
SUNERF_FORWARD_PRECISION=exact b bench_fwd_d512_exact --d-filter 512 --mode fwd --no-cpu-baseline
bash $R/tools/profile_round.sh
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d512_stats -- python3 $R/bench.py --d-filter 512 --no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact --steps 3 --warmup 1 --mode train > $O/d512_stats.log 2>&1
bash $R/tools/profile_d512_traffic.sh > $O/d512_traffic.log 2>&1
echo done
