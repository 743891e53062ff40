#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the d_filter = 512 training step (separate --pmc passes; run on the GPU box through gpurun)
set -e
R=/root/repo
O=$R/gpurun_out/d512_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
extra="--d-filter 512 --no-cpu-baseline --no-two-pass --no-small-batch --no-half --no-exact --steps 2 --warmup 1 --mode train"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $extra > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $extra > $O/write.log 2>&1
python3 $R/tools/hbm_traffic.py $O/hbm_traffic_d512.json train 32768 128 512 $O/fetch $O/write
cat $O/hbm_traffic_d512.json
