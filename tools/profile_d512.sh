set -e
R=/root/repo
O=$R/gpurun_out/d512
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py --d-filter 512 --no-two-pass --mode train > $O/train.json 2> $O/train.err
python3 $R/bench.py --d-filter 512 --no-two-pass --mode fwd > $O/fwd.json 2> $O/fwd.err
SUNERF_FORWARD_PRECISION=exact python3 $R/bench.py --d-filter 512 --no-two-pass --no-cpu-baseline --mode train > $O/train_exact.json 2> $O/train_exact.err
SUNERF_FORWARD_PRECISION=exact python3 $R/bench.py --d-filter 512 --no-two-pass --no-cpu-baseline --mode fwd > $O/fwd_exact.json 2> $O/fwd_exact.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --d-filter 512 --no-cpu-baseline --no-two-pass --no-half --steps 3 --warmup 1 --mode train > $O/stats.log 2>&1
echo done
