#!/bin/bash
# Copies the end-of-round measurement set (tools/final_profiles.sh, merged back under gpurun_out/) into profiles/<round>_*:
#   tools/copy_profiles.sh r4
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
tag=${1:?round tag, e.g. r4}
F=$R/gpurun_out/final
P=$R/gpurun_out/prof_round
for f in $F/bench_*.json; do cp $f $R/profiles/${tag}_$(basename $f); done
for mode in train fwd; do
  cp $(ls $P/${mode}_stats/*/*_kernel_stats.csv | head -1) $R/profiles/${tag}_${mode}_kernel_stats.csv
  python3 $R/tools/kernel_medians.py $P/${mode}_stats > $R/profiles/${tag}_${mode}_kernel_medians.txt
done
python3 $R/tools/pmc_summary.py $P/train_sq > $R/profiles/${tag}_train_pmc_sq.txt
cp $R/gpurun_out/hbm_traffic.json $R/profiles/hbm_traffic.json
if ls $F/d512_stats/*/*_kernel_stats.csv > /dev/null 2>&1; then cp $(ls $F/d512_stats/*/*_kernel_stats.csv | head -1) $R/profiles/${tag}_d512_train_kernel_stats.csv; fi
[ -f $R/gpurun_out/hbm_traffic_d512.json ] && cp $R/gpurun_out/hbm_traffic_d512.json $R/profiles/hbm_traffic_d512.json
ls -la $R/profiles | grep ${tag}_
