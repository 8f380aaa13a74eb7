#!/bin/bash
# Runs on the GPU box (through gpurun): kernel statistics of the Wilcoxon reduction at configs[2] / configs[4] / configs[3] column shapes
#   gpurun --timeout 900 -- 'bash scripts/gpu_wx_prof.sh [tag]'
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
TAG=${1:-wx}
mkdir -p "$OUT"
export TMPDIR=/tmp
export ABC_DIAG=1 ABC_WX_DEBUG=1
for shape in "1000000 32 16 8" "1000000 128 16 32" ${WX_BIG:+"10000000 64 32 8"}; do
  set -- $shape
  name=${TAG}_$1_$2_$3_$4
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o run --output-format csv -- \
      python3 "$ROOT/scripts/wx_time.py" $1 $2 $3 $4 5 dbg) > "$OUT/$name.log" 2>&1
  grep -E "WX_DEBUG|ranking" "$OUT/$name.log" | sort | uniq -c | sort -rn | head -4
  python3 scripts/kstats.py $(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1) k_wx k_pls k_gram k_project k_zstats | tee "$OUT/$name.kstats.txt"
done
