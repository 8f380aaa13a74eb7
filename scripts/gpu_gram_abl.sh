#!/bin/bash
# Runs on the GPU box: k_gram_i8 with phases of its tile loop left out (ABC_GRAM_ABL: 1 no conversion, 2 no byte products, 4 no
# refills of the raw tiles), kernel statistics of each -- where a tile's time goes
#   gpurun -- 'bash scripts/gpu_gram_abl.sh "1000000 128 16" 0 1 2 3 4 7'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp ABC_DIAG=1
SHAPE="$1"; shift
mkdir -p "$OUT"
for a in "$@"; do
  export ABC_GRAM_ABL=$a
  rm -rf "$OUT/prof_gabl_$a"
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_gabl_$a" -o run --output-format csv -- python3 "$ROOT/scripts/gram_time.py" $SHAPE) > "$OUT/gabl_$a.log" 2>&1
  echo "== ABC_GRAM_ABL=$a"
  python3 scripts/kstats.py $(find "$OUT/prof_gabl_$a" -name "*kernel_stats.csv" | head -1) k_gram_i8
done
