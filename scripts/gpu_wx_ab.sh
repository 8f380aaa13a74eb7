#!/bin/bash
# Runs on the GPU box: the Wilcoxon kernels' statistics for several builds of the library (build_ab/*.so), one shape
#   gpurun -- 'bash scripts/gpu_wx_ab.sh "1000000 32 16 8" build_ab/lib_a.so build_ab/lib_b.so'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
SHAPE="$1"; shift
for SO in "$@"; do
  name=ab_$(basename $SO .so)
  rm -rf "$OUT/prof_$name"
  (cd /tmp && ABCSMC_HIP_SO="$ROOT/$SO" rocprofv3 --kernel-trace --stats -d "$OUT/prof_$name" -o run --output-format csv -- \
      python3 "$ROOT/scripts/wx_time.py" $SHAPE 5) > "$OUT/$name.log" 2>&1
  echo "== $SO: $(grep ranking $OUT/$name.log)"
  python3 scripts/kstats.py $(find "$OUT/prof_$name" -name "*kernel_stats.csv" | head -1) k_wx_sweep
done
