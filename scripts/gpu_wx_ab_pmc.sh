#!/bin/bash
# Runs on the GPU box: SQ counters of the sweep kernels for several builds of the library, one shape
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; export TMPDIR=/tmp
SHAPE="$1"; shift
for SO in "$@"; do
  name=abpmc_$(basename $SO .so)
  rm -rf "$OUT/$name"
  (cd /tmp && ABCSMC_HIP_SO="$ROOT/$SO" rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
      -d "$OUT/$name" -o run --output-format csv -- python3 "$ROOT/scripts/wx_time.py" $SHAPE 3) > "$OUT/$name.log" 2>&1
  echo "== $SO"
  python3 scripts/pmc_by_kernel.py $(find "$OUT/$name" -name "*counter_collection.csv" | head -1) k_wx_sweep
done
