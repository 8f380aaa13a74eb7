#!/bin/bash
# Runs on the GPU box (through gpurun): SQ counters of the Wilcoxon kernels (two passes, counters only -- no trace domains)
#   gpurun --timeout 900 -- 'bash scripts/gpu_wx_pmc.sh "1000000 128 16 32" tag'
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
SHAPE=${1:-"1000000 128 16 32"}
TAG=${2:-pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set -d "$OUT/pmc_${TAG}_$i" -o run --output-format csv -- python3 "$ROOT/scripts/wx_time.py" $SHAPE 3) > "$OUT/pmc_${TAG}_$i.log" 2>&1
  python3 scripts/pmc_by_kernel.py $(find "$OUT/pmc_${TAG}_$i" -name "*counter_collection.csv" | head -1) k_wx
done
