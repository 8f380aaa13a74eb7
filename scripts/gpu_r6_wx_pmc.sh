#!/bin/bash
# round 6: SQ counters of the cascade's sweeps on the moved-count set of a config (levels 0 and fine side by side)
#   gpurun --timeout 900 -- 'bash scripts/gpu_r6_wx_pmc.sh 3'
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; C=${1:-3}
mkdir -p "$OUT"; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $set -d "$OUT/pmc_wxm_$i" -o run --output-format csv -- python3 "$ROOT/scripts/trace_step.py" $C moved 2) > "$OUT/pmc_wxm_$i.log" 2>&1
  python3 scripts/pmc_by_kernel.py $(find "$OUT/pmc_wxm_$i" -name "*counter_collection.csv" | head -1) k_wx_sweep
done
