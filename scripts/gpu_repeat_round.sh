#!/bin/bash
# Runs on the GPU box: scripts/repeat_check.py (bit-identical repeats of whole generations under the default component rule) as
# built, with the runtime serialising every launch, and with this round's orchestration switches flipped -- the hash of a shape
# must be the same in every line
#   gpurun --timeout 1200 -- 'bash scripts/gpu_repeat_round.sh 60'
set -u
export ABC_DIAG=1
R=${1:-60}
run() { echo "== $*"; env "$@" python3 scripts/repeat_check.py $R 2>&1 | grep -E "^ok|^FAIL|shapes"; }
run A=1
run AMD_SERIALIZE_KERNEL=3
run ABC_WX_FINISH_EARLY=1
run ABC_PROJECT_SEPARATE=1
run ABC_WX_L0_AFTER_GATHER=0
run ABC_WX_L0_AFTER_GATHER=1
run ABC_WX_INLINE=1
