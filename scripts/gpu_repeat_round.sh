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
# round 6's switches: round 5's deferred look, every test at level 0 (no "largest count first"), only the validation rows' scores
# kept, the fine levels' bins by the number of tests, the scores in a pass of their own, the reduction in stream order
run ABC_WX_DEFER=1
run ABC_WX_FIRST=0
run ABC_WX_FIRST=8
run ABC_SCORES_VALID_ONLY=1
run ABC_WX_BINS_BY_COUNT=1
run ABC_PROJECT_SEPARATE=1
run ABC_WX_INLINE=1
