#!/bin/bash
# round 6: scripts/moved_time.py under a list of environment settings, alternating   gpurun -- 'bash scripts/gpu_r6_moved_ab.sh 3 2 "A=1" "ABC_X=1"'
set -u
export TMPDIR=/tmp ABC_DIAG=1
CFG="$1"; REPS="$2"; shift 2
for r in $(seq 1 $REPS); do for e in "$@"; do echo -n "$e  "; env $e python3 scripts/moved_time.py $CFG 10 2>/dev/null | tail -1; done; done
