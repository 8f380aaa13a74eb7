#!/bin/bash
# round 6: the bench line of one config under a list of environment settings (diagnostic switches), alternating, REPS times
#   gpurun -- 'bash scripts/gpu_r6_envab.sh 5 2 "A=1" "ABC_WX_FIRST=3" "ABC_WX_FIRST=4"'
set -u
export TMPDIR=/tmp ABC_DIAG=1
CFG="$1"; REPS="$2"; shift 2
mkdir -p gpurun_out
for r in $(seq 1 $REPS); do
  for e in "$@"; do
    env $e python3 bench.py --config $CFG --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/envab.json
    python3 -c "
import json; d=json.loads(open('gpurun_out/envab.json').read()); print('%-40s step %.4f ms  streaming %.4f ms  set0 %.4f' % ('$e', d['ms_per_step'], d['roofline_streaming']['ms'], d['set0']['ms_per_step']))"
  done
done
