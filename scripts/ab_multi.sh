#!/bin/bash
# A/B/C... of several environment settings on ONE box (kernel clocks differ from box to box): bench.py REPS times each, interleaved
#   gpurun -- 'bash scripts/ab_multi.sh 2 "VAR=1" "VAR2=1 VAR3=1" "" -- [bench args]'      ("" = no setting)
REPS="$1"; shift
VARS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do VARS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for r in $(seq 1 $REPS); do
  for E in "${VARS[@]}"; do
    env $E python3 bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%-48s] step %.4f kde %.4f stream %.4f set0 %.4f sustained %.4f' % ('$E', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_streaming']['ms'], d['set0']['ms_per_step'], (d.get('sustained') or {}).get('ms_per_step', 0.0)))"
  done
done
