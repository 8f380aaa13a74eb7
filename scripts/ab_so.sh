#!/bin/bash
# A/B of two builds of the library on ONE box: bench.py REPS times each, interleaved; prints step, pair sums, the rest, set 0 and one stage
#   gpurun -- 'bash scripts/ab_so.sh 3 perturb gpurun_out/ab/lib_old.so gpurun_out/ab/lib_new.so'
REPS="$1"; STAGE="$2"; shift; shift
for r in $(seq 1 $REPS); do
  for SO in "$@"; do
    ABCSMC_HIP_SO="$PWD/$SO" python3 bench.py --no-cpu-baseline --no-extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('[%-28s] step %.4f kde %.4f stream %.4f set0 %.4f $STAGE %.4f' % ('$SO'.split('/')[-1], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_streaming']['ms'], d['set0']['ms_per_step'], d['stage_ms_per_step'].get('$STAGE', 0.0)))"
  done
done
