#!/bin/bash
# Runs on the GPU box: the bench line of one config without / with a diagnostic switch, alternating, REPS times each
#   gpurun -- 'bash scripts/gpu_env_ab_bench.sh ABC_PROJECT_SEPARATE 3 3'
set -u
export TMPDIR=/tmp ABC_DIAG=1
VAR="$1"; CFG="$2"; REPS="${3:-3}"
mkdir -p gpurun_out
for r in $(seq 1 $REPS); do
  for v in off on; do
    if [ $v = on ]; then export "$VAR"=1; else unset "$VAR"; fi
    python3 bench.py --config $CFG --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/envab_bench.json
    python3 -c "
import json; d=json.loads(open('gpurun_out/envab_bench.json').read()); print('$VAR $v: step %.4f ms  streaming %.4f ms frac %.4f' % (d['ms_per_step'], d['roofline_streaming']['ms'], d['roofline_streaming']['frac']))"
  done
done
