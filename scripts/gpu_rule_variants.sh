#!/bin/bash
# Runs on the GPU box: the bench line of one config under argmin PRESS, under the Wilcoxon rule, and under the rule with this round's
# orchestration changes switched off one at a time (diagnostic switches), alternating, REPS times
#   gpurun -- 'bash scripts/gpu_rule_variants.sh 3 2'
set -u
export TMPDIR=/tmp ABC_DIAG=1
CFG="$1"; REPS="${2:-2}"
mkdir -p gpurun_out
run() {   # label, rule, env assignments...
  local label="$1" rule="$2"; shift 2
  env "$@" python3 bench.py --config $CFG --rule $rule --steps 40 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/variant.json
  python3 -c "
import json; d=json.loads(open('gpurun_out/variant.json').read()); print('%-34s step %.4f ms  streaming %.4f ms' % ('$label', d['ms_per_step'], d['roofline_streaming']['ms']))"
}
for r in $(seq 1 $REPS); do
  run "press" press A=1
  run "wilcoxon" wilcoxon A=1
  run "wilcoxon, look deferred (round 5)" wilcoxon ABC_WX_DEFER=1
  run "wilcoxon, scores separate" wilcoxon ABC_PROJECT_SEPARATE=1
  run "wilcoxon, all tests at level 0" wilcoxon ABC_WX_FIRST=0
  run "wilcoxon, in stream order" wilcoxon ABC_WX_INLINE=1
done
