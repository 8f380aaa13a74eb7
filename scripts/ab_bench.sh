#!/bin/bash
# A/B of two environment settings on ONE box (kernel clocks differ from box to box): bench.py twice each, interleaved
#   gpurun -- 'bash scripts/ab_bench.sh "VAR=1" "VAR2=1" [bench args]'
A="$1"; B="$2"; shift 2
for r in 1 2; do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    env $E python3 bench.py --no-cpu-baseline --no-extra "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag [$E] step %.4f kde %.4f stream %.4f set0 %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline_streaming']['ms'], d['set0']['ms_per_step']))"
  done
done
