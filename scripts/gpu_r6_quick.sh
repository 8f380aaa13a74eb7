#!/bin/bash
# round 6 quick loop on the GPU box: a test selection (argument 1: a pytest -k expression, "none" to skip), then the bench line
# of the configs named in argument 2 (default "3 5") with the extra legs (moved-count leg included), and the cascade's debug lines
#   gpurun --timeout 1100 -- 'bash scripts/gpu_r6_quick.sh "wilcoxon or speculat" "3 5"'
set -u
R=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
K="${1:-none}"; CFGS="${2:-3 5}"
if [ "$K" != "none" ]; then
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/r6_tests.log 2>&1
  rc=$?; tail -8 gpurun_out/r6_tests.log
  [ $rc -ne 0 ] && exit $rc
fi
for c in $CFGS; do
  timeout -k 10 400 python3 bench.py --config $c --no-cpu-baseline > gpurun_out/r6_bench_$c.json 2> gpurun_out/r6_bench_$c.err || { tail -5 gpurun_out/r6_bench_$c.err; exit 1; }
  python3 - $c <<'PY'
import json, sys
c = sys.argv[1]
d = json.loads(open("gpurun_out/r6_bench_%s.json" % c).read().strip().splitlines()[-1])
rs = d["roofline_streaming"]
print("config %s: step %.4f ms  kde %.4f  stream %.4f (frac %.4f / as written %.4f)  set0 %.4f  sustained %.4f  press step %s" % (
    c, d["ms_per_step"], d["roofline"]["kernel_ms"], rs["ms"], rs["frac"], rs.get("frac_without_the_rules_pass", 0), d["set0"]["ms_per_step"],
    d["sustained"]["ms_per_step"], d["extra"].get("min_press_rule_step_ms")))
print("   moved:", json.dumps(d["extra"].get("moved_count")))
PY
  ABC_DIAG=1 ABC_WX_DEBUG=1 timeout -k 10 200 python3 bench.py --config $c --no-cpu-baseline --no-extra --steps 3 --warmup 1 2>&1 >/dev/null | grep WX_DEBUG | sort | uniq -c
done
