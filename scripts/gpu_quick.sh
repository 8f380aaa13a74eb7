#!/bin/bash
# quick loop on the GPU box: a test selection (argument 1, a pytest -k expression; "all" = the whole GPU suite), the default
# bench line, and the kernel timeline of one generation of configs[2]
#   gpurun --timeout 900 -- 'bash scripts/gpu_quick.sh "weight or generation"'
set -u
R=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
K="${1:-all}"
if [ "$K" = "all" ]; then timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > gpurun_out/quick_tests.log 2>&1
else timeout -k 10 800 python3 -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/quick_tests.log 2>&1; fi
rc=$?; tail -5 gpurun_out/quick_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail -5 gpurun_out/quick_bench.err; exit 1; }
rm -rf gpurun_out/trace_q
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_q -o t --output-format csv -- python3 $R/scripts/trace_step.py 3 ${2:-full} 5 > $R/gpurun_out/trace_q.log 2>&1)
python3 scripts/timeline.py $(find gpurun_out/trace_q -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline_q.txt
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/quick_bench.json"))
print("step %.4f ms  kde %.4f  stream %.4f (frac %.4f)  set0 %.4f (frac %.4f)  sustained %.4f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline_streaming"]["ms"], d["roofline_streaming"]["frac"], d["set0"]["ms_per_step"], d["set0"]["roofline_streaming"]["frac"], d["sustained"]["ms_per_step"]))
print(d["stage_ms_per_step"])
PY
