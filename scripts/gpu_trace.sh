#!/bin/bash
# kernel timeline of one generation of a bench configuration under rocprofv3 --kernel-trace (no tests, no bench)
#   gpurun -- 'bash scripts/gpu_trace.sh [config 3] [full|set0] [tag]'
R=$(pwd); export TMPDIR=/tmp
CFG="${1:-3}"; MODE="${2:-full}"; TAG="${3:-t}"
rm -rf gpurun_out/trace_$TAG
(cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$TAG -o t --output-format csv -- python3 $R/scripts/trace_step.py $CFG $MODE 5 > $R/gpurun_out/trace_$TAG.log 2>&1)
python3 scripts/timeline.py $(find gpurun_out/trace_$TAG -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline_$TAG.txt
cat gpurun_out/timeline_$TAG.txt
