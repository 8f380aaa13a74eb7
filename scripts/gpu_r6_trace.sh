#!/bin/bash
# round 6: kernel timeline of one generation -- arguments: config (3), mode (full | set0 | moved | moved0), label
#   gpurun -- 'bash scripts/gpu_r6_trace.sh 3 moved'
set -u
R=$(pwd); export TMPDIR=/tmp
mkdir -p gpurun_out
C="${1:-3}"; MODE="${2:-full}"
rm -rf gpurun_out/trace_$C$MODE
(cd /tmp && ABC_DIAG=1 ABC_WX_DEBUG=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/trace_$C$MODE -o t --output-format csv -- python3 $R/scripts/trace_step.py $C $MODE 5 > $R/gpurun_out/trace_$C$MODE.log 2>&1)
grep WX_DEBUG gpurun_out/trace_$C$MODE.log | sort | uniq -c
python3 scripts/timeline.py $(find gpurun_out/trace_$C$MODE -name "*kernel_trace.csv" | head -1) > gpurun_out/timeline_${C}_$MODE.txt
tail -3 gpurun_out/timeline_${C}_$MODE.txt
