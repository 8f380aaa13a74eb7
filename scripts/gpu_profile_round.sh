#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines and the rocprofv3 passes profiles/ is regenerated from.
#   gpurun --timeout 1200 -- 'bash scripts/gpu_profile_round.sh a'      bench lines, kernel statistics, timelines
#   gpurun --timeout 1200 -- 'bash scripts/gpu_profile_round.sh b'      the PMC passes
#   python scripts/summarize_profiles.py r04 prof pmc        (back in the container)
# Kernel stats and each PMC counter group are collected in SEPARATE rocprofv3 runs (no trace domains next to --pmc), and the
# program itself follows `--` (no env/bash hop after the profiler has initialised the GPU).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
PART=${1:-ab}
if [[ $PART == *a* ]]; then
# (configs 4 / 5 = BASELINE configs[3] / configs[4] at their STATED totals on this one GPU: 1e7 / 1e6 particles)
python3 bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
python3 bench.py --config 2 --no-cpu-baseline > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.err"
python3 bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --sustained-s 1.0 > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.err"
python3 bench.py --config 5 --no-cpu-baseline > "$OUT/bench_c5.json" 2> "$OUT/bench_c5.err"
for c in 2 3 5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-extra) > "$OUT/prof_c$c.log" 2>&1
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_c4" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/prof_c4.log" 2>&1
# (round 5: the timed region runs the drop-in's default rule, the Wilcoxon reduction -- the k_wx_* kernels are in prof_c*)
# the same generations under plain argmin PRESS (configs[2] and configs[4])
for c in 3 5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_p$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --rule press --steps 5 --warmup 2 --no-cpu-baseline --no-extra) > "$OUT/prof_p$c.log" 2>&1
done
# how the cascade settled the tests at the three column shapes (ABC_WX_DEBUG; not under the profiler: the report synchronises)
for shape in "1000000 32 16 8" "1000000 128 16 32" "10000000 64 32 8"; do
  ABC_DIAG=1 ABC_WX_DEBUG=1 python3 scripts/wx_time.py $shape 3 dbg 2>&1 | grep -E "WX_DEBUG|ranking" | sort | uniq -c | sort -rn | head -3
done > "$OUT/wx_debug.txt" 2>&1
# the kernel timeline of one generation: configs[2] weighted, first set, and with a component count the rule lowers (weighted and
# first set: bench.moved_count_data); configs[4] and configs[3] weighted and moved
for cm in "3 full" "3 set0" "3 moved" "3 moved0" "5 full" "5 moved" "4 full"; do
  set -- $cm; c=$1; m=$2
  (cd /tmp && rocprofv3 --kernel-trace -d "$OUT/trace_${c}_$m" -o t --output-format csv -- python3 "$ROOT/scripts/trace_step.py" $c $m 5) > "$OUT/trace_${c}_$m.log" 2>&1
  python3 scripts/timeline.py $(find "$OUT/trace_${c}_$m" -name "*kernel_trace.csv" | head -1) > "$OUT/timeline_${c}_$m.txt"
done
# how the cascade went, level by level (not under the profiler)
bash scripts/gpu_r6_levels.sh > "$OUT/wx_levels.txt" 2>&1
fi
if [[ $PART == *b* ]]; then
for c in 2 3; do
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_fetch_c$c.log" 2>&1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_write_c$c.log" 2>&1
done
# the Gram kernel: matrix-pipe utilisation, wait fractions
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES \
    -d "$OUT/pmc_sq_c3" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_sq_c3.log" 2>&1
(cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
    -d "$OUT/pmc_sq2_c3" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_sq2_c3.log" 2>&1
# the weight kernel: cycles (clock = GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy cycles, vector instructions
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
    -d "$OUT/pmc_kde_c3" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 3 --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_kde_c3.log" 2>&1
# the fp64 / i8 matrix-pipe kernels of the ranking at configs[3] (k_gram_dma8: 96 columns) and configs[4] (k_pilot_scale, k_gram_i8,
# k_gram_far: 144 columns; k_project_mfma: 32 components): matrix-pipe busy cycles against the kernel's cycles, vector instructions
for c in 4 5; do
  (cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
      -d "$OUT/pmc_mfma_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_mfma_c$c.log" 2>&1
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_fetch_c$c.log" 2>&1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_write_c$c.log" 2>&1
done
# the fp64 Gram at configs[4] for comparison (ABC_GRAM_FP64 is a diagnostic switch: ABC_DIAG=1 opens it)
ABC_DIAG=1 ABC_GRAM_FP64=1 python3 bench.py --config 5 --no-cpu-baseline --no-extra > "$OUT/bench_c5_fp64gram.json" 2> "$OUT/bench_c5_fp64gram.err"
export ABC_DIAG=1 ABC_GRAM_FP64=1
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
    -d "$OUT/pmc_mfma_c5f" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-extra) > "$OUT/pmc_mfma_c5f.log" 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_c5f" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-extra) > "$OUT/prof_c5f.log" 2>&1
unset ABC_DIAG ABC_GRAM_FP64
fi
# trim what travels back: only the stats / counter CSVs are needed
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
ls "$OUT"/prof_c3/ "$OUT"/pmc_fetch_c3/ 2>/dev/null | head
cat "$OUT/bench_c3.json"
