#!/bin/bash
# Runs on the GPU box (through gpurun): the bench lines and the rocprofv3 passes profiles/ is regenerated from.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_profile_round.sh'
#   python scripts/summarize_profiles.py r02 prof pmc        (back in the container)
# Kernel stats and each PMC counter group are collected in SEPARATE rocprofv3 runs (no trace domains next to --pmc), and the
# program itself follows `--` (no env/bash hop after the profiler has initialised the GPU).
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py > "$OUT/bench_c3.json" 2> "$OUT/bench_c3.err"
for c in 2 4 5; do
  python3 bench.py --config $c --no-cpu-baseline > "$OUT/bench_c$c.json" 2> "$OUT/bench_c$c.err"
done
for c in 2 3 4 5; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d "$OUT/prof_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 5 --warmup 2 --no-cpu-baseline) > "$OUT/prof_c$c.log" 2>&1
done
for c in 2 3; do
  (cd /tmp && rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline) > "$OUT/pmc_fetch_c$c.log" 2>&1
  (cd /tmp && rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write_c$c" -o runc --output-format csv -- \
      python3 "$ROOT/bench.py" --config $c --steps 2 --warmup 1 --no-cpu-baseline) > "$OUT/pmc_write_c$c.log" 2>&1
done
# the Gram kernel: matrix-pipe utilisation, wait fractions
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES \
    -d "$OUT/pmc_sq_c3" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 3 --steps 2 --warmup 1 --no-cpu-baseline) > "$OUT/pmc_sq_c3.log" 2>&1
(cd /tmp && rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU \
    -d "$OUT/pmc_sq2_c3" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 3 --steps 2 --warmup 1 --no-cpu-baseline) > "$OUT/pmc_sq2_c3.log" 2>&1
# the weight kernel: cycles (clock = GRBM_GUI_ACTIVE / 8 XCDs / duration), matrix-pipe busy cycles, vector instructions
(cd /tmp && rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU \
    -d "$OUT/pmc_kde_c3" -o runc --output-format csv -- \
    python3 "$ROOT/bench.py" --config 3 --steps 2 --warmup 1 --no-cpu-baseline) > "$OUT/pmc_kde_c3.log" 2>&1
# trim what travels back: only the stats / counter CSVs are needed
find "$OUT" -name '*_kernel_trace.csv' -size +8M -delete
ls "$OUT"/prof_c3/ "$OUT"/pmc_fetch_c3/ 2>/dev/null | head
cat "$OUT/bench_c3.json"
