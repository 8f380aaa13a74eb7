#!/bin/bash
# Runs on the GPU box (through gpurun): the randomised differential tests against the oracle on the final build of a round
#   gpurun --timeout 1200 -- 'bash scripts/gpu_fuzz_round.sh [part]'        part a: wilcoxon + wide gram + ranking; b: generations (+ large); c: sharded, weights, resample
set -u
OUT=gpurun_out
PART=${1:-abc}
mkdir -p $OUT
if [[ $PART == *a* ]]; then
python3 tests/fuzz/wilcoxon_fuzz.py $OUT/r05_wilcoxon_fuzz.json 100 505 > $OUT/fuzz_wilcoxon.log 2>&1; tail -2 $OUT/fuzz_wilcoxon.log
python3 tests/fuzz/wide_gram_fuzz.py $OUT/r05_wide_gram_fuzz.json 40 506 > $OUT/fuzz_wide_gram.log 2>&1; tail -1 $OUT/fuzz_wide_gram.log
python3 tests/fuzz/ranking_fuzz.py $OUT/r05_ranking_fuzz.json 150 507 > $OUT/fuzz_ranking.log 2>&1; tail -1 $OUT/fuzz_ranking.log
fi
if [[ $PART == *b* ]]; then
python3 tests/fuzz/generation_fuzz.py $OUT/r05_generation_fuzz.json 200 508 > $OUT/fuzz_generation.log 2>&1; tail -1 $OUT/fuzz_generation.log
FUZZ_LARGE=1 python3 tests/fuzz/generation_fuzz.py $OUT/r05_generation_fuzz_large.json 40 509 > $OUT/fuzz_generation_large.log 2>&1; tail -1 $OUT/fuzz_generation_large.log
fi
if [[ $PART == *c* ]]; then
python3 tests/fuzz/sharded_fuzz.py $OUT/r05_sharded_fuzz.json 24 510 > $OUT/fuzz_sharded.log 2>&1; tail -1 $OUT/fuzz_sharded.log
python3 tests/fuzz/weights_fuzz.py $OUT/r05_weights_fuzz.json 200 511 > $OUT/fuzz_weights.log 2>&1; tail -1 $OUT/fuzz_weights.log
python3 tests/fuzz/resample_fuzz.py $OUT/r05_resample_fuzz.json 150 512 > $OUT/fuzz_resample.log 2>&1; tail -1 $OUT/fuzz_resample.log
fi
